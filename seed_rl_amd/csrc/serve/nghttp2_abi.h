// The part of libnghttp2's public C API (nghttp2/nghttp2.h, v1.x ABI "14") that seedserve.cpp uses, declared here
// because this image ships the shared library (libnghttp2.so.14, a curl dependency) without its development header.
// Types and constants are restated from the library's published interface; the functions are resolved with dlsym at
// start-up (a missing library or symbol fails seedserve_create loudly).  nghttp2 does HTTP/2 framing, HPACK and flow
// control; everything gRPC (message framing, status trailers, method routing) is in seedserve.cpp.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <sys/types.h>

extern "C" {
struct nghttp2_session;
struct nghttp2_session_callbacks;
struct nghttp2_option;

struct nghttp2_frame_hd {          // first member of every member of the nghttp2_frame union
  size_t length;
  int32_t stream_id;
  uint8_t type;
  uint8_t flags;
  uint8_t reserved;
};
struct nghttp2_nv {
  uint8_t* name;
  uint8_t* value;
  size_t namelen;
  size_t valuelen;
  uint8_t flags;
};
struct nghttp2_settings_entry {
  int32_t settings_id;
  uint32_t value;
};
union nghttp2_data_source {
  int fd;
  void* ptr;
};
typedef ssize_t (*nghttp2_data_source_read_callback)(nghttp2_session* session, int32_t stream_id, uint8_t* buf,
                                                     size_t length, uint32_t* data_flags, nghttp2_data_source* source,
                                                     void* user_data);
struct nghttp2_data_provider {
  nghttp2_data_source source;
  nghttp2_data_source_read_callback read_callback;
};

enum {
  NGHTTP2_DATA = 0, NGHTTP2_HEADERS = 1, NGHTTP2_RST_STREAM = 3, NGHTTP2_SETTINGS = 4, NGHTTP2_GOAWAY = 7,
  NGHTTP2_FLAG_NONE = 0, NGHTTP2_FLAG_END_STREAM = 0x01, NGHTTP2_FLAG_END_HEADERS = 0x04,
  NGHTTP2_NV_FLAG_NONE = 0,
  NGHTTP2_SETTINGS_MAX_CONCURRENT_STREAMS = 3, NGHTTP2_SETTINGS_INITIAL_WINDOW_SIZE = 4, NGHTTP2_SETTINGS_MAX_FRAME_SIZE = 5,
  NGHTTP2_DATA_FLAG_NONE = 0, NGHTTP2_DATA_FLAG_EOF = 0x01, NGHTTP2_DATA_FLAG_NO_END_STREAM = 0x02,
  NGHTTP2_NO_ERROR = 0, NGHTTP2_INTERNAL_ERROR = 2,
  NGHTTP2_ERR_DEFERRED = -508, NGHTTP2_ERR_TEMPORAL_CALLBACK_FAILURE = -521, NGHTTP2_ERR_CALLBACK_FAILURE = -902
};

typedef int (*nghttp2_on_frame_recv_callback)(nghttp2_session*, const void* frame, void* user_data);
typedef int (*nghttp2_on_data_chunk_recv_callback)(nghttp2_session*, uint8_t flags, int32_t stream_id,
                                                   const uint8_t* data, size_t len, void* user_data);
typedef int (*nghttp2_on_stream_close_callback)(nghttp2_session*, int32_t stream_id, uint32_t error_code,
                                                void* user_data);
typedef int (*nghttp2_on_begin_headers_callback)(nghttp2_session*, const void* frame, void* user_data);
typedef int (*nghttp2_on_header_callback)(nghttp2_session*, const void* frame, const uint8_t* name, size_t namelen,
                                          const uint8_t* value, size_t valuelen, uint8_t flags, void* user_data);
}  // extern "C"

// name, return type, argument list
#define SEEDSERVE_NGHTTP2_FUNCS(X)                                                                                      \
  X(nghttp2_session_callbacks_new, int, (nghttp2_session_callbacks**))                                                  \
  X(nghttp2_session_callbacks_del, void, (nghttp2_session_callbacks*))                                                  \
  X(nghttp2_session_callbacks_set_on_frame_recv_callback, void, (nghttp2_session_callbacks*, nghttp2_on_frame_recv_callback)) \
  X(nghttp2_session_callbacks_set_on_data_chunk_recv_callback, void, (nghttp2_session_callbacks*, nghttp2_on_data_chunk_recv_callback)) \
  X(nghttp2_session_callbacks_set_on_stream_close_callback, void, (nghttp2_session_callbacks*, nghttp2_on_stream_close_callback)) \
  X(nghttp2_session_callbacks_set_on_begin_headers_callback, void, (nghttp2_session_callbacks*, nghttp2_on_begin_headers_callback)) \
  X(nghttp2_session_callbacks_set_on_header_callback, void, (nghttp2_session_callbacks*, nghttp2_on_header_callback))   \
  X(nghttp2_session_server_new, int, (nghttp2_session**, const nghttp2_session_callbacks*, void*))                      \
  X(nghttp2_session_del, void, (nghttp2_session*))                                                                      \
  X(nghttp2_session_mem_recv, ssize_t, (nghttp2_session*, const uint8_t*, size_t))                                      \
  X(nghttp2_session_mem_send, ssize_t, (nghttp2_session*, const uint8_t**))                                             \
  X(nghttp2_session_want_read, int, (nghttp2_session*))                                                                 \
  X(nghttp2_session_want_write, int, (nghttp2_session*))                                                                \
  X(nghttp2_submit_settings, int, (nghttp2_session*, uint8_t, const nghttp2_settings_entry*, size_t))                   \
  X(nghttp2_submit_response, int, (nghttp2_session*, int32_t, const nghttp2_nv*, size_t, const nghttp2_data_provider*)) \
  X(nghttp2_submit_trailer, int, (nghttp2_session*, int32_t, const nghttp2_nv*, size_t))                               \
  X(nghttp2_submit_rst_stream, int, (nghttp2_session*, uint8_t, int32_t, uint32_t))                                     \
  X(nghttp2_session_resume_data, int, (nghttp2_session*, int32_t))                                                      \
  X(nghttp2_session_set_local_window_size, int, (nghttp2_session*, uint8_t, int32_t, int32_t))                          \
  X(nghttp2_session_terminate_session, int, (nghttp2_session*, uint32_t))                                               \
  X(nghttp2_strerror, const char*, (int))

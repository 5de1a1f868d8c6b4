// libseedserve.so -- native gRPC front-end of the learner (include/seedserve.h).
//
// What the reference does with the gRPC C++ library + TensorFlow ops (/root/reference/grpc/ops/grpc.cc) is done here
// with epoll, libnghttp2 (HTTP/2 framing / HPACK / flow control only) and ~900 lines of C++:
//   TensorHandler::Init / ::Call         grpc.cc:141-233   -> handle_init / handle_call
//   completion-queue server threads      grpc.cc:366-475   -> IoThread (one epoll loop per thread, connections pinned)
//   verify_args / GetArgBatchSize        grpc.cc:527-589   -> verify_args (same error strings)
//   DynamicFn (server-side batching)     grpc.cc:591-861   -> Fn::place: rows are reserved under a lock, tensor bytes
//                                                             are copied OUTSIDE it straight from the message into
//                                                             the caller's pinned batch buffer
//   round-robin over functions bound     grpc.cc:193-205   -> Bucket::counter
//   under one name
// Wire format: grpc/service.proto:28-57 (CallRequest{function=1, tensor=2 repeated bytes}, CallResponse{tensor=1,
// status_code=2, status_error_message=3}) carrying serialized tensorflow.TensorProto (dtype=1, tensor_shape=2{dim=2
// {size=1}}, tensor_content=4, typed *_val fields with TF's "last value repeats" rule).
#include <arpa/inet.h>
#include <dlfcn.h>
#include <errno.h>
#include <fcntl.h>
#include <netdb.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <sys/epoll.h>
#include <sys/eventfd.h>
#include <sys/socket.h>
#include <sys/stat.h>
#include <sys/un.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../../include/seedserve.h"
#include "nghttp2_abi.h"

struct seedserve_server;

namespace {
using Server = ::seedserve_server;

// ---- errors ---------------------------------------------------------------------------------------------------- //
thread_local char g_err[512];
int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

// ---- libnghttp2, resolved at run time ---------------------------------------------------------------------------- //
struct Ng {
#define X(name, ret, args) ret(*name) args = nullptr;
  SEEDSERVE_NGHTTP2_FUNCS(X)
#undef X
  void* handle = nullptr;
  bool load(std::string* why) {
    for (const char* so : {"libnghttp2.so.14", "libnghttp2.so"}) {
      handle = dlopen(so, RTLD_NOW | RTLD_LOCAL);
      if (handle) break;
    }
    if (!handle) { *why = "cannot load libnghttp2.so.14 (HTTP/2 framing): " + std::string(dlerror()); return false; }
#define X(name, ret, args)                                          \
    name = reinterpret_cast<ret(*) args>(dlsym(handle, #name));     \
    if (!name) { *why = "libnghttp2 lacks " #name; return false; }
    SEEDSERVE_NGHTTP2_FUNCS(X)
#undef X
    return true;
  }
};
std::string g_ng_why;
Ng* ng() {
  static Ng* inst = [] {
    Ng* n = new Ng;
    if (!n->load(&g_ng_why)) { delete n; n = nullptr; }
    return n;
  }();
  return inst;
}

// ---- tensorflow DataType (types.proto) ---------------------------------------------------------------------------- //
enum { DT_FLOAT = 1, DT_DOUBLE = 2, DT_INT32 = 3, DT_UINT8 = 4, DT_INT16 = 5, DT_INT8 = 6, DT_STRING = 7, DT_INT64 = 9,
       DT_BOOL = 10, DT_UINT16 = 17, DT_HALF = 19, DT_UINT32 = 22, DT_UINT64 = 23 };
int dtype_size(int dt) {
  switch (dt) {
    case DT_FLOAT: case DT_INT32: case DT_UINT32: return 4;
    case DT_DOUBLE: case DT_INT64: case DT_UINT64: return 8;
    case DT_UINT8: case DT_INT8: case DT_BOOL: return 1;
    case DT_INT16: case DT_UINT16: case DT_HALF: return 2;
    default: return 0;
  }
}
const char* dtype_name(int dt) {                           // DataTypeString()
  switch (dt) {
    case DT_FLOAT: return "float"; case DT_DOUBLE: return "double"; case DT_INT32: return "int32";
    case DT_UINT8: return "uint8"; case DT_INT16: return "int16"; case DT_INT8: return "int8";
    case DT_STRING: return "string"; case DT_INT64: return "int64"; case DT_BOOL: return "bool";
    case DT_UINT16: return "uint16"; case DT_HALF: return "half"; case DT_UINT32: return "uint32";
    case DT_UINT64: return "uint64"; default: return "unknown";
  }
}
// tensorflow.error.Code == grpc status codes
enum { OK = 0, CANCELLED = 1, INVALID_ARGUMENT = 3, UNIMPLEMENTED = 12, INTERNAL = 13, UNAVAILABLE = 14 };

// ---- protobuf wire helpers ---------------------------------------------------------------------------------------- //
struct Reader {
  const uint8_t* p; const uint8_t* end; bool ok = true;
  Reader(const uint8_t* b, size_t n) : p(b), end(b + n) {}
  bool done() const { return p >= end || !ok; }
  uint64_t varint() {
    uint64_t v = 0; int s = 0;
    while (p < end && s < 64) {
      const uint8_t b = *p++;
      v |= (uint64_t)(b & 0x7F) << s;
      if (!(b & 0x80)) return v;
      s += 7;
    }
    ok = false; return 0;
  }
  bool bytes(const uint8_t** b, size_t* n) {
    const uint64_t len = varint();
    if (!ok || len > (uint64_t)(end - p)) { ok = false; return false; }
    *b = p; *n = (size_t)len; p += len; return true;
  }
  void skip(int wire) {
    const uint8_t* b; size_t n;
    switch (wire) {
      case 0: varint(); break;
      case 1: if (end - p < 8) ok = false; else p += 8; break;
      case 2: bytes(&b, &n); break;
      case 5: if (end - p < 4) ok = false; else p += 4; break;
      default: ok = false;
    }
  }
};
void put_varint(std::string* s, uint64_t v) {
  while (v >= 0x80) { s->push_back((char)(v | 0x80)); v >>= 7; }
  s->push_back((char)v);
}
void put_tag(std::string* s, int field, int wire) { put_varint(s, (uint64_t)(field << 3 | wire)); }
void put_bytes_field(std::string* s, int field, const void* b, size_t n) {
  put_tag(s, field, 2); put_varint(s, n); s->append((const char*)b, n);
}

struct TensorView {                                         // a parsed TensorProto, pointing into the message
  int dtype = 0; int rank = 0; int64_t dims[SEEDSERVE_MAX_RANK] = {0};
  bool rank_overflow = false;
  const uint8_t* content = nullptr; size_t content_len = 0;
  const uint8_t* vals = nullptr; size_t vals_len = 0; int vals_field = 0, vals_wire = 0;   // first typed *_val run
  int64_t elems() const { int64_t n = 1; for (int i = 0; i < rank; ++i) n *= dims[i]; return n; }
};
bool parse_shape(const uint8_t* b, size_t n, TensorView* t) {
  Reader r(b, n);
  while (!r.done()) {
    const uint64_t tag = r.varint();
    if (!r.ok) return false;
    if ((tag >> 3) == 2 && (tag & 7) == 2) {                // Dim
      const uint8_t* d; size_t dn;
      if (!r.bytes(&d, &dn)) return false;
      Reader dr(d, dn);
      int64_t size = 0;
      while (!dr.done()) {
        const uint64_t dt = dr.varint();
        if (!dr.ok) return false;
        if ((dt >> 3) == 1 && (dt & 7) == 0) size = (int64_t)dr.varint(); else dr.skip((int)(dt & 7));
      }
      if (!dr.ok) return false;
      if (t->rank < SEEDSERVE_MAX_RANK) t->dims[t->rank++] = size; else t->rank_overflow = true;
    } else {
      r.skip((int)(tag & 7));
    }
  }
  return r.ok;
}
bool parse_tensor(const uint8_t* b, size_t n, TensorView* t) {
  Reader r(b, n);
  while (!r.done()) {
    const uint64_t tag = r.varint();
    if (!r.ok) return false;
    const int field = (int)(tag >> 3), wire = (int)(tag & 7);
    const uint8_t* d; size_t dn;
    if (field == 1 && wire == 0) t->dtype = (int)r.varint();
    else if (field == 2 && wire == 2) { if (!r.bytes(&d, &dn) || !parse_shape(d, dn, t)) return false; }
    else if (field == 4 && wire == 2) { if (!r.bytes(&d, &dn)) return false; t->content = d; t->content_len = dn; }
    else if ((field == 5 || field == 6 || field == 7 || field == 10 || field == 11 || field == 13 || field == 16 ||
              field == 17) && wire == 2 && !t->vals) {      // packed typed values
      if (!r.bytes(&d, &dn)) return false;
      t->vals = d; t->vals_len = dn; t->vals_field = field; t->vals_wire = 2;
    } else r.skip(wire);
  }
  return r.ok;
}
std::string shape_str(const int64_t* dims, int rank) {     // TensorShape::DebugString()
  std::string s = "[";
  for (int i = 0; i < rank; ++i) { if (i) s += ","; s += std::to_string((long long)dims[i]); }
  return s + "]";
}

// ---- server objects ------------------------------------------------------------------------------------------------ //
struct IoThread;

struct Target { int io; uint64_t conn; int32_t stream; };  // where a response goes

struct Spec {
  int dtype, rank; int64_t dims[SEEDSERVE_MAX_RANK]; bool widen;
  int64_t row_elems;                                        // elements per row (dims[1:])
  int wire_size, store_size;
};

struct Pending { Target to; int start, count; bool batched; };
struct Slot {
  enum State { FREE, FILLING, READY, COMPUTING } state = FREE;
  int num_ready = 0;
  std::vector<Pending> pend;
};
struct Request {                                            // a verified call: views into the message bytes
  std::shared_ptr<std::string> keep;                        // set only when the call must wait for a free slot
  std::vector<TensorView> args; Target to; int count; bool batched; bool direct;
  const uint8_t* base = nullptr; size_t len = 0;
  void own() {                                              // copy the message once and re-point the views into the copy
    if (keep) return;
    keep = std::make_shared<std::string>((const char*)base, len);
    const ptrdiff_t d = (const uint8_t*)keep->data() - base;
    for (auto& a : args) { if (a.content) a.content += d; if (a.vals) a.vals += d; }
    base = (const uint8_t*)keep->data();
  }
};

struct Fn {
  int id; std::string name; int N;
  std::vector<Spec> in, out;
  int num_slots; std::vector<void*> inbuf, outbuf;
  std::mutex mu; std::condition_variable cv;
  std::vector<Slot> slots; int cur = -1, next_index = 0;
  std::deque<int> ready; std::deque<Request> overflow;
};
struct Bucket { std::vector<Fn*> fns; std::atomic<uint64_t> counter{0}; };

struct Stream {
  int method = 0;                                           // 1 Init, 2 Call
  std::string in; std::string out; size_t out_off = 0;
  bool deferred = false, remote_closed = false, finished = false, responded = false;
  int inflight = 0;
  std::string path, grpc_encoding;
};
struct Conn {
  uint64_t id; int fd; nghttp2_session* session = nullptr; IoThread* io;
  std::unordered_map<int32_t, Stream> streams;
  std::string wbuf; size_t woff = 0; bool want_out = false, dead = false;
};
struct OutItem { uint64_t conn; int32_t stream; std::string data; };

struct IoThread;
thread_local IoThread* t_current_io = nullptr;
struct IoThread {
  Server* srv; int index; int ep = -1, wake = -1; std::thread th;
  std::mutex mu; std::vector<OutItem> outbox; std::vector<int> new_fds;
  std::unordered_map<uint64_t, Conn*> conns;
  void post(OutItem&& it) {
    bool was_empty;
    { std::lock_guard<std::mutex> l(mu); was_empty = outbox.empty() && new_fds.empty(); outbox.push_back(std::move(it)); }
    if (was_empty && t_current_io != this) { uint64_t one = 1; (void)!write(wake, &one, 8); }
  }
  void adopt(int fd) {
    { std::lock_guard<std::mutex> l(mu); new_fds.push_back(fd); }
    uint64_t one = 1; (void)!write(wake, &one, 8);
  }
  void run();
  void add_conn(int fd);
  void close_conn(Conn* c);
  void on_readable(Conn* c);
  void flush(Conn* c);
};

}  // namespace

struct seedserve_server {
  std::vector<std::unique_ptr<IoThread>> io;
  std::vector<int> listeners; std::vector<std::string> unix_paths;
  std::vector<std::unique_ptr<Fn>> fns; std::map<std::string, std::unique_ptr<Bucket>> buckets;
  std::string init_response;
  std::atomic<bool> started{false}, shutdown{false};
  std::atomic<uint64_t> next_conn{1}, rr{0};
  std::atomic<uint64_t> n_conn{0}, n_stream{0}, n_call{0}, n_batch{0}, n_in{0}, n_out{0}, n_err{0};
  nghttp2_session_callbacks* cbs = nullptr;
};

namespace {
using Server_ = ::seedserve_server;

// ---- gRPC message framing / responses ------------------------------------------------------------------------------ //
std::string frame_message(const std::string& payload) {
  std::string m;
  m.reserve(payload.size() + 5);
  m.push_back(0);
  const uint32_t n = (uint32_t)payload.size();
  m.push_back((char)(n >> 24)); m.push_back((char)(n >> 16)); m.push_back((char)(n >> 8)); m.push_back((char)n);
  m += payload;
  return m;
}
std::string error_response(int code, const std::string& msg) {     // CallResponse with a status
  std::string p;
  put_tag(&p, 2, 0); put_varint(&p, (uint64_t)code);
  put_bytes_field(&p, 3, msg.data(), msg.size());
  return frame_message(p);
}
void encode_tensor(std::string* resp, const Spec& s, const uint8_t* rows, int count, bool batched) {
  std::string shape;
  auto dim = [&](int64_t d) { std::string dm; put_tag(&dm, 1, 0); put_varint(&dm, (uint64_t)d); put_bytes_field(&shape, 2, dm.data(), dm.size()); };
  if (batched) dim(count);
  for (int i = 1; i < s.rank; ++i) dim(s.dims[i]);
  std::string t;
  put_tag(&t, 1, 0); put_varint(&t, (uint64_t)s.dtype);
  put_bytes_field(&t, 2, shape.data(), shape.size());
  put_bytes_field(&t, 4, rows, (size_t)count * s.row_elems * s.wire_size);
  put_bytes_field(resp, 1, t.data(), t.size());
}

void respond(Server_* srv, const Target& to, std::string&& framed) {
  srv->n_out += framed.size();
  srv->io[to.io]->post(OutItem{to.conn, to.stream, std::move(framed)});
}

// ---- verify_args (grpc.cc:527-557), same messages --------------------------------------------------------------------- //
std::string verify_args(const std::vector<Spec>& specs, int batching_dims, const std::vector<TensorView>& args, bool full_shape) {
  char buf[256];
  if (specs.size() != args.size()) {
    snprintf(buf, sizeof buf, "Expects %zu arguments, but %zu is provided", specs.size(), args.size());
    return buf;
  }
  for (size_t i = 0; i < args.size(); ++i) {
    const Spec& s = specs[i]; const TensorView& a = args[i];
    const int64_t* edims = full_shape ? s.dims : s.dims + 1;
    const int erank = full_shape ? s.rank : s.rank - 1;
    if (a.rank_overflow || erank + batching_dims != a.rank) {
      snprintf(buf, sizeof buf, "Expects arg[%zu] to have shape with %d dimension(s), but had shape %s", i,
               erank + batching_dims, shape_str(a.dims, a.rank).c_str());
      return buf;
    }
    bool suffix = true;
    for (int d = 0; d < erank; ++d) suffix = suffix && a.dims[a.rank - erank + d] == edims[d];
    if (!suffix) {
      snprintf(buf, sizeof buf, "Expects arg[%zu] to have shape with suffix %s, but had shape %s", i,
               shape_str(edims, erank).c_str(), shape_str(a.dims, a.rank).c_str());
      return buf;
    }
    if (s.dtype != a.dtype) {
      snprintf(buf, sizeof buf, "Expects arg[%zu] to be %s but %s is provided", i, dtype_name(s.dtype), dtype_name(a.dtype));
      return buf;
    }
  }
  return "";
}

// Dry run of copy_rows' parse: true iff copying `count` rows of this argument cannot fail.  Called BEFORE rows are
// reserved (a request whose tensor bytes are malformed -- tensor_content of the wrong length, a truncated
// *_val run -- must be answered INVALID_ARGUMENT like the reference's Tensor::FromProto failure (grpc.cc:176-182) and
// must never reach a batch: the rows it had reserved kept an EARLIER batch's bytes -- another env's observation and
// run id -- and were computed on).
bool payload_ok(const Spec& s, const TensorView& a, int count) {
  const int64_t n = (int64_t)count * s.row_elems;
  if (a.content && a.content_len == (size_t)(n * s.wire_size)) return true;
  if (a.content_len != 0) return false;
  if (!a.vals) return true;                                  // no values at all: zeros
  Reader r(a.vals, a.vals_len);
  for (int64_t i = 0; i < n && !r.done(); ++i) {
    switch (a.vals_field) {
      case 5: if (r.end - r.p < 4) return false; r.p += 4; break;
      case 6: if (r.end - r.p < 8) return false; r.p += 8; break;
      default: (void)r.varint(); if (!r.ok) return false;
    }
  }
  return true;
}

// Copies `count` rows of one argument into the batch buffer at row `start` (the ONE copy of the tensor bytes).
bool copy_rows(const Spec& s, const TensorView& a, void* base, int start, int count) {
  const int64_t n = (int64_t)count * s.row_elems;
  uint8_t* dst = (uint8_t*)base + (int64_t)start * s.row_elems * s.store_size;
  if (a.content && a.content_len == (size_t)(n * s.wire_size)) {
    if (!s.widen) { memcpy(dst, a.content, (size_t)(n * s.wire_size)); return true; }
    const uint8_t* src = a.content; int64_t* d = (int64_t*)dst;
    for (int64_t i = 0; i < n; ++i) { int32_t v; memcpy(&v, src + 4 * i, 4); d[i] = v; }
    return true;
  }
  if (a.content_len != 0) return false;                      // tensor_content of the wrong length
  // typed *_val fields (Tensor::FromProto: missing trailing values repeat the last one; none at all = zeros)
  Reader r(a.vals, a.vals_len);
  double last_f = 0; int64_t last_i = 0;
  for (int64_t i = 0; i < n; ++i) {
    if (a.vals && !r.done()) {
      switch (a.vals_field) {
        case 5: { float f; if (r.end - r.p < 4) return false; memcpy(&f, r.p, 4); r.p += 4; last_f = f; last_i = (int64_t)f; break; }
        case 6: { double f; if (r.end - r.p < 8) return false; memcpy(&f, r.p, 8); r.p += 8; last_f = f; last_i = (int64_t)f; break; }
        default: last_i = (int64_t)r.varint(); last_f = (double)last_i; if (!r.ok) return false;
      }
    }
    uint8_t* e = dst + i * s.store_size;
    switch (s.dtype) {
      case DT_FLOAT: { float f = (float)last_f; memcpy(e, &f, 4); break; }
      case DT_DOUBLE: memcpy(e, &last_f, 8); break;
      case DT_INT32: case DT_UINT32: if (s.widen) memcpy(e, &last_i, 8); else { int32_t v = (int32_t)last_i; memcpy(e, &v, 4); } break;
      case DT_INT64: case DT_UINT64: memcpy(e, &last_i, 8); break;
      case DT_INT16: case DT_UINT16: case DT_HALF: { int16_t v = (int16_t)last_i; memcpy(e, &v, 2); break; }
      default: *e = (uint8_t)last_i;
    }
  }
  return true;
}

// ---- DynamicFn (grpc.cc:591-861) ------------------------------------------------------------------------------------- //
// Reserves rows for a verified request and copies its tensors.  Returns: 1 placed (and *counted: a batch was filled /
// a direct call started), 0 queued (no free slot right now), -1 rejected (response already sent).
int place(Server_* srv, Fn* fn, Request& rq, bool* counted) {
  *counted = false;
  int slot = -1, start = 0;
  {
    std::lock_guard<std::mutex> l(fn->mu);
    if (rq.direct) {                                         // exact-shape arguments: a computation of its own
      for (int s = 0; s < fn->num_slots; ++s)
        if (fn->slots[s].state == Slot::FREE) { slot = s; break; }
      if (slot < 0) { rq.own(); fn->overflow.push_back(std::move(rq)); return 0; }
      fn->slots[slot].state = Slot::FILLING;
    } else {
      if (fn->cur < 0) {
        for (int s = 0; s < fn->num_slots; ++s)
          if (fn->slots[s].state == Slot::FREE) { fn->cur = s; break; }
        if (fn->cur < 0) { rq.own(); fn->overflow.push_back(std::move(rq)); return 0; }
        fn->slots[fn->cur].state = Slot::FILLING; fn->next_index = 0;
      }
      if (fn->next_index + rq.count > fn->N) {               // a CHECK failure in the reference
        srv->n_err++;
        respond(srv, rq.to, error_response(INVALID_ARGUMENT, "Learner-side batch size exceeded"));
        return -1;
      }
      slot = fn->cur; start = fn->next_index;
      fn->next_index += rq.count;
      if (fn->next_index == fn->N) fn->cur = -1;             // the next call opens a new computation
    }
  }
  bool ok = true;
  for (size_t i = 0; i < rq.args.size() && ok; ++i)
    ok = copy_rows(fn->in[i], rq.args[i], fn->inbuf[(size_t)slot * fn->in.size() + i], start, rq.count);
  if (!ok) {
    // cannot happen after handle_call's payload_ok(); if it ever does, the reserved rows must not keep an earlier
    // batch's bytes: zero them in every argument (the caller still gets its slice of the result)
    for (size_t i = 0; i < rq.args.size(); ++i) {
      const Spec& sp = fn->in[i];
      memset((uint8_t*)fn->inbuf[(size_t)slot * fn->in.size() + i] + (int64_t)start * sp.row_elems * sp.store_size, 0,
             (size_t)((int64_t)rq.count * sp.row_elems * sp.store_size));
    }
  }
  bool full;
  {
    std::lock_guard<std::mutex> l(fn->mu);
    Slot& s = fn->slots[slot];
    s.pend.push_back(Pending{rq.to, start, rq.count, rq.batched});
    s.num_ready += rq.count;
    full = s.num_ready == fn->N;
    if (full) { s.state = Slot::READY; fn->ready.push_back(slot); }
  }
  if (!ok) srv->n_err++;
  if (full) { srv->n_batch++; fn->cv.notify_all(); *counted = true; }
  return 1;
}

void handle_call(Server_* srv, Conn* c, int32_t stream_id, const uint8_t* data, size_t len) {
  Target to{c->io->index, c->id, stream_id};
  srv->n_call++;
  Reader r(data, len);
  std::string function;
  std::vector<std::pair<const uint8_t*, size_t>> raw;
  while (!r.done()) {
    const uint64_t tag = r.varint();
    if (!r.ok) break;
    const uint8_t* b; size_t n;
    if ((tag >> 3) == 1 && (tag & 7) == 2) { if (r.bytes(&b, &n)) function.assign((const char*)b, n); }
    else if ((tag >> 3) == 2 && (tag & 7) == 2) { if (r.bytes(&b, &n)) raw.emplace_back(b, n); }
    else r.skip((int)(tag & 7));
  }
  Request rq;
  rq.base = data; rq.len = len; rq.to = to;
  bool parsed = r.ok;
  for (auto& t : raw) {
    TensorView v;
    parsed = parsed && parse_tensor(t.first, t.second, &v);
    rq.args.push_back(v);
  }
  if (!parsed) { srv->n_err++; respond(srv, to, error_response(INVALID_ARGUMENT, "Cannot parse TensorProto.")); return; }
  auto it = srv->buckets.find(function);
  if (it == srv->buckets.end()) {
    srv->n_err++;
    respond(srv, to, error_response(INTERNAL, "Function " + function + " not found"));
    return;
  }
  Bucket* bk = it->second.get();
  Fn* fn = bk->fns[bk->counter.load() % bk->fns.size()];
  // direct call (arguments of exactly the bound shapes) or a batched / single-step call (GetArgBatchSize)
  bool direct = !rq.args.empty() && rq.args[0].rank == fn->in[0].rank;
  for (int d = 0; direct && d < fn->in[0].rank; ++d) direct = rq.args[0].dims[d] == fn->in[0].dims[d];
  std::string err;
  if (direct) {
    err = verify_args(fn->in, 0, rq.args, true);
    rq.count = fn->N; rq.batched = true;
  } else {
    const bool batched = !rq.args.empty() && rq.args[0].rank == fn->in[0].rank;
    err = verify_args(fn->in, batched ? 1 : 0, rq.args, false);
    if (err.empty() && batched) {
      const int64_t n0 = rq.args[0].dims[0];
      for (size_t i = 1; i < rq.args.size(); ++i)
        if (rq.args[i].dims[0] != n0) {
          char buf[256];
          snprintf(buf, sizeof buf, "Expects arg[%zu] to start with the batching dimension %lld but had shape %s", i,
                   (long long)n0, shape_str(rq.args[i].dims, rq.args[i].rank).c_str());
          err = buf; break;
        }
      rq.count = (int)n0;
    } else if (err.empty()) rq.count = 1;
    rq.batched = batched;
    if (err.empty() && rq.count < 1) err = "Learner-side batch size exceeded";
  }
  rq.direct = direct;
  if (!err.empty()) { srv->n_err++; respond(srv, to, error_response(INVALID_ARGUMENT, err)); return; }
  for (size_t i = 0; i < rq.args.size(); ++i)
    if (!payload_ok(fn->in[i], rq.args[i], rq.count)) {      // before any row is reserved
      srv->n_err++;
      respond(srv, to, error_response(INVALID_ARGUMENT, "Cannot parse TensorProto."));
      return;
    }
  bool counted = false;
  place(srv, fn, rq, &counted);
  if (counted) bk->counter++;
}

// ---- nghttp2 callbacks ------------------------------------------------------------------------------------------------ //
ssize_t data_read_cb(nghttp2_session* session, int32_t stream_id, uint8_t* buf, size_t length, uint32_t* data_flags,
                     nghttp2_data_source* source, void* user_data) {
  (void)source;
  Conn* c = (Conn*)user_data;
  auto it = c->streams.find(stream_id);
  if (it == c->streams.end()) return NGHTTP2_ERR_TEMPORAL_CALLBACK_FAILURE;
  Stream& s = it->second;
  const size_t avail = s.out.size() - s.out_off;
  if (avail == 0) {
    if (s.method == 1 ? s.responded : (s.remote_closed && s.inflight == 0)) {     // everything said: trailers
      *data_flags |= NGHTTP2_DATA_FLAG_EOF | NGHTTP2_DATA_FLAG_NO_END_STREAM;
      static const char kStatus[] = "grpc-status", kZero[] = "0";
      nghttp2_nv tr[1] = {{(uint8_t*)kStatus, (uint8_t*)kZero, sizeof(kStatus) - 1, 1, NGHTTP2_NV_FLAG_NONE}};
      ng()->nghttp2_submit_trailer(session, stream_id, tr, 1);
      s.finished = true;
      return 0;
    }
    s.deferred = true;
    return NGHTTP2_ERR_DEFERRED;
  }
  const size_t n = avail < length ? avail : length;
  memcpy(buf, s.out.data() + s.out_off, n);
  s.out_off += n;
  if (s.out_off == s.out.size()) { s.out.clear(); s.out_off = 0; }
  return (ssize_t)n;
}

int on_begin_headers(nghttp2_session*, const void* frame, void* user_data) {
  const nghttp2_frame_hd* hd = (const nghttp2_frame_hd*)frame;
  Conn* c = (Conn*)user_data;
  if (hd->type == NGHTTP2_HEADERS && !c->streams.count(hd->stream_id)) {
    c->streams[hd->stream_id];
    c->io->srv->n_stream++;
  }
  return 0;
}
int on_header(nghttp2_session*, const void* frame, const uint8_t* name, size_t namelen, const uint8_t* value,
              size_t valuelen, uint8_t, void* user_data) {
  const nghttp2_frame_hd* hd = (const nghttp2_frame_hd*)frame;
  Conn* c = (Conn*)user_data;
  auto it = c->streams.find(hd->stream_id);
  if (it == c->streams.end()) return 0;
  if (namelen == 5 && !memcmp(name, ":path", 5)) it->second.path.assign((const char*)value, valuelen);
  else if (namelen == 13 && !memcmp(name, "grpc-encoding", 13)) it->second.grpc_encoding.assign((const char*)value, valuelen);
  return 0;
}
void trailers_only(Conn* c, int32_t stream_id, int status, const char* message) {
  static const char kS[] = ":status", k200[] = "200", kCt[] = "content-type", kGrpc[] = "application/grpc",
                    kGs[] = "grpc-status", kGm[] = "grpc-message";
  const std::string st = std::to_string(status);
  nghttp2_nv nva[4] = {{(uint8_t*)kS, (uint8_t*)k200, 7, 3, 0}, {(uint8_t*)kCt, (uint8_t*)kGrpc, 12, 16, 0},
                       {(uint8_t*)kGs, (uint8_t*)st.data(), 11, st.size(), 0},
                       {(uint8_t*)kGm, (uint8_t*)message, 12, strlen(message), 0}};
  ng()->nghttp2_submit_response(c->session, stream_id, nva, 4, nullptr);
}
void open_response(Conn* c, int32_t stream_id) {            // HEADERS now, DATA as responses arrive
  static const char kS[] = ":status", k200[] = "200", kCt[] = "content-type", kGrpc[] = "application/grpc";
  nghttp2_nv nva[2] = {{(uint8_t*)kS, (uint8_t*)k200, 7, 3, 0}, {(uint8_t*)kCt, (uint8_t*)kGrpc, 12, 16, 0}};
  nghttp2_data_provider prd;
  prd.source.ptr = nullptr;
  prd.read_callback = data_read_cb;
  ng()->nghttp2_submit_response(c->session, stream_id, nva, 2, &prd);
}
// Consumes complete gRPC messages (1-byte compressed flag + 4-byte big-endian length + payload) from the stream.
void drain_messages(Conn* c, int32_t stream_id, Stream& s) {
  Server_* srv = c->io->srv;
  size_t off = 0;
  while (s.in.size() - off >= 5) {
    const uint8_t* h = (const uint8_t*)s.in.data() + off;
    const uint32_t len = (uint32_t)h[1] << 24 | (uint32_t)h[2] << 16 | (uint32_t)h[3] << 8 | h[4];
    if (s.in.size() - off - 5 < len) break;
    if (h[0] != 0) {                                         // compressed message: not negotiated (no grpc-accept-encoding)
      ng()->nghttp2_submit_rst_stream(c->session, NGHTTP2_FLAG_NONE, stream_id, NGHTTP2_INTERNAL_ERROR);
      s.in.clear();
      return;
    }
    if (s.method == 2) {
      s.inflight++;
      handle_call(srv, c, stream_id, (const uint8_t*)s.in.data() + off + 5, len);
    }
    off += 5 + (size_t)len;
  }
  if (off) s.in.erase(0, off);
}
int on_data_chunk(nghttp2_session*, uint8_t, int32_t stream_id, const uint8_t* data, size_t len, void* user_data) {
  Conn* c = (Conn*)user_data;
  auto it = c->streams.find(stream_id);
  if (it == c->streams.end()) return 0;
  c->io->srv->n_in += len;
  Stream& s = it->second;
  s.in.append((const char*)data, len);                     // copy 1: the message assembler (chunks of <= 1 MiB frames)
  drain_messages(c, stream_id, s);                         // copy 2: tensor bytes -> pinned batch buffer
  return 0;
}
int on_frame_recv(nghttp2_session*, const void* frame, void* user_data) {
  const nghttp2_frame_hd* hd = (const nghttp2_frame_hd*)frame;
  Conn* c = (Conn*)user_data;
  Server_* srv = c->io->srv;
  if (hd->type != NGHTTP2_HEADERS && hd->type != NGHTTP2_DATA) return 0;
  auto it = c->streams.find(hd->stream_id);
  if (it == c->streams.end()) return 0;
  Stream& s = it->second;
  if (hd->type == NGHTTP2_HEADERS && (hd->flags & NGHTTP2_FLAG_END_HEADERS) && s.method == 0) {
    if (s.path == "/seed_rl.TensorService/Init") s.method = 1;
    else if (s.path == "/seed_rl.TensorService/Call") s.method = 2;
    if (s.method == 0) { trailers_only(c, hd->stream_id, UNIMPLEMENTED, "Method not found"); s.method = -1; }
    else if (!s.grpc_encoding.empty() && s.grpc_encoding != "identity") {
      trailers_only(c, hd->stream_id, UNIMPLEMENTED, "Compression is not supported"); s.method = -1;
    } else open_response(c, hd->stream_id);
  }
  if (hd->flags & NGHTTP2_FLAG_END_STREAM) {
    s.remote_closed = true;
    if (s.method == 1 && !s.responded) {                     // TensorHandler::Init
      s.out += frame_message(srv->init_response);
      s.responded = true;
    }
    if (s.method > 0 && s.deferred) { s.deferred = false; ng()->nghttp2_session_resume_data(c->session, hd->stream_id); }
  }
  return 0;
}
int on_stream_close(nghttp2_session*, int32_t stream_id, uint32_t, void* user_data) {
  ((Conn*)user_data)->streams.erase(stream_id);
  return 0;
}

// ---- I/O threads ------------------------------------------------------------------------------------------------------ //
void set_nonblock(int fd) { fcntl(fd, F_SETFL, fcntl(fd, F_GETFL, 0) | O_NONBLOCK); }

void IoThread::add_conn(int fd) {
  set_nonblock(fd);
  int one = 1;
  setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof one);               // fails harmlessly on unix sockets
  int sz = 4 << 20;
  setsockopt(fd, SOL_SOCKET, SO_RCVBUF, &sz, sizeof sz);
  setsockopt(fd, SOL_SOCKET, SO_SNDBUF, &sz, sizeof sz);
  Conn* c = new Conn;
  c->id = srv->next_conn++; c->fd = fd; c->io = this;
  if (ng()->nghttp2_session_server_new(&c->session, srv->cbs, c) != 0) { close(fd); delete c; return; }
  // large windows and frames: one inference request carries n x 7 KB (Atari) .. n x 20 KB (DMLab) of observation bytes
  nghttp2_settings_entry iv[3] = {{NGHTTP2_SETTINGS_MAX_CONCURRENT_STREAMS, 4096},
                                  {NGHTTP2_SETTINGS_INITIAL_WINDOW_SIZE, 64u << 20},
                                  {NGHTTP2_SETTINGS_MAX_FRAME_SIZE, 1u << 20}};
  ng()->nghttp2_submit_settings(c->session, NGHTTP2_FLAG_NONE, iv, 3);
  ng()->nghttp2_session_set_local_window_size(c->session, NGHTTP2_FLAG_NONE, 0, 1 << 30);
  conns[c->id] = c;
  srv->n_conn++;
  epoll_event ev{};
  ev.events = EPOLLIN; ev.data.u64 = c->id;
  epoll_ctl(ep, EPOLL_CTL_ADD, fd, &ev);
  flush(c);
}
void IoThread::close_conn(Conn* c) {
  epoll_ctl(ep, EPOLL_CTL_DEL, c->fd, nullptr);
  close(c->fd);
  if (c->session) ng()->nghttp2_session_del(c->session);
  conns.erase(c->id);
  delete c;
}
void IoThread::flush(Conn* c) {
  if (c->dead) return;
  for (;;) {
    if (c->woff == c->wbuf.size()) {
      c->wbuf.clear(); c->woff = 0;
      // refill from the session (bounded: one socket buffer's worth at a time)
      while (c->wbuf.size() < (4u << 20)) {
        const uint8_t* p;
        const ssize_t n = ng()->nghttp2_session_mem_send(c->session, &p);
        if (n < 0) { c->dead = true; return; }
        if (n == 0) break;
        c->wbuf.append((const char*)p, (size_t)n);
      }
      if (c->wbuf.empty()) break;
    }
    const ssize_t w = ::send(c->fd, c->wbuf.data() + c->woff, c->wbuf.size() - c->woff, MSG_NOSIGNAL);
    if (w > 0) { c->woff += (size_t)w; continue; }
    if (w < 0 && (errno == EAGAIN || errno == EWOULDBLOCK)) break;
    if (w < 0 && errno == EINTR) continue;
    c->dead = true; return;
  }
  const bool need_out = c->woff < c->wbuf.size();
  if (need_out != c->want_out) {
    c->want_out = need_out;
    epoll_event ev{};
    ev.events = EPOLLIN | (need_out ? EPOLLOUT : 0); ev.data.u64 = c->id;
    epoll_ctl(ep, EPOLL_CTL_MOD, c->fd, &ev);
  }
  if (!need_out && !ng()->nghttp2_session_want_read(c->session) && !ng()->nghttp2_session_want_write(c->session))
    c->dead = true;                                          // GOAWAY exchanged, nothing left to do
}
void IoThread::on_readable(Conn* c) {
  static thread_local std::vector<uint8_t> buf(1 << 20);
  for (int rounds = 0; rounds < 16 && !c->dead; ++rounds) {
    const ssize_t n = ::recv(c->fd, buf.data(), buf.size(), 0);
    if (n > 0) {
      const ssize_t used = ng()->nghttp2_session_mem_recv(c->session, buf.data(), (size_t)n);
      if (used < 0) { c->dead = true; break; }
      if ((size_t)n < buf.size()) break;
      continue;
    }
    if (n == 0) { c->dead = true; break; }
    if (errno == EAGAIN || errno == EWOULDBLOCK) break;
    if (errno == EINTR) continue;
    c->dead = true;
  }
}
void IoThread::run() {
  t_current_io = this;
  epoll_event evs[128];
  while (!srv->shutdown.load()) {
    const int n = epoll_wait(ep, evs, 128, 200);
    std::vector<uint64_t> touched;                           // ids, not pointers: a connection may close below
    for (int i = 0; i < n; ++i) {
      const uint64_t key = evs[i].data.u64;
      if (key == 0) {                                        // wake-up: adopted connections + responses
        uint64_t v; (void)!read(wake, &v, 8);
        continue;
      }
      if (key >> 62) {                                       // a listener (thread 0 only)
        const int lfd = (int)(key & 0xFFFFFFFFu);
        for (;;) {
          const int fd = accept4(lfd, nullptr, nullptr, SOCK_CLOEXEC);
          if (fd < 0) break;
          IoThread* t = srv->io[srv->rr++ % srv->io.size()].get();
          if (t == this) add_conn(fd); else t->adopt(fd);
        }
        continue;
      }
      auto it = conns.find(key);
      if (it == conns.end()) continue;
      Conn* c = it->second;
      if (evs[i].events & (EPOLLHUP | EPOLLERR)) c->dead = true;
      if (!c->dead && (evs[i].events & EPOLLIN)) on_readable(c);
      touched.push_back(c->id);
    }
    std::vector<OutItem> items; std::vector<int> fds;
    { std::lock_guard<std::mutex> l(mu); items.swap(outbox); fds.swap(new_fds); }
    for (int fd : fds) add_conn(fd);
    for (auto& it : items) {
      auto ci = conns.find(it.conn);
      if (ci == conns.end()) continue;
      Conn* c = ci->second;
      auto si = c->streams.find(it.stream);
      if (si == c->streams.end()) continue;
      Stream& s = si->second;
      s.out += it.data;
      if (s.inflight > 0) s.inflight--;
      if (s.deferred) { s.deferred = false; ng()->nghttp2_session_resume_data(c->session, it.stream); }
      touched.push_back(c->id);
    }
    std::sort(touched.begin(), touched.end());
    touched.erase(std::unique(touched.begin(), touched.end()), touched.end());
    for (uint64_t id : touched) {
      auto ci = conns.find(id);
      if (ci == conns.end()) continue;
      Conn* c = ci->second;
      if (!c->dead) flush(c);
      if (c->dead) close_conn(c);                            // erases it from `conns`; no other reference is kept
    }
  }
  // shutdown: nothing more is written (grpc.cc:336-343); connections close, clients see UNAVAILABLE
  std::vector<Conn*> all;
  for (auto& kv : conns) all.push_back(kv.second);
  for (Conn* c : all) close_conn(c);
}

int listen_on(Server_* srv, const std::string& addr, int* port_out) {
  *port_out = 0;
  int fd = -1;
  if (addr.rfind("unix:", 0) == 0) {
    std::string path = addr.substr(5);
    if (path.rfind("//", 0) == 0) path = path.substr(2);
    sockaddr_un sa{};
    sa.sun_family = AF_UNIX;
    if (path.size() >= sizeof(sa.sun_path)) return fail(-1, "unix socket path too long: %s", path.c_str());
    strcpy(sa.sun_path, path.c_str());
    unlink(path.c_str());
    fd = socket(AF_UNIX, SOCK_STREAM | SOCK_CLOEXEC, 0);
    if (fd < 0 || bind(fd, (sockaddr*)&sa, sizeof sa) != 0) {
      if (fd >= 0) close(fd);
      return fail(-1, "cannot bind %s: %s", addr.c_str(), strerror(errno));
    }
    srv->unix_paths.push_back(path);
  } else {
    const size_t colon = addr.rfind(':');
    if (colon == std::string::npos) return fail(-1, "address must be host:port or unix:path, got %s", addr.c_str());
    std::string host = addr.substr(0, colon), port = addr.substr(colon + 1);
    if (host.size() >= 2 && host.front() == '[' && host.back() == ']') host = host.substr(1, host.size() - 2);
    addrinfo hints{}, *res = nullptr;
    hints.ai_family = AF_UNSPEC; hints.ai_socktype = SOCK_STREAM; hints.ai_flags = AI_PASSIVE;
    const int rc = getaddrinfo(host.empty() || host == "*" ? nullptr : host.c_str(), port.c_str(), &hints, &res);
    if (rc != 0) return fail(-1, "cannot resolve %s: %s", addr.c_str(), gai_strerror(rc));
    for (addrinfo* a = res; a; a = a->ai_next) {
      fd = socket(a->ai_family, a->ai_socktype | SOCK_CLOEXEC, a->ai_protocol);
      if (fd < 0) continue;
      int one = 1;
      setsockopt(fd, SOL_SOCKET, SO_REUSEADDR, &one, sizeof one);
      if (bind(fd, a->ai_addr, a->ai_addrlen) == 0) {
        sockaddr_storage ss{}; socklen_t sl = sizeof ss;
        getsockname(fd, (sockaddr*)&ss, &sl);
        *port_out = ntohs(ss.ss_family == AF_INET6 ? ((sockaddr_in6*)&ss)->sin6_port : ((sockaddr_in*)&ss)->sin_port);
        break;
      }
      close(fd); fd = -1;
    }
    freeaddrinfo(res);
    if (fd < 0) return fail(-1, "cannot bind %s: %s", addr.c_str(), strerror(errno));
  }
  if (listen(fd, 1024) != 0) { close(fd); return fail(-1, "listen(%s): %s", addr.c_str(), strerror(errno)); }
  set_nonblock(fd);
  srv->listeners.push_back(fd);
  return 0;
}

}  // namespace

// ---- C ABI ------------------------------------------------------------------------------------------------------------ //
extern "C" {

const char* seedserve_last_error(void) { return g_err; }
int seedserve_abi_version(void) { return SEEDSERVE_ABI_VERSION; }

seedserve_server* seedserve_create(int num_io_threads) {
  if (!ng()) { fail(-1, "%s", g_ng_why.c_str()); return nullptr; }
  if (num_io_threads < 1) num_io_threads = 1;
  if (num_io_threads > 64) num_io_threads = 64;
  auto* s = new seedserve_server;
  ng()->nghttp2_session_callbacks_new(&s->cbs);
  ng()->nghttp2_session_callbacks_set_on_frame_recv_callback(s->cbs, on_frame_recv);
  ng()->nghttp2_session_callbacks_set_on_data_chunk_recv_callback(s->cbs, on_data_chunk);
  ng()->nghttp2_session_callbacks_set_on_stream_close_callback(s->cbs, on_stream_close);
  ng()->nghttp2_session_callbacks_set_on_begin_headers_callback(s->cbs, on_begin_headers);
  ng()->nghttp2_session_callbacks_set_on_header_callback(s->cbs, on_header);
  for (int i = 0; i < num_io_threads; ++i) {
    auto t = std::make_unique<IoThread>();
    t->srv = s; t->index = i;
    t->ep = epoll_create1(EPOLL_CLOEXEC);
    t->wake = eventfd(0, EFD_NONBLOCK | EFD_CLOEXEC);
    epoll_event ev{};
    ev.events = EPOLLIN; ev.data.u64 = 0;
    epoll_ctl(t->ep, EPOLL_CTL_ADD, t->wake, &ev);
    s->io.push_back(std::move(t));
  }
  return s;
}

int seedserve_listen(seedserve_server* s, const char* address) {
  if (!s || !address) return fail(-1, "seedserve_listen: null argument");
  if (s->started.load()) return fail(-1, "Server is already started");
  int port = 0;
  const int rc = listen_on(s, address, &port);
  return rc < 0 ? rc : port;
}

int seedserve_bind(seedserve_server* s, const char* name, int num_inputs, const seedserve_spec* inputs, int num_outputs,
                   const seedserve_spec* outputs, int num_slots, void* const* input_buffers,
                   void* const* output_buffers) {
  if (!s || !name || !inputs || num_inputs < 1 || num_outputs < 0 || num_slots < 1 || !input_buffers ||
      (num_outputs > 0 && (!outputs || !output_buffers)))
    return fail(-1, "seedserve_bind: bad arguments");
  if (s->started.load()) return fail(-1, "Server is already started");
  auto fn = std::make_unique<Fn>();
  fn->id = (int)s->fns.size(); fn->name = name; fn->num_slots = num_slots;
  auto conv = [&](const seedserve_spec& sp, bool input, Spec* o) -> int {
    if (sp.rank < 1 || sp.rank > SEEDSERVE_MAX_RANK) return fail(-1, "bind: every tensor needs the batch dimension (rank 1..%d)", SEEDSERVE_MAX_RANK);
    o->dtype = sp.dtype; o->rank = sp.rank; o->row_elems = 1;
    for (int i = 0; i < sp.rank; ++i) { o->dims[i] = sp.dims[i]; if (i) o->row_elems *= sp.dims[i]; }
    o->wire_size = dtype_size(sp.dtype);
    if (!o->wire_size) return fail(-1, "bind: unsupported dtype %d (numeric tensors only)", sp.dtype);
    o->widen = input && sp.widen_to_int64 && sp.dtype == DT_INT32;
    o->store_size = o->widen ? 8 : o->wire_size;
    return 0;
  };
  fn->in.resize(num_inputs); fn->out.resize(num_outputs);
  for (int i = 0; i < num_inputs; ++i) if (conv(inputs[i], true, &fn->in[i]) < 0) return -1;
  for (int i = 0; i < num_outputs; ++i) if (conv(outputs[i], false, &fn->out[i]) < 0) return -1;
  fn->N = (int)fn->in[0].dims[0];
  if (fn->N < 1) return fail(-1, "bind: batch dimension must be >= 1");
  for (auto& sp : fn->in) if (sp.dims[0] != fn->N) return fail(-1, "bind: every input must lead with the batch dimension %d (CanBatch)", fn->N);
  for (auto& sp : fn->out) if (sp.dims[0] != fn->N) return fail(-1, "bind: every output must lead with the batch dimension %d (CanBatch)", fn->N);
  fn->inbuf.assign(input_buffers, input_buffers + (size_t)num_slots * num_inputs);
  if (num_outputs) fn->outbuf.assign(output_buffers, output_buffers + (size_t)num_slots * num_outputs);
  for (void* p : fn->inbuf) if (!p) return fail(-1, "bind: null input buffer");
  for (void* p : fn->outbuf) if (!p) return fail(-1, "bind: null output buffer");
  fn->slots.resize(num_slots);
  auto& bk = s->buckets[name];
  if (!bk) bk = std::make_unique<Bucket>();
  bk->fns.push_back(fn.get());
  s->fns.push_back(std::move(fn));
  return (int)s->fns.size() - 1;
}

int seedserve_set_init_response(seedserve_server* s, const void* bytes, size_t len) {
  if (!s || (!bytes && len)) return fail(-1, "seedserve_set_init_response: null argument");
  s->init_response.assign((const char*)bytes, len);
  return 0;
}

int seedserve_start(seedserve_server* s) {
  if (!s) return fail(-1, "seedserve_start: null server");
  if (s->fns.empty()) return fail(-UNAVAILABLE, "No function was bound");
  if (s->started.exchange(true)) return fail(-INVALID_ARGUMENT, "Server is already started");
  for (int fd : s->listeners) {
    epoll_event ev{};
    ev.events = EPOLLIN; ev.data.u64 = (1ull << 62) | (uint32_t)fd;
    epoll_ctl(s->io[0]->ep, EPOLL_CTL_ADD, fd, &ev);
  }
  for (auto& t : s->io) { IoThread* p = t.get(); p->th = std::thread([p] { p->run(); }); }
  return 0;
}

int seedserve_next_batch(seedserve_server* s, int fn_id, int timeout_ms) {
  if (!s || fn_id < 0 || fn_id >= (int)s->fns.size()) return fail(-3, "seedserve_next_batch: bad function id");
  Fn* fn = s->fns[fn_id].get();
  std::unique_lock<std::mutex> l(fn->mu);
  const auto deadline = std::chrono::steady_clock::now() + std::chrono::milliseconds(timeout_ms < 0 ? 0 : timeout_ms);
  while (fn->ready.empty()) {
    if (s->shutdown.load()) return -2;
    if (fn->cv.wait_until(l, deadline) == std::cv_status::timeout && fn->ready.empty()) return s->shutdown.load() ? -2 : -1;
  }
  const int slot = fn->ready.front();
  fn->ready.pop_front();
  fn->slots[slot].state = Slot::COMPUTING;
  return slot;
}

int seedserve_complete(seedserve_server* s, int fn_id, int slot, int status_code, const char* message) {
  if (!s || fn_id < 0 || fn_id >= (int)s->fns.size()) return fail(-3, "seedserve_complete: bad function id");
  Fn* fn = s->fns[fn_id].get();
  if (slot < 0 || slot >= fn->num_slots) return fail(-3, "seedserve_complete: bad slot");
  std::vector<Pending> pend;
  {
    std::lock_guard<std::mutex> l(fn->mu);
    if (fn->slots[slot].state != Slot::COMPUTING) return fail(-3, "seedserve_complete: slot %d is not being computed", slot);
    pend.swap(fn->slots[slot].pend);
  }
  if (!s->shutdown.load()) {                                 // grpc.cc:336-343: nothing is written once the server shuts down
    for (const Pending& p : pend) {
      if (status_code != OK) { respond(s, p.to, error_response(status_code, message ? message : "")); continue; }
      std::string resp;
      for (size_t o = 0; o < fn->out.size(); ++o) {
        const Spec& sp = fn->out[o];
        const uint8_t* rows = (const uint8_t*)fn->outbuf[(size_t)slot * fn->out.size() + o] +
                              (int64_t)p.start * sp.row_elems * sp.wire_size;
        encode_tensor(&resp, sp, rows, p.count, p.batched);
      }
      respond(s, p.to, frame_message(resp));
    }
  }
  std::deque<Request> retry;
  {
    std::lock_guard<std::mutex> l(fn->mu);
    fn->slots[slot].state = Slot::FREE;
    fn->slots[slot].num_ready = 0;
    retry.swap(fn->overflow);
  }
  Bucket* bk = s->buckets[fn->name].get();
  for (auto& rq : retry) {                                    // calls that found every slot busy
    bool counted = false;
    place(s, fn, rq, &counted);
    if (counted) bk->counter++;
  }
  return 0;
}

int seedserve_shutdown(seedserve_server* s) {
  if (!s) return fail(-1, "seedserve_shutdown: null server");
  if (s->shutdown.exchange(true)) return 0;
  for (auto& fn : s->fns) { std::lock_guard<std::mutex> l(fn->mu); fn->cv.notify_all(); }
  if (s->started.load())
    for (auto& t : s->io) {
      uint64_t one = 1; (void)!write(t->wake, &one, 8);
      if (t->th.joinable()) t->th.join();
    }
  for (int fd : s->listeners) close(fd);
  s->listeners.clear();
  for (auto& p : s->unix_paths) unlink(p.c_str());
  return 0;
}

void seedserve_destroy(seedserve_server* s) {
  if (!s) return;
  seedserve_shutdown(s);
  for (auto& t : s->io) { close(t->ep); close(t->wake); }
  if (s->cbs) ng()->nghttp2_session_callbacks_del(s->cbs);
  delete s;
}

int seedserve_get_stats(seedserve_server* s, seedserve_stats* out) {
  if (!s || !out) return fail(-1, "seedserve_get_stats: null argument");
  out->connections = s->n_conn; out->streams = s->n_stream; out->calls = s->n_call; out->batches = s->n_batch;
  out->bytes_in = s->n_in; out->bytes_out = s->n_out; out->errors = s->n_err;
  return 0;
}

}  // extern "C"

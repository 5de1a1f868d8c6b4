/* seedserve.h -- C ABI of libseedserve.so: the native transport front-end of the learner.
 *
 * Replaces, for the actor -> learner inference path, the reference's C++ gRPC server ops
 * (/root/reference/grpc/ops/grpc.cc): TensorHandler (:141-233: function table, Init / streaming Call of
 * grpc/service.proto:28-57), the completion-queue server (:366-475), verify_args / GetArgBatchSize (:527-589) and
 * DynamicFn's server-side batching (:591-861) -- N single-step calls, or client-side batches of k rows, are gathered
 * into ONE invocation of the bound function and its outputs are sliced back to the callers.
 *
 * Design (host-only, no HIP, no torch): epoll I/O threads speak HTTP/2 (framing / HPACK / flow control through the
 * system's libnghttp2, loaded at run time) and the gRPC message framing themselves; a CallRequest is parsed in place
 * and every TensorProto.tensor_content is copied ONCE, straight into the caller-owned (pinned) batch buffers of the
 * bound function at the rows this call reserved.  A full batch is handed to the compute side through
 * seedserve_next_batch(); seedserve_complete() slices the output buffers into CallResponses and wakes the I/O threads.
 * Unmodified reference actors (grpc_client_call op / grpcio clients) connect: same service, methods, messages,
 * status codes and error strings.
 *
 * Thread safety: every function may be called from any thread; next_batch / complete are meant for one compute
 * thread per bound function.  All functions return 0 (or a non-negative value) on success and a negative code on
 * failure; seedserve_last_error() returns the calling thread's last message. */
#ifndef SEEDSERVE_H_
#define SEEDSERVE_H_
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SEEDSERVE_ABI_VERSION 1
#define SEEDSERVE_MAX_RANK 8

typedef struct seedserve_server seedserve_server;

/* One tensor of a bound function's flat signature (a tf.TensorSpec, grpc/python/ops.py `bind`): `dtype` is the
 * tensorflow DataType enum (DT_FLOAT 1, DT_INT32 3, DT_UINT8 4, DT_INT64 9, DT_BOOL 10, DT_UINT16 17, ...), dims[0] is the
 * batch dimension N shared by every input and output (CanBatch, grpc.cc:948-973).  widen_to_int64: an int32 input is
 * stored as int64 in the batch buffer (the device-side tables index with int64 ids); outputs ignore it. */
typedef struct seedserve_spec {
  int32_t dtype;
  int32_t rank;
  int64_t dims[SEEDSERVE_MAX_RANK];
  int32_t widen_to_int64;
} seedserve_spec;

typedef struct seedserve_stats {
  uint64_t connections, streams, calls, batches, bytes_in, bytes_out, errors;
} seedserve_stats;

const char* seedserve_last_error(void);
int seedserve_abi_version(void);

/* grpc.Server(server_addresses) of grpc/python/ops.py: nothing listens until seedserve_start. */
seedserve_server* seedserve_create(int num_io_threads);
/* "unix:/path", "unix:///path", "host:port", "[::]:port" / "localhost:port".  Port 0 picks a free port; the bound
 * port (0 for unix sockets) is returned. */
int seedserve_listen(seedserve_server*, const char* address);
/* server.bind(fn): `name` with batched inputs / outputs.  The caller owns num_slots sets of batch buffers
 * (input_buffers[slot * num_inputs + i] holds dims-shaped rows of input i, output_buffers likewise) and keeps them
 * alive until seedserve_destroy; pinned host memory makes the compute side's H2D / D2H copies asynchronous.
 * Binding the same name again adds another instance: calls go round-robin over the instances (grpc.cc:193-205).
 * Returns the function id. */
int seedserve_bind(seedserve_server*, const char* name, int num_inputs, const seedserve_spec* inputs, int num_outputs,
                   const seedserve_spec* outputs, int num_slots, void* const* input_buffers,
                   void* const* output_buffers);
/* The serialized seed_rl.InitResponse (method output signatures as tensorflow.StructuredValue) returned by `Init`;
 * built by the host language that knows the nests (seed_rl_amd/grpc_native.py). */
int seedserve_set_init_response(seedserve_server*, const void* bytes, size_t len);
int seedserve_start(seedserve_server*);
/* Compute side.  Blocks up to timeout_ms for a FULL batch of function fn_id: returns its slot (>= 0), -1 on timeout,
 * -2 after shutdown.  The slot's input buffers are complete and stay untouched until seedserve_complete(slot). */
int seedserve_next_batch(seedserve_server*, int fn_id, int timeout_ms);
/* The slot's output buffers hold the function's results (status_code 0) -- every caller gets its rows -- or the call
 * failed with tensorflow.error.Code status_code / message (all callers of the batch get that status). */
int seedserve_complete(seedserve_server*, int fn_id, int slot, int status_code, const char* message);
/* server.shutdown(): pending calls of unfilled batches get CANCELLED "Server shutdown." (grpc.cc:336-343 semantics:
 * nothing is written after shutdown; streams are closed), listeners close, I/O threads join. */
int seedserve_shutdown(seedserve_server*);
void seedserve_destroy(seedserve_server*);
int seedserve_get_stats(seedserve_server*, seedserve_stats* out);

#ifdef __cplusplus
}
#endif
#endif /* SEEDSERVE_H_ */

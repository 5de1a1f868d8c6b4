// Bookkeeping kernels of the learner-side batched inference step.
//
// Replaces the small-tensor part of the `inference` tf.function of
// /root/reference/agents/vtrace/learner.py:350-405 -- ~25 gather / scatter / where / boolean-mask ops on
// [inference_batch_size] tensors, several of which produce data-dependent shapes (tf.where -> tf.gather) -- with
// TWO launches that keep every shape static, so that the whole step has no host synchronisation and can be
// replayed from a HIP graph:
//   inference_pre  : id validation (range, duplicates through a per-env stamp), run-id compare + reset bookkeeping
//                    (:353-366), episode statistics (:373-378), previous action read (:381)
//   inference_post : store index advance, completed-unroll detection (utils.py:229-233), destination columns
//                    in the time-major training batch (exclusive scan over the batch), index lists for the
//                    row mover (store.hip) that appends the step, emits completed unrolls and carries the last
//                    step over (utils.py:237-255), action table update (:403); optionally SAMPLES the actions from the
//                    policy-head rows (Gumbel-max over Philox randoms, dmlab/networks.py:122) so that no eager op
//                    sits between the head GEMM and the bookkeeping.
//   rows_move_ops  : all data movement of the step as 4 launches of mutually independent row moves.
// One workgroup looping over 1024-row chunks; n = inference batch size <= 65536.  All integer work: exact.
#include "common.h"
#include "../../include/seedhip.h"

namespace {

constexpr int kMaxInferenceRows = 65536;

struct PreArgs {
  const long long* env_ids; const long long* run_ids; const float* reward; const float* raw_reward;
  const uint8_t* done; int n; int num_envs; int num_action_repeats;
  long long* run_ids_tab; long long* info_frames; float* info_return; float* info_raw_return;
  long long* actions_tab; long long* store_index;
  uint8_t* reset_mask; long long* prev_actions;
  float* episode_stats; int stats_capacity; int* stats_count; int* error_flag;
  long long* ids_safe; uint8_t* valid; int* stamp_tab; int* call_counter;
  uint8_t* will_complete; int full_length;                 // optional: rows whose unroll completes with this step
};

__global__ void __launch_bounds__(1024)
inference_pre_kernel(PreArgs a) {
  // this call's stamp: every thread reads the counter, thread 0 advances it once all have (one workgroup)
  const int stamp = *a.call_counter + 1;
  __syncthreads();
  if (threadIdx.x == 0) *a.call_counter = stamp;
  for (int i = threadIdx.x; i < a.n; i += 1024) {            // rows are independent: batches above 1024 rows just loop
    const long long e = a.env_ids[i];
    if (e < 0 || e >= a.num_envs) {
      // the reference raises (tf scatter / gather out of range); here the row is masked out of EVERY table access of
      // the step (valid = 0, ids_safe = 0 keeps reads in range) and the error is flagged
      atomicOr(a.error_flag, 1);
      a.reset_mask[i] = 0; a.prev_actions[i] = 0; a.ids_safe[i] = 0; a.valid[i] = 0;
      if (a.will_complete) a.will_complete[i] = 0;
      continue;
    }
    // duplicate ids in one batch are an error in the reference (utils.py:173-176): the first occurrence stamps the
    // env's slot, a later one finds this call's stamp there -- O(1) per row instead of an O(n) scan; a duplicate row is
    // masked out like an invalid one (two rows updating one env's tables would race)
    if (atomicExch(a.stamp_tab + e, stamp) == stamp) {
      atomicOr(a.error_flag, 2);
      a.reset_mask[i] = 0; a.prev_actions[i] = 0; a.ids_safe[i] = e; a.valid[i] = 0;
      if (a.will_complete) a.will_complete[i] = 0;
      continue;
    }
    a.ids_safe[i] = e; a.valid[i] = 1;
    const long long prev_run = a.run_ids_tab[e];
    const long long run = a.run_ids[i];
    a.run_ids_tab[e] = run;                                              // learner.py:354
    const bool reset = prev_run != run;                                  // :355-357
    long long frames = a.info_frames[e];
    float ret = a.info_return[e], raw = a.info_raw_return[e];
    long long act = a.actions_tab[e];
    if (reset) {                                                         // :360-366
      frames = 0; ret = 0.f; raw = 0.f; act = 0;
      a.store_index[e] = 0;                                              // UnrollStore.reset, overlap 0 (utils.py:207)
      a.actions_tab[e] = 0;
    }
    ret += a.reward[i]; raw += a.raw_reward[i];                          // :373
    if (a.done[i]) {                                                     // :374-377: report + reset the episode stats
      const int slot = atomicAdd(a.stats_count, 1);
      if (slot < a.stats_capacity) {
        a.episode_stats[3 * slot + 0] = (float)frames;
        a.episode_stats[3 * slot + 1] = ret;
        a.episode_stats[3 * slot + 2] = raw;
      }
      frames = 0; ret = 0.f; raw = 0.f;
    }
    frames += a.num_action_repeats;                                      // :378
    a.info_frames[e] = frames; a.info_return[e] = ret; a.info_raw_return[e] = raw;
    a.reset_mask[i] = reset ? 1 : 0;
    a.prev_actions[i] = act;                                             // :381
    // the unroll of env e completes with this step iff its (possibly just reset) write index is the last row: known
    // BEFORE the agent runs, so the previous agent state of exactly those envs can be set aside (learner.py:398-399)
    if (a.will_complete) a.will_complete[i] = (a.store_index[e] + 1 == a.full_length) ? 1 : 0;
  }
}

struct PostArgs {
  const long long* env_ids;    // ids_safe of inference_pre
  const uint8_t* valid;        // [n] rows to process (NULL = all)
  long long* actions;          // [n] in: the agent's actions; with `logits` set: OUT, sampled here
  const float* logits; int logits_ld; int num_actions;   // policy head rows (NULL = actions given)
  unsigned long long* rng;     // [2] seed, call counter (advanced by one per launch when sampling)
  int n; int num_envs; int full_length; int batch_capacity;
  long long* store_index; long long* actions_tab; int* batch_count;
  long long* append_rows;      // [n]          store row (idx*E + e) this step is written to
  uint8_t* complete;           // [n]          1 where the step completed an unroll that got a batch column
  uint8_t* carry;              // [n]          1 where the step completed an unroll (carried over to slot 0 regardless)
  long long* batch_cols;       // [n]          destination column in the training batch (valid where complete)
  long long* emit_env;         // [n]          compact list: env of the r-th completed unroll that got a column ...
  long long* emit_col;         // [n]          ... and that column; r < *emit_count
  int* emit_count;             // [1]          number of entries of the compact list
  const int* batch_start;      // [1] or NULL  ring head of the training batch: column = (start + fill + rank) % capacity
  long long* last_rows;        // [n]          (L-1)*E + e: the step carried over to slot 0 (utils.py:237-252)
  int* error_flag;
};

__global__ void __launch_bounds__(1024)
inference_post_kernel(PostArgs a) {
  __shared__ int s_wave[16];                                           // per-wave totals of the chunk being scanned
  __shared__ int s_base, s_start;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int L = a.full_length, E = a.num_envs;
  unsigned long long seed = 0, call = 0;
  if (a.logits) { seed = a.rng[0]; call = a.rng[1]; }
  if (tid == 0) { s_base = *a.batch_count; s_start = a.batch_start ? *a.batch_start : 0; }
  __syncthreads();
  if (a.logits && tid == 0) a.rng[1] = call + 1;                       // every thread has read the counter
  const int base = s_base, start = s_start;
  int running = 0;                                                     // completions in the chunks before this one (uniform)
  // 1024 rows per pass; the position of a completed unroll in the batch is its rank in env_ids order (like
  // tf.gather(env_ids, tf.where(...))): wave-level inclusive scan + the totals of the waves / chunks before
  for (int c0 = 0; c0 < a.n; c0 += 1024) {
    const int i = c0 + tid;
    long long e = 0;
    int done = 0;
    bool ok_row = false;
    if (i < a.n) {
      e = a.env_ids[i];
      ok_row = a.valid ? a.valid[i] != 0 : (e >= 0 && e < E);
      if (!ok_row) e = 0;
      long long act;
      if (a.logits) {                                                  // dmlab/networks.py:122 (sample in the head)
        act = seedhip::sample_categorical_row(a.logits + (long long)i * a.logits_ld, a.num_actions, seed, call, (unsigned)i);
        a.actions[i] = act;
      } else {
        act = a.actions[i];
      }
      if (ok_row) {
        const long long idx = a.store_index[e];
        a.append_rows[i] = idx * E + e;
        done = (idx + 1 == L) ? 1 : 0;
        if (idx + 1 > L) atomicOr(a.error_flag, 4);
        a.store_index[e] = done ? 1 : idx + 1;                         // utils.py:194, 254-255 (overlap 0)
        a.actions_tab[e] = act;                                        // learner.py:403
      } else {
        a.append_rows[i] = 0;
      }
      a.last_rows[i] = (long long)(L - 1) * E + e;
    }
    const unsigned long long votes = __ballot(done);
    const int in_wave = __popcll(votes & ((2ull << lane) - 1ull));     // inclusive rank inside the wave
    if (lane == 0) s_wave[wave] = __popcll(votes);
    __syncthreads();
    int before = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) { const int v = s_wave[w]; total += v; if (w < wave) before += v; }
    __syncthreads();                                                   // s_wave is rewritten by the next chunk
    if (i < a.n) {
      const int rank = running + before + in_wave - done;              // among the completions of the whole batch
      const int pos = base + rank;                                     // position in the batch, counted from its head
      // a completed unroll that finds the training batch full is dropped (flag 8) but its last step is STILL carried
      // to slot 0, so the env's next unroll starts from a consistent state
      const bool ok = done && pos < a.batch_capacity;
      int col = start + pos;                                           // the batch is a ring of columns
      if (col >= a.batch_capacity) col -= a.batch_capacity;
      a.complete[i] = ok ? 1 : 0;
      a.carry[i] = done ? 1 : 0;
      a.batch_cols[i] = ok ? col : 0;
      if (ok) {                                                        // rank among the accepted completions
        a.emit_env[rank] = e;
        a.emit_col[rank] = col;
      }
    }
    running += total;
  }
  if (tid == 0) {
    const int total = running;
    *a.batch_count = base + total > a.batch_capacity ? a.batch_capacity : base + total;
    if (base + total > a.batch_capacity) atomicOr(a.error_flag, 8);
    const int room = a.batch_capacity - base;
    *a.emit_count = total < room ? total : (room > 0 ? room : 0);
  }
}

// Completed unrolls -> training batch (what unroll_queue.enqueue_many + dequeue + make_time_major do in the reference,
// learner.py:396-397, 418-432), driven by the COMPACT list inference_post leaves on the device: work item (r, t) moves
// store row t * E + env_r of every field to batch row t * capacity + col_r.  Only the ~n / T unrolls that really
// completed are touched: the masked generic mover dispatched full_length * n row groups per field (24 k workgroups at
// n = 1024, 95 % of them exiting on the mask: 20 us), and the row lists it needed took another kernel to fill.
constexpr int kEmitFields = 16;
struct EmitArgs {
  void* dst[kEmitFields]; const void* src[kEmitFields]; long long row_bytes[kEmitFields]; int w[kEmitFields];
  int nfields; const long long* env; const long long* col; const int* count; int L, E, cap;
};
__global__ void __launch_bounds__(256)
emit_unrolls_kernel(EmitArgs a) {
  const int items = *a.count * a.L;
  for (int it = blockIdx.x; it < items; it += gridDim.x) {
    const int r = it / a.L, t = it - r * a.L;
    const long long srow = (long long)t * a.E + a.env[r], drow = (long long)t * a.cap + a.col[r];
    for (int f = 0; f < a.nfields; ++f) {
      const long long rb = a.row_bytes[f];
      const char* sp = (const char*)a.src[f] + srow * rb;
      char* dp = (char*)a.dst[f] + drow * rb;
      if (a.w[f] == 16) {
        for (long long e = threadIdx.x; e < rb / 16; e += 256) reinterpret_cast<uint4*>(dp)[e] = reinterpret_cast<const uint4*>(sp)[e];
      } else if (a.w[f] == 4) {
        for (long long e = threadIdx.x; e < rb / 4; e += 256) reinterpret_cast<uint32_t*>(dp)[e] = reinterpret_cast<const uint32_t*>(sp)[e];
      } else {
        for (long long e = threadIdx.x; e < rb; e += 256) dp[e] = sp[e];
      }
    }
  }
}

__global__ void __launch_bounds__(256)
categorical_sample_kernel(const float* __restrict__ logits, int ld, long long rows, int A,
                          unsigned long long* __restrict__ rng, long long* __restrict__ actions) {
  const unsigned long long seed = rng[0], call = rng[1];
  const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r < rows) actions[r] = seedhip::sample_categorical_row(logits + r * ld, A, seed, call, (unsigned)r);
}
__global__ void rng_advance_kernel(unsigned long long* rng) { rng[1] += 1; }

// Several independent row moves in ONE launch: dst[dst_rows[i]] = src[src_rows[i]] with per-operation row count, mask
// and pitches.  The ~8 data movements of an inference step (previous-state gather, append, emit, carry, state tables)
// become 4 launches of mutually independent operations.
// Every operation gets exactly the workgroups it needs (a 1-D grid with per-operation block ranges; r02c sized one grid
// dimension by the LARGEST operation, so the scalar fields of a step dispatched hundreds of empty workgroups beside the
// 28 KB frame-state rows), and no index is divided per element: long rows are cut into 1024-element pieces (one
// uniform division per workgroup), short rows are packed several to a workgroup (one division per thread).
constexpr int kMaxOps = 32;
constexpr int kPiece = 1024;                              // elements of a long row per workgroup
struct RowOp {
  void* dst; const void* src; long long dst_pitch; long long src_pitch;
  const long long* dst_rows; const long long* src_rows; const uint8_t* mask;
  unsigned n, row_elems, w;                               // rows; elements of w bytes per row (w = 16, 4 or 1)
  unsigned pieces;                                        // long rows (row_elems > 256): workgroups per row; else 0
  unsigned rows_per_wg;                                   // short rows: rows per workgroup
  int zero_where_masked;
};
struct OpsArgs { RowOp op[kMaxOps]; int blk_start[kMaxOps + 1]; int nops; };

__device__ __forceinline__ void move_elem(const RowOp& o, long long drow, long long srow, unsigned el, bool zero) {
  char* d = (char*)o.dst + drow * o.dst_pitch + (long long)el * o.w;
  const char* sp = (o.src && !zero) ? (const char*)o.src + srow * o.src_pitch + (long long)el * o.w : nullptr;
  if (o.w == 16) *reinterpret_cast<uint4*>(d) = sp ? *reinterpret_cast<const uint4*>(sp) : make_uint4(0, 0, 0, 0);
  else if (o.w == 4) *reinterpret_cast<uint32_t*>(d) = sp ? *reinterpret_cast<const uint32_t*>(sp) : 0u;
  else *d = sp ? *sp : (char)0;
}

__global__ void __launch_bounds__(256)
rows_move_ops_kernel(OpsArgs a) {
  int k = 0;
  while (k + 1 < a.nops && (int)blockIdx.x >= a.blk_start[k + 1]) ++k;       // scalar: <= 32 entries
  const RowOp& o = a.op[k];
  const unsigned lb = blockIdx.x - (unsigned)a.blk_start[k];
  unsigned r, el, el_end, step;
  if (o.pieces) {                                           // one piece of one long row
    r = lb / o.pieces;
    const unsigned part = lb - r * o.pieces;
    el = part * kPiece + threadIdx.x;
    el_end = (part + 1) * kPiece < o.row_elems ? (part + 1) * kPiece : o.row_elems;
    step = 256;
  } else {                                                  // several short rows
    const unsigned rl = threadIdx.x / o.row_elems;
    el = threadIdx.x - rl * o.row_elems;
    r = lb * o.rows_per_wg + rl;
    el_end = rl < o.rows_per_wg ? o.row_elems : 0;
    step = 0x40000000u;                                     // one element per thread
  }
  if (r >= o.n) return;
  bool zero = false;
  if (o.mask) {
    const bool m = o.mask[r] != 0;
    if (o.zero_where_masked) zero = m; else if (!m) return;
  }
  const long long drow = o.dst_rows ? o.dst_rows[r] : (long long)r;
  const long long srow = o.src_rows ? o.src_rows[r] : (long long)r;
  for (; el < el_end; el += step) move_elem(o, drow, srow, el, zero);
}

template <typename V>
__global__ void __launch_bounds__(256)
rows_move_masked_kernel(V* __restrict__ dst, const long long* __restrict__ dst_rows, const V* __restrict__ src,
                        const long long* __restrict__ src_rows, long long n, long long row_elems,
                        const uint8_t* __restrict__ mask, int zero_where_masked) {
  const long long total = n * row_elems;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const long long r = i / row_elems, el = i - r * row_elems;
    const bool m = mask[r] != 0;
    V v{};
    if (zero_where_masked) {
      if (!m && src) v = src[(src_rows ? src_rows[r] : r) * row_elems + el];
    } else {
      if (!m) continue;
      if (src) v = src[(src_rows ? src_rows[r] : r) * row_elems + el];
    }
    dst[(dst_rows ? dst_rows[r] : r) * row_elems + el] = v;
  }
}


// All fields of a structure in ONE launch (blockIdx.y = field): the per-field launches of the inference step
// (10 fields x {append, emit, carry}) collapse to three.
constexpr int kMaxFields = 16;
struct MultiArgs {
  void* dst[kMaxFields]; const void* src[kMaxFields]; long long row_bytes[kMaxFields];
  const long long* dst_rows; const long long* src_rows; long long n; const uint8_t* mask; int zero_where_masked;
};

__global__ void __launch_bounds__(256)
rows_move_multi_kernel(MultiArgs a) {
  const int f = blockIdx.y;
  const long long rb = a.row_bytes[f];
  const uintptr_t al = (uintptr_t)a.dst[f] | (uintptr_t)a.src[f] | (uintptr_t)rb;
  const int w = (al & 15) == 0 ? 16 : ((al & 3) == 0 ? 4 : 1);       // uniform per field
  const long long row_elems = rb / w;
  const long long total = a.n * row_elems;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const long long r = i / row_elems, el = i - r * row_elems;
    const bool m = a.mask ? a.mask[r] != 0 : true;
    bool zero = false;
    if (a.mask) {
      if (a.zero_where_masked) zero = m; else if (!m) continue;
    }
    const long long so = ((a.src_rows ? a.src_rows[r] : r) * row_elems + el) * w;
    const long long d_o = ((a.dst_rows ? a.dst_rows[r] : r) * row_elems + el) * w;
    char* d = (char*)a.dst[f] + d_o;
    const char* sp = a.src[f] ? (const char*)a.src[f] + so : nullptr;
    if (w == 16) *reinterpret_cast<uint4*>(d) = (sp && !zero) ? *reinterpret_cast<const uint4*>(sp) : make_uint4(0, 0, 0, 0);
    else if (w == 4) *reinterpret_cast<uint32_t*>(d) = (sp && !zero) ? *reinterpret_cast<const uint32_t*>(sp) : 0u;
    else *d = (sp && !zero) ? *sp : (char)0;
  }
}

}  // namespace

extern "C" int seedhip_rows_move_masked(void* dst, const long long* dst_rows, const void* src,
                                        const long long* src_rows, long long n, long long row_bytes,
                                        const uint8_t* row_mask, int zero_where_masked, void* stream) {
  SEEDHIP_REQUIRE(n >= 0 && row_bytes >= 1, "rows_move_masked: bad n / row_bytes");
  if (n == 0) return SEEDHIP_OK;
  SEEDHIP_REQUIRE(dst && row_mask, "rows_move_masked: null dst / mask");
  hipStream_t s = (hipStream_t)stream;
  const uintptr_t al = (uintptr_t)dst | (uintptr_t)src | (uintptr_t)row_bytes;
  const long long total_bytes = n * row_bytes;
  auto grid = [](long long elems) { long long g = (elems + 255) / 256; return (int)(g > 8192 ? 8192 : g); };
  if ((al & 15) == 0)
    hipLaunchKernelGGL(rows_move_masked_kernel<uint4>, dim3(grid(total_bytes / 16)), dim3(256), 0, s, (uint4*)dst,
                       dst_rows, (const uint4*)src, src_rows, n, row_bytes / 16, row_mask, zero_where_masked);
  else if ((al & 3) == 0)
    hipLaunchKernelGGL(rows_move_masked_kernel<uint32_t>, dim3(grid(total_bytes / 4)), dim3(256), 0, s, (uint32_t*)dst,
                       dst_rows, (const uint32_t*)src, src_rows, n, row_bytes / 4, row_mask, zero_where_masked);
  else
    hipLaunchKernelGGL(rows_move_masked_kernel<uint8_t>, dim3(grid(total_bytes)), dim3(256), 0, s, (uint8_t*)dst,
                       dst_rows, (const uint8_t*)src, src_rows, n, row_bytes, row_mask, zero_where_masked);
  return seedhip::check_launch("rows_move_masked_kernel");
}

extern "C" int seedhip_inference_pre(const long long* env_ids, const long long* run_ids, const float* reward,
                                     const float* raw_reward, const uint8_t* done, int n, int num_envs,
                                     int num_action_repeats, long long* run_ids_table, long long* info_frames,
                                     float* info_return, float* info_raw_return, long long* actions_table,
                                     long long* store_index, uint8_t* reset_mask, long long* prev_actions,
                                     float* episode_stats, int stats_capacity, int* stats_count, int* error_flag,
                                     long long* ids_safe, uint8_t* valid, int* stamp_table, int* call_counter,
                                     uint8_t* will_complete, int full_length, void* stream) {
  SEEDHIP_REQUIRE(n >= 1 && n <= kMaxInferenceRows && num_envs >= 1, "inference_pre: need 1 <= n <= %d", kMaxInferenceRows);
  SEEDHIP_REQUIRE(env_ids && run_ids && reward && raw_reward && done && run_ids_table && info_frames && info_return &&
                  info_raw_return && actions_table && store_index && reset_mask && prev_actions && episode_stats &&
                  stats_count && error_flag && ids_safe && valid && stamp_table && call_counter,
                  "inference_pre: null pointer");
  PreArgs a{env_ids, run_ids, reward, raw_reward, done, n, num_envs, num_action_repeats, run_ids_table, info_frames,
            info_return, info_raw_return, actions_table, store_index, reset_mask, prev_actions, episode_stats,
            stats_capacity, stats_count, error_flag, ids_safe, valid, stamp_table, call_counter, will_complete,
            full_length};
  hipLaunchKernelGGL(inference_pre_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, a);
  return seedhip::check_launch("inference_pre_kernel");
}

extern "C" int seedhip_inference_post(const long long* env_ids, const uint8_t* valid, long long* actions,
                                      const float* policy_logits, int logits_ld, int num_actions,
                                      unsigned long long* rng_state, int n, int num_envs,
                                      int full_length, int batch_capacity, long long* store_index,
                                      long long* actions_table, int* batch_count, long long* append_rows,
                                      uint8_t* complete, uint8_t* carry, long long* batch_cols, long long* emit_env,
                                      long long* emit_col, int* emit_count, long long* last_rows,
                                      int* error_flag, const int* batch_start, void* stream) {
  SEEDHIP_REQUIRE(n >= 1 && n <= kMaxInferenceRows && num_envs >= 1 && full_length >= 2 && batch_capacity >= 1,
                  "inference_post: bad sizes");
  SEEDHIP_REQUIRE(env_ids && actions && store_index && actions_table && batch_count && append_rows && complete &&
                  carry && batch_cols && emit_env && emit_col && emit_count && last_rows && error_flag,
                  "inference_post: null pointer");
  SEEDHIP_REQUIRE(!policy_logits || (rng_state && num_actions >= 1 && logits_ld >= num_actions),
                  "inference_post: sampling needs rng_state and 1 <= num_actions <= logits_ld");
  PostArgs a{env_ids, valid, actions, policy_logits, logits_ld, num_actions, rng_state, n, num_envs, full_length,
             batch_capacity, store_index, actions_table, batch_count, append_rows, complete, carry, batch_cols,
             emit_env, emit_col, emit_count, batch_start, last_rows, error_flag};
  hipLaunchKernelGGL(inference_post_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, a);
  return seedhip::check_launch("inference_post_kernel");
}

extern "C" int seedhip_emit_unrolls(int nfields, void* const* dst, const void* const* src, const long long* row_bytes,
                                    const long long* emit_env, const long long* emit_col, const int* emit_count,
                                    int max_unrolls, int full_length, int num_envs, int batch_capacity, void* stream) {
  SEEDHIP_REQUIRE(nfields >= 1 && nfields <= kEmitFields, "emit_unrolls: need 1 <= nfields <= %d", kEmitFields);
  SEEDHIP_REQUIRE(dst && src && row_bytes && emit_env && emit_col && emit_count, "emit_unrolls: null pointer");
  SEEDHIP_REQUIRE(max_unrolls >= 1 && full_length >= 1 && num_envs >= 1 && batch_capacity >= 1, "emit_unrolls: bad sizes");
  EmitArgs a;
  for (int f = 0; f < nfields; ++f) {
    SEEDHIP_REQUIRE(dst[f] && src[f] && row_bytes[f] >= 1, "emit_unrolls: bad field %d", f);
    a.dst[f] = dst[f]; a.src[f] = src[f]; a.row_bytes[f] = row_bytes[f];
    const uintptr_t al = (uintptr_t)dst[f] | (uintptr_t)src[f] | (uintptr_t)row_bytes[f];
    a.w[f] = (al & 15) == 0 ? 16 : ((al & 3) == 0 ? 4 : 1);
  }
  a.nfields = nfields; a.env = emit_env; a.col = emit_col; a.count = emit_count;
  a.L = full_length; a.E = num_envs; a.cap = batch_capacity;
  long long grid = (long long)max_unrolls * full_length;      // upper bound of the work items; the kernel reads the count
  if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL(emit_unrolls_kernel, dim3((int)grid), dim3(256), 0, (hipStream_t)stream, a);
  return seedhip::check_launch("emit_unrolls_kernel");
}

extern "C" int seedhip_categorical_sample(const float* logits, int ld, long long rows, int num_actions,
                                          unsigned long long* rng_state, long long* actions, void* stream) {
  SEEDHIP_REQUIRE(rows >= 0 && num_actions >= 1 && ld >= num_actions, "categorical_sample: bad rows / num_actions / ld");
  SEEDHIP_REQUIRE(rng_state, "categorical_sample: null rng_state");
  hipStream_t s = (hipStream_t)stream;
  if (rows > 0) {
    SEEDHIP_REQUIRE(logits && actions, "categorical_sample: null pointer");
    hipLaunchKernelGGL(categorical_sample_kernel, dim3(seedhip::cdiv(rows, 256)), dim3(256), 0, s, logits, ld, rows,
                       num_actions, rng_state, actions);
  }
  hipLaunchKernelGGL(rng_advance_kernel, dim3(1), dim3(1), 0, s, rng_state);
  return seedhip::check_launch("categorical_sample_kernel");
}

extern "C" int seedhip_rows_move_ops(int nops, const seedhip_row_op* ops, void* stream) {
  SEEDHIP_REQUIRE(nops >= 0 && nops <= kMaxOps, "rows_move_ops: need 0 <= nops <= %d", kMaxOps);
  if (nops == 0) return SEEDHIP_OK;
  SEEDHIP_REQUIRE(ops, "rows_move_ops: null ops");
  OpsArgs a;
  int k = 0;
  long long blocks = 0;
  for (int f = 0; f < nops; ++f) {
    const seedhip_row_op& o = ops[f];
    SEEDHIP_REQUIRE(o.n >= 0 && o.row_bytes >= 1, "rows_move_ops: bad op %d", f);
    if (o.n == 0) continue;
    SEEDHIP_REQUIRE(o.dst, "rows_move_ops: null dst in op %d", f);
    SEEDHIP_REQUIRE(o.n < (1LL << 31) && o.row_bytes < (1LL << 31), "rows_move_ops: op %d too large", f);
    RowOp& r = a.op[k];
    r.dst = o.dst; r.src = o.src;
    r.dst_pitch = o.dst_pitch ? o.dst_pitch : o.row_bytes; r.src_pitch = o.src_pitch ? o.src_pitch : o.row_bytes;
    r.dst_rows = o.dst_rows; r.src_rows = o.src_rows; r.n = (unsigned)o.n; r.mask = o.row_mask;
    r.zero_where_masked = o.zero_where_masked;
    const uintptr_t al = (uintptr_t)o.dst | (uintptr_t)o.src | (uintptr_t)o.row_bytes | (uintptr_t)r.dst_pitch | (uintptr_t)r.src_pitch;
    r.w = (al & 15) == 0 ? 16u : ((al & 3) == 0 ? 4u : 1u);
    r.row_elems = (unsigned)(o.row_bytes / r.w);
    long long nb;
    if (r.row_elems > 256) { r.pieces = (r.row_elems + kPiece - 1) / kPiece; r.rows_per_wg = 0; nb = (long long)r.n * r.pieces; }
    else { r.pieces = 0; r.rows_per_wg = 256u / r.row_elems; nb = ((long long)r.n + r.rows_per_wg - 1) / r.rows_per_wg; }
    a.blk_start[k] = (int)blocks;
    blocks += nb;
    SEEDHIP_REQUIRE(blocks < (1LL << 31), "rows_move_ops: too many workgroups");
    ++k;
  }
  if (k == 0) return SEEDHIP_OK;
  a.blk_start[k] = (int)blocks; a.nops = k;
  hipLaunchKernelGGL(rows_move_ops_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
  return seedhip::check_launch("rows_move_ops_kernel");
}

extern "C" int seedhip_rows_move_multi(int nfields, void* const* dst, const void* const* src, const long long* row_bytes,
                                       const long long* dst_rows, const long long* src_rows, long long n,
                                       const uint8_t* row_mask, int zero_where_masked, void* stream) {
  SEEDHIP_REQUIRE(nfields >= 1 && nfields <= kMaxFields && n >= 0, "rows_move_multi: need 1 <= nfields <= %d", kMaxFields);
  if (n == 0) return SEEDHIP_OK;
  SEEDHIP_REQUIRE(dst && src && row_bytes, "rows_move_multi: null descriptor arrays");
  MultiArgs a;
  long long max_elems = 1;
  for (int f = 0; f < nfields; ++f) {
    SEEDHIP_REQUIRE(dst[f] && row_bytes[f] >= 1, "rows_move_multi: bad field %d", f);
    a.dst[f] = dst[f]; a.src[f] = src[f]; a.row_bytes[f] = row_bytes[f];
    const long long e = n * ((row_bytes[f] + 15) / 16);
    if (e > max_elems) max_elems = e;
  }
  a.dst_rows = dst_rows; a.src_rows = src_rows; a.n = n; a.mask = row_mask; a.zero_where_masked = zero_where_masked;
  long long gx = (max_elems + 255) / 256; if (gx > 2048) gx = 2048;
  hipLaunchKernelGGL(rows_move_multi_kernel, dim3((int)gx, nfields), dim3(256), 0, (hipStream_t)stream, a);
  return seedhip::check_launch("rows_move_multi_kernel");
}

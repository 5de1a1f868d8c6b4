// Central inference as SIX launches for the frame-stacked Atari agents.
//
// The `inference` tf.function of /root/reference/agents/vtrace/learner.py:350-405 on the device store, specialised for
// agents whose only recurrent state is the frame stack (atari/networks.py:57-173).  inference.hip serves ANY agent with
// 16 launches per batch (bookkeeping, previous-state row moves, unpack of the bit-packed stacking state, the agent's
// kernels, sampling, three rounds of row moves, re-pack of the state): at 1 024 rows the ten launches around the four
// GEMM-class kernels took 90 of 155 us.  Here:
//   serve_begin   ALL bookkeeping that does not depend on the network: id validation, run-id resets (:353-366), episode
//                 statistics (:373-378), previous action (:381), store index advance and completed-unroll detection
//                 (common/utils.py:187-194, 229-233), batch columns by an exclusive scan in env_ids order, the frame
//                 stack's validity counter -- and, on its other workgroups, the first conv's weights / 255 split into
//                 the three bf16 planes its kernel keeps in registers (once per call instead of once per workgroup).
//   [stackconv.hip: seedhip_conv2d_stack_fwd_rows]  first conv straight from the request's frames + the three previous
//                 frames WHERE THE UNROLL STORE ALREADY HOLDS THEM (the observation field of steps idx-1..idx-3), writing
//                 the new frame into the store on the way: the bit-packed per-env stacking state (28 KB per env, unpacked
//                 and re-packed every step by two extra HBM passes) does not exist on this path.
//   [second conv, Dense partial sums: the library's kernels]
//   serve_finish  Dense split-K reduction + bias + ReLU as the A operand of the packed policy / baseline heads (same
//                 MFMA sequence as heads.hip: bit-identical logits), Gumbel-max action sampling (dmlab/networks.py:122),
//                 the step's scalar fields appended to the store (utils.py:187-194), action table update (:403).
//   serve_emit    completed unrolls -> the time-major training batch (learner.py:396-397, 418-432), last step carried
//                 to slot 0 (utils.py:237-255), first agent state of the NEXT unroll packed from the store's frames
//                 (learner.py:398-399; atari/networks.py:164-169 bit order) -- only for the ~n / T rows that completed.
// The frame stack as (frames in the store, validity counter): stack channel c >= 1 of env e at store slot idx is the
// observation at slot idx - c (slots below 0 wrap to L - 1 + (idx - c): slot 0 is the carried copy of slot L - 1), valid
// iff c <= stack_valid[e] and the step is not `done`; stack_valid' = min(3, done ? 1 : stack_valid + 1).  Identical to
// the reference's zeroing by cumulative-OR done masks (networks.py:131-157) because validity is a prefix in c.
// Needs full_length >= 5 (a history slot must not be the slot being written).  All integer / byte work: exact.
#include "common.h"
#include "../../include/seedhip.h"

namespace {

typedef float f32x4_t __attribute__((ext_vector_type(4)));
constexpr int kMaxRows = 65536;

// LDS accesses of one wave execute in program order; this keeps the compiler from moving the wave's later reads above
// its stores (stackconv.hip: wave_lds_fence).
__device__ __forceinline__ void seedhip_wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ------------------------------------------------------------------------------------------------------------------ //
// serve_begin
// ------------------------------------------------------------------------------------------------------------------ //
struct BeginArgs {
  seedhip_serve_step s; const float* w0; int cout0; uint4* w0_split;
  const float* heads_w; int feat, ldh; float4* heads_image;
};

// Packed heads W [feat][ldh] -> the B-operand register image of serve_finish's MFMA chain (v_mfma_f32_16x16x4_f32):
// image[blk][tile][lane] = W[16 blk + 4 (lane >> 4) + s][16 tile + (lane & 15)], s = 0..3 (zero for columns >= ldh) --
// heads_fwd_kernel's transposed LDS tile, element for element, without the LDS.
__device__ __forceinline__ void image_heads(const BeginArgs& a, int item) {
  const int lane = item & 63, tile = (item >> 6) & 1, blk = item >> 7;
  const int n = 16 * tile + (lane & 15), k0 = 16 * blk + 4 * (lane >> 4);
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (n < a.ldh) {
    v.x = a.heads_w[(k0 + 0) * a.ldh + n]; v.y = a.heads_w[(k0 + 1) * a.ldh + n];
    v.z = a.heads_w[(k0 + 2) * a.ldh + n]; v.w = a.heads_w[(k0 + 3) * a.ldh + n];
  }
  a.heads_image[item] = v;
}

// W / 255 -> three bf16 parts by truncation, in the register image of stackconv_fwd_bf16r_kernel:
// image[slice][G][part][lane] (uint4 = 8 bf16: kx = 0..7 of row ky = 4 (G & 1) + (lane >> 4), stack channel G >> 1,
// output channel 16 slice + (lane & 15)).  Same arithmetic as that kernel's own prologue: bit-identical operands.
__device__ __forceinline__ void split_conv0(const BeginArgs& a, int item) {
  const int lane = item & 63, G = (item >> 6) & 7, slice = item >> 9;
  const int kq = lane >> 4, j = lane & 15;
  const int c = G >> 1, ky = 4 * (G & 1) + kq;
  uint32_t part[3][4];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float w = a.w0[((ky * 8 + e) * 4 + c) * a.cout0 + 16 * slice + j] / 255.0f;
    const uint32_t hi = __float_as_uint(w) >> 16;
    const float r1 = w - __uint_as_float(hi << 16);
    const uint32_t mid = __float_as_uint(r1) >> 16;
    const uint32_t lo = __float_as_uint(r1 - __uint_as_float(mid << 16)) >> 16;
    const uint32_t v[3] = {hi, mid, lo};
#pragma unroll
    for (int s3 = 0; s3 < 3; ++s3) {
      if (e & 1) part[s3][e >> 1] |= v[s3] << 16; else part[s3][e >> 1] = v[s3];
    }
  }
#pragma unroll
  for (int s3 = 0; s3 < 3; ++s3)
    a.w0_split[((slice * 8 + G) * 3 + s3) * 64 + lane] = make_uint4(part[s3][0], part[s3][1], part[s3][2], part[s3][3]);
}

__global__ void __launch_bounds__(256)
split_conv0_kernel(const BeginArgs a) {
  const int items = (a.cout0 / 16) * 8 * 64;
  for (int it = blockIdx.x * 256 + threadIdx.x; it < items; it += gridDim.x * 256) split_conv0(a, it);
}

__global__ void __launch_bounds__(1024)
serve_begin_kernel(const BeginArgs a) {
  const seedhip_serve_step& s = a.s;
  const int tid = threadIdx.x;
  if (blockIdx.x > 0) {                                                // the first conv's weight planes, the heads' image
    const int items0 = a.w0_split ? (a.cout0 / 16) * 8 * 64 : 0, items1 = a.heads_image ? (a.feat / 16) * 128 : 0;
    for (int it = (blockIdx.x - 1) * 1024 + tid; it < items0 + items1; it += (gridDim.x - 1) * 1024) {
      if (it < items0) split_conv0(a, it); else image_heads(a, it - items0);
    }
    return;
  }
  __shared__ int s_wave[16];
  __shared__ int s_base, s_start, s_stamp;
  const int lane = tid & 63, wave = tid >> 6;
  const int L = s.full_length, E = s.num_envs;
  if (tid == 0) {
    s_stamp = *s.call_counter + 1;
    s_base = *s.batch_count;
    s_start = s.batch_start ? *s.batch_start : 0;
    if (s.rng_state) {                                                 // the sampler's (seed, call) of THIS step
      const unsigned long long seed = s.rng_state[0], call = s.rng_state[1];
      s.rng_snapshot[0] = seed; s.rng_snapshot[1] = call;
      s.rng_state[1] = call + 1;
    }
  }
  __syncthreads();
  const int stamp = s_stamp, base = s_base, start = s_start;
  if (tid == 0) *s.call_counter = stamp;
  int running = 0;                                                     // completions in the chunks before this one
  for (int c0 = 0; c0 < s.n; c0 += 1024) {
    const int i = c0 + tid;
    int done_unroll = 0, nvp = 0;
    long long e = 0;
    if (i < s.n) {
      e = s.env_ids[i];
      bool ok = e >= 0 && e < E;
      if (!ok) { atomicOr(s.error_flag, 1); e = 0; }
      // duplicate ids in one batch are an error in the reference (utils.py:173-176): the first occurrence stamps the
      // env's slot, a later one finds this call's stamp there
      else if (atomicExch(s.stamp_table + e, stamp) == stamp) { atomicOr(s.error_flag, 2); ok = false; }
      s.ids_safe[i] = e;
      s.valid[i] = ok ? 1 : 0;
      if (!ok) {                                                       // masked out of every table access of the step
        s.prev_actions[i] = 0; s.append_rows[i] = -1; s.nvalid[i] = 1; s.prev_valid[i] = 0;
#pragma unroll
        for (int c = 0; c < 4; ++c) s.hist_rows[4 * i + c] = 0;
      } else {
        const long long prev_run = s.run_ids_table[e], run = s.run_ids[i];
        s.run_ids_table[e] = run;                                      // learner.py:354
        const bool reset = prev_run != run;                            // :355-357
        long long frames = s.info_frames[e];
        float ret = s.info_return[e], raw = s.info_raw_return[e];
        long long act = s.actions_table[e];
        long long idx = s.store_index[e];
        nvp = s.stack_valid[e];
        if (reset) {                                                   // :360-366
          frames = 0; ret = 0.f; raw = 0.f; act = 0; idx = 0; nvp = 0;
          s.first_zero[e] = 1;                                         // the next unroll starts from the initial state
        }
        ret += s.reward[i]; raw += s.raw_reward[i];                    // :373
        const bool done = s.done[i] != 0;
        if (done) {                                                    // :374-377
          const int slot = atomicAdd(s.stats_count, 1);
          if (slot < s.stats_capacity) {
            s.episode_stats[3 * slot + 0] = (float)frames;
            s.episode_stats[3 * slot + 1] = ret;
            s.episode_stats[3 * slot + 2] = raw;
          }
          frames = 0; ret = 0.f; raw = 0.f;
        }
        frames += s.num_action_repeats;                                // :378
        s.info_frames[e] = frames; s.info_return[e] = ret; s.info_raw_return[e] = raw;
        s.prev_actions[i] = act;                                       // :381
        // frame stack (atari/networks.py:131-157 as a counter)
        const int nv = done ? 1 : 1 + nvp;
        s.nvalid[i] = (uint8_t)nv;
        s.prev_valid[i] = (uint8_t)nvp;
        s.stack_valid[e] = (uint8_t)(nv < 3 ? nv : 3);
        // store row of this step and of the three before it
        s.append_rows[i] = idx * E + e;
        s.hist_rows[4 * i] = idx * E + e;
#pragma unroll
        for (int c = 1; c < 4; ++c) {
          long long sl = idx - c;
          if (sl < 0) sl += L - 1;
          s.hist_rows[4 * i + c] = sl * E + e;
        }
        done_unroll = (idx + 1 == L) ? 1 : 0;
        if (idx + 1 > L) atomicOr(s.error_flag, 4);
        s.store_index[e] = done_unroll ? 1 : idx + 1;                  // utils.py:194, 254-255 (overlap 0)
      }
    }
    // rank of a completed unroll among the batch's completions, in env_ids order (tf.gather(env_ids, tf.where(..)))
    const unsigned long long votes = __ballot(done_unroll);
    const int in_wave = __popcll(votes & ((2ull << lane) - 1ull));
    if (lane == 0) s_wave[wave] = __popcll(votes);
    __syncthreads();
    int before = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) { const int v = s_wave[w]; total += v; if (w < wave) before += v; }
    __syncthreads();
    if (done_unroll) {
      const int rank = running + before + in_wave - 1;
      const int pos = base + rank;
      // an unroll that finds the training batch full is dropped (flag 8); its last step is still carried to slot 0 and
      // the next unroll's first state is still set aside (column -1)
      int col = start + pos;
      if (col >= s.batch_capacity) col -= s.batch_capacity;
      s.emit_env[rank] = e;
      s.emit_col[rank] = pos < s.batch_capacity ? col : -1;
      s.emit_row[rank] = i;
    }
    running += total;
  }
  if (tid == 0) {
    *s.batch_count = base + running > s.batch_capacity ? s.batch_capacity : base + running;
    if (base + running > s.batch_capacity) atomicOr(s.error_flag, 8);
    *s.emit_count = running;
  }
}

// ------------------------------------------------------------------------------------------------------------------ //
// serve_finish: one workgroup (sixteen waves) per 16 rows.
//   phase 1, all waves: x = relu(sum_z partial[z] + bias) for the tile's 16 rows, dense_epilogue_kernel's order (conv.hip),
//            into LDS -- at 1 024 rows the Dense layer's plan has 14 K-slices: one 16-byte column group per thread, eight
//            slices in flight;
//   phase 2, wave 0: the packed heads, heads_fwd_kernel's MFMA sequence (heads.hip: bit-identical logits), A operand
//            from LDS, B operand from the register image serve_begin wrote;
//   phase 3, wave 0: Gumbel-max sampling, four lanes per row (a lane takes every fourth Philox block; the first
//            maximum wins across lanes as in sample_categorical_row), append of the scalar fields, action table.
// The workgroups behind the tiles append the request frames (four each).
// ------------------------------------------------------------------------------------------------------------------ //
constexpr int kMaxFeat = 512, kMaxN = 32, kHeadPitch = 33, kFinThreads = 1024, kFramesPerCopy = 4;
struct FinishArgs {
  seedhip_serve_step s; seedhip_serve_fields f;
  const float* partial; int slices; const float* fc_bias; int feat;
  const float4* heads_image; const float* heads_b; int ldh, A;
  long long* actions;
  const uint8_t* obs; uint8_t* store_obs; long long hw;
};

__global__ void __launch_bounds__(kFinThreads)
serve_finish_kernel(const FinishArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const seedhip_serve_step& s = a.s;
  const int F = a.feat, LDX = F + 4;
  float* x = smem;                                                     // [16][F + 4]
  float* head = smem + 16 * LDX;                                       // [16][33] (+ pad to a 16-byte multiple)
  float4* img_lds = reinterpret_cast<float4*>(head + 16 * kHeadPitch + 16);   // [F / 16][2][64]: the heads' register image
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, kq = lane >> 4;
  const long long rows = s.n, total = rows * F;
  const int tiles = (int)((rows + 15) / 16);
  if ((int)blockIdx.x >= tiles) {
    // the workgroups behind the tiles append the request frames to the store (common/utils.py:187-194, the largest
    // field): kFramesPerCopy rows each, every thread's loads in flight before its first store
    const long long vec = a.hw >> 4;
    const long long i0 = (long long)((int)blockIdx.x - tiles) * kFramesPerCopy;
    constexpr int kMaxV = 2;                                           // vectors per thread and frame held in registers
    for (int fr = 0; fr < kFramesPerCopy; ++fr) {
      const long long i = i0 + fr;
      if (i >= rows) break;
      const long long arow = s.append_rows[i];
      if (arow < 0) continue;
      const uint4* src = reinterpret_cast<const uint4*>(a.obs + i * a.hw);
      uint4* dst = reinterpret_cast<uint4*>(a.store_obs + arow * a.hw);
      for (long long k0 = tid; k0 < vec; k0 += (long long)kMaxV * kFinThreads) {
        uint4 v[kMaxV];
#pragma unroll
        for (int q = 0; q < kMaxV; ++q) if (k0 + (long long)q * kFinThreads < vec) v[q] = src[k0 + (long long)q * kFinThreads];
#pragma unroll
        for (int q = 0; q < kMaxV; ++q) if (k0 + (long long)q * kFinThreads < vec) dst[k0 + (long long)q * kFinThreads] = v[q];
      }
    }
    return;
  }
  const long long row0 = (long long)blockIdx.x * 16;
  // the heads' image -> LDS by every wave (wave 0's MFMA chain read it from global memory: one L2 round trip per
  // 16-k block, sixteen in a row -- 12 of the kernel's 18 us)
  for (int idx = tid; idx < (F / 16) * 128; idx += kFinThreads) img_lds[idx] = a.heads_image[idx];
  {
    const int fq = F >> 2;
    for (int idx = tid; idx < 16 * fq; idx += kFinThreads) {
      const int rl = idx / fq, c = 4 * (idx - rl * fq);
      long long r = row0 + rl;
      if (r >= rows) r = rows - 1;
      const float* src = a.partial + r * F + c;
      f32x4_t v = *reinterpret_cast<const f32x4_t*>(src);
      for (int z0 = 1; z0 < a.slices; z0 += 8) {                       // eight slices in flight, summed in slice order
        f32x4_t t[8];
#pragma unroll
        for (int q = 0; q < 8; ++q)
          if (z0 + q < a.slices) t[q] = *reinterpret_cast<const f32x4_t*>(src + (long long)(z0 + q) * total);
#pragma unroll
        for (int q = 0; q < 8; ++q)
          if (z0 + q < a.slices) v += t[q];
      }
      v += *reinterpret_cast<const f32x4_t*>(a.fc_bias + c);
#pragma unroll
      for (int q = 0; q < 4; ++q) if (v[q] < 0.f) v[q] = 0.f;
      *reinterpret_cast<f32x4_t*>(x + rl * LDX + c) = v;
    }
  }
  __syncthreads();
  if (wave != 0) return;
  const int ntiles = a.ldh > 16 ? 2 : 1;
  float b0 = 0.f, b1 = 0.f;
  if (a.heads_b) { b0 = j < a.ldh ? a.heads_b[j] : 0.f; b1 = 16 + j < a.ldh ? a.heads_b[16 + j] : 0.f; }
  f32x4_t acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  const float* xr = x + j * LDX + 4 * kq;
  const float4* img = img_lds + lane;
  // what the row's lanes write at the end is requested now: one memory round trip under the MFMA chain, not behind it
  const int rl = lane >> 2, part = lane & 3;
  const long long i = row0 + rl;
  const bool live = i < rows;
  long long arow = -1, env = 0, pact = 0;
  float rew = 0.f; uint8_t dn = 0, ab = 0; int es = 0;
  unsigned long long seed = 0, call = 0;
  if (live) {
    arow = s.append_rows[i]; env = s.ids_safe[i]; pact = s.prev_actions[i]; rew = s.reward[i]; dn = s.done[i];
    ab = s.abandoned ? s.abandoned[i] : (uint8_t)0; es = s.episode_step ? s.episode_step[i] : 0;
    seed = s.rng_snapshot[0]; call = s.rng_snapshot[1];
  }
  // operands of block blk + 1 are read from LDS before block blk's MFMAs are issued (the loop did not unroll: every block
  // waited out its own LDS round trip)
  const int nblk = F / 16;
  f32x4_t v_n = *reinterpret_cast<const f32x4_t*>(xr);
  float4 bw0_n = img[0], bw1_n = img[64];
  for (int blk = 0; blk < nblk; ++blk) {
    const f32x4_t v = v_n;
    const float4 bw0 = bw0_n, bw1 = bw1_n;
    if (blk + 1 < nblk) {
      v_n = *reinterpret_cast<const f32x4_t*>(xr + 16 * (blk + 1));
      bw0_n = img[(2 * blk + 2) * 64];
      bw1_n = img[(2 * blk + 3) * 64];
    }
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(v[0], bw0.x, acc0, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(v[1], bw0.y, acc0, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(v[2], bw0.z, acc0, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(v[3], bw0.w, acc0, 0, 0, 0);
    if (ntiles == 2) {
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(v[0], bw1.x, acc1, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(v[1], bw1.y, acc1, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(v[2], bw1.z, acc1, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(v[3], bw1.w, acc1, 0, 0, 0);
    }
  }
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    head[(4 * kq + rr) * kHeadPitch + j] = acc0[rr] + b0;
    if (ntiles == 2) head[(4 * kq + rr) * kHeadPitch + 16 + j] = acc1[rr] + b1;
  }
  seedhip_wave_lds_fence();
  // lane (row = lane >> 2, part = lane & 3): Philox blocks part, part + 4, ... of the row (actions 4 blk .. 4 blk + 3)
  const float* hr = head + rl * kHeadPitch;
  float best = -INFINITY;
  int arg = 0x7fffffff;
  if (live) {
    const uint2 key = make_uint2((uint32_t)seed, (uint32_t)(seed >> 32));
    for (int a0 = 4 * part; a0 < a.A; a0 += 16) {
      const uint4 r = seedhip::philox4x32_10(make_uint4((uint32_t)call, (uint32_t)(call >> 32), (unsigned)i, (uint32_t)(a0 >> 2)), key);
      const uint32_t xs[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int ac = a0 + q;
        if (ac < a.A) {
          const float u = ((float)(xs[q] >> 9) + 0.5f) * (1.0f / 8388608.0f);
          const float v = hr[ac] - logf(-logf(u));
          if (v > best) { best = v; arg = ac; }
        }
      }
    }
  }
#pragma unroll
  for (int o = 1; o <= 2; o <<= 1) {                                   // first maximum over the row's four lanes
    const float ob = __shfl_xor(best, o, 64);
    const int oa = __shfl_xor(arg, o, 64);
    if (ob > best || (ob == best && oa < arg)) { best = ob; arg = oa; }
  }
  if (arg == 0x7fffffff) arg = 0;                                      // every candidate was NaN / -inf: sample_categorical_row's 0
  if (live) {
    if (part == 0) {
      const long long act = arg;
      a.actions[i] = act;
      if (arow >= 0) {
        s.actions_table[env] = act;
        a.f.prev_actions[arow] = pact;
        a.f.reward[arow] = rew;
        a.f.done[arow] = dn;
        a.f.abandoned[arow] = ab;
        a.f.episode_step[arow] = es;
        a.f.action[arow] = act;
        a.f.baseline[arow] = hr[a.A];
      }
    }
    if (arow >= 0)
      for (int c = part; c < a.A; c += 4) a.f.policy_logits[arow * a.A + c] = hr[c];
  }
}

// ------------------------------------------------------------------------------------------------------------------ //
// serve_emit: work item (r, u), u < L: u = 0 moves slot 0 AND slot L - 1 of completed unroll r (the latter also to slot 0:
// the carry; same thread, same element, in program order -- the one place where a row is read and rewritten), u = 1 ..
// L - 2 moves slot u, u = L - 1 hands the first agent state over and packs the next one.
// ------------------------------------------------------------------------------------------------------------------ //
constexpr int kEmitFields = 16, kEmitThreads = 256, kBigRow = 256, kStateParts = 1;
struct EmitArgs {
  seedhip_serve_step s;
  void* batch[kEmitFields]; void* store[kEmitFields]; long long row_bytes[kEmitFields]; int w[kEmitFields]; int nfields;
  int* first_table; int* batch_first; const uint8_t* store_obs; long long hw;
};

// One row of every field, source row rs -> destination rows rd0 / rd1 of d0 / d1 (a null table = no such copy).  Every
// load of the row is issued before the first store (field after field, each waiting for its own round trip, the nine
// fields of the V-trace unroll took nine memory latencies per item): small fields one unit (16 / 4 / 1 bytes) per thread
// across ALL small fields at once, big fields (frames) two vectors per thread in flight.
__device__ __forceinline__ void move_fields(const EmitArgs& a, long long rs, bool to0, long long rd0, bool to1, long long rd1,
                                            int tid) {
  // small fields: thread -> (field, unit)
  {
    int f = -1, k = 0, base = 0;
    for (int g = 0; g < a.nfields; ++g) {
      if (a.row_bytes[g] >= kBigRow) continue;
      const int units = (int)(a.row_bytes[g] / a.w[g]);
      if (f < 0 && tid < base + units) { f = g; k = tid - base; }
      base += units;
    }
    if (f >= 0) {
      const long long rb = a.row_bytes[f];
      const char* sp = (const char*)a.store[f] + rs * rb;
      char* p0 = to0 ? (char*)a.batch[f] + rd0 * rb : nullptr;
      char* p1 = to1 ? (char*)a.store[f] + rd1 * rb : nullptr;
      if (a.w[f] == 16) { const uint4 v = reinterpret_cast<const uint4*>(sp)[k]; if (p0) reinterpret_cast<uint4*>(p0)[k] = v; if (p1) reinterpret_cast<uint4*>(p1)[k] = v; }
      else if (a.w[f] == 4) { const uint32_t v = reinterpret_cast<const uint32_t*>(sp)[k]; if (p0) reinterpret_cast<uint32_t*>(p0)[k] = v; if (p1) reinterpret_cast<uint32_t*>(p1)[k] = v; }
      else { const unsigned char v = reinterpret_cast<const unsigned char*>(sp)[k]; if (p0) reinterpret_cast<unsigned char*>(p0)[k] = v; if (p1) reinterpret_cast<unsigned char*>(p1)[k] = v; }
    }
  }
  for (int f = 0; f < a.nfields; ++f) {
    const long long rb = a.row_bytes[f];
    if (rb < kBigRow) continue;
    const char* sp = (const char*)a.store[f] + rs * rb;
    char* p0 = to0 ? (char*)a.batch[f] + rd0 * rb : nullptr;
    char* p1 = to1 ? (char*)a.store[f] + rd1 * rb : nullptr;
    if (a.w[f] == 16) {
      const long long nv = rb >> 4;
      for (long long k0 = tid; k0 < nv; k0 += 2 * kEmitThreads) {
        const bool two = k0 + kEmitThreads < nv;
        const uint4 v0 = reinterpret_cast<const uint4*>(sp)[k0];
        uint4 v1 = make_uint4(0, 0, 0, 0);
        if (two) v1 = reinterpret_cast<const uint4*>(sp)[k0 + kEmitThreads];
        if (p0) { reinterpret_cast<uint4*>(p0)[k0] = v0; if (two) reinterpret_cast<uint4*>(p0)[k0 + kEmitThreads] = v1; }
        if (p1) { reinterpret_cast<uint4*>(p1)[k0] = v0; if (two) reinterpret_cast<uint4*>(p1)[k0 + kEmitThreads] = v1; }
      }
    } else {
      const int w = a.w[f];
      const long long nu = rb / w;
      for (long long k = tid; k < nu; k += kEmitThreads) {
        if (w == 4) { const uint32_t v = reinterpret_cast<const uint32_t*>(sp)[k]; if (p0) reinterpret_cast<uint32_t*>(p0)[k] = v; if (p1) reinterpret_cast<uint32_t*>(p1)[k] = v; }
        else { const unsigned char v = reinterpret_cast<const unsigned char*>(sp)[k]; if (p0) reinterpret_cast<unsigned char*>(p0)[k] = v; if (p1) reinterpret_cast<unsigned char*>(p1)[k] = v; }
      }
    }
  }
}

__global__ void __launch_bounds__(kEmitThreads)
serve_emit_kernel(const EmitArgs a) {
  const seedhip_serve_step& s = a.s;
  const int L = s.full_length, E = s.num_envs, cap = s.batch_capacity, tid = threadIdx.x;
  const int per = L - 1 + kStateParts;                                 // items per completed unroll
  const int items = *s.emit_count * per;
  for (int it = blockIdx.x; it < items; it += gridDim.x) {
    const int r = it / per, u = it - r * per;
    const long long e = s.emit_env[r], col = s.emit_col[r];
    if (u >= L - 1) {
      // learner.py:396-399: the unroll's first agent state goes with it; the state BEFORE this step (the three frames
      // in front of slot L - 1, zero outside the episode: atari/networks.py:164-169, MSB = newest) becomes the next one's
      constexpr int part = 0;
      const int i = s.emit_row[r];
      const int nvp = s.prev_valid[i];
      const bool zero_first = s.first_zero[e] != 0;
      uint4* ft = reinterpret_cast<uint4*>(a.first_table + e * a.hw);
      uint4* bf = col >= 0 ? reinterpret_cast<uint4*>(a.batch_first + col * a.hw) : nullptr;
      const unsigned* f1 = reinterpret_cast<const unsigned*>(a.store_obs + s.hist_rows[4 * i + 1] * a.hw);
      const unsigned* f2 = reinterpret_cast<const unsigned*>(a.store_obs + s.hist_rows[4 * i + 2] * a.hw);
      const unsigned* f3 = reinterpret_cast<const unsigned*>(a.store_obs + s.hist_rows[4 * i + 3] * a.hw);
      const long long nq = a.hw / 4, lo = nq * part / kStateParts, hi = nq * (part + 1) / kStateParts;
      for (long long k0 = lo + tid; k0 < hi; k0 += 2 * kEmitThreads) {
        const long long k1 = k0 + kEmitThreads;
        const bool two = k1 < hi;
        uint4 o0 = make_uint4(0, 0, 0, 0), o1 = o0;
        if (bf && !zero_first) { o0 = ft[k0]; if (two) o1 = ft[k1]; }
        unsigned x[2][3] = {{0u, 0u, 0u}, {0u, 0u, 0u}};
        if (nvp > 0) { x[0][0] = f1[k0]; if (two) x[1][0] = f1[k1]; }
        if (nvp > 1) { x[0][1] = f2[k0]; if (two) x[1][1] = f2[k1]; }
        if (nvp > 2) { x[0][2] = f3[k0]; if (two) x[1][2] = f3[k1]; }
        if (bf) { bf[k0] = o0; if (two) bf[k1] = o1; }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          if (h == 1 && !two) break;
          unsigned o[4];
#pragma unroll
          for (int q = 0; q < 4; ++q)
            o[q] = (((x[h][0] >> (8 * q)) & 0xFFu) << 16) | (((x[h][1] >> (8 * q)) & 0xFFu) << 8) | ((x[h][2] >> (8 * q)) & 0xFFu);
          ft[h ? k1 : k0] = make_uint4(o[0], o[1], o[2], o[3]);
        }
      }
      __syncthreads();                                                 // every thread has read first_zero[e]
      if (tid == 0) s.first_zero[e] = 0;
      continue;
    }
    if (u == 0) {
      // slot 0 -> batch slot 0; slot L - 1 -> batch slot L - 1 AND slot 0 (the carry, utils.py:237-255): same thread, same
      // element, in program order -- the one place where a row is read and rewritten
      move_fields(a, e, col >= 0, col, false, 0, tid);
      move_fields(a, (long long)(L - 1) * E + e, col >= 0, (long long)(L - 1) * cap + col, true, e, tid);
    } else if (col >= 0) {
      move_fields(a, (long long)u * E + e, true, (long long)u * cap + col, false, 0, tid);
    }
  }
}

}  // namespace

static int check_step(const seedhip_serve_step* s, const char* what) {
  SEEDHIP_REQUIRE(s, "%s: null step", what);
  SEEDHIP_REQUIRE(s->n >= 1 && s->n <= kMaxRows && s->num_envs >= 1 && s->full_length >= 5 && s->batch_capacity >= 1,
                  "%s: need 1 <= n <= %d, num_envs >= 1, full_length >= 5, batch_capacity >= 1", what, kMaxRows);
  SEEDHIP_REQUIRE(s->env_ids && s->run_ids && s->reward && s->raw_reward && s->done && s->run_ids_table && s->info_frames &&
                  s->info_return && s->info_raw_return && s->actions_table && s->store_index && s->stack_valid &&
                  s->first_zero && s->stamp_table && s->call_counter && s->episode_stats && s->stats_count &&
                  s->error_flag && s->batch_count && s->ids_safe && s->valid && s->prev_actions && s->append_rows &&
                  s->hist_rows && s->nvalid && s->prev_valid && s->emit_env && s->emit_col && s->emit_row &&
                  s->emit_count && s->rng_state && s->rng_snapshot, "%s: null pointer in the step", what);
  return SEEDHIP_OK;
}

extern "C" size_t seedhip_serve_conv0_split_bytes(int cout) { return cout > 0 ? (size_t)(cout / 16) * 8 * 3 * 64 * 16 : 0; }

extern "C" int seedhip_serve_split_conv0(const float* conv0_w, int conv0_cout, void* conv0_split, void* stream) {
  SEEDHIP_REQUIRE(conv0_w && conv0_split && conv0_cout >= 16 && conv0_cout % 16 == 0 && (((uintptr_t)conv0_split) & 15) == 0,
                  "serve_split_conv0: need conv0_w, cout %% 16 == 0 and a 16-byte aligned buffer");
  BeginArgs a;
  memset(&a, 0, sizeof(a));
  a.w0 = conv0_w; a.cout0 = conv0_cout; a.w0_split = (uint4*)conv0_split;
  hipLaunchKernelGGL(split_conv0_kernel, dim3(conv0_cout / 16 * 2), dim3(256), 0, (hipStream_t)stream, a);
  return seedhip::check_launch("split_conv0_kernel");
}

extern "C" size_t seedhip_serve_heads_image_bytes(int feat) { return feat > 0 ? (size_t)(feat / 16) * 128 * 16 : 0; }

extern "C" int seedhip_serve_begin(const seedhip_serve_step* step, const float* conv0_w, int conv0_cout, void* conv0_split,
                                   const float* heads_w, int feat, int ldh, void* heads_image, void* stream) {
  int rc = check_step(step, "serve_begin"); if (rc) return rc;
  SEEDHIP_REQUIRE(!conv0_split || (conv0_w && conv0_cout >= 16 && conv0_cout % 16 == 0 && (((uintptr_t)conv0_split) & 15) == 0),
                  "serve_begin: the weight planes need conv0_w, cout %% 16 == 0 and a 16-byte aligned buffer");
  SEEDHIP_REQUIRE(!heads_image || (heads_w && feat >= 64 && feat <= kMaxFeat && feat % 64 == 0 && ldh >= 4 && ldh <= kMaxN &&
                                   ldh % 4 == 0 && (((uintptr_t)heads_image) & 15) == 0),
                  "serve_begin: the heads image needs heads_w, feat %% 64 == 0, feat <= 512, ldh %% 4 == 0, ldh <= 32 and a 16-byte aligned buffer");
  BeginArgs a{*step, conv0_w, conv0_cout, (uint4*)conv0_split, heads_w, feat, ldh, (float4*)heads_image};
  const int work = (conv0_split ? conv0_cout / 16 * 512 : 0) + (heads_image ? feat / 16 * 128 : 0);
  const int extra = (work + 1023) / 1024;
  hipLaunchKernelGGL(serve_begin_kernel, dim3(1 + extra), dim3(1024), 0, (hipStream_t)stream, a);
  return seedhip::check_launch("serve_begin_kernel");
}

extern "C" int seedhip_serve_finish(const seedhip_serve_step* step, const seedhip_serve_fields* fields,
                                    const float* fc_partial, int slices, const float* fc_bias, int feat,
                                    const void* heads_image, const float* heads_b, int ldh, int num_actions,
                                    long long* actions, const uint8_t* obs, uint8_t* store_obs, long long hw,
                                    void* stream) {
  int rc = check_step(step, "serve_finish"); if (rc) return rc;
  SEEDHIP_REQUIRE(fields && fields->prev_actions && fields->reward && fields->done && fields->abandoned &&
                  fields->episode_step && fields->action && fields->policy_logits && fields->baseline,
                  "serve_finish: null store field");
  SEEDHIP_REQUIRE(fc_partial && fc_bias && heads_image && actions && slices >= 1, "serve_finish: null pointer / slices < 1");
  SEEDHIP_REQUIRE(feat >= 64 && feat <= kMaxFeat && feat % 64 == 0 && ldh >= 4 && ldh <= kMaxN && ldh % 4 == 0 &&
                  num_actions >= 1 && num_actions < ldh, "serve_finish: need feat %% 64 == 0, feat <= 512, ldh %% 4 == 0, ldh <= 32, num_actions < ldh");
  SEEDHIP_REQUIRE((((uintptr_t)fc_partial | (uintptr_t)fc_bias | (uintptr_t)heads_image) & 15) == 0,
                  "serve_finish: 16-byte aligned partial sums / bias / heads image");
  SEEDHIP_REQUIRE(obs && store_obs && hw >= 16 && hw % 16 == 0 && ((((uintptr_t)obs) | ((uintptr_t)store_obs)) & 15) == 0,
                  "serve_finish: frames must be 16-byte aligned rows of hw %% 16 == 0 bytes");
  FinishArgs a{*step, *fields, fc_partial, slices, fc_bias, feat, (const float4*)heads_image, heads_b, ldh, num_actions, actions,
               obs, store_obs, hw};
  const size_t lds = ((size_t)16 * (feat + 4) + 16 * kHeadPitch + 16) * sizeof(float) + (size_t)(feat / 16) * 128 * 16;
  // n <= 65536: one workgroup per 16 rows, then one per kFramesPerCopy frames to append
  const int grid = (step->n + 15) / 16 + (step->n + kFramesPerCopy - 1) / kFramesPerCopy;
  hipLaunchKernelGGL(serve_finish_kernel, dim3(grid), dim3(kFinThreads), lds, (hipStream_t)stream, a);
  return seedhip::check_launch("serve_finish_kernel");
}

extern "C" int seedhip_serve_emit(const seedhip_serve_step* step, int nfields, void* const* batch, void* const* store,
                                  const long long* row_bytes, int* first_table, int* batch_first,
                                  const uint8_t* store_obs, long long hw, void* stream) {
  int rc = check_step(step, "serve_emit"); if (rc) return rc;
  SEEDHIP_REQUIRE(nfields >= 1 && nfields <= kEmitFields && batch && store && row_bytes, "serve_emit: need 1 <= nfields <= %d", kEmitFields);
  SEEDHIP_REQUIRE(first_table && batch_first && store_obs && hw >= 16 && hw % 16 == 0 &&
                  ((((uintptr_t)first_table) | ((uintptr_t)batch_first) | ((uintptr_t)store_obs)) & 15) == 0,
                  "serve_emit: first-state tables / frames must be 16-byte aligned with hw %% 16 == 0");
  EmitArgs a;
  a.s = *step;
  for (int f = 0; f < nfields; ++f) {
    SEEDHIP_REQUIRE(batch[f] && store[f] && row_bytes[f] >= 1, "serve_emit: bad field %d", f);
    a.batch[f] = batch[f]; a.store[f] = store[f]; a.row_bytes[f] = row_bytes[f];
    const uintptr_t al = (uintptr_t)batch[f] | (uintptr_t)store[f] | (uintptr_t)row_bytes[f];
    a.w[f] = (al & 15) == 0 ? 16 : ((al & 3) == 0 ? 4 : 1);
  }
  a.nfields = nfields; a.first_table = first_table; a.batch_first = batch_first; a.store_obs = store_obs; a.hw = hw;
  int small_units = 0;
  for (int f = 0; f < nfields; ++f) if (row_bytes[f] < kBigRow) small_units += (int)(row_bytes[f] / a.w[f]);
  SEEDHIP_REQUIRE(small_units <= kEmitThreads, "serve_emit: the fields below %d bytes per row have %d units together (max %d)",
                  kBigRow, small_units, kEmitThreads);
  long long grid = (long long)step->n * step->full_length;             // upper bound; the kernel reads the count
  if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL(serve_emit_kernel, dim3((int)grid), dim3(kEmitThreads), 0, (hipStream_t)stream, a);
  return seedhip::check_launch("serve_emit_kernel");
}

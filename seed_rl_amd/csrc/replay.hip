// Prioritized replay sampling on the device.
//
// Replaces PrioritizedReplay.sample / update_priorities of /root/reference/common/utils.py:309-370:
//   prob_i = priority_i^alpha / sum_j priority_j^alpha over the first `limit` slots,
//   indices ~ Categorical(prob) (with replacement; the reference uses tf.random.categorical),
//   weights = ((1/limit) / prob[idx])^beta, normalised by their maximum (:349-353).
// The buffer itself (1e5 unrolls x 0.85 MB = 85 GB for Atari R2D2) lives in HBM -- it fits MI355X's 288 GB --
// and is moved by the row mover (store.hip).  Sampling = inclusive scan of priority^alpha (two-level, fixed
// summation order => deterministic) + one binary search per sample on caller-supplied uniforms.
#include "common.h"
#include "../../include/seedhip.h"

namespace {

constexpr int kScanBlock = 1024;

// level 1: per-block inclusive scan of p^alpha; block totals to sums[]
__global__ void __launch_bounds__(kScanBlock)
replay_scan_blocks_kernel(const float* __restrict__ prio, long long limit, float alpha, float* __restrict__ cdf,
                          float* __restrict__ sums) {
  __shared__ float s[kScanBlock];
  const long long i = (long long)blockIdx.x * kScanBlock + threadIdx.x;
  float v = 0.f;
  if (i < limit) v = alpha == 1.0f ? prio[i] : powf(prio[i], alpha);
  s[threadIdx.x] = v;
  __syncthreads();
  for (int off = 1; off < kScanBlock; off <<= 1) {
    const float t = threadIdx.x >= off ? s[threadIdx.x - off] : 0.f;
    __syncthreads();
    s[threadIdx.x] += t;
    __syncthreads();
  }
  if (i < limit) cdf[i] = s[threadIdx.x];
  if (threadIdx.x == kScanBlock - 1) sums[blockIdx.x] = s[kScanBlock - 1];
}

// level 2: exclusive scan of the block totals (nblocks <= 1024) in place; sums[nblocks] = grand total
__global__ void __launch_bounds__(kScanBlock)
replay_scan_sums_kernel(float* __restrict__ sums, int nblocks) {
  __shared__ float s[kScanBlock];
  const float v = threadIdx.x < nblocks ? sums[threadIdx.x] : 0.f;
  s[threadIdx.x] = v;
  __syncthreads();
  for (int off = 1; off < kScanBlock; off <<= 1) {
    const float t = threadIdx.x >= off ? s[threadIdx.x - off] : 0.f;
    __syncthreads();
    s[threadIdx.x] += t;
    __syncthreads();
  }
  if (threadIdx.x < nblocks) sums[threadIdx.x] = s[threadIdx.x] - v;
  if (threadIdx.x == kScanBlock - 1) sums[nblocks] = s[kScanBlock - 1];
}

__device__ __forceinline__ float cdf_at(const float* cdf, const float* sums, long long i) {
  return cdf[i] + sums[i / kScanBlock];
}

// one thread per sample: smallest index with cdf[idx] > u * total; raw importance weight
__global__ void __launch_bounds__(256)
replay_sample_kernel(const float* __restrict__ prio, float alpha, const float* __restrict__ cdf,
                     const float* __restrict__ sums, int nblocks, long long limit, const float* __restrict__ uniforms, int num_samples, float beta, long long* __restrict__ indices,
                     float* __restrict__ weights) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= num_samples) return;
  const float total = sums[nblocks];
  const float target = uniforms[s] * total;
  long long lo = 0, hi = limit - 1;
  while (lo < hi) {
    const long long mid = (lo + hi) >> 1;
    if (cdf_at(cdf, sums, mid) > target) hi = mid; else lo = mid + 1;
  }
  indices[s] = lo;
  const float mass = alpha == 1.0f ? prio[lo] : powf(prio[lo], alpha);   // not a cdf difference: full precision
  const float prob = mass / total;
  weights[s] = powf((1.0f / (float)limit) / prob, beta);                 // utils.py:350-352
}

__global__ void __launch_bounds__(256)
replay_normalize_kernel(float* __restrict__ weights, int n) {            // weights /= max(weights)   (:353)
  __shared__ float red[4];
  float m = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) m = fmaxf(m, weights[i]);
  m = seedhip::wave_max(m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  for (int i = threadIdx.x; i < n; i += 256) weights[i] = weights[i] / m;
}

}  // namespace

extern "C" size_t seedhip_replay_sample_workspace_bytes(long long limit) {
  const long long nblocks = (limit + kScanBlock - 1) / kScanBlock;
  return (size_t)(limit + nblocks + 8) * sizeof(float);
}

extern "C" int seedhip_replay_sample(const float* priorities, long long limit, float priority_exponent,
                                     float importance_sampling_exponent, const float* uniforms, int num_samples,
                                     long long* indices, float* weights, void* workspace, size_t workspace_bytes,
                                     void* stream) {
  SEEDHIP_REQUIRE(limit >= 1 && num_samples >= 1, "replay_sample: Cannot sample if replay buffer is empty");
  SEEDHIP_REQUIRE(limit <= (long long)kScanBlock * kScanBlock, "replay_sample: at most %d slots", kScanBlock * kScanBlock);
  SEEDHIP_REQUIRE(priorities && uniforms && indices && weights && workspace, "replay_sample: null pointer");
  SEEDHIP_REQUIRE(workspace_bytes >= seedhip_replay_sample_workspace_bytes(limit), "replay_sample: workspace too small");
  SEEDHIP_REQUIRE(priority_exponent > 0.f, "replay_sample: priority_exponent must be > 0 (0 = uniform: sample on the host side)");
  hipStream_t s = (hipStream_t)stream;
  const int nblocks = (int)((limit + kScanBlock - 1) / kScanBlock);
  float* cdf = (float*)workspace;
  float* sums = cdf + limit;
  hipLaunchKernelGGL(replay_scan_blocks_kernel, dim3(nblocks), dim3(kScanBlock), 0, s, priorities, limit,
                     priority_exponent, cdf, sums);
  hipLaunchKernelGGL(replay_scan_sums_kernel, dim3(1), dim3(kScanBlock), 0, s, sums, nblocks);
  hipLaunchKernelGGL(replay_sample_kernel, dim3(seedhip::cdiv(num_samples, 256)), dim3(256), 0, s, priorities,
                     priority_exponent, cdf, sums, nblocks, limit, uniforms, num_samples, importance_sampling_exponent, indices, weights);
  hipLaunchKernelGGL(replay_normalize_kernel, dim3(1), dim3(256), 0, s, weights, num_samples);
  return seedhip::check_launch("replay_sample");
}

// ---- R2D2 actor-side exploration and the replay's time-major row indices ----------------------------------------- //
namespace {
// apply_epsilon_greedy (agents/r2d2/learner.py:147-177): with probability epsilons[env_id] the action is replaced by a
// uniform random one.  Two Philox randoms per row, keyed like the categorical sampler (seed, call counter, row).
__global__ void __launch_bounds__(256)
epsilon_greedy_kernel(long long* __restrict__ actions, const long long* __restrict__ env_ids,
                      const float* __restrict__ epsilons, int n, int num_envs, int num_actions,
                      const unsigned long long* __restrict__ rng, uint8_t* __restrict__ replaced) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned long long seed = rng[0], call = rng[1];
  const uint4 r = seedhip::philox4x32_10(make_uint4((uint32_t)call, (uint32_t)(call >> 32), (uint32_t)i, 0x45505347u),
                                         make_uint2((uint32_t)seed, (uint32_t)(seed >> 32)));
  const long long id = env_ids[i];
  const float eps = (id >= 0 && id < num_envs) ? epsilons[id] : 0.0f;
  const float prob = (float)(r.x >> 8) * (1.0f / 16777216.0f);                 // U[0,1): tf.random.uniform(shape)
  // tf.random.uniform(maxval=num_actions, dtype=int32): uniform over [0, num_actions)
  const int random_action = (int)(((unsigned long long)r.y * (unsigned long long)num_actions) >> 32);
  const bool take = prob < eps;                                                  // tf.where(probs < epsilons, random, actions)
  if (take) actions[i] = random_action;
  if (replaced) replaced[i] = take ? 1 : 0;
}
__global__ void rng_advance1_kernel(unsigned long long* rng) { rng[1] += 1; }

// Row indices that move unrolls between the replay's [slot][t] rows and a time-major [t][column] batch in ONE pass
// (utils.make_time_major of agents/r2d2/learner.py:453-457 folded into the gather): k = t * B + b ->
// replay_rows[k] = slots[b] * T1 + t, batch_rows[k] = k.
__global__ void __launch_bounds__(256)
replay_time_rows_kernel(const long long* __restrict__ slots, int B, int T1, long long* __restrict__ replay_rows,
                        long long* __restrict__ batch_rows) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= B * T1) return;
  const int t = k / B, b = k - t * B;
  replay_rows[k] = slots[b] * T1 + t;
  batch_rows[k] = k;
}
}  // namespace

extern "C" int seedhip_epsilon_greedy(long long* actions, const long long* env_ids, const float* epsilons, int n,
                                      int num_envs, int num_actions, unsigned long long* rng_state, uint8_t* replaced,
                                      void* stream) {
  SEEDHIP_REQUIRE(n >= 0 && num_envs >= 1 && num_actions >= 1, "epsilon_greedy: bad sizes");
  SEEDHIP_REQUIRE(rng_state, "epsilon_greedy: null rng_state");
  hipStream_t s = (hipStream_t)stream;
  if (n > 0) {
    SEEDHIP_REQUIRE(actions && env_ids && epsilons, "epsilon_greedy: null pointer");
    hipLaunchKernelGGL(epsilon_greedy_kernel, dim3(seedhip::cdiv(n, 256)), dim3(256), 0, s, actions, env_ids, epsilons,
                       n, num_envs, num_actions, rng_state, replaced);
  }
  hipLaunchKernelGGL(rng_advance1_kernel, dim3(1), dim3(1), 0, s, rng_state);
  return seedhip::check_launch("epsilon_greedy_kernel");
}

extern "C" int seedhip_replay_time_rows(const long long* slots, int num_unrolls, int steps, long long* replay_rows,
                                        long long* batch_rows, void* stream) {
  SEEDHIP_REQUIRE(num_unrolls >= 0 && steps >= 1, "replay_time_rows: bad sizes");
  if (num_unrolls == 0) return SEEDHIP_OK;
  SEEDHIP_REQUIRE(slots && replay_rows && batch_rows, "replay_time_rows: null pointer");
  SEEDHIP_REQUIRE((long long)num_unrolls * steps < (1LL << 31), "replay_time_rows: too many rows");
  hipLaunchKernelGGL(replay_time_rows_kernel, dim3(seedhip::cdiv((long long)num_unrolls * steps, 256)), dim3(256), 0,
                     (hipStream_t)stream, slots, num_unrolls, steps, replay_rows, batch_rows);
  return seedhip::check_launch("replay_time_rows_kernel");
}

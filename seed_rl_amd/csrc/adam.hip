// Flat fused Adam.
//
// Replaces the per-variable Keras OptimizerV2 Adam update issued by
// /root/reference/agents/vtrace/learner.py:272-275 (39 small kernels for
// ImpalaDeep) with ONE launch over the flat fp32 parameter buffer.
// Keras semantics:
//   m += (g - m)(1 - b1);  v += (g*g - v)(1 - b2);  p -= lr_t * m / (sqrt(v) + eps)
// with lr_t = lr * sqrt(1 - b2^t) / (1 - b1^t) computed by the host in fp64.
// `grad_scale` folds the data-parallel 1/world (mean) or 1 (reference SUM) in.
// `clamp_index` (>= 0) names ONE element that carries a Keras variable constraint (the learner's entropy-cost
// parameter, agents/vtrace/learner.py:225-232: clip to [-20/speed, 20/speed]), applied after its update; -1 = none.
// HBM-bound: 16 B read + 12 B written per parameter.
#include "common.h"
#include "../../include/seedhip.h"

namespace {
__global__ void __launch_bounds__(256)
adam_flat_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                 float* __restrict__ v, long long n, float lr_host, const float* __restrict__ lr_dev,
                 float one_minus_b1, float one_minus_b2, float eps, float grad_scale,
                 long long clamp_index, float clamp_lo, float clamp_hi, const int* __restrict__ guard) {
  // guard (may be null): a device word that is non-zero when this step's gradients are INVALID (a bounded wait of the
  // LSTM sequence kernels timed out: lstm_step.hip) -- the whole update is skipped, parameters and moments stay as
  // they are, and the host demotes the agent to the per-step kernels when it sees the flag (networks.py)
  if (guard && guard[0] != 0) return;
  const float lr_t = lr_dev ? lr_dev[0] : lr_host;   // device scalar: the step can sit in a captured HIP graph
  const long long stride = (long long)gridDim.x * blockDim.x;
  const long long n4 = n >> 2;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    float4 pp = reinterpret_cast<float4*>(p)[i];
    const float4 gg = reinterpret_cast<const float4*>(g)[i];
    float4 mm = reinterpret_cast<float4*>(m)[i];
    float4 vv = reinterpret_cast<float4*>(v)[i];
#define UPD(c)                                                   \
    { const float gs = gg.c * grad_scale;                        \
      mm.c += (gs - mm.c) * one_minus_b1;                        \
      vv.c += (gs * gs - vv.c) * one_minus_b2;                   \
      pp.c -= lr_t * mm.c / (sqrtf(vv.c) + eps); }
    UPD(x) UPD(y) UPD(z) UPD(w)
#undef UPD
    if (i == (clamp_index >> 2)) {                 // Keras variable constraint, applied after the update
      float q[4] = {pp.x, pp.y, pp.z, pp.w};
      const int c = (int)(clamp_index & 3);
      q[c] = fminf(fmaxf(q[c], clamp_lo), clamp_hi);
      pp = make_float4(q[0], q[1], q[2], q[3]);
    }
    reinterpret_cast<float4*>(p)[i] = pp;
    reinterpret_cast<float4*>(m)[i] = mm;
    reinterpret_cast<float4*>(v)[i] = vv;
  }
  // tail
  for (long long i = (n4 << 2) + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float gs = g[i] * grad_scale;
    float mm = m[i], vv = v[i];
    mm += (gs - mm) * one_minus_b1;
    vv += (gs * gs - vv) * one_minus_b2;
    float pn = p[i] - lr_t * mm / (sqrtf(vv) + eps);
    if (i == clamp_index) pn = fminf(fmaxf(pn, clamp_lo), clamp_hi);
    p[i] = pn;
    m[i] = mm; v[i] = vv;
  }
}

// sum of squares partials for clip_by_global_norm (r2d2/learner.py:606-609).
__global__ void __launch_bounds__(256)
sumsq_kernel(const float* __restrict__ g, long long n, float* __restrict__ partials) {
  __shared__ float s[4];
  float acc = 0.f;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) acc += g[i] * g[i];
  acc = seedhip::wave_sum(acc);
  if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partials[blockIdx.x] = (s[0] + s[1]) + (s[2] + s[3]);
}
__global__ void sumsq_final_kernel(const float* __restrict__ partials, int n, float* __restrict__ out) {
  float acc = 0.f;
  for (int i = threadIdx.x; i < n; i += 64) acc += partials[i];
  acc = seedhip::wave_sum(acc);
  if (threadIdx.x == 0) out[0] = acc;
}
// g *= clip / max(norm, clip) with norm = sqrt(*sumsq) read on device (no host sync).
__global__ void __launch_bounds__(256)
clip_scale_kernel(float* __restrict__ g, long long n, const float* __restrict__ sumsq, float clip) {
  const float norm = sqrtf(sumsq[0]);
  const float scale = clip / fmaxf(norm, clip);
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) g[i] *= scale;
}
constexpr int kSumsqBlocks = 512;
}  // namespace

extern "C" int seedhip_adam_flat(float* params, const float* grads, float* m, float* v, long long n,
                                 float lr_t, float beta_1, float beta_2, float epsilon, float grad_scale,
                                 long long clamp_index, float clamp_lo, float clamp_hi, void* stream) {
  return seedhip_adam_flat_guarded(params, grads, m, v, n, lr_t, nullptr, beta_1, beta_2, epsilon, grad_scale, clamp_index,
                                   clamp_lo, clamp_hi, nullptr, stream);
}
extern "C" int seedhip_adam_flat_dev_lr(float* params, const float* grads, float* m, float* v, long long n,
                                        const float* lr_t_device, float beta_1, float beta_2, float epsilon,
                                        float grad_scale, long long clamp_index, float clamp_lo, float clamp_hi,
                                        void* stream) {
  SEEDHIP_REQUIRE(n == 0 || lr_t_device, "adam: null pointer");
  return seedhip_adam_flat_guarded(params, grads, m, v, n, 0.f, lr_t_device, beta_1, beta_2, epsilon, grad_scale, clamp_index,
                                   clamp_lo, clamp_hi, nullptr, stream);
}

extern "C" int seedhip_adam_flat_guarded(float* params, const float* grads, float* m, float* v, long long n,
                                         float lr_t, const float* lr_t_device, float beta_1, float beta_2, float epsilon,
                                         float grad_scale, long long clamp_index, float clamp_lo, float clamp_hi,
                                         const int* skip_if_nonzero, void* stream) {
  SEEDHIP_REQUIRE(n >= 0, "adam: negative n");
  SEEDHIP_REQUIRE(clamp_index < n, "adam: clamp_index %lld out of range", clamp_index);
  if (n == 0) return SEEDHIP_OK;
  SEEDHIP_REQUIRE(params && grads && m && v, "adam: null pointer");
  SEEDHIP_REQUIRE(((((uintptr_t)params) | ((uintptr_t)grads) | ((uintptr_t)m) | ((uintptr_t)v)) & 15) == 0,
                  "adam: buffers must be 16-byte aligned");
  int blocks = seedhip::cdiv(n / 4 + 1, 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(adam_flat_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, params, grads, m, v, n,
                     lr_t, lr_t_device, 1.0f - beta_1, 1.0f - beta_2, epsilon, grad_scale, clamp_index,
                     clamp_lo, clamp_hi, skip_if_nonzero);
  return seedhip::check_launch("adam_flat_kernel");
}

extern "C" size_t seedhip_global_norm_workspace_bytes(void) { return (kSumsqBlocks + 4) * sizeof(float); }

extern "C" int seedhip_clip_by_global_norm(float* grads, long long n, float clip_norm, float* sumsq_out,
                                           void* workspace, size_t workspace_bytes, void* stream) {
  SEEDHIP_REQUIRE(n >= 0 && grads && sumsq_out && workspace, "clip_by_global_norm: bad args");
  SEEDHIP_REQUIRE(workspace_bytes >= seedhip_global_norm_workspace_bytes(), "clip_by_global_norm: workspace too small");
  if (n == 0) return SEEDHIP_OK;
  hipStream_t s = (hipStream_t)stream;
  float* partials = (float*)workspace;
  hipLaunchKernelGGL(sumsq_kernel, dim3(kSumsqBlocks), dim3(256), 0, s, grads, n, partials);
  hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(64), 0, s, partials, kSumsqBlocks, sumsq_out);
  if (clip_norm > 0.f) {
    int blocks = seedhip::cdiv(n, 256); if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(clip_scale_kernel, dim3(blocks), dim3(256), 0, s, grads, n, sumsq_out, clip_norm);
  }
  return seedhip::check_launch("clip_by_global_norm");
}

// R2D2 learner math: dueling Q head and the n-step double-Q loss.
//
//   dueling_fwd / dueling_bwd : /root/reference/atari/networks.py:273-285 (_head):
//       advantage -= mean(advantage); q = value + advantage; action = argmax(q)   (+ autodiff)
//   r2d2_loss_fwd_bwd : /root/reference/agents/r2d2/learner.py:258-330
//       (compute_loss_and_priorities_from_agent_outputs) with value rescaling h / h^-1 (:180-192),
//       the n-step Bellman target (:195-255; NOT Retrace -- SURVEY.md section 0, D2), the
//       per-sequence priorities, and the gradient seed of `reduce_mean(loss * importance_weights)`
//       (:604) wrt the training network's q-values.
// All arithmetic fp32 in the reference's op order (file compiled with -ffp-contract=off).
// Tiny tensors ([T,B] ~ 20k elements): two launches, latency-bound.
#include "common.h"
#include "../../include/seedhip.h"

namespace {

// ---- dueling head ------------------------------------------------------------------------ //
// va [rows, ld]: columns 0..A-1 = advantage, column A = value.
__global__ void __launch_bounds__(256)
dueling_fwd_kernel(const float* __restrict__ va, int ld, long long rows, int A, float* __restrict__ q,
                   int* __restrict__ action) {
  const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  const float* row = va + r * ld;
  float s = 0.f;
  for (int a = 0; a < A; ++a) s += row[a];
  const float mean = s / (float)A;
  const float v = row[A];
  float best = -INFINITY; int bi = 0;
  for (int a = 0; a < A; ++a) {
    const float adv = row[a] - mean;                   // networks.py:279
    const float qa = v + adv;                          // :282
    q[r * A + a] = qa;
    if (qa > best) { best = qa; bi = a; }              // tf.argmax: first maximum
  }
  if (action) action[r] = bi;
}

// d_va[:, a] = dq[a] - mean(dq); d_va[:, A] = sum(dq)
__global__ void __launch_bounds__(256)
dueling_bwd_kernel(const float* __restrict__ dq, long long rows, int A, float* __restrict__ d_va, int ld) {
  const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  float s = 0.f;
  for (int a = 0; a < A; ++a) s += dq[r * A + a];
  const float mean = s / (float)A;
  float* row = d_va + r * ld;
  for (int a = 0; a < A; ++a) row[a] = dq[r * A + a] - mean;
  row[A] = s;
  for (int a = A + 1; a < ld; ++a) row[a] = 0.f;
}

// ---- loss ------------------------------------------------------------------------------------ //
__device__ __forceinline__ float sgn(float x) { return x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f); }
__device__ __forceinline__ float h_fn(float x, float eps) {          // learner.py:180-183
  return sgn(x) * (sqrtf(fabsf(x) + 1.f) - 1.f) + eps * x;
}
__device__ __forceinline__ float h_inv(float x, float eps) {         // learner.py:186-192
  const float inner = sqrtf(1.f + 4.f * eps * (fabsf(x) + 1.f + eps));
  const float u = (inner - 1.f) / (2.f * eps);
  return sgn(x) * (u * u - 1.f);
}

// phase 1, one thread per (t,b): replay_q = Q(s_t, a_t); qtarget_max = h^-1(Q_target(s_t, argmax_a Q(s_t,a)))
__global__ void __launch_bounds__(256)
r2d2_rows_kernel(const float* __restrict__ tq, const float* __restrict__ gq, const int* __restrict__ actions,
                 long long rows, int A, float eps, float* __restrict__ replay_q, float* __restrict__ qmax) {
  const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  const float* t = tq + r * A;
  float best = -INFINITY; int bi = 0;
  for (int a = 0; a < A; ++a) if (t[a] > best) { best = t[a]; bi = a; }
  replay_q[r] = t[actions[r]];                                        // :287-293
  qmax[r] = h_inv(gq[r * A + bi], eps);                               // :295-304
}

struct GammaPow { float g[8]; };

// phase 2, one thread per column b: n-step target (:233-255), shift (:315-317), h (:319), TD errors,
// priorities (:324-326), loss (:329), gradient seed.
__global__ void __launch_bounds__(64)
r2d2_cols_kernel(const float* __restrict__ replay_q, const float* __restrict__ qmax, const float* __restrict__ rewards,
                 const uint8_t* __restrict__ done, const int* __restrict__ actions, const float* __restrict__ iw,
                 int T, int B, int A, float gamma, int n_steps, GammaPow gp, float eta, float eps, float inv_denom,
                 float* __restrict__ scratch /* [T+n, B] */, float* __restrict__ loss_b, float* __restrict__ prio_b,
                 float* __restrict__ d_tq /* [T,B,A] */, float* __restrict__ partial /* [gridDim.x] */) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  float wl = 0.f;
  if (b < B) {
    const int L = T + n_steps;                       // length of the padded target array
    // bellman_target = concat([0], q_target, [q_target[-1] / gamma^k for k = 1..n-1])
    scratch[b] = 0.f;
    for (int t = 0; t < T; ++t) scratch[(long long)(t + 1) * B + b] = qmax[(long long)t * B + b];
    const float last = qmax[(long long)(T - 1) * B + b];
    for (int k = 1; k < n_steps; ++k) scratch[(long long)(T + k) * B + b] = last / gp.g[k];
    // n passes of  bt[i] = r[i] + gamma * (1 - done[i]) * bt[i+1]  over arrays that shrink by one each pass;
    // rewards / done are zero-padded by n (index >= T reads 0 / false).
    int len = L;                                     // current length of bt
    for (int pass = 0; pass < n_steps; ++pass) {
      const int rl = T + n_steps - 1 - pass;         // length of rewards/done after dropping the last element
      for (int i = 0; i < rl; ++i) {
        const float r = i < T ? rewards[(long long)i * B + b] : 0.f;
        const float nd = (i < T && done[(long long)i * B + b]) ? 0.f : 1.f;
        scratch[(long long)i * B + b] = r + (gamma * nd) * scratch[(long long)(i + 1) * B + b];
      }
      len = rl;
    }
    (void)len;                                       // == T
    float mx = 0.f, sum_abs = 0.f, sum_sq = 0.f;
    const float w = iw ? iw[b] : 1.f;
    for (int t = 0; t < T - 1; ++t) {
      const float tgt = h_fn(scratch[(long long)(t + 1) * B + b], eps);        // bellman_target[1:], then h
      const float td = tgt - replay_q[(long long)t * B + b];                   // replay_q[:-1]
      const float ad = fabsf(td);
      mx = fmaxf(mx, ad); sum_abs += ad; sum_sq += ad * ad;
      // d(0.5 * td^2 * w / denom) / d q[t, b, a_t] = -td * w / denom
      float* row = d_tq + ((long long)t * B + b) * A;
      const int act = actions[(long long)t * B + b];
      for (int a = 0; a < A; ++a) row[a] = (a == act) ? -td * w * inv_denom : 0.f;
    }
    float* row = d_tq + ((long long)(T - 1) * B + b) * A;
    for (int a = 0; a < A; ++a) row[a] = 0.f;
    const float lb = 0.5f * sum_sq;
    loss_b[b] = lb;
    prio_b[b] = eta * mx + (1.f - eta) * (sum_abs / (float)(T - 1));
    wl = lb * w;
  }
  wl = seedhip::wave_sum(wl);
  if ((threadIdx.x & 63) == 0) partial[blockIdx.x] = wl;
}

__global__ void r2d2_total_kernel(const float* __restrict__ partial, int n, float inv_denom, float* __restrict__ total) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    float s = 0.f;
    for (int i = 0; i < n; ++i) s += partial[i];
    total[0] = s * inv_denom;
  }
}

}  // namespace

extern "C" int seedhip_dueling_fwd(const float* va, int ld, long long rows, int A, float* q, int* action,
                                   void* stream) {
  SEEDHIP_REQUIRE(va && q && rows >= 1 && A >= 1 && ld >= A + 1, "dueling_fwd: bad argument");
  hipLaunchKernelGGL(dueling_fwd_kernel, dim3(seedhip::cdiv(rows, 256)), dim3(256), 0, (hipStream_t)stream, va, ld,
                     rows, A, q, action);
  return seedhip::check_launch("dueling_fwd_kernel");
}

extern "C" int seedhip_dueling_bwd(const float* dq, long long rows, int A, float* d_va, int ld, void* stream) {
  SEEDHIP_REQUIRE(dq && d_va && rows >= 1 && A >= 1 && ld >= A + 1, "dueling_bwd: bad argument");
  hipLaunchKernelGGL(dueling_bwd_kernel, dim3(seedhip::cdiv(rows, 256)), dim3(256), 0, (hipStream_t)stream, dq, rows,
                     A, d_va, ld);
  return seedhip::check_launch("dueling_bwd_kernel");
}

extern "C" size_t seedhip_r2d2_loss_workspace_bytes(int T, int B, int n_steps) {
  // replay_q [T,B] + qmax [T,B] + scratch [T+n, B] + partial [ceil(B/64)]
  return ((size_t)(3 * T + n_steps) * B + (size_t)(B + 63) / 64 + 16) * sizeof(float);
}

extern "C" int seedhip_r2d2_loss_fwd_bwd(const float* training_q, const float* target_q, const int* actions,
                                         const float* rewards, const uint8_t* done,
                                         const float* importance_weights, int T, int B, int A, float gamma,
                                         int n_steps, float eta, float epsilon, float mean_denominator,
                                         float* loss_per_sequence, float* priorities, float* d_training_q,
                                         float* total_loss, void* workspace, size_t workspace_bytes, void* stream) {
  SEEDHIP_REQUIRE(T >= 2 && B >= 1 && A >= 1, "r2d2_loss: need T >= 2, B >= 1, A >= 1");
  SEEDHIP_REQUIRE(n_steps >= 1 && n_steps <= 8, "r2d2_loss: n_steps must be in [1, 8]");
  SEEDHIP_REQUIRE(training_q && target_q && actions && rewards && done && loss_per_sequence && priorities &&
                  d_training_q && total_loss && workspace, "r2d2_loss: null pointer");
  SEEDHIP_REQUIRE(workspace_bytes >= seedhip_r2d2_loss_workspace_bytes(T, B, n_steps), "r2d2_loss: workspace too small");
  SEEDHIP_REQUIRE(mean_denominator > 0.f && epsilon > 0.f, "r2d2_loss: mean_denominator and epsilon must be > 0");
  hipStream_t s = (hipStream_t)stream;
  float* replay_q = (float*)workspace;
  float* qmax = replay_q + (size_t)T * B;
  float* scratch = qmax + (size_t)T * B;
  float* partial = scratch + (size_t)(T + n_steps) * B;
  const long long rows = (long long)T * B;
  hipLaunchKernelGGL(r2d2_rows_kernel, dim3(seedhip::cdiv(rows, 256)), dim3(256), 0, s, training_q, target_q, actions,
                     rows, A, epsilon, replay_q, qmax);
  int rc = seedhip::check_launch("r2d2_rows_kernel"); if (rc) return rc;
  GammaPow gp;                          // the reference divides by gamma**k evaluated in double (learner.py:236-238)
  {
    const double gd = (double)gamma;
    double acc = 1.0;
    for (int k = 0; k < 8; ++k) { gp.g[k] = (float)acc; acc *= gd; }
  }
  const int nblk = (B + 63) / 64;
  hipLaunchKernelGGL(r2d2_cols_kernel, dim3(nblk), dim3(64), 0, s, replay_q, qmax, rewards, done, actions,
                     importance_weights, T, B, A, gamma, n_steps, gp, eta, epsilon, 1.0f / mean_denominator, scratch,
                     loss_per_sequence, priorities, d_training_q, partial);
  rc = seedhip::check_launch("r2d2_cols_kernel"); if (rc) return rc;
  hipLaunchKernelGGL(r2d2_total_kernel, dim3(1), dim3(64), 0, s, partial, nblk, 1.0f / mean_denominator, total_loss);
  return seedhip::check_launch("r2d2_total_kernel");
}

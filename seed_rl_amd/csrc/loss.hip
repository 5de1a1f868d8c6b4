// Fused IMPALA loss head, forward + backward (SURVEY.md 8(a) a2 + a3 + a1).
//
// Replaces /root/reference/agents/vtrace/learner.py:82-157 (the part of
// compute_loss after the agent unroll): categorical log-prob x2
// (common/parametric_distribution.py:69-74,94-95), V-trace
// (common/vtrace.py:84-148), policy / baseline / entropy / KL losses and the
// logged scalars -- ~30 elementwise/reduction TF ops + 12 logging reductions +
// their autodiff -- with ONE launch (+ a 1-block finalize for the scalars).
//
// Work decomposition (time-major [T+1,B,A] rows):
//   * a workgroup owns CB adjacent batch columns for ALL T+1 steps, so the
//     sequential time recursion never leaves the workgroup;
//   * phase 1 (row-parallel): 8 lanes per (t,b) row; the 8-lane groups of a wave
//     read adjacent rows => contiguous, coalesced logits reads; max / sum-exp /
//     entropy reductions are 3-step wavefront shuffles (xor 1,2,4);
//     per-row scalars (log-probs, entropy, lse, reward, discount, value) are
//     staged in LDS;
//   * phase 2 (column-serial): CB lanes run the T-step V-trace recursion out of
//     LDS (log-rho / c clipping + discounted-return recursion);
//   * phase 3 (row-parallel): gradients wrt policy_logits and baseline
//      are written coalesced; loss partial sums are
//     reduced wave->block deterministically (no atomics).
// HBM-bound: algorithmic bytes per (t,b) = 8A+13 (+4 for int64 actions) read,
// 4A+4 (+8 with vs/pg_adv emitted) written.
//
// Compiled with -ffp-contract=off so the V-trace arithmetic rounds exactly like
// the reference's unfused fp32 ops.
#include "common.h"
#include "../../include/seedhip.h"

namespace {

constexpr int kLPR = 8;            // lanes per row
constexpr int kThreads = 256;
constexpr int kGroups = kThreads / kLPR;
constexpr int kNumPartials = 8;    // per-block partial sums

__device__ __forceinline__ float grp_max(float v) {
  v = fmaxf(v, __shfl_xor(v, 1, 64)); v = fmaxf(v, __shfl_xor(v, 2, 64)); v = fmaxf(v, __shfl_xor(v, 4, 64));
  return v;
}
__device__ __forceinline__ float grp_sum(float v) {
  v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64);
  return v;
}

__device__ __forceinline__ long long load_action(const void* actions, int elem_size, long long idx) {
  return elem_size == 8 ? ((const long long*)actions)[idx] : (long long)((const int*)actions)[idx];
}

struct LossParams {
  const float* tgt_logits;   // [T+1,B,A] learner_outputs.policy_logits
  const float* baseline;     // [T+1,B]   learner_outputs.baseline
  const float* beh_logits;   // [T+1,B,A] agent_outputs.policy_logits
  const void* actions;       // [T+1,B]   agent_outputs.action (int32/int64)
  const float* rewards;      // [T+1,B]   env_outputs.reward
  const uint8_t* done;       // [T+1,B]   env_outputs.done
  int action_elem_size;
  int T, B, A;
  int logits_ld, baseline_ld;   // row strides (floats) of the learner logits / baseline (and their grads)
  float entropy_cost, baseline_cost, kl_cost, discounting, lambda_, max_abs_reward;
  float clip_rho, clip_pg_rho;
  float inv_n;               // 1 / mean_denominator
  const float* ec_param;     // learnable entropy cost: cost = exp(ec_mul * ec_param[0]) (learner.py:225-234); null = fixed
  float ec_mul;
  float* d_logits;           // [T+1,B,A]
  float* d_baseline;         // [T+1,B]
  float* vs;                 // [T,B] or null
  float* pg_adv;             // [T,B] or null
  float* partials;           // [nblocks, kNumPartials]
};

// EPL = elements per lane (A <= kLPR * EPL).
template <int EPL, int CB>
__global__ void __launch_bounds__(kThreads)
impala_loss_kernel(LossParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int T = p.T, B = p.B, A = p.A;
  const int T1 = T + 1;
  // LDS carve: 8 arrays of [T1][CB].
  float* s_tlp = smem;                 // target log-prob
  float* s_blp = s_tlp + T1 * CB;      // behaviour log-prob
  float* s_ent = s_blp + T1 * CB;      // entropy of target policy
  float* s_lse = s_ent + T1 * CB;      // logsumexp(target logits)
  float* s_rew = s_lse + T1 * CB;      // reward[t+1] (clipped)
  float* s_dis = s_rew + T1 * CB;      // discount[t] = (~done[t+1]) * gamma
  float* s_val = s_dis + T1 * CB;      // baseline[t]
  float* s_pg = s_val + T1 * CB;       // pg_adv[t]; after phase 2
  float* s_vs = s_pg + T1 * CB;        // vs[t]
  __shared__ float s_red[kThreads / 64][kNumPartials];

  const int tid = threadIdx.x;
  const int sub = tid & (kLPR - 1);
  const int grp = tid / kLPR;
  const int b0 = blockIdx.x * CB;
  const int nrows = T1 * CB;

  // ---------------- phase 1: per-row categorical statistics ---------------- //
  for (int r = grp; r < nrows; r += kGroups) {
    const int t = r / CB, c = r - t * CB;
    const int b = b0 + c;
    if (b >= B) continue;                           // uniform within the 8-lane group
    const long long tb = (long long)t * B + b;
    if (sub == 0) s_val[r] = p.baseline[tb * p.baseline_ld];
    if (t >= T) continue;                           // bootstrap row: value only
    const long long row = tb * A;                   // behaviour logits: contiguous
    const long long rowt = tb * p.logits_ld;        // learner logits: strided
    const long long act = load_action(p.actions, p.action_elem_size, tb);
    float x[EPL], y[EPL];
    float mx = -INFINITY, my = -INFINITY;
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
      const int a = sub + e * kLPR;
      x[e] = a < A ? p.tgt_logits[rowt + a] : -INFINITY;
      y[e] = a < A ? p.beh_logits[row + a] : -INFINITY;
      mx = fmaxf(mx, x[e]); my = fmaxf(my, y[e]);
    }
    mx = grp_max(mx); my = grp_max(my);
    float sx = 0.f, sy = 0.f;
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
      const int a = sub + e * kLPR;
      if (a < A) { sx += expf(x[e] - mx); sy += expf(y[e] - my); }
    }
    sx = grp_sum(sx); sy = grp_sum(sy);
    const float lsx = logf(sx), lsy = logf(sy);     // log_softmax = (x - max) - log(sum)
    float ent = 0.f, xa = 0.f, ya = 0.f;
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
      const int a = sub + e * kLPR;
      if (a < A) {
        const float ls = (x[e] - mx) - lsx;
        ent += expf(ls) * ls;
        if (a == act) { xa = ls; ya = (y[e] - my) - lsy; }
      }
    }
    ent = -grp_sum(ent); xa = grp_sum(xa); ya = grp_sum(ya);
    if (sub == 0) {
      s_tlp[r] = xa; s_blp[r] = ya; s_ent[r] = ent; s_lse[r] = mx + lsx;
      const long long tb1 = tb + B;                 // env_outputs[1:], learner.py:87
      float rw = p.rewards[tb1];
      if (p.max_abs_reward != 0.f) rw = fminf(fmaxf(rw, -p.max_abs_reward), p.max_abs_reward);
      s_rew[r] = rw;
      s_dis[r] = (p.done[tb1] ? 0.f : 1.f) * p.discounting;   // learner.py:93
    }
  }
  __syncthreads();

  // ---------------- phase 2: V-trace recursion per column ------------------ //
  if (tid < CB && b0 + tid < B) {
    const int c = tid;
    const bool has_rho = p.clip_rho >= 0.f, has_pg = p.clip_pg_rho >= 0.f;
    const float boot = s_val[T * CB + c];           // learner.py:82
    float acc = 0.f, vs_next = boot, v_next = boot;
    for (int t = T - 1; t >= 0; --t) {
      const int r = t * CB + c;
      const float rho = expf(s_tlp[r] - s_blp[r]);
      const float crho = has_rho ? fminf(p.clip_rho, rho) : rho;
      const float cs = fminf(1.0f, rho) * p.lambda_;
      const float d = s_dis[r], rw = s_rew[r], v = s_val[r];
      const float delta = crho * ((rw + d * v_next) - v);
      acc = delta + (d * cs) * acc;
      const float vs = acc + v;
      const float cpg = has_pg ? fminf(p.clip_pg_rho, rho) : rho;
      const float pg = cpg * ((rw + d * vs_next) - v);
      s_vs[r] = vs; s_pg[r] = pg;
      vs_next = vs; v_next = v;
    }
  }
  __syncthreads();

  // ---------------- phase 3: gradients + loss partial sums ----------------- //
  // entropy cost: a flag value, or exp(speed * param) of the learner's Lagrange-style parameter; either way a constant
  // for these gradients (stop_gradient, learner.py:121)
  const float ec = p.ec_param ? expf(p.ec_mul * p.ec_param[0]) : p.entropy_cost;
  float acc_pg = 0.f, acc_v2 = 0.f, acc_ent = 0.f, acc_kl = 0.f, acc_val = 0.f, acc_maxa = 0.f;
  for (int r = grp; r < nrows; r += kGroups) {
    const int t = r / CB, c = r - t * CB;
    const int b = b0 + c;
    if (b >= B) continue;
    const long long tb = (long long)t * B + b;
    const long long row = tb * p.logits_ld;
    if (t >= T) {                                   // bootstrap row: no gradient (Appendix C)
#pragma unroll
      for (int e = 0; e < EPL; ++e) { const int a = sub + e * kLPR; if (a < A) p.d_logits[row + a] = 0.f; }
      if (sub == 0) p.d_baseline[tb * p.baseline_ld] = 0.f;
      continue;
    }
    const long long act = load_action(p.actions, p.action_elem_size, tb);
    const float lse = s_lse[r], ent = s_ent[r], pg = s_pg[r], vs = s_vs[r], v = s_val[r];
    const float coef = (pg + p.kl_cost) * p.inv_n;  // policy-gradient + KL terms share (1[j=a]-p_j)
    const float ecn = ec * p.inv_n;
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
      const int a = sub + e * kLPR;
      if (a < A) {
        const float ls = p.tgt_logits[row + a] - lse;
        const float pr = expf(ls);
        const float onehot = (a == act) ? 1.f : 0.f;
        p.d_logits[row + a] = -coef * (onehot - pr) + ecn * pr * (ls + ent);
      }
    }
    if (sub == 0) {
      const float verr = vs - v;
      p.d_baseline[tb * p.baseline_ld] = p.baseline_cost * (v - vs) * p.inv_n;
      if (p.vs) p.vs[tb] = vs;
      if (p.pg_adv) p.pg_adv[tb] = pg;
      acc_pg += s_tlp[r] * pg; acc_v2 += verr * verr; acc_ent += ent;
      acc_kl += s_blp[r] - s_tlp[r]; acc_val += v;
      acc_maxa = fmaxf(acc_maxa, fabsf((float)act));
    }
  }
  // wave -> block reduction in a fixed order.
  acc_pg = seedhip::wave_sum(acc_pg); acc_v2 = seedhip::wave_sum(acc_v2);
  acc_ent = seedhip::wave_sum(acc_ent); acc_kl = seedhip::wave_sum(acc_kl);
  acc_val = seedhip::wave_sum(acc_val); acc_maxa = seedhip::wave_max(acc_maxa);
  const int wave = tid >> 6;
  if ((tid & 63) == 0) {
    s_red[wave][0] = acc_pg; s_red[wave][1] = acc_v2; s_red[wave][2] = acc_ent;
    s_red[wave][3] = acc_kl; s_red[wave][4] = acc_val; s_red[wave][5] = acc_maxa;
  }
  __syncthreads();
  if (tid == 0) {
    float o[kNumPartials] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int w = 0; w < kThreads / 64; ++w) {
      for (int k = 0; k < 5; ++k) o[k] += s_red[w][k];
      o[5] = fmaxf(o[5], s_red[w][5]);
    }
    for (int k = 0; k < kNumPartials; ++k) p.partials[(long long)blockIdx.x * kNumPartials + k] = o[k];
  }
}

// scalars[]: see SEEDHIP_LOSS_* indices in seedhip.h.
// Entropy-cost adjustment (learner.py:127-135, :225-234): with a learnable parameter theta the cost is
// c = exp(speed * theta); entropy_adjustment_loss = c * stop_gradient(mean(H) - target) when a target entropy is set
// (its only gradient: d/dtheta = speed * c * (mean(H) - target)), and 0 * c otherwise (gradient 0, never None).
__global__ void impala_loss_finalize_kernel(const float* __restrict__ partials, int nblocks, float inv_n,
                                            float entropy_cost, float baseline_cost, float kl_cost,
                                            const float* __restrict__ ec_param, float ec_mul, int has_target,
                                            float target_entropy_share, float* __restrict__ d_ec_param,
                                            float* __restrict__ scalars) {
  // One wave; each lane sums a strided subset in fixed order, then a shuffle tree.
  float o[6] = {0, 0, 0, 0, 0, 0};
  for (int i = threadIdx.x; i < nblocks; i += 64) {
    for (int k = 0; k < 5; ++k) o[k] += partials[(long long)i * kNumPartials + k];
    o[5] = fmaxf(o[5], partials[(long long)i * kNumPartials + 5]);
  }
  for (int k = 0; k < 5; ++k) o[k] = seedhip::wave_sum(o[k]);
  o[5] = seedhip::wave_max(o[5]);
  if (threadIdx.x == 0) {
    if (ec_param) entropy_cost = expf(ec_mul * ec_param[0]);
    const float policy_loss = -(o[0] * inv_n);                        // learner.py:111-112
    const float mse = o[1] * inv_n;
    const float v_loss = baseline_cost * 0.5f * mse;                  // :115-116
    const float entropy = o[2] * inv_n;                               // :119-120
    const float entropy_loss = entropy_cost * -entropy;               // :121
    const float kl_mean = o[3] * inv_n;
    const float kl_loss = kl_cost * kl_mean;                          // :124-125
    float adjustment = 0.f;                                           // :128-132
    if (has_target) adjustment = entropy_cost * (entropy - target_entropy_share);
    if (d_ec_param) d_ec_param[0] = has_target ? ec_mul * entropy_cost * (entropy - target_entropy_share) : 0.f;
    scalars[SEEDHIP_LOSS_TOTAL] = policy_loss + v_loss + entropy_loss + kl_loss + adjustment;  // :134-135
    scalars[SEEDHIP_LOSS_POLICY] = policy_loss;
    scalars[SEEDHIP_LOSS_V] = v_loss;
    scalars[SEEDHIP_LOSS_ENTROPY] = entropy_loss;
    scalars[SEEDHIP_LOSS_KL] = kl_loss;
    scalars[SEEDHIP_LOSS_ENTROPY_MEAN] = entropy;
    scalars[SEEDHIP_LOSS_KL_MEAN] = kl_mean;
    scalars[SEEDHIP_LOSS_VALUE_MEAN] = o[4] * inv_n;                  // :138-140
    scalars[SEEDHIP_LOSS_V_L2_ERROR] = sqrtf(mse);                    // :141
    scalars[SEEDHIP_LOSS_MAX_ACTION_ABS] = o[5];                      // :152-153
    scalars[SEEDHIP_LOSS_ENTROPY_COST] = entropy_cost;                // :155
    scalars[SEEDHIP_LOSS_ENTROPY_ADJUSTMENT] = adjustment;
  }
}

// Columns per workgroup.  8 keeps whole 32-byte sectors of the [T+1, B] scalars per workgroup; at the learner's own
// sizes (B = 512 per GPU: 64 workgroups of which each spends its V-trace phase on 8 lanes) the launch is latency-bound
// and under-fills the 256 CUs, so narrower column groups are used until the grid reaches the CU count: B = 512 -> 2
// columns x 256 workgroups (r02: 31 -> ~15 us per launch).
inline int pick_cb(int B) { return B >= 8 * 256 ? 8 : (B >= 4 * 256 ? 4 : 2); }

template <int EPL, int CB>
void launch_loss_cb(const LossParams& p, int nblocks, size_t lds, hipStream_t s) {
  if (lds > 48 * 1024)
    (void)hipFuncSetAttribute((const void*)impala_loss_kernel<EPL, CB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((impala_loss_kernel<EPL, CB>), dim3(nblocks), dim3(kThreads), lds, s, p);
}
template <int EPL>
void launch_loss(const LossParams& p, int cb, int nblocks, size_t lds, hipStream_t s) {
  if (cb == 8) launch_loss_cb<EPL, 8>(p, nblocks, lds, s);
  else if (cb == 4) launch_loss_cb<EPL, 4>(p, nblocks, lds, s);
  else launch_loss_cb<EPL, 2>(p, nblocks, lds, s);
}

}  // namespace

extern "C" size_t seedhip_impala_loss_workspace_bytes(int T, int B) {
  (void)T;
  const int nblocks = (B + 1) / 2;                             // the narrowest column group (pick_cb)
  return (size_t)nblocks * kNumPartials * sizeof(float);
}

namespace {
int impala_loss_impl(
    const float* learner_policy_logits, int logits_ld, const float* learner_baseline, int baseline_ld,
    const float* behaviour_policy_logits, const void* actions, int action_elem_size,
    const float* rewards, const uint8_t* done, int T, int B, int A,
    float entropy_cost, float baseline_cost, float kl_cost, float discounting, float lambda_,
    float max_abs_reward, float clip_rho_threshold, float clip_pg_rho_threshold,
    float mean_denominator, float* d_policy_logits, float* d_baseline, float* vs, float* pg_advantages,
    float* scalars, void* workspace, size_t workspace_bytes, void* stream,
    const float* ec_param, float ec_mul, int has_target, float target_share, float* d_ec_param) {
  SEEDHIP_REQUIRE(T >= 1 && B >= 1 && A >= 1, "impala_loss: need T>=1,B>=1,A>=1 (got %d,%d,%d)", T, B, A);
  SEEDHIP_REQUIRE(A <= kLPR * 16, "impala_loss: A=%d > %d unsupported", A, kLPR * 16);
  SEEDHIP_REQUIRE(logits_ld >= A && baseline_ld >= 1, "impala_loss: bad row strides");
  SEEDHIP_REQUIRE(action_elem_size == 4 || action_elem_size == 8, "impala_loss: action_elem_size must be 4 or 8");
  SEEDHIP_REQUIRE(learner_policy_logits && learner_baseline && behaviour_policy_logits && actions && rewards &&
                  done && d_policy_logits && d_baseline && scalars && workspace, "impala_loss: null pointer");
  SEEDHIP_REQUIRE(workspace_bytes >= seedhip_impala_loss_workspace_bytes(T, B), "impala_loss: workspace too small");
  SEEDHIP_REQUIRE(mean_denominator > 0.f, "impala_loss: mean_denominator must be > 0");
  const int cb = pick_cb(B);
  const size_t lds = (size_t)(T + 1) * cb * 9 * sizeof(float);
  SEEDHIP_REQUIRE(lds <= 150 * 1024, "impala_loss: T=%d too long for LDS staging", T);
  LossParams p;
  p.tgt_logits = learner_policy_logits; p.baseline = learner_baseline; p.beh_logits = behaviour_policy_logits;
  p.actions = actions; p.rewards = rewards; p.done = done; p.action_elem_size = action_elem_size;
  p.T = T; p.B = B; p.A = A; p.logits_ld = logits_ld; p.baseline_ld = baseline_ld;
  p.entropy_cost = entropy_cost; p.baseline_cost = baseline_cost; p.kl_cost = kl_cost;
  p.discounting = discounting; p.lambda_ = lambda_; p.max_abs_reward = max_abs_reward;
  p.clip_rho = clip_rho_threshold; p.clip_pg_rho = clip_pg_rho_threshold;
  p.inv_n = 1.0f / mean_denominator;
  p.d_logits = d_policy_logits; p.d_baseline = d_baseline; p.vs = vs; p.pg_adv = pg_advantages;
  p.partials = (float*)workspace;
  p.ec_param = ec_param; p.ec_mul = ec_mul;
  hipStream_t s = (hipStream_t)stream;
  const int nblocks = (B + cb - 1) / cb;
  if (A <= kLPR) launch_loss<1>(p, cb, nblocks, lds, s);
  else if (A <= 2 * kLPR) launch_loss<2>(p, cb, nblocks, lds, s);
  else if (A <= 4 * kLPR) launch_loss<4>(p, cb, nblocks, lds, s);
  else if (A <= 8 * kLPR) launch_loss<8>(p, cb, nblocks, lds, s);
  else launch_loss<16>(p, cb, nblocks, lds, s);
  int rc = seedhip::check_launch("impala_loss_kernel");
  if (rc) return rc;
  hipLaunchKernelGGL(impala_loss_finalize_kernel, dim3(1), dim3(64), 0, s, (const float*)workspace, nblocks,
                     p.inv_n, entropy_cost, baseline_cost, kl_cost, ec_param, ec_mul, has_target, target_share,
                     d_ec_param, scalars);
  return seedhip::check_launch("impala_loss_finalize_kernel");
}
}  // namespace

extern "C" int seedhip_impala_loss_fwd_bwd(
    const float* learner_policy_logits, int logits_ld, const float* learner_baseline, int baseline_ld,
    const float* behaviour_policy_logits, const void* actions, int action_elem_size,
    const float* rewards, const uint8_t* done, int T, int B, int A,
    float entropy_cost, float baseline_cost, float kl_cost, float discounting, float lambda_,
    float max_abs_reward, float clip_rho_threshold, float clip_pg_rho_threshold,
    float mean_denominator, float* d_policy_logits, float* d_baseline, float* vs, float* pg_advantages,
    float* scalars, void* workspace, size_t workspace_bytes, void* stream) {
  return impala_loss_impl(learner_policy_logits, logits_ld, learner_baseline, baseline_ld, behaviour_policy_logits,
                          actions, action_elem_size, rewards, done, T, B, A, entropy_cost, baseline_cost, kl_cost,
                          discounting, lambda_, max_abs_reward, clip_rho_threshold, clip_pg_rho_threshold,
                          mean_denominator, d_policy_logits, d_baseline, vs, pg_advantages, scalars, workspace,
                          workspace_bytes, stream, nullptr, 0.f, 0, 0.f, nullptr);
}

extern "C" int seedhip_impala_loss_fwd_bwd_adaptive(
    const float* learner_policy_logits, int logits_ld, const float* learner_baseline, int baseline_ld,
    const float* behaviour_policy_logits, const void* actions, int action_elem_size,
    const float* rewards, const uint8_t* done, int T, int B, int A,
    const float* entropy_cost_param, float entropy_cost_adjustment_speed, int has_target_entropy,
    float target_entropy, float* d_entropy_cost_param,
    float baseline_cost, float kl_cost, float discounting, float lambda_,
    float max_abs_reward, float clip_rho_threshold, float clip_pg_rho_threshold,
    float mean_denominator, float* d_policy_logits, float* d_baseline, float* vs, float* pg_advantages,
    float* scalars, void* workspace, size_t workspace_bytes, void* stream) {
  SEEDHIP_REQUIRE(entropy_cost_param && d_entropy_cost_param, "impala_loss_adaptive: null entropy-cost parameter");
  return impala_loss_impl(learner_policy_logits, logits_ld, learner_baseline, baseline_ld, behaviour_policy_logits,
                          actions, action_elem_size, rewards, done, T, B, A, 0.f, baseline_cost, kl_cost,
                          discounting, lambda_, max_abs_reward, clip_rho_threshold, clip_pg_rho_threshold,
                          mean_denominator, d_policy_logits, d_baseline, vs, pg_advantages, scalars, workspace,
                          workspace_bytes, stream, entropy_cost_param, entropy_cost_adjustment_speed,
                          has_target_entropy, target_entropy, d_entropy_cost_param);
}

// ---- categorical log_prob / entropy (common/parametric_distribution.py:69-74) ---- //
namespace {
template <int EPL>
__global__ void __launch_bounds__(kThreads)
categorical_kernel(const float* __restrict__ logits, const void* __restrict__ actions, int action_elem_size,
                   long long rows, int A, float* __restrict__ log_prob, float* __restrict__ entropy) {
  const int sub = threadIdx.x & (kLPR - 1);
  const long long r = ((long long)blockIdx.x * kThreads + threadIdx.x) / kLPR;
  if (r >= rows) return;
  const long long row = r * A;
  const long long act = actions ? load_action(actions, action_elem_size, r) : -1;
  float x[EPL];
  float mx = -INFINITY;
#pragma unroll
  for (int e = 0; e < EPL; ++e) {
    const int a = sub + e * kLPR;
    x[e] = a < A ? logits[row + a] : -INFINITY;
    mx = fmaxf(mx, x[e]);
  }
  mx = grp_max(mx);
  float sx = 0.f;
#pragma unroll
  for (int e = 0; e < EPL; ++e) { const int a = sub + e * kLPR; if (a < A) sx += expf(x[e] - mx); }
  sx = grp_sum(sx);
  const float lsx = logf(sx);
  float ent = 0.f, xa = 0.f;
#pragma unroll
  for (int e = 0; e < EPL; ++e) {
    const int a = sub + e * kLPR;
    if (a < A) { const float ls = (x[e] - mx) - lsx; ent += expf(ls) * ls; if (a == act) xa = ls; }
  }
  ent = -grp_sum(ent); xa = grp_sum(xa);
  if (sub == 0) { if (log_prob) log_prob[r] = xa; if (entropy) entropy[r] = ent; }
}
}  // namespace

extern "C" int seedhip_categorical_log_prob_entropy(const float* logits, const void* actions, int action_elem_size,
                                                    long long rows, int A, float* log_prob, float* entropy,
                                                    void* stream) {
  SEEDHIP_REQUIRE(rows >= 0 && A >= 1 && A <= kLPR * 16, "categorical: bad rows=%lld A=%d", rows, A);
  if (rows == 0) return SEEDHIP_OK;
  SEEDHIP_REQUIRE(logits && (log_prob || entropy), "categorical: null pointer");
  SEEDHIP_REQUIRE(!log_prob || actions, "categorical: log_prob needs actions");
  SEEDHIP_REQUIRE(!actions || action_elem_size == 4 || action_elem_size == 8, "categorical: action_elem_size");
  const int nblocks = seedhip::cdiv(rows * kLPR, kThreads);
  hipStream_t s = (hipStream_t)stream;
#define L(E) hipLaunchKernelGGL((categorical_kernel<E>), dim3(nblocks), dim3(kThreads), 0, s, logits, actions, \
                                action_elem_size, rows, A, log_prob, entropy)
  if (A <= kLPR) L(1); else if (A <= 2 * kLPR) L(2); else if (A <= 4 * kLPR) L(4); else if (A <= 8 * kLPR) L(8); else L(16);
#undef L
  return seedhip::check_launch("categorical_kernel");
}

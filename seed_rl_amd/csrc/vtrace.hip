// V-trace reverse-time scan.
//
// Replaces /root/reference/common/vtrace.py:84-148 (from_importance_weights):
// ~100 tiny TF ops (4 [B]-vector ops x T + concat/stack/reverse) become ONE
// launch.  One lane owns one (or 4 adjacent) batch column(s); time-major [T,B]
// rows make every load/store a fully coalesced 256 B (1 KiB for the float4
// form) wave access.  Loads are independent of the recurrence, so they are
// issued U timesteps ahead of the serial `acc` chain.
//
// HBM-bound: 28 B per (t,b) element (5 fp32 reads + 2 fp32 writes) + 4 B per
// column (bootstrap).  Arithmetic order and rounding mirror the reference
// exactly (file compiled with -ffp-contract=off; IEEE expf, no fast-math):
//   acc = delta + (discount * c) * acc            (vtrace.py:128)
#include "common.h"
#include "../../include/seedhip.h"

namespace {

template <int V> struct Vec;
template <> struct Vec<1> { using type = float; };
template <> struct Vec<4> { using type = float4; };

template <int V> __device__ __forceinline__ void ld(const float* p, float (&o)[V]) {
  if constexpr (V == 4) { float4 v = *reinterpret_cast<const float4*>(p); o[0]=v.x; o[1]=v.y; o[2]=v.z; o[3]=v.w; }
  else o[0] = *p;
}
template <int V, bool NT = false> __device__ __forceinline__ void st(float* p, const float (&o)[V]) {
  if constexpr (V == 4) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    const f4 v = {o[0], o[1], o[2], o[3]};
    if constexpr (NT) __builtin_nontemporal_store(v, reinterpret_cast<f4*>(p));      // written once, read by a later kernel
    else *reinterpret_cast<f4*>(p) = v;
  } else *p = o[0];
}
template <int V, bool NT> __device__ __forceinline__ void ldx(const float* p, float (&o)[V]) {
  if constexpr (V == 4 && NT) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    const f4 v = __builtin_nontemporal_load(reinterpret_cast<const f4*>(p));
    o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3];
  } else ld<V>(p, o);
}

// U = timesteps whose loads are in flight together.
template <int V, int U, int NT = 0>                     // NT bit 0: non-temporal stores, bit 1: non-temporal loads
__global__ void __launch_bounds__(256)
vtrace_scan_kernel(const float* __restrict__ tgt, const float* __restrict__ beh,
                   const float* __restrict__ disc, const float* __restrict__ rew,
                   const float* __restrict__ val, const float* __restrict__ boot,
                   float clip_rho, float clip_pg, float lambda, int T, long long B, long long row_ld,
                   float* __restrict__ vs_out, float* __restrict__ pg_out) {
  const long long col = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * V;
  if (col >= B) return;
  float acc[V], vs_next[V], v_next[V];
  {
    float b[V]; ld<V>(boot + col, b);
#pragma unroll
    for (int i = 0; i < V; ++i) { acc[i] = 0.f; vs_next[i] = b[i]; v_next[i] = b[i]; }
  }
  const bool has_rho = clip_rho >= 0.f, has_pg = clip_pg >= 0.f;
  int t = T - 1;
  while (t >= 0) {
    const int n = (t + 1 < U) ? t + 1 : U;      // steps in this chunk
    float a_t[U][V], a_b[U][V], a_d[U][V], a_r[U][V], a_v[U][V];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (u < n) {
        const long long off = (long long)(t - u) * row_ld + col;
        ldx<V, (NT & 2) != 0>(tgt + off, a_t[u]); ldx<V, (NT & 2) != 0>(beh + off, a_b[u]); ldx<V, (NT & 2) != 0>(disc + off, a_d[u]);
        ldx<V, (NT & 2) != 0>(rew + off, a_r[u]); ldx<V, (NT & 2) != 0>(val + off, a_v[u]);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (u < n) {
        float o_vs[V], o_pg[V];
#pragma unroll
        for (int i = 0; i < V; ++i) {
          const float log_rho = a_t[u][i] - a_b[u][i];                 // :84
          const float rho = expf(log_rho);                              // :110
          const float crho = has_rho ? fminf(clip_rho, rho) : rho;     // :111-114
          const float c = fminf(1.0f, rho) * lambda;                   // :116-117
          const float d = a_d[u][i], r = a_r[u][i], v = a_v[u][i];
          const float delta = crho * ((r + d * v_next[i]) - v);        // :122
          acc[i] = delta + (d * c) * acc[i];                           // :128
          const float vs = acc[i] + v;                                 // :133
          const float cpg = has_pg ? fminf(clip_pg, rho) : rho;        // :138-142
          o_pg[i] = cpg * ((r + d * vs_next[i]) - v);                  // :143-144
          o_vs[i] = vs; vs_next[i] = vs; v_next[i] = v;
        }
        const long long off = (long long)(t - u) * row_ld + col;
        st<V, (NT & 1) != 0>(vs_out + off, o_vs); st<V, (NT & 1) != 0>(pg_out + off, o_pg);
      }
    }
    t -= n;
  }
}

}  // namespace

extern "C" int seedhip_vtrace_from_importance_weights(
    const float* target_action_log_probs, const float* behaviour_action_log_probs,
    const float* discounts, const float* rewards, const float* values,
    const float* bootstrap_value, float clip_rho_threshold, float clip_pg_rho_threshold,
    float lambda_, int T, long long B, float* vs, float* pg_advantages, void* stream) {
  SEEDHIP_REQUIRE(T >= 0 && B >= 0, "vtrace: negative T=%d or B=%lld", T, B);
  if (T == 0 || B == 0) return SEEDHIP_OK;
  SEEDHIP_REQUIRE(target_action_log_probs && behaviour_action_log_probs && discounts && rewards &&
                  values && bootstrap_value && vs && pg_advantages, "vtrace: null pointer");
  hipStream_t s = (hipStream_t)stream;
  auto aligned16 = [](const void* p) { return (((uintptr_t)p) & 15) == 0; };
  const bool vec4 = (B % 4 == 0) && B >= (1 << 16) && aligned16(target_action_log_probs) &&
                    aligned16(behaviour_action_log_probs) && aligned16(discounts) &&
                    aligned16(rewards) && aligned16(values) && aligned16(bootstrap_value) &&
                    aligned16(vs) && aligned16(pg_advantages);
  if (vec4) {
    const int block = 256;
    // Cache policy by working set (measured on MI355X, tools/bench_vtrace.py, GB/s of the 28 B per element):
    //   <= 0.3 GB (B <= 2^19; the 256 MB infinity cache still helps): plain loads / stores, one step in flight
    //                                                     4.8 / 6.3 / 6.7 TB/s at 2^17 / 2^18 / 2^19 (non-temporal: 3.4 / 5.2 / 6.0)
    //   0.6-1.2 GB (B = 2^20, 2^21): non-temporal loads AND stores, one step in flight     5.8-6.1 / 5.3 TB/s (plain: 5.1-5.2)
    //   2.3 GB (B = 2^22): non-temporal, two steps in flight                               4.8-5.0 TB/s (plain: 4.6)
    static const int forced = getenv("SEEDHIP_VTRACE_VARIANT") ? atoi(getenv("SEEDHIP_VTRACE_VARIANT")) : -1;
    const long long bytes = 28LL * T * B;
    const int variant = forced >= 0 ? forced : (bytes <= (400LL << 20) ? 6 : (bytes <= (1500LL << 20) ? 7 : 4));
    // Column chunks (r6; VERDICT r5 item 9: 0.73 of HBM at B = 2^20 fell to 0.58 at 2^22).  Every workgroup walks its
    // columns at its own t, so one launch touches all T rows of all seven arrays over the column range in flight: at
    // B = 2^22 that is 140 row streams 16 MB apart over 2.35 GB.  Launching the same problem as column chunks (row stride
    // B, 2^19 columns = 0.29 GB of footprint each) bounds what is open at a time: 496 -> 447 us (4.77 -> 5.29 TB/s, 0.66
    // of HBM) at 2^22; chunks of 2^20 / 2^21 give 470 / 485 us, B <= 2^21 does not move (tools/bench_vtrace.py with
    // SEEDHIP_VTRACE_CHUNK).  The grid was never capped; the loss is the footprint (page / DRAM-row locality), not the launch.
    static const long long forced_chunk = getenv("SEEDHIP_VTRACE_CHUNK") ? atoll(getenv("SEEDHIP_VTRACE_CHUNK")) : -1;
    long long chunk = forced_chunk >= 0 ? forced_chunk : (bytes > (1500LL << 20) ? (1LL << 19) : 0);
    if (chunk <= 0 || chunk > B) chunk = B;
    chunk = chunk / 4 * 4;
#define SEEDHIP_VT(U_, NT_)                                                                                        \
    for (long long c0 = 0; c0 < B; c0 += chunk) {                                                                 \
      const long long bc = B - c0 < chunk ? B - c0 : chunk;                                                       \
      hipLaunchKernelGGL((vtrace_scan_kernel<4, U_, NT_>), dim3(seedhip::cdiv(bc / 4, block)), dim3(block), 0, s, \
                         target_action_log_probs + c0, behaviour_action_log_probs + c0, discounts + c0,           \
                         rewards + c0, values + c0, bootstrap_value + c0, clip_rho_threshold,                     \
                         clip_pg_rho_threshold, lambda_, T, bc, B, vs + c0, pg_advantages + c0);                  \
    }
    switch (variant) {
      case 1: SEEDHIP_VT(4, 0); break;
      case 2: SEEDHIP_VT(2, 1); break;
      case 3: SEEDHIP_VT(4, 1); break;
      case 4: SEEDHIP_VT(2, 3); break;
      case 5: SEEDHIP_VT(4, 3); break;
      case 6: SEEDHIP_VT(1, 0); break;
      case 7: SEEDHIP_VT(1, 3); break;
      default: SEEDHIP_VT(2, 0); break;
    }
#undef SEEDHIP_VT
  } else {
    // Small B: spread the columns over as many CUs as possible (64-lane blocks).
    const int block = (B >= (1 << 15)) ? 256 : 64;
    hipLaunchKernelGGL((vtrace_scan_kernel<1, 4>), dim3(seedhip::cdiv(B, block)), dim3(block), 0, s,
                       target_action_log_probs, behaviour_action_log_probs, discounts, rewards, values,
                       bootstrap_value, clip_rho_threshold, clip_pg_rho_threshold, lambda_, T, B, B, vs,
                       pg_advantages);
  }
  return seedhip::check_launch("vtrace_scan_kernel");
}

// Common helpers for the seedhip C-ABI library (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#define SEEDHIP_OK 0
#define SEEDHIP_ERR_INVALID (-1)     // bad argument (shape / null pointer / unsupported size)
#define SEEDHIP_ERR_LAUNCH (-2)      // hipGetLastError() != hipSuccess after launch
#define SEEDHIP_ERR_UNSUPPORTED (-3)

namespace seedhip {

// Thread-local last-error string (never throws, never aborts).
char* err_buf();
int fail(int code, const char* fmt, ...);

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(SEEDHIP_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
  return SEEDHIP_OK;
}

constexpr int kWave = 64;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// ---- counter-based random numbers: Philox4x32-10 (Salmon et al., SC'11) ------------------------------------- //
// A pure function of (counter, key): no generator state to carry through a HIP graph except one device counter that
// the sampling kernels advance themselves.
__device__ __forceinline__ uint4 philox4x32_10(uint4 c, uint2 k) {
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c.x), lo0 = 0xD2511F53u * c.x;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c.z), lo1 = 0xCD9E8D57u * c.z;
    c = make_uint4(hi1 ^ c.y ^ k.x, lo1, hi0 ^ c.w ^ k.y, lo0);
    k.x += 0x9E3779B9u; k.y += 0xBB67AE85u;
  }
  return c;
}

// One categorical sample from unnormalised logits by the Gumbel-max trick: argmax_a (logit_a - log(-log u_a)),
// u_a ~ U(0,1) -- distributed exactly like tfd.Categorical(logits).sample() of the reference head
// (dmlab/networks.py:122, common/parametric_distribution.py:94-95); first maximum wins.  Deterministic in
// (seed, call counter, row): the fused inference kernel and the stand-alone sampler give the same action.
__device__ __forceinline__ int sample_categorical_row(const float* __restrict__ logits, int A, unsigned long long seed,
                                                      unsigned long long call, unsigned int row) {
  const uint2 key = make_uint2((uint32_t)seed, (uint32_t)(seed >> 32));
  float best = -INFINITY;
  int arg = 0;
  for (int a0 = 0; a0 < A; a0 += 4) {
    const uint4 r = philox4x32_10(make_uint4((uint32_t)call, (uint32_t)(call >> 32), row, (uint32_t)(a0 >> 2)), key);
    const uint32_t x[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int a = a0 + j;
      if (a < A) {
        const float u = ((float)(x[j] >> 9) + 0.5f) * (1.0f / 8388608.0f);      // in (0, 1) exactly: 23 random bits + 1/2
        const float v = logits[a] - logf(-logf(u));
        if (v > best) { best = v; arg = a; }
      }
    }
  }
  return arg;
}

}  // namespace seedhip

#define SEEDHIP_REQUIRE(cond, ...) \
  do { if (!(cond)) return seedhip::fail(SEEDHIP_ERR_INVALID, __VA_ARGS__); } while (0)

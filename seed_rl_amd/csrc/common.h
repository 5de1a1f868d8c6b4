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

}  // namespace seedhip

#define SEEDHIP_REQUIRE(cond, ...) \
  do { if (!(cond)) return seedhip::fail(SEEDHIP_ERR_INVALID, __VA_ARGS__); } while (0)

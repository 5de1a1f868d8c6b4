// Host-side launch helpers shared by conv.hip and stackconv.hip.
#pragma once
#include "common.h"

namespace seedhip {

// out[i] = sum_z partial[z][i]  (fixed order => deterministic), float4 vectorised.
static __global__ void __launch_bounds__(256)
reduce_slices_kernel(const float* __restrict__ partial, int slices, long long n, float* __restrict__ out) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  const long long n4 = n >> 2;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    float4 a = reinterpret_cast<const float4*>(partial)[i];
    for (int z = 1; z < slices; ++z) {
      const float4 b = reinterpret_cast<const float4*>(partial + (long long)z * n)[i];
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    reinterpret_cast<float4*>(out)[i] = a;
  }
  for (long long i = (n4 << 2) + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    float a = partial[i];
    for (int z = 1; z < slices; ++z) a += partial[(long long)z * n + i];
    out[i] = a;
  }
}

// Many slices (persistent split-K workgroups write one slice each): 16 float4 columns x 16 slice lanes per
// block; lane q sums slices q, q+16, ... (independent loads in flight), then the 16 lanes are combined through
// LDS in a fixed order.  Deterministic, and 16x more parallel than one thread walking all slices of a column.
static __global__ void __launch_bounds__(256)
reduce_slices_wide_kernel(const float* __restrict__ partial, int slices, long long n, float* __restrict__ out) {
  __shared__ float4 red[16][17];
  const int col = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const long long n4 = n >> 2;
  const long long c4 = (long long)blockIdx.x * 16 + col;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c4 < n4) {
    for (int z = sl; z < slices; z += 16) {
      const float4 b = reinterpret_cast<const float4*>(partial + (long long)z * n)[c4];
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
  }
  red[sl][col] = a;
  __syncthreads();
  if (sl == 0 && c4 < n4) {
    float4 t = red[0][col];
    for (int q = 1; q < 16; ++q) { const float4 b = red[q][col]; t.x += b.x; t.y += b.y; t.z += b.z; t.w += b.w; }
    reinterpret_cast<float4*>(out)[c4] = t;
  }
}

static inline void reduce_slices(const float* partial, int slices, long long n, float* out, hipStream_t s) {
  if (slices >= 32 && (n & 3) == 0 && ((((uintptr_t)partial) | ((uintptr_t)out)) & 15) == 0) {
    hipLaunchKernelGGL(reduce_slices_wide_kernel, dim3(cdiv(n >> 2, 16)), dim3(256), 0, s, partial, slices, n, out);
    return;
  }
  int blocks = cdiv(n / 4 + 1, 256); if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(reduce_slices_kernel, dim3(blocks), dim3(256), 0, s, partial, slices, n, out);
}

// pixels per split-K slice for weight gradients: enough slices to fill the chip,
// few enough that the partial-sum traffic stays small.
static inline int pick_k_per_slice(long long pixels, long long tiles_mn) {
  long long want_slices = (1024 + tiles_mn - 1) / tiles_mn;     // ~4 workgroups per CU
  if (want_slices < 1) want_slices = 1;
  if (want_slices > 256) want_slices = 256;
  long long per = (pixels + want_slices - 1) / want_slices;
  per = ((per + 63) / 64) * 64;
  if (per < 256) per = 256;
  return (int)per;
}
static inline long long tiles_for(int M, int N) {
  int bm, bn;
  if (N <= 16) { bm = 256; bn = 16; } else if (N <= 32) { bm = 128; bn = 32; } else if (N <= 48) { bm = 128; bn = 48; } else { bm = 128; bn = 64; }
  return (long long)((M + bm - 1) / bm) * ((N + bn - 1) / bn);
}


}  // namespace seedhip

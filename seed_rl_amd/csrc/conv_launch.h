// Host-side launch helpers shared by conv.hip and stackconv.hip.
#pragma once
#include "common.h"

namespace seedhip {

// One reduction job: out[i] = sum_z partial[z][i], i < n (fixed order => deterministic).  The kernels below take TWO
// jobs with the same slice count -- a weight gradient and its bias gradient -- so that the pair costs one launch
// (each of these launches is a few microseconds of drain + dispatch against ~1 us of work).
struct ReduceJob { const float* partial; long long n; float* out; };

// float4 vectorised, one thread walks all slices of its columns.  Blocks [0, blocks_a) serve job a, the rest job b.
static __global__ void __launch_bounds__(256)
reduce_slices_kernel(const ReduceJob ja, const ReduceJob jb, int blocks_a, int slices) {
  const bool first = (int)blockIdx.x < blocks_a;
  const ReduceJob j = first ? ja : jb;
  const long long bx = first ? blockIdx.x : blockIdx.x - blocks_a, nb = first ? blocks_a : gridDim.x - blocks_a;
  const float* __restrict__ partial = j.partial; float* __restrict__ out = j.out; const long long n = j.n;
  const long long stride = nb * blockDim.x;
  const long long n4 = n >> 2;
  for (long long i = bx * blockDim.x + threadIdx.x; i < n4; i += stride) {
    float4 a = reinterpret_cast<const float4*>(partial)[i];
    for (int z = 1; z < slices; ++z) {
      const float4 b = reinterpret_cast<const float4*>(partial + (long long)z * n)[i];
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    reinterpret_cast<float4*>(out)[i] = a;
  }
  for (long long i = (n4 << 2) + bx * blockDim.x + threadIdx.x; i < n; i += stride) {
    float a = partial[i];
    for (int z = 1; z < slices; ++z) a += partial[(long long)z * n + i];
    out[i] = a;
  }
}

// Many slices (persistent split-K workgroups write one slice each): 8 float4 columns x 32 slice lanes per block (a wave
// touches eight 128-byte runs per load instruction); lane q sums slices q, q + 32, ... with four independent loads in
// flight per round, then the 32 lanes are combined through LDS in a fixed order.  Deterministic.  (r5: 16 columns x 16
// lanes left half the chip idle on an [8192]-float gradient -- 128 blocks -- and walked 32 slices per thread: 10.4 us
// for 512 slices of the second Atari conv's gradient, 135 launches per cfg3 step.)
static __global__ void __launch_bounds__(256)
reduce_slices_wide_kernel(const ReduceJob ja, const ReduceJob jb, int blocks_a, int slices) {
  __shared__ float4 red[32][9];
  const bool first = (int)blockIdx.x < blocks_a;
  const ReduceJob j = first ? ja : jb;
  const long long bx = first ? blockIdx.x : blockIdx.x - blocks_a;
  const float* __restrict__ partial = j.partial; float* __restrict__ out = j.out; const long long n = j.n;
  const int col = threadIdx.x & 7, sl = threadIdx.x >> 3;
  const long long n4 = n >> 2;
  const long long c4 = bx * 8 + col;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c4 < n4) {
    int z = sl;
    for (; z + 96 < slices; z += 128) {
      const float4 b0 = reinterpret_cast<const float4*>(partial + (long long)z * n)[c4];
      const float4 b1 = reinterpret_cast<const float4*>(partial + (long long)(z + 32) * n)[c4];
      const float4 b2 = reinterpret_cast<const float4*>(partial + (long long)(z + 64) * n)[c4];
      const float4 b3 = reinterpret_cast<const float4*>(partial + (long long)(z + 96) * n)[c4];
      a.x += b0.x; a.y += b0.y; a.z += b0.z; a.w += b0.w;
      a.x += b1.x; a.y += b1.y; a.z += b1.z; a.w += b1.w;
      a.x += b2.x; a.y += b2.y; a.z += b2.z; a.w += b2.w;
      a.x += b3.x; a.y += b3.y; a.z += b3.z; a.w += b3.w;
    }
    for (; z < slices; z += 32) {
      const float4 b = reinterpret_cast<const float4*>(partial + (long long)z * n)[c4];
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
  }
  red[sl][col] = a;
  __syncthreads();
  if (sl == 0 && c4 < n4) {
    float4 t = red[0][col];
    for (int q = 1; q < 32; ++q) { const float4 b = red[q][col]; t.x += b.x; t.y += b.y; t.z += b.z; t.w += b.w; }
    reinterpret_cast<float4*>(out)[c4] = t;
  }
}

// (pw, nw, dw) and, when db != nullptr, (pb, nb, db): same slice count; one launch when both jobs take the same
// kernel (so every output keeps the summation order it had as a separate launch).
static inline void reduce_slices2(const float* pw, long long nw, float* dw, const float* pb, long long nb, float* db,
                                  int slices, hipStream_t s) {
  auto wide_ok = [&](const float* p, long long n, const float* o) {
    return slices >= 32 && (n & 3) == 0 && ((((uintptr_t)p) | ((uintptr_t)o)) & 15) == 0;
  };
  const bool wa = wide_ok(pw, nw, dw);
  if (db && wide_ok(pb, nb, db) != wa) {                       // mixed: two launches, as before
    reduce_slices2(pw, nw, dw, nullptr, 0, nullptr, slices, s);
    reduce_slices2(pb, nb, db, nullptr, 0, nullptr, slices, s);
    return;
  }
  const ReduceJob ja{pw, nw, dw};
  const ReduceJob jb{db ? pb : pw, db ? nb : 0, db ? db : dw};
  if (wa) {
    const int ba = cdiv(nw >> 2, 8), bb = jb.n ? cdiv(jb.n >> 2, 8) : 0;
    hipLaunchKernelGGL(reduce_slices_wide_kernel, dim3(ba + bb), dim3(256), 0, s, ja, jb, ba, slices);
    return;
  }
  int ba = cdiv(nw / 4 + 1, 256); if (ba > 1024) ba = 1024;
  int bb = jb.n ? cdiv(jb.n / 4 + 1, 256) : 0; if (bb > 1024) bb = 1024;
  hipLaunchKernelGGL(reduce_slices_kernel, dim3(ba + bb), dim3(256), 0, s, ja, jb, ba, slices);
}

static inline void reduce_slices(const float* partial, int slices, long long n, float* out, hipStream_t s) {
  reduce_slices2(partial, n, out, nullptr, 0, nullptr, slices, s);
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

// Generic fp32 implicit-GEMM core on CDNA4 matrix cores (v_mfma_f32_16x16x4_f32).
//
// Every dense contraction on the learner hot path (conv forward / data-grad /
// weight-grad, Dense layers, LSTM projections) is phrased
// as C[M,N] = sum_k A[m,k] * B[k,n] where A and B are *accessors* supplied by a
// "problem" struct (im2col gather, transposed weights, u8 frames, ...).  The
// accessors are plain HOST+DEVICE functions, so the index math is unit-tested on
// the CPU (tests/host/emul.cpp) with the same code the GPU runs.
//
// Why fp32 MFMA: the reference computes in fp32 (no mixed precision anywhere);
// v_mfma_f32_16x16x4_f32 is an exact fp32 fmaf chain at the 157 TF vector-peak
// rate, 2.4x a VALU GEMM, so parity and speed are not in tension.
//
// Tiling (64-wide wavefronts, 4 waves = 256 threads per workgroup):
//   * workgroup tile BM x BN = (WM*MR*16) x (WN*NR*16); each wave owns MR x NR
//     16x16 accumulators (4 VGPRs each);
//   * A/B k-tiles (BK deep) are gathered global -> registers (16 B per lane) -> LDS: LDS is
//     double buffered and there are TWO register stages, so the loads of tile kt+2 are issued
//     while tile kt is multiplied and tile kt+1 waits in registers; ONE barrier per k-tile;
//   * LDS layouts make the MFMA fragment reads bank-conflict free:
//     A_s[BM][BK+2] (row stride == 2 mod 32 dwords, lanes (i,kq) -> bank 2i+kq),
//     B_s[BK][LDB] with LDB == 16 mod 32 (lanes (kq,j) -> bank 16kq+j).
//   * blockIdx.z = "slice": split-K chunk (weight-grad) or stride-parity class
//     (data-grad of strided convs).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define SH_HD __host__ __device__ __forceinline__
#else
#define SH_HD inline
struct float4 { float x, y, z, w; };
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
#endif

namespace seedhip {

// Division by a runtime-constant divisor with one mul-hi + shift (valid for 0 <= x < 2^31).
struct FastDiv {
  uint32_t d, mul, shift;
  void init(uint32_t div) {
    d = div;
    if (div <= 1) { mul = 0; shift = 0; return; }
    uint32_t l = 0;
    while ((1ull << l) < div) ++l;
    const uint32_t p = 31 + l;
    mul = (uint32_t)(((1ull << p) + div - 1) / div);
    shift = p - 32;
  }
  SH_HD uint32_t div(uint32_t x) const {
    if (d <= 1) return x;
#if defined(__HIP_DEVICE_COMPILE__)
    return __umulhi(x, mul) >> shift;
#else
    return (uint32_t)(((uint64_t)x * mul) >> 32) >> shift;
#endif
  }
  SH_HD void divmod(uint32_t x, uint32_t& q, uint32_t& r) const { q = div(x); r = x - q * d; }
};

SH_HD float f4_get(const float4& v, int i) { return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w; }
SH_HD float relu_if(float v, int on) { return (on && v < 0.f) ? 0.f : v; }

#if defined(__HIPCC__)

typedef float f32x4_t __attribute__((ext_vector_type(4)));

template <int BN> struct LdB { static constexpr int value = (BN % 32 == 16) ? BN : BN + 16; };

// P: problem (see conv_problems.h).  Required interface:
//   static constexpr bool kAVecK, kBVecN, kColSumB;
//   int M, N;                                   GEMM extents
//   void k_range(int z, int& k0, int& k1)       reduction range of slice z (k0 % 4 == 0)
//   ARow a_row(int m, int z); float4 load_a(const ARow&, int k, int z)
//        kAVecK: elements (m, k..k+3); else (m..m+3, k)
//   BCol b_col(int n, int z); float4 load_b(const BCol&, int k, int z)
//        kBVecN: elements (k, n..n+3); else (k..k+3, n)
//   void store(int m, int n, float v, int z); void store_colsum(int n, float v, int z)
template <class P, int WM, int WN, int MR, int NR, int BK>
__global__ void __launch_bounds__(256)
igemm_kernel(const P p) {
  static_assert(WM * WN == 4, "4 waves per workgroup");
  static_assert(BK % 4 == 0, "BK multiple of 4");
  constexpr int BM = WM * MR * 16, BN = WN * NR * 16;
  constexpr int LDA = BK + 2;
  constexpr int LDB = LdB<BN>::value;
  constexpr int NVA = BM * BK / 4;                 // float4 vectors in an A tile
  constexpr int NVB = BK * BN / 4;
  constexpr int VA = (NVA + 255) / 256, VB = (NVB + 255) / 256;
  constexpr int A_CHUNKS = P::kAVecK ? BK / 4 : BM / 4;   // vectors along the fast axis
  constexpr int B_CHUNKS = P::kBVecN ? BN / 4 : BK / 4;
  static_assert(NVA % 256 == 0 || NVA < 256, "A tile vectors must tile 256 threads");
  static_assert(VB == 1, "one B vector per thread (B tile <= 256 float4)");
  static_assert(!P::kColSumB || P::kBVecN, "column sums need n-contiguous B staging");

  __shared__ __attribute__((aligned(16))) float smem[2 * BM * LDA + 2 * BK * LDB];
  float* As = smem;
  float* Bs = smem + 2 * BM * LDA;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int z = blockIdx.z;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  int k0, k1;
  p.k_range(z, k0, k1);
  const int nkt = (k1 - k0 + BK - 1) / BK;

  // Per-thread staging assignments (fixed across k-tiles).
  typename P::ARow arow[VA];
  int a_slow[VA], a_fast[VA];
#pragma unroll
  for (int i = 0; i < VA; ++i) {
    const int v = tid + i * 256;
    a_slow[i] = v / A_CHUNKS; a_fast[i] = (v % A_CHUNKS) * 4;
    if constexpr (P::kAVecK) arow[i] = p.a_row(m0 + a_slow[i], z);      // slow = m, fast = k
    else arow[i] = p.a_row(m0 + a_fast[i], z);                           // slow = k, fast = m
  }
  typename P::BCol bcol[VB];
  int b_slow[VB], b_fast[VB];
#pragma unroll
  for (int i = 0; i < VB; ++i) {
    const int v = tid + i * 256;
    b_slow[i] = v / B_CHUNKS; b_fast[i] = (v % B_CHUNKS) * 4;
    if constexpr (P::kBVecN) bcol[i] = p.b_col(n0 + b_fast[i], z);      // slow = k, fast = n
    else bcol[i] = p.b_col(n0 + b_slow[i], z);                           // slow = n, fast = k
  }

  float4 ra0[VA], rb0[VB], ra1[VA], rb1[VB];     // two register stages: tiles kt+1 and kt+2 in flight
  float4 colsum[VB];
#pragma unroll
  for (int i = 0; i < VB; ++i) colsum[i] = make_float4(0.f, 0.f, 0.f, 0.f);

  auto load_tile = [&](int kt, float4 (&ra)[VA], float4 (&rb)[VB]) {
    const int kb = k0 + kt * BK;
#pragma unroll
    for (int i = 0; i < VA; ++i) {
      if (NVA >= 256 || tid + i * 256 < NVA) {
        const int k = kb + (P::kAVecK ? a_fast[i] : a_slow[i]);
        ra[i] = (k < k1) ? p.load_a(arow[i], k, z) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
#pragma unroll
    for (int i = 0; i < VB; ++i) {
      if (NVB >= 256 || tid + i * 256 < NVB) {
        const int k = kb + (P::kBVecN ? b_slow[i] : b_fast[i]);
        rb[i] = (k < k1) ? p.load_b(bcol[i], k, z) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  };
  auto store_tile = [&](int buf, float4 (&ra)[VA], float4 (&rb)[VB]) {
    float* A = As + buf * BM * LDA;
    float* B = Bs + buf * BK * LDB;
#pragma unroll
    for (int i = 0; i < VA; ++i) {
      if (NVA >= 256 || tid + i * 256 < NVA) {
        if constexpr (P::kAVecK) {
          float* d = A + a_slow[i] * LDA + a_fast[i];        // 8-B aligned (LDA even, fast % 4 == 0)
          *reinterpret_cast<float2*>(d) = make_float2(ra[i].x, ra[i].y);
          *reinterpret_cast<float2*>(d + 2) = make_float2(ra[i].z, ra[i].w);
        } else {
          float* d = A + a_fast[i] * LDA + a_slow[i];
          d[0] = ra[i].x; d[LDA] = ra[i].y; d[2 * LDA] = ra[i].z; d[3 * LDA] = ra[i].w;
        }
      }
    }
#pragma unroll
    for (int i = 0; i < VB; ++i) {
      if (NVB >= 256 || tid + i * 256 < NVB) {
        if constexpr (P::kColSumB) {                   // bias gradient: summed when the tile is consumed, not when its
          colsum[i].x += rb[i].x; colsum[i].y += rb[i].y; colsum[i].z += rb[i].z; colsum[i].w += rb[i].w;   // load is issued
        }
        if constexpr (P::kBVecN) {
          *reinterpret_cast<float4*>(B + b_slow[i] * LDB + b_fast[i]) = rb[i];   // LDB % 4 == 0
        } else {
          float* d = B + b_fast[i] * LDB + b_slow[i];
          d[0] = rb[i].x; d[LDB] = rb[i].y; d[2 * LDB] = rb[i].z; d[3 * LDB] = rb[i].w;
        }
      }
    }
  };

  f32x4_t acc[MR][NR];
#pragma unroll
  for (int i = 0; i < MR; ++i)
#pragma unroll
    for (int j = 0; j < NR; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // Pipeline: LDS double buffer + TWO register stages.  While tile kt is multiplied out of LDS, tile kt+1
  // sits in one register set (stored to the other LDS buffer after the MFMAs) and the global loads of tile
  // kt+2 are issued into the other set -- two k-tiles (~2 x 1000 matrix-pipe cycles) of load latency cover,
  // which matters at the 1-2 workgroups per CU these layer shapes give.
  if (nkt > 0) { load_tile(0, ra0, rb0); store_tile(0, ra0, rb0); }
  if (nkt > 1) load_tile(1, ra1, rb1);
  __syncthreads();

  const int a_frag_off = (wm * MR * 16 + (lane & 15)) * LDA + (lane >> 4);
  const int b_frag_off = (lane >> 4) * LDB + wn * NR * 16 + (lane & 15);
  auto step = [&](int kt, float4 (&la)[VA], float4 (&lb)[VB], float4 (&sa)[VA], float4 (&sb)[VB]) {
    const int buf = kt & 1;
    if (kt + 2 < nkt) load_tile(kt + 2, la, lb);
    const float* A = As + buf * BM * LDA + a_frag_off;
    const float* B = Bs + buf * BK * LDB + b_frag_off;
#pragma unroll
    for (int kk = 0; kk < BK / 4; ++kk) {
      float af[MR], bf[NR];
#pragma unroll
      for (int i = 0; i < MR; ++i) af[i] = A[i * 16 * LDA + kk * 4];
#pragma unroll
      for (int j = 0; j < NR; ++j) bf[j] = B[kk * 4 * LDB + j * 16];
#pragma unroll
      for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < NR; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nkt) store_tile(buf ^ 1, sa, sb);
    __syncthreads();
  };
  for (int kt = 0; kt < nkt; kt += 2) {
    step(kt, ra0, rb0, ra1, rb1);                    // even: tile kt+2 -> set 0, tile kt+1 (set 1) -> LDS
    if (kt + 1 < nkt) step(kt + 1, ra1, rb1, ra0, rb0);
  }

  // Epilogue: C[row = 4*(lane>>4) + reg][col = lane & 15] per 16x16 tile.
#pragma unroll
  for (int i = 0; i < MR; ++i) {
#pragma unroll
    for (int j = 0; j < NR; ++j) {
      const int n = n0 + wn * NR * 16 + j * 16 + (lane & 15);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wm * MR * 16 + i * 16 + 4 * (lane >> 4) + r;
        if (m < p.M && n < p.N) p.store(m, n, acc[i][j][r], z);
      }
    }
  }

  if constexpr (P::kColSumB) {
    // Column sums of B over this slice's k range (bias gradient): only the first M-tile emits.
    if (blockIdx.x == 0) {
      __syncthreads();
      float* red = smem;                               // reuse: [256 / B_CHUNKS][BN] floats
      if (NVB >= 256 || tid < NVB) {
        float* d = red + (tid / B_CHUNKS) * BN + b_fast[0];
        d[0] = colsum[0].x; d[1] = colsum[0].y; d[2] = colsum[0].z; d[3] = colsum[0].w;
      }
      __syncthreads();
      constexpr int ROWS = (NVB >= 256 ? 256 : NVB) / B_CHUNKS;
      if (tid < BN) {
        float s = 0.f;
        for (int r = 0; r < ROWS; ++r) s += red[r * BN + tid];
        if (n0 + tid < p.N) p.store_colsum(n0 + tid, s, z);
      }
    }
  }
}

template <class P, int WM, int WN, int MR, int NR, int BK>
inline void launch_igemm(const P& p, int slices, hipStream_t s) {
  constexpr int BM = WM * MR * 16, BN = WN * NR * 16;
  dim3 grid((p.M + BM - 1) / BM, (p.N + BN - 1) / BN, slices);
  hipLaunchKernelGGL((igemm_kernel<P, WM, WN, MR, NR, BK>), grid, dim3(256), 0, s, p);
}

// Tile-shape dispatch on the GEMM N extent (output channels).
template <class P>
inline void launch_igemm_auto(const P& p, int slices, hipStream_t s) {
  if (p.N <= 16) launch_igemm<P, 4, 1, 4, 1, 16>(p, slices, s);          // 256 x 16
  else if (p.N <= 32) launch_igemm<P, 4, 1, 2, 2, 16>(p, slices, s);     // 128 x 32
  else if (p.N <= 48) launch_igemm<P, 4, 1, 2, 3, 16>(p, slices, s);     // 128 x 48
  else {
    // 128 x 64 tiles over N; 64 x 64 when that leaves the chip under ~2 workgroups per CU (latency hiding
    // needs several waves per SIMD: one wave cannot cover its own staging VALU under the 32-cycle MFMAs)
    const long long wgs = (long long)((p.M + 127) / 128) * ((p.N + 63) / 64) * slices;
    if (wgs < 512 && p.M > 64) launch_igemm<P, 2, 2, 2, 2, 16>(p, slices, s);
    else launch_igemm<P, 2, 2, 4, 2, 16>(p, slices, s);
  }
}

#endif  // __HIPCC__

}  // namespace seedhip

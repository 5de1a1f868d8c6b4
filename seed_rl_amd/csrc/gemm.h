// fp32 GEMM on the matrix cores (v_mfma_f32_16x16x4_f32) for the Dense layers and the LSTM projections
// (dmlab/networks.py:105-124,152-171, atari/networks.py:176-251):
//     C[m, n] = epilogue( sum_k A(m, k) * B(k, n) )
// Each operand is either "k-contiguous" (KC: element (x, k) at base[x*ld + k]) or "outer-contiguous"
// (OC: element (x, k) at base[k*ld + x]).  The three Dense GEMMs in Keras layouts (kernel [in, out]) are
//     forward        y  = x  W     : A = x  (KC),  B = W  (OC)
//     data gradient  dx = dy W^T   : A = dy (KC),  B = W  (KC: W[in, out] read as B(k = out, n = in))
//     weight gradient dW = x^T dy  : A = x  (OC),  B = dy (OC), k = the batch row
// and in every case each operand moves 16 bytes at a time -- global -> registers -> LDS -> MFMA fragment -- with
// no transposition anywhere:
//   * KC operand: LDS rows [x][k] (stride 40 floats: conflict-free b128).  A lane reads 4 consecutive k of its row
//     as one ds_read_b128 and feeds 4 MFMAs through the k-permutation (lane (x, kq) holds k = 4*kq + kk at step kk;
//     any bijection of k works as long as A and B agree).
//   * OC operand: LDS rows [k][x] exactly as in memory.  A lane reads 4 consecutive x of ONE k row as one b128 and
//     these feed 4 different 16-wide tiles: tile e of a wave owns x = 4*lane_x + e (an x-interleaved tile assignment,
//     undone in the epilogue).  For C this makes every lane own 4 consecutive n: 16-byte stores.
// Either way 8 ds_read_b128 feed 64 MFMAs (0.125 LDS reads per MFMA against 0.75 in the generic implicit-GEMM
// core); that matters because each ds_read costs the fp32 matrix pipe ~14 cycles (tools/probes/mfma_probe2.hip).
// One LDS buffer + register prefetch of the next k-tile; BK = 32; split-K over blockIdx.z with a deterministic
// second-pass reduction (fixed slice order).
#pragma once
#include "common.h"
#include "gemm_geom.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>

namespace seedhip {
namespace gemm {

typedef float f32x4_t __attribute__((ext_vector_type(4)));

// Staging of one operand tile [X rows/cols = BX][BK] through registers into LDS.
// OC rows are BX floats; with 2-wide fragments (ds_read_b64: lane groups {0-31}, {32-63} = two k rows 4 apart each)
// the row stride must move 4 rows by 32 banks: BX + 8.  4-wide fragments (b128, 16 lanes = 256 contiguous bytes per
// group) are conflict free at any stride.
template <int BX> struct LdOC { static constexpr int value = (BX == 64 || BX == 32) ? BX + 8 : BX; };

// (uniform pointer -> buffer resource over `bytes`; readfirstlane keeps the descriptor in SGPRs)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_view(const float* p, long long bytes) {
  const uint64_t ab = reinterpret_cast<uint64_t>(p);
  const uint64_t sb = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((unsigned)(ab >> 32)) << 32) |
                      (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((unsigned)ab);
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(sb), 0, __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000);
}
__device__ __forceinline__ float4 view_load(const __amdgpu_buffer_rsrc_t& r, unsigned byte_off) {
  typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
  const u32x4_t t = __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0);
  return make_float4(__uint_as_float(t[0]), __uint_as_float(t[1]), __uint_as_float(t[2]), __uint_as_float(t[3]));
}
constexpr unsigned kViewOOB = 0x80000000u;                 // past num_records of any view (< 2 GB): the load returns zeros

template <int BX, bool KC>
struct Stager {
  static constexpr int kVecs = BX * BK / 4 / 256;
  static constexpr int kLdOC = LdOC<BX>::value;
  static constexpr int kLdsFloats = KC ? BX * LD_KC : BK * kLdOC;
  const float* base[kVecs];   // pointer at k = 0 of this thread's vector (null: out of range in x)
  int lds_off[kVecs];
  int krow[kVecs];            // k offset of the vector inside a tile
  long long kstride;          // floats per unit k
  float4 r[kVecs];
  // VALU diet (a SIMD's VALU work does not overlap its MFMAs): when the operand fits a 2 GB buffer view, a vector
  // costs one 32-bit multiply-add for its byte offset and -- only in a ragged last k-tile -- one select that pushes
  // it out of range (the hardware returns zeros); no 64-bit pointer arithmetic, no branch, no select on the data.
  bool fast;
  unsigned boff[kVecs];       // byte offset at k = 0, or kViewOOB: out of range in x
  unsigned kstep;             // bytes per unit k
  __amdgpu_buffer_rsrc_t rsrc;

  // extent_k: the operand's extent along k (rows of an outer-contiguous operand); X its extent along x
  __device__ void init(const float* p, long long ld, int x0, int X, int tid, int extent_k = 0, bool allow = false) {
    const long long bytes = (KC ? (long long)X * ld : (long long)extent_k * ld) * 4;
    fast = allow && bytes > 0 && bytes < (1LL << 31) - 64;
    rsrc = make_view(p, fast ? bytes : 0);
#pragma unroll
    for (int i = 0; i < kVecs; ++i) {
      const int v = tid + i * 256;
      if (KC) {
        const int row = v >> 3, kc = (v & 7) * 4;
        krow[i] = kc; lds_off[i] = row * LD_KC + kc;
        base[i] = (x0 + row < X) ? p + (long long)(x0 + row) * ld + kc : nullptr;
        boff[i] = (x0 + row < X) ? (unsigned)(((long long)(x0 + row) * ld + kc) * 4) : kViewOOB;
      } else {
        const int kr = v / (BX / 4), x4 = (v % (BX / 4)) * 4;
        krow[i] = kr; lds_off[i] = kr * kLdOC + x4;
        base[i] = (x0 + x4 < X) ? p + (long long)kr * ld + x0 + x4 : nullptr;     // X % 4 == 0: all in or all out
        boff[i] = (x0 + x4 < X) ? (unsigned)(((long long)kr * ld + x0 + x4) * 4) : kViewOOB;
      }
    }
    kstride = KC ? 1 : ld;
    kstep = (unsigned)(KC ? 4 : ld * 4);
  }
  __device__ void load(int k, int k1, bool relu) {
    if (fast) {
      const unsigned kb = (unsigned)k * kstep;
      if (k + BK <= k1) {                       // uniform: a full k-tile needs no per-vector k check
#pragma unroll
        for (int i = 0; i < kVecs; ++i) r[i] = view_load(rsrc, boff[i] == kViewOOB ? kViewOOB : boff[i] + kb);
      } else {
#pragma unroll
        for (int i = 0; i < kVecs; ++i) r[i] = view_load(rsrc, (boff[i] == kViewOOB || k + krow[i] >= k1) ? kViewOOB : boff[i] + kb);
      }
      if (relu) {
#pragma unroll
        for (int i = 0; i < kVecs; ++i) { r[i].x = fmaxf(r[i].x, 0.f); r[i].y = fmaxf(r[i].y, 0.f); r[i].z = fmaxf(r[i].z, 0.f); r[i].w = fmaxf(r[i].w, 0.f); }
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < kVecs; ++i) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (base[i] && k + krow[i] < k1) {        // KC: K % 4 == 0 so the vector is entirely in or out
        v = *reinterpret_cast<const float4*>(base[i] + (long long)k * kstride);
        if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      }
      r[i] = v;
    }
  }
  __device__ void store(float* lds) const {
#pragma unroll
    for (int i = 0; i < kVecs; ++i) *reinterpret_cast<float4*>(lds + lds_off[i]) = r[i];
  }
};

template <int BX>
struct GatherStager {                        // KC only; same interface as Stager
  static constexpr int kVecs = BX * BK / 4 / 256;
  static constexpr int kLdOC = BX;
  static constexpr int kLdsFloats = BX * LD_KC;
  const float* base[kVecs];                  // row base + this thread's k offset inside a tile (null: row out of range)
  int y0[kVecs], x0[kVecs];
  int lds_off[kVecs];
  int kc;
  const Gather* g;
  float4 r[kVecs];

  __device__ void init(const float* p, const Gather& gg, int /*nkt*/, int xbase, int X, int tid) {
    g = &gg;
    kc = (tid & 7) * 4;
#pragma unroll
    for (int i = 0; i < kVecs; ++i) {
      const int v = tid + i * 256, row = v >> 3;
      lds_off[i] = row * LD_KC + kc;
      const float* b = nullptr;
      int yy = 0, xx = 0;
      if (xbase + row < X) {
        long long off;
        if (gather_row(gg, xbase + row, off, yy, xx)) b = p + off;
      }
      base[i] = b; y0[i] = yy; x0[i] = xx;
    }
  }
  __device__ void load(int k, int k1, bool relu) {
    int toff, dy, dx;                                        // same tap for all of this thread's vectors
    bool tap_ok;
    gather_tap(*g, k + kc, toff, dy, dx, tap_ok);
#pragma unroll
    for (int i = 0; i < kVecs; ++i) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      const int y = y0[i] + dy, x = x0[i] + dx;
      if (base[i] && tap_ok && gather_inside(*g, y, x)) {
        v = *reinterpret_cast<const float4*>(base[i] + toff);
        if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      }
      r[i] = v;
    }
  }
  __device__ void store(float* lds) const {
#pragma unroll
    for (int i = 0; i < kVecs; ++i) *reinterpret_cast<float4*>(lds + lds_off[i]) = r[i];
  }
};

// Outer-contiguous gathered operand: the im2col matrix of a convolution read TRANSPOSED, for the weight gradient
// dW[(ky, kx, c), co] = sum_pixels X[pixel; ky, kx, c] * dY[pixel, co].  The GEMM row x is the dW row (tap, channel):
// 4 consecutive channels of one tap are one float4 of the NHWC input; the reduction index k is the output pixel.
// Geometry = the forward conv's `Gather` (pixel -> (img, a, b), tap offsets, border predicate); the per-thread tap
// part is fixed for the whole launch, only the pixel part changes from k-tile to k-tile.
template <int BX>
struct GatherOCStager {
  static constexpr int kVecs = BX * BK / 4 / 256;
  static constexpr int kLdOC = LdOC<BX>::value;
  static constexpr int kLdsFloats = BK * kLdOC;
  static constexpr int kRowStep = 256 / (BX / 4);       // k rows between a thread's consecutive vectors
  int lds_off[kVecs], krow[kVecs];
  int toff, tdy, tdx;                        // this thread's tap: offset and border shift (same x4 for all its vectors)
  bool x_ok;
  const float* p0;
  const Gather* g;
  float4 r[kVecs];
  // VALU diet (a SIMD's VALU work does not overlap its MFMAs; this stager used to cost 6 VALU per MFMA): when the
  // tensor fits a 2 GB buffer view the loads go through a buffer resource -- ONE 32-bit byte offset per vector, set
  // out of range for border taps / tail rows (the hardware returns zeros: no select, no branch) -- and the rows after
  // a thread's first are stepped through (u, v, w) instead of divided.
  bool fast, step_ok;
  uint32_t rows_v;                           // extent of v (d1.d / d2.d)
  __amdgpu_buffer_rsrc_t rsrc;

  __device__ void init(const float* p, const Gather& gg, int /*nkt*/, int xbase, int X, int tid) {
    g = &gg; p0 = p;
    const int x4 = (tid % (BX / 4)) * 4, xg = xbase + x4;
    x_ok = xg < X;
    const int xc = x_ok ? xg : 0;
    bool tap_ok;
    gather_tap(gg, xc, toff, tdy, tdx, tap_ok);
    x_ok = x_ok && tap_ok;
#pragma unroll
    for (int i = 0; i < kVecs; ++i) {
      const int v = tid + i * 256;
      krow[i] = v / (BX / 4);
      lds_off[i] = krow[i] * kLdOC + x4;
    }
    fast = gg.extent > 0 && gg.extent < (1LL << 29) && !gg.coord_uv;
    rows_v = gg.d2.div(gg.d1.d);
    step_ok = (uint32_t)kRowStep <= gg.d2.d;
    rsrc = make_view(p, fast ? gg.extent * 4 : 0);
  }
  __device__ void load(int k, int k1, bool relu) {
    if (fast) {
      typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
      uint32_t u, v, w;
      gather_decode(*g, (uint32_t)(k + krow[0]), u, v, w);
#pragma unroll
      for (int i = 0; i < kVecs; ++i) {
        if (i) {
          if (step_ok) gather_step(*g, rows_v, kRowStep, u, v, w);
          else gather_decode(*g, (uint32_t)(k + krow[i]), u, v, w);
        }
        unsigned off;
        const bool inside = gather_elem32(*g, u, v, w, toff, tdy, tdx, off);
        const bool ok = x_ok && k + krow[i] < k1 && inside;
        const u32x4_t t = __builtin_amdgcn_raw_buffer_load_b128(rsrc, ok ? off : kViewOOB, 0, 0);
        r[i] = make_float4(__uint_as_float(t[0]), __uint_as_float(t[1]), __uint_as_float(t[2]), __uint_as_float(t[3]));
      }
      if (relu) {
#pragma unroll
        for (int i = 0; i < kVecs; ++i) { r[i].x = fmaxf(r[i].x, 0.f); r[i].y = fmaxf(r[i].y, 0.f); r[i].z = fmaxf(r[i].z, 0.f); r[i].w = fmaxf(r[i].w, 0.f); }
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < kVecs; ++i) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      const int pix = k + krow[i];
      if (x_ok && pix < k1) {
        long long off;
        int y0, x0;
        gather_row(*g, pix, off, y0, x0);
        if (gather_inside(*g, y0 + tdy, x0 + tdx)) {
          v = *reinterpret_cast<const float4*>(p0 + off + toff);
          if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        }
      }
      r[i] = v;
    }
  }
  __device__ void store(float* lds) const {
#pragma unroll
    for (int i = 0; i < kVecs; ++i) *reinterpret_cast<float4*>(lds + lds_off[i]) = r[i];
  }
};

template <int BX, bool KC, bool G> struct PickStager { typedef Stager<BX, KC> type; };
template <int BX> struct PickStager<BX, false, true> { typedef GatherOCStager<BX> type; };
template <int BX> struct PickStager<BX, true, true> { typedef GatherStager<BX> type; };

template <int R> struct FragVec;
template <> struct FragVec<4> { typedef f32x4_t type; };
template <> struct FragVec<2> { typedef float type __attribute__((ext_vector_type(2))); };

// AG / BG: the (KC) operand is gathered (struct Gather).  SCATTER: conv data-gradient epilogue, row m = super-pixel
// (img, a, b), column n = (py, px, ci) -> dx[img, es*a + py, es*b + px, ci].
// WN: waves along N -- 2: 2 x 2 waves, tile (2*MR*16) x (2*NR*16); 1: 4 x 1 waves, tile (4*MR*16) x (NR*16) for
// N <= 32 (the 32-channel conv layers).  MR, NR in {2, 4}.
template <int MR, int NR, bool AKC, bool BKC, bool AG = false, bool BG = false, bool SCATTER = false, int WN = 2>
__global__ void __launch_bounds__(256)
gemm_kernel(const Params p) {
  constexpr int WM = 4 / WN;
  constexpr int BM = WM * MR * 16, BN = WN * NR * 16;
  constexpr bool kSwap = AKC && BKC;
  typedef typename PickStager<BM, AKC, AG>::type SA;
  typedef typename PickStager<BN, BKC, BG>::type SB;
  __shared__ __attribute__((aligned(16))) float smem[SA::kLdsFloats + SB::kLdsFloats];
  float* As = smem;
  float* Bs = smem + SA::kLdsFloats;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN, lx = lane & 15, kq = lane >> 4;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int k0 = blockIdx.z * p.k_per_slice;
  int k1 = k0 + p.k_per_slice; if (k1 > p.K) k1 = p.K;
  const int nkt = (k1 - k0 + BK - 1) / BK;

  SA sa; SB sb;
  // buffer-view staging (see Stager) for the Dense GEMMs, where it measured faster (cfg2 FC trio: 0.482 -> 0.466 ms);
  // next to a gathered A (conv forward, cfg5) the weights as a buffer-view B measured 16-18 % SLOWER than pointer loads
  constexpr bool kView = !AG && !BG;
  if constexpr (AG) sa.init(p.A, p.ga, nkt, m0, p.M, tid); else sa.init(p.A, p.lda, m0, p.M, tid, p.K, kView);
  if constexpr (BG) sb.init(p.B, p.gb, nkt, n0, p.N, tid); else sb.init(p.B, p.ldb, n0, p.N, tid, p.K, kView);
  const bool do_colsum = !BKC && p.partial_colsum && blockIdx.x == 0;
  float4 csum[SB::kVecs];
#pragma unroll
  for (int i = 0; i < SB::kVecs; ++i) csum[i] = make_float4(0.f, 0.f, 0.f, 0.f);

  f32x4_t acc[MR][NR];
#pragma unroll
  for (int i = 0; i < MR; ++i)
#pragma unroll
    for (int j = 0; j < NR; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // fragment base addresses: KC [x][k] rows of this lane's x, 4 consecutive k at 4*kq; OC [k][x] row 4*kq (+kk),
  // R consecutive x at R*lx
  constexpr int LDA_OC = SA::kLdOC, LDB_OC = SB::kLdOC;
  const float* a_frag = AKC ? As + (wm * MR * 16 + lx) * LD_KC + 4 * kq : As + (4 * kq) * LDA_OC + wm * MR * 16 + MR * lx;
  const float* b_frag = BKC ? Bs + (wn * NR * 16 + lx) * LD_KC + 4 * kq : Bs + (4 * kq) * LDB_OC + wn * NR * 16 + NR * lx;

  // k-tiles to visit: all of them -- or, for a conv data gradient in image-block x position order (Gather::blk: this
  // tile's rows share one super-pixel), only those whose tap lies inside dY: the others are exact zeros for every row
  // (DQN conv3, 3x3 'valid' on 9x9: 40 % of the (position, tap) pairs)
  uint32_t kmask = 0;
  bool skipk = false;
  if constexpr (AG && SCATTER) {
    if (p.ga.blk && nkt <= 32) {
      skipk = true;
      uint32_t t = (uint32_t)m0 / (uint32_t)p.ga.blk, ib, pos, v, w;
      p.ga.d1.divmod(t, ib, pos);
      p.ga.d2.divmod(pos, v, w);
      for (int kt = 0; kt < nkt; ++kt) {
        int toff, dy, dx; bool tap_ok;
        gather_tap(p.ga, k0 + kt * BK, toff, dy, dx, tap_ok);
        if (tap_ok && gather_inside(p.ga, (int)v * p.ga.cy + p.ga.oy0 + dy, (int)w * p.ga.cx + p.ga.ox0 + dx)) kmask |= 1u << kt;
      }
      kmask = __builtin_amdgcn_readfirstlane(kmask);
    }
  }
  const int nvisit = skipk ? __popc(kmask) : nkt;
  int kt_seq = 0;
  auto next_kt = [&]() -> int {
    if (!skipk) return kt_seq++;
    const int b = __ffs(kmask) - 1;
    kmask &= kmask - 1;
    return b;
  };
  if (nvisit > 0) { const int kt = next_kt(); sa.load(k0 + kt * BK, k1, p.a_relu != 0); sb.load(k0 + kt * BK, k1, false); }
  for (int it = 0; it < nvisit; ++it) {
    __syncthreads();                                       // previous tile consumed
    sa.store(As); sb.store(Bs);
    if (do_colsum) {
#pragma unroll
      for (int i = 0; i < SB::kVecs; ++i) { csum[i].x += sb.r[i].x; csum[i].y += sb.r[i].y; csum[i].z += sb.r[i].z; csum[i].w += sb.r[i].w; }
    }
    __syncthreads();
    if (it + 1 < nvisit) { const int kt = next_kt(); sa.load(k0 + kt * BK, k1, p.a_relu != 0); sb.load(k0 + kt * BK, k1, false); }
#pragma unroll
    for (int h = 0; h < 2; ++h) {                          // two 16-deep halves; lane (x, kq) holds k = 16h + 4kq + kk
      f32x4_t a_kc[MR], b_kc[NR];
      typename FragVec<MR>::type a_oc[4];
      typename FragVec<NR>::type b_oc[4];
      if (AKC) {
#pragma unroll
        for (int i = 0; i < MR; ++i) a_kc[i] = *reinterpret_cast<const f32x4_t*>(a_frag + i * 16 * LD_KC + h * 16);
      } else {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) a_oc[kk] = *reinterpret_cast<const typename FragVec<MR>::type*>(a_frag + (h * 16 + kk) * LDA_OC);
      }
      if (BKC) {
#pragma unroll
        for (int j = 0; j < NR; ++j) b_kc[j] = *reinterpret_cast<const f32x4_t*>(b_frag + j * 16 * LD_KC + h * 16);
      } else {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) b_oc[kk] = *reinterpret_cast<const typename FragVec<NR>::type*>(b_frag + (h * 16 + kk) * LDB_OC);
      }
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int i = 0; i < MR; ++i)
#pragma unroll
          for (int j = 0; j < NR; ++j)
          {
            // both operands k-contiguous (Dense data gradient): the operands are SWAPPED -- B supplies the instruction's
            // rows -- so that a lane ends up with four consecutive n of ONE row m: 16-byte epilogue accesses (mask load,
            // store) instead of four 4-byte ones per accumulator
            if constexpr (kSwap)
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(b_kc[j][kk], a_kc[i][kk], acc[i][j], 0, 0, 0);
            else
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(AKC ? a_kc[i][kk] : a_oc[kk][i], BKC ? b_kc[j][kk] : b_oc[kk][j],
                                                               acc[i][j], 0, 0, 0);
          }
    }
  }

  // bias-gradient column sums of this slice: per-thread partials -> LDS [groups][BN] -> fixed-order sum
  if (do_colsum) {
    __syncthreads();
    constexpr int kPerRow = BN / 4, kGroups = 256 / kPerRow;
    static_assert(kGroups * BN <= SA::kLdsFloats + SB::kLdsFloats, "colsum scratch");
    float4 t = csum[0];
#pragma unroll
    for (int i = 1; i < SB::kVecs; ++i) { t.x += csum[i].x; t.y += csum[i].y; t.z += csum[i].z; t.w += csum[i].w; }
    *reinterpret_cast<float4*>(smem + (tid / kPerRow) * BN + (tid % kPerRow) * 4) = t;
    __syncthreads();
    if (tid < BN && n0 + tid < p.N) {
      float s = 0.f;
#pragma unroll
      for (int g = 0; g < kGroups; ++g) s += smem[g * BN + tid];
      p.partial_colsum[(long long)blockIdx.z * p.N + n0 + tid] = s;
    }
  }

  // epilogue.  MFMA C layout: row 4*kq + r, column lx of each 16x16 tile; tile (i, j) of the wave covers
  //   m = wm*MR*16 + (AKC ? 16*i + row : MR*row + i),   n = wn*NR*16 + (BKC ? 16*j + col : NR*col + j)
  if constexpr (kSwap) {
    // swapped operands: accumulator (i, j) of lane (lx, kq) holds C[m = 16 i + lx][n = 16 j + 4 kq + r]
    const bool vec = (p.ldc & 3) == 0 && (p.N & 3) == 0 &&
        ((((uintptr_t)p.C) | ((uintptr_t)p.mask) | ((uintptr_t)p.add) | ((uintptr_t)p.residual) | ((uintptr_t)p.bias) |
          ((uintptr_t)p.partial)) & 15) == 0;
#pragma unroll
    for (int i = 0; i < MR; ++i) {
      const int m = m0 + wm * MR * 16 + 16 * i + lx;
      if (m >= p.M) continue;
#pragma unroll
      for (int j = 0; j < NR; ++j) {
        const int n = n0 + wn * NR * 16 + 16 * j + 4 * kq;
        if (n >= p.N) continue;
        f32x4_t v = acc[i][j];
        if constexpr (SCATTER) {
          // conv data gradient: column n = (py, px, ci) with ci fastest -- four consecutive n are four consecutive
          // channels of ONE dX pixel when cin % 4 == 0: one scatter address, one 16-byte mask load and store
          const bool quad = vec && (p.gb.d2.d & 3u) == 0;
          if (quad) {
            long long at;
            if (!scatter_addr(p.ga, p.gb, p.es, p.eih, p.eiw, p.ldc, m, n, at)) continue;
            if (p.mask) {
              const f32x4_t mv = *reinterpret_cast<const f32x4_t*>(p.mask + at);
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] = mv[r] > 0.f ? v[r] : 0.f;
            }
            if (p.add) v += *reinterpret_cast<const f32x4_t*>(p.add + at);
            *reinterpret_cast<f32x4_t*>(p.C + at) = v;
          } else {
            for (int r = 0; r < 4; ++r) {
              long long at;
              if (n + r >= p.N || !scatter_addr(p.ga, p.gb, p.es, p.eih, p.eiw, p.ldc, m, n + r, at)) continue;
              float o = v[r];
              if (p.mask && !(p.mask[at] > 0.f)) o = 0.f;
              if (p.add) o += p.add[at];
              p.C[at] = o;
            }
          }
          continue;
        }
        if (p.partial) {
          float* dst = p.partial + ((long long)blockIdx.z * p.M + m) * p.N + n;
          if (vec) *reinterpret_cast<f32x4_t*>(dst) = v;
          else { for (int r = 0; r < 4; ++r) if (n + r < p.N) dst[r] = v[r]; }
          continue;
        }
        const long long at = (long long)m * p.ldc + n;
        if (vec) {
          if (p.bias) v += *reinterpret_cast<const f32x4_t*>(p.bias + n);
          if (p.residual) v += *reinterpret_cast<const f32x4_t*>(p.residual + at);
          if (p.out_relu) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
          }
          if (p.mask) {
            const f32x4_t mv = *reinterpret_cast<const f32x4_t*>(p.mask + at);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = mv[r] > 0.f ? v[r] : 0.f;
          }
          if (p.add) v += *reinterpret_cast<const f32x4_t*>(p.add + at);
          *reinterpret_cast<f32x4_t*>(p.C + at) = v;
        } else {
          for (int r = 0; r < 4; ++r) {
            if (n + r >= p.N) break;
            float o = v[r];
            if (p.bias) o += p.bias[n + r];
            if (p.residual) o += p.residual[at + r];
            if (p.out_relu && o < 0.f) o = 0.f;
            if (p.mask && !(p.mask[at + r] > 0.f)) o = 0.f;
            if (p.add) o += p.add[at + r];
            p.C[at + r] = o;
          }
        }
      }
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < MR; ++i) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 4 * kq + r;
      const int m = m0 + wm * MR * 16 + (AKC ? 16 * i + row : MR * row + i);
      if (m >= p.M) continue;
      float v[NR];
#pragma unroll
      for (int j = 0; j < NR; ++j) v[j] = acc[i][j][r];
      if (SCATTER) {
#pragma unroll
        for (int j = 0; j < NR; ++j) {
          const int n = n0 + wn * NR * 16 + 16 * j + lx;
          long long at;
          if (n >= p.N || !scatter_addr(p.ga, p.gb, p.es, p.eih, p.eiw, p.ldc, m, n, at)) continue;
          float o = v[j];
          if (p.mask && !(p.mask[at] > 0.f)) o = 0.f;
          if (p.add) o += p.add[at];
          p.C[at] = o;
        }
      } else if (BKC) {
#pragma unroll
        for (int j = 0; j < NR; ++j) {
          const int n = n0 + wn * NR * 16 + 16 * j + lx;
          if (n >= p.N) continue;
          float o = v[j];
          if (p.partial) { p.partial[((long long)blockIdx.z * p.M + m) * p.N + n] = o; continue; }
          const long long at = (long long)m * p.ldc + n;
          if (p.bias) o += p.bias[n];
          if (p.residual) o += p.residual[at];
          if (p.out_relu && o < 0.f) o = 0.f;
          if (p.mask && !(p.mask[at] > 0.f)) o = 0.f;
          if (p.add) o += p.add[at];
          p.C[at] = o;
        }
      } else {
        const int n = n0 + wn * NR * 16 + NR * lx;        // NR consecutive columns; N % 4 == 0
        if (n >= p.N) continue;
        typedef typename FragVec<NR>::type vec_t;
        if (p.partial) {
          vec_t o;
#pragma unroll
          for (int j = 0; j < NR; ++j) o[j] = v[j];
          *reinterpret_cast<vec_t*>(p.partial + ((long long)blockIdx.z * p.M + m) * p.N + n) = o;
          continue;
        }
        const long long at = (long long)m * p.ldc + n;
#pragma unroll
        for (int j = 0; j < NR; ++j) {
          if (n + j >= p.N) break;
          float o = v[j];
          if (p.bias) o += p.bias[n + j];
          if (p.residual) o += p.residual[at + j];
          if (p.out_relu && o < 0.f) o = 0.f;
          if (p.mask && !(p.mask[at + j] > 0.f)) o = 0.f;
          if (p.add) o += p.add[at + j];
          v[j] = o;
        }
        if ((p.ldc & 3) == 0 && n + NR <= p.N) {
          vec_t o;
#pragma unroll
          for (int j = 0; j < NR; ++j) o[j] = v[j];
          *reinterpret_cast<vec_t*>(p.C + at) = o;
        } else {
#pragma unroll
          for (int j = 0; j < NR; ++j) if (n + j < p.N) p.C[at + j] = v[j];
        }
      }
    }
  }
}

// Tile shape and split-K from a small cost model calibrated on MI355X (tools/bench_kernels.py gemm):
//   * a workgroup costs (k-tiles + ovh) units of MR*NR work; ovh = prologue + epilogue in k-tile units
//     (64x64: 2.7, 128x64: 4.7, 128x128: 7) -- short K wants small tiles, long K the 128x128 tile;
//   * a CU saturates the matrix pipe with `sat` resident workgroups (4 / 2.5 / 2) and holds at most `occ` (5 / 3 / 2);
//     fewer resident workgroups each run at 1/sat of the CU rate; the busiest CU sets the time;
//   * split-K adds the reduce pass: (slices + 1) * M * N floats through HBM + one launch.
struct Plan { int mr, nr, slices, k_per_slice; double model_us; int wn; };
// bias44 scales the modelled cost of the 128x128 tile: the weight-gradient GEMMs (both operands outer-contiguous, long
// K split into slices) run 5-15 % faster on it than the model predicts (tools/bench_kernels.py gemm).
inline Plan plan(int M, int N, int K, double bias44 = 1.0) {
  struct Tile { int mr, nr, occ; double sat, ovh; int wn; };
  // 2x2-wave tiles 64x64 / 128x64 / 128x128, and for N <= 32 the 4x1-wave tiles 128x32 / 256x32
  static const Tile kTiles[5] = {{2, 2, 5, 4.0, 2.7, 2}, {4, 2, 3, 2.5, 4.7, 2}, {4, 4, 2, 2.0, 6.0, 2},
                                 {2, 2, 5, 4.0, 2.7, 1}, {4, 2, 3, 2.5, 4.7, 1}};
  static const int kSlices[] = {1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64, 96, 128, 192, 256, 384, 512};
  constexpr int force = 0, force_s = 0;                     // (r2 / r3: plans forced from the environment for the cost model's calibration)
  Plan best{2, 2, 1, (K + BK - 1) / BK * BK, 1e30, 2};
  for (const Tile& t : kTiles) {
    if (force && force != t.mr * 10 + t.nr + 100 * (t.wn == 1)) continue;
    if (t.wn == 1 && N > 16 * t.nr) continue;                                // 4x1 tiles are one column tile wide
    const int bm = (4 / t.wn) * t.mr * 16, bn = t.wn * t.nr * 16;
    const long long tiles = (long long)((M + bm - 1) / bm) * ((N + bn - 1) / bn);
    for (int s : kSlices) {
      if (force_s) { if (s != force_s) continue; }
      else if (s > 1 && K / s < 4 * BK) break;
      int per = (K + s - 1) / s; per = (per + BK - 1) / BK * BK;
      const int slices = (K + per - 1) / per;
      const double wg_units = (per / BK + t.ovh) * t.mr * t.nr;              // work of one workgroup
      const double wpc = (double)tiles * slices / 256.0;                     // workgroups per CU
      const double full = floor(wpc / t.occ), rem = ceil(wpc - full * t.occ - 1e-9);
      const double serial = full * t.occ + (rem > 0 ? (rem > t.sat ? rem : t.sat) : 0.0);   // workgroup-times, busiest CU
      // one unit = 32x32x32 MACs; a saturated CU retires 128 MAC/cycle (4 SIMDs x 1024 MACs / 32 cycles) at 2.4 GHz,
      // of which this kernel sustains ~80%
      const double unit_us = 32.0 * 32 * 32 / 128.0 / 2400.0 / 0.80;
      double us = serial * wg_units * unit_us;
      if (slices > 1) us += 4.0 + (slices + 1.0) * M * N * 4.0 / 3.0e6;
      if (t.mr == 4 && t.nr == 4) us *= bias44;
      if (us < best.model_us) best = Plan{t.mr, t.nr, slices, per, us, t.wn};
    }
  }
  return best;
}

template <bool AKC, bool BKC, bool AG = false, bool BG = false, bool SCATTER = false>
inline void launch(const Params& p, const Plan& pl, hipStream_t s) {
  const int bm = (4 / pl.wn) * pl.mr * 16, bn = pl.wn * pl.nr * 16;
  dim3 grid((p.M + bm - 1) / bm, (p.N + bn - 1) / bn, pl.slices);
  if (pl.wn == 1 && pl.mr == 4) hipLaunchKernelGGL((gemm_kernel<4, 2, AKC, BKC, AG, BG, SCATTER, 1>), grid, dim3(256), 0, s, p);
  else if (pl.wn == 1) hipLaunchKernelGGL((gemm_kernel<2, 2, AKC, BKC, AG, BG, SCATTER, 1>), grid, dim3(256), 0, s, p);
  else if (pl.mr == 4 && pl.nr == 4) hipLaunchKernelGGL((gemm_kernel<4, 4, AKC, BKC, AG, BG, SCATTER>), grid, dim3(256), 0, s, p);
  else if (pl.mr == 4 && pl.nr == 2) hipLaunchKernelGGL((gemm_kernel<4, 2, AKC, BKC, AG, BG, SCATTER>), grid, dim3(256), 0, s, p);
  else hipLaunchKernelGGL((gemm_kernel<2, 2, AKC, BKC, AG, BG, SCATTER>), grid, dim3(256), 0, s, p);
}

}  // namespace gemm
}  // namespace seedhip

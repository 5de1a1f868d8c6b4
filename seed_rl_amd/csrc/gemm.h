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
#include "igemm.h"
#include "../../include/seedhip.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>

namespace seedhip {
namespace gemm {

typedef float f32x4_t __attribute__((ext_vector_type(4)));

// A k-contiguous operand whose rows are GATHERED instead of dense: the im2col row of a convolution (forward), the
// dY taps of a (super-)pixel (data gradient), the Keras kernel re-indexed by (parity class, ci) (data gradient B).
//   row x -> (u, v, w) by two divisions;  row base = const0 + u*s0 + v*s1 + w*s2
//   k -> tap = k >> cshift (C = 1 << cshift contiguous floats per tap: the channels), tap -> (ty, tx) = divmod(tap, tw)
//   element address = row base + ty*tsy + tx*tsx + (k & (C-1)); it reads as zero unless tap < ntaps and
//   (y0 + ty*ey, x0 + tx*ex) lies inside [0, vh) x [0, vw), with (y0, x0) = (ya*cy + oy0, xb*cx + ox0) and
//   (ya, xb) = (v, w) (or (u, v) when coord_uv): 'same' padding and map borders cost nothing but the predicate.
// A 16-byte vector never straddles taps (C is a power of two >= 4).
struct Gather {
  FastDiv d1, d2, d_tw;
  long long s0, const0; int s1, s2;
  int coord_uv, cshift, ntaps;
  int tsy, tsx;
  int cy, oy0, ey, cx, ox0, ex, vh, vw, all_valid;
};

struct Params {
  const float* A; long long lda; int a_relu;
  const float* B; long long ldb;
  int M, N, K;
  int k_per_slice;                          // split-K over blockIdx.z (multiple of BK)
  float* partial;                           // [slices][M][N] raw sums, or null: fused epilogue below
  float* partial_colsum;                    // [slices][N]: sum_k B(k, n) (bias gradient; OC B only), or null
  float* C; long long ldc;
  const float* bias; const float* residual; int out_relu;      // forward epilogue
  const float* mask; const float* add;                         // data-gradient epilogue (indexed like C)
  Gather ga, gb;                            // gathered operands (conv kernels below); unused by the Dense GEMMs
  int es, eih, eiw;                         // scatter epilogue (conv data gradient): stride, input map extents
};

constexpr int BK = 32, LD_KC = BK + 8;

// Staging of one operand tile [X rows/cols = BX][BK] through registers into LDS.
// OC rows are BX floats; with 2-wide fragments (ds_read_b64: lane groups {0-31}, {32-63} = two k rows 4 apart each)
// the row stride must move 4 rows by 32 banks: BX + 8.  4-wide fragments (b128, 16 lanes = 256 contiguous bytes per
// group) are conflict free at any stride.
template <int BX> struct LdOC { static constexpr int value = (BX == 64 || BX == 32) ? BX + 8 : BX; };

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

  __device__ void init(const float* p, long long ld, int x0, int X, int tid) {
#pragma unroll
    for (int i = 0; i < kVecs; ++i) {
      const int v = tid + i * 256;
      if (KC) {
        const int row = v >> 3, kc = (v & 7) * 4;
        krow[i] = kc; lds_off[i] = row * LD_KC + kc;
        base[i] = (x0 + row < X) ? p + (long long)(x0 + row) * ld + kc : nullptr;
      } else {
        const int kr = v / (BX / 4), x4 = (v % (BX / 4)) * 4;
        krow[i] = kr; lds_off[i] = kr * kLdOC + x4;
        base[i] = (x0 + x4 < X) ? p + (long long)kr * ld + x0 + x4 : nullptr;     // X % 4 == 0: all in or all out
      }
    }
    kstride = KC ? 1 : ld;
  }
  __device__ void load(int k, int k1, bool relu) {
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
        uint32_t u, rem, vv, ww;
        gg.d1.divmod((uint32_t)(xbase + row), u, rem);
        gg.d2.divmod(rem, vv, ww);
        b = p + gg.const0 + (long long)u * gg.s0 + (long long)vv * gg.s1 + (long long)ww * gg.s2;
        yy = (int)(gg.coord_uv ? u : vv) * gg.cy + gg.oy0;
        xx = (int)(gg.coord_uv ? vv : ww) * gg.cx + gg.ox0;
      }
      base[i] = b; y0[i] = yy; x0[i] = xx;
    }
  }
  __device__ void load(int k, int k1, bool relu) {
    const int kk = k + kc;                                  // same tap for all of this thread's vectors
    const int tap = kk >> g->cshift, kin = kk & ((1 << g->cshift) - 1);
    uint32_t ty, tx;
    g->d_tw.divmod((uint32_t)tap, ty, tx);
    const int toff = (int)ty * g->tsy + (int)tx * g->tsx + kin;
    const int dy = (int)ty * g->ey, dx = (int)tx * g->ex;
    const bool tap_ok = tap < g->ntaps;
#pragma unroll
    for (int i = 0; i < kVecs; ++i) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      const int y = y0[i] + dy, x = x0[i] + dx;
      if (base[i] && tap_ok && (g->all_valid || (y >= 0 && y < g->vh && x >= 0 && x < g->vw))) {
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
  int lds_off[kVecs], krow[kVecs];
  int toff, tdy, tdx;                        // this thread's tap: offset and border shift (same x4 for all its vectors)
  bool x_ok;
  const float* p0;
  const Gather* g;
  float4 r[kVecs];

  __device__ void init(const float* p, const Gather& gg, int /*nkt*/, int xbase, int X, int tid) {
    g = &gg; p0 = p + gg.const0;
    const int x4 = (tid % (BX / 4)) * 4, xg = xbase + x4;
    x_ok = xg < X;
    const int xc = x_ok ? xg : 0;
    const int tap = xc >> gg.cshift, kin = xc & ((1 << gg.cshift) - 1);
    uint32_t ty, tx;
    gg.d_tw.divmod((uint32_t)tap, ty, tx);
    toff = (int)ty * gg.tsy + (int)tx * gg.tsx + kin;
    tdy = (int)ty * gg.ey + gg.oy0; tdx = (int)tx * gg.ex + gg.ox0;
#pragma unroll
    for (int i = 0; i < kVecs; ++i) {
      const int v = tid + i * 256;
      krow[i] = v / (BX / 4);
      lds_off[i] = krow[i] * kLdOC + x4;
    }
  }
  __device__ void load(int k, int k1, bool relu) {
#pragma unroll
    for (int i = 0; i < kVecs; ++i) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      const int pix = k + krow[i];
      if (x_ok && pix < k1) {
        uint32_t u, rem, a, b;
        g->d1.divmod((uint32_t)pix, u, rem);
        g->d2.divmod(rem, a, b);
        const int y = (int)a * g->cy + tdy, x = (int)b * g->cx + tdx;
        if (g->all_valid || (y >= 0 && y < g->vh && x >= 0 && x < g->vw)) {
          v = *reinterpret_cast<const float4*>(p0 + (long long)u * g->s0 + (long long)a * g->s1 + (long long)b * g->s2 + toff);
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
  if constexpr (AG) sa.init(p.A, p.ga, nkt, m0, p.M, tid); else sa.init(p.A, p.lda, m0, p.M, tid);
  if constexpr (BG) sb.init(p.B, p.gb, nkt, n0, p.N, tid); else sb.init(p.B, p.ldb, n0, p.N, tid);
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

  if (nkt > 0) { sa.load(k0, k1, p.a_relu != 0); sb.load(k0, k1, false); }
  for (int kt = 0; kt < nkt; ++kt) {
    __syncthreads();                                       // previous tile consumed
    sa.store(As); sb.store(Bs);
    if (do_colsum) {
#pragma unroll
      for (int i = 0; i < SB::kVecs; ++i) { csum[i].x += sb.r[i].x; csum[i].y += sb.r[i].y; csum[i].z += sb.r[i].z; csum[i].w += sb.r[i].w; }
    }
    __syncthreads();
    if (kt + 1 < nkt) { sa.load(k0 + (kt + 1) * BK, k1, p.a_relu != 0); sb.load(k0 + (kt + 1) * BK, k1, false); }
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
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(AKC ? a_kc[i][kk] : a_oc[kk][i], BKC ? b_kc[j][kk] : b_oc[kk][j],
                                                             acc[i][j], 0, 0, 0);
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
        uint32_t img, rem, sa_, sb_;
        p.ga.d1.divmod((uint32_t)m, img, rem);
        p.ga.d2.divmod(rem, sa_, sb_);
#pragma unroll
        for (int j = 0; j < NR; ++j) {
          const int n = n0 + wn * NR * 16 + 16 * j + lx;
          if (n >= p.N) continue;
          uint32_t py, rem2, px, ci;
          p.gb.d1.divmod((uint32_t)n, py, rem2);
          p.gb.d2.divmod(rem2, px, ci);
          const int oy = (int)sa_ * p.es + (int)py, ox = (int)sb_ * p.es + (int)px;
          if (oy >= p.eih || ox >= p.eiw) continue;
          const long long at = (((long long)img * p.eih + oy) * p.eiw + ox) * p.ldc + ci;
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
  static const int force = getenv("SEEDHIP_GEMM_TILE") ? atoi(getenv("SEEDHIP_GEMM_TILE")) : 0;
  static const int force_s = getenv("SEEDHIP_GEMM_SLICES") ? atoi(getenv("SEEDHIP_GEMM_SLICES")) : 0;
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
  if (getenv("SEEDHIP_GEMM_DEBUG")) {
    static long long last = -1;
    const long long key = ((long long)M << 40) ^ ((long long)N << 20) ^ K;
    if (key != last) {
      last = key;
      fprintf(stderr, "[gemm] M=%d N=%d K=%d -> tile %dx%d slices %d (model %.1f us)\n", M, N, K,
              (4 / best.wn) * best.mr * 16, best.wn * best.nr * 16, best.slices, best.model_us);
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

// ---- convolutions as gather-GEMMs (see struct Gather) ------------------------------------------------------ //
inline int log2_exact(int v) { int l = 0; while ((1 << l) < v) ++l; return (1 << l) == v ? l : -1; }

// Forward: m = output pixel, k = (ky, kx, ci), n = co, B = the Keras kernel as stored (OC).  Any stride and padding;
// needs cin a power of two >= 4, ld_in % 4 == 0, cout % 4 == 0.
inline bool conv_fwd_setup(Params& p, const seedhip_conv_geom* g) {
  const int cs = log2_exact(g->cin);
  if (cs < 2 || g->ld_in % 4 || g->cout % 4 || g->ld_out % 4) return false;
  memset(&p, 0, sizeof(p));
  p.M = g->n_img * g->oh * g->ow; p.N = g->cout; p.K = g->kh * g->kw * g->cin; p.k_per_slice = (p.K + BK - 1) / BK * BK;
  p.ldb = g->cout; p.ldc = g->ld_out;
  Gather& a = p.ga;
  a.d1.init(g->oh * g->ow); a.d2.init(g->ow); a.d_tw.init(g->kw);
  a.s0 = (long long)g->ih * g->iw * g->ld_in; a.s1 = g->stride * g->iw * g->ld_in; a.s2 = g->stride * g->ld_in;
  a.const0 = -((long long)g->pad_t * g->iw + g->pad_l) * g->ld_in;
  a.cshift = cs; a.ntaps = g->kh * g->kw; a.tsy = g->iw * g->ld_in; a.tsx = g->ld_in;
  a.cy = g->stride; a.oy0 = -g->pad_t; a.ey = 1; a.cx = g->stride; a.ox0 = -g->pad_l; a.ex = 1; a.vh = g->ih; a.vw = g->iw;
  a.all_valid = g->pad_t == 0 && g->pad_l == 0 && (g->oh - 1) * g->stride + g->kh <= g->ih &&
                (g->ow - 1) * g->stride + g->kw <= g->iw;
  return true;
}

// Data gradient: m = super-pixel (a, b) covering input pixels (s*a+py, s*b+px), n = (py, px, ci), k = (jy, jx, co):
//   dX[s*a+py, s*b+px, ci] = sum dY[a-jy, b-jx, co] * W[py+s*jy, px+s*jx, ci, co]                  (pad 0)
// one GEMM for all stride-parity classes; for stride 1 with padding the same with dY[y+pad-jy, x+pad-jx].
// A rows = dY taps (zero outside the map), B rows = the kernel's co-contiguous rows re-indexed by (py, px, ci).
// Needs kh % s == kw % s == 0, cout a power of two >= 4, and pad 0 unless s == 1.
inline bool conv_dgrad_setup(Params& p, const seedhip_conv_geom* g) {
  const int s = g->stride, cs = log2_exact(g->cout);
  if (cs < 2 || g->kh % s || g->kw % s || g->ld_out % 4 || ((g->pad_t || g->pad_l) && s != 1)) return false;
  const int jh = g->kh / s, jw = g->kw / s;
  memset(&p, 0, sizeof(p));
  const int gh = (g->ih + s - 1) / s, gw = (g->iw + s - 1) / s;
  p.M = g->n_img * gh * gw; p.N = s * s * g->cin; p.K = jh * jw * g->cout; p.k_per_slice = (p.K + BK - 1) / BK * BK;
  p.ldc = g->ld_in; p.es = s; p.eih = g->ih; p.eiw = g->iw;
  Gather& a = p.ga;
  a.d1.init(gh * gw); a.d2.init(gw); a.d_tw.init(jw);
  a.s0 = (long long)g->oh * g->ow * g->ld_out; a.s1 = g->ow * g->ld_out; a.s2 = g->ld_out;
  a.const0 = ((long long)g->pad_t * g->ow + g->pad_l) * g->ld_out;
  a.cshift = cs; a.ntaps = jh * jw; a.tsy = -g->ow * g->ld_out; a.tsx = -g->ld_out;
  a.cy = 1; a.oy0 = g->pad_t; a.ey = -1; a.cx = 1; a.ox0 = g->pad_l; a.ex = -1; a.vh = g->oh; a.vw = g->ow;
  Gather& b = p.gb;
  b.d1.init(s * g->cin); b.d2.init(g->cin); b.d_tw.init(jw);
  b.s0 = (long long)g->kw * g->cin * g->cout; b.s1 = g->cin * g->cout; b.s2 = g->cout;
  b.cshift = cs; b.ntaps = jh * jw; b.tsy = s * g->kw * g->cin * g->cout; b.tsx = s * g->cin * g->cout;
  b.all_valid = 1;
  return true;
}

// Weight gradient: m = dW row (ky, kx, c), n = co, k = output pixel; A = the input gathered per tap (OC, above),
// B = dY [pixel, co] as stored.  Same geometry requirements as the forward.
inline bool conv_wgrad_setup(Params& p, const seedhip_conv_geom* g) {
  Params f;
  if (!conv_fwd_setup(f, g)) return false;
  memset(&p, 0, sizeof(p));
  p.ga = f.ga;
  p.M = g->kh * g->kw * g->cin; p.N = g->cout; p.K = g->n_img * g->oh * g->ow;
  p.ldb = g->ld_out;
  return true;
}

}  // namespace gemm
}  // namespace seedhip

// bf16x6 Dense GEMM, second structure (r4): 128 x 256 x 32 tiles on 8 waves, the SMALL operand pre-split.
//     C[m, n] = epilogue( sum_k A(m, k) * B(k, n) ),   A, B, C fp32 in HBM, same arithmetic as xgemm.h
//     (exact three-way bf16 split of both operands, the six products above 2^-25, fp32 accumulation in the matrix core).
//
// What xgemm.h's leave-one-out probes said (cfg2 forward 110 us): an empty loop 34 us, everything
// but the MFMAs 62, the MFMAs ADD 48 on top -- the workgroup's phases (load wait -> split + LDS writes -> barrier ->
// fragment reads -> MFMAs -> barrier) run one after the other, two workgroups of 200 registers per CU do not hide each
// other, and a SIMD cannot issue more than ~5 other instructions per 32-cycle bf16 MFMA: splitting BOTH operands of a
// 128 x 128 tile in the kernel is 3.7 VALU + 1 LDS access per MFMA before any address arithmetic.  So:
//   * the small operand of each Dense GEMM (the weights of the forward and the data gradient, dY of the weight
//     gradient; 2.6 / 11 MB at cfg2) is split ONCE per call by xsplit_kernel into the exact LDS image of its k-tile
//     slabs: [slab = (x-tile, k-tile)][plane hi/mid/lo][k / 8][x][8 k] bf16 -- staging it is a straight 16-byte copy
//     (no VALU), fragment reads are conflict free without a swizzle;
//   * the large operand (the activations, 111 MB) is read ONCE as fp32 -- the tile spans all 256 output columns --
//     and split on the way into LDS by waves 0-3 (4 float4 per thread and k-tile) while waves 4-7 copy the slab;
//     the data gradient has BOTH operands pre-split (dY, W) and no VALU work at all in its loop;
//   * two LDS stages of 72 KB; the two halves of the workgroup (one wave of each per SIMD) run HALF A PERIOD APART:
//     while waves 0-3 multiply k-tile j, waves 4-7 read their fragments, write their share of tile j + 1 and
//     re-request their registers for tile j + 2, then the roles swap (a barrier at each hand-over) -- a SIMD's matrix
//     pipe always has one wave's MFMAs while the other wave's LDS / VALU / VMEM work issues beside them (the first cut,
//     all eight waves in the same phase, added the two: 50 us of staging + 45 us of MFMAs = 91-100 us);
//     consecutive MFMAs go to different accumulators;
//   * one workgroup per CU (144 KB of LDS), split-K so that the grid is ~one round of the chip, partial sums through the
//     callers' deterministic second pass.
// Issue budget per SIMD and k-tile: 96 MFMAs (3072 cycles) against 48 fragment reads, 12 + 12 copies (waves 4-7) and
// 88 split VALU + 12 LDS writes + 4 loads (waves 0-3): ~2 other instructions per MFMA.
#pragma once
#include "xgemm.h"

namespace seedhip {
namespace xg8 {

using gemm::Params;
using gemm::kViewOOB;
using gemm::make_view;
using xg::bf16x8_t;
using xg::f32x4_t;
using xg::f32x16_t;
using xg::u32x4_t;

constexpr int BM = 128, BN = 256, BK = 32;
constexpr int kAPlane = BM * 64, kBPlane = BN * 64;              // bytes of one bf16 plane of a k-tile
constexpr int kAImg = 3 * kAPlane, kBImg = 3 * kBPlane;          // 24 KB, 48 KB
constexpr int kStage = kAImg + kBImg;                            // 72 KB
constexpr int kLds = 2 * kStage;                                 // 144 KB: one workgroup per CU
static_assert(kAPlane == xg::kPlane, "the in-kernel stagers of xgemm.h write 128-row planes");

// ---- pre-split: fp32 matrix -> k-tile slabs of bf16 planes ------------------------------------------------------ //
// Operand view: element (x, k), x < X, k < K, at src[x * ld + k] (kc) or src[k * ld + x] (!kc).  Slab (xt, kt) of XT
// rows x 32 k lives at dst + (xt * nkt + kt) * 3 * XT * 64 bytes as [plane][c = k / 8][x][8 k]; rows / k beyond the
// operand are zeros.  colsum (outer-contiguous operands only): colsum[kt * X + x] = sum over the slab's 32 k of the
// element -- the bias gradient's partial sums (one per k-tile; fixed order).
struct SplitArgs {
  const float* src; long long ld; int X, K, kc, relu;
  unsigned char* dst; int nkt;
  float* colsum;
};

__device__ __forceinline__ void split8(const float (&v)[8], u32x4_t& h, u32x4_t& m, u32x4_t& l) {
  unsigned hh[4], mm[4], ll[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) xg::split2(v[2 * j], v[2 * j + 1], hh[j], mm[j], ll[j]);
  h = u32x4_t{hh[0], hh[1], hh[2], hh[3]}; m = u32x4_t{mm[0], mm[1], mm[2], mm[3]}; l = u32x4_t{ll[0], ll[1], ll[2], ll[3]};
}

template <int XT>
__global__ void __launch_bounds__(XT)
xsplit_kernel(const SplitArgs a) {
  __shared__ float4 cs[4][XT / 4];
  const int kt = blockIdx.x, xt = blockIdx.y, t = threadIdx.x;
  unsigned char* slab = a.dst + ((size_t)xt * a.nkt + kt) * (size_t)(3 * XT * 64);
  const int k0 = kt * BK, x0 = xt * XT;
  if (a.kc) {
    // thread = row x: its 32 consecutive k (128 contiguous bytes: eight 16-byte loads), four chunks
    const int x = x0 + t;
    const float* row = a.src + (long long)x * a.ld + k0;
    float4 r[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {                              // K % 4 == 0: a quad is entirely in or out
      r[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (x < a.X && k0 + 4 * i < a.K) r[i] = *reinterpret_cast<const float4*>(row + 4 * i);
      if (a.relu) { r[i].x = fmaxf(r[i].x, 0.f); r[i].y = fmaxf(r[i].y, 0.f); r[i].z = fmaxf(r[i].z, 0.f); r[i].w = fmaxf(r[i].w, 0.f); }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float v[8] = {r[2 * c].x, r[2 * c].y, r[2 * c].z, r[2 * c].w, r[2 * c + 1].x, r[2 * c + 1].y, r[2 * c + 1].z, r[2 * c + 1].w};
      u32x4_t h, m, l;
      split8(v, h, m, l);
      unsigned char* d = slab + ((size_t)c * XT + t) * 16;
      *reinterpret_cast<u32x4_t*>(d) = h;
      *reinterpret_cast<u32x4_t*>(d + XT * 64) = m;
      *reinterpret_cast<u32x4_t*>(d + 2 * XT * 64) = l;
    }
    return;
  }
  // outer-contiguous: thread (c = t / (XT / 4), xq = t % (XT / 4)) loads rows k0 + 8 c + i of columns x0 + 4 xq .. + 3
  const int xq = t % (XT / 4), c = t / (XT / 4);
  const int x = x0 + 4 * xq;
  float4 r[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int k = k0 + 8 * c + i;
    r[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (k < a.K && x < a.X) {                                  // X % 4 == 0: a quad is entirely in or out
      r[i] = *reinterpret_cast<const float4*>(a.src + (long long)k * a.ld + x);
      if (a.relu) { r[i].x = fmaxf(r[i].x, 0.f); r[i].y = fmaxf(r[i].y, 0.f); r[i].z = fmaxf(r[i].z, 0.f); r[i].w = fmaxf(r[i].w, 0.f); }
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = e == 0 ? r[i].x : e == 1 ? r[i].y : e == 2 ? r[i].z : r[i].w;
    u32x4_t h, m, l;
    split8(v, h, m, l);
    unsigned char* d = slab + ((size_t)c * XT + 4 * xq + e) * 16;
    *reinterpret_cast<u32x4_t*>(d) = h;
    *reinterpret_cast<u32x4_t*>(d + XT * 64) = m;
    *reinterpret_cast<u32x4_t*>(d + 2 * XT * 64) = l;
  }
  if (a.colsum) {
    float4 s = r[0];
#pragma unroll
    for (int i = 1; i < 8; ++i) { s.x += r[i].x; s.y += r[i].y; s.z += r[i].z; s.w += r[i].w; }
    cs[c][xq] = s;
    __syncthreads();
    if (c == 0 && x < a.X) {
      float4 o = cs[0][xq];
#pragma unroll
      for (int cc = 1; cc < 4; ++cc) { const float4 p = cs[cc][xq]; o.x += p.x; o.y += p.y; o.z += p.z; o.w += p.w; }
      *reinterpret_cast<float4*>(a.colsum + (long long)kt * a.X + x) = o;
    }
  }
}

inline size_t slab_bytes(int X, int K, int XT) {
  return (size_t)((X + XT - 1) / XT) * ((K + BK - 1) / BK) * (size_t)(3 * XT * 64);
}

template <int XT>
inline void launch_split(const float* src, long long ld, int X, int K, bool kc, bool relu, void* dst, float* colsum,
                         hipStream_t s) {
  SplitArgs a;
  a.src = src; a.ld = ld; a.X = X; a.K = K; a.kc = kc; a.relu = relu; a.dst = (unsigned char*)dst;
  a.nkt = (K + BK - 1) / BK; a.colsum = colsum;
  hipLaunchKernelGGL(xsplit_kernel<XT>, dim3(a.nkt, (X + XT - 1) / XT), dim3(XT), 0, s, a);
}

// ---- the GEMM ----------------------------------------------------------------------------------------------------- //
struct Args {
  Params p;
  const unsigned char* Ap;      // pre-split A slabs (AMODE 2), XT = 128
  const unsigned char* Bp;      // pre-split B slabs, XT = 256
  const float* colsum_kt;       // [nkt][N] per-k-tile column sums of B (xsplit), or null; summed per slice by m-tile 0
  int mt, nt, slices, kt_per_slice, nkt;
  unsigned long long* trace;    // EXP & 128 builds: s_memtime stamps of workgroup 0 (tools/trace_x8.py), [group][k-tile][8]
  int slice_major;              // item order (slice, m-tile, n-tile) instead of (m-tile, slice, n-tile)
  int prio;                     // s_setprio of a wave's MEM segment (its COMP segment runs at 0)
};

// ---- in-kernel stagers of the fp32 operand A for 256 threads (group A), four 16-byte vectors each ------------------ //
// k-contiguous: thread (row = u >> 3, q = u & 7) loads k = 4 q .. 4 q + 3 of rows row + 32 i; LDS layout = xgemm.h's
// kc_chunk (rows 32 apart share the swizzle).
struct StageKC4 {
  unsigned voff, step, wofs;
  int nvalid, kq4;
  __device__ void init(int u, long long ld, int x0, int X) {
    const int q = u & 7, row = u >> 3;
    voff = (unsigned)(((long long)(x0 + row) * ld + 4 * q) * 4);
    int nv = (X - x0 - row + 31) >> 5;
    nvalid = nv < 0 ? 0 : (nv > 4 ? 4 : nv);
    step = (unsigned)(ld * 128);                             // 32 rows
    kq4 = 4 * q;
    wofs = (unsigned)(xg::kc_chunk(row, q >> 1) + (q & 1) * 8);
  }
  __device__ void load(f32x4_t (&r)[4], const __amdgpu_buffer_rsrc_t& rs, int k, int k1) {
    const bool kin = k + kq4 < k1;                           // K % 4 == 0: a vector is entirely in or out
    const unsigned kb = (unsigned)k * 4u;
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = xg::view_load_s(rs, (kin && i < nvalid) ? voff : kViewOOB, kb + (unsigned)i * step);
  }
  template <int EXP>
  __device__ void store(f32x4_t (&r)[4], unsigned char* lds, bool relu) const {
    if (relu) {
#pragma unroll
      for (int i = 0; i < 4; ++i) xg::relu4(r[i]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      unsigned h0, m0, l0, h1, m1, l1;
      if (EXP & 4) {
        h0 = m0 = l0 = __builtin_amdgcn_perm(__float_as_uint(r[i][1]), __float_as_uint(r[i][0]), 0x07060302u);
        h1 = m1 = l1 = __builtin_amdgcn_perm(__float_as_uint(r[i][3]), __float_as_uint(r[i][2]), 0x07060302u);
      } else {
        xg::split2s(r[i][0], r[i][1], h0, m0, l0);
        xg::split2s(r[i][2], r[i][3], h1, m1, l1);
      }
      unsigned char* d = lds + wofs + i * (32 * 64);
      *reinterpret_cast<xg::u32x2_t*>(d) = xg::u32x2_t{h0, h1};
      *reinterpret_cast<xg::u32x2_t*>(d + kAPlane) = xg::u32x2_t{m0, m1};
      *reinterpret_cast<xg::u32x2_t*>(d + 2 * kAPlane) = xg::u32x2_t{l0, l1};
    }
  }
};
// outer-contiguous: thread (c = u >> 6, kh = (u >> 5) & 1, xq = u & 31) loads x = 4 xq .. 4 xq + 3 of rows
// k = 8 c + 4 kh + i and writes, per plane and x, the 8-byte half (4 k) of chunk (x, c); LDS layout = xgemm.h's oc_chunk.
struct StageOC4 {
  unsigned voff, ld4;
  int kr, xq, c, kh;
  bool xin;
  __device__ void init(int u, long long ld, int x0, int X) {
    xq = u & 31; kh = (u >> 5) & 1; c = u >> 6;
    kr = 8 * c + 4 * kh;
    xin = x0 + 4 * xq < X;                                   // X % 4 == 0
    voff = (unsigned)(((long long)kr * ld + x0 + 4 * xq) * 4);
    ld4 = (unsigned)(ld * 4);
  }
  __device__ void load(f32x4_t (&r)[4], const __amdgpu_buffer_rsrc_t& rs, int k, int k1) {
    const unsigned kb = (unsigned)k * ld4;
    const int nrow = k1 - k - kr;
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = xg::view_load_s(rs, (xin && i < nrow) ? voff : kViewOOB, kb + (unsigned)i * ld4);
  }
  template <int EXP>
  __device__ void store(f32x4_t (&r)[4], unsigned char* lds, bool relu) const {
    if (relu) {
#pragma unroll
      for (int i = 0; i < 4; ++i) xg::relu4(r[i]);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      unsigned h0, m0, l0, h1, m1, l1;
      if (EXP & 4) {
        h0 = m0 = l0 = __builtin_amdgcn_perm(__float_as_uint(r[1][e]), __float_as_uint(r[0][e]), 0x07060302u);
        h1 = m1 = l1 = __builtin_amdgcn_perm(__float_as_uint(r[3][e]), __float_as_uint(r[2][e]), 0x07060302u);
      } else {
        xg::split2s(r[0][e], r[1][e], h0, m0, l0);
        xg::split2s(r[2][e], r[3][e], h1, m1, l1);
      }
      unsigned char* d = lds + xg::oc_chunk(4 * xq + e, c) + kh * 8;
      *reinterpret_cast<xg::u32x2_t*>(d) = xg::u32x2_t{h0, h1};
      *reinterpret_cast<xg::u32x2_t*>(d + kAPlane) = xg::u32x2_t{m0, m1};
      *reinterpret_cast<xg::u32x2_t*>(d + 2 * kAPlane) = xg::u32x2_t{l0, l1};
    }
  }
};
// ---- the same for ALL 512 threads, two vectors each (DMA builds: group 1's B slab comes by LDS-DMA, so both groups
// share the split of operand A: r4b trace, 2174 cycles of MEM for waves 0-3 against 1600 of COMP) ------------------- //
struct StageKC2 {
  unsigned voff, step, wofs;
  int nvalid, kq4;
  __device__ void init(int u, long long ld, int x0, int X) {
    const int q = u & 7, row = u >> 3;                       // row 0..63; rows row + 64 i
    voff = (unsigned)(((long long)(x0 + row) * ld + 4 * q) * 4);
    int nv = (X - x0 - row + 63) >> 6;
    nvalid = nv < 0 ? 0 : (nv > 2 ? 2 : nv);
    step = (unsigned)(ld * 256);
    kq4 = 4 * q;
    wofs = (unsigned)(xg::kc_chunk(row, q >> 1) + (q & 1) * 8);
  }
  __device__ void load(f32x4_t (&r)[2], const __amdgpu_buffer_rsrc_t& rs, int k, int k1) {
    const bool kin = k + kq4 < k1;
    const unsigned kb = (unsigned)k * 4u;
#pragma unroll
    for (int i = 0; i < 2; ++i) r[i] = xg::view_load_s(rs, (kin && i < nvalid) ? voff : kViewOOB, kb + (unsigned)i * step);
  }
  __device__ void store(f32x4_t (&r)[2], unsigned char* lds, bool relu) const {
    if (relu) { xg::relu4(r[0]); xg::relu4(r[1]); }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      unsigned h0, m0, l0, h1, m1, l1;
      xg::split2(r[i][0], r[i][1], h0, m0, l0);
      xg::split2(r[i][2], r[i][3], h1, m1, l1);
      unsigned char* d = lds + wofs + i * (64 * 64);
      *reinterpret_cast<xg::u32x2_t*>(d) = xg::u32x2_t{h0, h1};
      *reinterpret_cast<xg::u32x2_t*>(d + kAPlane) = xg::u32x2_t{m0, m1};
      *reinterpret_cast<xg::u32x2_t*>(d + 2 * kAPlane) = xg::u32x2_t{l0, l1};
    }
  }
};
// outer-contiguous: thread (c = u >> 7, kp = (u >> 5) & 3, xq = u & 31) loads x = 4 xq .. 4 xq + 3 of rows k = 8 c + 2 kp
// and + 1 and writes, per plane and x, the 4-byte word (k, k + 1) of chunk (x, c)
struct StageOC2 {
  unsigned voff, ld4;
  int kr, xq, c, kp;
  bool xin;
  __device__ void init(int u, long long ld, int x0, int X) {
    xq = u & 31; kp = (u >> 5) & 3; c = u >> 7;
    kr = 8 * c + 2 * kp;
    xin = x0 + 4 * xq < X;
    voff = (unsigned)(((long long)kr * ld + x0 + 4 * xq) * 4);
    ld4 = (unsigned)(ld * 4);
  }
  __device__ void load(f32x4_t (&r)[2], const __amdgpu_buffer_rsrc_t& rs, int k, int k1) {
    const unsigned kb = (unsigned)k * ld4;
    const int nrow = k1 - k - kr;
#pragma unroll
    for (int i = 0; i < 2; ++i) r[i] = xg::view_load_s(rs, (xin && i < nrow) ? voff : kViewOOB, kb + (unsigned)i * ld4);
  }
  __device__ void store(f32x4_t (&r)[2], unsigned char* lds, bool relu) const {
    if (relu) { xg::relu4(r[0]); xg::relu4(r[1]); }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      unsigned h, m, l;
      xg::split2(r[0][e], r[1][e], h, m, l);
      unsigned char* d = lds + xg::oc_chunk(4 * xq + e, c) + kp * 4;
      *reinterpret_cast<unsigned*>(d) = h;
      *reinterpret_cast<unsigned*>(d + kAPlane) = m;
      *reinterpret_cast<unsigned*>(d + 2 * kAPlane) = l;
    }
  }
};
template <int AMODE> struct PickStage2 { typedef StageKC2 type; };
template <> struct PickStage2<1> { typedef StageOC2 type; };
template <int AMODE> struct PickStage4 { typedef StageKC4 type; };
template <> struct PickStage4<1> { typedef StageOC4 type; };

// AMODE: 0 A fp32, element (m, k) at A[m * lda + k]; 1 A fp32 at A[k * lda + m]; 2 A pre-split slabs.
// GROUP: 0 = waves 0-3, which bring operand A (split it in the kernel, or copy its slab), 1 = waves 4-7, which copy the
// B slab.  A SIMD holds one wave of each group and the groups run HALF A PERIOD APART -- while one multiplies k-tile
// j (48 MFMAs, 1536 cycles of its SIMD's matrix pipe) the other reads its fragments of the next tile, writes its share
// of the tile after that into the other LDS stage and re-requests its registers -- with a workgroup barrier at each
// hand-over (two per k-tile).  Hazards: tile j + 1 is written in MEM(j) (group 0 between barriers 2j and 2j + 1, group 1
// between 2j + 1 and 2j + 2) into the stage whose previous tile j - 1 both groups finished READING before barrier 2j;
// it is complete at barrier 2j + 2, before either group's MEM(j + 1).
// EXP (probes, garbage results): 4 no split arithmetic, 8 no MFMAs, 32 no fragment reads
// DMA (r4b): group 1 brings its B slab by LDS-DMA (`buffer_load_dwordx4 ... lds`, 1 KB per wave instruction: no VGPRs, no
// ds_write) -- the slab IS the LDS image.  Tile j + 1 is requested right behind barrier 2j, at the HEAD of the group's
// COMP(j - 1) segment (the stage's previous tile j - 1 was last read before that barrier), flies under the whole period
// and is waited for (vmcnt(0)) at the end of the group's MEM(j), before barrier 2j + 2.  The hand-over barriers are raw
// `s_barrier`s behind `s_waitcnt lgkmcnt(0)`: `__syncthreads()` would drain the DMA at EVERY barrier (it waits vmcnt(0)).
template <int AMODE, int EXP, int GROUP, bool DMA>
__device__ __forceinline__ void xg8_body(const Args& g, unsigned char* smem, const int wave) {
  const Params& p = g.p;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wm = wave >> 2, wn = wave & 3, lx = lane & 31, half = lane >> 5;

  // work item: XCD b % 8 runs a contiguous range of (m-tile, slice, n-tile) items, n-tile fastest
  const int item = xg::work_item(blockIdx.x, g.mt * g.nt * g.slices);
  // slice-major: the workgroups of one XCD then share ONE k-range of the pre-split operand (cfg2 forward: 1.3 MB of
  // the 4 MB of W planes per XCD instead of all of them beside the streaming activations in a 4 MB L2)
  const int ntile = item % g.nt, rest = item / g.nt;
  const int slice = g.slice_major ? rest / g.mt : rest % g.slices, mtile = g.slice_major ? rest % g.mt : rest / g.slices;
  const int m0 = mtile * BM, n0 = ntile * BN;
  const int kt0 = slice * g.kt_per_slice;
  int nkt = g.nkt - kt0; if (nkt > g.kt_per_slice) nkt = g.kt_per_slice;
  const int k_end = (kt0 + nkt) * BK < p.K ? (kt0 + nkt) * BK : p.K;

  // ---- staging ---- //
  constexpr bool kSplitA = AMODE != 2;
  constexpr bool kSplitter = kSplitA && GROUP == 0;
  constexpr bool kDmaB = DMA && GROUP == 1;
  constexpr bool kShare = DMA && kSplitA;               // DMA builds: all 512 threads split operand A, two vectors each
  constexpr int kCopies = GROUP == 1 ? (kDmaB ? 1 : 12) : 6;   // 16-byte chunks of a slab per thread and k-tile (registers)
  typename PickStage2<AMODE>::type sa2;
  f32x4_t ra2[2];
  typename PickStage4<AMODE>::type sa;
  f32x4_t ra[4];
  u32x4_t rc[kCopies];
  const int u = tid & 255;
  const __amdgpu_buffer_rsrc_t va = make_view(p.A, kSplitter ? ((AMODE == 0 ? (long long)p.M : (long long)p.K) * p.lda * 4) : 0);
  const __amdgpu_buffer_rsrc_t vs = GROUP == 1 ? make_view((const float*)g.Bp, (long long)((size_t)g.nt * g.nkt * kBImg))
                                               : make_view((const float*)g.Ap, kSplitA ? 0 : (long long)((size_t)g.mt * g.nkt * kAImg));
  if constexpr (kSplitter && !kShare) sa.init(u, p.lda, m0, p.M);
  const __amdgpu_buffer_rsrc_t va2 = make_view(p.A, kShare ? ((AMODE == 0 ? (long long)p.M : (long long)p.K) * p.lda * 4) : 0);
  if constexpr (kShare) sa2.init(tid, p.lda, m0, p.M);
  const unsigned slab0 = GROUP == 1 ? (unsigned)((ntile * g.nkt + kt0) * kBImg) : (unsigned)((mtile * g.nkt + kt0) * kAImg);
  constexpr unsigned kSlab = GROUP == 1 ? kBImg : kAImg;
  constexpr int kImgOff = GROUP == 1 ? kAImg : 0;        // where this group's image starts inside a stage

  auto dma_tile = [&](int j, unsigned char* stage) {     // DMA: k-tile j of the B slab -> stage (asynchronous)
    const int jj = j < nkt ? j : nkt - 1;
    const unsigned so = slab0 + (unsigned)jj * kSlab;
    typedef __attribute__((address_space(3))) void lds_void_t;
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      const int piece = (wave - 4) + 4 * i;              // 48 pieces of 1 KB, twelve per wave
      __builtin_amdgcn_raw_ptr_buffer_load_lds(vs, (lds_void_t*)(stage + kAImg + piece * 1024), 16, 16u * (unsigned)lane,
                                               __builtin_amdgcn_readfirstlane(so + (unsigned)piece * 1024u), 0, 0);
    }
  };
  auto load_tile = [&](int j) {                          // k-tile j of this slice -> registers (clamped past the end)
    const int jj = j < nkt ? j : nkt - 1;
    if constexpr (kShare) { sa2.load(ra2, va2, (kt0 + jj) * BK, k_end); }
    else if constexpr (kDmaB) { (void)jj; }
    else if constexpr (kSplitter) { sa.load(ra, va, (kt0 + jj) * BK, k_end); }
    else {
      const unsigned so = slab0 + (unsigned)jj * kSlab;
#pragma unroll
      for (int i = 0; i < kCopies; ++i)
        rc[i] = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(vs, (unsigned)u * 16u, __builtin_amdgcn_readfirstlane(so + (unsigned)i * 4096u), 0));
    }
  };
  auto store_tile = [&](unsigned char* stage) {          // registers -> LDS stage (the image IS the slab: chunk i at 16 i)
    if constexpr (kShare) { sa2.store(ra2, stage, p.a_relu != 0); }
    else if constexpr (kDmaB) { (void)stage; }
    else if constexpr (kSplitter) { sa.template store<EXP>(ra, stage, p.a_relu != 0); }
    else {
#pragma unroll
      for (int i = 0; i < kCopies; ++i) *reinterpret_cast<u32x4_t*>(stage + kImgOff + (u + 256 * i) * 16) = rc[i];
    }
  };

  // ---- fragments: all of a k-tile (two 16-deep steps) ---- //
  // A: xgemm.h's layouts when split in the kernel, [plane][c][x][8 k] when pre-split; B always the latter.
  int a_off[2], b_off[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    a_off[s] = kSplitA ? xg::frag_off<AMODE == 0>(wm * 64 + lx, half, s) : ((2 * s + half) * BM + wm * 64 + lx) * 16;
    b_off[s] = kAImg + ((2 * s + half) * BN + wn * 64 + lx) * 16;
  }
  constexpr int a_tile = kSplitA ? xg::tile_stride<AMODE == 0>() : 32 * 16;
  bf16x8_t fa[2][2][3], fb[2][2][3];                     // [k-step][tile][plane]
  auto read_frags = [&](const unsigned char* stage) {
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          if (EXP & 32) {
            fa[s][t][q] = __builtin_bit_cast(bf16x8_t, u32x4_t{(unsigned)a_off[s], (unsigned)t, (unsigned)q, 1u});
            fb[s][t][q] = __builtin_bit_cast(bf16x8_t, u32x4_t{(unsigned)b_off[s], (unsigned)t, (unsigned)q, 2u});
            continue;
          }
          fa[s][t][q] = *reinterpret_cast<const bf16x8_t*>(stage + a_off[s] + t * a_tile + q * kAPlane);
          fb[s][t][q] = *reinterpret_cast<const bf16x8_t*>(stage + b_off[s] + t * (32 * 16) + q * kBPlane);
        }
  };

  f32x16_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  // six products per accumulator, smallest first; consecutive MFMAs on DIFFERENT accumulators (a dependent 32 x 32 x 16
  // waits for its predecessor's last pass).  MFMA rows = n (operand B), columns = m (operand A).
  auto comp = [&]() {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      if (EXP & 8) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int q = 0; q < 3; ++q) asm volatile("" :: "v"(fa[s][t][q]), "v"(fb[s][t][q]));
        continue;
      }
#define XG8_PRODUCT(QB, QA) \
      _Pragma("unroll") for (int i = 0; i < 2; ++i) \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) \
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[s][j][QB], fa[s][i][QA], acc[i][j], 0, 0, 0);
      XG8_PRODUCT(2, 0)   // bl ah
      XG8_PRODUCT(0, 2)   // bh al
      XG8_PRODUCT(1, 1)   // bm am
      XG8_PRODUCT(1, 0)   // bm ah
      XG8_PRODUCT(0, 1)   // bh am
      XG8_PRODUCT(0, 0)   // bh ah
#undef XG8_PRODUCT
    }
  };
  unsigned char* st0 = smem;
  unsigned char* st1 = smem + kStage;
  const int prio = g.prio;
  auto mem = [&](int j) {                                // fragments of tile j; tile j + 1 to LDS; tile j + 2 requested
    if (prio == 1) __builtin_amdgcn_s_setprio(1); else if (prio == 2) __builtin_amdgcn_s_setprio(2); else if (prio == 3) __builtin_amdgcn_s_setprio(3);
    read_frags((j & 1) ? st1 : st0);                     // issued first: their latency passes under the staging
    store_tile((j & 1) ? st0 : st1);                     // (a clamped copy of the last tile behind the end)
    load_tile(j + 2);
    __builtin_amdgcn_sched_barrier(0);
    if (prio) __builtin_amdgcn_s_setprio(0);
  };
  auto hand_over = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (DMA) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // NOT vmcnt: a DMA may be in flight
    else __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
  };

  const bool tr = (EXP & 128) && g.trace != nullptr && blockIdx.x == 0 && (wave & 3) == 0 && lane == 0;
  unsigned long long* tb = g.trace + GROUP * 64 * 8;
  auto stamp = [&](int j, int k) { if ((EXP & 128) && tr && j < 64) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); tb[j * 8 + k] = __builtin_amdgcn_s_memtime(); } };
  // publish: the group's DMA pieces have landed (its own newer register loads -- kShare: two -- may stay in flight)
  auto publish = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (kShare) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  if constexpr (kDmaB) dma_tile(0, st0);
  if constexpr (!kDmaB || kShare) { load_tile(0); store_tile(st0); load_tile(1); }
  if constexpr (DMA) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); else __syncthreads();
  if constexpr (GROUP == 0) {
    for (int j = 0; j < nkt; ++j) { stamp(j, 0); mem(j); stamp(j, 1); hand_over(); stamp(j, 2); comp(); stamp(j, 3); hand_over(); stamp(j, 4); }
  } else if constexpr (kDmaB) {
    // MEM of this group: fragments, its share of operand A (kShare), then the wait for the DMA pieces requested a period ago
    auto mem1 = [&](int j) {
      read_frags((j & 1) ? st1 : st0);
      if constexpr (kShare) { store_tile((j & 1) ? st0 : st1); load_tile(j + 2); }
    };
    dma_tile(1, st1);                                    // behind barrier 0: tile 1 flies under group 0's MEM(0)
    hand_over();                                         // barrier 1
    mem1(0);
    publish();                                           // barrier 2: tile 1 complete
    for (int j = 1; j < nkt; ++j) {
      stamp(j, 0);
      dma_tile(j + 1, (j & 1) ? st0 : st1);              // behind barrier 2j: stage (j + 1) & 1, tile j - 1 is read out
      comp();                                            // COMP(j - 1)
      stamp(j, 1);
      hand_over();                                       // barrier 2j + 1
      stamp(j, 2);
      mem1(j);                                           // MEM(j)
      stamp(j, 3);
      publish();                                         // barrier 2j + 2: tile j + 1 complete
      stamp(j, 4);
    }
    comp();
  } else {
    hand_over(); mem(0); hand_over();
    for (int j = 1; j < nkt; ++j) { stamp(j, 0); comp(); stamp(j, 1); hand_over(); stamp(j, 2); mem(j); stamp(j, 3); hand_over(); stamp(j, 4); }
    comp();
  }

  // bias gradient: this slice's column sums from the per-k-tile sums xsplit left (m-tile 0 only; fixed order)
  if (g.colsum_kt && p.partial_colsum && mtile == 0 && tid < BN / 4 && n0 + 4 * tid < p.N) {
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int j = 0; j < nkt; ++j) {
      const float4 o = *reinterpret_cast<const float4*>(g.colsum_kt + (long long)(kt0 + j) * p.N + n0 + 4 * tid);
      s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w;
    }
    *reinterpret_cast<float4*>(p.partial_colsum + (long long)slice * p.N + n0 + 4 * tid) = s;
  }

  // epilogue.  Accumulator (i, j) register e of lane (lx, half): m = m0 + wm*64 + 32 i + lx,
  //   n = n0 + wn*64 + 32 j + 8 (e >> 2) + 4 half + (e & 3): quads of four consecutive n.
  const int nb = n0 + wn * 64 + 4 * half;                // quad (j, q) starts at nb + 32 j + 8 q
  const bool has_bias = p.bias != nullptr, has_mask = p.mask != nullptr, has_res = p.residual != nullptr, has_add = p.add != nullptr;
  f32x4_t bv[8];
#pragma unroll
  for (int jq = 0; jq < 8; ++jq) bv[jq] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  if (has_bias && !p.partial) {
#pragma unroll
    for (int jq = 0; jq < 8; ++jq) {
      const int n = nb + 32 * (jq >> 2) + 8 * (jq & 3);
      if (n < p.N) bv[jq] = *reinterpret_cast<const f32x4_t*>(p.bias + n);
    }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = m0 + wm * 64 + 32 * i + lx;
    const bool m_ok = m < p.M;
    if (p.partial) {
#pragma unroll
      for (int jq = 0; jq < 8; ++jq) {
        const int j = jq >> 2, q = jq & 3, n = nb + 32 * j + 8 * q;
        if (m_ok && n < p.N) {
          const f32x4_t v = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
          *reinterpret_cast<f32x4_t*>(p.partial + ((long long)slice * p.M + m) * p.N + n) = v;
        }
      }
      continue;
    }
    const long long row = (long long)m * p.ldc;
    f32x4_t ex[8], ad[8];
    if (has_mask || has_res) {                           // (mask and residual never come together: backward / forward)
      const float* src = has_mask ? p.mask : p.residual;
#pragma unroll
      for (int jq = 0; jq < 8; ++jq) {
        const int n = nb + 32 * (jq >> 2) + 8 * (jq & 3);
        ex[jq] = (m_ok && n < p.N) ? *reinterpret_cast<const f32x4_t*>(src + row + n) : f32x4_t{0.f, 0.f, 0.f, 0.f};
      }
    }
    if (has_add) {
#pragma unroll
      for (int jq = 0; jq < 8; ++jq) {
        const int n = nb + 32 * (jq >> 2) + 8 * (jq & 3);
        ad[jq] = (m_ok && n < p.N) ? *reinterpret_cast<const f32x4_t*>(p.add + row + n) : f32x4_t{0.f, 0.f, 0.f, 0.f};
      }
    }
#pragma unroll
    for (int jq = 0; jq < 8; ++jq) {
      const int j = jq >> 2, q = jq & 3, n = nb + 32 * j + 8 * q;
      f32x4_t v = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
      v += bv[jq];
      if (has_res) v += ex[jq];
      if (p.out_relu) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
      }
      if (has_mask) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = ex[jq][e] > 0.f ? v[e] : 0.f;
      }
      if (has_add) v += ad[jq];
      if (m_ok && n < p.N) *reinterpret_cast<f32x4_t*>(p.C + row + n) = v;
    }
  }
}

template <int AMODE, int EXP = 0, bool DMA = false>
__global__ void __launch_bounds__(512, 2)
xg8_kernel(const Args g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  if (wave < 4) xg8_body<AMODE, EXP, 0, DMA>(g, smem, wave);
  else xg8_body<AMODE, EXP, 1, DMA>(g, smem, wave);
}

// ---- host side ------------------------------------------------------------------------------------------------ //
struct Plan { bool ok; int mt, nt, nkt, slices, kt_per_slice; size_t a_planes, b_planes, colsum; };

// mode bits (SEEDHIP_X8, default 5): 1 Dense forward, 2 Dense data gradient, 4 Dense weight gradient.  The data gradient
// stays on xgemm.h by default: its K is the layer's output width (256 at cfg2: 8 k-tiles per workgroup) and its
// epilogue moves 2 x 111 MB, which ONE workgroup per CU cannot hide under another's MFMAs (measured 142-156 us against
// 126-135; r4).
inline int mode() { static const int m = xg::env_int("SEEDHIP_X8", 5); return m; }

// Served: N >= 192 (a 256-column tile at least three quarters full), 16-byte rows, operands and slabs below 2 GB.
// a_pre: A pre-split as well (the data gradient).  want_colsum: room for xsplit's per-k-tile column sums.
inline Plan plan(int M, int N, int K, long long a_bytes, bool a_pre, bool want_colsum, bool must_split = false) {
  Plan pl{false, 0, 0, 0, 1, 0, 0, 0, 0};
  if (M < 256 || N < 192 || K < 64 || (K & 3) || (N & 3) || (M & 3)) return pl;
  pl.mt = (M + BM - 1) / BM; pl.nt = (N + BN - 1) / BN; pl.nkt = (K + BK - 1) / BK;
  pl.a_planes = a_pre ? (size_t)pl.mt * pl.nkt * kAImg : 0;
  pl.b_planes = (size_t)pl.nt * pl.nkt * kBImg;
  pl.colsum = want_colsum ? (size_t)pl.nkt * N * sizeof(float) : 0;
  if (a_bytes >= (1LL << 31) - 64 || pl.a_planes >= (1ULL << 31) - 64 || pl.b_planes >= (1ULL << 31) - 64) return pl;
  const long long tiles = (long long)pl.mt * pl.nt;
  if (tiles > (1 << 20)) return pl;
  // slices: one workgroup per CU; cost ~ rounds(tiles * s) * (k-tiles per slice + fixed) + the partial sums' round trip
  int s = 1;
  constexpr int force = 0;
  if (force > 0) s = force;
  else {
    const long long slots = xg::cu_count();
    const double kFixed = 5.0, us_per_ktile = 1.5, bytes_per_us = 4.0e6;
    int smax = pl.nkt / 8; if (smax > 64) smax = 64; if (smax < 1) smax = 1;
    double best = -1.0;
    for (int c = 1; c <= smax; ++c) {
      const int per = (pl.nkt + c - 1) / c;
      const long long rounds = (tiles * c + slots - 1) / slots;
      double cost = (double)rounds * (per + kFixed) * us_per_ktile;
      if (c > 1) cost += (double)(c + 1) * M * N * 4.0 / bytes_per_us;
      if (best < 0.0 || cost < best) { best = cost; s = c; }
    }
  }
  if (must_split && s < 2 && pl.nkt >= 2) s = 2;
  const int per = (pl.nkt + s - 1) / s;
  pl.slices = (pl.nkt + per - 1) / per; pl.kt_per_slice = per;
  pl.ok = true;
  return pl;
}
inline size_t partial_bytes(int M, int N, const Plan& pl) { return pl.slices > 1 ? (size_t)pl.slices * M * N * sizeof(float) : 0; }
inline size_t al256(size_t b) { return (b + 255) & ~(size_t)255; }
// workspace beyond the caller's partial-sum area: [A planes][B planes][colsum]
inline size_t planes_bytes(const Plan& pl) { return al256(pl.a_planes) + al256(pl.b_planes) + al256(pl.colsum); }

template <int AMODE>
inline bool launch(const Params& p, const Plan& pl, const void* Ap, const void* Bp, const float* colsum_kt, hipStream_t s) {
  Args g;
  g.p = p; g.Ap = (const unsigned char*)Ap; g.Bp = (const unsigned char*)Bp; g.colsum_kt = colsum_kt;
  g.mt = pl.mt; g.nt = pl.nt; g.slices = pl.slices; g.kt_per_slice = pl.kt_per_slice; g.nkt = pl.nkt;
  g.slice_major = 1; g.prio = 0; g.trace = nullptr;           // (r4's order / priority / timing-experiment knobs measured +-3 %: gone)
  const int blocks = pl.mt * pl.nt * pl.slices;
  static const bool ok = hipFuncSetAttribute((const void*)xg8_kernel<AMODE, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds) == hipSuccess;
  if (!ok) return false;
  hipLaunchKernelGGL((xg8_kernel<AMODE, 0>), dim3(blocks), dim3(512), kLds, s, g);
  return true;
}

}  // namespace xg8
}  // namespace seedhip

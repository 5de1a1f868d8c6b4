// fp32 GEMM on the BF16 matrix pipe through the exact three-way split ("bf16x6"), for the large Dense layers
// (atari/networks.py:176-251 Dense(256) / dmlab/networks.py:105-124,152-171 Dense + LSTM input projections):
//     C[m, n] = epilogue( sum_k A(m, k) * B(k, n) ),        A, B, C fp32 in HBM.
//
// Why: v_mfma_f32_16x16x4_f32 runs at the fp32 VECTOR rate (64 FLOP/clk/SIMD, 157 TF/s) and shares the SIMD's issue
// with every VALU instruction of the kernel; v_mfma_f32_32x32x16_bf16 is a separate pipe at 16x that rate.  Any fp32
// number is the exact sum of three bf16 numbers, v = h + m + l with h = bf16_rn(v), m = bf16_rn(v - h),
// l = v - h - m (8 + 8 + 8 significant bits; both subtractions are exact in fp32; tests/test_bf16_split.py), so
//     a * b = (ah + am + al) * (bh + bm + bl)
// and the six products  ah bh, ah bm, am bh, ah bl, al bh, am bm  carry everything above 2^-25 |a b|: what is dropped
// (am bl + al bm + al bl) is bounded by 2^-26 |a b| per term pair -- a quarter of ONE fp32 rounding of the accumulator,
// with random sign (round-to-nearest splits).  Every product of two bf16 numbers is exact in fp32 and the matrix core
// accumulates in fp32: the kernel evaluates the same real-number sum as the fp32 MFMA up to that term and fp32
// summation order, at 6/16 of its matrix-pipe time (ceiling 2500 / 6 = 417 algorithmic TF/s against 157).
// tests/test_gpu_kernels.py::test_x6_gemm_fp32_accuracy holds it to the error of torch's own fp32 matmul against an
// fp64 evaluation at K = 2592 / 3136.
//
// Structure (256 threads = 2 x 2 waves, 128 x 128 x 32 tiles, one LDS buffer + register prefetch like gemm.h):
//   * operands come from HBM as fp32, 16 bytes per lane and load, through buffer views (hardware zero fill for ragged
//     tiles; ONE byte-offset register per thread, the k-tile / row advance is a uniform soffset);
//     waves 0-1 stage operand A, waves 2-3 operand B: eight float4 per thread and k-tile;
//   * the split happens ONCE per staged element, on the way from the load registers into LDS (v_cvt_pk_bf16_f32:
//     11 VALU per element pair -- VALU work that runs beside the bf16 matrix pipe, unlike beside the fp32 one);
//   * LDS holds three bf16 planes per operand with the reduction index contiguous in runs of 8 (one ds_read_b128 = one
//     MFMA operand): k-contiguous operands as [x][32 k] rows of 64 bytes with the 16-byte chunk index XOR-swizzled by
//     (x >> 2) & 3, outer-contiguous operands as [k / 8][x][8 k] -- the thread that loaded rows k..k+7 of four x
//     packs them into four 16-byte chunks, no transposition pass; both are conflict free for the fragment reads;
//   * the MFMA's row operand (i) is GEMM-B's n, its column operand (j) GEMM-A's m: a lane ends up with four
//     consecutive n of one row m per accumulator quad -> 16-byte epilogue loads / stores.
// Split-K over the grid with the deterministic second pass of gemm.h's callers; the workgroup -> tile map gives each
// XCD a contiguous run of tiles (neighbouring tiles share an operand panel in that XCD's L2).
#pragma once
#include "gemm.h"

namespace seedhip {
namespace xg {

using gemm::Params;
using gemm::kViewOOB;
using gemm::make_view;

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));

constexpr int BK = 32;                       // fp32 reduction elements per k-tile = two 16-deep MFMA steps
constexpr int BX = 128;                      // rows of A / columns of B per workgroup tile
constexpr int kPlane = BX * 64;              // bytes of one bf16 plane of one operand tile
constexpr int kOperand = 3 * kPlane;
constexpr int kLds = 2 * kOperand;           // 48 KB: three workgroups per CU

// (a, b) -> packed bf16 pairs of the three parts (element a in the low half)
__device__ __forceinline__ void split2(float a, float b, unsigned& h, unsigned& m, unsigned& l) {
  const f32x2_t v = {a, b};
  h = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
  const f32x2_t r1 = {a - __uint_as_float(h << 16), b - __uint_as_float(h & 0xFFFF0000u)};
  m = __builtin_bit_cast(unsigned, __builtin_convertvector(r1, bf16x2_t));
  const f32x2_t r2 = {r1[0] - __uint_as_float(m << 16), r1[1] - __uint_as_float(m & 0xFFFF0000u)};
  l = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, bf16x2_t));
}

// The h part of the in-kernel splits below.  SEEDHIP_SPLIT_H_RNE = 0: truncation (one v_and); 1: round to nearest even
// with integer operations (bits + 0x7FFF + ((bits >> 16) & 1), mask: v_bfe + v_add3 + v_and) -- m and l stay exact
// truncations of an exactly representable remainder, but the remainder then has either sign, and the dropped products
// am bl + al bm + al bl lose the common sign that truncating h gives them (r6 A/B: profiles/r06_split_rne_ab.txt).
#ifndef SEEDHIP_SPLIT_H_RNE
#define SEEDHIP_SPLIT_H_RNE 0
#endif
__device__ __forceinline__ unsigned hi_part(unsigned bits) {
  return SEEDHIP_SPLIT_H_RNE ? ((bits + 0x7FFFu + ((bits >> 16) & 1u)) & 0xFFFF0000u) : (bits & 0xFFFF0000u);
}
__device__ __forceinline__ u32x2_t hi_part(u32x2_t bits) {
  return SEEDHIP_SPLIT_H_RNE ? ((bits + 0x7FFFu + ((bits >> 16) & 1u)) & 0xFFFF0000u) : (bits & 0xFFFF0000u);
}

// the same split by TRUNCATION (plain integer / fp32 VALU operations only: v_cvt_pk_bf16_f32 issues at a quarter of
// their rate): h = top 8 significant bits, m = top 8 of the rest, l = the remaining <= 8 bits; h + m + l == v exactly,
// every part has the sign of v, so the dropped products am bl + al bm (<= 2^-23 |a b|) share the sign of a b
__device__ __forceinline__ void split2_trunc(float a, float b, unsigned& h, unsigned& m, unsigned& l) {
  const unsigned ha = hi_part(__float_as_uint(a)), hb = hi_part(__float_as_uint(b));
  const float ra = a - __uint_as_float(ha), rb = b - __uint_as_float(hb);
  const unsigned ma = __float_as_uint(ra) & 0xFFFF0000u, mb = __float_as_uint(rb) & 0xFFFF0000u;
  const float la = ra - __uint_as_float(ma), lb = rb - __uint_as_float(mb);
  h = __builtin_amdgcn_perm(hb, ha, 0x07060302u);
  m = __builtin_amdgcn_perm(mb, ma, 0x07060302u);
  l = __builtin_amdgcn_perm(__float_as_uint(lb), __float_as_uint(la), 0x07060302u);
}
// The split the IN-KERNEL stagers of xgemm.h / xgemm8.h use for the operand they split on the way into LDS: by truncation
// since r5 (plain full-rate VALU; the three v_cvt_pk_bf16_f32 of the round-to-nearest split issue at a quarter of that
// rate and were most of the staging half's period: cfg2 Dense forward 97 -> 90 us, weight gradient 108 -> 97, data
// gradient 117 -> 107, step 1.009 -> 0.987 ms, same-box A/B of two builds through SEEDHIP_LIB).  The operand that is
// split ONCE per call (xsplit_kernel) keeps round-to-nearest, as do the weights of wfx.h / wdx.h / wsx.h / wsy.h: one
// truncated and one rounded operand leave the dropped products am bl + al bm without a common sign.
__device__ __forceinline__ void split2s(float a, float b, unsigned& h, unsigned& m, unsigned& l) { split2_trunc(a, b, h, m, l); }
template <int EXP>
__device__ __forceinline__ void split2x(float a, float b, unsigned& h, unsigned& m, unsigned& l) {
  if (EXP & 256) split2_trunc(a, b, h, m, l); else split2s(a, b, h, m, l);
}

__device__ __forceinline__ f32x4_t view_load_s(const __amdgpu_buffer_rsrc_t& r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(r, voff, __builtin_amdgcn_readfirstlane(soff), 0));
}
// The same load hidden from hipcc's s_waitcnt bookkeeping (xgemm_ws_kernel's stagers): with a multi-slot register
// ring filled across loop iterations the compiler falls back to draining the whole queue before every use, i.e. it
// waits for the YOUNGEST request.  Here the loads are opaque asm statements and the ring slot is claimed by
// wait_slot<N>: s_waitcnt vmcnt(N) naming every register of the slot read-write, so that no consumer is scheduled
// above it (guide 5.7 form (ii)).  The leading s_nop covers a soffset fresh from a VALU-written SGPR.
typedef unsigned sgpr128_t __attribute__((ext_vector_type(4)));
// (raw buffer descriptor words: base[31:0], base[47:32] (stride 0), num_records, flags as gemm::make_view)
__device__ __forceinline__ sgpr128_t make_view_words(const float* p, long long bytes) {
  const uint64_t ab = reinterpret_cast<uint64_t>(p);
  return sgpr128_t{(unsigned)__builtin_amdgcn_readfirstlane((unsigned)ab),
                   (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(ab >> 32)) & 0xFFFFu,
                   (unsigned)__builtin_amdgcn_readfirstlane((int)bytes), 0x00020000u};
}
__device__ __forceinline__ f32x4_t asm_load(const sgpr128_t& d, unsigned voff, unsigned soff) {
  f32x4_t v;
  asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(v) : "v"(voff), "s"(d), "s"(__builtin_amdgcn_readfirstlane(soff)));
  return v;
}
// An item that an earlier `asm volatile("buffer_load_dwordx4 ...")` request may still be writing.  The compiler knows
// nothing about the request: to it the destination registers hold a value from the asm statement on, and it is free to
// COPY them (a tied "+v" operand of a later `s_waitcnt` statement allocated elsewhere, a live-range split) before the
// data has arrived -- cgx.h's first build read the first unit's items from such copies.  take_item waits (at most N
// younger vector-memory operations outstanding; they retire in order) and moves the registers out INSIDE one asm
// statement whose inputs are plain "v" operands: the only reads of the in-flight registers sit behind the wait.
// tests/test_isa_structure.py checks the compiled kernels for reads in front of it.
template <int N>
__device__ __forceinline__ f32x4_t take_item(const f32x4_t& r) {
  const f32x2_t a = __builtin_shufflevector(r, r, 0, 1), b = __builtin_shufflevector(r, r, 2, 3);
  f32x2_t lo, hi;
  asm volatile("s_waitcnt vmcnt(%4)\n\tv_mov_b64 %0, %2\n\tv_mov_b64 %1, %3" : "=&v"(lo), "=&v"(hi) : "v"(a), "v"(b), "n"(N));
  return f32x4_t{lo[0], lo[1], hi[0], hi[1]};
}
// The same without the wait, for kernels that wait ONCE for a whole set of items (`asm volatile("s_waitcnt vmcnt(0)")`, no
// operands) and consume them one by one behind it: volatile asm statements keep their order, and an input-only operand
// gives the compiler no reason to copy the in-flight registers (a tied "+v" operand on the wait did: wsy.h copied the
// next round's items into the loop's registers in the loop pre-header, in front of the wait).
__device__ __forceinline__ f32x4_t move_item(const f32x4_t& r) {
  const f32x2_t a = __builtin_shufflevector(r, r, 0, 1), b = __builtin_shufflevector(r, r, 2, 3);
  f32x2_t lo, hi;
  asm volatile("v_mov_b64 %0, %2\n\tv_mov_b64 %1, %3" : "=&v"(lo), "=&v"(hi) : "v"(a), "v"(b));
  return f32x4_t{lo[0], lo[1], hi[0], hi[1]};
}
template <int N>
__device__ __forceinline__ unsigned take_word(unsigned r) {
  unsigned o;
  asm volatile("s_waitcnt vmcnt(%2)\n\tv_mov_b32 %0, %1" : "=&v"(o) : "v"(r), "n"(N));
  return o;
}
__device__ __forceinline__ f32x4_t any_load(const sgpr128_t& d, unsigned voff, unsigned soff) { return asm_load(d, voff, soff); }
__device__ __forceinline__ f32x4_t any_load(const __amdgpu_buffer_rsrc_t& r, unsigned voff, unsigned soff) { return view_load_s(r, voff, soff); }
template <int N>
__device__ __forceinline__ void wait_slot(f32x4_t (&r)[8]) {
  asm volatile("s_waitcnt vmcnt(%8)"
               : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "n"(N));
}

__device__ __forceinline__ void relu4(f32x4_t& v) {
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
}

// LDS byte offset (inside one plane) of the 16-byte chunk holding k = 8 c .. 8 c + 7 of row / column x
SH_HD int kc_chunk(int x, int c) { return x * 64 + ((c ^ ((x >> 2) & 3)) << 4); }
SH_HD int oc_slot(int x) { return 4 * (x >> 2) + ((x & 3) ^ ((x >> 3) & 3)); }
SH_HD int oc_chunk(int x, int c) { return (c * BX + oc_slot(x)) << 4; }

// ---- staging of one operand tile [BX][32] by 128 threads (u = 0..127), eight float4 each ------------------------ //
// k-contiguous: element (x, k) at base[x * ld + k].  Thread (row = u >> 3, q = u & 7) loads k = 4 q .. 4 q + 3 of rows
// row + 16 i (a wave instruction = 8 rows x 128 contiguous bytes) and writes 8 bytes per plane and vector.
struct StageKC {
  unsigned voff, step, wofs;
  int nvalid, kq4;
  __device__ void init(int u, long long ld, int x0, int X) {
    const int q = u & 7, row = u >> 3;
    voff = (unsigned)(((long long)(x0 + row) * ld + 4 * q) * 4);
    int nv = (X - x0 - row + 15) >> 4;
    nvalid = nv < 0 ? 0 : (nv > 8 ? 8 : nv);
    step = (unsigned)(ld * 64);                              // 16 rows
    kq4 = 4 * q;
    wofs = (unsigned)(kc_chunk(row, q >> 1) + (q & 1) * 8);  // rows 16 apart share the swizzle
  }
  template <typename RS>
  __device__ void load(f32x4_t (&r)[8], const RS& rs, int k, int k1) {
    const bool kin = k + kq4 < k1;                           // K % 4 == 0: a vector is entirely in or out
    const unsigned kb = (unsigned)k * 4u;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const unsigned vo = (kin && i < nvalid) ? voff : kViewOOB;
      r[i] = any_load(rs, vo, kb + (unsigned)i * step);
    }
  }
  // (the ReLU of an operand is applied here, not behind the loads: a use of the load registers right where they are
  // requested would wait for them and kill the prefetch)
  template <int EXP = 0>
  __device__ void store(f32x4_t (&r)[8], unsigned char* lds, bool relu) const {
    if (relu) {
#pragma unroll
      for (int i = 0; i < 8; ++i) relu4(r[i]);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      unsigned h0, m0, l0, h1, m1, l1;
      if (EXP & 4) {
        h0 = m0 = l0 = __builtin_amdgcn_perm(__float_as_uint(r[i][1]), __float_as_uint(r[i][0]), 0x07060302u);
        h1 = m1 = l1 = __builtin_amdgcn_perm(__float_as_uint(r[i][3]), __float_as_uint(r[i][2]), 0x07060302u);
      } else {
        split2x<EXP>(r[i][0], r[i][1], h0, m0, l0);
        split2x<EXP>(r[i][2], r[i][3], h1, m1, l1);
      }
      if (EXP & 16) { asm volatile("" :: "v"(h0), "v"(m0), "v"(l0), "v"(h1), "v"(m1), "v"(l1)); continue; }
      unsigned char* d = lds + wofs + i * (16 * 64);
      *reinterpret_cast<u32x2_t*>(d) = u32x2_t{h0, h1};
      *reinterpret_cast<u32x2_t*>(d + kPlane) = u32x2_t{m0, m1};
      *reinterpret_cast<u32x2_t*>(d + 2 * kPlane) = u32x2_t{l0, l1};
    }
  }
  __device__ float4 colsum(const f32x4_t (&)[8]) const { return make_float4(0.f, 0.f, 0.f, 0.f); }
};
// outer-contiguous: element (x, k) at base[k * ld + x].  Thread (c = u >> 5, xg = u & 31) loads x = 4 xg .. 4 xg + 3
// of rows k = 8 c + i (a wave instruction = 2 rows x 512 contiguous bytes) and writes, per plane, the four 16-byte
// chunks (x, k = 8 c .. 8 c + 7): element pairs (k, k + 1) come from two load registers, so the "transposition" is
// the choice of cvt_pk operands.
struct StageOC {
  unsigned voff, ld4;
  int c8, xg;
  bool xin;
  __device__ void init(int u, long long ld, int x0, int X) {
    xg = u & 31; c8 = (u >> 5) * 8;
    xin = x0 + 4 * xg < X;                                   // X % 4 == 0
    voff = (unsigned)(((long long)c8 * ld + x0 + 4 * xg) * 4);
    ld4 = (unsigned)(ld * 4);
  }
  template <typename RS>
  __device__ void load(f32x4_t (&r)[8], const RS& rs, int k, int k1) {
    const unsigned kb = (unsigned)k * ld4;
    const int nrow = k1 - k - c8;                            // rows of this thread inside the operand
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const unsigned vo = (xin && i < nrow) ? voff : kViewOOB;
      r[i] = any_load(rs, vo, kb + (unsigned)i * ld4);
    }
  }
  static __device__ __forceinline__ float comp(const f32x4_t& v, int e) { return v[e]; }
  template <int EXP = 0>
  __device__ void store(f32x4_t (&r)[8], unsigned char* lds, bool relu) const {
    if (relu) {
#pragma unroll
      for (int i = 0; i < 8; ++i) relu4(r[i]);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {                            // e, j are compile-time constants after unrolling
      unsigned h[4], m[4], l[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (EXP & 4) h[j] = m[j] = l[j] = __builtin_amdgcn_perm(__float_as_uint(comp(r[2 * j + 1], e)), __float_as_uint(comp(r[2 * j], e)), 0x07060302u);
        else split2x<EXP>(comp(r[2 * j], e), comp(r[2 * j + 1], e), h[j], m[j], l[j]);
      }
      if (EXP & 16) { asm volatile("" :: "v"(h[0]), "v"(h[1]), "v"(h[2]), "v"(h[3]), "v"(m[0]), "v"(m[1]), "v"(m[2]), "v"(m[3]), "v"(l[0]), "v"(l[1]), "v"(l[2]), "v"(l[3])); continue; }
      unsigned char* d = lds + oc_chunk(4 * xg + e, c8 >> 3);
      *reinterpret_cast<u32x4_t*>(d) = u32x4_t{h[0], h[1], h[2], h[3]};
      *reinterpret_cast<u32x4_t*>(d + kPlane) = u32x4_t{m[0], m[1], m[2], m[3]};
      *reinterpret_cast<u32x4_t*>(d + 2 * kPlane) = u32x4_t{l[0], l[1], l[2], l[3]};
    }
  }
  __device__ float4 colsum(const f32x4_t (&r)[8]) const {     // sum over this thread's 8 rows (bias gradient)
    f32x4_t t = r[0];
#pragma unroll
    for (int i = 1; i < 8; ++i) t += r[i];
    return make_float4(t[0], t[1], t[2], t[3]);
  }
};
template <bool KC> struct PickStage { typedef StageKC type; };
template <> struct PickStage<false> { typedef StageOC type; };

// fragment byte offset of lane (x, half) for 16-deep step s inside one plane (tile t adds 32 rows / columns)
template <bool KC> __device__ __forceinline__ int frag_off(int x, int half, int s) {
  return KC ? kc_chunk(x, 2 * s + half) : oc_chunk(x, 2 * s + half);
}
template <bool KC> constexpr int tile_stride() { return KC ? 32 * 64 : 32 * 16; }

struct Grid { int mt, nt, slices; unsigned long long* trace; };   // trace: s_memtime stamps of workgroup 0 (EXP & 128)

// workgroup b -> work item: XCD b % 8 runs a contiguous range of the (m-tile, slice, n-tile) items, n-tile fastest
__device__ __forceinline__ int work_item(int b, int total) {
  const int xcd = b & 7, idx = b >> 3, per = total >> 3, rem = total & 7;
  return xcd * per + (xcd < rem ? xcd : rem) + idx;
}

// EXP (leave-one-out probes, instantiated for the forward kernel only; results are garbage): 2 no global loads after
// the first k-tile, 4 no split arithmetic (the high halves go to all three planes), 8 no MFMAs, 16 no LDS writes,
// 32 no fragment reads
template <bool AKC, bool BKC, int EXP = 0>
__global__ void __launch_bounds__(256, 2)
xgemm_kernel(const Params p, const Grid g) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[kLds];
  unsigned char* As = smem;
  unsigned char* Bs = smem + kOperand;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, lx = lane & 31, half = lane >> 5;

  const int item = work_item(blockIdx.x, g.mt * g.nt * g.slices);
  const int ntile = item % g.nt, rest = item / g.nt, slice = rest % g.slices, mtile = rest / g.slices;
  const int m0 = mtile * BX, n0 = ntile * BX;
  const int k0 = slice * p.k_per_slice;
  int k1 = k0 + p.k_per_slice; if (k1 > p.K) k1 = p.K;
  const int nkt = (k1 - k0 + BK - 1) / BK;

  typedef typename PickStage<AKC>::type SA;
  typedef typename PickStage<BKC>::type SB;
  // one of the two stagers is live per wave (waves 0-1: A, waves 2-3: B); the load registers are shared
  SA sa; SB sb;
  f32x4_t r[8];
  const bool stage_a = __builtin_amdgcn_readfirstlane(tid) < 128;      // a scalar branch
  const int u = tid & 127;
  const __amdgpu_buffer_rsrc_t ra = make_view(p.A, (AKC ? (long long)p.M * p.lda : (long long)p.K * p.lda) * 4);
  const __amdgpu_buffer_rsrc_t rb = make_view(p.B, (BKC ? (long long)p.N * p.ldb : (long long)p.K * p.ldb) * 4);
  if (stage_a) sa.init(u, p.lda, m0, p.M); else sb.init(u, p.ldb, n0, p.N);
  const bool do_colsum = !BKC && p.partial_colsum && mtile == 0;
  float4 csum = make_float4(0.f, 0.f, 0.f, 0.f);

  f32x16_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  int a_off[2], b_off[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    a_off[s] = frag_off<AKC>(wm * 64 + lx, half, s);
    b_off[s] = frag_off<BKC>(wn * 64 + lx, half, s);
  }

  if (nkt > 0) {
    if (stage_a) sa.load(r, ra, k0, k1); else sb.load(r, rb, k0, k1);
  }
  for (int kt = 0; kt < nkt; ++kt) {
    __syncthreads();                                         // previous tile consumed
    if (stage_a) sa.template store<EXP>(r, As, p.a_relu != 0);
    else {
      sb.template store<EXP>(r, Bs, false);
      if (do_colsum) { const float4 t = sb.colsum(r); csum.x += t.x; csum.y += t.y; csum.z += t.z; csum.w += t.w; }
    }
    __syncthreads();
    if (kt + 1 < nkt && !(EXP & 2)) {
      const int k = k0 + (kt + 1) * BK;
      if (stage_a) sa.load(r, ra, k, k1); else sb.load(r, rb, k, k1);
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8_t fa[2][3], fb[2][3];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          if (EXP & 32) {
            fa[t][q] = __builtin_bit_cast(bf16x8_t, u32x4_t{(unsigned)a_off[s], (unsigned)kt, (unsigned)t, (unsigned)q});
            fb[t][q] = __builtin_bit_cast(bf16x8_t, u32x4_t{(unsigned)b_off[s], (unsigned)kt, (unsigned)t, (unsigned)q});
            continue;
          }
          fa[t][q] = *reinterpret_cast<const bf16x8_t*>(As + a_off[s] + t * tile_stride<AKC>() + q * kPlane);
          fb[t][q] = *reinterpret_cast<const bf16x8_t*>(Bs + b_off[s] + t * tile_stride<BKC>() + q * kPlane);
        }
      if (EXP & 8) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int q = 0; q < 3; ++q) asm volatile("" :: "v"(fa[t][q]), "v"(fb[t][q]));
        continue;
      }
      // six products per (m-tile, n-tile), smallest first; MFMA rows = n (operand B), columns = m (operand A)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          f32x16_t c = acc[i][j];
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[j][2], fa[i][0], c, 0, 0, 0);   // bl ah
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[j][0], fa[i][2], c, 0, 0, 0);   // bh al
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[j][1], fa[i][1], c, 0, 0, 0);   // bm am
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[j][1], fa[i][0], c, 0, 0, 0);   // bm ah
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[j][0], fa[i][1], c, 0, 0, 0);   // bh am
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[j][0], fa[i][0], c, 0, 0, 0);   // bh ah
          acc[i][j] = c;
        }
    }
  }

  // bias-gradient column sums of this slice: the four row-chunk threads of a column quad are summed in fixed order
  if (do_colsum) {
    __syncthreads();
    float4* scr = reinterpret_cast<float4*>(smem);
    if (!stage_a) scr[u] = csum;                             // u = c * 32 + xg
    __syncthreads();
    if (tid < 32 && n0 + 4 * tid < p.N) {
      float4 t = scr[tid];
#pragma unroll
      for (int c = 1; c < 4; ++c) { const float4 o = scr[c * 32 + tid]; t.x += o.x; t.y += o.y; t.z += o.z; t.w += o.w; }
      *reinterpret_cast<float4*>(p.partial_colsum + (long long)slice * p.N + n0 + 4 * tid) = t;
    }
  }

  // epilogue.  Accumulator (i, j) register r of lane (lx, half): m = m0 + wm*64 + 32 i + lx,
  //   n = n0 + wn*64 + 32 j + 8 (r >> 2) + 4 half + (r & 3): quads of four consecutive n
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = m0 + wm * 64 + 32 * i + lx;
    if (m >= p.M) continue;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n = n0 + wn * 64 + 32 * j + 8 * q + 4 * half;
        if (n >= p.N) continue;                              // N % 4 == 0
        f32x4_t v = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
        if (p.partial) {
          *reinterpret_cast<f32x4_t*>(p.partial + ((long long)slice * p.M + m) * p.N + n) = v;
          continue;
        }
        const long long at = (long long)m * p.ldc + n;
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
        if (p.mask_bits) {                                   // (ldc % 4 == 0: `at` is a multiple of four)
          const unsigned mb = p.mask_bits[at >> 2];
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = ((mb >> r) & 1u) ? v[r] : 0.f;
        }
        if (p.add) v += *reinterpret_cast<const f32x4_t*>(p.add + at);
        *reinterpret_cast<f32x4_t*>(p.C + at) = v;
      }
    }
  }
}

// ---- host side ---------------------------------------------------------------------------------------------- //
struct Plan { bool ok; int slices, k_per_slice; Grid grid; };

inline int env_int(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
// mode bits (SEEDHIP_X6, default 7): 1 Dense forward, 2 Dense data gradient, 4 Dense weight gradient
inline int mode() { static const int m = env_int("SEEDHIP_X6", 7); return m; }

inline int cu_count() {
  static const int n = [] { int dev = 0, v = 0; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev); return v > 0 ? v : 256; }();
  return n;
}

// Served: operands of 16-byte aligned rows (ld % 4 == 0, K % 4 == 0, N % 4 == 0) below 2 GB each, and enough work
// to fill the chip with 128 x 128 tiles; small GEMMs (recurrent steps, inference batches, the heads) stay on gemm.h.
inline Plan plan(int M, int N, int K, long long a_bytes, long long b_bytes, bool must_split = false) {
  Plan pl{false, 1, 0, {0, 0, 0, nullptr}};
  if (M < 256 || N < 128 || K < 64 || (K & 3) || (N & 3)) return pl;
  if (a_bytes >= (1LL << 31) - 64 || b_bytes >= (1LL << 31) - 64) return pl;
  const int mt = (M + BX - 1) / BX, nt = (N + BX - 1) / BX;
  const long long tiles = (long long)mt * nt;
  if (tiles > (1 << 20)) return pl;
  // slices: the kernel holds 200 registers per lane = two workgroups per CU, so the chip runs 2 x CUs workgroups at a
  // time and a grid of 1.3 such rounds costs two (r4: the forward's 672 workgroups on 512 slots took two rounds of
  // 21 k-tiles where 504 take one of 27).  Cost of s slices ~ rounds(tiles * s) * (k-tiles per slice + kFixed), the
  // fixed part being a workgroup's prologue and epilogue, plus the partial sums' round trip; at least 8 k-tiles each.
  const int nkt = (K + BK - 1) / BK;
  int s = 1;
  constexpr int force = 0;
  if (force > 0) s = force;
  else {
    const long long slots = 2LL * cu_count();
    const double kFixed = 6.0, us_per_ktile = 2.0, bytes_per_us = 4.0e6;   // measured on the cfg2 / cfg5 Dense shapes
    int smax = nkt / 8; if (smax > 64) smax = 64; if (smax < 1) smax = 1;
    double best = -1.0;
    for (int c = 1; c <= smax; ++c) {
      const int per = (nkt + c - 1) / c;
      const long long rounds = (tiles * c + slots - 1) / slots;
      double cost = (double)rounds * (per + kFixed) * us_per_ktile;
      if (c > 1) cost += (double)(c + 1) * M * N * 4.0 / bytes_per_us;     // the partial sums' round trip
      if (best < 0.0 || cost < best) { best = cost; s = c; }
    }
  }
  if (must_split && s < 2 && nkt >= 2) s = 2;
  int per = (nkt + s - 1) / s;
  s = (nkt + per - 1) / per;
  pl.ok = true; pl.slices = s; pl.k_per_slice = per * BK;
  pl.grid = Grid{mt, nt, s, nullptr};
  return pl;
}
inline size_t partial_bytes(int M, int N, const Plan& pl) { return pl.slices > 1 ? (size_t)pl.slices * M * N * sizeof(float) : 0; }

template <bool AKC, bool BKC>
inline void launch(Params p, const Plan& pl, hipStream_t s) {
  p.k_per_slice = pl.k_per_slice;
  const int blocks = pl.grid.mt * pl.grid.nt * pl.grid.slices;
  // (r4's persistent wave-specialised variant -- 142 vs 110 us forward -- and the leave-one-out timing builds are gone)
  hipLaunchKernelGGL((xgemm_kernel<AKC, BKC, 0>), dim3(blocks), dim3(256), 0, s, p, pl.grid);
}

}  // namespace xg
}  // namespace seedhip

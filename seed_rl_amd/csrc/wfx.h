// The second Atari conv's forward -- Conv2D(32, 4, 2) on the 20 x 20 x 16 output of the first
// (/root/reference/atari/networks.py:236) -- on the BF16 matrix pipe through the exact three-way operand split of
// xgemm.h ("bf16x6": fp32 = h + m + l, six of the nine plane products; same arithmetic, same error bound).
//
// Why: wfw.h's fp32-MFMA kernel needs 140 us in the step for 11.2 GFLOP (80 TF/s of the fp32 pipe's 157; its 128 MFMAs
// per round already run back to back), and v_mfma_f32_32x32x16_bf16 does the six products of one (32 pixel x 32 channel
// x 16 k) block in 192 cycles against 1024 for the fp32 instruction's 128 passes -- 27 us of matrix time for the layer.
//
// Structure: one 8-wave workgroup per CU (157.7 KB of LDS), a persistent run of images per workgroup.
//   * a ROUND is 128 consecutive pixels of the run = four tiles of 32 (tiles cross image boundaries).  Two waves share a
//     tile and split the REDUCTION: wave (tile, kh) multiplies the eight taps of kernel rows 2 kh, 2 kh + 1 -- 48 MFMAs
//     per round -- and the pair lands on one SIMD, so one wave's split / address / store work overlaps the other's
//     multiplications.  (One wave per SIMD with all 16 taps needed 412 registers and serialised everything: 91 us.)
//   * WEIGHTS: lane (co = lane & 31, kq = lane >> 5) holds W[tap][ci = 8 kq .. 8 kq + 7][co] of its eight taps as three
//     bf16 planes -- 96 registers, split once in the prologue (round to nearest); they are the MFMA's row operand, so a
//     lane ends up with four consecutive output channels of one pixel per accumulator quad;
//   * INPUT: the run is a contiguous array of image rows (20 per image, 1 280 bytes each).  Every row is loaded ONCE
//     (32-byte items, coalesced), split in registers (by truncation: full-rate VALU) and written as three bf16 planes
//     into a ring of 72 rows; rows of one parity sit `kPitch` = 41 sixteen-byte slots apart, 41 = 9 (mod 16): a step
//     to the next output row (two input rows down, nine pixels on) continues the slot sequence of the previous one, and
//     the 16 lanes of a ds_read_b128 phase ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}: the pixel -> lane map follows
//     these groups) take 16 consecutive pixels: 16 different slots except across an image boundary (average conflict
//     factor 1.2, against 2.0 for a plain [row][ix][ci] layout);
//   * per round and wave: pixel operands two taps ahead (three register buffers), one row-item request per tap in
//     taps 0-5 (for round r + 2), one half item split and written per tap in taps 2-7 (for round r + 1, from the
//     registers requested a round ago: asm loads with counted waits -- the compiler's own bookkeeping would drain the
//     queue at the loop head, see xgemm.h), pieces pinned between the MFMAs with sched_barrier;
//   * OUTPUT: the second half's waves hand their partial sums over through a 4 KB block per tile; the first half's add
//     theirs (bias in the accumulator's start), activate, and write the sums back into the block as [pixel][32 channel]
//     rows with the 16-byte chunk index XOR-swizzled by the pixel -- read back at the end of the round they leave as
//     1 KB of consecutive addresses per store instruction.  Two barriers per round (block free / block + rows published).
// Ring safety: rounds r and r + 1 together span at most 35 rows of one parity (enumerated in
// tests/test_wfx_layout.py); kRU = 36.
//
// Measured (8 442 images, MI355X): 74 us against wfw.h's 121.  Where the rest goes (SEEDHIP_WFX_EXP builds, results
// wrong): without the split 72, without split and loads 59, without MFMAs 58, nothing but operand reads + outputs 37:
// the pieces ADD rather than overlap -- VALU work does not hide beside the streaming bf16 MFMAs of the SIMD's other
// wave (the same finding as xgemm8.h), and the in-kernel split is 5.5 VALU instructions per input element.
#pragma once
#include "common.h"
#include <vector>
#include "xgemm.h"
#include "../../include/seedhip.h"

namespace seedhip {
namespace wfx {

using xg::bf16x8_t;
using xg::f32x4_t;
using xg::f32x16_t;
using xg::u32x4_t;
using xg::sgpr128_t;

constexpr int kIH = 20, kIW = 20, kOW = 9, kP = 81;          // the one geometry this kernel is built for
constexpr int kRowBytes = kIW * 16 * 4;                      // 1 280 bytes of fp32 per image row
constexpr int kRU = 36;                                      // ring rows per parity
constexpr int kPitch = 656;                                  // bytes of one row in one plane: 4 sub-rows x 160 + 16
constexpr int kPar = kRU * kPitch;                           // 23 616
constexpr int kPlane = 2 * kPar;                             // 47 232
constexpr int kLds = 3 * kPlane;                             // 141 696
constexpr int kRound = 128;                                  // pixels per round
constexpr int kItems = 3;                                    // 32-byte items per thread and round (512 threads: <= 38 rows)
constexpr int kPart = 4 * 4096;                              // partial sums of the second tap half, one 4 KB block per tile
constexpr unsigned kOut = 0x80000000u;

struct Params {
  const float* X; const float* W; const float* bias; float* Y;
  int n_img, per_wg, in_relu, out_relu;
  unsigned long long* trace;              // (TRACE builds only: [workgroup][8 waves][32 rounds][8 stamps])
  unsigned char* bits;                    // r5, optional (needs out_relu): [pixel][8] bytes, bit r of byte q = Y[pixel][4 q + r] > 0
};

__device__ __forceinline__ f32x4_t load16(const sgpr128_t& d, unsigned voff, unsigned soff) {
  f32x4_t v;
  asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(v) : "v"(voff), "s"(d), "s"(soff));
  return v;
}
__device__ __forceinline__ f32x4_t load16b(const sgpr128_t& d, unsigned voff, unsigned soff) {
  f32x4_t v;
  asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:16" : "=v"(v) : "v"(voff), "s"(d), "s"(soff));
  return v;
}
template <int N>
__device__ __forceinline__ void wait_set(f32x4_t (&r)[kItems][2]) {
  static_assert(kItems == 3, "operand list below");
  asm volatile("s_waitcnt vmcnt(%6)"
               : "+v"(r[0][0]), "+v"(r[0][1]), "+v"(r[1][0]), "+v"(r[1][1]), "+v"(r[2][0]), "+v"(r[2][1]) : "n"(N));
}
template <int N>
__device__ __forceinline__ void wait_item(f32x4_t (&r)[2]) {
  asm volatile("s_waitcnt vmcnt(%2)" : "+v"(r[0]), "+v"(r[1]) : "n"(N));
}

// rows [0, end_row(r)) of the run are what rounds 0..r read
__device__ __forceinline__ int end_row(int r, int total, int rows) {
  int pl = kRound * r + kRound - 1; if (pl > total - 1) pl = total - 1;
  if (pl < 0) return 0;
  const unsigned li = (unsigned)pl / (unsigned)kP, pix = (unsigned)pl - li * kP;
  const int e = (int)(kIH * li + 2 * (pix / (unsigned)kOW) + 4);
  return e < rows ? e : rows;
}

// EXP (timing experiments only, results wrong): 1 no split / LDS writes, 2 no loads, 4 no MFMAs, 8 no operand reads
template <bool TRACE, bool RELU_IN, int EXP = 0, bool BITS = false>
__global__ void __launch_bounds__(512, 2)
wfx_kernel(const Params p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, kq = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tile = wave & 3, kh = wave >> 2;                 // pixel tile of the round; half of the taps (ky = 2 kh, 2 kh + 1)
  // ds_read_b128 serves a wave in the lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} (and + 32): the lanes of a
  // group take 16 CONSECUTIVE pixels of the tile, which is what the ring's slot sequence keeps apart
  const int l5 = lane & 31;
  const int px = l5 < 4 ? l5 : l5 < 12 ? l5 + 12 : l5 < 16 ? l5 - 8 : l5 < 20 ? l5 + 8 : l5 < 28 ? l5 - 12 : l5;
  const int img0 = blockIdx.x * p.per_wg;
  int nimg = p.n_img - img0; if (nimg > p.per_wg) nimg = p.per_wg;
  if (nimg <= 0) return;
  const int total = nimg * kP, rows = nimg * kIH;
  const int rounds = (total + kRound - 1) / kRound;
  const sgpr128_t xd = xg::make_view_words(p.X + (long long)img0 * (kIH * kIW * 16), (long long)rows * kRowBytes);

  // ---- weights of this wave's eight taps: three planes of W[8 kh + t][8 kq + e][co = lane & 31], e = 0..7 -------- //
  bf16x8_t wh[8], wm[8], wl[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = p.W[((8 * kh + t) * 16 + 8 * kq + e) * 32 + l5];
    u32x4_t h, m, l;
#pragma unroll
    for (int e = 0; e < 4; ++e) { unsigned a, b, c; xg::split2(v[2 * e], v[2 * e + 1], a, b, c); h[e] = a; m[e] = b; l[e] = c; }
    wh[t] = __builtin_bit_cast(bf16x8_t, h); wm[t] = __builtin_bit_cast(bf16x8_t, m); wl[t] = __builtin_bit_cast(bf16x8_t, l);
  }
  f32x16_t acc0;                                             // the accumulator's start: bias in the first half's waves
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int q = 0; q < 4; ++q) acc0[4 * g + q] = (p.bias && kh == 0) ? p.bias[8 * g + 4 * kq + q] : 0.f;

  // ---- staging ------------------------------------------------------------------------------------------------- //
  f32x4_t ld[2][kItems][2];
  auto issue1 = [&](f32x4_t (&s)[kItems][2], int lo, int hi, int i) {   // load i (item i / 2, half i % 2) of rows [lo, hi)
    const unsigned n = (unsigned)(hi - lo) * 40u;
    const unsigned soff = __builtin_amdgcn_readfirstlane((unsigned)lo * (unsigned)kRowBytes);
    const unsigned q = (unsigned)tid + 512u * (i >> 1);
    const unsigned voff = (q < n && !(EXP & 64)) ? q * 32u : kOut;
    if (i & 1) s[i >> 1][1] = load16b(xd, voff, soff); else s[i >> 1][0] = load16(xd, voff, soff);
  };
  // Item k of rows [lo, hi) goes to LDS in two halves (four values -> 8 bytes per plane).  Branch free: items past the
  // rows were loaded as zeros and go to a pad slot nobody reads.
  auto put_addr = [&](int k, int lo, int hi) -> unsigned {
    const unsigned q = (unsigned)tid + 512u * k;
    const unsigned rr = q / 40u, c = q - rr * 40u, grow = (unsigned)lo + rr, ix = c >> 1, half = c & 1u;
    const unsigned u = (grow >> 1) % (unsigned)kRU;
    const unsigned dst = (grow & 1u) * kPar + u * kPitch + (half * 2u + (ix & 1u)) * 160u + (ix >> 1) * 16u;
    return q < (unsigned)(hi - lo) * 40u ? dst : 640u;
  };
  auto put_pair = [&](float f0, float f1, unsigned& h, unsigned& m, unsigned& l) {
    if (RELU_IN) { f0 = fmaxf(f0, 0.f); f1 = fmaxf(f1, 0.f); }
    // by truncation (plain full-rate VALU; v_cvt_pk_bf16_f32 issues at a quarter of that): still h + m + l == f exactly.
    // The weights are split with round-to-nearest, so the dropped products am wl + al wm keep a random sign; their
    // bound doubles to 2^-24 |x w|, half an ulp of the product.
    const xg::f32x2_t x = {f0, f1};
    const xg::u32x2_t xu = xg::hi_part(__builtin_bit_cast(xg::u32x2_t, x));
    const xg::f32x2_t r1 = x - __builtin_bit_cast(xg::f32x2_t, xu);                 // (v_pk_add_f32)
    const xg::u32x2_t ru = __builtin_bit_cast(xg::u32x2_t, r1) & 0xFFFF0000u;
    const xg::u32x2_t r2 = __builtin_bit_cast(xg::u32x2_t, r1 - __builtin_bit_cast(xg::f32x2_t, ru));
    h = __builtin_amdgcn_perm(xu[1], xu[0], 0x07060302u);
    m = __builtin_amdgcn_perm(ru[1], ru[0], 0x07060302u);
    l = __builtin_amdgcn_perm(r2[1], r2[0], 0x07060302u);
  };
  auto put_half = [&](const f32x4_t& it, int k, int j, int lo, int hi) {
    unsigned h[2], m[2], l[2];
    put_pair(it[0], it[1], h[0], m[0], l[0]); put_pair(it[2], it[3], h[1], m[1], l[1]);
    const unsigned dst = put_addr(k, lo, hi) + 8 * j;
    *reinterpret_cast<xg::u32x2_t*>(smem + dst) = xg::u32x2_t{h[0], h[1]};
    *reinterpret_cast<xg::u32x2_t*>(smem + dst + kPlane) = xg::u32x2_t{m[0], m[1]};
    *reinterpret_cast<xg::u32x2_t*>(smem + dst + 2 * kPlane) = xg::u32x2_t{l[0], l[1]};
  };

  unsigned long long* tg = TRACE ? p.trace + ((long long)blockIdx.x * 8 + wave) * 256 : nullptr;
  auto stamp = [&](int r, int k) {
    if (TRACE && r < 31) { const unsigned long long c = __builtin_amdgcn_s_memtime(); if (lane == 0) tg[r * 8 + k] = c; }
    if (TRACE && r < 31 && k == 0) { const unsigned long long c = __builtin_amdgcn_s_memrealtime(); if (lane == 0) tg[r * 8 + 7] = c; }   // (100 MHz)
  };
  const int e0 = end_row(0, total, rows), e1 = end_row(1, total, rows);
#pragma unroll
  for (int i = 0; i < 2 * kItems; ++i) issue1(ld[0], 0, e0, i);
#pragma unroll
  for (int i = 0; i < 2 * kItems; ++i) issue1(ld[1], e0, e1, i);
  wait_set<2 * kItems>(ld[0]);
#pragma unroll
  for (int k = 0; k < kItems; ++k) { put_half(ld[0][k][0], k, 0, 0, e0); put_half(ld[0][k][1], k, 1, 0, e0); }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

  const long long ybase = (long long)img0 * kP * 32;
  unsigned char* blk = smem + kLds + tile * 4096;            // this tile's 4 KB exchange block
  unsigned char* part = blk + lane * 16;                     // the second half's partial sums: [quad][lane] x 16 bytes
  f32x16_t acc = acc0;
  // The first half's waves finish round r - 1 at the start of round r: their partner's partial sums were published by
  // the barrier in between, and while they add and activate, the partner (same SIMD) is multiplying already.  The sums
  // go back into the block as [pixel][32 channels] rows (16-byte chunk c of pixel x at chunk c ^ (x & 7)); the wave
  // stores them at the END of its round, 1 KB of consecutive addresses per instruction -- behind the round's counted
  // waits, which a pending store would stretch (stores and loads share vmcnt and do not retire in order).
  auto finish = [&](int rr) {                                // rr: the round whose tile is being finished
    if (kh == 0) {
      const int Pf = kRound * rr + 32 * tile + px;
      f32x4_t q4[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) q4[g] = *reinterpret_cast<const f32x4_t*>(part + g * 1024);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4_t v;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float y = acc[4 * g + q] + q4[g][q];
          v[q] = p.out_relu ? __builtin_amdgcn_fmed3f(y, 0.f, __builtin_inff()) : y;
        }
        *reinterpret_cast<f32x4_t*>(blk + px * 128 + (((2 * g + kq) ^ (px & 7)) << 4)) = v;
        if (BITS && Pf < total) {                          // the next layer's data gradient reads this byte instead of the 16
          const u32x4_t bu = __builtin_bit_cast(u32x4_t, v);  // (post-ReLU: > 0 <=> bit pattern != 0, see stackconv.hip)
          const unsigned m01 = ((bu[1] < 1u ? bu[1] : 1u) << 1) | (bu[0] < 1u ? bu[0] : 1u);
          const unsigned m23 = ((bu[3] < 1u ? bu[3] : 1u) << 1) | (bu[2] < 1u ? bu[2] : 1u);
          p.bits[((long long)img0 * kP + Pf) * 8 + 2 * g + kq] = (unsigned char)((m23 << 2) | m01);
        }
      }
    }
  };
  // outputs of round r, from the block: piece i = 8 pixels = 1 KB
  f32x4_t ov;
  auto out_read = [&](int i) {
    const int pr = 8 * i + (lane >> 3);
    ov = *reinterpret_cast<const f32x4_t*>(blk + pr * 128 + (((lane & 7) ^ (pr & 7)) << 4));
  };
  auto out_store = [&](int r, int i) {
    const int P = kRound * r + 32 * tile + 8 * i + (lane >> 3);
    if (P < total) *reinterpret_cast<f32x4_t*>(p.Y + ybase + (long long)P * 32 + 4 * (lane & 7)) = ov;
  };
  // one round: reads rows of round r, writes the rows of r + 1 from `wr`, requests the rows of r + 2 into `nx`
  auto round = [&](int r, f32x4_t (&wr)[kItems][2], f32x4_t (&nx)[kItems][2]) {
    stamp(r, 0);
    const int lo1 = end_row(r, total, rows), hi1 = end_row(r + 1, total, rows), hi2 = end_row(r + 2, total, rows);
    const int P = kRound * r + 32 * tile + px;
    const unsigned Pc = (unsigned)(P < total ? P : total - 1);
    const unsigned li = Pc / (unsigned)kP, pix = Pc - li * kP, oy = pix / (unsigned)kOW, ox = pix - oy * kOW;
    const unsigned u0 = 10u * li + oy + (unsigned)kh;        // both of this wave's kernel rows sit in row pair u0
    const unsigned o0 = (u0 % (unsigned)kRU) * kPitch + ox * 16u + (unsigned)kq * 320u;
    unsigned o1 = o0 + kPlane, o2 = o0 + 2 * kPlane;        // a base per plane: every tap offset is an immediate
    asm volatile("" : "+v"(o1), "+v"(o2));                   // (pinned: otherwise re-derived from o0 with an add per read)
    const unsigned char* bpl[3] = {smem + o0, smem + o1, smem + o2};
    bf16x8_t xb[3][3];
    auto fetch = [&](bf16x8_t (&x)[3], int t) {              // tap t of this wave: ky = 2 kh + (t >> 2), kx = t & 3
      const int off = (t >> 2) * kPar + (t & 1) * 160 + ((t >> 1) & 1) * 16;
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) {
        if (EXP & 8) x[pl] = wl[(t + pl) & 7]; else x[pl] = *reinterpret_cast<const bf16x8_t*>(bpl[pl] + off);
      }
    };
    fetch(xb[0], 0);                                         // operands are requested two taps ahead (three buffers)
    fetch(xb[1], 1);
    if (r > 0) finish(r - 1);
    acc = acc0;
    stamp(r, 1);
#define WFX_SB __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      WFX_SB
      if (t + 2 < 8) fetch(xb[(t + 2) % 3], t + 2);
      const bf16x8_t (&x)[3] = xb[t % 3];
      // `wr` item k was requested in taps 2k, 2k + 1 of the previous round.  Younger LOADS at tap 2 + 2k: the rest of
      // `wr` (4 - 2k) and this round's first 2 + 2k requests, 6 in all; stores do not retire in order with loads, but
      // they can only make the counter larger (the wait longer), never let a load of item k pass for complete.
      if (t >= 2 && !(t & 1) && !(EXP & 2) && !(EXP & 32)) wait_item<2 * kItems>(wr[(t - 2) >> 1]);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl[t], x[0], acc, 0, 0, 0);
      if (t < 2 * kItems && !(EXP & 2)) issue1(nx, hi1, hi2, t);
      WFX_SB
      if (!(EXP & 4)) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[t], x[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wm[t], x[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wm[t], x[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[t], x[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[t], x[0], acc, 0, 0, 0);
      } else {
        acc[t] += (float)x[1][0] + (float)x[2][1];
      }
      if (t >= 2 && !(EXP & 1)) put_half(wr[(t - 2) >> 1][(t - 2) & 1], (t - 2) >> 1, (t - 2) & 1, lo1, hi1);
    }
    WFX_SB
#undef WFX_SB
    if (kh == 0 && r > 0) {                                  // behind the taps (early in the round they slowed it by 5 %)
#pragma unroll
      for (int i = 0; i < 4; ++i) { out_read(i); out_store(r - 1, i); }
    }
    stamp(r, 2);
    // two barriers: behind the first every first-half wave has read the previous round's outputs out of the block, so
    // the block can be rewritten; the second publishes it together with the rows.
    // (A flag from the partner instead of the first barrier was not faster.)
    asm volatile("s_barrier" ::: "memory");
    if (kh == 1) {
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<f32x4_t*>(part + g * 1024) = f32x4_t{acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
    }
    stamp(r, 3);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    stamp(r, 4);
  };
  for (int r = 0; r < rounds; r += 2) {
    round(r, ld[1], ld[0]);
    if (r + 1 < rounds) round(r + 1, ld[0], ld[1]);
  }
  finish(rounds - 1);
  if (kh == 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { out_read(i); out_store(rounds - 1, i); }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

inline bool plan(Params& p, const seedhip_conv_geom* g) {
  if (g->pad_t || g->pad_l || g->kh != 4 || g->kw != 4 || g->stride != 2 || g->cin != 16 || g->cout != 32 || g->ld_in != 16 ||
      g->ld_out != 32 || g->ih != kIH || g->iw != kIW || g->oh != kOW || g->ow != kOW)
    return false;
  constexpr int min_img = 256;      // faster than the fp32 kernels from inference batches on (273 images: 9.9 vs 12.7 us forward)
  if (g->n_img < min_img) return false;
  memset(&p, 0, sizeof(p));
  p.n_img = g->n_img;
  return true;
}

inline int launch(Params& p, hipStream_t s) {
  static const int cus = xg::cu_count();
  p.per_wg = (p.n_img + cus - 1) / cus;
  const int grid = (p.n_img + p.per_wg - 1) / p.per_wg;
  constexpr int kBytes = kLds + kPart;
#define WFX_GO(R_, B_) { \
    static const bool ok = hipFuncSetAttribute((const void*)wfx_kernel<false, R_, 0, B_>, hipFuncAttributeMaxDynamicSharedMemorySize, kBytes) == hipSuccess; \
    if (!ok) return -1; \
    hipLaunchKernelGGL((wfx_kernel<false, R_, 0, B_>), dim3(grid), dim3(512), kBytes, s, p); }
  if (p.bits) { if (p.in_relu) WFX_GO(true, true) else WFX_GO(false, true) }
  else if (p.in_relu) WFX_GO(true, false) else WFX_GO(false, false)
#undef WFX_GO
  return check_launch("wfx_kernel");
}

}  // namespace wfx
}  // namespace seedhip

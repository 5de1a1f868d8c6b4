// Data gradient of the second Atari conv -- Conv2D(32, 4, 2) on 20 x 20 x 16 (/root/reference/atari/networks.py:236;
// the gradient TensorFlow derives for it) -- on the BF16 matrix pipe through the exact three-way split (xgemm.h, wfx.h):
//     dX[img, 2a + py, 2b + px, ci] = mask * sum_{dy, dx, co} dY[img, a - dy, b - dx, co] W[py + 2 dy, px + 2 dx, ci, co]
// The four parity classes (py, px) of a 2 x 2 "super-pixel" (a, b) read the SAME four dY pixels: one GEMM with
// M = super-pixels (10 x 10 per image), N = 4 classes x 16 channels = 64, K = 4 taps x 32 channels = 128
// (wsgemm.h's formulation; that fp32-MFMA kernel needs 171 us in the cfg2 step).
//
// Structure: wfx.h's, transposed -- and simpler, because N = 64 splits between the two waves of a tile instead of the
// reduction: one 8-wave workgroup per CU over a run of images; a ROUND is 128 consecutive super-pixels = four tiles of
// 32; wave (tile, py) computes the input rows of parity py: eight 16-deep steps (dy, dx, channel half) of six plane
// products, 48 MFMAs per round, no exchange between waves and ONE barrier per round.
//   * WEIGHTS: lane (n = lane & 31 -> px = n >> 4, ci = n & 15; kq) holds W[py + 2 dy][px + 2 dx][ci][16 c + 8 kq ..] for
//     dy, dx, c in {0, 1} as three bf16 planes: 96 registers;
//   * dY: the run as an array of PADDED rows (11 per image: a zero row above and below the nine, a zero pixel left and
//     right), staged once into a ring of 32 rows x 3 planes; a row holds its four 8-channel chunks one after the other,
//     11 sixteen-byte slots each, rows 58 slots apart (58 = 10 (mod 16): the next super-pixel row continues the slot
//     sequence, conflict factor 1.12).  Border items are requested out of range: the buffer load returns the zeros;
//   * OUTPUT: a lane ends up with four channels of one pixel per accumulator quad; the wave passes them through its own
//     4 KB block ([super-pixel][128 bytes], chunk index swizzled by the super-pixel) and reads them back 1 KB per store
//     instruction -- eight consecutive super-pixels of one input row.  The ReLU mask (the layer's input, same addresses
//     as dX) comes by LDS-DMA into a second 4 KB area, requested a whole round before it is needed.
//
// Measured (8 442 images, MI355X, microbenchmark with the mask): 140-155 us against wsgemm.h's 189.  SEEDHIP_WDX_EXP
// builds (results wrong): without the output path 99, operand reads + MFMAs alone 79, operand reads alone 39; without the
// mask 141.  As in wfx.h the parts add up instead of overlapping; three placements of the output path (end of the
// round / spread over the next round's steps with counted waits / with the memory phases of a SIMD's two waves
// staggered) are within 5 % of each other.
#pragma once
#include <type_traits>
#include "wfx.h"

namespace seedhip {
namespace wdx {

using xg::bf16x8_t;
using xg::f32x4_t;
using xg::f32x16_t;
using xg::u32x4_t;
using xg::sgpr128_t;

constexpr int kSP = 100;                                     // super-pixels per image
constexpr int kPR = 11;                                      // padded dY rows per image
constexpr int kChunk = 12 * 16;                              // bytes between the 8-channel chunks of a row (11 slots used)
constexpr int kRS = 58 * 16;                                 // bytes of a padded row in one plane
constexpr int kR = 32;                                       // ring rows (two rounds span at most 31)
constexpr int kPlane = kR * kRS;                             // 29 696
constexpr int kRing = 3 * kPlane;                            // 89 088
constexpr int kLds = kRing + 8 * 4096 + 8 * 4096;            // ring + a 4 KB output block and a 4 KB mask area per wave: 154 624
constexpr int kRound = 128;
constexpr int kItems = 2;                                    // 32-byte items per thread and round (<= 15 new rows x 44)
constexpr unsigned kOut = 0x80000000u;

struct Params {
  const float* dY; const float* W; const float* X; float* dX;
  const unsigned char* bits;              // r5: the ReLU mask as bytes [pixel][cin / 4], bit r of byte q = X[pixel][4 q + r] > 0
  int n_img, per_wg;                      //     (seedhip_conv2d_stack_fwd_bits): 17 MB at cfg2 where X is 275 MB
};

// padded rows [0, end_row(r)) of the run are what rounds 0..r read
__device__ __forceinline__ int end_row(int r, int total, int rows) {
  int sl = kRound * r + kRound - 1; if (sl > total - 1) sl = total - 1;
  if (sl < 0) return 0;
  const unsigned li = (unsigned)sl / (unsigned)kSP, sp = (unsigned)sl - li * kSP;
  const int e = (int)(kPR * li + sp / 10u + 2);
  return e < rows ? e : rows;
}

// MASK: 0 none, 1 the fp32 activation read for its sign (by LDS-DMA), 2 the byte mask (one byte per lane and piece, straight
// into a register a round ahead).  EXP (timing experiments only, results wrong; not instantiated by the library):
// 1 no split / LDS writes, 2 no row loads, 4 no MFMAs, 8 no output path
template <int MASK, int EXP = 0>
__global__ void __launch_bounds__(512, 2)
wdx_kernel(const Params p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, kq = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tile = wave & 3, py = wave >> 2;                 // super-pixel tile of the round; row parity of its outputs
  const int l5 = lane & 31;                                  // (ds_read_b128 lane groups: see wfx.h)
  const int sl = l5 < 4 ? l5 : l5 < 12 ? l5 + 12 : l5 < 16 ? l5 - 8 : l5 < 20 ? l5 + 8 : l5 < 28 ? l5 - 12 : l5;
  const int img0 = blockIdx.x * p.per_wg;
  int nimg = p.n_img - img0; if (nimg > p.per_wg) nimg = p.per_wg;
  if (nimg <= 0) return;
  const int total = nimg * kSP, rows = nimg * kPR;
  const int rounds = (total + kRound - 1) / kRound;
  const sgpr128_t yd = xg::make_view_words(p.dY + (long long)img0 * (81 * 32), (long long)nimg * (81 * 32 * 4));
  const __amdgpu_buffer_rsrc_t xv = gemm::make_view(MASK == 1 ? p.X + (long long)img0 * 6400 : p.dX, (long long)nimg * 25600);
  const sgpr128_t bv = xg::make_view_words(reinterpret_cast<const float*>(MASK == 2 ? p.bits + (long long)img0 * 1600 : (const unsigned char*)p.dX),
                                           (long long)nimg * 1600);
  unsigned mb[4] = {0u, 0u, 0u, 0u};                         // MASK == 2: the lane's mask byte of piece j
  const __amdgpu_buffer_rsrc_t ov = gemm::make_view(p.dX + (long long)img0 * 6400, (long long)nimg * 25600);

  // ---- weights: step s = 4 dy + 2 dx + c: W[py + 2 dy][px + 2 dx][ci][16 c + 8 kq + e], e = 0..7 ------------------ //
  bf16x8_t wh[8], wm[8], wl[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    const int tap = (py + 2 * (s >> 2)) * 4 + (l5 >> 4) + 2 * ((s >> 1) & 1);
    const float* src = p.W + (tap * 16 + (l5 & 15)) * 32 + 16 * (s & 1) + 8 * kq;
    const f32x4_t v0 = *reinterpret_cast<const f32x4_t*>(src), v1 = *reinterpret_cast<const f32x4_t*>(src + 4);
    u32x4_t h, m, l;
    unsigned a, b, c;
    xg::split2(v0[0], v0[1], a, b, c); h[0] = a; m[0] = b; l[0] = c;
    xg::split2(v0[2], v0[3], a, b, c); h[1] = a; m[1] = b; l[1] = c;
    xg::split2(v1[0], v1[1], a, b, c); h[2] = a; m[2] = b; l[2] = c;
    xg::split2(v1[2], v1[3], a, b, c); h[3] = a; m[3] = b; l[3] = c;
    wh[s] = __builtin_bit_cast(bf16x8_t, h); wm[s] = __builtin_bit_cast(bf16x8_t, m); wl[s] = __builtin_bit_cast(bf16x8_t, l);
  }

  // ---- staging: item q of padded rows [lo, hi) = chunk c of padded pixel pc of row lo + q / 44 -------------------- //
  f32x4_t ld[2][kItems][2];
  auto item_src = [&](int k, int lo, int hi) -> unsigned {   // byte offset into the run's dY, or out of range
    const unsigned q = (unsigned)tid + 512u * k;
    const unsigned rr = q / 44u, rem = q - rr * 44u, pc = rem >> 2, c = rem & 3u;
    const unsigned prow = (unsigned)lo + rr, li = prow / (unsigned)kPR, r1 = prow - li * kPR;
    const bool in = prow < (unsigned)hi && r1 - 1u < 9u && pc - 1u < 9u;
    return in ? (((li * 9u + r1 - 1u) * 9u + pc - 1u) * 32u + 8u * c) * 4u : kOut;
  };
  auto issue1 = [&](f32x4_t (&s)[kItems][2], int lo, int hi, int i) {
    const unsigned voff = item_src(i >> 1, lo, hi);
    if (i & 1) s[i >> 1][1] = wfx::load16b(yd, voff, 0u); else s[i >> 1][0] = wfx::load16(yd, voff, 0u);
  };
  auto put_half = [&](const f32x4_t& it, int k, int j, int lo, int hi) {
    const unsigned q = (unsigned)tid + 512u * k;
    const unsigned rr = q / 44u, rem = q - rr * 44u, pc = rem >> 2, c = rem & 3u;
    const unsigned prow = (unsigned)lo + rr;
    unsigned dst = (prow & (unsigned)(kR - 1)) * kRS + c * kChunk + pc * 16u + 8u * j;
    dst = prow < (unsigned)hi ? dst : (unsigned)(kRS - 16);  // (past the rows: a pad slot of row 0)
    unsigned h[2], m[2], l[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const xg::f32x2_t x = {it[2 * e], it[2 * e + 1]};
      const xg::u32x2_t xu = xg::hi_part(__builtin_bit_cast(xg::u32x2_t, x));     // split by truncation (wfx.h)
      const xg::f32x2_t r1 = x - __builtin_bit_cast(xg::f32x2_t, xu);
      const xg::u32x2_t ru = __builtin_bit_cast(xg::u32x2_t, r1) & 0xFFFF0000u;
      const xg::u32x2_t r2 = __builtin_bit_cast(xg::u32x2_t, r1 - __builtin_bit_cast(xg::f32x2_t, ru));
      h[e] = __builtin_amdgcn_perm(xu[1], xu[0], 0x07060302u);
      m[e] = __builtin_amdgcn_perm(ru[1], ru[0], 0x07060302u);
      l[e] = __builtin_amdgcn_perm(r2[1], r2[0], 0x07060302u);
    }
    *reinterpret_cast<xg::u32x2_t*>(smem + dst) = xg::u32x2_t{h[0], h[1]};
    *reinterpret_cast<xg::u32x2_t*>(smem + dst + kPlane) = xg::u32x2_t{m[0], m[1]};
    *reinterpret_cast<xg::u32x2_t*>(smem + dst + 2 * kPlane) = xg::u32x2_t{l[0], l[1]};
  };

  const int e0 = end_row(0, total, rows), e1 = end_row(1, total, rows);
#pragma unroll
  for (int i = 0; i < 2 * kItems; ++i) issue1(ld[0], 0, e0, i);
#pragma unroll
  for (int i = 0; i < 2 * kItems; ++i) issue1(ld[1], e0, e1, i);
  // (r6) The items are read BEHIND an operand-free wait through xg::move_item: the tied "+v" operands of wait_set let
  // hipcc allocate them elsewhere and copy the in-flight registers in FRONT of the waiting statement -- it did, at the
  // round loop's back edge (tools/isa_inflight.py with the wait COUNT held against the requests issued behind: eight
  // v_mov_b64 of items requested a round earlier, a race their ~1 us of head start won in every test).
  asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * kItems));
#pragma unroll
  for (int k = 0; k < kItems; ++k) {
    const f32x4_t i0 = xg::move_item(ld[0][k][0]), i1 = xg::move_item(ld[0][k][1]);
    put_half(i0, k, 0, 0, e0); put_half(i1, k, 1, 0, e0);
  }

  unsigned char* blk = smem + kRing + wave * 4096;           // this wave's outputs, [super-pixel][128 B], chunks swizzled
  unsigned char* msk = smem + kRing + 8 * 4096 + wave * 4096;   // this wave's mask pieces, [piece][lane] x 16 bytes

  // byte offset (in dX / X of the run) of the 16 bytes lane L of piece j = 0..3 holds: super-pixel 8 j + (L >> 3) of
  // the tile, chunk L & 7 of the 128 bytes of its two pixels in input row 2 a + py; out of range past the run
  unsigned oofs[4];
  auto out_offset = [&](int r, int j) {
    const unsigned P = (unsigned)(kRound * r + 32 * tile + 8 * j) + (unsigned)(lane >> 3);
    const unsigned li = P / (unsigned)kSP, sp = P - li * kSP, a = sp / 10u, b = sp - a * 10u;
    oofs[j] = P < (unsigned)total ? ((li * 20u + 2u * a + (unsigned)py) * 20u + 2u * b) * 64u + (unsigned)(lane & 7) * 16u : kOut;
  };
  auto mask_request = [&](int j) {                           // 1 KB piece j into the wave's mask area
    typedef __attribute__((address_space(3))) void lds_void_t;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(xv, (lds_void_t*)(msk + j * 1024), 16, oofs[j], 0, 0, 0);
  };
  // the byte of the lane's 16 output bytes: the mask is indexed like dX / 16
  // (an asm load like the row items': with the builtin hipcc guards the re-used register with counted waits that also
  // wait for the output stores in flight -- vmcnt(1) in front of the fourth request -- and drains the queue at every
  // step of the first round; the wave's own full wait at the head of its memory phase covers these loads)
  auto bits_request = [&](int j) {
    const unsigned voff = oofs[j] == kOut ? kOut : oofs[j] >> 4;
    asm volatile("buffer_load_ubyte %0, %1, %2, 0 offen" : "=v"(mb[j]) : "v"(voff), "s"(bv));
  };
  auto bits_wait = [&]() { asm volatile("s_waitcnt vmcnt(0)" : "+v"(mb[0]), "+v"(mb[1]), "+v"(mb[2]), "+v"(mb[3])); };
  auto out_piece = [&](int j) {                              // 8 super-pixels = 1 KB of consecutive addresses
    const int spl = 8 * j + (lane >> 3);
    f32x4_t v = *reinterpret_cast<const f32x4_t*>(blk + spl * 128 + (((lane & 7) ^ (spl & 7)) << 4));
    if (MASK == 1) {
      const f32x4_t mk = *reinterpret_cast<const f32x4_t*>(msk + j * 1024 + lane * 16);
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = mk[q] > 0.f ? v[q] : 0.f;
    }
    if (MASK == 2) {
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = ((mb[j] >> q) & 1u) ? v[q] : 0.f;
    }
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), ov, oofs[j], 0, 0);
  };
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

  f32x16_t acc;
  // One round.  Vector-memory queue of a wave, oldest first, when round r starts: [row items of round r + 1: 4]
  // [mask of round r - 1's tile... consumed] ... -- see the counted waits below.
  auto round = [&](auto PH, int r, f32x4_t (&wr)[kItems][2], f32x4_t (&nx)[kItems][2]) {
    constexpr int ph = decltype(PH)::value;                  // the wave's memory phase: steps 4 ph .. 4 ph + 3
    const int lo1 = end_row(r, total, rows), hi1 = end_row(r + 1, total, rows), hi2 = end_row(r + 2, total, rows);
    const int P = kRound * r + 32 * tile + sl;
    const unsigned Pc = (unsigned)(P < total ? P : total - 1);
    const unsigned li = Pc / (unsigned)kSP, sp = Pc - li * kSP, a = sp / 10u, b = sp - a * 10u;
    const unsigned prow = kPR * li + a + 1u;                 // padded row of dy = 0; dy = 1 reads the one above
    const unsigned inrow = (unsigned)kq * kChunk + b * 16u;
    unsigned o[2][3];
#pragma unroll
    for (int dy = 0; dy < 2; ++dy) {
      o[dy][0] = ((prow - dy) & (unsigned)(kR - 1)) * kRS + inrow;
      o[dy][1] = o[dy][0] + kPlane; o[dy][2] = o[dy][0] + 2 * kPlane;
      asm volatile("" : "+v"(o[dy][1]), "+v"(o[dy][2]));     // (a base per plane: every step offset is an immediate)
    }
    bf16x8_t xb[2][3];
    auto fetch = [&](bf16x8_t (&x)[3], int s) {              // step s: dy = s >> 2, dx = (s >> 1) & 1 (padded pixel b + 1 - dx), channels 16 (s & 1) + 8 kq
      const int off = (s & 1) * 2 * kChunk + (1 - ((s >> 1) & 1)) * 16;
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) x[pl] = *reinterpret_cast<const bf16x8_t*>(smem + o[s >> 2][pl] + off);
    };
    fetch(xb[0], 0);
    // ALL vector-memory work of a wave happens in four consecutive steps of the round -- row items requested, previous
    // outputs stored, mask pieces re-requested -- steps 0-3 in the waves with py = 0, steps 4-7 in their SIMD partners
    // (the 64 B/clk path to memory is shared by the CU: one wave of a SIMD waits on it while the other multiplies).
    // When a wave's phase begins, everything in its queue is a whole round old: one full wait, no counting.  (Counted
    // waits BETWEEN stores wait for the stores' acknowledgements: loads and stores share vmcnt.)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
    f32x4_t item[2 * kItems];                                // the round's row items behind their wait
#define WDX_SB __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      WDX_SB
      if (s + 1 < 8) fetch(xb[(s + 1) & 1], s + 1);
      const bf16x8_t (&x)[3] = xb[s & 1];
      const int j = s & 3;
      const bool mine = (s >> 2) == ph;
      if (mine && j == 0) {
        asm volatile("s_waitcnt vmcnt(0)");
#pragma unroll
        for (int k = 0; k < 2 * kItems; ++k) item[k] = xg::move_item(wr[k >> 1][k & 1]);
        if (MASK == 2) bits_wait();
      }
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl[s], x[0], acc, 0, 0, 0);
      if (mine && !(EXP & 2)) issue1(nx, hi1, hi2, j);
      WDX_SB
      if (!(EXP & 4)) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[s], x[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wm[s], x[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wm[s], x[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[s], x[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[s], x[0], acc, 0, 0, 0);
      } else {
        acc[s] += (float)x[1][0] + (float)x[2][1];
      }
      if (mine) {
        if (!(EXP & 1)) put_half(item[j], j >> 1, j & 1, lo1, hi1);
        if (!(EXP & 8)) {
          // piece j of the PREVIOUS round's outputs leaves (its sums wait in the block, its mask piece was requested in
          // this step of the previous round), then the same slot takes the mask of THIS round's tile
          if (r > 0) out_piece(j);
          out_offset(r, j);
          if (MASK == 1) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); mask_request(j); }
          if (MASK == 2) bits_request(j);
        }
      }
    }
    WDX_SB
#undef WDX_SB
    // the tile's sums into the wave's own block (a lane holds four channels of one pixel per quad; the stores of the
    // next round want 1 KB of consecutive addresses): its previous content left in steps 4-7
    if (!(EXP & 8)) {
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<f32x4_t*>(blk + sl * 128 + (((2 * g + kq) ^ (sl & 7)) << 4)) =
            f32x4_t{acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
    } else if (acc[0] == 123.f) out_piece(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");                  // (the rows for round r + 1 were written before the lgkmcnt(0) above)
  };
  if (py == 0) {
    for (int r = 0; r < rounds; r += 2) {
      round(std::integral_constant<int, 0>(), r, ld[1], ld[0]);
      if (r + 1 < rounds) round(std::integral_constant<int, 0>(), r + 1, ld[0], ld[1]);
    }
  } else {
    for (int r = 0; r < rounds; r += 2) {
      round(std::integral_constant<int, 1>(), r, ld[1], ld[0]);
      if (r + 1 < rounds) round(std::integral_constant<int, 1>(), r + 1, ld[0], ld[1]);
    }
  }
  // the last round's outputs (all four pieces read before the first store: a 16-byte store's data registers are not
  // handed to the next read at once, tools/isa_store_hazard.py)
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  if (MASK == 2) bits_wait();
  if (!(EXP & 8)) {
    f32x4_t v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int spl = 8 * j + (lane >> 3);
      v[j] = *reinterpret_cast<const f32x4_t*>(blk + spl * 128 + (((lane & 7) ^ (spl & 7)) << 4));
      if (MASK == 1) {
        const f32x4_t mk = *reinterpret_cast<const f32x4_t*>(msk + j * 1024 + lane * 16);
#pragma unroll
        for (int q = 0; q < 4; ++q) v[j][q] = mk[q] > 0.f ? v[j][q] : 0.f;
      }
      if (MASK == 2) {
#pragma unroll
        for (int q = 0; q < 4; ++q) v[j][q] = ((mb[j] >> q) & 1u) ? v[j][q] : 0.f;
      }
      asm volatile("" : "+v"(v[j]));
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v[j]), ov, oofs[j], 0, 0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

inline bool plan(Params& p, const seedhip_conv_geom* g) {
  if (g->pad_t || g->pad_l || g->kh != 4 || g->kw != 4 || g->stride != 2 || g->cin != 16 || g->cout != 32 || g->ld_in != 16 ||
      g->ld_out != 32 || g->ih != 20 || g->iw != 20 || g->oh != 9 || g->ow != 9)
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
#define WDX_GO(M_) { \
    static const bool ok = hipFuncSetAttribute((const void*)wdx_kernel<M_>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds) == hipSuccess; \
    if (!ok) return -1; \
    hipLaunchKernelGGL((wdx_kernel<M_>), dim3(grid), dim3(512), kLds, s, p); }
  if (p.bits) WDX_GO(2) else if (p.X) WDX_GO(1) else WDX_GO(0)
#undef WDX_GO
  return check_launch("wdx_kernel");
}

}  // namespace wdx
}  // namespace seedhip

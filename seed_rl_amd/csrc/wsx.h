// The 32 -> 32 channel 3 x 3 'same' convolutions of ImpalaDeep's second and third stacks
// (/root/reference/agents/vtrace/networks.py is the MLP agent; dmlab/networks.py:26-60 the residual stacks meant here:
// Conv2D(32, 3, padding='same') on 18 x 24 and 9 x 12 maps), forward AND data gradient, on the BF16 matrix pipe through
// the exact three-way operand split (xgemm.h; the machine of wfx.h / wdx.h):
//     forward        Y  = act( bias + residual + sum_{ky, kx, ci} relu?(X)[y + ky - 1, x + kx - 1, ci] W[ky, kx, ci, co] )
//     data gradient  dX = (mask > 0 ? sum_{ky, kx, co} dY[y + ky - 1, x + kx - 1, co] W[2 - ky, 2 - kx, ci, co] : 0) + add
// -- the same kernel with the weights read flipped and transposed, and the other epilogue (halo_fwd.h's two modes; that
// fp32-MFMA kernel needs 490 us per call at 5 376 images, 86 TF/s, and cfg3 calls it ten times a step).
//
// Structure: one 8-wave workgroup per CU over a run of images; a ROUND is 128 consecutive pixels of the run = four tiles
// of 32; the two waves of a tile split the reduction by input-channel halves (nine taps x 16 channels = nine 16-deep
// steps of six plane products, 54 MFMAs per round, 108 weight registers) and sit on one SIMD; partial sums cross through
// a 4 KB block per tile as in wfx.h.
//   * INPUT rows are PADDED (one pixel left and right, one zero row above and below each image: border items are
//     requested out of range and the buffer load returns the zeros), loaded once, split once (by truncation), three
//     bf16 planes in a ring of 16 (18 x 24 maps) / 32 (9 x 12) rows; a row holds its four 8-channel chunks one after the
//     other, rows `kRS` slots apart with kRS = map width (mod 16): the next image row continues the 16-byte slot sequence
//     (conflict factor 1.0 / 1.1).  Every element is used by nine taps: the split costs half of what it does in wfx.h
//     per MFMA;
//   * EPILOGUE operands (residual / ReLU mask, skip-path add) have the output's addresses: they come by LDS-DMA, 1 KB
//     per instruction, a round ahead; outputs leave as 1 KB stores through the tile's swizzled block (wfx.h);
//   * all vector-memory work of a wave happens in four consecutive steps (first wave of a tile: steps 0-3, second:
//     4-7) behind ONE full wait (wdx.h).
// Ring bounds and slot sequence are enumerated in tests/test_wfx_layout.py.
#pragma once
#include <type_traits>
#include "wfx.h"

namespace seedhip {
namespace wsx {

using xg::bf16x8_t;
using xg::f32x4_t;
using xg::f32x16_t;
using xg::u32x4_t;
using xg::sgpr128_t;

constexpr int kRound = 128;
constexpr int kItems = 2;                                    // 32-byte items per thread and round
constexpr unsigned kOut = 0x80000000u;

template <int H, int W>
struct Geo {
  static constexpr int kH = H, kW = W, kHP = H + 2, kWP = W + 2, kPX = H * W;
  static constexpr int kChunk = kWP * 16;                    // bytes between the four 8-channel chunks of a padded row
  static constexpr int kRSslots = 4 * kWP + ((W - 4 * kWP) % 16 + 16) % 16;   // = W (mod 16)
  static constexpr int kRS = kRSslots * 16;
  static constexpr int kR = W >= 24 ? 16 : 32;               // ring rows (two rounds span 16 / 30)
  static constexpr int kPlane = kR * kRS;
  static constexpr int kRing = 3 * kPlane;
  static constexpr int kDump = kRing + 4 * 4096 + 8 * 4096;  // 64 bytes nobody reads: items past the rows land there
  static constexpr int kLds = kDump + 64;                    // ring + a block and two operand areas per tile + dump
  static constexpr int kRowItems = 4 * kWP;
  static_assert(kRSslots % 16 == W % 16 && kRS >= 4 * kChunk, "row pitch");
};

struct Params {
  const float* X; const float* Wt; const float* bias; const float* A; const float* B; float* Y;   // A: residual / mask, B: add
  int n_img, per_wg, in_relu, out_relu;
  // ReLU masks as bytes (one per four channels: bit q of byte [pixel][quad] = x[pixel][4 quad + q] > 0, the layout of
  // seedhip_conv2d_fwd_bits): the forward writes the sign of its OUTPUT from the epilogue's registers (out_bits: 64
  // consecutive bytes per wave instruction); the data gradient of the layer that consumes that tensor through a ReLU
  // reads them (mask_bits) instead of the fp32 activation A -- 1/16 of its bytes.
  unsigned char* out_bits; const unsigned char* mask_bits;
};
__device__ __forceinline__ unsigned load_u8(const sgpr128_t& d, unsigned voff) {
  unsigned v;
  asm volatile("buffer_load_ubyte %0, %1, %2, 0 offen" : "=v"(v) : "v"(voff), "s"(d));
  return v;
}
__device__ __forceinline__ unsigned sign_bits(const f32x4_t& v) {
  return (v[0] > 0.f ? 1u : 0u) | (v[1] > 0.f ? 2u : 0u) | (v[2] > 0.f ? 4u : 0u) | (v[3] > 0.f ? 8u : 0u);
}

template <int N>
__device__ __forceinline__ void wait_set(f32x4_t (&r)[kItems][2]) {
  static_assert(kItems == 2, "operand list below");
  asm volatile("s_waitcnt vmcnt(%4)" : "+v"(r[0][0]), "+v"(r[0][1]), "+v"(r[1][0]), "+v"(r[1][1]) : "n"(N));
}

// padded rows [0, end_row(r)) of the run are what rounds 0..r read
template <typename G>
__device__ __forceinline__ int end_row(int r, int total, int rows) {
  int pl = kRound * r + kRound - 1; if (pl > total - 1) pl = total - 1;
  if (pl < 0) return 0;
  const unsigned li = (unsigned)pl / (unsigned)G::kPX, rem = (unsigned)pl - li * G::kPX;
  const int e = (int)(G::kHP * li + rem / (unsigned)G::kW + 3);
  return e < rows ? e : rows;
}

// DG: data-gradient mode (weights flipped and transposed; epilogue mask / add instead of bias / residual / ReLU)
template <typename G, bool DG>
__global__ void __launch_bounds__(512, 2)
wsx_kernel(const Params p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, kq = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tile = wave & 3, kh = wave >> 2;                 // pixel tile of the round; input-channel half
  const int l5 = lane & 31;                                  // (ds_read_b128 lane groups: see wfx.h)
  const int px = l5 < 4 ? l5 : l5 < 12 ? l5 + 12 : l5 < 16 ? l5 - 8 : l5 < 20 ? l5 + 8 : l5 < 28 ? l5 - 12 : l5;
  const int img0 = blockIdx.x * p.per_wg;
  int nimg = p.n_img - img0; if (nimg > p.per_wg) nimg = p.per_wg;
  if (nimg <= 0) return;
  const int total = nimg * G::kPX, rows = nimg * G::kHP;
  const int rounds = (total + kRound - 1) / kRound;
  const long long run_bytes = (long long)nimg * G::kPX * 128;
  const sgpr128_t xd = xg::make_view_words(p.X + (long long)img0 * G::kPX * 32, run_bytes);
  const __amdgpu_buffer_rsrc_t av = gemm::make_view((p.A ? p.A : p.Y) + (long long)img0 * G::kPX * 32, run_bytes);
  const __amdgpu_buffer_rsrc_t bv = gemm::make_view((p.B ? p.B : p.Y) + (long long)img0 * G::kPX * 32, run_bytes);
  const __amdgpu_buffer_rsrc_t ov = gemm::make_view(p.Y + (long long)img0 * G::kPX * 32, run_bytes);
  const bool has_a = p.A != nullptr, has_b = p.B != nullptr;
  const bool has_m = DG && p.mask_bits != nullptr, emit = !DG && p.out_bits != nullptr;
  const sgpr128_t md = xg::make_view_words(reinterpret_cast<const float*>((has_m ? p.mask_bits : (const unsigned char*)p.Y) + (long long)img0 * G::kPX * 8), has_m ? run_bytes >> 4 : 0);
  const __amdgpu_buffer_rsrc_t ev = gemm::make_view(reinterpret_cast<const float*>((emit ? p.out_bits : (unsigned char*)p.Y) + (long long)img0 * G::kPX * 8), emit ? run_bytes >> 4 : 0);
  unsigned mb[4] = {0u, 0u, 0u, 0u};

  // ---- weights of this wave's channel half: step t = 3 ky + kx, W_eff[t][ci = 16 kh + 8 kq + e][co = lane & 31] ----- //
  bf16x8_t wh[9], wm[9], wl[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    float v[8];
    if (DG) {                                                // w[2 - ky][2 - kx][co][ci]: eight consecutive ci
      const float* src = p.Wt + (((8 - t) * 32 + l5) * 32 + 16 * kh + 8 * kq);
      const f32x4_t v0 = *reinterpret_cast<const f32x4_t*>(src), v1 = *reinterpret_cast<const f32x4_t*>(src + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[e] = v0[e]; v[4 + e] = v1[e]; }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = p.Wt[(t * 32 + 16 * kh + 8 * kq + e) * 32 + l5];
    }
    u32x4_t h, m, l;
#pragma unroll
    for (int e = 0; e < 4; ++e) { unsigned a, b, c; xg::split2(v[2 * e], v[2 * e + 1], a, b, c); h[e] = a; m[e] = b; l[e] = c; }
    wh[t] = __builtin_bit_cast(bf16x8_t, h); wm[t] = __builtin_bit_cast(bf16x8_t, m); wl[t] = __builtin_bit_cast(bf16x8_t, l);
  }
  f32x16_t acc0;                                             // the accumulator's start: bias in the first half's waves
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int q = 0; q < 4; ++q) acc0[4 * g + q] = (!DG && p.bias && kh == 0) ? p.bias[8 * g + 4 * kq + q] : 0.f;

  // ---- staging: item q of padded rows [lo, hi) = chunk c of padded pixel pc of row lo + q / kRowItems -------------- //
  f32x4_t ld[2][kItems][2];
  auto item_src = [&](int k, int lo, int hi) -> unsigned {   // byte offset into the run's input, or out of range
    const unsigned q = (unsigned)tid + 512u * k;
    const unsigned rr = q / (unsigned)G::kRowItems, rem = q - rr * G::kRowItems, pc = rem >> 2, c = rem & 3u;
    const unsigned prow = (unsigned)lo + rr, li = prow / (unsigned)G::kHP, r1 = prow - li * G::kHP;
    const bool in = prow < (unsigned)hi && r1 - 1u < (unsigned)G::kH && pc - 1u < (unsigned)G::kW;
    return in ? (((li * G::kH + r1 - 1u) * G::kW + pc - 1u) * 32u + 8u * c) * 4u : kOut;
  };
  auto issue1 = [&](f32x4_t (&s)[kItems][2], int lo, int hi, int i) {
    const unsigned voff = item_src(i >> 1, lo, hi);
    if (i & 1) s[i >> 1][1] = wfx::load16b(xd, voff, 0u); else s[i >> 1][0] = wfx::load16(xd, voff, 0u);
  };
  auto put_half = [&](const f32x4_t& it, int k, int j, int lo, int hi) {
    const unsigned q = (unsigned)tid + 512u * k;
    const unsigned rr = q / (unsigned)G::kRowItems, rem = q - rr * G::kRowItems, pc = rem >> 2, c = rem & 3u;
    const unsigned prow = (unsigned)lo + rr;
    unsigned dst = (prow & (unsigned)(G::kR - 1)) * G::kRS + c * G::kChunk + pc * 16u + 8u * j;
    dst = prow < (unsigned)hi ? dst : (unsigned)(G::kDump + 8 * j);      // (items past the rows were loaded as zeros)
    unsigned h[2], m[2], l[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      xg::f32x2_t x = {it[2 * e], it[2 * e + 1]};
      if (!DG && p.in_relu) { x[0] = fmaxf(x[0], 0.f); x[1] = fmaxf(x[1], 0.f); }
      const xg::u32x2_t xu = xg::hi_part(__builtin_bit_cast(xg::u32x2_t, x));     // split by truncation (wfx.h)
      const xg::f32x2_t r1 = x - __builtin_bit_cast(xg::f32x2_t, xu);
      const xg::u32x2_t ru = __builtin_bit_cast(xg::u32x2_t, r1) & 0xFFFF0000u;
      const xg::u32x2_t r2 = __builtin_bit_cast(xg::u32x2_t, r1 - __builtin_bit_cast(xg::f32x2_t, ru));
      h[e] = __builtin_amdgcn_perm(xu[1], xu[0], 0x07060302u);
      m[e] = __builtin_amdgcn_perm(ru[1], ru[0], 0x07060302u);
      l[e] = __builtin_amdgcn_perm(r2[1], r2[0], 0x07060302u);
    }
    const bool real = prow < (unsigned)hi;
    *reinterpret_cast<xg::u32x2_t*>(smem + dst) = xg::u32x2_t{h[0], h[1]};
    *reinterpret_cast<xg::u32x2_t*>(smem + dst + (real ? G::kPlane : 16)) = xg::u32x2_t{m[0], m[1]};
    *reinterpret_cast<xg::u32x2_t*>(smem + dst + (real ? 2 * G::kPlane : 32)) = xg::u32x2_t{l[0], l[1]};
  };

  const int e0 = end_row<G>(0, total, rows), e1 = end_row<G>(1, total, rows);
#pragma unroll
  for (int i = 0; i < 2 * kItems; ++i) issue1(ld[0], 0, e0, i);
#pragma unroll
  for (int i = 0; i < 2 * kItems; ++i) issue1(ld[1], e0, e1, i);
  wait_set<2 * kItems>(ld[0]);
#pragma unroll
  for (int k = 0; k < kItems; ++k) { put_half(ld[0][k][0], k, 0, 0, e0); put_half(ld[0][k][1], k, 1, 0, e0); }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

  unsigned char* blk = smem + G::kRing + tile * 4096;        // partial sums [quad][lane], then outputs [pixel][128 B] swizzled
  unsigned char* part = blk + lane * 16;
  unsigned char* ara = smem + G::kRing + 4 * 4096 + tile * 8192;   // operand A pieces [piece][lane] x 16 bytes; B 4 KB further
  f32x16_t acc = acc0;
  auto tile_offset = [&](int r) -> unsigned {                // piece j adds 1 KB: 8 pixels x 128 bytes
    const unsigned P = (unsigned)(kRound * r + 32 * tile) + (unsigned)(lane >> 3);
    return P * 128u + (unsigned)(lane & 7) * 16u;
  };
  auto piece_ok = [&](int r, int j) -> bool { return (unsigned)(kRound * r + 32 * tile + 8 * j) + (unsigned)(lane >> 3) < (unsigned)total; };
  // first half: add the partner's partial sums, write [pixel][chunk ^ (pixel & 7)] rows back into the block
  auto finish = [&]() {
    if (kh == 0) {
      f32x4_t q4[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) q4[g] = *reinterpret_cast<const f32x4_t*>(part + g * 1024);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4_t v;
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = acc[4 * g + q] + q4[g][q];
        *reinterpret_cast<f32x4_t*>(blk + px * 128 + (((2 * g + kq) ^ (px & 7)) << 4)) = v;
      }
    }
  };
  auto out_piece = [&](int r, int j) {                       // 8 pixels = 1 KB of consecutive addresses, of round r
    const int pr = 8 * j + (lane >> 3);
    f32x4_t v = *reinterpret_cast<const f32x4_t*>(blk + pr * 128 + (((lane & 7) ^ (pr & 7)) << 4));
    if (has_a) {
      const f32x4_t a = *reinterpret_cast<const f32x4_t*>(ara + j * 1024 + lane * 16);
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = DG ? (a[q] > 0.f ? v[q] : 0.f) : v[q] + a[q];
    }
    if (has_m) {
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = (mb[j] >> q) & 1u ? v[q] : 0.f;
    }
    if (DG && has_b) {
      const f32x4_t b = *reinterpret_cast<const f32x4_t*>(ara + 4096 + j * 1024 + lane * 16);
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] += b[q];
    }
    if (!DG && p.out_relu) {
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = fmaxf(v[q], 0.f);
    }
    const unsigned off = piece_ok(r, j) ? tile_offset(r) + 1024u * j : kOut;
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), ov, off, 0, 0);
    if (emit) __builtin_amdgcn_raw_buffer_store_b8((unsigned char)sign_bits(v), ev, off == kOut ? kOut : off >> 4, 0, 0);
  };
  auto operands_request = [&](int r, int j) {                // piece j of round r's tile into the areas (LDS-DMA)
    typedef __attribute__((address_space(3))) void lds_void_t;
    const unsigned off = piece_ok(r, j) ? tile_offset(r) + 1024u * j : kOut;
    if (has_a) __builtin_amdgcn_raw_ptr_buffer_load_lds(av, (lds_void_t*)(ara + j * 1024), 16, off, 0, 0, 0);
    if (has_m) mb[j] = load_u8(md, off == kOut ? kOut : off >> 4);       // this lane's four channels: one byte
    if (DG && has_b) __builtin_amdgcn_raw_ptr_buffer_load_lds(bv, (lds_void_t*)(ara + 4096 + j * 1024), 16, off, 0, 0, 0);
  };

  auto round = [&](auto PH, int r, f32x4_t (&wr)[kItems][2], f32x4_t (&nx)[kItems][2]) {
    constexpr int ph = decltype(PH)::value;                  // the wave's memory phase: steps 4 ph .. 4 ph + 3
    const int lo1 = end_row<G>(r, total, rows), hi1 = end_row<G>(r + 1, total, rows), hi2 = end_row<G>(r + 2, total, rows);
    const int P = kRound * r + 32 * tile + px;
    const unsigned Pc = (unsigned)(P < total ? P : total - 1);
    const unsigned li = Pc / (unsigned)G::kPX, rem = Pc - li * G::kPX, y = rem / (unsigned)G::kW, x = rem - y * G::kW;
    const unsigned prow = G::kHP * li + y;                   // padded row of ky = 0
    const unsigned inrow = (unsigned)(2 * kh + kq) * G::kChunk + x * 16u;
    unsigned o[3][3];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      o[ky][0] = ((prow + ky) & (unsigned)(G::kR - 1)) * G::kRS + inrow;
      o[ky][1] = o[ky][0] + G::kPlane; o[ky][2] = o[ky][0] + 2 * G::kPlane;
      asm volatile("" : "+v"(o[ky][1]), "+v"(o[ky][2]));     // (a base per plane: the kx offsets are immediates)
    }
    bf16x8_t xb[3][3];
    auto fetch = [&](bf16x8_t (&xx)[3], int s) {             // step s: ky = s / 3, kx = s % 3 (padded pixel x + kx)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) xx[pl] = *reinterpret_cast<const bf16x8_t*>(smem + o[s / 3][pl] + (s % 3) * 16);
    };
    fetch(xb[0], 0);
    fetch(xb[1], 1);
    if (r > 0) finish();
    acc = acc0;
#define WSX_SB __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < 9; ++s) {
      WSX_SB
      if (s + 2 < 9) fetch(xb[(s + 2) % 3], s + 2);
      const bf16x8_t (&xx)[3] = xb[s % 3];
      const int j = s & 3;
      const bool mine = s < 8 && (s >> 2) == ph;
      // all vector-memory work of the round in four consecutive steps, behind one full wait: everything in the queue
      // is a round old by then (wdx.h)
      if (mine && j == 0) {
        wait_set<0>(wr);
        asm volatile("" : "+v"(mb[0]), "+v"(mb[1]), "+v"(mb[2]), "+v"(mb[3]));   // (requested a round ago, as the row items)
      }
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl[s], xx[0], acc, 0, 0, 0);
      if (mine) issue1(nx, hi1, hi2, j);
      WSX_SB
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[s], xx[2], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wm[s], xx[1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wm[s], xx[0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[s], xx[1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[s], xx[0], acc, 0, 0, 0);
      if (mine) {
        put_half(wr[j >> 1][j & 1], j >> 1, j & 1, lo1, hi1);
        if (ph == 0) {
          // piece j of the PREVIOUS round's outputs leaves (sums in the block since this round's start, operands
          // requested in this step of the previous round), then the same slots take this round's operands
          if (r > 0) out_piece(r - 1, j);
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          operands_request(r, j);
        }
      }
    }
    WSX_SB
#undef WSX_SB
    // two barriers: behind the first the block has been read out and may take the new partial sums; the second
    // publishes them together with the rows of the next round
    asm volatile("s_barrier" ::: "memory");
    if (kh == 1) {
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<f32x4_t*>(part + g * 1024) = f32x4_t{acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  };
  if (kh == 0) {
    for (int r = 0; r < rounds; r += 2) {
      round(std::integral_constant<int, 0>(), r, ld[1], ld[0]);
      if (r + 1 < rounds) round(std::integral_constant<int, 0>(), r + 1, ld[0], ld[1]);
    }
    // the last round's outputs (all four pieces read before the first store, tools/isa_store_hazard.py)
    finish();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(mb[0]), "+v"(mb[1]), "+v"(mb[2]), "+v"(mb[3]) :: "memory");
    f32x4_t v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int pr = 8 * j + (lane >> 3);
      v[j] = *reinterpret_cast<const f32x4_t*>(blk + pr * 128 + (((lane & 7) ^ (pr & 7)) << 4));
      if (has_a) {
        const f32x4_t a = *reinterpret_cast<const f32x4_t*>(ara + j * 1024 + lane * 16);
#pragma unroll
        for (int q = 0; q < 4; ++q) v[j][q] = DG ? (a[q] > 0.f ? v[j][q] : 0.f) : v[j][q] + a[q];
      }
      if (has_m) {
#pragma unroll
        for (int q = 0; q < 4; ++q) v[j][q] = (mb[j] >> q) & 1u ? v[j][q] : 0.f;
      }
      if (DG && has_b) {
        const f32x4_t b = *reinterpret_cast<const f32x4_t*>(ara + 4096 + j * 1024 + lane * 16);
#pragma unroll
        for (int q = 0; q < 4; ++q) v[j][q] += b[q];
      }
      if (!DG && p.out_relu) {
#pragma unroll
        for (int q = 0; q < 4; ++q) v[j][q] = fmaxf(v[j][q], 0.f);
      }
      asm volatile("" : "+v"(v[j]));
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const unsigned off = piece_ok(rounds - 1, j) ? tile_offset(rounds - 1) + 1024u * j : kOut;
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v[j]), ov, off, 0, 0);
      if (emit) __builtin_amdgcn_raw_buffer_store_b8((unsigned char)sign_bits(v[j]), ev, off == kOut ? kOut : off >> 4, 0, 0);
    }
  } else {
    for (int r = 0; r < rounds; r += 2) {
      round(std::integral_constant<int, 1>(), r, ld[1], ld[0]);
      if (r + 1 < rounds) round(std::integral_constant<int, 1>(), r + 1, ld[0], ld[1]);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// served: 3 x 3, stride 1, 'same', 32 -> 32 channels, dense rows, 18 x 24 or 9 x 12 maps
inline int geometry(const seedhip_conv_geom* g) {
  if (g->kh != 3 || g->kw != 3 || g->stride != 1 || g->pad_t != 1 || g->pad_l != 1 || g->cin != 32 || g->cout != 32 ||
      g->ld_in != 32 || g->ld_out != 32 || g->oh != g->ih || g->ow != g->iw)
    return 0;
  constexpr int min_img = 512;
  if (g->n_img < min_img) return 0;
  if (g->ih == 18 && g->iw == 24) return 1;
  if (g->ih == 9 && g->iw == 12) return 2;
  return 0;
}

template <typename G, bool DG>
inline int launch_one(Params& p, hipStream_t s) {
  static const int cus = xg::cu_count();
  p.per_wg = (p.n_img + cus - 1) / cus;
  const int grid = (p.n_img + p.per_wg - 1) / p.per_wg;
  static const bool ok = hipFuncSetAttribute((const void*)wsx_kernel<G, DG>, hipFuncAttributeMaxDynamicSharedMemorySize, G::kLds) == hipSuccess;
  if (!ok) return -1;
  hipLaunchKernelGGL((wsx_kernel<G, DG>), dim3(grid), dim3(512), G::kLds, s, p);
  return check_launch("wsx_kernel");
}

inline int launch(int geo, bool dg, Params& p, hipStream_t s) {
  if (geo == 1) return dg ? launch_one<Geo<18, 24>, true>(p, s) : launch_one<Geo<18, 24>, false>(p, s);
  if (geo == 2) return dg ? launch_one<Geo<9, 12>, true>(p, s) : launch_one<Geo<9, 12>, false>(p, s);
  return -1;
}

}  // namespace wsx
}  // namespace seedhip

// The 16 -> 16 channel 3 x 3 'same' convolutions of ImpalaDeep's first stack (dmlab/networks.py:26-60: the residual
// blocks on the 36 x 48 x 16 map behind the first pool), forward and data gradient, on the BF16 matrix pipe through the
// exact three-way operand split -- wsx.h's machine for 16 channels (halo_fwd_kernel<3, 1>: 515 us per call at 5 376
// images, eight calls per cfg3 step):
//   * 16 output channels fill a 16 x 16 x 32 MFMA's rows; its 32-deep reduction is TWO taps x 16 input channels (the
//     ninth tap pairs with zero weights: five steps), so a wave holds ALL the weights (60 registers): no reduction split,
//     no exchange between waves, one barrier per round;
//   * a wave owns two tiles of 16 consecutive pixels (a ROUND is 256 pixels of the run); an accumulator quad is four
//     consecutive channels of one pixel and the 16 lanes of a column group cover 16 consecutive pixels: a store
//     instruction writes 1 KB of consecutive addresses straight from the registers;
//   * padded input rows in a ring of 16 x 3 planes, the two 8-channel chunks of a row 64 slots apart (a multiple of 16:
//     the lane groups of ds_read_b128 mix the two chunks), 48 pixels per row = three whole 16-lane groups: conflict free;
//   * epilogue operands (residual / ReLU mask, skip add) are loaded into registers in the round's memory phase and meet
//     its accumulators in the NEXT round's phase, where the outputs leave; row items likewise use ONE register set (an
//     item is written to the ring and requested again at once): every load has a whole round to arrive.  All
//     vector-memory work of a wave sits in two consecutive steps behind one full wait, the two waves of a SIMD two
//     steps apart.
#pragma once
#include <type_traits>
#include "wsx.h"

namespace seedhip {
namespace wsy {

using xg::bf16x8_t;
using xg::f32x4_t;
using xg::u32x4_t;
using xg::sgpr128_t;

constexpr int kRound = 256;
constexpr int kItems = 2;                                    // 32-byte items per thread and round
constexpr unsigned kOut = 0x80000000u;

template <int H, int W>
struct Geo {
  static_assert(W % 16 == 0 && (H * W) % 16 == 0, "16-pixel tiles never cross an image row");
  static constexpr int kH = H, kW = W, kHP = H + 2, kWP = W + 2, kPX = H * W;
  static constexpr int kChunk = 64 * 16;                     // bytes between the two 8-channel chunks of a padded row
  static constexpr int kRS = 2 * kChunk;
  static constexpr int kR = 16;                              // ring rows (two rounds span 16)
  static constexpr int kPlane = kR * kRS;                    // 32 768
  static constexpr int kRing = 3 * kPlane;                   // 98 304
  static constexpr int kDump = kRing;
  static constexpr int kLds = kRing + 64;
  static constexpr int kRowItems = 2 * kWP;
  static_assert(kWP <= 64, "chunk pitch");
};

using wsx::Params;

template <typename G>
__device__ __forceinline__ int end_row(int r, int total, int rows) {
  int pl = kRound * r + kRound - 1; if (pl > total - 1) pl = total - 1;
  if (pl < 0) return 0;
  const unsigned li = (unsigned)pl / (unsigned)G::kPX, rem = (unsigned)pl - li * G::kPX;
  const int e = (int)(G::kHP * li + rem / (unsigned)G::kW + 3);
  return e < rows ? e : rows;
}

template <typename G, bool DG>
__global__ void __launch_bounds__(512, 2)
wsy_kernel(const Params p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, kq = lane >> 4, l4 = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int img0 = blockIdx.x * p.per_wg;
  int nimg = p.n_img - img0; if (nimg > p.per_wg) nimg = p.per_wg;
  if (nimg <= 0) return;
  const int total = nimg * G::kPX, rows = nimg * G::kHP;
  const int rounds = (total + kRound - 1) / kRound;
  const long long run_bytes = (long long)nimg * G::kPX * 64;
  const sgpr128_t xd = xg::make_view_words(p.X + (long long)img0 * G::kPX * 16, run_bytes);
  const sgpr128_t ad = xg::make_view_words((p.A ? p.A : p.Y) + (long long)img0 * G::kPX * 16, run_bytes);
  const sgpr128_t bd = xg::make_view_words((p.B ? p.B : p.Y) + (long long)img0 * G::kPX * 16, run_bytes);
  const __amdgpu_buffer_rsrc_t ov = gemm::make_view(p.Y + (long long)img0 * G::kPX * 16, run_bytes);
  const bool has_a = p.A != nullptr, has_b = DG && p.B != nullptr;
  const bool has_m = DG && p.mask_bits != nullptr, emit = !DG && p.out_bits != nullptr;   // ReLU masks as bytes: wsx.h
  const sgpr128_t md = xg::make_view_words(reinterpret_cast<const float*>((has_m ? p.mask_bits : (const unsigned char*)p.Y) + (long long)img0 * G::kPX * 4), has_m ? run_bytes >> 4 : 0);
  const __amdgpu_buffer_rsrc_t ev = gemm::make_view(reinterpret_cast<const float*>((emit ? p.out_bits : (unsigned char*)p.Y) + (long long)img0 * G::kPX * 4), emit ? run_bytes >> 4 : 0);
  unsigned mb[2] = {0u, 0u};

  // ---- weights: step s, lane (co = lane & 15, kq): tap t = 2 s + (kq >> 1), ci = 8 (kq & 1) + e; tap 9 = zeros ------ //
  bf16x8_t wh[5], wm[5], wl[5];
#pragma unroll
  for (int s = 0; s < 5; ++s) {
    const int t = 2 * s + (kq >> 1), tc = t < 9 ? t : 8;
    float v[8];
    if (DG) {                                                // w[2 - ky][2 - kx][co][ci]: eight consecutive ci
      const float* src = p.Wt + (((8 - tc) * 16 + l4) * 16 + 8 * (kq & 1));
      const f32x4_t v0 = *reinterpret_cast<const f32x4_t*>(src), v1 = *reinterpret_cast<const f32x4_t*>(src + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[e] = v0[e]; v[4 + e] = v1[e]; }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = p.Wt[(tc * 16 + 8 * (kq & 1) + e) * 16 + l4];
    }
    if (t >= 9) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = 0.f;
    }
    u32x4_t h, m, l;
#pragma unroll
    for (int e = 0; e < 4; ++e) { unsigned a, b, c; xg::split2(v[2 * e], v[2 * e + 1], a, b, c); h[e] = a; m[e] = b; l[e] = c; }
    wh[s] = __builtin_bit_cast(bf16x8_t, h); wm[s] = __builtin_bit_cast(bf16x8_t, m); wl[s] = __builtin_bit_cast(bf16x8_t, l);
  }
  f32x4_t acc0;                                              // rows co = 4 kq + q of the 16 x 16 result
#pragma unroll
  for (int q = 0; q < 4; ++q) acc0[q] = (!DG && p.bias) ? p.bias[4 * kq + q] : 0.f;

  // ---- staging: item q of padded rows [lo, hi) = chunk c of padded pixel pc of row lo + q / kRowItems -------------- //
  f32x4_t ld[1][kItems][2];
  auto item_src = [&](int k, int lo, int hi) -> unsigned {
    const unsigned q = (unsigned)tid + 512u * k;
    const unsigned rr = q / (unsigned)G::kRowItems, rem = q - rr * G::kRowItems, pc = rem >> 1, c = rem & 1u;
    const unsigned prow = (unsigned)lo + rr, li = prow / (unsigned)G::kHP, r1 = prow - li * G::kHP;
    const bool in = prow < (unsigned)hi && r1 - 1u < (unsigned)G::kH && pc - 1u < (unsigned)G::kW;
    return in ? (((li * G::kH + r1 - 1u) * G::kW + pc - 1u) * 16u + 8u * c) * 4u : kOut;
  };
  auto issue1 = [&](f32x4_t (&s)[kItems][2], int lo, int hi, int i) {
    const unsigned voff = item_src(i >> 1, lo, hi);
    if (i & 1) s[i >> 1][1] = wfx::load16b(xd, voff, 0u); else s[i >> 1][0] = wfx::load16(xd, voff, 0u);
  };
  auto put_half = [&](const f32x4_t& it_in_flight, int k, int j, int lo, int hi) {
    const f32x4_t it = xg::move_item(it_in_flight);          // (behind the round's one wait; no other read of the item's registers)
    const unsigned q = (unsigned)tid + 512u * k;
    const unsigned rr = q / (unsigned)G::kRowItems, rem = q - rr * G::kRowItems, pc = rem >> 1, c = rem & 1u;
    const unsigned prow = (unsigned)lo + rr;
    const bool real = prow < (unsigned)hi;
    const unsigned dst = real ? (prow & (unsigned)(G::kR - 1)) * G::kRS + c * G::kChunk + pc * 16u + 8u * j : (unsigned)(G::kDump + 8 * j);
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
    *reinterpret_cast<xg::u32x2_t*>(smem + dst) = xg::u32x2_t{h[0], h[1]};
    *reinterpret_cast<xg::u32x2_t*>(smem + dst + (real ? G::kPlane : 16)) = xg::u32x2_t{m[0], m[1]};
    *reinterpret_cast<xg::u32x2_t*>(smem + dst + (real ? 2 * G::kPlane : 32)) = xg::u32x2_t{l[0], l[1]};
  };

  // ONE register set for the row items: an item is written to the ring and its registers are requested again at once,
  // for the rows of the round after next -- a whole round in flight, consumed in the same step of the next round
  const int e0 = end_row<G>(0, total, rows), e1 = end_row<G>(1, total, rows);
  f32x4_t (&set)[kItems][2] = ld[0];
#pragma unroll
  for (int i = 0; i < 2 * kItems; ++i) issue1(set, 0, e0, i);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // (no operands: put_half moves the items out, xg::move_item)
#pragma unroll
  for (int k = 0; k < kItems; ++k) { put_half(set[k][0], k, 0, 0, e0); put_half(set[k][1], k, 1, 0, e0); }
#pragma unroll
  for (int i = 0; i < 2 * kItems; ++i) issue1(set, e0, e1, i);

  // byte offset of this lane's 16 output bytes of tile u of round r (pixel 256 r + 32 wave + 16 u + (lane & 15),
  // channels 4 kq ..), or out of range
  auto out_offset = [&](int r, int u) -> unsigned {
    const unsigned P = (unsigned)(kRound * r + 32 * wave + 16 * u + l4);
    return P < (unsigned)total ? P * 64u + (unsigned)kq * 16u : kOut;
  };
  // The outputs of round r leave in round r + 1's memory phase: their epilogue operands (residual / mask, add) were
  // requested into `oa`, `ob` in round r's phase -- a round in flight, one register set -- and the accumulators wait in
  // `prev`.
  f32x4_t oa[2], ob[2], prev[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) { oa[u] = f32x4_t{0.f, 0.f, 0.f, 0.f}; ob[u] = oa[u]; prev[u] = oa[u]; }
  auto out_prev = [&](int r) {                               // outputs of round r from prev, oa, ob
    f32x4_t v[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      v[u] = prev[u];
      if (has_a) {
#pragma unroll
        for (int q = 0; q < 4; ++q) v[u][q] = DG ? (oa[u][q] > 0.f ? v[u][q] : 0.f) : v[u][q] + oa[u][q];
      }
      if (has_m) {
#pragma unroll
        for (int q = 0; q < 4; ++q) v[u][q] = (mb[u] >> q) & 1u ? v[u][q] : 0.f;
      }
      if (has_b) {
#pragma unroll
        for (int q = 0; q < 4; ++q) v[u][q] += ob[u][q];
      }
      if (!DG && p.out_relu) {
#pragma unroll
        for (int q = 0; q < 4; ++q) v[u][q] = fmaxf(v[u][q], 0.f);
      }
      asm volatile("" : "+v"(v[u]));
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const unsigned off = out_offset(r, u);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v[u]), ov, off, 0, 0);
      if (emit) __builtin_amdgcn_raw_buffer_store_b8((unsigned char)wsx::sign_bits(v[u]), ev, off == kOut ? kOut : off >> 4, 0, 0);
    }
  };
  auto operands = [&](int r) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const unsigned off = out_offset(r, u);
      if (has_a) oa[u] = wfx::load16(ad, off, 0u);
      if (has_m) mb[u] = wsx::load_u8(md, off == kOut ? kOut : off >> 4);
      if (has_b) ob[u] = wfx::load16(bd, off, 0u);
    }
  };
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

  auto round = [&](auto PH, int r) {
    constexpr int ph = decltype(PH)::value;                  // the wave's memory phase: steps 2 ph, 2 ph + 1
    const int lo1 = end_row<G>(r, total, rows), hi1 = end_row<G>(r + 1, total, rows), hi2 = end_row<G>(r + 2, total, rows);
    // per tile and step: the lane's tap (two taps per step, the upper half of the lanes takes the second)
    unsigned o01[2][5];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int P = kRound * r + 32 * wave + 16 * u + l4;
      const unsigned Pc = (unsigned)(P < total ? P : total - 1);
      const unsigned li = Pc / (unsigned)G::kPX, rem = Pc - li * G::kPX, y = rem / (unsigned)G::kW, x = rem - y * G::kW;
      const unsigned prow = G::kHP * li + y;
#pragma unroll
      for (int s = 0; s < 5; ++s) {
        const int ta = 2 * s, tb = 2 * s + 1 < 9 ? 2 * s + 1 : 8;
        const unsigned oa_ = ((prow + ta / 3) & (unsigned)(G::kR - 1)) * G::kRS + (x + ta % 3) * 16u;
        const unsigned ob_ = ((prow + tb / 3) & (unsigned)(G::kR - 1)) * G::kRS + (x + tb % 3) * 16u;
        o01[u][s] = ((kq >> 1) ? ob_ : oa_) + (unsigned)(kq & 1) * G::kChunk;
      }
    }
    f32x4_t acc[2] = {acc0, acc0};
    bf16x8_t xb[2][2][3];
    auto fetch = [&](bf16x8_t (&xx)[2][3], int s) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        xx[u][0] = *reinterpret_cast<const bf16x8_t*>(smem + o01[u][s]);
        xx[u][1] = *reinterpret_cast<const bf16x8_t*>(smem + o01[u][s] + G::kPlane);
        xx[u][2] = *reinterpret_cast<const bf16x8_t*>(smem + (o01[u][s] + 2 * G::kPlane));
      }
    };
    fetch(xb[0], 0);
#define WSY_SB __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < 5; ++s) {
      WSY_SB
      if (s + 1 < 5) fetch(xb[(s + 1) & 1], s + 1);
      const bf16x8_t (&xx)[2][3] = xb[s & 1];
      const bool mine = (s >> 1) == ph;                      // steps 0-1 or 2-3
      const int j = s & 1;
      // all vector-memory work of the round in two consecutive steps behind one full wait: everything in the queue is
      // a round old by then (the rows of round r + 1, the previous round's operands and stores)
      if (mine && j == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("" : "+v"(oa[0]), "+v"(oa[1]), "+v"(ob[0]), "+v"(ob[1]), "+v"(mb[0]), "+v"(mb[1]));
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[s], xx[u][0], acc[u], 0, 0, 0);
      if (mine && j == 0) {
        if (r > 0) out_prev(r - 1);
        operands(r);
      }
      WSY_SB
#pragma unroll
      for (int u = 0; u < 2; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[s], xx[u][2], acc[u], 0, 0, 0);
#pragma unroll
      for (int u = 0; u < 2; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm[s], xx[u][1], acc[u], 0, 0, 0);
#pragma unroll
      for (int u = 0; u < 2; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm[s], xx[u][0], acc[u], 0, 0, 0);
#pragma unroll
      for (int u = 0; u < 2; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[s], xx[u][1], acc[u], 0, 0, 0);
#pragma unroll
      for (int u = 0; u < 2; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[s], xx[u][0], acc[u], 0, 0, 0);
      if (mine) {
        put_half(set[j][0], j, 0, lo1, hi1); put_half(set[j][1], j, 1, lo1, hi1);
        issue1(set, hi1, hi2, 2 * j); issue1(set, hi1, hi2, 2 * j + 1);
      }
    }
    WSY_SB
#undef WSY_SB
    prev[0] = acc[0]; prev[1] = acc[1];
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  };
  if (wave < 4) {
    for (int r = 0; r < rounds; ++r) round(std::integral_constant<int, 0>(), r);
  } else {
    for (int r = 0; r < rounds; ++r) round(std::integral_constant<int, 1>(), r);
  }
  // the last round's outputs
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(oa[0]), "+v"(oa[1]), "+v"(ob[0]), "+v"(ob[1]), "+v"(mb[0]), "+v"(mb[1]));
  out_prev(rounds - 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// served: 3 x 3, stride 1, 'same', 16 -> 16 channels, dense rows, 36 x 48 maps
inline int geometry(const seedhip_conv_geom* g) {
  if (g->kh != 3 || g->kw != 3 || g->stride != 1 || g->pad_t != 1 || g->pad_l != 1 || g->cin != 16 || g->cout != 16 ||
      g->ld_in != 16 || g->ld_out != 16 || g->oh != g->ih || g->ow != g->iw)
    return 0;
  constexpr int min_img = 256;
  if (g->n_img < min_img) return 0;
  return (g->ih == 36 && g->iw == 48) ? 1 : 0;
}

template <typename G, bool DG>
inline int launch_one16(Params& p, hipStream_t s) {
  static const int cus = xg::cu_count();
  p.per_wg = (p.n_img + cus - 1) / cus;
  const int grid = (p.n_img + p.per_wg - 1) / p.per_wg;
  static const bool ok = hipFuncSetAttribute((const void*)wsy_kernel<G, DG>, hipFuncAttributeMaxDynamicSharedMemorySize, G::kLds) == hipSuccess;
  if (!ok) return -1;
  hipLaunchKernelGGL((wsy_kernel<G, DG>), dim3(grid), dim3(512), G::kLds, s, p);
  return check_launch("wsy_kernel");
}

inline int launch(int geo, bool dg, Params& p, hipStream_t s) {
  if (geo == 1) return dg ? launch_one16<Geo<36, 48>, true>(p, s) : launch_one16<Geo<36, 48>, false>(p, s);
  return -1;
}

}  // namespace wsy
}  // namespace seedhip

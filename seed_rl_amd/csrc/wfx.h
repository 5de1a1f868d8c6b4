// The second Atari conv's forward -- Conv2D(32, 4, 2) on the 20 x 20 x 16 output of the first
// (/root/reference/atari/networks.py:236) -- on the BF16 matrix pipe through the exact three-way operand split of
// xgemm.h ("bf16x6": fp32 = h + m + l, six of the nine plane products; same arithmetic, same error bound).
//
// Why: wfw.h's fp32-MFMA kernel needs 140 us for 11.2 GFLOP (80 TF/s of the fp32 pipe's 157; its 128 MFMAs per round
// already run back to back), and v_mfma_f32_32x32x16_bf16 does the six products of one (32 pixel x 32 channel x 16 k)
// block in 192 cycles against 1024 for the fp32 instruction's 128 passes -- 27 us of matrix time for the layer.
//
// Structure: one 4-wave workgroup per CU (141.7 KB of LDS), a persistent run of images per workgroup.
//   * WEIGHTS: lane (co = lane & 31, kq = lane >> 5) holds W[tap][ci = 8 kq .. 8 kq + 7][co] of all 16 taps as three
//     bf16 planes -- 192 registers, split once in the prologue; they are the MFMA's row operand, so a lane ends up with
//     four consecutive output channels of one pixel per accumulator quad (16-byte stores, bias and ReLU fused);
//   * INPUT: the run is a contiguous array of image rows (20 per image, 1 280 bytes each).  Every row is loaded ONCE
//     (32-byte items, coalesced), split in registers and written as three bf16 planes into a ring of 72 rows; rows of
//     one parity sit `kPitch` = 41 sixteen-byte slots apart, 41 = 9 (mod 16): a step to the next output row (two input
//     rows down, nine pixels on) continues the slot sequence of the previous one, so the 16 lanes of a ds_read_b128
//     phase -- 16 consecutive pixels of the run, any tap -- hit 16 different slots except across an image boundary
//     (average conflict factor 1.19, against 2.0 for a plain [row][ix][ci] layout);
//   * a ROUND is 128 consecutive pixels of the run (32 per wave, tiles cross image boundaries): 16 taps x 6 MFMAs per
//     wave, the pixel operand of tap t + 1 read while tap t multiplies.  The rows of round r + 1 are split and written
//     between the taps of round r (from registers loaded during round r - 1), the loads for round r + 2 are issued at
//     its start: two register sets, the loads are asm statements with counted waits (the compiler's own bookkeeping
//     would drain the queue at the loop head, see xgemm.h); one `s_barrier` per round, without `vmcnt(0)`.
// Ring safety: rounds r and r + 1 together span at most 35 rows of one parity (enumerated in
// tests/test_wfx_layout.py); kRU = 36.
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
constexpr int kItems = 6;                                    // 32-byte items per thread and round (<= 38 rows)
constexpr unsigned kOut = 0x80000000u;

struct Params {
  const float* X; const float* W; const float* bias; float* Y;
  int n_img, per_wg, in_relu, out_relu;
  unsigned long long* trace;              // SEEDHIP_WFX_TRACE: [workgroup][wave][32 rounds][8 stamps]
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
  static_assert(kItems == 6, "operand list below");
  asm volatile("s_waitcnt vmcnt(%12)"
               : "+v"(r[0][0]), "+v"(r[0][1]), "+v"(r[1][0]), "+v"(r[1][1]), "+v"(r[2][0]), "+v"(r[2][1]),
                 "+v"(r[3][0]), "+v"(r[3][1]), "+v"(r[4][0]), "+v"(r[4][1]), "+v"(r[5][0]), "+v"(r[5][1]) : "n"(N));
}

// rows [0, end_row(r)) of the run are what rounds 0..r read
__device__ __forceinline__ int end_row(int r, int total, int rows) {
  int pl = kRound * r + kRound - 1; if (pl > total - 1) pl = total - 1;
  if (pl < 0) return 0;
  const unsigned li = (unsigned)pl / (unsigned)kP, pix = (unsigned)pl - li * kP;
  const int e = (int)(kIH * li + 2 * (pix / (unsigned)kOW) + 4);
  return e < rows ? e : rows;
}

template <bool TRACE, bool RELU_IN>
__global__ void __launch_bounds__(256, 1)
wfx_kernel(const Params p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, px = lane & 31, kq = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int img0 = blockIdx.x * p.per_wg;
  int nimg = p.n_img - img0; if (nimg > p.per_wg) nimg = p.per_wg;
  if (nimg <= 0) return;
  const int total = nimg * kP, rows = nimg * kIH;
  const int rounds = (total + kRound - 1) / kRound;
  const sgpr128_t xd = xg::make_view_words(p.X + (long long)img0 * (kIH * kIW * 16), (long long)rows * kRowBytes);

  // ---- weights: three planes of W[t][8 kq + e][co = px], e = 0..7 ------------------------------------------------ //
  bf16x8_t wh[16], wm[16], wl[16];
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = p.W[(t * 16 + 8 * kq + e) * 32 + px];
    u32x4_t h, m, l;
#pragma unroll
    for (int e = 0; e < 4; ++e) { unsigned a, b, c; xg::split2(v[2 * e], v[2 * e + 1], a, b, c); h[e] = a; m[e] = b; l[e] = c; }
    wh[t] = __builtin_bit_cast(bf16x8_t, h); wm[t] = __builtin_bit_cast(bf16x8_t, m); wl[t] = __builtin_bit_cast(bf16x8_t, l);
  }
  f32x16_t bias16;
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int q = 0; q < 4; ++q) bias16[4 * g + q] = p.bias ? p.bias[8 * g + 4 * kq + q] : 0.f;

  // ---- staging ------------------------------------------------------------------------------------------------- //
  f32x4_t ld[2][kItems][2];
  auto issue = [&](f32x4_t (&s)[kItems][2], int lo, int hi, int k0 = 0, int k1 = kItems) {   // rows [lo, hi) of the run -> registers
    const unsigned n = (unsigned)(hi - lo) * 40u;
    const unsigned soff = __builtin_amdgcn_readfirstlane((unsigned)lo * (unsigned)kRowBytes);
#pragma unroll
    for (int k = k0; k < k1; ++k) {
      const unsigned q = (unsigned)tid + 256u * k;
      const unsigned voff = q < n ? q * 32u : kOut;
      s[k][0] = load16(xd, voff, soff);
      s[k][1] = load16b(xd, voff, soff);
    }
  };
  // Item k of rows [lo, hi) goes to LDS in two halves (four values -> 8 bytes per plane), each in three pieces that the
  // round places into the gaps between its MFMAs: address, two pair splits, the writes.  Branch free: items past the
  // rows were loaded as zeros and go to a pad slot nobody reads.
  auto put_addr = [&](int k, int lo, int hi) -> unsigned {
    const unsigned q = (unsigned)tid + 256u * k;
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
    xg::split2_trunc(f0, f1, h, m, l);
  };
  auto put_write = [&](unsigned dst, const unsigned (&h)[2], const unsigned (&m)[2], const unsigned (&l)[2]) {
    *reinterpret_cast<xg::u32x2_t*>(smem + dst) = xg::u32x2_t{h[0], h[1]};
    *reinterpret_cast<xg::u32x2_t*>(smem + dst + kPlane) = xg::u32x2_t{m[0], m[1]};
    *reinterpret_cast<xg::u32x2_t*>(smem + dst + 2 * kPlane) = xg::u32x2_t{l[0], l[1]};
  };
  auto put_half = [&](const f32x4_t& it, int k, int j, int lo, int hi) {
    unsigned h[2], m[2], l[2];
    put_pair(it[0], it[1], h[0], m[0], l[0]); put_pair(it[2], it[3], h[1], m[1], l[1]);
    put_write(put_addr(k, lo, hi) + 8 * j, h, m, l);
  };
  auto put = [&](const f32x4_t (&it)[2], int k, int lo, int hi) { put_half(it[0], k, 0, lo, hi); put_half(it[1], k, 1, lo, hi); };

  unsigned long long* tl = reinterpret_cast<unsigned long long*>(smem + kLds) + wave * 256;   // (TRACE builds: 8 KB more)
  auto stamp = [&](int r, int k) {
    if (TRACE && r < 32) { const unsigned long long c = __builtin_amdgcn_s_memtime(); if (lane == 0) tl[r * 8 + k] = c; }
  };
  const int e0 = end_row(0, total, rows), e1 = end_row(1, total, rows);
  issue(ld[0], 0, e0);
  issue(ld[1], e0, e1);
  wait_set<2 * kItems>(ld[0]);
#pragma unroll
  for (int k = 0; k < kItems; ++k) put(ld[0][k], k, 0, e0);
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

  const long long ybase = (long long)img0 * kP * 32;
  // one round: reads rows of round r, writes the rows of r + 1 from `wr`, requests the rows of r + 2 into `nx`
  // outputs of the previous round: stored behind this round's wait for `wr` (see there)
  f32x16_t pv = bias16; float* po = nullptr; bool plive = false;
  auto flush = [&]() {
    if (plive) {
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<f32x4_t*>(po + 8 * g) = f32x4_t{pv[4 * g], pv[4 * g + 1], pv[4 * g + 2], pv[4 * g + 3]};
    }
  };
  auto round = [&](int r, f32x4_t (&wr)[kItems][2], f32x4_t (&nx)[kItems][2]) {
    stamp(r, 0);
    const int lo1 = end_row(r, total, rows), hi1 = end_row(r + 1, total, rows), hi2 = end_row(r + 2, total, rows);
    const int P = kRound * r + 32 * wave + px;
    const bool live = P < total;
    const unsigned Pc = (unsigned)(live ? P : total - 1);
    const unsigned li = Pc / (unsigned)kP, pix = Pc - li * kP, oy = pix / (unsigned)kOW, ox = pix - oy * kOW;
    const unsigned u0 = 10u * li + oy;
    const unsigned inrow = ox * 16u + (unsigned)kq * 320u;
    const unsigned char* b0 = smem + (u0 % (unsigned)kRU) * kPitch + inrow;
    const unsigned char* b1 = smem + ((u0 + 1u) % (unsigned)kRU) * kPitch + inrow;
    f32x16_t acc = bias16;
    bf16x8_t xb[3][3];
    auto fetch = [&](bf16x8_t (&x)[3], int t) {
      const int ky = t >> 2, kx = t & 3;
      const unsigned char* b = ((ky >> 1) ? b1 : b0) + (ky & 1) * kPar + (kx & 1) * 160 + (kx >> 1) * 16;
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) x[pl] = *reinterpret_cast<const bf16x8_t*>(b + pl * kPlane);
    };
    fetch(xb[0], 0);                                         // the LDS round trip is longer than a tap's six MFMAs:
    fetch(xb[1], 1);                                         // operands are requested two taps ahead (three buffers)
    __builtin_amdgcn_sched_barrier(0);
    // `wr` was requested over the first taps of the previous round, the outputs stored before that: everything in the
    // queue is at least 13 taps old.  (No counted wait: loads and stores do not retire in order with each other.)
    wait_set<0>(wr);
    stamp(r, 1);
    flush();
    stamp(r, 2);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      if (t + 2 < 16) fetch(xb[(t + 2) % 3], t + 2);
      const bf16x8_t (&x)[3] = xb[t % 3];
      // six MFMAs, the pieces of one half item pinned into the gaps between them (one wave per SIMD: nothing else
      // would overlap the split's VALU work with the matrix pipe)
      const bool has = t >= 2 && t < 2 + 2 * kItems;
      const int pk = has ? (t - 2) >> 1 : 0, pj = (t - 2) & 1;
      const f32x4_t& it = wr[pk][pj];
      unsigned dst = 0, sh[2], sm[2], sl[2];
#define WFX_SB __builtin_amdgcn_sched_barrier(0);
      WFX_SB
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl[t], x[0], acc, 0, 0, 0);
      if (has) dst = put_addr(pk, lo1, hi1) + 8 * pj;
      WFX_SB
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[t], x[2], acc, 0, 0, 0);
      if (has) put_pair(it[0], it[1], sh[0], sm[0], sl[0]);
      WFX_SB
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wm[t], x[1], acc, 0, 0, 0);
      WFX_SB
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wm[t], x[0], acc, 0, 0, 0);
      if (has) put_pair(it[2], it[3], sh[1], sm[1], sl[1]);
      WFX_SB
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[t], x[1], acc, 0, 0, 0);
      WFX_SB
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[t], x[0], acc, 0, 0, 0);
      if (has) put_write(dst, sh, sm, sl);
      WFX_SB
#undef WFX_SB
      if (t < 3) issue(nx, hi1, hi2, 2 * t, 2 * t + 2);        // 48 KB per CU and round: not in one burst behind the barrier
      if (t == 0) stamp(r, 3);
      if (t == 1) stamp(r, 4);
      if (t == 1 + 2 * kItems) stamp(r, 5);
    }
    stamp(r, 6);
    po = p.Y + ybase + (long long)P * 32 + 4 * kq; plive = live;
#pragma unroll
    for (int q = 0; q < 16; ++q) pv[q] = p.out_relu ? __builtin_amdgcn_fmed3f(acc[q], 0.f, __builtin_inff()) : acc[q];
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    stamp(r, 7);
  };
  for (int r = 0; r < rounds; r += 2) {
    round(r, ld[1], ld[0]);
    if (r + 1 < rounds) round(r + 1, ld[0], ld[1]);
  }
  flush();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (TRACE && p.trace && lane < 32) {
    for (int k = lane; k < 256; k += 32) p.trace[((long long)blockIdx.x * 4 + wave) * 256 + k] = tl[k];
  }
}

inline bool plan(Params& p, const seedhip_conv_geom* g) {
  if (g->pad_t || g->pad_l || g->kh != 4 || g->kw != 4 || g->stride != 2 || g->cin != 16 || g->cout != 32 || g->ld_in != 16 ||
      g->ld_out != 32 || g->ih != kIH || g->iw != kIW || g->oh != kOW || g->ow != kOW)
    return false;
  if (g->n_img < 2048) return false;
  memset(&p, 0, sizeof(p));
  p.n_img = g->n_img;
  return true;
}

inline int launch(Params& p, hipStream_t s) {
  static const int cus = xg::cu_count();
  p.per_wg = (p.n_img + cus - 1) / cus;
  const int grid = (p.n_img + p.per_wg - 1) / p.per_wg;
  static const int trace = xg::env_int("SEEDHIP_WFX_TRACE", 0);
  if (trace && !p.in_relu) {                                               // per-round cycle stamps of workgroup 0 and one in the middle, to stderr
    static unsigned long long* buf = nullptr;
    const size_t n = (size_t)grid * 4 * 256;
    if (!buf && hipMalloc(&buf, (size_t)1024 * 4 * 256 * 8) != hipSuccess) return -1;
    if (hipFuncSetAttribute((const void*)wfx_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds + 8192) != hipSuccess) return -1;
    (void)hipMemsetAsync(buf, 0, n * 8, s);
    p.trace = buf;
    hipLaunchKernelGGL((wfx_kernel<true, false>), dim3(grid), dim3(256), kLds + 8192, s, p);
    std::vector<unsigned long long> h(n);
    (void)hipStreamSynchronize(s);
    (void)hipMemcpy(h.data(), buf, n * 8, hipMemcpyDeviceToHost);
    static int shown = 0;
    if (shown++ < 2) {
      for (int wg : {0, grid / 2}) {
        for (int w = 0; w < 4; w += 3) {
          const unsigned long long* t = h.data() + ((size_t)wg * 4 + w) * 256;
          fprintf(stderr, "wfx trace wg %d wave %d: round | wait | flush+addr | tap 0 | tap 1 | taps 2-13 | taps 14-15 | relu+barrier | (s_memtime ticks)\n", wg, w);
          for (int r = 0; r < 24 && t[r * 8]; ++r)
            fprintf(stderr, "  %2d | %6llu %6llu %6llu %6llu %6llu %6llu %6llu | round %6llu\n", r, t[r * 8 + 1] - t[r * 8], t[r * 8 + 2] - t[r * 8 + 1],
                    t[r * 8 + 3] - t[r * 8 + 2], t[r * 8 + 4] - t[r * 8 + 3], t[r * 8 + 5] - t[r * 8 + 4], t[r * 8 + 6] - t[r * 8 + 5],
                    t[r * 8 + 7] - t[r * 8 + 6], t[r * 8 + 7] - t[r * 8]);
        }
      }
    }
    return check_launch("wfx_kernel(trace)");
  }
  if (p.in_relu) {
    static const bool ok = hipFuncSetAttribute((const void*)wfx_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds) == hipSuccess;
    if (!ok) return -1;
    hipLaunchKernelGGL((wfx_kernel<false, true>), dim3(grid), dim3(256), kLds, s, p);
  } else {
    static const bool ok = hipFuncSetAttribute((const void*)wfx_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds) == hipSuccess;
    if (!ok) return -1;
    hipLaunchKernelGGL((wfx_kernel<false, false>), dim3(grid), dim3(256), kLds, s, p);
  }
  return check_launch("wfx_kernel");
}

}  // namespace wfx
}  // namespace seedhip

// Weight gradients of the small convolutions on the BF16 matrix pipe through the exact three-way operand split of
// xgemm.h ("bf16x6": fp32 = h + m + l, six of the nine plane products, fp32 accumulation) -- the second Atari conv
// (/root/reference/atari/networks.py:236, Conv2D(32, 4, 2) on 20 x 20 x 16) and ImpalaDeep's 3 x 3 'same' layers
// (/root/reference/dmlab/networks.py:26-60):
//     dW[(ky, kx, ci), co] = sum over images and output pixels of X[pixel; ky, kx, ci] * dY[pixel, co].
//
// Why it was the last GEMM class on the fp32 pipe (wsw.h / halo_wgrad.h): the reduction index is the PIXEL, and in NHWC
// the pixel is the strided index of both operands -- an MFMA lane wants eight consecutive k of one row.  CDNA4's
// transposing LDS read does that for free: ds_read_b64_tr_b16 hands lane c of a 16-lane group the four k-values of
// column c of a [4 k][16 columns] block whose four rows are addressed individually (lane 4 j + q of the group supplies
// the address of row j, columns 4 q .. 4 q + 3).  "Columns" are channels (contiguous in NHWC), "rows" are pixels at ANY
// address: stride, tap offset and zero padding of the im2col matrix are just per-lane addresses, nothing is gathered.
//
// Structure (256 threads = 4 waves, two workgroups per CU, each a persistent run of UNITS; unit = BR output rows of one
// image -- the whole image where it fits).  Measured (MI355X, r5, rocprofv3 clock): second Atari conv 100.5 us against
// wsw.h's 143, ImpalaDeep 16->16 @36x48 288 (halo_wgrad.h: 525), 16->32 @36x48 475 (820), 32->32 @18x24 220 (427),
// 32->32 @9x12 66 (127); cfg2 step 1.036 -> 1.013 ms, cfg3 16.09 -> 14.30 ms on one box.
//   * a unit's input rows ((BR - 1) S + KH of them; rows outside a 'same'-padded image are requested out of range and
//     come back as zeros) and its dY rows are contiguous in HBM: 16-byte items, coalesced, every byte once per unit;
//     loaded into registers ONE UNIT AHEAD, each item requested again the moment its registers are consumed (the
//     requests fly under the rest of the split and the whole MFMA phase of the current unit);
//   * each element is split ONCE (by truncation: plain full-rate VALU, xgemm.h split2_trunc) on its way into LDS:
//     three bf16 planes of X as [row][x + pad][ci] (pad columns zeroed once) and of dY as [pixel][co] (+ zero pixels up to
//     a k-step multiple); one ds_write_b64 per plane and item;
//   * MFMA phase: wave (wm, wk) owns the tiles of tap group wm (accumulators in registers for the whole launch) and
//     every WK-th k-step; a k-step is 16 pixels (v_mfma_f32_32x32x16_bf16, 32 output channels: tile = 2 taps x 16 or
//     1 tap x 32 input channels) or 32 pixels (v_mfma_f32_16x16x32_bf16, 16 output channels: tile = 1 tap x 16);
//     per k-step 6 transposing reads for dY's planes, per tile 6 for X's and 6 MFMAs (products al bh, ah bl, am bm,
//     am bh, ah bm, ah bh: small terms first).  The per-lane pixel addresses of a unit's k-steps are unit-independent
//     and live in registers; tap and plane offsets are instruction immediates;
//   * pixels past the unit's last one (k-step padding) read dY's zero pixels, and X at the unit's LAST real pixel: a
//     non-finite X there is already part of the true sum of that row of dW, so non-finite inputs give non-finite
//     outputs in exactly the rows where the fp32 evaluation has them;
//   * two barriers per unit (planes written / planes consumed); the two workgroups of a CU run out of phase, one's split
//     phase under the other's MFMAs;
//   * epilogue: the waves' partial sums are added in LDS in a fixed order, one partial slice per workgroup + the deterministic
//     second-pass reduction the other weight gradients use; the bias gradient is the loader threads' running sums of
//     the dY items they split anyway.
// Bank conflicts: a transposing read is served in two 32-lane halves; the k-index -> pixel map (pix_of) makes a half
// cover 8 consecutive pixels x 32 bytes (16 channels) or 4 pixels x 64 bytes: 256 consecutive bytes except at row ends.
// Compiled as its own translation unit (wgx.hip); conv.hip sees wgx_api.h only.
#pragma once
#include "common.h"
#include "xgemm.h"
#include "wgx_api.h"

namespace seedhip {
namespace wgx {

using xg::bf16x8_t;
using xg::f32x4_t;
using xg::f32x16_t;
using xg::u32x2_t;
typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef short s16x8_t __attribute__((ext_vector_type(8)));
constexpr unsigned kOut = 0x80000000u;

template <int KH_, int KW_, int S_, int PAD_, int CIN_, int COUT_, int IH_, int IW_, int OH_, int OW_, int BR_, int WM_, int WK_>
struct Geo {
  static constexpr int KH = KH_, KW = KW_, S = S_, PAD = PAD_, CIN = CIN_, COUT = COUT_, IH = IH_, IW = IW_, OH = OH_,
                       OW = OW_, BR = BR_, WM = WM_, WK = WK_;
  static constexpr bool M32 = COUT == 32;                     // 32x32x16 MFMAs (k-step 16 pixels) / 16x16x32 (32 pixels)
  static constexpr int KS = M32 ? 16 : 32;
  static constexpr int TROWS = M32 ? 32 : 16;                 // dW rows of one accumulator tile
  static constexpr int TPT = TROWS / CIN;                     // taps per tile
  static constexpr int NTAPS = KH * KW, TILES = (NTAPS + TPT - 1) / TPT, M = NTAPS * CIN;
  static constexpr int TPW = (TILES + WM - 1) / WM;           // tiles per wave
  static constexpr int ACCN = M32 ? 16 : 4;
  static constexpr int NB = OH / BR;                          // units per image
  static constexpr int XR = (BR - 1) * S + KH;                // input rows of a unit
  static constexpr int RP = IW + 2 * PAD;                     // pixels of a padded LDS row
  static constexpr int PIXB = CIN * 2;                        // bytes of a pixel in one plane
  static constexpr int XPL = XR * RP * PIXB;                  // one plane of X
  static constexpr int NPIX = BR * OW, KSTEPS = (NPIX + KS - 1) / KS, KPW = (KSTEPS + WK - 1) / WK;
  static constexpr int YPIXB = COUT * 2;
  static constexpr int YPL = KSTEPS * KS * YPIXB;             // one plane of dY incl. the zero pixels
  static constexpr int YOFF = 3 * XPL;
  static constexpr int LDS = YOFF + 3 * YPL;
  static constexpr int XITEMS = XR * IW * CIN / 4, NXI = (XITEMS + 255) / 256;
  static constexpr int YITEMS = NPIX * COUT / 4, NYI = (YITEMS + 255) / 256;
  static constexpr int ROWB = IW * CIN * 4, IMGB = IH * ROWB, YIMGB = OH * OW * COUT * 4, YUNITB = NPIX * COUT * 4;
  static constexpr int RED = M * COUT * 4;                     // bytes of the epilogue's [M][COUT] block
  static_assert(CIN == 16 || CIN == 32, "input channels");
  static_assert(COUT == 16 || COUT == 32, "output channels");
  static_assert(TROWS >= CIN && OH % BR == 0 && WM * WK == 4, "tiling");
  static_assert(XPL % 16 == 0 && YPL % 16 == 0 && LDS % 16 == 0, "alignment");
  static_assert(2 * XPL + ((KH - 1) * RP + KW) * PIXB < 65536 && 2 * YPL < 65536, "immediate offsets");
  static_assert(RED + 4096 <= LDS, "epilogue blocks fit the planes' LDS");
  static_assert(LDS <= 80 * 1024, "two workgroups per CU");
};

struct Params {
  const float* X; const float* dY; float* partial_w; float* partial_b;
  int n_img, units, per_wg;
  long long x_bytes, y_bytes;
};

// pixel (within the k-step) of element j of transposing read rd, 16-lane group g
template <bool M32>
__device__ __forceinline__ int pix_of(int g, int rd, int j) {
  return M32 ? 8 * (g >> 1) + 4 * rd + j : 16 * (g >> 1) + 8 * rd + 4 * (g & 1) + j;
}

__device__ __forceinline__ s16x4_t tr_read(const unsigned char* smem, unsigned off) {
  typedef __attribute__((address_space(3))) s16x4_t lds_s16x4_t;
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(smem + off));
}
__device__ __forceinline__ bf16x8_t operand(const unsigned char* smem, unsigned o0, unsigned o1) {
  const s16x4_t a = tr_read(smem, o0), b = tr_read(smem, o1);
  const s16x8_t v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  return __builtin_bit_cast(bf16x8_t, v);
}

template <class G> struct Acc { typedef f32x16_t type; };
template <class G, bool M32 = G::M32> struct AccT { typedef f32x16_t type; };
template <class G> struct AccT<G, false> { typedef f32x4_t type; };

template <class G, bool RELU>
__global__ void __launch_bounds__(256, 2)
wgx_kernel(const Params p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  typedef typename AccT<G>::type acc_t;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / G::WK, wk = wave - wm * G::WK;
  const int u0 = blockIdx.x * p.per_wg;
  int u1 = u0 + p.per_wg; if (u1 > p.units) u1 = p.units;
  if (u0 >= u1) return;

  for (int i = tid * 16; i < G::LDS; i += 256 * 16) *reinterpret_cast<xg::u32x4_t*>(smem + i) = xg::u32x4_t{0u, 0u, 0u, 0u};

  const __amdgpu_buffer_rsrc_t xr = gemm::make_view(p.X, p.x_bytes), yr = gemm::make_view(p.dY, p.y_bytes);

  // ---- staging: item i = tid + 256 j of a unit's X rows / dY rows; only X's LDS offsets need a table (row / pixel
  // decode), everything else is tid * 16 plus immediates / scalar offsets --------------------------------------------- //
  unsigned xdst[G::NXI];
#pragma unroll
  for (int j = 0; j < G::NXI; ++j) {
    const int i = tid + 256 * j;
    constexpr int per_row = G::IW * G::CIN / 4, per_pix = G::CIN / 4;
    const int r = i / per_row, rem = i - r * per_row, x = rem / per_pix, q = rem - x * per_pix;
    xdst[j] = (unsigned)(((r * G::RP + x + G::PAD) * G::CIN + 4 * q) * 2);
  }
  const unsigned i16 = (unsigned)tid * 16u;
  f32x4_t lx[G::NXI], ly[G::NYI];
  // request item j of unit u (asynchronous); rows above / below the image and items past the unit come back as zeros
  auto issue_x = [&](int u, int j) {
    const int img = u / G::NB, band = u - img * G::NB;
    const int rowoff = (band * G::BR * G::S - G::PAD) * G::ROWB;       // first input row of the unit (may be -PAD rows)
    const unsigned off = (unsigned)(rowoff + 4096 * j) + i16;
    const bool in = off < (unsigned)G::IMGB && (j + 1 < G::NXI || tid + 256 * j < G::XITEMS);
    lx[j] = xg::view_load_s(xr, in ? off : kOut, (unsigned)img * (unsigned)G::IMGB);
  };
  auto issue_y = [&](int u, int j) {
    const int img = u / G::NB, band = u - img * G::NB;
    const unsigned ys = (unsigned)img * (unsigned)G::YIMGB + (unsigned)band * (unsigned)G::YUNITB;
    ly[j] = xg::view_load_s(yr, (j + 1 < G::NYI || tid + 256 * j < G::YITEMS) ? i16 : kOut, ys + (unsigned)(4096 * j));
  };
  // bias gradient: a plain sum of dY over ~10^6 largely cancelling terms -- the per-thread running sums are kept in
  // fp64 (one conversion + add per unit and component; a unit's own <= 6 items are added in fp32 first) and combined in
  // fp64 across the workgroup's threads: only the 512 per-workgroup partials are added in fp32 (r5: the per-tensor fp64
  // gate of the full-size test had a bias gradient 1.8x further from the truth than torch's pairwise fp32 sum)
  double bsum[4] = {0.0, 0.0, 0.0, 0.0};
  auto put3 = [&](const f32x4_t& v, unsigned dst, int plane) {
    unsigned h0, m0, l0, h1, m1, l1;
    xg::split2_trunc(v[0], v[1], h0, m0, l0);
    xg::split2_trunc(v[2], v[3], h1, m1, l1);
    *reinterpret_cast<u32x2_t*>(smem + dst) = u32x2_t{h0, h1};
    *reinterpret_cast<u32x2_t*>(smem + dst + plane) = u32x2_t{m0, m1};
    *reinterpret_cast<u32x2_t*>(smem + dst + 2 * plane) = u32x2_t{l0, l1};
  };
  // registers -> three planes in LDS; every item is requested again (for the NEXT unit `un`) as soon as its registers
  // are free: the requests fly under the rest of this split and the whole MFMA phase
  auto put = [&](int un) {
    const bool more = un < u1;
#pragma unroll
    for (int j = 0; j < G::NXI; ++j) {
      f32x4_t v = lx[j];
      if (more) issue_x(un, j);
      if (RELU) xg::relu4(v);
      if (j + 1 < G::NXI || tid + 256 * j < G::XITEMS) put3(v, xdst[j], G::XPL);
    }
    f32x4_t usum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < G::NYI; ++j) {
      const f32x4_t v = ly[j];
      if (more) issue_y(un, j);
      if (j + 1 < G::NYI || tid + 256 * j < G::YITEMS) { usum += v; put3(v, (unsigned)G::YOFF + (unsigned)tid * 8u + 2048u * j, G::YPL); }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) bsum[c] += (double)usum[c];
  };

  // ---- per-lane operand addresses of this wave's k-steps (unit independent) ------------------------------------ //
  const int g = lane >> 4, c16 = lane & 15, jrow = c16 >> 2, q = c16 & 3;
  constexpr bool kPairAdj = G::TPT == 2 && (G::KW % 2 == 0);  // a tile's two taps are x-neighbours: fold into the base
  unsigned xb[G::KPW][2], yb[G::KPW][2];
#pragma unroll
  for (int jj = 0; jj < G::KPW; ++jj)
#pragma unroll
    for (int rd = 0; rd < 2; ++rd) {
      const int pix = (wk + jj * G::WK) * G::KS + pix_of<G::M32>(g, rd, jrow);
      const int pc = pix < G::NPIX ? pix : G::NPIX - 1;
      const int oy = pc / G::OW, ox = pc - oy * G::OW;
      const int half = (g & 1) * 16;                         // second 16 columns of a 32-wide block
      const int xch = (G::M32 && G::CIN == 32) ? half + 4 * q : 4 * q;
      const int ych = G::M32 ? half + 4 * q : 4 * q;
      xb[jj][rd] = (unsigned)(((oy * G::S) * G::RP + ox * G::S) * G::PIXB + xch * 2 + (kPairAdj ? (g & 1) * G::PIXB : 0));
      yb[jj][rd] = (unsigned)(G::YOFF + pix * G::YPIXB + ych * 2);
    }
  auto tap_off = [](int tap) { return ((tap / G::KW) * G::RP + (tap % G::KW)) * G::PIXB; };

  acc_t acc[G::TPW];
#pragma unroll
  for (int t = 0; t < G::TPW; ++t)
#pragma unroll
    for (int r = 0; r < G::ACCN; ++r) acc[t][r] = 0.f;

  auto mfma = [&](const bf16x8_t& a, const bf16x8_t& b, acc_t& c) {
    if constexpr (G::M32) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    else c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  };
  // The MFMA phase of one unit for tap group WMI (compile time: every tap offset is an immediate).  The (k-step, tile)
  // pairs form one static sequence; the operand reads of pair s + 1 -- X's planes, and dY's when it opens a k-step --
  // are requested BEFORE the six MFMAs of pair s and pinned there (sched_barrier): left alone, hipcc hoists the reads of
  // the whole unrolled phase and spills, and reads requested at the head of their own k-step leave the matrix pipe idle
  // for an LDS round trip per k-step (r5: 24 of 108 us at the second Atari conv).
  auto phase = [&](auto wmi) {
    constexpr int WMI = decltype(wmi)::value;
    constexpr int NT = (G::TILES - WMI * G::TPW) < G::TPW ? (G::TILES - WMI * G::TPW) : G::TPW;   // this group's tiles
    auto load_a = [&](int jj, int t, bf16x8_t (&a)[3]) {
      const int tt = WMI * G::TPW + t;
      unsigned o0 = xb[jj][0], o1 = xb[jj][1];
      int imm = 0;
      if (G::TPT == 2 && !kPairAdj) {                        // the lane's tap of the pair (odd tap count: the last tile's
        const int ta = 2 * tt, tb = 2 * tt + 1 < G::NTAPS ? 2 * tt + 1 : 2 * tt;   // second half repeats the first)
        const unsigned d = (g & 1) ? (unsigned)tap_off(tb) : (unsigned)tap_off(ta);
        o0 += d; o1 += d;
      } else {
        imm = tap_off(tt * G::TPT);
      }
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) a[pl] = operand(smem, o0 + imm + pl * G::XPL, o1 + imm + pl * G::XPL);
    };
    auto load_b = [&](int jj, bf16x8_t (&b)[3]) {
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) b[pl] = operand(smem, yb[jj][0] + pl * G::YPL, yb[jj][1] + pl * G::YPL);
    };
    auto live = [&](int jj) { return wk + jj * G::WK < G::KSTEPS; };   // (uniform; only a wave's LAST k-step can be dead)
    bf16x8_t b[2][3], a[2][3];
    if (live(0)) { load_b(0, b[0]); load_a(0, 0, a[0]); }
#pragma unroll
    for (int jj = 0; jj < G::KPW; ++jj) {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int s = jj * NT + t;
        if (t + 1 < NT) {
          if (live(jj)) load_a(jj, t + 1, a[(s + 1) & 1]);
        } else if (jj + 1 < G::KPW) {
          if (live(jj + 1)) { load_b(jj + 1, b[(jj + 1) & 1]); load_a(jj + 1, 0, a[(s + 1) & 1]); }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (live(jj)) {
          const bf16x8_t (&x)[3] = a[s & 1];
          const bf16x8_t (&y)[3] = b[jj & 1];
          mfma(x[2], y[0], acc[t]);
          mfma(x[0], y[2], acc[t]);
          mfma(x[1], y[1], acc[t]);
          mfma(x[1], y[0], acc[t]);
          mfma(x[0], y[1], acc[t]);
          mfma(x[0], y[0], acc[t]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };

  // The unit loop, instantiated per tap group (branching on wm inside it makes hipcc copy the accumulators around).
  // Two such workgroups share a CU and drift freely: one's split phase (VALU, LDS writes) under the other's MFMAs.  (r5
  // also tried ONE 8-wave workgroup whose two teams of four waves run exactly half a step apart behind workgroup-wide
  // barriers -- half the partial slices --: 1.015 vs 1.013 ms for cfg2's step, 14.43 vs 14.30 for cfg3's; a team's
  // latency bubbles are lost when the other team may not run ahead.  profiles/r05_wgx_step_ab.txt)
  auto run = [&](auto wmi) {
    for (int u = u0; u < u1; ++u) {
      put(u + 1);
      __syncthreads();
      phase(wmi);
      __syncthreads();
    }
  };
#pragma unroll
  for (int j = 0; j < G::NXI; ++j) issue_x(u0, j);
#pragma unroll
  for (int j = 0; j < G::NYI; ++j) issue_y(u0, j);
  __syncthreads();                                           // LDS zeroed
  if (G::WM == 1 || wm == 0) run(std::integral_constant<int, 0>());
  else if (G::WM == 2 || wm == 1) run(std::integral_constant<int, 1>());
  else if (wm == 2) run(std::integral_constant<int, G::WM == 4 ? 2 : 0>());
  else run(std::integral_constant<int, G::WM == 4 ? 3 : 0>());

  // ---- bias gradient: thread t summed output channels 4 (t % (COUT / 4)) .. + 3 ---------------------------------- //
  if (p.partial_b) {
    double* bred = reinterpret_cast<double*>(smem);          // [256 threads][4] (the planes are dead by now)
#pragma unroll
    for (int c = 0; c < 4; ++c) bred[tid * 4 + c] = bsum[c];
    __syncthreads();
    if (tid < G::COUT) {
      constexpr int per = G::COUT / 4;
      double s = 0.0;
      for (int t = tid >> 2; t < 256; t += per) s += bred[t * 4 + (tid & 3)];
      p.partial_b[(long long)blockIdx.x * G::COUT + tid] = (float)s;
    }
    __syncthreads();                                         // (the slice block below reuses these bytes)
  }
  // ---- the workgroup's slice: the wk-waves add their tiles into an [M][COUT] block in LDS one after the other (fixed
  // order), which then leaves as consecutive 16-byte stores -------------------------------------------------------- //
  float* red = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int w2 = 0; w2 < G::WK; ++w2) {
    if (wk == w2) {
#pragma unroll
      for (int t = 0; t < G::TPW; ++t) {
        const int tt = wm * G::TPW + t;
#pragma unroll
        for (int r = 0; r < G::ACCN; ++r) {
          const int row = G::M32 ? 32 * tt + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3) : 16 * tt + 4 * (lane >> 4) + r;
          const int col = G::M32 ? (lane & 31) : (lane & 15);
          if (tt < G::TILES && row < G::M) {
            if (w2 == 0) red[row * G::COUT + col] = acc[t][r]; else red[row * G::COUT + col] += acc[t][r];
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __syncthreads();
  }
  f32x4_t* pw = reinterpret_cast<f32x4_t*>(p.partial_w + (long long)blockIdx.x * (G::M * G::COUT));
  for (int i = tid; i < G::M * G::COUT / 4; i += 256) {
    pw[i] = *reinterpret_cast<const f32x4_t*>(smem + i * 16);
    // (a 16-byte store reads its data registers over several cycles; hipcc reuses them for the next ds_read in the very
    // next issue slot -- tools/isa_store_hazard.py -- so keep a wait state between the two)
    asm volatile("s_nop 4" ::: "memory");
  }
}

// ---- the served geometries ----------------------------------------------------------------------------------------- //
//                 KH KW S PAD CIN COUT IH  IW  OH  OW  BR WM WK
typedef Geo<4, 4, 2, 0, 16, 32, 20, 20, 9, 9, 9, 4, 1> GeoAtari2;       // second Atari conv: wave = kernel row
typedef Geo<3, 3, 1, 1, 16, 16, 36, 48, 36, 48, 6, 2, 2> GeoDeep16;     // ImpalaDeep stack 0 residual layers
typedef Geo<3, 3, 1, 1, 16, 32, 36, 48, 36, 48, 4, 1, 4> GeoDeep16x32;  // stack 1 entry layer
typedef Geo<3, 3, 1, 1, 32, 32, 18, 24, 18, 24, 6, 2, 2> GeoDeep32a;    // stack 1 residual / stack 2 entry layers
typedef Geo<3, 3, 1, 1, 32, 32, 9, 12, 9, 12, 9, 2, 2> GeoDeep32b;      // stack 2 residual layers

template <class G>
inline bool matches(const seedhip_conv_geom* g) {
  return g->kh == G::KH && g->kw == G::KW && g->stride == G::S && g->pad_t == G::PAD && g->pad_l == G::PAD && g->cin == G::CIN &&
         g->cout == G::COUT && g->ih == G::IH && g->iw == G::IW && g->oh == G::OH && g->ow == G::OW && g->ld_in == G::CIN &&
         g->ld_out == G::COUT;
}
inline int which(const seedhip_conv_geom* g) {
  if (matches<GeoAtari2>(g)) return 1;
  if (matches<GeoDeep16>(g)) return 2;
  if (matches<GeoDeep16x32>(g)) return 3;
  if (matches<GeoDeep32a>(g)) return 4;
  if (matches<GeoDeep32b>(g)) return 5;
  return 0;
}
inline int units_per_image(int k) { return k == 1 ? GeoAtari2::NB : k == 2 ? GeoDeep16::NB : k == 3 ? GeoDeep16x32::NB : k == 4 ? GeoDeep32a::NB : GeoDeep32b::NB; }

// grid (= partial slices) for n_img images of geometry k: two workgroups per CU, contiguous runs of units
int grid_for(int k, int n_img, int* per_wg_out) {
  static const int cus = xg::cu_count();
  const int units = n_img * units_per_image(k);
  int grid = units < 2 * cus ? units : 2 * cus;
  const int per_wg = (units + grid - 1) / grid;
  grid = (units + per_wg - 1) / per_wg;
  if (per_wg_out) *per_wg_out = per_wg;
  return grid;
}

// Eligibility (geometry only).  0: not served.
int plan(const seedhip_conv_geom* g) {
  const int k = which(g);
  if (!k) return 0;
  if (g->n_img < 32) return 0;                               // (a handful of images: the generic kernels' grids fill better)
  const long long xb = (long long)g->n_img * g->ih * g->iw * g->cin * 4, yb = (long long)g->n_img * g->oh * g->ow * g->cout * 4;
  if (xb >= (1LL << 31) - (1 << 22) || yb >= (1LL << 31) - (1 << 22)) return 0;
  return k;
}

template <class G, bool RELU>
inline int launch_one(Params& p, int grid, hipStream_t s) {
  static const bool ok = hipFuncSetAttribute((const void*)wgx_kernel<G, RELU>, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS) == hipSuccess;
  if (!ok) return fail(SEEDHIP_ERR_LAUNCH, "wgx_kernel: LDS attribute");
  hipLaunchKernelGGL((wgx_kernel<G, RELU>), dim3(grid), dim3(256), G::LDS, s, p);
  return check_launch("wgx_kernel");
}
template <class G>
inline int launch_geo(Params& p, int in_relu, int grid, hipStream_t s) {
  return in_relu ? launch_one<G, true>(p, grid, s) : launch_one<G, false>(p, grid, s);
}

// partial_w [grid][M][COUT], partial_b [grid][COUT] (or null); returns the slice count through *slices
int launch(int k, const seedhip_conv_geom* g, const float* X, int in_relu, const float* dY, float* partial_w,
                  float* partial_b, int* slices, hipStream_t s) {
  Params p;
  memset(&p, 0, sizeof(p));
  p.X = X; p.dY = dY; p.partial_w = partial_w; p.partial_b = partial_b; p.n_img = g->n_img;
  p.units = g->n_img * units_per_image(k);
  const int grid = grid_for(k, g->n_img, &p.per_wg);
  p.x_bytes = (long long)g->n_img * g->ih * g->iw * g->cin * 4;
  p.y_bytes = (long long)g->n_img * g->oh * g->ow * g->cout * 4;
  *slices = grid;
  switch (k) {
    case 1: return launch_geo<GeoAtari2>(p, in_relu, grid, s);
    case 2: return launch_geo<GeoDeep16>(p, in_relu, grid, s);
    case 3: return launch_geo<GeoDeep16x32>(p, in_relu, grid, s);
    case 4: return launch_geo<GeoDeep32a>(p, in_relu, grid, s);
    default: return launch_geo<GeoDeep32b>(p, in_relu, grid, s);
  }
}

}  // namespace wgx
}  // namespace seedhip

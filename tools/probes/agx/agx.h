// The shallow Atari torso's second convolution -- Conv2D(32, 4, 2, 'valid') + ReLU on the 20 x 20 x 16 map of the first
// (/root/reference/atari/networks.py:236) -- forward and data gradient at training batch sizes on cgx.h's machine
// (bf16 matrix pipe, exact three-way split of both operands: xgemm.h).  wfx.h / wdx.h (r4) stage rows through a ring and
// spend 6.7 / 8.0 VALU instructions per MFMA with the pipe 0.32 / 0.39 busy (r5 counters); here whole images are staged
// once per unit as three bf16 planes, even and odd columns of a row apart (a tap's consecutive output pixels read
// consecutive slots), weights live in registers for the whole launch, one 8-wave workgroup per CU:
//   * FORWARD: units of three images = a run of 243 output pixels = eight 32-pixel tiles.  A wave is (tile slot 0..3) x
//     (half of the 16 taps): 8 reduction steps of one tap x 16 channels (96 weight registers), 48 MFMAs per tile; the two
//     halves of a slot exchange two accumulator quads each through LDS (one barrier per round of four tiles) and every
//     wave adds, applies bias + ReLU, stores two quads (and their ReLU-mask bytes: seedhip_conv2d_fwd_bits);
//   * DATA GRADIENT: dX[2q + py, 2r + px, ci] = sum_{j, i} sum_co dY[q - j, r - i, co] W[py + 2j, px + 2i, ci, co] -- four
//     stride-parity classes of 2 x 2 taps over dY zero-padded by one.  The two classes (py, 0), (py, 1) read the SAME dY
//     pixels and have 16 input channels each: stacked they fill the 32 rows of one MFMA.  A wave is (py) x (tile slot);
//     its reduction is 4 taps x 32 channels = 8 steps (96 registers), NOTHING crosses between waves: units of five images
//     (16 tiles of the 10 x 10 class grid), two barriers per unit; the ReLU mask of the first conv's output in the
//     epilogue as bytes (seedhip_conv2d_bwd_data_bits) or fp32.
// Staging, in-flight items and their waits: fgx.h / cgx.h (xg::take_item).
#pragma once
#include <type_traits>
#include "common.h"
#include "xgemm.h"
#include "agx_api.h"

namespace seedhip {
namespace agx {

using xg::bf16x8_t;
using xg::f32x4_t;
using xg::f32x16_t;
using xg::u32x2_t;
using xg::u32x4_t;
constexpr unsigned kOut = 0x80000000u;
constexpr int odd64(int bytes) { return (((bytes + 63) / 64) | 1) * 64; }

struct Params {
  const float* X; const float* W; const float* bias; const float* mask; float* Y;
  unsigned char* bits_out; const unsigned char* bits_in;
  int n_img, units, per_wg, out_relu;
  long long x_bytes, y_bytes;
};

__device__ __forceinline__ f32x4_t quad(const f32x16_t& a, int g) { return f32x4_t{a[4 * g], a[4 * g + 1], a[4 * g + 2], a[4 * g + 3]}; }
__device__ __forceinline__ unsigned sign_bits(const f32x4_t& v) {
  return (v[0] > 0.f ? 1u : 0u) | (v[1] > 0.f ? 2u : 0u) | (v[2] > 0.f ? 4u : 0u) | (v[3] > 0.f ? 8u : 0u);
}
#define AGX_MFMA6(ACC, WL, WM, WH, X)                                              \
  ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(WL, X[0], ACC, 0, 0, 0);          \
  ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(WH, X[2], ACC, 0, 0, 0);          \
  ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(WM, X[1], ACC, 0, 0, 0);          \
  ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(WM, X[0], ACC, 0, 0, 0);          \
  ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(WH, X[1], ACC, 0, 0, 0);          \
  ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(WH, X[0], ACC, 0, 0, 0);

// ================================================= forward ====================================================== //
struct Fwd {
  static constexpr int G = 3, IH = 20, IW = 20, HW = 10, OH = 9, OW = 9, CIN = 16, COUT = 32;
  static constexpr int SLOTS = G * IH * IW, CBP = odd64(SLOTS * 16), XPL = 2 * CBP, XBYTES = 3 * XPL;
  static constexpr int EXSLOT = 8 * 2048, BIAS = XBYTES + 2 * EXSLOT, LDS = BIAS + 128 + 64;
  static constexpr int ITEMS = G * IH * IW * 4, NXI = (ITEMS + 511) / 512;
  static constexpr int NP = G * OH * OW, T = (NP + 31) / 32, TPW = T / 4;
  static_assert(T % 4 == 0 && LDS <= 160 * 1024, "tiles per slot; one workgroup per CU");
  static constexpr int slot(int img, int row, int col) { return (img * IH + row) * IW + (col & 1) * HW + (col >> 1); }
  static constexpr int tap_off(int ky, int kx) { return (ky * IW + (kx & 1) * HW + (kx >> 1)) * 16; }
};

__global__ void __launch_bounds__(512, 2)
agx_fwd_kernel(const Params p) {
  typedef Fwd G;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ts = wave & 3, kh = wave >> 2;                   // tile slot; taps 8 kh .. 8 kh + 7
  const int px = lane & 31, kb = lane >> 5;
  const int u0 = blockIdx.x * p.per_wg;
  int u1 = u0 + p.per_wg; if (u1 > p.units) u1 = p.units;
  if (u0 >= u1) return;

  for (int i = tid * 16; i < G::XBYTES; i += 512 * 16) *reinterpret_cast<u32x4_t*>(smem + i) = u32x4_t{0u, 0u, 0u, 0u};
  const __amdgpu_buffer_rsrc_t xr = gemm::make_view(p.X, p.x_bytes), yr = gemm::make_view(p.Y, p.y_bytes);
  const __amdgpu_buffer_rsrc_t br = gemm::make_view(reinterpret_cast<const float*>(p.bits_out ? p.bits_out : (unsigned char*)p.Y), p.bits_out ? p.y_bytes >> 4 : 0);
  float* bias_lds = reinterpret_cast<float*>(smem + G::BIAS);
  if (tid < 32) bias_lds[tid] = p.bias ? p.bias[tid] : 0.f;

  // ---- weights: step s = tap 8 kh + s; rows = the 32 output channels, reduction = input channels 8 kb .. 8 kb + 7 ---- //
  bf16x8_t wh[8], wm[8], wl[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = p.W[((8 * kh + s) * 16 + 8 * kb + e) * 32 + px];
    u32x4_t h, m, l;
#pragma unroll
    for (int e = 0; e < 4; ++e) { unsigned a, b, c; xg::split2(v[2 * e], v[2 * e + 1], a, b, c); h[e] = a; m[e] = b; l[e] = c; }
    wh[s] = __builtin_bit_cast(bf16x8_t, h); wm[s] = __builtin_bit_cast(bf16x8_t, m); wl[s] = __builtin_bit_cast(bf16x8_t, l);
  }

  // ---- staging: item i = tid + 512 j = quad q of input pixel i / 4 of the unit's images (contiguous in HBM).  Sixteen
  //      consecutive items = four pixels x two blocks x two halves: 32 banks once (blocks an odd multiple of 64 bytes apart) //
  auto item_dst = [&](int j) -> unsigned {                   // (recomputed where used: registers; pinned against hoisting)
    int tv = tid;
    asm volatile("" : "+v"(tv));
    const unsigned i = (unsigned)tv + 512u * j, pix = i >> 2, q = i & 3u;
    const unsigned img = pix / 400u, rem = pix - img * 400u, r = rem / 20u, c = rem - r * 20u;
    const unsigned dst = (q >> 1) * G::CBP + ((img * G::IH + r) * G::IW + (c & 1u) * G::HW + (c >> 1)) * 16u + (q & 1u) * 8u;
    return i < (unsigned)G::ITEMS ? dst : kOut;
  };
  const unsigned i16 = (unsigned)tid * 16u;
  f32x4_t lx[G::NXI];
  auto issue_x = [&](int u, int j, bool more) __attribute__((always_inline)) {
    const unsigned off = 8192u * (unsigned)j + i16;
    const long long img0 = (long long)u * G::G;
    const unsigned lim = (unsigned)(((long long)p.n_img - img0 < G::G ? (long long)p.n_img - img0 : G::G) * (400 * 64));
    const unsigned voff = (more && off < lim) ? (unsigned)(img0 * (400 * 64)) + off : kOut;
    asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(lx[j]) : "v"(voff), "s"(xr));
  };
  constexpr int kQueue = 2 * G::TPW;                         // per unit behind the staging requests: >= two output quads per tile
  auto put = [&](int un, auto first) __attribute__((always_inline)) {
    const bool more = un < u1;
#pragma unroll
    for (int j = 0; j < G::NXI; ++j) {
      const f32x4_t it = decltype(first)::value ? xg::take_item<G::NXI - 1>(lx[j]) : xg::take_item<G::NXI - 1 + kQueue>(lx[j]);
      unsigned h0, m0, l0, h1, m1, l1;
      xg::split2_trunc(it[0], it[1], h0, m0, l0);
      xg::split2_trunc(it[2], it[3], h1, m1, l1);
      const unsigned dst = item_dst(j);
      constexpr unsigned kDump = (unsigned)(G::LDS - 64);
      *reinterpret_cast<u32x2_t*>(smem + (dst == kOut ? kDump : dst)) = u32x2_t{h0, h1};
      *reinterpret_cast<u32x2_t*>(smem + (dst == kOut ? kDump + 16u : dst + G::XPL)) = u32x2_t{m0, m1};
      *reinterpret_cast<u32x2_t*>(smem + (dst == kOut ? kDump + 32u : dst + 2 * G::XPL)) = u32x2_t{l0, l1};
      issue_x(un, j, more);
    }
  };

  // ---- this lane's pixels: tile ts + 4 k of the unit's run, pixel P = 32 tile + px ----------------------------------- //
  unsigned pb[G::TPW];
#pragma unroll
  for (int k = 0; k < G::TPW; ++k) {
    int P = 32 * (ts + 4 * k) + px; if (P >= G::NP) P = 0;
    const int img = P / 81, rem = P - img * 81, oy = rem / 9, ox = rem - oy * 9;
    pb[k] = (unsigned)(kb * G::CBP + G::slot(img, 2 * oy, 2 * ox) * 16);
  }
  unsigned char* exs = smem + G::XBYTES;
  unsigned parity = 0;

  auto compute = [&](auto KH, int u) __attribute__((always_inline)) {
    constexpr int kH = decltype(KH)::value;                  // this wave finishes quads 2 kH, 2 kH + 1: channels 16 kH + 8 g + 4 kb ..
    long long left = ((long long)p.n_img - (long long)u * G::G) * 81;
    const int npx = left < G::NP ? (int)left : G::NP;
    const unsigned ys = (unsigned)((long long)u * G::NP * 128);
#pragma unroll
    for (int k = 0; k < G::TPW; ++k) {
      const int P = 32 * (ts + 4 * k) + px;
      const unsigned o0 = P < npx ? (unsigned)(P * 32 + 16 * kH + 4 * kb) * 4u : kOut;
      f32x16_t acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      bf16x8_t xv[2][3];
      auto fetch = [&](int s, bf16x8_t (&x)[3]) {              // tap 8 kH + s = (2 kH + s / 4, s % 4)
        const int off = G::tap_off(2 * kH + s / 4, s % 4);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) x[pl] = *reinterpret_cast<const bf16x8_t*>(smem + pb[k] + off + pl * G::XPL);
      };
      fetch(0, xv[0]);
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        if (s + 1 < 8) fetch(s + 1, xv[(s + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
        const bf16x8_t (&x)[3] = xv[s & 1];
        AGX_MFMA6(acc, wl[s], wm[s], wh[s], x)
        __builtin_amdgcn_sched_barrier(0);
      }
      // exchange slot: [tile slot][receiving half][two quads][lane] x 16 bytes
      unsigned char* slot = exs + parity * G::EXSLOT + ts * 4096 + lane * 16;
#pragma unroll
      for (int g = 0; g < 2; ++g) *reinterpret_cast<f32x4_t*>(slot + (1 - kH) * 2048 + g * 1024) = quad(acc, 2 * (1 - kH) + g);
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      f32x4_t o[2];
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const f32x4_t own = quad(acc, 2 * kH + g), got = *reinterpret_cast<const f32x4_t*>(slot + kH * 2048 + g * 1024);
        const f32x4_t b4 = *reinterpret_cast<const f32x4_t*>(bias_lds + 16 * kH + 8 * g + 4 * kb);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          o[g][e] = b4[e] + (kH == 0 ? own[e] + got[e] : got[e] + own[e]);        // (taps 0-7 first, whoever adds)
          if (p.out_relu) o[g][e] = fmaxf(o[g][e], 0.f);
        }
        asm volatile("" : "+v"(o[g]));
      }
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, o[g]), yr, o0 == kOut ? kOut : o0 + 32u * g, ys, 0);
        asm volatile("s_nop 1" ::: "memory");
      }
      if (p.bits_out) {
#pragma unroll
        for (int g = 0; g < 2; ++g)
          __builtin_amdgcn_raw_buffer_store_b8((unsigned char)sign_bits(o[g]), br, o0 == kOut ? kOut : (o0 >> 4) + 2u * g, ys >> 4, 0);
      }
      parity ^= 1u;
    }
  };
  auto run = [&](auto KH) __attribute__((always_inline)) {
    auto step = [&](int u, auto first) __attribute__((always_inline)) {
      put(u + 1, first);
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      compute(KH, u);                                        // (its last barrier also frees the planes for the next put)
    };
    step(u0, std::true_type());
    for (int u = u0 + 1; u < u1; ++u) step(u, std::false_type());
  };
#pragma unroll
  for (int s = 0; s < 8; ++s) asm volatile("" :: "v"(wh[s]), "v"(wm[s]), "v"(wl[s]));   // (weights finished before the first requests: cgx.h)
#pragma unroll
  for (int j = 0; j < G::NXI; ++j) issue_x(u0, j, true);
  __syncthreads();                                           // LDS zeroed, bias in place
  if (kh == 0) run(std::integral_constant<int, 0>()); else run(std::integral_constant<int, 1>());
}

// ============================================== data gradient =================================================== //
struct Dgr {
  static constexpr int G = 5, IH = 9, IW = 9, IHP = 11, IWP = 11, SLOTS = G * IHP * IWP;
  static constexpr int CBP = odd64(SLOTS * 16), XPL = 4 * CBP, XBYTES = 3 * XPL, LDS = XBYTES + 64;
  static constexpr int ITEMS = G * IH * IW * 8, NXI = (ITEMS + 511) / 512;
  static constexpr int GH = 10, GW = 10, NP = G * GH * GW, T = (NP + 31) / 32, TPW = T / 4;
  static constexpr int XH = 20, XW = 20, XC = 16;
  static_assert(T % 4 == 0 && LDS <= 160 * 1024, "tiles per slot; one workgroup per CU");
};

__global__ void __launch_bounds__(512, 2)
agx_dg_kernel(const Params p) {
  typedef Dgr G;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int py = wave & 1, ts = wave >> 1;                   // row parity of the two classes this wave computes; tile slot
  const int px = lane & 31, kb = lane >> 5;
  const int u0 = blockIdx.x * p.per_wg;
  int u1 = u0 + p.per_wg; if (u1 > p.units) u1 = p.units;
  if (u0 >= u1) return;

  for (int i = tid * 16; i < G::XBYTES; i += 512 * 16) *reinterpret_cast<u32x4_t*>(smem + i) = u32x4_t{0u, 0u, 0u, 0u};
  const __amdgpu_buffer_rsrc_t xr = gemm::make_view(p.X, p.x_bytes), yr = gemm::make_view(p.Y, p.y_bytes);
  const __amdgpu_buffer_rsrc_t mr = gemm::make_view(p.mask ? p.mask : p.Y, p.mask ? p.y_bytes : 0);
  const __amdgpu_buffer_rsrc_t br = gemm::make_view(reinterpret_cast<const float*>(p.bits_in ? p.bits_in : (const unsigned char*)p.Y), p.bits_in ? p.y_bytes >> 4 : 0);
  const bool has_mask = p.mask != nullptr, has_bits = p.bits_in != nullptr;

  // ---- weights: rows = (column parity pxc = row / 16, input channel ci = row % 16); step s = 2 (2 j + i) + sub: tap
  //      (py + 2 j, pxc + 2 i), output channels 16 sub + 8 kb .. + 7 of W (eight consecutive floats) ------------------- //
  bf16x8_t wh[8], wm[8], wl[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    const int j = s >> 2, i = (s >> 1) & 1, sub = s & 1, pxc = px >> 4, ci = px & 15;
    const float* src = p.W + (((py + 2 * j) * 4 + pxc + 2 * i) * 16 + ci) * 32 + 16 * sub + 8 * kb;
    const f32x4_t v0 = *reinterpret_cast<const f32x4_t*>(src), v1 = *reinterpret_cast<const f32x4_t*>(src + 4);
    const float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
    u32x4_t h, m, l;
#pragma unroll
    for (int e = 0; e < 4; ++e) { unsigned a, b, c; xg::split2(v[2 * e], v[2 * e + 1], a, b, c); h[e] = a; m[e] = b; l[e] = c; }
    wh[s] = __builtin_bit_cast(bf16x8_t, h); wm[s] = __builtin_bit_cast(bf16x8_t, m); wl[s] = __builtin_bit_cast(bf16x8_t, l);
  }

  // ---- staging of dY: item i = ti + 512 j = quad q (of 8) of pixel i / 8; lanes permuted inside each 64-item chunk
  //      (eight pixels x eight quads): a 16-lane group holds quads 4 g' .. 4 g' + 3 (two blocks) of four pixels ------------ //
  const int ti = (tid & ~63) + (((lane >> 4) >> 1) * 4 + ((lane >> 2) & 3)) * 8 + 4 * ((lane >> 4) & 1) + (lane & 3);
  auto item_dst = [&](int j) -> unsigned {
    int tv = ti;
    asm volatile("" : "+v"(tv));
    const unsigned i = (unsigned)tv + 512u * j, pix = i >> 3, q = i & 7u;
    const unsigned img = pix / 81u, rem = pix - img * 81u, r = rem / 9u, c = rem - r * 9u;
    const unsigned dst = (q >> 1) * G::CBP + ((img * G::IHP + r + 1) * G::IWP + c + 1) * 16u + (q & 1u) * 8u;
    return i < (unsigned)G::ITEMS ? dst : kOut;
  };
  const unsigned i16 = (unsigned)ti * 16u;
  f32x4_t lx[G::NXI];
  auto issue_x = [&](int u, int j, bool more) __attribute__((always_inline)) {
    const unsigned off = 8192u * (unsigned)j + i16;
    const long long img0 = (long long)u * G::G;
    const unsigned lim = (unsigned)(((long long)p.n_img - img0 < G::G ? (long long)p.n_img - img0 : G::G) * (81 * 128));
    const unsigned voff = (more && off < lim) ? (unsigned)(img0 * (81 * 128)) + off : kOut;
    asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(lx[j]) : "v"(voff), "s"(xr));
  };
  constexpr int kQueue = 4 * G::TPW;                         // per unit behind the staging requests: >= four output quads per tile
  auto put = [&](int un, auto first) __attribute__((always_inline)) {
    const bool more = un < u1;
#pragma unroll
    for (int j = 0; j < G::NXI; ++j) {
      const f32x4_t it = decltype(first)::value ? xg::take_item<G::NXI - 1>(lx[j]) : xg::take_item<G::NXI - 1 + kQueue>(lx[j]);
      unsigned h0, m0, l0, h1, m1, l1;
      xg::split2_trunc(it[0], it[1], h0, m0, l0);
      xg::split2_trunc(it[2], it[3], h1, m1, l1);
      const unsigned dst = item_dst(j);
      constexpr unsigned kDump = (unsigned)(G::LDS - 64);
      *reinterpret_cast<u32x2_t*>(smem + (dst == kOut ? kDump : dst)) = u32x2_t{h0, h1};
      *reinterpret_cast<u32x2_t*>(smem + (dst == kOut ? kDump + 16u : dst + G::XPL)) = u32x2_t{m0, m1};
      *reinterpret_cast<u32x2_t*>(smem + (dst == kOut ? kDump + 32u : dst + 2 * G::XPL)) = u32x2_t{l0, l1};
      issue_x(un, j, more);
    }
  };

  // ---- this lane's pixels: tile ts + 4 k, class-grid pixel P = 32 tile + px = (image, q, r) --------------------------- //
  unsigned pb[G::TPW];
#pragma unroll
  for (int k = 0; k < G::TPW; ++k) {
    int P = 32 * (ts + 4 * k) + px; if (P >= G::NP) P = 0;
    const int img = P / 100, rem = P - img * 100, q = rem / 10, r = rem - q * 10;
    pb[k] = (unsigned)(kb * G::CBP + ((img * G::IHP + q) * G::IWP + r) * 16);       // dY[q - 1, r - 1] in padded slots
  }
  auto out_off = [&](int k) -> unsigned {                    // dX[image, 2 q + py, 2 r, channel 4 kb] (recomputed per tile, pinned)
    int pv = px;
    asm volatile("" : "+v"(pv));
    const unsigned P = 32u * (ts + 4 * k) + (unsigned)pv;
    const unsigned img = P / 100u, rem = P - img * 100u, q = rem / 10u, r = rem - q * 10u;
    return (((img * G::XH + 2 * q + py) * G::XW + 2 * r) * G::XC + 4 * kb) * 4u;
  };
  auto compute = [&](int u) __attribute__((always_inline)) {
    long long left = ((long long)p.n_img - (long long)u * G::G) * 100;
    const int npx = left < G::NP ? (int)left : G::NP;
    const unsigned ys = (unsigned)((long long)u * G::G * (400 * 64));
#pragma unroll
    for (int k = 0; k < G::TPW; ++k) {
      const unsigned o0 = (32 * (ts + 4 * k) + px) < npx ? out_off(k) : kOut;
      unsigned mb[4] = {0u, 0u, 0u, 0u};                       // the mask bytes fly under the tile's MFMAs
      if (has_bits) {
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4)
          mb[g4] = (unsigned)__builtin_amdgcn_raw_buffer_load_b8(br, o0 == kOut ? kOut : (o0 + (unsigned)((g4 >> 1) * 64 + (g4 & 1) * 32)) >> 4, ys >> 4, 0);
      }
      f32x16_t acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      bf16x8_t xv[2][3];
      auto fetch = [&](int s, bf16x8_t (&x)[3]) {              // tap (j, i) reads dY[q - j, r - i]; channel block 2 sub (+ kb in pb)
        const int j = s >> 2, i = (s >> 1) & 1, sub = s & 1;
        const int off = ((1 - j) * G::IWP + (1 - i)) * 16 + 2 * sub * G::CBP;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) x[pl] = *reinterpret_cast<const bf16x8_t*>(smem + pb[k] + off + pl * G::XPL);
      };
      fetch(0, xv[0]);
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        if (s + 1 < 8) fetch(s + 1, xv[(s + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
        const bf16x8_t (&x)[3] = xv[s & 1];
        AGX_MFMA6(acc, wl[s], wm[s], wh[s], x)
        __builtin_amdgcn_sched_barrier(0);
      }
      // quad g4: column parity g4 / 2, input channels 8 (g4 % 2) + 4 kb ..: dX pixel (2 q + py, 2 r + g4 / 2)
      f32x4_t o[4];
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        o[g4] = quad(acc, g4);
        const unsigned og = o0 == kOut ? kOut : o0 + (unsigned)((g4 >> 1) * 64 + (g4 & 1) * 32);
        if (has_bits) {
#pragma unroll
          for (int e = 0; e < 4; ++e) o[g4][e] = (mb[g4] >> e) & 1u ? o[g4][e] : 0.f;
        } else if (has_mask) {
          const f32x4_t mk = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(mr, og, ys, 0));
#pragma unroll
          for (int e = 0; e < 4; ++e) o[g4][e] = mk[e] > 0.f ? o[g4][e] : 0.f;
        }
        asm volatile("" : "+v"(o[g4]));
      }
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const unsigned og = o0 == kOut ? kOut : o0 + (unsigned)((g4 >> 1) * 64 + (g4 & 1) * 32);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, o[g4]), yr, og, ys, 0);
        asm volatile("s_nop 1" ::: "memory");
      }
    }
  };
  auto step = [&](int u, auto first) __attribute__((always_inline)) {
    put(u + 1, first);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    compute(u);
    asm volatile("s_barrier" ::: "memory");
  };
#pragma unroll
  for (int s = 0; s < 8; ++s) asm volatile("" :: "v"(wh[s]), "v"(wm[s]), "v"(wl[s]));
#pragma unroll
  for (int j = 0; j < G::NXI; ++j) issue_x(u0, j, true);
  __syncthreads();                                           // LDS zeroed
  step(u0, std::true_type());
  for (int u = u0 + 1; u < u1; ++u) step(u, std::false_type());
}
#undef AGX_MFMA6

// ---- host ------------------------------------------------------------------------------------------------------- //
inline bool geometry(const seedhip_conv_geom* g) {
  return g->kh == 4 && g->kw == 4 && g->stride == 2 && g->pad_t == 0 && g->pad_l == 0 && g->cin == 16 && g->cout == 32 &&
         g->ih == 20 && g->iw == 20 && g->oh == 9 && g->ow == 9 && g->ld_in == 16 && g->ld_out == 32;
}
bool plan(const seedhip_conv_geom* g) {
  if (!geometry(g) || g->n_img < kMinImages) return false;
  return (long long)g->n_img * 400 * 16 * 4 < (1LL << 31) - (1 << 22);
}
template <class G, class K>
inline int launch_k(K kernel, Params& p, hipStream_t s, const char* what) {
  static const int cus = xg::cu_count();
  p.units = (p.n_img + G::G - 1) / G::G;
  int grid = p.units < cus ? p.units : cus;
  p.per_wg = (p.units + grid - 1) / grid;
  grid = (p.units + p.per_wg - 1) / p.per_wg;
  if (hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS) != hipSuccess)
    return fail(SEEDHIP_ERR_LAUNCH, "agx: LDS attribute");
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(512), G::LDS, s, p);
  return check_launch(what);
}
int launch_fwd(const seedhip_conv_geom* g, const float* X, const float* W, const float* bias, float* Y, int out_relu,
               unsigned char* relu_bits, hipStream_t s) {
  Params p;
  memset(&p, 0, sizeof(p));
  p.X = X; p.W = W; p.bias = bias; p.Y = Y; p.n_img = g->n_img; p.out_relu = out_relu; p.bits_out = relu_bits;
  p.x_bytes = (long long)g->n_img * 400 * 16 * 4; p.y_bytes = (long long)g->n_img * 81 * 32 * 4;
  return launch_k<Fwd>(agx_fwd_kernel, p, s, "agx_fwd_kernel");
}
int launch_dgrad(const seedhip_conv_geom* g, const float* dY, const float* W, float* dX, const float* relu_mask,
                 const unsigned char* relu_bits, hipStream_t s) {
  Params p;
  memset(&p, 0, sizeof(p));
  p.X = dY; p.W = W; p.mask = relu_mask; p.bits_in = relu_bits; p.Y = dX; p.n_img = g->n_img;
  p.x_bytes = (long long)g->n_img * 81 * 32 * 4; p.y_bytes = (long long)g->n_img * 400 * 16 * 4;
  return launch_k<Dgr>(agx_dg_kernel, p, s, "agx_dg_kernel");
}

}  // namespace agx
}  // namespace seedhip

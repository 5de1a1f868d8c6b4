// Data gradient of the DQN torso's second convolution -- Conv2D(64, 4, 2, 'valid') on the 20 x 20 x 32 map
// (/root/reference/atari/networks.py:233-252) -- on the bf16 matrix pipe (cgx.h's machine; fp32 MFMA before: 1.33 ms per
// cfg5 step).
//     dX[2q + py, 2r + px, ci] = sum_{j, i in {0, 1}} sum_co dY[q - j, r - i, co] W[py + 2j, px + 2i, ci, co]
// : four stride-PARITY classes (py, px), each a 2 x 2-tap convolution of dY (9 x 9 x 64, zero-padded by one) onto a
// 10 x 10 grid with its own quarter of the kernel.  The eight waves are (class) x (half of the 64 dY channels): a wave
// holds its class's weights for its half (8 steps of 16: tap x 16-channel sub-block, 96 registers), every wave
// multiplies every 32-pixel tile of the class grid -- the four classes read the SAME dY pixels --, and the two halves of a
// class exchange two accumulator quads each through LDS (2 KB per wave and tile), so every wave finishes and stores two
// quads (8 of the 32 input channels x 32 pixels... 16 bytes per lane and quad).  Staging, planes, waits: cgx.h.
#pragma once
#include "cgx.h"

namespace seedhip {
namespace cgx {

struct Dg2 {
  static constexpr int G = 2, IH = 9, IW = 9, PAD = 1, IHP = 11, IWP = 11, SLOTS = G * IHP * IWP;
  static constexpr int CBP = (((SLOTS * 16 + 63) / 64) | 1) * 64, XPL = 8 * CBP, XBYTES = 3 * XPL;
  static constexpr int EXSLOT = 8 * 2048, LDS = XBYTES + 2 * EXSLOT + 64;
  static constexpr int ITEMS = G * IH * IW * 16, NXI = (ITEMS + 511) / 512;
  static constexpr int GH = 10, GW = 10, NP = G * GH * GW, T = (NP + 31) / 32;   // class grid
  static constexpr int XH = 20, XW = 20, XC = 32;                                 // dX map
};

__global__ void __launch_bounds__(512, 2)
cgx_dg2_kernel(const Params p) {
  typedef Dg2 G;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cls = wave & 3, kh = wave >> 2, py = cls >> 1, pxc = cls & 1;
  const int px = lane & 31, kb = lane >> 5;
  const int u0 = blockIdx.x * p.per_wg;
  int u1 = u0 + p.per_wg; if (u1 > p.units) u1 = p.units;
  if (u0 >= u1) return;

  for (int i = tid * 16; i < G::XBYTES; i += 512 * 16) *reinterpret_cast<u32x4_t*>(smem + i) = u32x4_t{0u, 0u, 0u, 0u};
  const __amdgpu_buffer_rsrc_t xr = gemm::make_view(p.X, p.x_bytes), yr = gemm::make_view(p.Y, p.y_bytes);
  const __amdgpu_buffer_rsrc_t mr = gemm::make_view(p.mask ? p.mask : p.Y, p.mask ? p.y_bytes : 0);
  const bool has_mask = p.mask != nullptr;

  // ---- weights: rows = the 32 input channels of W; step s = 2 (2 j + i) + sub: tap (py + 2 j, px + 2 i), output channels
  //      32 kh + 16 sub + 8 kb .. + 7 of W (eight consecutive floats) -------------------------------------------------- //
  bf16x8_t wh[8], wm[8], wl[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    const int j = s >> 2, i = (s >> 1) & 1, sub = s & 1;
    const float* src = p.W + (((py + 2 * j) * 4 + pxc + 2 * i) * 32 + px) * 64 + 32 * kh + 16 * sub + 8 * kb;
    const f32x4_t v0 = *reinterpret_cast<const f32x4_t*>(src), v1 = *reinterpret_cast<const f32x4_t*>(src + 4);
    const float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
    u32x4_t h, m, l;
#pragma unroll
    for (int e = 0; e < 4; ++e) { unsigned a, b, c; xg::split2(v[2 * e], v[2 * e + 1], a, b, c); h[e] = a; m[e] = b; l[e] = c; }
    wh[s] = __builtin_bit_cast(bf16x8_t, h); wm[s] = __builtin_bit_cast(bf16x8_t, m); wl[s] = __builtin_bit_cast(bf16x8_t, l);
  }

  // ---- staging of dY (cgx.h): item i = ti + 512 j = quad q of pixel i / 16 of the unit's two images ------------------ //
  const int ti = (tid & ~63) + ((lane >> 2) & 3) * 16 + 4 * (lane >> 4) + (lane & 3);
  auto item_dst = [&](int j) -> unsigned {                   // (recomputed where it is used: registers; `tv` pinned against hoisting)
    int tv = ti;
    asm volatile("" : "+v"(tv));
    const unsigned i = (unsigned)tv + 512u * j, pix = i >> 4, q = i & 15u;
    const unsigned img = pix / 81u, rem = pix - img * 81u, r = rem / 9u, c = rem - r * 9u;
    return i < (unsigned)G::ITEMS ? (q >> 1) * G::CBP + ((img * G::IHP + r + 1) * G::IWP + c + 1) * 16u + (q & 1u) * 8u : kOut;
  };
  const unsigned i16 = (unsigned)ti * 16u;
  f32x4_t lx[G::NXI];
  auto issue_x = [&](int u, int j, bool more) __attribute__((always_inline)) {
    const unsigned off = 8192u * (unsigned)j + i16;
    const long long img0 = (long long)u * G::G;
    const unsigned lim = (unsigned)(((long long)p.n_img - img0 < G::G ? (long long)p.n_img - img0 : G::G) * (81 * 256));
    const unsigned voff = (more && off < lim) ? (unsigned)(img0 * (81 * 256)) + off : kOut;
    asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(lx[j]) : "v"(voff), "s"(xr));
  };
  constexpr int kQueue = 2 * G::T;                           // per unit, behind the staging requests: two output quads per tile
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

  // ---- this lane's pixels: tile t, class-grid pixel P = 32 t + px of the unit = (image, q, r) ------------------------ //
  unsigned pb[G::T];
#pragma unroll
  for (int t = 0; t < G::T; ++t) {
    int P = 32 * t + px; if (P >= G::NP) P = 0;
    const int img = P / (G::GH * G::GW), rem = P - img * (G::GH * G::GW), q = rem / G::GW, r = rem - q * G::GW;
    pb[t] = (unsigned)((4 * kh + kb) * G::CBP + ((img * G::IHP + q) * G::IWP + r) * 16);     // dY[q - 1, r - 1] in padded slots
  }
  auto out_off = [&](int t) -> unsigned {                    // (recomputed per tile, pinned like item_dst)
    int pv = px;
    asm volatile("" : "+v"(pv));
    const unsigned P = 32u * t + (unsigned)pv;
    const unsigned img = P / (unsigned)(G::GH * G::GW), rem = P - img * (G::GH * G::GW), q = rem / (unsigned)G::GW, r = rem - q * G::GW;
    return (((img * G::XH + 2 * q + py) * G::XW + 2 * r + pxc) * G::XC + 16 * kh + 4 * kb) * 4u;
  };
  unsigned char* exs = smem + G::XBYTES;
  unsigned parity = 0;

  auto quad = [](const f32x16_t& a, int g) -> f32x4_t { return f32x4_t{a[4 * g], a[4 * g + 1], a[4 * g + 2], a[4 * g + 3]}; };
  auto compute = [&](auto KH, int u) __attribute__((always_inline)) {
    constexpr int kH = decltype(KH)::value;                  // this wave finishes quads 2 kH, 2 kH + 1 (input channels 16 kH + 8 g' + 4 kb ..)
    long long left = ((long long)p.n_img - (long long)u * G::G) * (G::GH * G::GW);
    const int npx = left < G::NP ? (int)left : G::NP;
    const unsigned ys = (unsigned)((long long)u * G::G * (G::XH * G::XW * G::XC * 4));
#pragma unroll
    for (int t = 0; t < G::T; ++t) {
      const unsigned o0 = (32 * t + px) < npx ? out_off(t) : kOut;
      f32x16_t acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      bf16x8_t xv[2][3];
      auto fetch = [&](int s, bf16x8_t (&x)[3]) {              // step s: tap (j, i) reads dY[q - j, r - i]; sub-block 2 sub (+ kb in pb)
        const int j = s >> 2, i = (s >> 1) & 1, sub = s & 1;
        const int off = ((1 - j) * G::IWP + (1 - i)) * 16 + 2 * sub * G::CBP;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) x[pl] = *reinterpret_cast<const bf16x8_t*>(smem + pb[t] + off + pl * G::XPL);
      };
      fetch(0, xv[0]);
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        if (s + 1 < 8) fetch(s + 1, xv[(s + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
        const bf16x8_t (&x)[3] = xv[s & 1];
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl[s], x[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[s], x[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wm[s], x[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wm[s], x[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[s], x[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[s], x[0], acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      // (the mask quads are requested here, behind the MFMAs: eight registers the loop does not have; the barrier hides them)
      f32x4_t mk[2];
      if (has_mask) {
#pragma unroll
        for (int g = 0; g < 2; ++g) mk[g] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(mr, o0 == kOut ? kOut : o0 + 32u * g, ys, 0));
      }
      // exchange slot: [class][receiving half][two quads][lane] x 16 bytes
      unsigned char* slot = exs + parity * G::EXSLOT + cls * 4096 + lane * 16;
#pragma unroll
      for (int g = 0; g < 2; ++g) *reinterpret_cast<f32x4_t*>(slot + (1 - kH) * 2048 + g * 1024) = quad(acc, 2 * (1 - kH) + g);
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      f32x4_t o[2];
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const f32x4_t own = quad(acc, 2 * kH + g), got = *reinterpret_cast<const f32x4_t*>(slot + kH * 2048 + g * 1024);
#pragma unroll
        for (int e = 0; e < 4; ++e) o[g][e] = kH == 0 ? own[e] + got[e] : got[e] + own[e];   // (channel half 0's part first, whoever adds)
        if (has_mask) {
#pragma unroll
          for (int e = 0; e < 4; ++e) o[g][e] = mk[g][e] > 0.f ? o[g][e] : 0.f;
        }
        asm volatile("" : "+v"(o[g]));
      }
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, o[g]), yr, o0 == kOut ? kOut : o0 + 32u * g, ys, 0);
        asm volatile("s_nop 1" ::: "memory");
      }
      parity ^= 1u;
    }
  };
  auto run = [&](auto KH) __attribute__((always_inline)) {
    auto step = [&](int u, auto first) __attribute__((always_inline)) {
      put(u + 1, first);
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      compute(KH, u);
    };
    step(u0, std::true_type());
    for (int u = u0 + 1; u < u1; ++u) step(u, std::false_type());
  };
#pragma unroll
  for (int s = 0; s < 8; ++s) asm volatile("" :: "v"(wh[s]), "v"(wm[s]), "v"(wl[s]));   // (weights finished before the first requests: cgx.h)
#pragma unroll
  for (int j = 0; j < G::NXI; ++j) issue_x(u0, j, true);
  __syncthreads();                                           // LDS zeroed
  if (kh == 0) run(std::integral_constant<int, 0>()); else run(std::integral_constant<int, 1>());
}

bool plan_dgrad2(const seedhip_conv_geom* g) {
  if (!geometry2(g) || g->n_img < 512) return false;
  return (long long)g->n_img * 400 * 32 * 4 < (1LL << 31) - (1 << 22);
}
int launch_dgrad2(const seedhip_conv_geom* g, const float* dY, const float* W, float* dX, const float* relu_mask, hipStream_t s) {
  Params p;
  memset(&p, 0, sizeof(p));
  p.X = dY; p.W = W; p.mask = relu_mask; p.Y = dX; p.n_img = g->n_img;
  p.x_bytes = (long long)g->n_img * 81 * 64 * 4; p.y_bytes = (long long)g->n_img * 400 * 32 * 4;
  static const int cus = xg::cu_count();
  p.units = (p.n_img + Dg2::G - 1) / Dg2::G;
  int grid = p.units < cus ? p.units : cus;
  p.per_wg = (p.units + grid - 1) / grid;
  grid = (p.units + p.per_wg - 1) / p.per_wg;
  static const bool ok = hipFuncSetAttribute((const void*)cgx_dg2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, Dg2::LDS) == hipSuccess;
  if (!ok) return fail(SEEDHIP_ERR_LAUNCH, "cgx_dg2_kernel: LDS attribute");
  hipLaunchKernelGGL(cgx_dg2_kernel, dim3(grid), dim3(512), Dg2::LDS, s, p);
  return check_launch("cgx_dg2_kernel");
}

}  // namespace cgx
}  // namespace seedhip

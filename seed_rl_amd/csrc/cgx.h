// The DQN torso's third convolution -- Conv2D(64, 3, 1, 'valid') on the 9 x 9 x 64 map behind the second conv
// (/root/reference/atari/networks.py:233-252: the conv body of DuelingLSTMDQNNet) -- forward and data gradient on the BF16
// matrix pipe through the exact three-way operand split (xgemm.h: "bf16x6").  It ran on the fp32-MFMA gather GEMM
// (gemm.h; 2.17 + 0.83 ms of a cfg5 step at 0.65 of that pipe's peak).
//
// fgx.h's machine for 64 input channels.  There a wave holds ALL weights of its 32 output channels (9 taps x 16 input
// channels: 108 registers); with 64 input channels a wave can hold a QUARTER of the reduction: the eight waves of a
// workgroup are (output-channel half) x (16-channel block of the input), every wave multiplies every pixel tile against
// its 108 registers, and the four partial sums of a tile meet in LDS: every wave owns one quad of the tile's accumulator
// registers and receives the other three waves' partial sums of that quad (1 KB each), one barrier, it adds them in a
// fixed order, applies the epilogue and stores 16 bytes per lane (two exchange slots: the next tile's sums go to the
// other one).
//   * unit = G whole images (the maps are 9 x 9 / 7 x 7: a band would be a fraction of a tile): their pixels are one run
//     of G x OH x OW consecutive output pixels in HBM, cut into 32-pixel tiles wherever they fall; inputs = contiguous
//     16-byte items requested one unit ahead (asm loads, hand-counted `s_waitcnt`, fgx.h), split once per element by
//     truncation into three bf16 planes [8-channel block][image][row + PAD][column + PAD][8];
//   * the data gradient is the same convolution of dY over a map zero-padded by 2 (the pad slots are zeroed once and never
//     written), weights read flipped and transposed, ReLU mask of the layer's input in the epilogue;
//   * blocks of a plane an ODD multiple of 64 bytes apart: a `ds_write_b64` group of the staging holds four pixels of two
//     neighbouring blocks (16 + 16 banks), a `ds_read_b128` group sixteen pixels of one block.
#pragma once
#include <type_traits>
#include "common.h"
#include "xgemm.h"
#include "cgx_api.h"

namespace seedhip {
namespace cgx {

using xg::bf16x8_t;
using xg::f32x4_t;
using xg::f32x16_t;
using xg::u32x2_t;
using xg::u32x4_t;
constexpr unsigned kOut = 0x80000000u;

// KS = kernel size (3, stride 1 | 4, stride 2), CIN = 64 | 32 input channels, 64 output channels.  A wave's share of the
// reduction is CIN / 4 channels of every tap, in 16-deep steps: one tap x 16 channels (CIN = 64: the lane's half kb
// picks the 8-channel block) or two taps x 8 channels (CIN = 32: kb picks the tap).  Stride 2 stores a row's even and odd
// columns apart ([row][parity][column / 2]): consecutive output pixels of a tap then read consecutive slots.
template <int KS_, int CIN_, int IH_, int IW_, int OH_, int OW_, int PAD_, int G_, bool DG_>
struct Geo {
  static constexpr int KS = KS_, S = KS_ == 4 ? 2 : 1, CIN = CIN_, IH = IH_, IW = IW_, OH = OH_, OW = OW_, PAD = PAD_, G = G_, C = 64;
  static constexpr bool DG = DG_;
  static constexpr int TAPS = KS * KS, NSTEP = TAPS * (CIN / 4) / 16, QP = CIN / 4, NBLK = CIN / 8;
  static constexpr int IHP = IH + 2 * PAD, IWP = IW + 2 * PAD, SLOTS = G * IHP * IWP, HW = IWP / 2;
  static constexpr int CB64 = (SLOTS * 16 + 63) / 64;
  static constexpr int CBP = (CB64 | 1) * 64;                  // bytes of one 8-channel block of a plane: odd multiple of 64
  static constexpr int XPL = NBLK * CBP, XBYTES = 3 * XPL;
  static_assert((KS == 3 && CIN == 64) || (KS == 4 && CIN == 32 && IWP % 2 == 0 && PAD == 0 && !DG), "served forms");
  // slot of padded pixel (row, col) of image img; byte offset of tap (ky, kx) relative to a pixel's tap (0, 0)
  static constexpr int slot(int img, int row, int col) { return (img * IHP + row) * IWP + (S == 2 ? (col & 1) * HW + (col >> 1) : col); }
  static constexpr int tap_off(int ky, int kx) { return (ky * IWP + (S == 2 ? (kx & 1) * HW + (kx >> 1) : kx)) * 16; }
  static constexpr int EXW = 4096, EXSLOT = 2 * 3 * EXW;      // (channel half) x (owner quad x three senders x 1 KB)
  static constexpr int LDS = XBYTES + 2 * EXSLOT + 256 + 64;  // + the bias + a dump for the items past a unit
  static constexpr int IPX = G * IH * IW, NP = G * OH * OW, T = (NP + 31) / 32;
  static constexpr int ITEMS = IPX * QP, NXI = (ITEMS + 511) / 512;
  static_assert(OH == (IH + 2 * PAD - KS) / S + 1 && OW == (IW + 2 * PAD - KS) / S + 1, "valid convolution of the padded map");
  static_assert(LDS <= 160 * 1024, "one 8-wave workgroup per CU");
};

struct Params {
  const float* X; const float* W; const float* bias; const float* mask; float* Y;
  int n_img, units, per_wg, out_relu;
  long long x_bytes, y_bytes;
};

template <class G>
__global__ void __launch_bounds__(512, 2)
cgx_kernel(const Params p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ch = wave & 1, kq = wave >> 1;                   // output-channel half; 16-channel block of the reduction
  const int px = lane & 31, kb = lane >> 5;
  const int u0 = blockIdx.x * p.per_wg;
  int u1 = u0 + p.per_wg; if (u1 > p.units) u1 = p.units;
  if (u0 >= u1) return;

  for (int i = tid * 16; i < G::XBYTES; i += 512 * 16) *reinterpret_cast<u32x4_t*>(smem + i) = u32x4_t{0u, 0u, 0u, 0u};
  const __amdgpu_buffer_rsrc_t xr = gemm::make_view(p.X, p.x_bytes), yr = gemm::make_view(p.Y, p.y_bytes);
  const __amdgpu_buffer_rsrc_t mr = gemm::make_view(p.mask ? p.mask : p.Y, p.mask ? p.y_bytes : 0);
  const bool has_mask = G::DG && p.mask != nullptr;

  // ---- weights: rows = this wave's 32 output channels, reduction elements 8 kb .. 8 kb + 7 of its 16-channel block ---- //
  bf16x8_t wh[G::NSTEP], wm[G::NSTEP], wl[G::NSTEP];
#pragma unroll
  for (int t = 0; t < G::NSTEP; ++t) {
    float v[8];
    if (G::DG) {                                             // W[8 - t][row = ci of W][reduction = co of W]: eight consecutive
      const float* src = p.W + ((8 - t) * 64 + 32 * ch + px) * 64 + 16 * kq + 8 * kb;
      const f32x4_t v0 = *reinterpret_cast<const f32x4_t*>(src), v1 = *reinterpret_cast<const f32x4_t*>(src + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[e] = v0[e]; v[4 + e] = v1[e]; }
    } else if (G::CIN == 64) {                               // W[t][reduction = ci][row = co]
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = p.W[(t * 64 + 16 * kq + 8 * kb + e) * 64 + 32 * ch + px];
    } else {                                                 // step t = taps 2 t, 2 t + 1 (kb) x the wave's 8 channels
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = p.W[((2 * t + kb) * 32 + 8 * kq + e) * 64 + 32 * ch + px];
    }
    u32x4_t h, m, l;
#pragma unroll
    for (int e = 0; e < 4; ++e) { unsigned a, b, c; xg::split2(v[2 * e], v[2 * e + 1], a, b, c); h[e] = a; m[e] = b; l[e] = c; }
    wh[t] = __builtin_bit_cast(bf16x8_t, h); wm[t] = __builtin_bit_cast(bf16x8_t, m); wl[t] = __builtin_bit_cast(bf16x8_t, l);
  }
  // accumulator register r of a lane: output channel 32 ch + 8 (r / 4) + 4 kb + r % 4 of pixel px; the bias waits in LDS
  // for the wave that finishes a tile (sixteen registers the forward does not have)
  float* bias_lds = reinterpret_cast<float*>(smem + G::XBYTES + 2 * G::EXSLOT);
  if (tid < 64) bias_lds[tid] = (!G::DG && p.bias) ? p.bias[tid] : 0.f;

  // ---- staging: item i = ti + 512 j = quad q of input pixel i / 16 of the unit (contiguous in HBM) ------------------ //
  // lanes permuted inside each 64-item chunk (four pixels x 16 quads): a 16-lane group holds quads 4 g .. 4 g + 3 (two
  // neighbouring blocks) of the four pixels
  const int ti = (tid & ~63) + (((lane >> 4) / (G::QP / 4)) * 4 + ((lane >> 2) & 3)) * G::QP + 4 * ((lane >> 4) % (G::QP / 4)) + (lane & 3);
  // LDS offset of item j, recomputed where it is used (eight registers the forward does not have; `tv` is pinned so that
  // the compiler does not hoist the table back out of the unit loop)
  // (a register table where the kernel has registers left: the second conv's forward, the third conv's data gradient)
  constexpr bool kTable = G::NSTEP < 9 || G::DG;
  auto item_dst = [&](int j) -> unsigned {
    int tv = ti;
    if (!kTable) asm volatile("" : "+v"(tv));
    const unsigned i = (unsigned)tv + 512u * j, pix = i / (unsigned)G::QP, q = i % (unsigned)G::QP;
    const unsigned img = pix / (unsigned)(G::IH * G::IW), rem = pix - img * (G::IH * G::IW), r = rem / (unsigned)G::IW, c = rem - r * G::IW;
    const unsigned col = c + G::PAD, sl = (img * G::IHP + r + G::PAD) * G::IWP + (G::S == 2 ? (col & 1u) * G::HW + (col >> 1) : col);
    const unsigned dst = (q >> 1) * G::CBP + sl * 16u + (q & 1u) * 8u;
    return i < (unsigned)G::ITEMS ? dst : kOut;
  };
  const unsigned i16 = (unsigned)ti * 16u;
  f32x4_t lx[G::NXI];
  auto issue_x = [&](int u, int j, bool more) __attribute__((always_inline)) {
    const unsigned off = 8192u * (unsigned)j + i16;           // inside the unit's G images
    const long long img0 = (long long)u * G::G;
    const unsigned lim = (unsigned)(((long long)p.n_img - img0 < G::G ? (long long)p.n_img - img0 : G::G) * (G::IH * G::IW * G::CIN * 4));
    const bool in = more && off < lim;
    const unsigned voff = in ? (unsigned)(img0 * (G::IH * G::IW * G::CIN * 4)) + off : kOut;
    asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(lx[j]) : "v"(voff), "s"(xr));
  };

  // ---- this lane's pixels: tile t, pixel P = 32 t + px of the unit's run ------------------------------------------ //
  unsigned pb[G::T];
#pragma unroll
  for (int t = 0; t < G::T; ++t) {
    int P = 32 * t + px; if (P >= G::NP) P = 0;              // (past the run: any slot; its outputs are not stored)
    const int img = P / (G::OH * G::OW), rem = P - img * (G::OH * G::OW), oy = rem / G::OW, ox = rem - oy * G::OW;
    pb[t] = G::CIN == 64 ? (unsigned)((2 * kq + kb) * G::CBP + G::slot(img, oy, ox) * 16)
                         : (unsigned)(kq * G::CBP + G::slot(img, G::S * oy, G::S * ox) * 16 + G::tap_off(0, kb));
  }
  unsigned char* exs = smem + G::XBYTES;
  unsigned parity = 0;                                       // exchange slot of the next tile

  // Every wave finishes ONE quad of its tile: quad g (output channels 32 ch + 8 g + 4 kb ..) belongs to the wave with
  // kq = g, the other three send it their partial sums of that quad (1 KB each) -- the same LDS traffic as one finishing
  // wave per tile, but no wave has four times the epilogue of the others in front of the next barrier.
  constexpr int kQueue = G::T;                               // per unit, behind the staging requests: one output quad per tile
  // (the data gradient's mask quads come on top when there is a mask: counting fewer only waits a little longer)
  auto put = [&](int un, auto first) __attribute__((always_inline)) {
    const bool more = un < u1;
#pragma unroll
    for (int j = 0; j < G::NXI; ++j) {
      const f32x4_t it = decltype(first)::value ? xg::take_item<G::NXI - 1>(lx[j]) : xg::take_item<G::NXI - 1 + kQueue>(lx[j]);
      {
        // branch free (an item past the unit was requested out of range -- zeros -- and lands in a pad nobody reads): a
        // use under a branch made the compiler COPY the item's registers in front of the wait, i.e. before the data
        unsigned h0, m0, l0, h1, m1, l1;
        xg::split2_trunc(it[0], it[1], h0, m0, l0);
        xg::split2_trunc(it[2], it[3], h1, m1, l1);
        const unsigned dst = item_dst(j);
        constexpr unsigned kDump = (unsigned)(G::LDS - 64);
        *reinterpret_cast<u32x2_t*>(smem + (dst == kOut ? kDump : dst)) = u32x2_t{h0, h1};
        *reinterpret_cast<u32x2_t*>(smem + (dst == kOut ? kDump + 16u : dst + G::XPL)) = u32x2_t{m0, m1};
        *reinterpret_cast<u32x2_t*>(smem + (dst == kOut ? kDump + 32u : dst + 2 * G::XPL)) = u32x2_t{l0, l1};
      }
      issue_x(un, j, more);
    }
  };
  auto quad = [](const f32x16_t& a, int g) -> f32x4_t { return f32x4_t{a[4 * g], a[4 * g + 1], a[4 * g + 2], a[4 * g + 3]}; };
  auto compute = [&](auto KQ, int u) __attribute__((always_inline)) {
    constexpr int kQ = decltype(KQ)::value;                  // (a copy of the loop per kq: the quads are register indices)
    long long left = ((long long)p.n_img - (long long)u * G::G) * (G::OH * G::OW);
    const int npx = left < G::NP ? (int)left : G::NP;          // output pixels of this unit
    const unsigned ys = (unsigned)((long long)u * G::NP * 256);
#pragma unroll
    for (int t = 0; t < G::T; ++t) {
      const int P = 32 * t + px;
      const unsigned ob = P < npx ? (unsigned)(P * 64 + 32 * ch + 8 * kQ + 4 * kb) * 4u : kOut;
      f32x4_t mk;
      if (G::DG && has_mask) mk = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(mr, ob, ys, 0));
      f32x16_t acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      bf16x8_t xv[2][3];
      auto fetch = [&](int st, bf16x8_t (&x)[3]) {             // step st: tap st (CIN 64) or taps 2 st + kb (CIN 32; kb's part is in pb)
        const int off = G::CIN == 64 ? G::tap_off(st / G::KS, st % G::KS) : G::tap_off((2 * st) / G::KS, (2 * st) % G::KS);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) x[pl] = *reinterpret_cast<const bf16x8_t*>(smem + pb[t] + off + pl * G::XPL);
      };
      fetch(0, xv[0]);
#pragma unroll
      for (int tap = 0; tap < G::NSTEP; ++tap) {
        if (tap + 1 < G::NSTEP) fetch(tap + 1, xv[(tap + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
        const bf16x8_t (&x)[3] = xv[tap & 1];
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl[tap], x[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[tap], x[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wm[tap], x[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wm[tap], x[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[tap], x[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[tap], x[0], acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      // slot: [channel half][owner quad][sender 0 .. 2][lane] x 16 bytes; sender index = kq of the sender, minus one above the owner
      unsigned char* slot = exs + parity * G::EXSLOT + ch * (3 * G::EXW) + lane * 16;
#pragma unroll
      for (int g = 0; g < 4; ++g)
        if (g != kQ) *reinterpret_cast<f32x4_t*>(slot + g * 3072 + (kQ < g ? kQ : kQ - 1) * 1024) = quad(acc, g);
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      // own quad + the other three quarters of the reduction in a fixed order, the epilogue, one 16-byte store
      f32x4_t o = quad(acc, kQ);
      if (!G::DG) {
        const f32x4_t b4 = *reinterpret_cast<const f32x4_t*>(bias_lds + 32 * ch + 8 * kQ + 4 * kb);
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] += b4[e];
      }
#pragma unroll
      for (int w = 0; w < 3; ++w) {
        const f32x4_t q4 = *reinterpret_cast<const f32x4_t*>(slot + kQ * 3072 + w * 1024);
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] += q4[e];
      }
      if (G::DG && has_mask) {
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = mk[e] > 0.f ? o[e] : 0.f;
      }
      if (!G::DG && p.out_relu) {
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = fmaxf(o[e], 0.f);
      }
      asm volatile("" : "+v"(o));
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, o), yr, ob, ys, 0);
      asm volatile("s_nop 1" ::: "memory");
      parity ^= 1u;
    }
  };
  auto run = [&](auto KQ) __attribute__((always_inline)) {
    auto step = [&](int u, auto first) __attribute__((always_inline)) {
      put(u + 1, first);
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      compute(KQ, u);                                        // (its last barrier also frees the planes for the next put)
    };
    step(u0, std::true_type());
    for (int u = u0 + 1; u < u1; ++u) step(u, std::false_type());
  };

  // The weights are finished BEFORE the first requests go out: a request's destination registers are written when the
  // data arrives, so nothing between the request and its `s_waitcnt` may make the compiler move or spill them -- with the
  // weight preparation (its temporaries, 253 registers in use) scheduled behind the requests it did, and the first
  // unit's first items were read from the copies.
#pragma unroll
  for (int t = 0; t < G::NSTEP; ++t) asm volatile("" :: "v"(wh[t]), "v"(wm[t]), "v"(wl[t]));
#pragma unroll
  for (int j = 0; j < G::NXI; ++j) issue_x(u0, j, true);
  __syncthreads();                                           // LDS zeroed
  if (kq == 0) run(std::integral_constant<int, 0>());
  else if (kq == 1) run(std::integral_constant<int, 1>());
  else if (kq == 2) run(std::integral_constant<int, 2>());
  else run(std::integral_constant<int, 3>());
}

// ---- served geometries ------------------------------------------------------------------------------------------ //
typedef Geo<3, 64, 9, 9, 7, 7, 0, 3, false> GeoFwd;        // conv 3: three images per unit: 147 pixels = five tiles
typedef Geo<3, 64, 7, 7, 9, 9, 2, 2, true> GeoDgrad;       // its data gradient: dY 7 x 7 zero-padded to 11 x 11, two images = six tiles
typedef Geo<4, 32, 20, 20, 9, 9, 0, 1, false> GeoFwd2;     // conv 2 (4 x 4 / 2, 32 -> 64): one 20 x 20 image per unit: 81 pixels = three tiles

inline bool geometry(const seedhip_conv_geom* g) {
  return g->kh == 3 && g->kw == 3 && g->stride == 1 && g->pad_t == 0 && g->pad_l == 0 && g->cin == 64 && g->cout == 64 &&
         g->ih == 9 && g->iw == 9 && g->oh == 7 && g->ow == 7 && g->ld_in == 64 && g->ld_out == 64;
}
inline bool geometry2(const seedhip_conv_geom* g) {
  return g->kh == 4 && g->kw == 4 && g->stride == 2 && g->pad_t == 0 && g->pad_l == 0 && g->cin == 32 && g->cout == 64 &&
         g->ih == 20 && g->iw == 20 && g->oh == 9 && g->ow == 9 && g->ld_in == 32 && g->ld_out == 64;
}
bool plan(const seedhip_conv_geom* g) {
  if (!geometry(g) || g->n_img < 1024) return false;
  return (long long)g->n_img * 81 * 64 * 4 < (1LL << 31) - (1 << 22);
}
bool plan_fwd2(const seedhip_conv_geom* g) {
  if (!geometry2(g) || g->n_img < 512) return false;
  return (long long)g->n_img * 400 * 32 * 4 < (1LL << 31) - (1 << 22);
}

template <class G>
inline int launch_geo(Params& p, hipStream_t s) {
  static const int cus = xg::cu_count();
  p.units = (p.n_img + G::G - 1) / G::G;
  int grid = p.units < cus ? p.units : cus;
  p.per_wg = (p.units + grid - 1) / grid;
  grid = (p.units + p.per_wg - 1) / p.per_wg;
  static const bool ok = hipFuncSetAttribute((const void*)cgx_kernel<G>, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS) == hipSuccess;
  if (!ok) return fail(SEEDHIP_ERR_LAUNCH, "cgx_kernel: LDS attribute");
  hipLaunchKernelGGL((cgx_kernel<G>), dim3(grid), dim3(512), G::LDS, s, p);
  return check_launch("cgx_kernel");
}

int launch_fwd(const seedhip_conv_geom* g, const float* X, const float* W, const float* bias, float* Y, int out_relu, hipStream_t s) {
  Params p;
  memset(&p, 0, sizeof(p));
  p.X = X; p.W = W; p.bias = bias; p.Y = Y; p.n_img = g->n_img; p.out_relu = out_relu;
  if (geometry2(g)) {
    p.x_bytes = (long long)g->n_img * 400 * 32 * 4; p.y_bytes = (long long)g->n_img * 81 * 64 * 4;
    return launch_geo<GeoFwd2>(p, s);
  }
  p.x_bytes = (long long)g->n_img * 81 * 64 * 4; p.y_bytes = (long long)g->n_img * 49 * 64 * 4;
  return launch_geo<GeoFwd>(p, s);
}
int launch_dgrad(const seedhip_conv_geom* g, const float* dY, const float* W, float* dX, const float* relu_mask, hipStream_t s) {
  Params p;
  memset(&p, 0, sizeof(p));
  p.X = dY; p.W = W; p.mask = relu_mask; p.Y = dX; p.n_img = g->n_img;
  p.x_bytes = (long long)g->n_img * 49 * 64 * 4; p.y_bytes = (long long)g->n_img * 81 * 64 * 4;
  return launch_geo<GeoDgrad>(p, s);
}

}  // namespace cgx
}  // namespace seedhip

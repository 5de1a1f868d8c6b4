// Forward / data gradient of ImpalaDeep's stack-entry convolution on the 36 x 48 map -- Conv2D(32, 3, 'same') on the 16
// channels behind the first pool (/root/reference/dmlab/networks.py:31-37: `conv_out = Conv2D(num_ch, 3, padding='same')`
// opening every stack) and the gradient TensorFlow derives for its input -- on the BF16 matrix pipe through the exact
// three-way operand split (xgemm.h: "bf16x6").  The other 3 x 3 layers have wsx.h / wsy.h (32 -> 32 and 16 -> 16: those
// hard-code their channel counts); this layer ran on the fp32-MFMA halo kernels (0.99 + 1.02 ms per cfg3 step, r5
// rocprofv3), the last convolution of the step on that pipe.
//
// Built on wgx.h's staging, not on wfx.h's ring: a UNIT is a band of BR output rows of one image; its BR + 2 input rows
// (rows outside the image requested out of range: zeros) are contiguous 16-byte items, loaded into registers one unit
// ahead, split ONCE per element by truncation and written as three bf16 planes [channel block of 8][row][x + 1][8]
// (16 bytes per pixel and block: the 16 lanes of a ds_read_b128 group take 16 consecutive pixels = 256 consecutive
// bytes, no bank conflict; the pad columns are zeroed once).  MFMA operands SWAPPED as in the other conv kernels: the
// weights are the instruction's rows (one tap = one reduction step: 16 input channels on v_mfma_f32_32x32x16_bf16 for
// the forward, 32 on v_mfma_f32_16x16x32_bf16 for the data gradient), split once in the prologue into 108 registers
// per lane (round to nearest), the pixels its columns: a lane ends up with four consecutive output channels of one
// pixel per accumulator quad and stores them as 16 bytes.  Each wave owns TPW pixel tiles of the band; the pixel operands
// of tap t + 1 are requested before the six MFMAs of tap t; two barriers per unit, two 4-wave workgroups per CU.
// DG = data gradient: dX[p, ci] = sum_{tap, co} dY[p + 1 - tap, co] W[tap, ci, co] is the same convolution of dY with the
// kernel flipped and transposed, read straight out of W (eight consecutive co per lane).
#pragma once
#include "common.h"
#include "xgemm.h"
#include "fgx_api.h"
#include <type_traits>

namespace seedhip {
namespace fgx {

using xg::bf16x8_t;
using xg::f32x4_t;
using xg::f32x16_t;
using xg::u32x2_t;
using xg::u32x4_t;
constexpr unsigned kOut = 0x80000000u;

template <int CIN_, int COUT_, int IH_, int IW_, int BR_, int SUB_, bool DG_>
struct Geo {
  static constexpr int CIN = CIN_, COUT = COUT_, IH = IH_, IW = IW_, BR = BR_, SUB = SUB_;
  static constexpr bool DG = DG_;
  static constexpr bool M32 = COUT == 32;                     // 32 x 32 x 16 (K = 16 = CIN) or 16 x 16 x 32 (K = 32 = CIN)
  static_assert((M32 && CIN == 16) || (!M32 && COUT == 16 && CIN == 32), "one tap per reduction step");
  static_assert(CIN == 16 || CIN == 32, "lane permutation of the staging");
  static constexpr int PT = M32 ? 32 : 16;                    // pixels per tile
  static constexpr int ACCN = M32 ? 16 : 4;
  static constexpr int NB = IH / BR;                          // bands per image; a unit = SUB consecutive bands of the batch
  static constexpr int XR = BR + 2, RP = IW + 2;              // input rows of a unit; pixels of a padded LDS row
  static constexpr int CB = CIN / 8;                          // 8-channel blocks
  static constexpr int CBP = (XR * RP * 16 + 255) / 256 * 256;  // bytes of one channel block of one plane (a multiple of 64 banks:
                                                              // the 16 lanes of a read group that span two blocks do not collide)
  static constexpr int SUBP = CB * CBP;                       // one band of one plane
  static constexpr int XPL = SUB * SUBP;                      // one plane
  static constexpr int LDS = 3 * XPL;
  static constexpr int NPIX = BR * IW, BTILES = NPIX / PT, TILES = SUB * BTILES, TPW = TILES / 4;
  static constexpr int BITEMS = XR * IW * CIN / 4, XITEMS = SUB * BITEMS, NXI = (XITEMS + 255) / 256;
  static constexpr int ROWB = IW * CIN * 4, IMGB = IH * ROWB, YIMGB = IH * IW * COUT * 4;
  static_assert(IH % BR == 0 && NPIX % PT == 0 && TILES % 4 == 0 && BTILES % TPW == 0 && SUB <= 2, "tiling");
  static_assert(LDS <= 80 * 1024, "two workgroups per CU");
};

struct Params {
  const float* X; const float* W; const float* bias; float* Y;
  int n_img, bands, units, per_wg;
  long long x_bytes, y_bytes;
  // POOL (data gradient behind the stack's max-pool): X = gradient of the POOLED map [n, IH / 2, IW / 2, CIN], arg = the
  // pool's argmax bytes, DA = the pre-pool gradient [n, IH, IW, CIN] this kernel also writes (the weight gradient reads it)
  const unsigned char* arg; float* DA; long long p_bytes;
};

template <class G, bool M32 = G::M32> struct AccT { typedef f32x16_t type; };
template <class G> struct AccT<G, false> { typedef f32x4_t type; };

template <class G, bool POOL = false>
__global__ void __launch_bounds__(256, 2)
fgx_kernel(const Params p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  typedef typename AccT<G>::type acc_t;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int px = lane % G::PT, kb = lane / G::PT;            // pixel of the tile / 8-channel block of the reduction step
  const int u0 = blockIdx.x * p.per_wg;
  int u1 = u0 + p.per_wg; if (u1 > p.units) u1 = p.units;
  if (u0 >= u1) return;

  for (int i = tid * 16; i < G::LDS; i += 256 * 16) *reinterpret_cast<u32x4_t*>(smem + i) = u32x4_t{0u, 0u, 0u, 0u};
  const __amdgpu_buffer_rsrc_t xr = gemm::make_view(p.X, POOL ? p.p_bytes : p.x_bytes), yr = gemm::make_view(p.Y, p.y_bytes);

  // ---- weights: the instruction's rows.  Lane (row = lane % PT, kb): reduction elements 8 kb .. 8 kb + 7 of tap t ---- //
  bf16x8_t wh[9], wm[9], wl[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    float v[8];
    if (G::DG) {                                             // row = ci of W, reduction = co: W[8 - t][row][8 kb + e]
      const float* src = p.W + ((8 - t) * G::COUT + px) * G::CIN + 8 * kb;     // (COUT here = W's cin, CIN = W's cout)
      const f32x4_t v0 = *reinterpret_cast<const f32x4_t*>(src), v1 = *reinterpret_cast<const f32x4_t*>(src + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[e] = v0[e]; v[4 + e] = v1[e]; }
    } else {                                                 // row = co, reduction = ci: W[t][8 kb + e][row]
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = p.W[(t * G::CIN + 8 * kb + e) * G::COUT + px];
    }
    u32x4_t h, m, l;
#pragma unroll
    for (int e = 0; e < 4; ++e) { unsigned a, b, c; xg::split2(v[2 * e], v[2 * e + 1], a, b, c); h[e] = a; m[e] = b; l[e] = c; }
    wh[t] = __builtin_bit_cast(bf16x8_t, h); wm[t] = __builtin_bit_cast(bf16x8_t, m); wl[t] = __builtin_bit_cast(bf16x8_t, l);
  }
  // accumulator register r of a lane: output channel row(r) of pixel px
  acc_t acc0;
#pragma unroll
  for (int r = 0; r < G::ACCN; ++r) {
    const int row = G::M32 ? 8 * (r >> 2) + 4 * kb + (r & 3) : 4 * kb + r;
    acc0[r] = (!G::DG && p.bias) ? p.bias[row] : 0.f;
  }

  // ---- staging (wgx.h): item i = tid + 256 j of the unit's input rows ------------------------------------------- //
  // Lanes permuted inside each 64-item (1 KB) chunk so that the 16 lanes of a ds_write_b64 group hold the two halves
  // of ONE channel block of 8 consecutive pixels = 32 banks once (in item order they would hold every block of a few
  // pixels, and the blocks are a multiple of 32 banks apart).
  const int lw = lane & 15, lg = lane >> 4;
  const int pxl = G::CB == 4 ? (lw >> 1) : (lg >> 1) * 8 + (lw >> 1);
  const int ti = (tid & ~63) + pxl * (2 * G::CB) + 2 * (G::CB == 4 ? lg : (lg & 1)) + (lw & 1);
  unsigned xdst[G::NXI];
#pragma unroll
  for (int j = 0; j < G::NXI; ++j) {
    const int i = ti + 256 * j;
    constexpr int per_row = G::IW * G::CIN / 4, per_pix = G::CIN / 4;
    const int sb = i / G::BITEMS, ib = i - sb * G::BITEMS;
    const int r = ib / per_row, rem = ib - r * per_row, x = rem / per_pix, q = rem - x * per_pix;
    xdst[j] = (unsigned)(sb * G::SUBP + (q >> 1) * G::CBP + (r * G::RP + x + 1) * 16 + (q & 1) * 8);
  }
  const unsigned i16 = (unsigned)ti * 16u;
  f32x4_t lx[G::NXI];
  // band b of the batch: image b / NB, rows (b % NB) BR - 1 .. + XR - 1 (rows outside the image: requested out of range)
  auto issue_x = [&](int u, int j, bool more) {
    const int b0 = u * G::SUB, b1 = b0 + G::SUB - 1;
    const int img0 = b0 / G::NB, img1 = b1 / G::NB;
    const int row0 = ((b0 - img0 * G::NB) * G::BR - 1) * G::ROWB, row1 = ((b1 - img1 * G::NB) * G::BR - 1) * G::ROWB - G::BITEMS * 16;
    const bool second = G::SUB > 1 && (256 * j >= G::BITEMS || (256 * (j + 1) > G::BITEMS && ti + 256 * j >= G::BITEMS));
    const unsigned off = (unsigned)((second ? row1 : row0) + 4096 * j) + i16;
    const bool in = off < (unsigned)G::IMGB && (j + 1 < G::NXI || ti + 256 * j < G::XITEMS) && (!second || b1 < p.bands) && more;
    const unsigned voff = in ? (unsigned)(second ? img1 : img0) * (unsigned)G::IMGB + off : kOut;
    // (asm: the compiler's own s_waitcnt for a builtin load here also waits for the previous unit's output stores)
    asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(lx[j]) : "v"(voff), "s"(xr));
  };
  // Vector memory operations retire in order: item j of this unit is the oldest request in flight when it is needed,
  // behind it the NXI - 1 - j younger items, the previous unit's output stores (none before the first unit) and the j
  // items of the next unit already requested.
  constexpr int kStores = G::TPW * (G::M32 ? 4 : 1);
  auto put = [&](int un, auto first) __attribute__((always_inline)) {
    const bool more = un < u1;
#pragma unroll
    for (int j = 0; j < G::NXI; ++j) {
      const f32x4_t it = decltype(first)::value ? xg::take_item<G::NXI - 1>(lx[j]) : xg::take_item<G::NXI - 1 + kStores>(lx[j]);
      if (j + 1 < G::NXI || ti + 256 * j < G::XITEMS) {
        unsigned h0, m0, l0, h1, m1, l1;
        xg::split2_trunc(it[0], it[1], h0, m0, l0);
        xg::split2_trunc(it[2], it[3], h1, m1, l1);
        *reinterpret_cast<u32x2_t*>(smem + xdst[j]) = u32x2_t{h0, h1};
        *reinterpret_cast<u32x2_t*>(smem + xdst[j] + G::XPL) = u32x2_t{m0, m1};
        *reinterpret_cast<u32x2_t*>(smem + xdst[j] + 2 * G::XPL) = u32x2_t{l0, l1};
      }
      // the next unit's item into the SAME registers, after their last use (requested before: the loop-carried value needs
      // a second register set and the copy at the loop's end waits for the prefetch AND the output stores with vmcnt(0));
      // past the last unit: requested out of range, no branch
      issue_x(un, j, more);
    }
  };


  // ---- POOL staging: the max-pool's backward in the loader ------------------------------------------------------- //
  // MaxPool2D(3, 2, 'same') on an even map: window (k, m) covers rows 2k .. 2k + 2, columns 2m .. 2m + 2, so the 2 x 2
  // block of pre-pool pixels (2k, 2k + 1) x (2m, 2m + 1) is reached by the four windows (k - 1 | k, m - 1 | m) and by
  // no other: an item = that block x 4 channels asks for the four windows' gradient quads and argmax bytes (1 KB of
  // consecutive addresses per 8 lanes), rebuilds the four pixel quads with pool.hip's maxpool_bwd_pair_kernel's terms
  // in its order (bit-identical to the separate pass), writes the band's OWN rows to DA -- the weight gradient reads
  // that tensor; this kernel no longer does -- and splits them into the planes like the plain loader.  A unit's 6 input
  // rows are the row pairs P = 0 .. 3 = rows 4b - 2 + 2P, + 1 (the first and the last row of those 8 are not staged):
  // 4 x 24 x 8 = 768 items, three per thread, P the same for a whole wave.
  constexpr int PH = G::IH / 2, PW = G::IW / 2, C = G::CIN, Q = C / 4, kPoolItems = 3;
  static_assert(!POOL || (G::DG && G::SUB == 1 && G::BR == 4 && G::IH % 2 == 0 && G::IW % 2 == 0 && 4 * PW * Q == kPoolItems * 256), "pool items");
  const __amdgpu_buffer_rsrc_t ar = gemm::make_view(reinterpret_cast<const float*>(POOL ? p.arg : nullptr), POOL ? p.p_bytes >> 2 : 0);
  const __amdgpu_buffer_rsrc_t dr = gemm::make_view(POOL ? p.DA : nullptr, POOL ? p.x_bytes : 0);
  // item j of wave w = chunk w + 4 j of the 12 (row pair P, eight column pairs) chunks; inside a chunk the 16 lanes of a
  // ds_write_b64 group hold the two quads of ONE 8-channel block (the blocks are a multiple of 32 banks apart) of
  // eight column pairs
  static_assert(!POOL || (Q == 8 && PW % 8 == 0), "chunk = 8 column pairs x 8 quads");
  const int pml = (lane >> 1) & 7, pq = 2 * (lane >> 4) + (lane & 1);
  f32x4_t pv[kPoolItems][4];
  unsigned pa[kPoolItems][4];
  auto issue_pool = [&](int u, int j, bool more) __attribute__((always_inline)) {
    const int chunk = wave + 4 * j, P = chunk / (PW / 8), m = (chunk % (PW / 8)) * 8 + pml, q = pq;
    const int img = u / G::NB, b = u - img * G::NB, k = 2 * b - 1 + P;                 // window rows k - 1, k
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int row = k - 1 + (t >> 1), col = m - 1 + (t & 1);
      const bool ok = more && (unsigned)row < (unsigned)PH && col >= 0;
      const unsigned e = (unsigned)(((img * PH + row) * PW + col) * C + 4 * q);         // element of the pooled map
      const unsigned vd = ok ? e * 4u : kOut, va = ok ? e : kOut;
      asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(pv[j][t]) : "v"(vd), "s"(xr));
      asm volatile("buffer_load_dword %0, %1, %2, 0 offen" : "=v"(pa[j][t]) : "v"(va), "s"(ar));
    }
  };
  constexpr int kStoresP = G::TPW;                           // output stores of the previous unit's compute
  auto pool_item = [&](auto J, int un, auto first) __attribute__((always_inline)) {
    constexpr int j = decltype(J)::value;
    const bool more = un < u1;
    const int u = un - 1, img = u / G::NB, b = u - img * G::NB;
    {
      // younger than item j's eight loads: the later items of this unit, the previous unit's output stores, and per
      // earlier item of this pass its four DA stores and its eight new loads
      constexpr int kYoung = (kPoolItems - 1 - j) * 8 + j * 12;
      constexpr int kWait = kYoung + (decltype(first)::value ? 0 : kStoresP);
      f32x4_t tv4[4];
      unsigned ta[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) { tv4[t] = xg::take_item<kWait>(pv[j][t]); ta[t] = xg::take_word<kWait>(pa[j][t]); }
      const int chunk = wave + 4 * j, P = chunk / (PW / 8), m = (chunk % (PW / 8)) * 8 + pml, q = pq;
      // v[y][x]: pixel (2k + y, 2m + x); window t = 2 a + bc: rows k - 1 + a, columns m - 1 + bc; argmax code = 3 ky + kx
      f32x4_t v[2][2];
#pragma unroll
      for (int y = 0; y < 2; ++y)
#pragma unroll
        for (int x = 0; x < 2; ++x) v[y][x] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int bc = 0; bc < 2; ++bc) {
          const int t = 2 * a + bc;
#pragma unroll
          for (int y = 0; y < 2; ++y) {
            if (y == 1 && a == 0) continue;                  // the odd row lies in window row k only
#pragma unroll
            for (int x = 0; x < 2; ++x) {
              if (x == 1 && bc == 0) continue;               // the odd column lies in window column m only
              const unsigned w = (unsigned)((y == 1 ? 1 : (a == 0 ? 2 : 0)) * 3 + (x == 1 ? 1 : (bc == 0 ? 2 : 0)));
#pragma unroll
              for (int c = 0; c < 4; ++c)
                if (((ta[t] >> (8 * c)) & 255u) == w) v[y][x][c] += tv4[t][c];
            }
          }
        }
      // the band's own rows (P = 1, 2) to DA; every item issues its four stores (out of range elsewhere: the count above)
      const int r0 = 4 * b - 2 + 2 * P;
      const bool own = P == 1 || P == 2;
#pragma unroll
      for (int y = 0; y < 2; ++y)
#pragma unroll
        for (int x = 0; x < 2; ++x) {
          f32x4_t o = v[y][x];
          asm volatile("" : "+v"(o));
          const unsigned off = (unsigned)((((img * G::IH + r0 + y) * G::IW + 2 * m + x) * C + 4 * q) * 4);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, o), dr, own ? off : kOut, 0, 0);
          asm volatile("s_nop 1" ::: "memory");
        }
      // planes: LDS rows 2P - 1 and 2P of the unit's six
#pragma unroll
      for (int y = 0; y < 2; ++y) {
        const int rl = 2 * P - 1 + y;
        if (rl < 0 || rl >= G::XR) continue;
#pragma unroll
        for (int x = 0; x < 2; ++x) {
          const unsigned dst = (unsigned)((q >> 1) * G::CBP + (rl * G::RP + 2 * m + x + 1) * 16 + (q & 1) * 8);
          unsigned h0, m0, l0, h1, m1, l1;
          xg::split2_trunc(v[y][x][0], v[y][x][1], h0, m0, l0);
          xg::split2_trunc(v[y][x][2], v[y][x][3], h1, m1, l1);
          *reinterpret_cast<u32x2_t*>(smem + dst) = u32x2_t{h0, h1};
          *reinterpret_cast<u32x2_t*>(smem + dst + G::XPL) = u32x2_t{m0, m1};
          *reinterpret_cast<u32x2_t*>(smem + dst + 2 * G::XPL) = u32x2_t{l0, l1};
        }
      }
      issue_pool(un, j, more);
    }
  };
  auto put_pool = [&](int un, auto first) __attribute__((always_inline)) {
    pool_item(std::integral_constant<int, 0>(), un, first);
    pool_item(std::integral_constant<int, 1>(), un, first);
    pool_item(std::integral_constant<int, 2>(), un, first);
  };

  // ---- this wave's tiles: LDS offset of the lane's pixel (tap (0, 0)) and its output offset inside the unit ------ //
  unsigned pb[G::TPW], ob[G::TPW];
#pragma unroll
  for (int t = 0; t < G::TPW; ++t) {
    const int tile = wave * G::TPW + t, sb = tile / G::BTILES;
    const int pix = (tile - sb * G::BTILES) * G::PT + px;
    const int oy = pix / G::IW, ox = pix - oy * G::IW;
    pb[t] = (unsigned)(sb * G::SUBP + kb * G::CBP + (oy * G::RP + ox) * 16);
    ob[t] = (unsigned)(pix * G::COUT + (G::M32 ? 4 * kb : 4 * kb)) * 4u;
  }
  auto mfma = [&](const bf16x8_t& a, const bf16x8_t& b, acc_t& c) {
    if constexpr (G::M32) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    else c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  };
  auto compute = [&](int u) __attribute__((always_inline)) {
    const int b = u * G::SUB + (wave * G::TPW) / G::BTILES;    // this wave's band (all its tiles lie in one)
    // a band past the batch (second half of the last unit): the wave multiplies whatever the ring holds and its stores go
    // out of range -- NOT an early return: the loaders' waits count this unit's stores (r6: tools/isa_inflight.py --cfg
    // follows every path; the return was only ever taken in the last unit, behind which nothing is waited for)
    const unsigned dead = b >= p.bands ? kOut : 0u;
    const unsigned ys = b >= p.bands ? 0u : (unsigned)b * (unsigned)(G::NPIX * G::COUT * 4);
#pragma unroll
    for (int t = 0; t < G::TPW; ++t) {
      acc_t acc = acc0;
      bf16x8_t xv[2][3];
      auto fetch = [&](int tap, bf16x8_t (&x)[3]) {
        const int off = ((tap / 3) * G::RP + (tap % 3)) * 16;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) x[pl] = *reinterpret_cast<const bf16x8_t*>(smem + pb[t] + off + pl * G::XPL);
      };
      fetch(0, xv[0]);
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        if (tap + 1 < 9) fetch(tap + 1, xv[(tap + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
        const bf16x8_t (&x)[3] = xv[tap & 1];
        mfma(wl[tap], x[0], acc);
        mfma(wh[tap], x[2], acc);
        mfma(wm[tap], x[1], acc);
        mfma(wm[tap], x[0], acc);
        mfma(wh[tap], x[1], acc);
        mfma(wh[tap], x[0], acc);
        __builtin_amdgcn_sched_barrier(0);
      }
      // outputs: four consecutive channels of the lane's pixel per quad (finished and pinned before the first store)
      if constexpr (G::M32) {
        f32x4_t o[4];
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) { o[g4] = f32x4_t{acc[4 * g4], acc[4 * g4 + 1], acc[4 * g4 + 2], acc[4 * g4 + 3]}; asm volatile("" : "+v"(o[g4])); }
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, o[g4]), yr, (ob[t] + (unsigned)(32 * g4)) | dead, ys, 0);
          asm volatile("s_nop 1" ::: "memory");
        }
      } else {
        f32x4_t o = acc;
        asm volatile("" : "+v"(o));
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, o), yr, ob[t] | dead, ys, 0);
        asm volatile("s_nop 1" ::: "memory");
      }
    }
  };

  // (the weights are finished before the first requests go out: cgx.h)
#pragma unroll
  for (int t = 0; t < 9; ++t) asm volatile("" :: "v"(wh[t]), "v"(wm[t]), "v"(wl[t]));
  if constexpr (POOL) {
#pragma unroll
    for (int j = 0; j < kPoolItems; ++j) issue_pool(u0, j, true);
  } else {
#pragma unroll
    for (int j = 0; j < G::NXI; ++j) issue_x(u0, j, true);
  }
  __syncthreads();                                           // LDS zeroed
  auto step = [&](int u, auto first) __attribute__((always_inline)) {
    if constexpr (POOL) put_pool(u + 1, first); else put(u + 1, first);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // (not __syncthreads(): its fence waits for the output
    compute(u);                                                        //  stores and the prefetch with vmcnt(0))
    asm volatile("s_barrier" ::: "memory");
  };
  // (first unit peeled: its items have no output stores behind them)
  step(u0, std::true_type());
  for (int u = u0 + 1; u < u1; ++u) step(u, std::false_type());
}

// ---- served geometries ------------------------------------------------------------------------------------------ //
typedef Geo<16, 32, 36, 48, 4, 2, false> GeoFwd;            // forward: 16 -> 32 on 36 x 48: two 4-row bands (2 x 6 tiles of 32 pixels)
typedef Geo<32, 16, 36, 48, 4, 1, true> GeoDgrad;           // data gradient: dY (32) -> dX (16): one 4-row band (12 tiles of 16)

inline bool geometry(const seedhip_conv_geom* g) {
  return g->kh == 3 && g->kw == 3 && g->stride == 1 && g->pad_t == 1 && g->pad_l == 1 && g->cin == 16 && g->cout == 32 &&
         g->ih == 36 && g->iw == 48 && g->oh == 36 && g->ow == 48 && g->ld_in == 16 && g->ld_out == 32;
}
bool plan(const seedhip_conv_geom* g) {
  if (!geometry(g) || g->n_img < 64) return false;
  return (long long)g->n_img * 36 * 48 * 32 * 4 < (1LL << 31) - (1 << 22);
}

template <class G, bool POOL = false>
inline int launch_geo(Params& p, hipStream_t s) {
  static const int cus = xg::cu_count();
  p.bands = p.n_img * G::NB;
  p.units = (p.bands + G::SUB - 1) / G::SUB;
  int grid = p.units < 2 * cus ? p.units : 2 * cus;
  p.per_wg = (p.units + grid - 1) / grid;
  grid = (p.units + p.per_wg - 1) / p.per_wg;
  static const bool ok = hipFuncSetAttribute((const void*)fgx_kernel<G, POOL>, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS) == hipSuccess;
  if (!ok) return fail(SEEDHIP_ERR_LAUNCH, "fgx_kernel: LDS attribute");
  hipLaunchKernelGGL((fgx_kernel<G, POOL>), dim3(grid), dim3(256), G::LDS, s, p);
  return check_launch("fgx_kernel");
}

int launch_fwd(const seedhip_conv_geom* g, const float* X, const float* W, const float* bias, float* Y, hipStream_t s) {
  Params p;
  memset(&p, 0, sizeof(p));
  p.X = X; p.W = W; p.bias = bias; p.Y = Y; p.n_img = g->n_img;
  p.x_bytes = (long long)g->n_img * 36 * 48 * 16 * 4; p.y_bytes = (long long)g->n_img * 36 * 48 * 32 * 4;
  return launch_geo<GeoFwd>(p, s);
}
int launch_dgrad(const seedhip_conv_geom* g, const float* dY, const float* W, float* dX, hipStream_t s) {
  Params p;
  memset(&p, 0, sizeof(p));
  p.X = dY; p.W = W; p.bias = nullptr; p.Y = dX; p.n_img = g->n_img;
  p.x_bytes = (long long)g->n_img * 36 * 48 * 32 * 4; p.y_bytes = (long long)g->n_img * 36 * 48 * 16 * 4;
  return launch_geo<GeoDgrad>(p, s);
}

// dX AND the pre-pool gradient DA from the pooled map's gradient + the pool's argmax bytes (the max-pool backward fused)
int launch_dgrad_pool(const seedhip_conv_geom* g, const float* dpooled, const unsigned char* argmax, const float* W, float* dX,
                      float* DA, hipStream_t s) {
  Params p;
  memset(&p, 0, sizeof(p));
  p.X = dpooled; p.arg = argmax; p.DA = DA; p.W = W; p.bias = nullptr; p.Y = dX; p.n_img = g->n_img;
  p.p_bytes = (long long)g->n_img * 18 * 24 * 32 * 4;
  p.x_bytes = (long long)g->n_img * 36 * 48 * 32 * 4; p.y_bytes = (long long)g->n_img * 36 * 48 * 16 * 4;
  return launch_geo<GeoDgrad, true>(p, s);
}

}  // namespace fgx
}  // namespace seedhip

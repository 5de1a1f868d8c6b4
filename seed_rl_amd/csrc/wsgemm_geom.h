// Geometry of the weight-stationary conv kernels (wsgemm.h) shared by the device kernels and the HOST: the launch
// parameters, the shape plans, and the index arithmetic of the specialised kernel (row decode and stepping, A byte
// offsets and tap validity, the W' re-indexing of the data gradient, the dX scatter offset) as plain HOST+DEVICE
// functions -- tests/host/emul.cpp runs the second Atari conv through exactly these on the CPU.
#pragma once
#include <string.h>
#include "igemm.h"                              // FastDiv, SH_HD
#include "../../include/seedhip.h"

namespace seedhip {
namespace wsgemm {

constexpr int BK = 32, LDA = BK + 8, kMaxTiles = 16;

struct Params {
  int mode;                                   // 0 forward, 1 data gradient
  const float* A; int a_relu;                 // forward: layer input; data gradient: dY
  const float* W;                             // Keras kernel [kh, kw, cin, cout]
  int kh, kw, cin, cout, s;
  int M, N, K, nkt;                           // GEMM extents; nkt = K / 32
  int gh, gw;                                 // grid of m per image (output pixels | super-pixels)
  FastDiv d_g, d_gw;                          // m -> (img, rem) -> (a, b)
  unsigned a_img_stride, a_row_stride, a_col_stride;   // floats: row base = img*.. + a*.. + b*..
  int tile_off[kMaxTiles];                    // floats added to the row base for k-tile t (may be negative)
  int tile_dy[kMaxTiles], tile_dx[kMaxTiles]; // k-tile t of row (a, b) is valid iff 0 <= a+dy < vh && 0 <= b+dx < vw
  int vh, vw;
  // forward epilogue: out[m*ldc + n] = act(acc + bias[n])
  float* C; int ldc; const float* bias; int out_relu; const float* residual;
  // data-gradient epilogue: dx[img, s*a+py, s*b+px, ci] = mask(acc) + add
  int ih, iw, ld_in; const float* mask; const float* add;
  int ntiles;
  long long a_bytes;                          // extent of A in bytes (buffer-resource range of the specialised kernel)
  long long c_bytes;                          // extent of C (and of mask / add / residual, which are indexed like C) in bytes
  int pow2, l_cout, l_cin, l_s, l_jw;         // data gradient: cout, cin, s, kw / s all powers of two -> W' index math by shifts
  // data gradient, ws_tab_kernel<.., BITS = true> only: the ReLU mask as one BYTE per four channels, indexed like C / 16
  // (bit r of byte [pixel][q] = activation[pixel][4 q + r] > 0; written by the producing forward kernel) instead of the
  // fp32 activation itself: 1/16 of the mask traffic
  const unsigned char* mask_bits;
};


// (img, a, b) of GEMM row m, and of m + step (step < gw): one division pair per lane and tile, the rest is stepped
SH_HD void ws_locate(const Params& p, uint32_t m, uint32_t& img, uint32_t& a, uint32_t& b) {
  uint32_t rem;
  p.d_g.divmod(m, img, rem);
  p.d_gw.divmod(rem, a, b);
}
SH_HD void ws_advance(const Params& p, uint32_t step, uint32_t& img, uint32_t& a, uint32_t& b) {
  b += step;
  if (b >= (uint32_t)p.gw) { b -= (uint32_t)p.gw; if (++a >= (uint32_t)p.gh) { a = 0; ++img; } }
}
SH_HD uint32_t ws_mul24(uint32_t x, uint32_t y) {          // both < 2^24 (rows / columns of a map, strides of a row)
#if defined(__HIP_DEVICE_COMPILE__)
  return __umul24(x, y);
#else
  return x * y;
#endif
}
// byte offset of A element (row (img, a, b), float kc of k-tile 0); the k-tile adds 4 * tile_off[t]
SH_HD unsigned ws_row_byte(const Params& p, uint32_t img, uint32_t a, uint32_t b, int kc) {
  return (img * p.a_img_stride + ws_mul24(a, p.a_row_stride) + ws_mul24(b, p.a_col_stride) + (unsigned)kc) * 4u;
}
// data gradient: k-tile t of super-pixel (a, b) reads dY[a + dy, b + dx], zero outside the map
SH_HD bool ws_tap_ok(const Params& p, uint32_t a, uint32_t b, int dy, int dx) {
  return (unsigned)((int)a + dy) < (unsigned)p.vh && (unsigned)((int)b + dx) < (unsigned)p.vw;
}
// data gradient: W'[k][n] = W[src], k = (jy, jx, co), n = (py, px, ci)
SH_HD int ws_wprime_src(const Params& p, int k, int n) {
  if (p.pow2) {
    const int co = k & (p.cout - 1), tap = k >> p.l_cout, jy = tap >> p.l_jw, jx = tap & ((1 << p.l_jw) - 1);
    const int ci = n & (p.cin - 1), cls = n >> p.l_cin, py = cls >> p.l_s, px = cls & (p.s - 1);
    return (((((py + (jy << p.l_s)) * p.kw + px + (jx << p.l_s)) << p.l_cin) + ci) << p.l_cout) + co;
  }
  const int jw = p.kw / p.s;
  const int co = k % p.cout, tap = k / p.cout, jy = tap / jw, jx = tap - jy * jw;
  const int ci = n % p.cin, cls = n / p.cin, py = cls / p.s, px = cls - py * p.s;
  return (((py + p.s * jy) * p.kw + px + p.s * jx) * p.cin + ci) * p.cout + co;
}
// data gradient: the part of the dX offset that depends on the column n = (py, px, ci) only, and (py, px)
SH_HD unsigned ws_dgrad_col(const Params& p, int n, int& py, int& px) {
  const int cls = n / p.cin, ci = n - cls * p.cin;
  py = cls / p.s; px = cls - py * p.s;
  return (unsigned)((py * p.iw + px) * p.ld_in + ci);
}
// data gradient: element offset into dX of (super-pixel (img, a, b), column constant e_const), 0xffffffff when the
// pixel lies outside the map (odd extents)
SH_HD unsigned ws_dgrad_at(const Params& p, uint32_t img, uint32_t a, uint32_t b, unsigned e_const, int py, int px, bool exact) {
  if (!exact && ((int)a * p.s + py >= p.ih || (int)b * p.s + px >= p.iw)) return 0xffffffffu;
  return img * (unsigned)(p.ih * p.iw * p.ld_in) + ws_mul24(a, (unsigned)(p.s * p.iw * p.ld_in)) +
         ws_mul24(b, (unsigned)(p.s * p.ld_in)) + e_const;
}

struct Plan { bool ok; int mr, nr, grid; size_t lds; };

// Fills the geometry for the forward of a 'valid' conv; ok = false when the shape is outside this kernel's range.
inline Plan plan_fwd(Params& p, const seedhip_conv_geom* g) {
  Plan pl; memset(&pl, 0, sizeof(pl));
  const int seg = g->kw * g->cin;
  if (g->pad_t || g->pad_l || g->ld_in != g->cin || seg % BK || (g->cout != 32 && g->cout != 64) || g->ld_out % 4) return pl;
  const int K = g->kh * seg, N = g->cout, tiles_per_row = seg / BK;
  if (K / BK > kMaxTiles || (K / BK) % 2 || (long long)K * (N == 32 ? N + 8 : N) * 4 > 40 * 1024) return pl;
  if ((long long)g->n_img * g->ih * g->iw * g->ld_in >= (1LL << 31)) return pl;
  memset(&p, 0, sizeof(p));
  p.mode = 0; p.kh = g->kh; p.kw = g->kw; p.cin = g->cin; p.cout = g->cout; p.s = g->stride;
  p.M = g->n_img * g->oh * g->ow; p.N = N; p.K = K; p.nkt = K / BK;
  p.gh = g->oh; p.gw = g->ow; p.d_g.init(g->oh * g->ow); p.d_gw.init(g->ow);
  p.a_img_stride = (unsigned)(g->ih * g->iw * g->ld_in); p.a_row_stride = (unsigned)(g->stride * g->iw * g->ld_in);
  p.a_col_stride = (unsigned)(g->stride * g->ld_in);
  for (int t = 0; t < p.nkt; ++t) {
    p.tile_off[t] = (t / tiles_per_row) * g->iw * g->ld_in + (t % tiles_per_row) * BK;
    p.tile_dy[t] = 0; p.tile_dx[t] = 0;
  }
  p.vh = g->oh; p.vw = g->ow;
  p.ldc = g->ld_out;
  p.a_bytes = (long long)g->n_img * g->ih * g->iw * g->ld_in * 4;
  p.c_bytes = (long long)g->n_img * g->oh * g->ow * g->ld_out * 4;
  pl.nr = N / 16; pl.ok = true;
  return pl;
}

// Data gradient of a 'valid' conv whose kernel extents are multiples of the stride.
inline Plan plan_dgrad(Params& p, const seedhip_conv_geom* g) {
  Plan pl; memset(&pl, 0, sizeof(pl));
  const int s = g->stride;
  if (g->pad_t || g->pad_l || g->kh % s || g->kw % s || g->cout % BK || g->cin % 4 || g->ld_in % 4 || g->ld_out % 4) return pl;
  const int N = s * s * g->cin, jh = g->kh / s, jw = g->kw / s, K = jh * jw * g->cout, per_tap = g->cout / BK;
  if ((N != 32 && N != 64) || K / BK > kMaxTiles || (K / BK) % 2 || (long long)K * (N == 32 ? N + 8 : N) * 4 > 40 * 1024) return pl;
  if ((long long)g->n_img * g->oh * g->ow * g->ld_out >= (1LL << 31)) return pl;
  if ((long long)g->n_img * g->ih * g->iw * g->ld_in >= (1LL << 32) - 1) return pl;     // dX offsets are 32-bit
  memset(&p, 0, sizeof(p));
  p.mode = 1; p.kh = g->kh; p.kw = g->kw; p.cin = g->cin; p.cout = g->cout; p.s = s;
  p.gh = (g->ih + s - 1) / s; p.gw = (g->iw + s - 1) / s;
  p.M = g->n_img * p.gh * p.gw; p.N = N; p.K = K; p.nkt = K / BK;
  p.d_g.init(p.gh * p.gw); p.d_gw.init(p.gw);
  p.a_img_stride = (unsigned)(g->oh * g->ow * g->ld_out); p.a_row_stride = (unsigned)(g->ow * g->ld_out);
  p.a_col_stride = (unsigned)g->ld_out;
  for (int t = 0; t < p.nkt; ++t) {
    const int tap = t / per_tap, jy = tap / jw, jx = tap % jw;
    p.tile_off[t] = -(jy * g->ow + jx) * g->ld_out + (t % per_tap) * BK;
    p.tile_dy[t] = -jy; p.tile_dx[t] = -jx;
  }
  p.vh = g->oh; p.vw = g->ow;
  p.ih = g->ih; p.iw = g->iw; p.ld_in = g->ld_in;
  p.a_bytes = (long long)g->n_img * g->oh * g->ow * g->ld_out * 4;
  p.c_bytes = (long long)g->n_img * g->ih * g->iw * g->ld_in * 4;
  {
    auto lg = [](int v) { int l = 0; while ((1 << l) < v) ++l; return (1 << l) == v ? l : -1; };
    p.l_cout = lg(g->cout); p.l_cin = lg(g->cin); p.l_s = lg(s); p.l_jw = lg(jw);
    p.pow2 = p.l_cout >= 0 && p.l_cin >= 0 && p.l_s >= 0 && p.l_jw >= 0;
  }
  pl.nr = N / 16; pl.ok = true;
  return pl;
}


}  // namespace wsgemm
}  // namespace seedhip

// Geometry of the GEMM core (gemm.h) that is shared by the device kernels and the HOST: operand descriptors, the
// launch parameters, and the index arithmetic of gathered (convolution) operands as plain HOST+DEVICE functions --
// tests/host/emul.cpp executes convolutions through exactly these functions on the CPU, so the address / predicate
// math the GPU stagers use is unit-tested without a GPU (as conv_problems.h is for the implicit-GEMM core).
#pragma once
#include <string.h>
#include "igemm.h"                              // FastDiv, SH_HD
#include "../../include/seedhip.h"

namespace seedhip {
namespace gemm {

// A k-contiguous operand whose rows are GATHERED instead of dense: the im2col row of a convolution (forward), the
// dY taps of a (super-)pixel (data gradient), the Keras kernel re-indexed by (parity class, ci) (data gradient B).
//   row x -> (u, v, w) by two divisions;  row base = const0 + u*s0 + v*s1 + w*s2
//   k -> tap = k >> cshift (C = 1 << cshift contiguous floats per tap: the channels), tap -> (ty, tx) = divmod(tap, tw)
//   element address = row base + ty*tsy + tx*tsx + (k & (C-1)); it reads as zero unless tap < ntaps and
//   (y0 + ty*ey, x0 + tx*ex) lies inside [0, vh) x [0, vw), with (y0, x0) = (ya*cy + oy0, xb*cx + ox0) and
//   (ya, xb) = (v, w) (or (u, v) when coord_uv): 'same' padding and map borders cost nothing but the predicate.
// A 16-byte vector never straddles taps (C is a power of two >= 4).
struct Gather {
  FastDiv d1, d2, d_tw;
  long long s0, const0; int s1, s2;
  int coord_uv, cshift, ntaps;
  int tsy, tsx;
  int cy, oy0, ey, cx, ox0, ex, vh, vw, all_valid;
  long long extent;                         // floats in the gathered tensor (0 = unknown): bounds of the buffer view
  // Row order "block of images x position" (data gradients of 'valid' convs, conv.hip): row x = (ib * G + pos) * blk + r
  // is super-pixel `pos` of image ib * blk + r, G = d1.d positions per image.  A workgroup tile of blk rows then shares
  // ONE position, so whether a tap (k-tile) falls outside dY is uniform and the tile skips it (gemm.h); consecutive
  // tiles walk the positions of one image block, whose dY stays in L2.  blk = 0: rows are (image, position).
  int blk, n_img;
  FastDiv d_blk;
};

struct Params {
  const float* A; long long lda; int a_relu;
  const float* B; long long ldb;
  int M, N, K;
  int k_per_slice;                          // split-K over blockIdx.z (multiple of BK)
  float* partial;                           // [slices][M][N] raw sums, or null: fused epilogue below
  float* partial_colsum;                    // [slices][N]: sum_k B(k, n) (bias gradient; OC B only), or null
  float* C; long long ldc;
  const float* bias; const float* residual; int out_relu;      // forward epilogue
  const float* mask; const float* add;                         // data-gradient epilogue (indexed like C)
  const unsigned char* mask_bits;           // xgemm.h only (r5): the ReLU mask as bytes, indexed like C / 4 (bit r = element 4 q + r > 0)
  Gather ga, gb;                            // gathered operands (conv kernels below); unused by the Dense GEMMs
  int es, eih, eiw;                         // scatter epilogue (conv data gradient): stride, input map extents
};

constexpr int BK = 32, LD_KC = BK + 8;


// ---- index arithmetic of a gathered operand (see struct Gather) -------------------------------------------- //
// Row part: float offset of the row base (relative to the operand pointer) and the border coordinates (y0, x0).
// row x -> (u, v, w) = (image, grid row, grid column); false: a padding row of the image-block order (no such image)
SH_HD bool gather_locate(const Gather& g, uint32_t x, uint32_t& u, uint32_t& v, uint32_t& w) {
  uint32_t rem;
  if (g.blk) {
    uint32_t t, r, ib;
    g.d_blk.divmod(x, t, r);
    g.d1.divmod(t, ib, rem);
    g.d2.divmod(rem, v, w);
    u = ib * (uint32_t)g.blk + r;
    return u < (uint32_t)g.n_img;
  }
  g.d1.divmod(x, u, rem);
  g.d2.divmod(rem, v, w);
  return true;
}
SH_HD bool gather_row(const Gather& g, int x, long long& base, int& y0, int& x0) {
  uint32_t u, v, w;
  const bool ok = gather_locate(g, (uint32_t)x, u, v, w);
  base = g.const0 + (long long)u * g.s0 + (long long)v * g.s1 + (long long)w * g.s2;
  y0 = (int)(g.coord_uv ? u : v) * g.cy + g.oy0;
  x0 = (int)(g.coord_uv ? v : w) * g.cx + g.ox0;
  return ok;
}
// Tap part of reduction / row index k: float offset added to the row base, border shift, and whether the tap exists.
SH_HD void gather_tap(const Gather& g, int k, int& toff, int& dy, int& dx, bool& tap_ok) {
  const int tap = k >> g.cshift, kin = k & ((1 << g.cshift) - 1);
  uint32_t ty, tx;
  g.d_tw.divmod((uint32_t)tap, ty, tx);
  toff = (int)ty * g.tsy + (int)tx * g.tsx + kin;
  dy = (int)ty * g.ey; dx = (int)tx * g.ex;
  tap_ok = tap < g.ntaps;
}
// The same row arithmetic in the form the VALU-lean stagers use (gemm.h GatherOCStager): one division pair per k-tile
// (gather_decode), the following rows STEPPED (gather_step, step <= extent of w), element offset as 32-bit bytes
// (gather_elem32; valid when the tensor is < 2 GB and !coord_uv).  tests/host/emul.cpp walks the weight gradient's
// reduction index with these and checks them against gather_row / gather_inside.
SH_HD void gather_decode(const Gather& g, uint32_t x, uint32_t& u, uint32_t& v, uint32_t& w) {
  uint32_t rem;
  g.d1.divmod(x, u, rem);
  g.d2.divmod(rem, v, w);
}
SH_HD void gather_step(const Gather& g, uint32_t rows_v, uint32_t step, uint32_t& u, uint32_t& v, uint32_t& w) {
  w += step;
  if (w >= g.d2.d) { w -= g.d2.d; if (++v >= rows_v) { v = 0; ++u; } }
}
SH_HD bool gather_elem32(const Gather& g, uint32_t u, uint32_t v, uint32_t w, int toff, int tdy, int tdx, unsigned& byte_off) {
  const int y = (int)v * g.cy + g.oy0 + tdy, x = (int)w * g.cx + g.ox0 + tdx;
  byte_off = (unsigned)((int)g.const0 + toff + (int)u * (int)g.s0 + (int)v * g.s1 + (int)w * g.s2) * 4u;
  return g.all_valid || ((unsigned)y < (unsigned)g.vh && (unsigned)x < (unsigned)g.vw);
}
SH_HD bool gather_inside(const Gather& g, int y, int x) {
  return g.all_valid || (y >= 0 && y < g.vh && x >= 0 && x < g.vw);
}
// Scatter epilogue of the conv data gradient: GEMM element (m = super-pixel, n = (py, px, ci)) -> offset into dX
// (returns false when the pixel lies outside the input map: odd extents).
SH_HD bool scatter_addr(const Gather& ga, const Gather& gb, int es, int eih, int eiw, long long ldc, int m, int n,
                        long long& at) {
  uint32_t img, sa, sb, py, rem2, px, ci;
  if (!gather_locate(ga, (uint32_t)m, img, sa, sb)) return false;
  gb.d1.divmod((uint32_t)n, py, rem2);
  gb.d2.divmod(rem2, px, ci);
  const int oy = (int)sa * es + (int)py, ox = (int)sb * es + (int)px;
  if (oy >= eih || ox >= eiw) return false;
  at = (((long long)img * eih + oy) * eiw + ox) * ldc + ci;
  return true;
}

// ---- convolutions as gather-GEMMs (see struct Gather) ------------------------------------------------------ //
inline int log2_exact(int v) { int l = 0; while ((1 << l) < v) ++l; return (1 << l) == v ? l : -1; }

// Forward: m = output pixel, k = (ky, kx, ci), n = co, B = the Keras kernel as stored (OC).  Any stride and padding;
// needs cin a power of two >= 4, ld_in % 4 == 0, cout % 4 == 0.
inline bool conv_fwd_setup(Params& p, const seedhip_conv_geom* g) {
  const int cs = log2_exact(g->cin);
  if (cs < 2 || g->ld_in % 4 || g->cout % 4 || g->ld_out % 4) return false;
  memset(&p, 0, sizeof(p));
  p.M = g->n_img * g->oh * g->ow; p.N = g->cout; p.K = g->kh * g->kw * g->cin; p.k_per_slice = (p.K + BK - 1) / BK * BK;
  p.ldb = g->cout; p.ldc = g->ld_out;
  Gather& a = p.ga;
  a.d1.init(g->oh * g->ow); a.d2.init(g->ow); a.d_tw.init(g->kw);
  a.s0 = (long long)g->ih * g->iw * g->ld_in; a.s1 = g->stride * g->iw * g->ld_in; a.s2 = g->stride * g->ld_in;
  a.const0 = -((long long)g->pad_t * g->iw + g->pad_l) * g->ld_in;
  a.cshift = cs; a.ntaps = g->kh * g->kw; a.tsy = g->iw * g->ld_in; a.tsx = g->ld_in;
  a.cy = g->stride; a.oy0 = -g->pad_t; a.ey = 1; a.cx = g->stride; a.ox0 = -g->pad_l; a.ex = 1; a.vh = g->ih; a.vw = g->iw;
  a.all_valid = g->pad_t == 0 && g->pad_l == 0 && (g->oh - 1) * g->stride + g->kh <= g->ih &&
                (g->ow - 1) * g->stride + g->kw <= g->iw;
  a.extent = (long long)g->n_img * g->ih * g->iw * g->ld_in;
  return true;
}

// Data gradient: m = super-pixel (a, b) covering input pixels (s*a+py, s*b+px), n = (py, px, ci), k = (jy, jx, co):
//   dX[s*a+py, s*b+px, ci] = sum dY[a-jy, b-jx, co] * W[py+s*jy, px+s*jx, ci, co]                  (pad 0)
// one GEMM for all stride-parity classes; for stride 1 with padding the same with dY[y+pad-jy, x+pad-jx].
// A rows = dY taps (zero outside the map), B rows = the kernel's co-contiguous rows re-indexed by (py, px, ci).
// Needs kh % s == kw % s == 0, cout a power of two >= 4, and pad 0 unless s == 1.
inline bool conv_dgrad_setup(Params& p, const seedhip_conv_geom* g) {
  const int s = g->stride, cs = log2_exact(g->cout);
  if (cs < 2 || g->kh % s || g->kw % s || g->ld_out % 4 || ((g->pad_t || g->pad_l) && s != 1)) return false;
  const int jh = g->kh / s, jw = g->kw / s;
  memset(&p, 0, sizeof(p));
  const int gh = (g->ih + s - 1) / s, gw = (g->iw + s - 1) / s;
  p.M = g->n_img * gh * gw; p.N = s * s * g->cin; p.K = jh * jw * g->cout; p.k_per_slice = (p.K + BK - 1) / BK * BK;
  p.ldc = g->ld_in; p.es = s; p.eih = g->ih; p.eiw = g->iw;
  Gather& a = p.ga;
  a.d1.init(gh * gw); a.d2.init(gw); a.d_tw.init(jw);
  a.s0 = (long long)g->oh * g->ow * g->ld_out; a.s1 = g->ow * g->ld_out; a.s2 = g->ld_out;
  a.const0 = ((long long)g->pad_t * g->ow + g->pad_l) * g->ld_out;
  a.cshift = cs; a.ntaps = jh * jw; a.tsy = -g->ow * g->ld_out; a.tsx = -g->ld_out;
  a.cy = 1; a.oy0 = g->pad_t; a.ey = -1; a.cx = 1; a.ox0 = g->pad_l; a.ex = -1; a.vh = g->oh; a.vw = g->ow;
  Gather& b = p.gb;
  b.d1.init(s * g->cin); b.d2.init(g->cin); b.d_tw.init(jw);
  b.s0 = (long long)g->kw * g->cin * g->cout; b.s1 = g->cin * g->cout; b.s2 = g->cout;
  b.cshift = cs; b.ntaps = jh * jw; b.tsy = s * g->kw * g->cin * g->cout; b.tsx = s * g->cin * g->cout;
  b.all_valid = 1;
  return true;
}

// Weight gradient: m = dW row (ky, kx, c), n = co, k = output pixel; A = the input gathered per tap (OC, above),
// B = dY [pixel, co] as stored.  Same geometry requirements as the forward.
inline bool conv_wgrad_setup(Params& p, const seedhip_conv_geom* g) {
  Params f;
  if (!conv_fwd_setup(f, g)) return false;
  memset(&p, 0, sizeof(p));
  p.ga = f.ga;
  p.M = g->kh * g->kw * g->cin; p.N = g->cout; p.K = g->n_img * g->oh * g->ow;
  p.ldb = g->ld_out;
  return true;
}

}  // namespace gemm
}  // namespace seedhip

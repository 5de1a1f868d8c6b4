// Direct convolution with the input band staged ONCE in LDS: forward of small-kernel Conv2D layers and,
// through a weight-index transform, their data gradient.
//
// Used by seedhip_conv2d_fwd / seedhip_conv2d_bwd_data for the conv stacks of
// /root/reference/dmlab/networks.py:31-60 (3x3 'same', stride 1) and the second Atari conv
// (/root/reference/atari/networks.py:236: 4x4 stride 2) and their TF autodiff wrt the layer input.
//
// The implicit-GEMM core gathers every input element kh*kw times from global memory with per-element
// index arithmetic and is VALU-bound on these shapes.  Here a persistent workgroup walks (image, row-band)
// tiles; the input band + halo is copied to LDS once (ReLU / u8->/255 applied once per element, next tile
// prefetched into registers), the layer's weights sit in LDS for the whole launch, and the MFMA operands are
// ds_read_b128 with no div/mod:
//   k-group = 4 consecutive input channels of one tap; MFMA step kk of a 16-deep slice takes element kk of
//   lane-group kq's k-group (k = 4*kq + kk), so one b128 read feeds 4 MFMAs.
//   A (rows = 16 output channels): weights, LDS image [slice][co-tile][lane][4]  (linear, conflict free)
//   B (cols = 16 output pixels):   X[(oy*s+ky)*twp + ox*s+kx][c..c+3]; pixel stride chosen so that the four
//                                  16-lane groups of ds_read_b128 hit distinct 16-B slots
//   D: lane holds 4 consecutive output channels of one pixel -> 16-byte fully coalesced stores with the
//      epilogue (bias, ReLU, residual | ReLU-mask, accumulate) fused.
// Data gradient = the same kernel per stride-parity class (py,px) of the input pixel: a stride-1 correlation of
// dY with the taps ky = py + s*j (flipped), output written to the strided positions of that class -- no
// structural zeros are multiplied.
#pragma once
#include "common.h"
#include "igemm.h"
#include "../../include/seedhip.h"

namespace seedhip {
namespace halo {

typedef float f32x4_t __attribute__((ext_vector_type(4)));

struct FwdClass {                                  // one launch computes up to 4 "classes" off the same input tile
  int kh, kw, pad_t, pad_l;                        // the class's kernel and pads (input side)
  int oh, ow;                                      // its output grid
  int oy0, ox0;                                    // placement offset: out[n, oy*so + oy0, ox*so + ox0, :]
  int w_py, w_px;                                  // data-gradient parity (wmode 1)
  int nslices, w_off;                              // k-slices and start (in slices) inside the LDS weight image
  int r_off, c_off;                                // tile-row / tile-column of this class's first tap
  FastDiv d_ow;
};

struct FwdParams {
  const void* in; int in_dtype, in_relu;        // input [n_img, ih, iw, ld_in]
  const float* w;                                // original Keras kernel [w_kh, w_kw, w_cin, w_cout]
  int wmode;                                     // 0: forward, W_eff[tap][c][co] = w[tap][c][co]
                                                 // 1: data gradient class (w_py, w_px): W_eff[(j,i)][c][co] =
                                                 //    w[w_py + w_s*(kh-1-j)][w_px + w_s*(kw-1-i)][co][c]
  int w_kh, w_kw, w_cin, w_cout, w_s;
  int n_img, ih, iw, cin, stride;                // input tensor; stride of the forward conv (1 for data gradients)
  int cout;
  float* out; int OH, OW, ld_out, so;            // output tensor and placement stride
  const float* bias; int out_relu; const float* residual;
  const float* mask; const float* add;           // data-gradient epilogue (indexed like out)
  int ld_in;
  int ncls; FwdClass cls[4];
  int oh_max;                                    // bands run over max_c oh
  int tile_pad_t, tile_pad_l;                    // input row/col of tile origin = y0*stride - tile_pad_t, -tile_pad_l
  int TH, bands, ntiles, thp, twp, xs;           // tiling; xs = LDS pixel stride (floats)
  int cgs;                                       // k-groups (of 4 channels) per tap = ceil(cin / 4), power of 2
  int cgs_shift;
  int total_slices;
};

template <int MT, int NT>
__global__ void __launch_bounds__(256)
halo_fwd_kernel(const FwdParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int w_floats = p.total_slices * NT * 256;
  float* w_lds = smem;
  int* koff_tab = reinterpret_cast<int*>(smem + w_floats);      // [total_slices][4]: LDS offset of (slice, lane group kq)
  float* x_lds = smem + w_floats + p.total_slices * 4;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kq = lane >> 4, j = lane & 15;

  // ---- weights -> LDS, once per workgroup: per class [slice][nt][lane][kk] ----
  for (int ci = 0; ci < p.ncls; ++ci) {
    const FwdClass& c = p.cls[ci];
    const int ntaps = c.kh * c.kw;
    float* wl = w_lds + c.w_off * NT * 256;
    for (int idx = tid; idx < c.nslices * NT * 256; idx += 256) {
      const int kk = idx & 3, l = (idx >> 2) & 63, rest = idx >> 8;
      const int nt = rest % NT, slice = rest / NT;
      const int G = slice * 4 + (l >> 4);
      const int tap = G >> p.cgs_shift, ch = ((G & (p.cgs - 1)) << 2) + kk;
      const int co = nt * 16 + (l & 15);
      float v = 0.f;
      if (tap < ntaps && ch < p.cin && co < p.cout) {
        const int ty = tap / c.kw, tx = tap - ty * c.kw;
        if (p.wmode == 0) {
          v = p.w[((long long)tap * p.w_cin + ch) * p.w_cout + co];
        } else {
          const int ky = c.w_py + p.w_s * (c.kh - 1 - ty), kx = c.w_px + p.w_s * (c.kw - 1 - tx);
          v = p.w[((long long)(ky * p.w_kw + kx) * p.w_cin + co) * p.w_cout + ch];
        }
      }
      wl[idx] = v;
    }
  }

  // per-slice operand offsets, once per workgroup: the slice loop then costs one ds_read instead of ~15 integer
  // instructions (SQ counters of cfg3, r02c: 7.8 VALU + 2.3 SALU per MFMA in this kernel, matrix pipe 39 % busy)
  for (int idx = tid; idx < p.total_slices * 4; idx += 256) {
    const int sl = idx >> 2, q = idx & 3;
    int ci = 0;
    while (ci + 1 < p.ncls && sl >= p.cls[ci + 1].w_off) ++ci;
    const FwdClass& c = p.cls[ci];
    const int G = (sl - c.w_off) * 4 + q;
    int tap = G >> p.cgs_shift;
    const int cg = G & (p.cgs - 1);
    if (tap > c.kh * c.kw - 1) tap = c.kh * c.kw - 1;     // padded k-groups carry zero weights
    const int ty = tap / c.kw, tx = tap - ty * c.kw;
    koff_tab[idx] = (ty * p.twp + tx) * p.xs + (cg << 2);
  }

  // ---- tile pipeline (as halo_wgrad.h) ----
  constexpr int kXV = 7;
  const bool vec = p.in_dtype == 0 && (p.cin & 3) == 0 && (p.ld_in & 3) == 0;
  float4 xr[kXV];
  auto band_of = [&](int tile, int& n, int& y0, int& th) {
    n = tile / p.bands;
    const int band = tile - n * p.bands;
    y0 = band * p.TH;
    th = (y0 + p.TH <= p.oh_max) ? p.TH : p.oh_max - y0;
  };
  // A thread's staging vectors have the SAME tile coordinates in every tile: decode them once (two integer divisions
  // per vector) instead of in every load and every store (four divisions per vector and tile).
  int st_row[kXV], st_goff[kXV], st_lds[kXV];            // tile row; offset in the image relative to the band's first
  {                                                      // row, or -1 (column outside the map); LDS offset
    const int c4 = p.cin >> 2, per_row = p.twp * (c4 > 0 ? c4 : 1), rowf = p.twp * p.xs;
#pragma unroll
    for (int u = 0; u < kXV; ++u) {
      const int v = tid + u * 256;
      const int r = v / per_row, rem = v - r * per_row;
      const int xcol = rem / (c4 > 0 ? c4 : 1), cq = rem - xcol * c4;
      const int ix = xcol - p.tile_pad_l;
      st_row[u] = r;
      st_goff[u] = (ix >= 0 && ix < p.iw) ? ((r - p.tile_pad_t) * p.iw + ix) * p.ld_in + 4 * cq : -1;
      st_lds[u] = r * rowf + xcol * p.xs + 4 * cq;
    }
  }
  auto load_tile = [&](int tile) {
    if (!vec) return;
    int n, y0, th; band_of(tile, n, y0, th);
    const int nrows = p.thp - (p.TH - th) * p.stride;
    const int iy0 = y0 * p.stride - p.tile_pad_t;
    const float* src = (const float*)p.in + ((long long)n * p.ih + y0 * p.stride) * p.iw * p.ld_in;
#pragma unroll
    for (int u = 0; u < kXV; ++u) {
      float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
      const int iy = iy0 + st_row[u];
      if (st_row[u] < nrows && st_goff[u] != -1 && iy >= 0 && iy < p.ih)
        val = *reinterpret_cast<const float4*>(src + st_goff[u]);
      xr[u] = val;
    }
  };
  auto store_tile = [&](int tile) {
    int n, y0, th; band_of(tile, n, y0, th);
    const int rowf = p.twp * p.xs;
    const int nrows = p.thp - (p.TH - th) * p.stride;
    if (vec) {
#pragma unroll
      for (int u = 0; u < kXV; ++u) {
        if (st_row[u] < nrows) {
          float4 val = xr[u];
          if (p.in_relu) { val.x = fmaxf(val.x, 0.f); val.y = fmaxf(val.y, 0.f); val.z = fmaxf(val.z, 0.f); val.w = fmaxf(val.w, 0.f); }
          *reinterpret_cast<float4*>(x_lds + st_lds[u]) = val;
        }
      }
    } else {                                           // u8 / odd channel counts: channels padded to 4*cgs with zeros
      const int cp = p.cgs << 2;
      const int per_row = p.twp * cp;
      for (int v = tid; v < nrows * per_row; v += 256) {
        const int r = v / per_row, rem = v - r * per_row;
        const int xcol = rem / cp, ch = rem - xcol * cp;
        const int iy = y0 * p.stride - p.tile_pad_t + r, ix = xcol - p.tile_pad_l;
        float val = 0.f;
        if (ch < p.cin && iy >= 0 && iy < p.ih && ix >= 0 && ix < p.iw) {
          const long long off = (((long long)n * p.ih + iy) * p.iw + ix) * p.ld_in + ch;
          val = p.in_dtype == 1 ? (float)((const uint8_t*)p.in)[off] / 255.0f : ((const float*)p.in)[off];
          if (p.in_relu) val = fmaxf(val, 0.f);
        }
        x_lds[r * rowf + xcol * p.xs + ch] = val;
      }
    }
  };

  if ((int)blockIdx.x < p.ntiles) load_tile(blockIdx.x);
  for (int tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
    int n, y0, th; band_of(tile, n, y0, th);
    __syncthreads();
    store_tile(tile);
    __syncthreads();
    if (tile + (int)gridDim.x < p.ntiles) load_tile(tile + gridDim.x);

    for (int ci = 0; ci < p.ncls; ++ci) {
      const FwdClass& c = p.cls[ci];
      const int thc = (y0 + th <= c.oh) ? th : c.oh - y0;      // this class may have fewer rows than the band
      if (thc <= 0) continue;
      const int npix = thc * c.ow;
      const int ntile16 = (npix + 15) >> 4;
      const float* wl = w_lds + c.w_off * NT * 256;
      for (int t0 = wave * MT; t0 < ntile16; t0 += 4 * MT) {
        int xbase[MT], opy[MT], opx[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          int pix = (t0 + m) * 16 + j;
          if (pix > npix - 1) pix = npix - 1;
          uint32_t py, px;
          c.d_ow.divmod((uint32_t)pix, py, px);
          opy[m] = (int)py; opx[m] = (int)px;
          xbase[m] = (((int)py * p.stride + c.r_off) * p.twp + (int)px * p.stride + c.c_off) * p.xs;
        }
        f32x4_t acc[MT][NT];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[m][nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        const int* kt = koff_tab + c.w_off * 4 + kq;
        for (int s = 0; s < c.nslices; ++s) {
          const int koff = kt[s * 4];
          f32x4_t a[NT], b[MT];
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) a[nt] = *reinterpret_cast<const f32x4_t*>(wl + ((s * NT + nt) * 64 + lane) * 4);
#pragma unroll
          for (int m = 0; m < MT; ++m) b[m] = *reinterpret_cast<const f32x4_t*>(x_lds + xbase[m] + koff);
#pragma unroll
          for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
              for (int nt = 0; nt < NT; ++nt)
                acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[nt][kk], b[m][kk], acc[m][nt], 0, 0, 0);
        }
        // ---- epilogue: lane holds channels nt*16 + 4*kq + {0..3} of pixel (t0+m)*16 + j ----
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          const int pix = (t0 + m) * 16 + j;
          if (t0 + m >= ntile16 || pix >= npix) continue;
          const int oy = (y0 + opy[m]) * p.so + c.oy0, ox = opx[m] * p.so + c.ox0;
          if (oy < 0 || oy >= p.OH || ox < 0 || ox >= p.OW) continue;
          const long long obase = (long long)n * p.OH * p.OW * p.ld_out + (oy * p.OW + ox) * p.ld_out;
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            const int co = nt * 16 + 4 * kq;
            if (co >= p.cout) continue;
            f32x4_t v = acc[m][nt];
            if (p.bias) { const float4 bv = *reinterpret_cast<const float4*>(p.bias + co); v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w; }
            if (p.residual) { const float4 rv = *reinterpret_cast<const float4*>(p.residual + obase + co); v[0] += rv.x; v[1] += rv.y; v[2] += rv.z; v[3] += rv.w; }
            if (p.out_relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
            if (p.mask) {
              const float4 mv = *reinterpret_cast<const float4*>(p.mask + obase + co);
              if (!(mv.x > 0.f)) v[0] = 0.f; if (!(mv.y > 0.f)) v[1] = 0.f; if (!(mv.z > 0.f)) v[2] = 0.f; if (!(mv.w > 0.f)) v[3] = 0.f;
            }
            if (p.add) { const float4 av = *reinterpret_cast<const float4*>(p.add + obase + co); v[0] += av.x; v[1] += av.y; v[2] += av.z; v[3] += av.w; }
            *reinterpret_cast<float4*>(p.out + obase + co) = make_float4(v[0], v[1], v[2], v[3]);
          }
        }
      }
    }
  }
}

// ---- host side ------------------------------------------------------------------------------ //
struct FwdPlan { bool ok; int MT, NT, TH, grid; size_t lds; };

// Describes one class before tiling.
struct ClassSpec { int kh, kw, pad_t, pad_l, oh, ow, oy0, ox0, w_py, w_px; };

// Fills p.cls / tiling for `ncls` classes that share the input tensor already described in p
// (n_img, ih, iw, cin, stride, cout, ld_in, ld_out).  Returns ok=false if the shape is outside the kernel's range.
inline FwdPlan plan_fwd(FwdParams& p, const ClassSpec* cs, int ncls, bool u8) {
  FwdPlan pl; memset(&pl, 0, sizeof(pl));
  if (ncls < 1 || ncls > 4 || p.stride > 2 || p.cout > 32 || p.cout % 4 != 0 || p.ld_out % 4 != 0) return pl;
  if (!u8 && (p.cin % 4 != 0 || p.ld_in % 4 != 0) && p.cin > 4) return pl;
  int cgs = 1, sh = 0;
  while (cgs * 4 < p.cin) { cgs <<= 1; ++sh; }
  if (cgs > 16) return pl;
  if (!u8 && p.cin % 4 == 0 && p.cin != 4 * cgs) return pl;   // vector fill leaves no zeroed pad channels
  p.cgs = cgs; p.cgs_shift = sh;
  pl.NT = (p.cout + 15) / 16;
  int S = cgs;                                                 // LDS pixel stride in 16-B slots
  while ((p.stride * S) % 4 != 2) ++S;
  p.xs = 4 * S;
  int max_pt = 0, max_pl = 0, oh_max = 0, ow_max = 0, total = 0;
  for (int i = 0; i < ncls; ++i) {
    if (cs[i].kh > 4 || cs[i].kw > 4 || cs[i].kh < 1 || cs[i].kw < 1 || cs[i].oh < 1 || cs[i].ow < 1) return pl;
    if (cs[i].pad_t > max_pt) max_pt = cs[i].pad_t;
    if (cs[i].pad_l > max_pl) max_pl = cs[i].pad_l;
    if (cs[i].oh > oh_max) oh_max = cs[i].oh;
    if (cs[i].ow > ow_max) ow_max = cs[i].ow;
  }
  if (oh_max * ow_max < 16) return pl;
  int ext_h = 0, ext_w = 0;                                    // rows/cols the taps reach beyond (q-1)*stride
  for (int i = 0; i < ncls; ++i) {
    FwdClass& c = p.cls[i];
    c.kh = cs[i].kh; c.kw = cs[i].kw; c.pad_t = cs[i].pad_t; c.pad_l = cs[i].pad_l; c.oh = cs[i].oh; c.ow = cs[i].ow;
    c.oy0 = cs[i].oy0; c.ox0 = cs[i].ox0; c.w_py = cs[i].w_py; c.w_px = cs[i].w_px;
    c.r_off = max_pt - c.pad_t; c.c_off = max_pl - c.pad_l;
    c.nslices = (c.kh * c.kw * cgs + 3) / 4; c.w_off = total; total += c.nslices;
    c.d_ow.init(c.ow);
    if (c.kh + c.r_off > ext_h) ext_h = c.kh + c.r_off;
    if (c.kw + c.c_off > ext_w) ext_w = c.kw + c.c_off;
  }
  p.ncls = ncls; p.total_slices = total; p.oh_max = oh_max; p.tile_pad_t = max_pt; p.tile_pad_l = max_pl;
  p.twp = (ow_max - 1) * p.stride + ext_w;
  const size_t w_b = (size_t)total * pl.NT * 1024;
  // Band height TH and pixel tiles per wave MT: maximise the share of MFMA tile slots that carry pixels
  // (a band's 16-pixel tiles are dealt to 4 waves x MT at a time; e.g. 36x48 maps: TH=6, MT=5 -> 18 of 20
  // slots, where TH=6, MT=4 would fill 18 of 32), among bands that fit LDS and the register prefetch.
  int best_th = 0, best_mt = 0;
  double best_u = -1.0;
  const int th_hi = (320 + ow_max - 1) / ow_max < oh_max ? (320 + ow_max - 1) / ow_max : oh_max;
  for (int th = 1; th <= th_hi; ++th) {
    const int thp = (th - 1) * p.stride + ext_h;
    const size_t x_b = (size_t)thp * p.twp * p.xs * 4;
    const size_t x_src = (size_t)thp * p.twp * p.cin * 4;                       // bytes prefetched in registers
    if (!(w_b + (size_t)total * 16 + x_b <= 64 * 1024 && (u8 || p.cin % 4 != 0 || x_src <= 7 * 256 * 16))) continue;
    const int nb = (oh_max + th - 1) / th, last = oh_max - (nb - 1) * th;
    const int t_full = (th * ow_max + 15) / 16, t_last = (last * ow_max + 15) / 16;
    for (int mt = 2; mt <= 5; ++mt) {
      const int round = 4 * mt;
      const double slots = (double)(nb - 1) * ((t_full + round - 1) / round) * round + (double)((t_last + round - 1) / round) * round;
      double u = ((double)(nb - 1) * t_full + t_last) / slots;
      u *= 1.0 - 0.02 * nb / (double)oh_max;                                     // fewer bands: fewer barriers / halo re-reads
      if (u > best_u + 1e-9) { best_u = u; best_th = th; best_mt = mt; }
    }
  }
  if (!best_th) return pl;
  {
    const int thp = (best_th - 1) * p.stride + ext_h;
    pl.lds = w_b + (size_t)total * 16 + (size_t)thp * p.twp * p.xs * 4;
  }
  const int th = best_th;
  pl.TH = th; p.TH = th; p.thp = (th - 1) * p.stride + ext_h;
  p.bands = (oh_max + th - 1) / th; p.ntiles = p.n_img * p.bands;
  pl.MT = best_mt;
  int per_cu = (int)((160 * 1024) / pl.lds); if (per_cu > 3) per_cu = 3; if (per_cu < 1) per_cu = 1;
  const long long mg = 256LL * per_cu;
  pl.grid = (int)(p.ntiles < mg ? p.ntiles : mg);
  pl.ok = true;
  return pl;
}

inline int launch_fwd_kernel(const FwdParams& p, const FwdPlan& pl, hipStream_t s) {
#define SEEDHIP_HF(MT_, NT_)                                                                                     \
  if (pl.MT == MT_ && pl.NT == NT_) {                                                                            \
    if (pl.lds > 64 * 1024)                                                                                      \
      (void)hipFuncSetAttribute((const void*)halo_fwd_kernel<MT_, NT_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.lds); \
    hipLaunchKernelGGL((halo_fwd_kernel<MT_, NT_>), dim3(pl.grid), dim3(256), pl.lds, s, p);                     \
    return check_launch("halo_fwd_kernel");                                                                      \
  }
  SEEDHIP_HF(4, 1) SEEDHIP_HF(4, 2) SEEDHIP_HF(2, 1) SEEDHIP_HF(2, 2) SEEDHIP_HF(3, 1) SEEDHIP_HF(3, 2)
  SEEDHIP_HF(5, 1) SEEDHIP_HF(5, 2)
#undef SEEDHIP_HF
  return fail(SEEDHIP_ERR_UNSUPPORTED, "halo_fwd: no kernel for MT=%d NT=%d", pl.MT, pl.NT);
}

}  // namespace halo
}  // namespace seedhip

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

// DG = data-gradient epilogue (relu mask, skip-path add) instead of the forward one (bias, residual, relu): the two
// never mix (conv.hip), and a compile-time split keeps the epilogue straight-line.
template <int MT, int NT, bool DG, int XV>
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
  // the X tile starts as zeros: halo columns outside the map are never written again (vector path)
  for (int idx = tid; idx < p.thp * p.twp * p.xs; idx += 256) x_lds[idx] = 0.f;

  // ---- tile pipeline (as halo_wgrad.h) ----
  constexpr int kXV = XV;                                // float4 registers per thread for the prefetched X tile
  constexpr int kNoRow = 0x7fff;
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
  uint32_t st_pack[kXV];                                 // tile row (kNoRow: column outside the map) << 16 | LDS byte offset
  uint32_t st_goff[kXV];                                 // byte offset from the tile's first input row
  {
    const int c4 = p.cin >> 2, per_row = p.twp * (c4 > 0 ? c4 : 1), rowf = p.twp * p.xs;
#pragma unroll
    for (int u = 0; u < kXV; ++u) {
      const int v = tid + u * 256;
      const int r = v / per_row, rem = v - r * per_row;
      const int xcol = rem / (c4 > 0 ? c4 : 1), cq = rem - xcol * c4;
      const int ix = xcol - p.tile_pad_l;
      const bool col_ok = ix >= 0 && ix < p.iw && r < kNoRow;
      st_goff[u] = col_ok ? (uint32_t)((r * p.iw + ix) * p.ld_in + 4 * cq) * 4u : 0u;
      st_pack[u] = (uint32_t)(col_ok ? r : kNoRow) << 16 | ((uint32_t)(r * rowf + xcol * p.xs + 4 * cq) * 4u & 0xffffu);
    }
  }
  auto load_tile = [&](int tile) {
    if (!vec) return;
    int n, y0, th; band_of(tile, n, y0, th);
    const int nrows = p.thp - (p.TH - th) * p.stride;
    const int iy0 = y0 * p.stride - p.tile_pad_t;                    // image row of tile row 0 (may be negative)
    const int lo = iy0 < 0 ? -iy0 : 0;
    int hi = p.ih - iy0 < nrows ? p.ih - iy0 : nrows;
    const uint32_t cnt = hi > lo ? (uint32_t)(hi - lo) : 0u;        // tile rows [lo, lo + cnt) lie inside the image
    const char* src = reinterpret_cast<const char*>((const float*)p.in + ((long long)n * p.ih + iy0) * p.iw * p.ld_in);
#pragma unroll
    for (int u = 0; u < kXV; ++u) {
      float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
      if ((st_pack[u] >> 16) - (uint32_t)lo < cnt) val = *reinterpret_cast<const float4*>(src + st_goff[u]);
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
        if (st_pack[u] < ((uint32_t)nrows << 16)) {
          float4 val = xr[u];
          if (p.in_relu) { val.x = fmaxf(val.x, 0.f); val.y = fmaxf(val.y, 0.f); val.z = fmaxf(val.z, 0.f); val.w = fmaxf(val.w, 0.f); }
          *reinterpret_cast<float4*>(reinterpret_cast<char*>(x_lds) + (st_pack[u] & 0xffffu)) = val;
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

  // ---- epilogue operands: eA = residual (forward) or relu mask (data gradient), eB = skip-path add ----
  // Small tiles fetch them BEFORE the k loop, so that their latency hides under the MFMAs; the big ones (registers)
  // fetch them all at once after it.  Offsets are 32-bit from a per-image base.
  constexpr bool kPrefetch = MT * NT <= 6;
  const float* pA = DG ? p.mask : p.residual;
  const float* pB = DG ? p.add : nullptr;
  const float floor_v = (!DG && p.out_relu) ? 0.f : -INFINITY;
  float4 bias_f[NT];
  bool ch_ok[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int co = nt * 16 + 4 * kq;
    ch_ok[nt] = co < p.cout;
    bias_f[nt] = (!DG && p.bias && ch_ok[nt]) ? *reinterpret_cast<const float4*>(p.bias + co) : make_float4(0.f, 0.f, 0.f, 0.f);
  }

  if ((int)blockIdx.x < p.ntiles) load_tile(blockIdx.x);
  for (int tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
    int n, y0, th; band_of(tile, n, y0, th);
    __syncthreads();
    store_tile(tile);
    __syncthreads();
    if (tile + (int)gridDim.x < p.ntiles) load_tile(tile + gridDim.x);
    const long long img = (long long)n * p.OH * p.OW * p.ld_out;
    char* out_n = reinterpret_cast<char*>(p.out + img);
    const char* a_n = reinterpret_cast<const char*>(pA + img);
    const char* b_n = reinterpret_cast<const char*>(pB + img);

    for (int ci = 0; ci < p.ncls; ++ci) {
      const FwdClass& c = p.cls[ci];
      const int thc = (y0 + th <= c.oh) ? th : c.oh - y0;      // this class may have fewer rows than the band
      if (thc <= 0) continue;
      const int npix = thc * c.ow;
      const int ntile16 = (npix + 15) >> 4;
      const float* wl = w_lds + c.w_off * NT * 256;
      for (int t0 = wave * MT; t0 < ntile16; t0 += 4 * MT) {
        int xbase[MT];
        uint32_t ooff[MT];                                 // byte offset of the lane's 4 channels in the output image
        bool ook[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          const int pix0 = (t0 + m) * 16 + j;
          const int pix = pix0 > npix - 1 ? npix - 1 : pix0;
          uint32_t py, px;
          c.d_ow.divmod((uint32_t)pix, py, px);
          xbase[m] = (((int)py * p.stride + c.r_off) * p.twp + (int)px * p.stride + c.c_off) * p.xs;
          const int oy = (y0 + (int)py) * p.so + c.oy0, ox = (int)px * p.so + c.ox0;
          ook[m] = pix0 < npix && (uint32_t)oy < (uint32_t)p.OH && (uint32_t)ox < (uint32_t)p.OW;
          ooff[m] = ook[m] ? (uint32_t)((oy * p.OW + ox) * p.ld_out + 4 * kq) * 4u : 0u;   // 0: a safe address for masked lanes
        }
        float4 eA[MT][NT], eB[MT][NT];
        auto ld_pinned = [](const char* q) { return *reinterpret_cast<const float4*>(q); };
        auto fetch_extras = [&]() {
          if (pA) {
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
              for (int nt = 0; nt < NT; ++nt)
                eA[m][nt] = ld_pinned(a_n + (ch_ok[nt] ? ooff[m] + nt * 64u : 0u));
          }
          if (DG && pB) {
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
              for (int nt = 0; nt < NT; ++nt)
                eB[m][nt] = ld_pinned(b_n + (ch_ok[nt] ? ooff[m] + nt * 64u : 0u));
          }
        };
        if constexpr (kPrefetch) {
          fetch_extras();
          // compiler barrier for memory operations: without it the loads just requested are SUNK to their first use
          // behind the k loop, and the epilogue waits out a full memory round trip per tile
          asm volatile("" ::: "memory");
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
        if constexpr (!kPrefetch) fetch_extras();
        // every output is FINISHED (all epilogue operands consumed) before the first store is issued: the stores sit in
        // predicated blocks, and a block that still needs a prefetched operand while an earlier block's store is in
        // flight waits with s_waitcnt vmcnt(0) -- loads and stores share the counter and do not retire in order with
        // each other --, i.e. for that store's write latency: MT * NT - 1 serialised round trips per tile
        float4 outv[MT][NT];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            float4 v = make_float4(acc[m][nt][0], acc[m][nt][1], acc[m][nt][2], acc[m][nt][3]);
            if constexpr (!DG) {
              v.x += bias_f[nt].x; v.y += bias_f[nt].y; v.z += bias_f[nt].z; v.w += bias_f[nt].w;
              if (pA) { v.x += eA[m][nt].x; v.y += eA[m][nt].y; v.z += eA[m][nt].z; v.w += eA[m][nt].w; }
              v.x = fmaxf(v.x, floor_v); v.y = fmaxf(v.y, floor_v); v.z = fmaxf(v.z, floor_v); v.w = fmaxf(v.w, floor_v);
            } else {
              if (pA) {
                if (!(eA[m][nt].x > 0.f)) v.x = 0.f; if (!(eA[m][nt].y > 0.f)) v.y = 0.f;
                if (!(eA[m][nt].z > 0.f)) v.z = 0.f; if (!(eA[m][nt].w > 0.f)) v.w = 0.f;
              }
              if (pB) { v.x += eB[m][nt].x; v.y += eB[m][nt].y; v.z += eB[m][nt].z; v.w += eB[m][nt].w; }
            }
            // (pinned here: the compiler would otherwise sink the arithmetic back into the predicated store blocks)
            asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));
            outv[m][nt] = v;
          }
        }
#pragma unroll
        for (int m = 0; m < MT; ++m) {
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            if (ook[m] && ch_ok[nt]) *reinterpret_cast<float4*>(out_n + ooff[m] + nt * 64u) = outv[m][nt];
        }
      }
    }
  }
}

// ---- host side ------------------------------------------------------------------------------ //
struct FwdPlan { bool ok; int MT, NT, TH, XV, grid; size_t lds; };

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
  // two budgets: the standard one (64 KB of LDS: 2+ workgroups per CU whatever else; 7 prefetch vectors), and -- only
  // when that leaves more than a fifth of the MFMA tile slots empty (32 input channels on 48-pixel rows: bands of 2
  // rows, 6 of 8 slots, 2x halo re-reads) -- a larger one (80 KB, 10 vectors)
  for (int tier = 0; tier < 2 && best_u < 0.8; ++tier) {
    const size_t lds_cap = tier ? 80 * 1024 : 64 * 1024, src_cap = (size_t)(tier ? 10 : 7) * 256 * 16;
    for (int th = 1; th <= th_hi; ++th) {
      const int thp = (th - 1) * p.stride + ext_h;
      const size_t x_b = (size_t)thp * p.twp * p.xs * 4;
      const size_t x_src = (size_t)thp * p.twp * p.cin * 4;                     // bytes prefetched in registers
      if (!(w_b + (size_t)total * 16 + x_b <= lds_cap && x_b < 65536 && (u8 || p.cin % 4 != 0 || x_src <= src_cap))) continue;
      const int nb = (oh_max + th - 1) / th, last = oh_max - (nb - 1) * th;
      const int t_full = (th * ow_max + 15) / 16, t_last = (last * ow_max + 15) / 16;
      for (int mt = 2; mt <= 5; ++mt) {
        const int round = 4 * mt;
        const double slots = (double)(nb - 1) * ((t_full + round - 1) / round) * round + (double)((t_last + round - 1) / round) * round;
        double u = ((double)(nb - 1) * t_full + t_last) / slots;
        u *= 1.0 - 0.02 * nb / (double)oh_max;                                   // fewer bands: fewer barriers / halo re-reads
        if (u > best_u + 1e-9) { best_u = u; best_th = th; best_mt = mt; }
      }
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
  {                                                          // prefetch registers the band needs: 5 or 7 float4
    const size_t x_src = (size_t)p.thp * p.twp * p.cin * 4;
    pl.XV = (!u8 && p.cin % 4 == 0 && x_src <= 5 * 256 * 16) ? 5 : (x_src <= 7 * 256 * 16 || u8 || p.cin % 4 != 0 ? 7 : 10);
  }
  int per_cu = (int)((160 * 1024) / pl.lds); if (per_cu > 4) per_cu = 4; if (per_cu < 1) per_cu = 1;
  const long long mg = 256LL * per_cu;
  pl.grid = (int)(p.ntiles < mg ? p.ntiles : mg);
  pl.ok = true;
  return pl;
}

inline int launch_fwd_kernel(const FwdParams& p, const FwdPlan& pl, hipStream_t s) {
  const bool dg = p.wmode == 1;
  if (dg ? (p.bias || p.residual || p.out_relu) : (p.mask || p.add))
    return fail(SEEDHIP_ERR_UNSUPPORTED, "halo_fwd: forward and data-gradient epilogues do not mix");
  if ((long long)p.OH * p.OW * p.ld_out * 4 >= (1LL << 31) || (long long)(p.thp + p.ih) * p.iw * p.ld_in * 4 >= (1LL << 31))
    return fail(SEEDHIP_ERR_UNSUPPORTED, "halo_fwd: one image exceeds the 32-bit offsets of the kernel");
#define SEEDHIP_HF3(MT_, NT_, DG_, XV_)                                                                          \
  {                                                                                                              \
    if (pl.lds > 64 * 1024)                                                                                      \
      (void)hipFuncSetAttribute((const void*)halo_fwd_kernel<MT_, NT_, DG_, XV_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.lds); \
    static int occ_lds = -1, occ = 0;            /* resident workgroups per CU of this variant at this LDS size */ \
    if (occ_lds != (int)pl.lds) {                                                                                \
      int o = 0;                                                                                                 \
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&o, (const void*)halo_fwd_kernel<MT_, NT_, DG_, XV_>, 256, pl.lds) != hipSuccess || o < 1) o = 1; \
      occ = o; occ_lds = (int)pl.lds;                                                                            \
    }                                                                                                            \
    const long long mg = 256LL * occ;                                                                            \
    const int grid = (int)(p.ntiles < mg ? p.ntiles : mg);                                                       \
    hipLaunchKernelGGL((halo_fwd_kernel<MT_, NT_, DG_, XV_>), dim3(grid), dim3(256), pl.lds, s, p);              \
    return check_launch("halo_fwd_kernel");                                                                      \
  }
#define SEEDHIP_HF2(MT_, NT_, DG_)                                                                               \
  { if (pl.XV == 5) SEEDHIP_HF3(MT_, NT_, DG_, 5) else if (pl.XV == 7) SEEDHIP_HF3(MT_, NT_, DG_, 7) else SEEDHIP_HF3(MT_, NT_, DG_, 10) }
#define SEEDHIP_HF(MT_, NT_)                                                                                     \
  if (pl.MT == MT_ && pl.NT == NT_) {                                                                            \
    if (dg) SEEDHIP_HF2(MT_, NT_, true) else SEEDHIP_HF2(MT_, NT_, false)                                        \
  }
  SEEDHIP_HF(4, 1) SEEDHIP_HF(4, 2) SEEDHIP_HF(2, 1) SEEDHIP_HF(2, 2) SEEDHIP_HF(3, 1) SEEDHIP_HF(3, 2)
  SEEDHIP_HF(5, 1) SEEDHIP_HF(5, 2)
#undef SEEDHIP_HF
#undef SEEDHIP_HF2
#undef SEEDHIP_HF3
  return fail(SEEDHIP_ERR_UNSUPPORTED, "halo_fwd: no kernel for MT=%d NT=%d", pl.MT, pl.NT);
}

}  // namespace halo
}  // namespace seedhip

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

struct FwdParams {
  const void* in; int in_dtype, in_relu;        // input [n_img, ih, iw, ld_in]
  const float* w;                                // original Keras kernel [w_kh, w_kw, w_cin, w_cout]
  int wmode;                                     // 0: forward, W_eff[tap][c][co] = w[tap][c][co]
                                                 // 1: data gradient class (w_py, w_px): W_eff[(j,i)][c][co] =
                                                 //    w[w_py + w_s*(kh-1-j)][w_px + w_s*(kw-1-i)][co][c]
  int w_kh, w_kw, w_cin, w_cout, w_py, w_px, w_s;
  int n_img, ih, iw, cin, kh, kw, stride, pad_t, pad_l;   // the convolution this launch computes (input side)
  int oh, ow, cout;                              // its output grid / channels
  float* out; int OH, OW, ld_out, so, oy0, ox0;  // placement: out[n, oy*so + oy0, ox*so + ox0, :]
  const float* bias; int out_relu; const float* residual;
  const float* mask; const float* add;           // data-gradient epilogue (indexed like out)
  int ld_in;
  int TH, bands, ntiles, thp, twp, xs;           // tiling; xs = LDS pixel stride (floats)
  int cgs;                                       // k-groups (of 4 channels) per tap = ceil(cin / 4), power of 2
  int cgs_shift;
  int nslices;                                   // ceil(kh*kw*cgs / 4)
  FastDiv d_ow;
};

template <int MT, int NT>
__global__ void __launch_bounds__(256)
halo_fwd_kernel(const FwdParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int w_floats = p.nslices * NT * 256;
  float* w_lds = smem;
  float* x_lds = smem + w_floats;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kq = lane >> 4, j = lane & 15;
  const int ntaps = p.kh * p.kw;

  // ---- weights -> LDS, once per workgroup: [slice][nt][lane][kk] ----
  for (int idx = tid; idx < w_floats; idx += 256) {
    const int kk = idx & 3, l = (idx >> 2) & 63, rest = idx >> 8;
    const int nt = rest % NT, slice = rest / NT;
    const int G = slice * 4 + (l >> 4);
    const int tap = G >> p.cgs_shift, c = ((G & (p.cgs - 1)) << 2) + kk;
    const int co = nt * 16 + (l & 15);
    float v = 0.f;
    if (tap < ntaps && c < p.cin && co < p.cout) {
      const int ty = tap / p.kw, tx = tap - ty * p.kw;
      if (p.wmode == 0) {
        v = p.w[((long long)tap * p.w_cin + c) * p.w_cout + co];
      } else {
        const int ky = p.w_py + p.w_s * (p.kh - 1 - ty), kx = p.w_px + p.w_s * (p.kw - 1 - tx);
        v = p.w[((long long)(ky * p.w_kw + kx) * p.w_cin + co) * p.w_cout + c];
      }
    }
    w_lds[idx] = v;
  }

  // ---- tile pipeline (as halo_wgrad.h) ----
  constexpr int kXV = 7;
  const bool vec = p.in_dtype == 0 && (p.cin & 3) == 0 && (p.ld_in & 3) == 0;
  float4 xr[kXV];
  auto band_of = [&](int tile, int& n, int& y0, int& th) {
    n = tile / p.bands;
    const int band = tile - n * p.bands;
    y0 = band * p.TH;
    th = (y0 + p.TH <= p.oh) ? p.TH : p.oh - y0;
  };
  auto load_tile = [&](int tile) {
    if (!vec) return;
    int n, y0, th; band_of(tile, n, y0, th);
    const int c4 = p.cin >> 2;
    const int per_row = p.twp * c4;
    const int nvec = ((th - 1) * p.stride + p.kh) * per_row;
#pragma unroll
    for (int u = 0; u < kXV; ++u) {
      const int v = tid + u * 256;
      float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
      if (v < nvec) {
        const int r = v / per_row, rem = v - r * per_row;
        const int xcol = rem / c4, cq = rem - xcol * c4;
        const int iy = y0 * p.stride - p.pad_t + r, ix = xcol - p.pad_l;
        if (iy >= 0 && iy < p.ih && ix >= 0 && ix < p.iw)
          val = *reinterpret_cast<const float4*>((const float*)p.in + (((long long)n * p.ih + iy) * p.iw + ix) * p.ld_in + 4 * cq);
      }
      xr[u] = val;
    }
  };
  auto store_tile = [&](int tile) {
    int n, y0, th; band_of(tile, n, y0, th);
    const int rowf = p.twp * p.xs;
    const int nrows = (th - 1) * p.stride + p.kh;
    if (vec) {
      const int c4 = p.cin >> 2;
      const int per_row = p.twp * c4;
#pragma unroll
      for (int u = 0; u < kXV; ++u) {
        const int v = tid + u * 256;
        if (v < nrows * per_row) {
          const int r = v / per_row, rem = v - r * per_row;
          const int xcol = rem / c4, cq = rem - xcol * c4;
          float4 val = xr[u];
          if (p.in_relu) { val.x = fmaxf(val.x, 0.f); val.y = fmaxf(val.y, 0.f); val.z = fmaxf(val.z, 0.f); val.w = fmaxf(val.w, 0.f); }
          *reinterpret_cast<float4*>(x_lds + r * rowf + xcol * p.xs + 4 * cq) = val;
        }
      }
    } else {                                           // u8 / odd channel counts: channels padded to 4*cgs with zeros
      const int cp = p.cgs << 2;
      const int per_row = p.twp * cp;
      for (int v = tid; v < nrows * per_row; v += 256) {
        const int r = v / per_row, rem = v - r * per_row;
        const int xcol = rem / cp, c = rem - xcol * cp;
        const int iy = y0 * p.stride - p.pad_t + r, ix = xcol - p.pad_l;
        float val = 0.f;
        if (c < p.cin && iy >= 0 && iy < p.ih && ix >= 0 && ix < p.iw) {
          const long long off = (((long long)n * p.ih + iy) * p.iw + ix) * p.ld_in + c;
          val = p.in_dtype == 1 ? (float)((const uint8_t*)p.in)[off] / 255.0f : ((const float*)p.in)[off];
          if (p.in_relu) val = fmaxf(val, 0.f);
        }
        x_lds[r * rowf + xcol * p.xs + c] = val;
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

    const int npix = th * p.ow;
    const int ntile16 = (npix + 15) >> 4;
    for (int t0 = wave * MT; t0 < ntile16; t0 += 4 * MT) {
      int xbase[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        int pix = (t0 + m) * 16 + j;
        if (pix > npix - 1) pix = npix - 1;
        uint32_t py, px;
        p.d_ow.divmod((uint32_t)pix, py, px);
        xbase[m] = ((int)py * p.stride * p.twp + (int)px * p.stride) * p.xs;
      }
      f32x4_t acc[MT][NT];
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[m][nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      for (int s = 0; s < p.nslices; ++s) {
        const int G = s * 4 + kq;
        int tap = G >> p.cgs_shift;
        const int cg = G & (p.cgs - 1);
        if (tap > ntaps - 1) tap = ntaps - 1;             // padded k-groups carry zero weights
        const int ty = tap / p.kw, tx = tap - ty * p.kw;
        const int koff = (ty * p.twp + tx) * p.xs + (cg << 2);
        f32x4_t a[NT], b[MT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) a[nt] = *reinterpret_cast<const f32x4_t*>(w_lds + ((s * NT + nt) * 64 + lane) * 4);
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
        uint32_t py, px;
        p.d_ow.divmod((uint32_t)pix, py, px);
        const int oy = (y0 + (int)py) * p.so + p.oy0, ox = (int)px * p.so + p.ox0;
        if (oy < 0 || oy >= p.OH || ox < 0 || ox >= p.OW) continue;
        const long long obase = (((long long)n * p.OH + oy) * p.OW + ox) * p.ld_out;
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

// ---- host side ------------------------------------------------------------------------------ //
struct FwdPlan { bool ok; int MT, NT, TH, grid, xs, cgs, cgs_shift, nslices, thp, twp; size_t lds; };

// The convolution as the kernel sees it: input dims, kernel, stride, pads, output grid, channels.
inline FwdPlan plan_fwd(int n_img, int ih, int iw, int cin, int kh, int kw, int stride, int oh, int ow, int cout,
                        int ld_in, int ld_out, bool u8) {
  FwdPlan pl; memset(&pl, 0, sizeof(pl));
  if (kh > 4 || kw > 4 || stride > 2 || cout > 32 || cout % 4 != 0 || ld_out % 4 != 0 || oh * ow < 16) return pl;
  if (!u8 && (cin % 4 != 0 || ld_in % 4 != 0) && cin > 4) return pl;
  int cgs = 1, sh = 0;
  while (cgs * 4 < cin) { cgs <<= 1; ++sh; }
  if (cgs > 16) return pl;
  if (!u8 && cin % 4 == 0 && cin != 4 * cgs) return pl;       // vector fill leaves no zeroed pad channels
  pl.cgs = cgs; pl.cgs_shift = sh;
  pl.nslices = (kh * kw * cgs + 3) / 4;
  pl.NT = (cout + 15) / 16;
  // LDS pixel stride in 16-B slots: >= cgs and (stride * S) % 4 == 2  => conflict-free ds_read_b128
  int S = cgs;
  while ((stride * S) % 4 != 2) ++S;
  pl.xs = 4 * S;
  pl.twp = (ow - 1) * stride + kw;
  const size_t w_b = (size_t)pl.nslices * pl.NT * 1024;
  int th = (256 + ow - 1) / ow; if (th > oh) th = oh;
  // prefer a band whose 16-pixel tile count is a multiple of 4 waves x MT
  for (;; --th) {
    const size_t x_b = (size_t)((th - 1) * stride + kh) * pl.twp * pl.xs * 4;
    const size_t x_src = (size_t)((th - 1) * stride + kh) * pl.twp * cin * 4;       // bytes prefetched in registers
    const bool fits = w_b + x_b <= 64 * 1024 && (u8 || cin % 4 != 0 || x_src <= 7 * 256 * 16);
    if (fits || th == 1) { if (!fits) return pl; pl.lds = w_b + x_b; break; }
  }
  pl.TH = th; pl.thp = (th - 1) * stride + kh;
  const int tiles16 = (th * ow + 15) / 16;
  pl.MT = tiles16 >= 12 ? 4 : 2;
  const int bands = (oh + th - 1) / th;
  const long long ntiles = (long long)n_img * bands;
  int per_cu = (int)((160 * 1024) / pl.lds); if (per_cu > 3) per_cu = 3; if (per_cu < 1) per_cu = 1;
  const long long mg = 256LL * per_cu;
  pl.grid = (int)(ntiles < mg ? ntiles : mg);
  pl.ok = true;
  return pl;
}

inline void fill_tiling(FwdParams& p, const FwdPlan& pl) {
  p.TH = pl.TH; p.bands = (p.oh + pl.TH - 1) / pl.TH; p.ntiles = p.n_img * p.bands;
  p.thp = pl.thp; p.twp = pl.twp; p.xs = pl.xs; p.cgs = pl.cgs; p.cgs_shift = pl.cgs_shift; p.nslices = pl.nslices;
  p.d_ow.init(p.ow);
}

inline int launch_fwd_kernel(const FwdParams& p, const FwdPlan& pl, hipStream_t s) {
#define SEEDHIP_HF(MT_, NT_)                                                                                     \
  if (pl.MT == MT_ && pl.NT == NT_) {                                                                            \
    if (pl.lds > 64 * 1024)                                                                                      \
      (void)hipFuncSetAttribute((const void*)halo_fwd_kernel<MT_, NT_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.lds); \
    hipLaunchKernelGGL((halo_fwd_kernel<MT_, NT_>), dim3(pl.grid), dim3(256), pl.lds, s, p);                     \
    return check_launch("halo_fwd_kernel");                                                                      \
  }
  SEEDHIP_HF(4, 1) SEEDHIP_HF(4, 2) SEEDHIP_HF(2, 1) SEEDHIP_HF(2, 2)
#undef SEEDHIP_HF
  return fail(SEEDHIP_ERR_UNSUPPORTED, "halo_fwd: no kernel for MT=%d NT=%d", pl.MT, pl.NT);
}

}  // namespace halo
}  // namespace seedhip

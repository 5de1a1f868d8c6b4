// Weight gradient of 3x3 stride-1 convolutions with the input staged ONCE per spatial tile.
//
// Used by seedhip_conv2d_bwd_weight for the ResNet stacks of
// /root/reference/dmlab/networks.py:31-60 (TF autodiff of Conv2D wrt kernel / bias).
//
//   dW[ky,kx,c,co] = sum_{n,oy,ox} X[n, oy+ky-pad, ox+kx-pad, c] * dY[n,oy,ox,co],  db[co] = sum dY
//
// The implicit-GEMM formulation gathers every input element 9 times from global memory with
// per-element index arithmetic (VALU-bound: 4-16% of the fp32 MFMA peak on these shapes).  Here
// a workgroup walks (image, row-band) tiles: the input band + halo (after ReLU / u8->/255
// conversion) and the dY band are copied to LDS once with plain coalesced float4 copies, and
// all 9 taps read them from LDS with addresses of the form  pixel_base + per-lane tap offset --
// no div/mod per operand.  MFMA roles (v_mfma_f32_16x16x4_f32, 4 pixels reduced per issue):
//   A: rows = 16 consecutive rows of dW viewed as [9*cin, cout] (row = tap*cin + c),
//      lane (i, kq) supplies X[pixel 4g+kq shifted by the row's tap][c];
//   B: cols = 16 output channels, lane (kq, j) supplies dY[pixel 4g+kq][co].
// The dW rows are split over MSPLIT waves (each MTW row-tiles), pixel groups over the other
// 4/MSPLIT waves; accumulators stay in registers across all tiles of the persistent workgroup,
// then waves are summed through LDS in a fixed order and ONE partial slice per workgroup goes
// to the workspace (reduced by reduce_slices: deterministic, no atomics).
#pragma once
#include "common.h"
#include "igemm.h"
#include "conv_launch.h"
#include "../../include/seedhip.h"

namespace seedhip {
namespace halo {

typedef float f32x4_t __attribute__((ext_vector_type(4)));

struct WgradParams {
  const void* in; int in_dtype, in_relu;
  const float* dy;
  float* partial_w;            // [grid][rows*cout]
  float* partial_b;            // [grid][cout] or null
  int n_img, ih, iw, cin, oh, ow, cout, ld_in, ld_out, pad;
  int TH;                      // output rows per tile
  int bands;                   // ceil(oh / TH)
  int ntiles;                  // n_img * bands
  int rows;                    // 9 * cin
  int xs;                      // LDS pixel stride of the X tile (floats)
  int twp;                     // tile width incl. halo = ow + 2
  FastDiv d_ow;
};

template <int MTW, int NT, int MSPLIT>
__global__ void __launch_bounds__(256)
halo_wgrad_kernel(const WgradParams p) {
  constexpr int PP = 4 / MSPLIT;                      // pixel partitions
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int x_floats = ((p.TH + 2) * p.twp * p.xs + 3) & ~3;
  float* xs_lds = smem;
  float* dy_lds = smem + x_floats;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kq = lane >> 4, i = lane & 15;
  const int ms = wave % MSPLIT, pp = wave / MSPLIT;
  const int coutp = NT * 16;

  // per-lane LDS offset of each of this wave's dW rows (tap shift + channel)
  int lane_off[MTW];
  bool row_ok[MTW];
#pragma unroll
  for (int mt = 0; mt < MTW; ++mt) {
    const int R = (ms * MTW + mt) * 16 + i;
    row_ok[mt] = R < p.rows;
    const int Rc = row_ok[mt] ? R : 0;
    const int tap = Rc / p.cin, c = Rc - tap * p.cin;
    lane_off[mt] = ((tap / 3) * p.twp + (tap % 3)) * p.xs + c;
  }
  f32x4_t acc[MTW][NT];
#pragma unroll
  for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  float bsum[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) bsum[nt] = 0.f;

  for (int tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
    const int n = tile / p.bands, band = tile - n * p.bands;
    const int y0 = band * p.TH;
    const int th = (y0 + p.TH <= p.oh) ? p.TH : p.oh - y0;
    __syncthreads();                                   // previous tile fully consumed
    // ---- X band + halo -> LDS (zero outside the image), converted once ----
    {
      const int rowf = p.twp * p.xs;                   // floats per LDS row
      const int nrows = th + 2;
      if (p.in_dtype == 0 && (p.cin & 3) == 0 && (p.ld_in & 3) == 0) {
        const int c4 = p.cin >> 2;
        const int per_row = p.twp * c4;
        for (int v = tid; v < nrows * per_row; v += 256) {
          const int r = v / per_row, rem = v - r * per_row;
          const int xcol = rem / c4, cq = rem - xcol * c4;
          const int iy = y0 - p.pad + r, ix = xcol - p.pad;
          float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
          if (iy >= 0 && iy < p.ih && ix >= 0 && ix < p.iw) {
            val = *reinterpret_cast<const float4*>((const float*)p.in + (((long long)n * p.ih + iy) * p.iw + ix) * p.ld_in + 4 * cq);
            if (p.in_relu) { val.x = fmaxf(val.x, 0.f); val.y = fmaxf(val.y, 0.f); val.z = fmaxf(val.z, 0.f); val.w = fmaxf(val.w, 0.f); }
          }
          *reinterpret_cast<float4*>(xs_lds + r * rowf + xcol * p.xs + 4 * cq) = val;
        }
      } else {
        const int per_row = p.twp * p.cin;
        for (int v = tid; v < nrows * per_row; v += 256) {
          const int r = v / per_row, rem = v - r * per_row;
          const int xcol = rem / p.cin, c = rem - xcol * p.cin;
          const int iy = y0 - p.pad + r, ix = xcol - p.pad;
          float val = 0.f;
          if (iy >= 0 && iy < p.ih && ix >= 0 && ix < p.iw) {
            const long long off = (((long long)n * p.ih + iy) * p.iw + ix) * p.ld_in + c;
            val = p.in_dtype == 1 ? (float)((const uint8_t*)p.in)[off] / 255.0f : ((const float*)p.in)[off];
            if (p.in_relu) val = fmaxf(val, 0.f);
          }
          xs_lds[r * rowf + xcol * p.xs + c] = val;
        }
      }
    }
    // ---- dY band -> LDS ----
    {
      const int c4 = coutp >> 2;
      const int npix = th * p.ow;
      const float* src = p.dy + ((long long)n * p.oh + y0) * p.ow * p.ld_out;
      for (int v = tid; v < npix * c4; v += 256) {
        const int pix = v / c4, cq = v - pix * c4;
        *reinterpret_cast<float4*>(dy_lds + pix * coutp + 4 * cq) =
            *reinterpret_cast<const float4*>(src + (long long)pix * p.ld_out + 4 * cq);
      }
    }
    __syncthreads();
    // ---- MFMA over this wave's pixel groups ----
    const int G = (th * p.ow) >> 2;                    // ow % 4 == 0
    for (int g = pp; g < G; g += PP) {
      const int pix = 4 * g + kq;
      uint32_t py, px;
      p.d_ow.divmod((uint32_t)pix, py, px);
      const float* xb = xs_lds + ((int)py * p.twp + (int)px) * p.xs;
      float b[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) { b[nt] = dy_lds[pix * coutp + nt * 16 + i]; if (ms == 0) bsum[nt] += b[nt]; }
      float a[MTW];
#pragma unroll
      for (int mt = 0; mt < MTW; ++mt) a[mt] = row_ok[mt] ? xb[lane_off[mt]] : 0.f;
#pragma unroll
      for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt], b[nt], acc[mt][nt], 0, 0, 0);
    }
  }

  // ---- reduce the PP pixel partitions through LDS (fixed order), write one slice ----
  __syncthreads();
  float* red = smem;                                   // [MSPLIT*MTW*16 rows][coutp]
  for (int q = 0; q < PP; ++q) {
    if (pp == q) {
#pragma unroll
      for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = (ms * MTW + mt) * 16 + 4 * kq + r;
            float* d = red + row * coutp + nt * 16 + i;
            *d = (q == 0) ? acc[mt][nt][r] : *d + acc[mt][nt][r];
          }
    }
    __syncthreads();
  }
  float* pw = p.partial_w + (long long)blockIdx.x * p.rows * p.cout;
  for (int v = tid; v < p.rows * coutp; v += 256) {
    const int row = v / coutp, co = v - row * coutp;
    if (co < p.cout) pw[row * p.cout + co] = red[v];
  }
  if (p.partial_b) {
    __syncthreads();
    float* rb = smem;                                  // [4 waves][coutp]
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      float s = bsum[nt];
      s += __shfl_xor(s, 16, 64);
      s += __shfl_xor(s, 32, 64);
      if (lane < 16) rb[wave * coutp + nt * 16 + lane] = s;
    }
    __syncthreads();
    if (tid < p.cout) {
      float s = 0.f;
      for (int w = 0; w < 4; ++w) s += rb[w * coutp + tid];     // waves with ms != 0 hold zeros
      p.partial_b[(long long)blockIdx.x * p.cout + tid] = s;
    }
  }
}

// ---- host side ------------------------------------------------------------------------------ //
struct WgradPlan { bool ok; int MTW, NT, MSPLIT, TH, grid; size_t lds; size_t ws_bytes; };

inline WgradPlan plan_wgrad(const seedhip_conv_geom* g) {
  WgradPlan pl; memset(&pl, 0, sizeof(pl));
  if (!(g->kh == 3 && g->kw == 3 && g->stride == 1 && g->pad_t == g->pad_l && (g->pad_t == 0 || g->pad_t == 1))) return pl;
  if (g->ow % 4 != 0 || g->cout % 16 != 0 || g->cout > 64 || g->ld_out % 4 != 0) return pl;
  if (g->oh != g->ih + 2 * g->pad_t - 2 || g->ow != g->iw + 2 * g->pad_l - 2) return pl;
  const int rows = 9 * g->cin;
  const int MT = (rows + 15) / 16;
  int msplit = 1;
  if (MT > 9) msplit = (MT + 8) / 9;                   // <= 9 row-tiles per wave
  if (msplit == 3) msplit = 4;
  if (msplit > 4) return pl;
  pl.MSPLIT = msplit;
  pl.MTW = (MT + msplit - 1) / msplit;
  pl.NT = g->cout / 16;
  if (!((pl.MTW == 2 && pl.NT == 1) || (pl.MTW == 9 && (pl.NT == 1 || pl.NT == 2 || pl.NT == 4)))) return pl;
  if (pl.MTW == 9 && pl.NT == 4 && pl.MSPLIT != 4) return pl;
  if (pl.MTW == 9 && pl.NT <= 2 && pl.MSPLIT > 2) return pl;
  // rows per band: ~192-256 output pixels
  int th = 256 / g->ow; if (th < 1) th = 1; if (th > g->oh) th = g->oh;
  const int twp = g->ow + 2;
  for (;; --th) {
    const size_t x_b = (size_t)(th + 2) * twp * g->cin * 4, dy_b = (size_t)th * g->ow * g->cout * 4;
    size_t red_b = (size_t)msplit * pl.MTW * 16 * g->cout * 4;
    size_t need = x_b + dy_b + 16; if (need < red_b) need = red_b;
    if (need <= 64 * 1024 || th == 1) { pl.lds = need; break; }
  }
  if (pl.lds > 150 * 1024) return pl;
  pl.TH = th;
  const int bands = (g->oh + th - 1) / th;
  const long long ntiles = (long long)g->n_img * bands;
  int per_cu = (int)((160 * 1024) / pl.lds); if (per_cu > 2) per_cu = 2; if (per_cu < 1) per_cu = 1;
  const long long mg = 256LL * per_cu;
  pl.grid = (int)(ntiles < mg ? ntiles : mg);
  pl.ws_bytes = (size_t)pl.grid * ((size_t)rows * g->cout + g->cout) * sizeof(float);
  pl.ok = true;
  return pl;
}

inline int launch_wgrad(const seedhip_conv_geom* g, const WgradPlan& pl, const void* in, int in_dtype, int in_relu,
                        const float* dy, float* dw, float* dbias, void* workspace, hipStream_t s) {
  WgradParams p;
  p.in = in; p.in_dtype = in_dtype; p.in_relu = in_relu; p.dy = dy;
  p.n_img = g->n_img; p.ih = g->ih; p.iw = g->iw; p.cin = g->cin; p.oh = g->oh; p.ow = g->ow; p.cout = g->cout;
  p.ld_in = g->ld_in; p.ld_out = g->ld_out; p.pad = g->pad_t;
  p.TH = pl.TH; p.bands = (g->oh + pl.TH - 1) / pl.TH; p.ntiles = g->n_img * p.bands;
  p.rows = 9 * g->cin; p.xs = g->cin; p.twp = g->ow + 2;
  p.d_ow.init(g->ow);
  p.partial_w = (float*)workspace;
  p.partial_b = dbias ? (float*)workspace + (size_t)pl.grid * p.rows * g->cout : nullptr;
#define SEEDHIP_HALO_LAUNCH(MTW_, NT_, MS_)                                                                        \
  do {                                                                                                             \
    if (pl.lds > 64 * 1024)                                                                                        \
      (void)hipFuncSetAttribute((const void*)halo_wgrad_kernel<MTW_, NT_, MS_>,                                    \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.lds);                          \
    hipLaunchKernelGGL((halo_wgrad_kernel<MTW_, NT_, MS_>), dim3(pl.grid), dim3(256), pl.lds, s, p);               \
  } while (0)
  if (pl.MTW == 2) SEEDHIP_HALO_LAUNCH(2, 1, 1);
  else if (pl.NT == 1 && pl.MSPLIT == 1) SEEDHIP_HALO_LAUNCH(9, 1, 1);
  else if (pl.NT == 2 && pl.MSPLIT == 1) SEEDHIP_HALO_LAUNCH(9, 2, 1);
  else if (pl.NT == 1 && pl.MSPLIT == 2) SEEDHIP_HALO_LAUNCH(9, 1, 2);
  else if (pl.NT == 2 && pl.MSPLIT == 2) SEEDHIP_HALO_LAUNCH(9, 2, 2);
  else SEEDHIP_HALO_LAUNCH(9, 4, 4);
#undef SEEDHIP_HALO_LAUNCH
  int rc = check_launch("halo_wgrad_kernel"); if (rc) return rc;
  reduce_slices(p.partial_w, pl.grid, (long long)p.rows * g->cout, dw, s);
  if (dbias) reduce_slices(p.partial_b, pl.grid, g->cout, dbias, s);
  return check_launch("halo_wgrad reduce");
}

}  // namespace halo
}  // namespace seedhip

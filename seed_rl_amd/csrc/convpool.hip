// First ImpalaDeep stage fused: Conv2D(16, 3, 'same') on uint8 frames (x/255) + MaxPool2D(3, 2, 'same'), forward and
// weight gradient, without ever materialising the pre-pool activation.
//
// Replaces /root/reference/dmlab/networks.py:31-37 (conv -> max-pool of stack 0, with the x/255 of :98-100) and the TF
// autodiff of that pair wrt the conv kernel / bias (MaxPoolGrad -> Conv2DBackpropFilter).
//
// Why fused: at T=20, B=256 the 72x96x16 fp32 pre-pool tensor is 2.4 GB.  Unfused it is written by the conv, read by
// the pool, its gradient written by the pool backward and read by the conv weight gradient: ~10 GB of HBM traffic
// (6.3 ms measured) around 32 GFLOP of arithmetic.  Fused, the step reads the uint8 frames (111 MB) and writes /
// reads only the pooled tensor (0.6 GB) and the argmax bytes (0.15 GB).
//
// The conv has K = 27 and 16 output channels: too thin for the matrix cores to matter, so both kernels are plain
// fp32 FMA on the vector ALU (v_pk_fma_f32: two channels per instruction; FMA chain in (ky, kx, c) order):
//   * a workgroup walks (image, band of 4 pooled rows) tiles; the band's input rows + halo are staged in LDS as fp32
//     (x/255, zero padding, channels padded to 4 so that a pixel's 3x3x3 window is nine 16-byte reads);
//   * wave w owns output channels 4w..4w+3 and keeps their 27 x 4 weights (forward) or 27 x 4 gradient accumulators
//     (backward, across ALL tiles of the persistent workgroup) in registers; lanes are pixels;
//   * forward: conv band -> LDS -> 3x3/2 max with TF 'SAME' windows and first-max argmax (same byte code as
//     pool.hip) -> pooled tensor + argmax;
//   * backward: pooled gradient + argmax of the band (+ one pooled row above: windows overlap) staged in LDS; per conv
//     pixel the pre-pool gradient is gathered from its <= 4 windows (no atomics) and fed to the dW/db FMAs; at the end the 64
//     lanes are reduced with a fixed shuffle tree and ONE partial slice per workgroup is written (deterministic
//     second-pass reduction, conv_launch.h).
#include "common.h"
#include "conv_launch.h"
#include "igemm.h"
#include "../../include/seedhip.h"

namespace {

constexpr int CIN = 3, COUT = 16, PB = 4;                 // pooled rows per band
// v_pk_fma_f32: two fp32 FMAs per lane per instruction (the 157 TF/s vector peak is the packed rate)
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 pk_fma(float x, f2 w, f2 a) { return __builtin_elementwise_fma(f2{x, x}, w, a); }
constexpr int kConvRows = 2 * PB + 1, kInRows = 2 * PB + 3;

struct Geom {
  int n, ih, iw;                                          // conv map (input and conv output: 'same', stride 1)
  int ph, pw, pt, pl;                                     // pooled map, TF 'SAME' pads of the pool
  int bands, ntiles;
  seedhip::FastDiv d_iw, d_wp, d_pw, d_pw4, d_bands;     // the loops divide by these every iteration: mul-hi instead
};

__host__ __device__ inline int xin_floats(int iw) { return kInRows * (iw + 2) * 4 + 256; }     // + the b/255 table
__host__ __device__ inline int cbuf_floats(int iw) { return kConvRows * iw * COUT; }

// Input rows [r0, r0 + kInRows) of image n -> LDS fp32 [row][col + 1][4], x/255, zeros outside the map / 4th channel.
// Split in two so that the bytes of the NEXT tile fly while the current one computes: fetch (global -> registers,
// 3 bytes + a valid flag packed per pixel) and commit (registers -> LDS).
constexpr int kInPre = 5;                                 // pixels per thread: kInRows * (iw + 2) <= 5 * 256 for iw <= 114
struct InPrefetch { uint32_t v[kInPre]; };
__device__ __forceinline__ void fetch_input(const Geom& g, const uint8_t* __restrict__ x, int n, int r0, InPrefetch& pre, int tid) {
  const int wp = g.iw + 2;
#pragma unroll
  for (int u = 0; u < kInPre; ++u) {
    const int idx = tid + u * 256;
    uint32_t v = 0;
    if (idx < kInRows * wp) {
      uint32_t r, c;
      g.d_wp.divmod((uint32_t)idx, r, c);
      const int iy = r0 + (int)r, ix = (int)c - 1;
      if (iy >= 0 && iy < g.ih && ix >= 0 && ix < g.iw) {
        const uint8_t* s = x + (((long long)n * g.ih + iy) * g.iw + ix) * CIN;
        v = (uint32_t)s[0] | ((uint32_t)s[1] << 8) | ((uint32_t)s[2] << 16);
      }
    }
    pre.v[u] = v;
  }
}
// lut[b] = b / 255 (the correctly rounded fp32 quotient the reference computes), built once per workgroup
__device__ __forceinline__ void commit_input(const Geom& g, const InPrefetch& pre, float* xin, const float* lut, int tid) {
  const int wp = g.iw + 2;
#pragma unroll
  for (int u = 0; u < kInPre; ++u) {
    const int idx = tid + u * 256;
    if (idx < kInRows * wp) {
      const uint32_t v = pre.v[u];                        // out-of-map pixels hold 0 = the 'same' zero padding
      *reinterpret_cast<float4*>(xin + idx * 4) =
          make_float4(lut[v & 255u], lut[(v >> 8) & 255u], lut[(v >> 16) & 255u], 0.f);
    }
  }
}

__device__ __forceinline__ void load_window(const float* xin, int wp, int r, int xcol, float4 (&win)[3][3]) {
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
      win[ky][kx] = *reinterpret_cast<const float4*>(xin + ((r + ky) * wp + xcol + kx) * 4);
}

__global__ void __launch_bounds__(256)
convpool_fwd_kernel(const Geom g, const uint8_t* __restrict__ x, const float* __restrict__ w,
                    const float* __restrict__ bias, float* __restrict__ pooled, uint8_t* __restrict__ argmax) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* xin = smem;
  float* lut = smem + xin_floats(g.iw) - 256;
  float* cbuf = smem + xin_floats(g.iw);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wp = g.iw + 2;
  lut[tid] = (float)tid / 255.0f;                         // visible after the first barrier of the tile loop

  f2 wr[3][3][CIN][2];                                    // this wave's 4 output channels, as two pairs
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
      for (int c = 0; c < CIN; ++c) {
        const float4 t = *reinterpret_cast<const float4*>(w + ((ky * 3 + kx) * CIN + c) * COUT + 4 * wave);
        wr[ky][kx][c][0] = f2{t.x, t.y}; wr[ky][kx][c][1] = f2{t.z, t.w};
      }
  float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (bias) b4 = *reinterpret_cast<const float4*>(bias + 4 * wave);

  InPrefetch pre;
  if ((int)blockIdx.x < g.ntiles) {
    uint32_t n, band;
    g.d_bands.divmod(blockIdx.x, n, band);
    fetch_input(g, x, (int)n, 2 * (int)band * PB - g.pt - 1, pre, tid);
  }
  for (int tile = blockIdx.x; tile < g.ntiles; tile += gridDim.x) {
    uint32_t un, uband;
    g.d_bands.divmod((uint32_t)tile, un, uband);
    const int n = (int)un, band = (int)uband;
    const int i0 = band * PB;                             // first pooled row of the band
    const int cy0 = 2 * i0 - g.pt;                        // first conv row the band's windows touch
    __syncthreads();                                      // previous tile's pool phase is done with cbuf / conv with xin
    commit_input(g, pre, xin, lut, tid);
    __syncthreads();
    if (tile + (int)gridDim.x < g.ntiles) {               // next tile's bytes fly under this tile's conv + pool
      uint32_t n2, band2;
      g.d_bands.divmod((uint32_t)(tile + gridDim.x), n2, band2);
      fetch_input(g, x, (int)n2, 2 * (int)band2 * PB - g.pt - 1, pre, tid);
    }
    // ---- conv rows cy0 .. cy0 + kConvRows - 1 (rows outside the map are never read by the pool) ----
    for (int pix = lane; pix < kConvRows * g.iw; pix += 64) {
      uint32_t ur, uxc;
      g.d_iw.divmod((uint32_t)pix, ur, uxc);
      const int r = (int)ur, xc = (int)uxc;
      float4 win[3][3];
      load_window(xin, wp, r, xc, win);
      f2 a01 = f2{b4.x, b4.y}, a23 = f2{b4.z, b4.w};
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const float xv[3] = {win[ky][kx].x, win[ky][kx].y, win[ky][kx].z};
#pragma unroll
          for (int c = 0; c < CIN; ++c) {
            a01 = pk_fma(xv[c], wr[ky][kx][c][0], a01);
            a23 = pk_fma(xv[c], wr[ky][kx][c][1], a23);
          }
        }
      // cbuf is channel-quad major [quad][row][x] x 16 B: lanes (consecutive x) write consecutive 16-byte slots
      reinterpret_cast<float4*>(cbuf)[(wave * kConvRows + r) * g.iw + xc] = make_float4(a01[0], a01[1], a23[0], a23[1]);
    }
    __syncthreads();
    // ---- 3x3 / 2 max-pool of the band out of LDS: item = (pooled pixel, channel quad) ----
    const int rows = (i0 + PB <= g.ph) ? PB : g.ph - i0;
    for (int item = tid; item < rows * g.pw * 4; item += 256) {
      uint32_t ucq, pp, upi, upj;                          // item = (quad, pooled row, pooled col): lanes share the quad
      g.d_pw.divmod((uint32_t)item, pp, upj);              // pp = quad * rows + pi
      const int pj = (int)upj;
      ucq = pp / (uint32_t)rows;
      upi = pp - ucq * (uint32_t)rows;
      const int cq = (int)ucq, pi = (int)upi;
      float4 best = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
      int bi0 = 0, bi1 = 0, bi2 = 0, bi3 = 0;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const int iy = 2 * (i0 + pi) - g.pt + ky;
        if (iy < 0 || iy >= g.ih) continue;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int ix = 2 * pj - g.pl + kx;
          if (ix < 0 || ix >= g.iw) continue;
          const float4 v = reinterpret_cast<const float4*>(cbuf)[(cq * kConvRows + (iy - cy0)) * g.iw + ix];
          const int code = ky * 3 + kx;
          if (v.x > best.x) { best.x = v.x; bi0 = code; }
          if (v.y > best.y) { best.y = v.y; bi1 = code; }
          if (v.z > best.z) { best.z = v.z; bi2 = code; }
          if (v.w > best.w) { best.w = v.w; bi3 = code; }
        }
      }
      const long long o = (((long long)n * g.ph + i0 + pi) * g.pw + pj) * 4 + cq;
      reinterpret_cast<float4*>(pooled)[o] = best;
      reinterpret_cast<uchar4*>(argmax)[o] = make_uchar4((uint8_t)bi0, (uint8_t)bi1, (uint8_t)bi2, (uint8_t)bi3);
    }
  }
}

// Backward.  Band b owns conv rows [2*i0 - pt, 2*i0 - pt + 2*PB) (no overlap between bands); their pre-pool gradient
// gathers from pooled rows i0-1 .. i0+PB-1, which are staged in LDS (gradient + argmax) with the input rows.
__host__ __device__ inline int dyp_floats(int pw) { return (PB + 1) * pw * COUT; }

__global__ void __launch_bounds__(256)
convpool_bwd_kernel(const Geom g, const uint8_t* __restrict__ x, const float* __restrict__ dpooled,
                    const uint8_t* __restrict__ argmax, float* __restrict__ partial_w, float* __restrict__ partial_b) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* xin = smem;
  float* lut = smem + xin_floats(g.iw) - 256;
  float* dyp = smem + xin_floats(g.iw);                               // [(PB+1) pooled rows][pw][COUT]
  uint8_t* argl = reinterpret_cast<uint8_t*>(dyp + dyp_floats(g.pw)); // same shape, bytes
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wp = g.iw + 2;
  lut[tid] = (float)tid / 255.0f;

  f2 acc[3][3][CIN][2];
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
      for (int c = 0; c < CIN; ++c) { acc[ky][kx][c][0] = f2{0.f, 0.f}; acc[ky][kx][c][1] = f2{0.f, 0.f}; }
  float accb[4] = {0.f, 0.f, 0.f, 0.f};

  // pooled rows i0-1 .. i0+PB-1 (gradient and argmax), 16 bytes / 4 bytes per (pixel, channel quad): prefetched like
  // the input bytes ((PB+1) * pw * 4 <= 5 * 256 items for pw <= 64)
  InPrefetch pre;
  float4 pd[kInPre];
  uchar4 pa[kInPre];
  auto fetch_tile = [&](int t) {
    uint32_t un, uband;
    g.d_bands.divmod((uint32_t)t, un, uband);
    const int n = (int)un, i0 = (int)uband * PB;
    fetch_input(g, x, n, 2 * i0 - g.pt - 1, pre, tid);
#pragma unroll
    for (int u = 0; u < kInPre; ++u) {
      const int idx = tid + u * 256;
      float4 d = make_float4(0.f, 0.f, 0.f, 0.f);
      uchar4 am = make_uchar4(255, 255, 255, 255);
      if (idx < (PB + 1) * g.pw * 4) {
        uint32_t pr, rest;
        g.d_pw4.divmod((uint32_t)idx, pr, rest);
        const int oy = i0 - 1 + (int)pr;
        if (oy >= 0 && oy < g.ph) {
          const long long o = ((long long)n * g.ph + oy) * g.pw * 4 + rest;
          d = reinterpret_cast<const float4*>(dpooled)[o];
          am = reinterpret_cast<const uchar4*>(argmax)[o];
        }
      }
      pd[u] = d; pa[u] = am;
    }
  };
  if ((int)blockIdx.x < g.ntiles) fetch_tile(blockIdx.x);
  for (int tile = blockIdx.x; tile < g.ntiles; tile += gridDim.x) {
    uint32_t un, uband;
    g.d_bands.divmod((uint32_t)tile, un, uband);
    const int band = (int)uband;
    const int i0 = band * PB;
    const int cy0 = 2 * i0 - g.pt;                        // first conv row owned by the band
    int crow = 2 * PB;                                    // conv rows owned: the last band takes what is left
    if (band == g.bands - 1) crow = g.ih - cy0;
    const int cfirst = cy0 < 0 ? 0 : cy0;
    __syncthreads();
    commit_input(g, pre, xin, lut, tid);
#pragma unroll
    for (int u = 0; u < kInPre; ++u) {
      const int idx = tid + u * 256;
      if (idx < (PB + 1) * g.pw * 4) {                      // global order (row, col, quad) -> LDS [quad][row][col]
        const int q = idx & 3, rc = idx >> 2;
        reinterpret_cast<float4*>(dyp)[q * (PB + 1) * g.pw + rc] = pd[u];
        reinterpret_cast<uchar4*>(argl)[q * (PB + 1) * g.pw + rc] = pa[u];
      }
    }
    __syncthreads();
    if (tile + (int)gridDim.x < g.ntiles) fetch_tile(tile + gridDim.x);
    // ---- per owned conv pixel: pre-pool gradient of this wave's 4 channels (gather over <= 4 windows, out of LDS),
    //      then dW[ky][kx][c][4w..4w+3] += x(window) * dpre; db += dpre ----
    for (int pix = lane; pix < (cy0 + crow - cfirst) * g.iw; pix += 64) {
      uint32_t urr, uix;
      g.d_iw.divmod((uint32_t)pix, urr, uix);
      const int ix = (int)uix, iy = cfirst + (int)urr;
      float4 d = make_float4(0.f, 0.f, 0.f, 0.f);
      const int y0 = iy + g.pt, x0 = ix + g.pl;
      for (int oy = (y0 - 1) >> 1; oy <= (y0 >> 1); ++oy) {
        if (oy < 0 || oy >= g.ph) continue;
        const int ky = y0 - 2 * oy;
        if (ky < 0 || ky > 2) continue;
        for (int ox = (x0 - 1) >> 1; ox <= (x0 >> 1); ++ox) {
          if (ox < 0 || ox >= g.pw) continue;
          const int kx = x0 - 2 * ox;
          if (kx < 0 || kx > 2) continue;
          const int o = (wave * (PB + 1) + oy - (i0 - 1)) * g.pw + ox;
          const uchar4 am = reinterpret_cast<const uchar4*>(argl)[o];
          const float4 dv = reinterpret_cast<const float4*>(dyp)[o];
          const int code = ky * 3 + kx;
          if (am.x == code) d.x += dv.x;
          if (am.y == code) d.y += dv.y;
          if (am.z == code) d.z += dv.z;
          if (am.w == code) d.w += dv.w;
        }
      }
      float4 win[3][3];
      load_window(xin, wp, iy - cy0, ix, win);             // conv row iy: input rows iy-1 .. iy+1 = xin rows (iy - cy0) ..
      accb[0] += d.x; accb[1] += d.y; accb[2] += d.z; accb[3] += d.w;
      const f2 d01 = f2{d.x, d.y}, d23 = f2{d.z, d.w};
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const float xv[3] = {win[ky][kx].x, win[ky][kx].y, win[ky][kx].z};
#pragma unroll
          for (int c = 0; c < CIN; ++c) {
            acc[ky][kx][c][0] = pk_fma(xv[c], d01, acc[ky][kx][c][0]);
            acc[ky][kx][c][1] = pk_fma(xv[c], d23, acc[ky][kx][c][1]);
          }
        }
    }
  }

  // ---- lanes -> one value (fixed xor tree), one partial slice per workgroup ----
  float* pw = partial_w + (long long)blockIdx.x * 27 * COUT;
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
      for (int c = 0; c < CIN; ++c)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float v = acc[ky][kx][c][j >> 1][j & 1];
#pragma unroll
          for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
          if (lane == 0) pw[((ky * 3 + kx) * CIN + c) * COUT + 4 * wave + j] = v;
        }
  if (partial_b) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float v = accb[j];
#pragma unroll
      for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
      if (lane == 0) partial_b[(long long)blockIdx.x * COUT + 4 * wave + j] = v;
    }
  }
}

int make_geom(int n, int ih, int iw, int cin, int cout, Geom* g, const char* what) {
  SEEDHIP_REQUIRE(cin == CIN && cout == COUT, "%s: built for %d input and %d output channels", what, CIN, COUT);
  SEEDHIP_REQUIRE(n >= 1 && ih >= 3 && iw >= 3 && iw <= 114, "%s: need n >= 1, ih >= 3, 3 <= iw <= 114", what);
  g->n = n; g->ih = ih; g->iw = iw;
  g->ph = (ih + 1) / 2; g->pw = (iw + 1) / 2;
  const int padh = (g->ph - 1) * 2 + 3 - ih, padw = (g->pw - 1) * 2 + 3 - iw;
  g->pt = (padh > 0 ? padh : 0) / 2; g->pl = (padw > 0 ? padw : 0) / 2;
  g->bands = (g->ph + PB - 1) / PB; g->ntiles = n * g->bands;
  g->d_iw.init(iw); g->d_wp.init(iw + 2); g->d_pw.init(g->pw); g->d_pw4.init(g->pw * 4); g->d_bands.init(g->bands);
  return SEEDHIP_OK;
}
int grid_for(const Geom& g) { return g.ntiles < 512 ? g.ntiles : 512; }

}  // namespace

extern "C" int seedhip_conv3x3_u8_pool_fwd(const uint8_t* x, int n, int ih, int iw, int cin, const float* w,
                                           const float* bias, int cout, float* pooled, uint8_t* argmax, void* stream) {
  Geom g;
  int rc = make_geom(n, ih, iw, cin, cout, &g, "conv3x3_u8_pool_fwd"); if (rc) return rc;
  SEEDHIP_REQUIRE(x && w && pooled && argmax, "conv3x3_u8_pool_fwd: null pointer");
  SEEDHIP_REQUIRE(((((uintptr_t)w) | ((uintptr_t)bias) | ((uintptr_t)pooled)) & 15) == 0 && (((uintptr_t)argmax) & 3) == 0,
                  "conv3x3_u8_pool_fwd: w / bias / pooled must be 16-byte aligned, argmax 4-byte aligned");
  const size_t lds = (size_t)(xin_floats(iw) + cbuf_floats(iw)) * sizeof(float);
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute((const void*)convpool_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(convpool_fwd_kernel, dim3(grid_for(g)), dim3(256), lds, (hipStream_t)stream, g, x, w, bias, pooled,
                     argmax);
  return seedhip::check_launch("convpool_fwd_kernel");
}

extern "C" size_t seedhip_conv3x3_u8_pool_bwd_workspace_bytes(int n, int ih, int iw) {
  Geom g;
  if (make_geom(n, ih, iw, CIN, COUT, &g, "conv3x3_u8_pool_bwd")) return 0;
  return (size_t)grid_for(g) * (27 * COUT + COUT) * sizeof(float);
}

extern "C" int seedhip_conv3x3_u8_pool_bwd(const uint8_t* x, int n, int ih, int iw, int cin, const float* dpooled,
                                           const uint8_t* argmax, int cout, float* dw, float* dbias, void* workspace,
                                           size_t workspace_bytes, void* stream) {
  Geom g;
  int rc = make_geom(n, ih, iw, cin, cout, &g, "conv3x3_u8_pool_bwd"); if (rc) return rc;
  SEEDHIP_REQUIRE(x && dpooled && argmax && dw && workspace, "conv3x3_u8_pool_bwd: null pointer");
  SEEDHIP_REQUIRE((((uintptr_t)dpooled) & 15) == 0 && (((uintptr_t)argmax) & 3) == 0,
                  "conv3x3_u8_pool_bwd: dpooled must be 16-byte aligned, argmax 4-byte aligned");
  SEEDHIP_REQUIRE(workspace_bytes >= seedhip_conv3x3_u8_pool_bwd_workspace_bytes(n, ih, iw),
                  "conv3x3_u8_pool_bwd: workspace too small");
  const int grid = grid_for(g);
  float* pw = (float*)workspace;
  float* pb = dbias ? pw + (size_t)grid * 27 * COUT : nullptr;
  const size_t lds = (size_t)(xin_floats(iw) + dyp_floats(g.pw)) * sizeof(float) + (size_t)(PB + 1) * g.pw * COUT;
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute((const void*)convpool_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(convpool_bwd_kernel, dim3(grid), dim3(256), lds, s, g, x, dpooled, argmax, pw, pb);
  rc = seedhip::check_launch("convpool_bwd_kernel"); if (rc) return rc;
  seedhip::reduce_slices2(pw, 27LL * COUT, dw, pb, COUT, dbias, grid, s);
  return seedhip::check_launch("conv3x3_u8_pool_bwd");
}

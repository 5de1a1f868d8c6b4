// MaxPool2D(pool 3, strides 2, padding 'same') forward / backward, NHWC fp32.
//
// Replaces the Keras MaxPool2D of /root/reference/dmlab/networks.py:36-37 (and its TF
// autodiff, MaxPoolGrad).  TF 'SAME' semantics: out = ceil(in/2),
// pad_total = (out-1)*2 + 3 - in, pad_before = pad_total/2 (0 for even sizes, 1 for odd),
// padding value -inf; window i covers rows [2i - pad_before, 2i - pad_before + 2].
// The forward also records the window argmax (0..8, first max in row-major window order, as
// TF/Eigen do) so that the backward is a pure gather: every input pixel looks at the <= 4
// windows covering it -- no atomics, deterministic.
// HBM-bound: forward reads 4C B/pixel in, writes (4C+C)/4 B/pixel; one thread owns 4 channels.
#include "common.h"
#include "../../include/seedhip.h"

namespace {

struct PoolGeom { int n, ih, iw, c, oh, ow, pt, pl; };

__global__ void __launch_bounds__(256)
maxpool_fwd_kernel(PoolGeom g, const float* __restrict__ x, float* __restrict__ y, uint8_t* __restrict__ arg) {
  const int c4 = g.c >> 2;
  const long long total = (long long)g.n * g.oh * g.ow * c4;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int cq = (int)(i % c4);
    long long r = i / c4;
    const int ox = (int)(r % g.ow); r /= g.ow;
    const int oy = (int)(r % g.oh);
    const int n = (int)(r / g.oh);
    float4 best = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    int bi[4] = {0, 0, 0, 0};
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = 2 * oy - g.pt + ky;
      if (iy < 0 || iy >= g.ih) continue;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int ix = 2 * ox - g.pl + kx;
        if (ix < 0 || ix >= g.iw) continue;
        const float4 v = *reinterpret_cast<const float4*>(x + (((long long)n * g.ih + iy) * g.iw + ix) * g.c + 4 * cq);
        const int w = ky * 3 + kx;
        if (v.x > best.x) { best.x = v.x; bi[0] = w; }
        if (v.y > best.y) { best.y = v.y; bi[1] = w; }
        if (v.z > best.z) { best.z = v.z; bi[2] = w; }
        if (v.w > best.w) { best.w = v.w; bi[3] = w; }
      }
    }
    reinterpret_cast<float4*>(y)[i] = best;
    reinterpret_cast<uchar4*>(arg)[i] = make_uchar4((uint8_t)bi[0], (uint8_t)bi[1], (uint8_t)bi[2], (uint8_t)bi[3]);
  }
}

__global__ void __launch_bounds__(256)
maxpool_bwd_kernel(PoolGeom g, const float* __restrict__ dy, const uint8_t* __restrict__ arg, float* __restrict__ dx) {
  const int c4 = g.c >> 2;
  const long long total = (long long)g.n * g.ih * g.iw * c4;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int cq = (int)(i % c4);
    long long r = i / c4;
    const int ix = (int)(r % g.iw); r /= g.iw;
    const int iy = (int)(r % g.ih);
    const int n = (int)(r / g.ih);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    // windows oy with 2*oy - pt <= iy <= 2*oy - pt + 2
    const int y0 = iy + g.pt, x0 = ix + g.pl;
    for (int oy = (y0 - 1) >> 1; oy <= (y0 >> 1); ++oy) {       // (y0-2+1)/2 rounded up == (y0-1)>>1 for y0>=1
      if (oy < 0 || oy >= g.oh) continue;
      const int ky = y0 - 2 * oy;
      if (ky < 0 || ky > 2) continue;
      for (int ox = (x0 - 1) >> 1; ox <= (x0 >> 1); ++ox) {
        if (ox < 0 || ox >= g.ow) continue;
        const int kx = x0 - 2 * ox;
        if (kx < 0 || kx > 2) continue;
        const long long o = (((long long)n * g.oh + oy) * g.ow + ox) * c4 + cq;
        const uchar4 a = reinterpret_cast<const uchar4*>(arg)[o];
        const float4 d = reinterpret_cast<const float4*>(dy)[o];
        const int w = ky * 3 + kx;
        if (a.x == w) acc.x += d.x;
        if (a.y == w) acc.y += d.y;
        if (a.z == w) acc.z += d.z;
        if (a.w == w) acc.w += d.w;
      }
    }
    reinterpret_cast<float4*>(dx)[i] = acc;
  }
}

int make_geom(int n, int ih, int iw, int c, PoolGeom* g, const char* what) {
  SEEDHIP_REQUIRE(n >= 1 && ih >= 1 && iw >= 1 && c >= 4 && c % 4 == 0, "%s: need n,ih,iw >= 1 and c %% 4 == 0", what);
  g->n = n; g->ih = ih; g->iw = iw; g->c = c;
  g->oh = (ih + 1) / 2; g->ow = (iw + 1) / 2;
  const int ph = (g->oh - 1) * 2 + 3 - ih, pw = (g->ow - 1) * 2 + 3 - iw;
  g->pt = (ph > 0 ? ph : 0) / 2; g->pl = (pw > 0 ? pw : 0) / 2;
  return SEEDHIP_OK;
}
int grid_for(long long n) { long long b = (n + 255) / 256; return (int)(b > 8192 ? 8192 : (b < 1 ? 1 : b)); }

}  // namespace

extern "C" int seedhip_maxpool3x3s2_same_fwd(int n, int ih, int iw, int c, const float* x, float* y, uint8_t* argmax,
                                             void* stream) {
  PoolGeom g;
  int rc = make_geom(n, ih, iw, c, &g, "maxpool_fwd"); if (rc) return rc;
  SEEDHIP_REQUIRE(x && y && argmax, "maxpool_fwd: null pointer");
  hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(grid_for((long long)n * g.oh * g.ow * (c / 4))), dim3(256), 0,
                     (hipStream_t)stream, g, x, y, argmax);
  return seedhip::check_launch("maxpool_fwd_kernel");
}

extern "C" int seedhip_maxpool3x3s2_same_bwd(int n, int ih, int iw, int c, const float* dy, const uint8_t* argmax,
                                             float* dx, void* stream) {
  PoolGeom g;
  int rc = make_geom(n, ih, iw, c, &g, "maxpool_bwd"); if (rc) return rc;
  SEEDHIP_REQUIRE(dy && dx && argmax, "maxpool_bwd: null pointer");
  hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(grid_for((long long)n * ih * iw * (c / 4))), dim3(256), 0,
                     (hipStream_t)stream, g, dy, argmax, dx);
  return seedhip::check_launch("maxpool_bwd_kernel");
}

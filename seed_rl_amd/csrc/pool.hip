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
#include "igemm.h"
#include "../../include/seedhip.h"

namespace {

struct PoolGeom {
  int n, ih, iw, c, oh, ow, pt, pl;
  seedhip::FastDiv d_c4, d_ow, d_oh, d_iw, d_ih;         // 32-bit mul-hi decodes (the item counts fit 31 bits: make_geom)
};

// One thread owns 4 channels of one pooled pixel.  Interior windows (all 9 taps inside the map: all but the last row /
// column of an even map) take the straight-line path: nine independent 16-byte loads in flight, no border tests.
__global__ void __launch_bounds__(256)
maxpool_fwd_kernel(PoolGeom g, const float* __restrict__ x, float* __restrict__ y, uint8_t* __restrict__ arg,
                   uint8_t* __restrict__ bits) {
  const uint32_t c4 = (uint32_t)g.c >> 2;
  const uint32_t total = (uint32_t)g.n * g.oh * g.ow * c4;
  const uint32_t stride = gridDim.x * blockDim.x;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    uint32_t r, r2, cq, ox, oy, n;
    g.d_c4.divmod(i, r, cq);
    g.d_ow.divmod(r, r2, ox);
    g.d_oh.divmod(r2, n, oy);
    const float* img = x + (long long)n * g.ih * g.iw * g.c + 4 * cq;
    const int iy0 = 2 * (int)oy - g.pt, ix0 = 2 * (int)ox - g.pl;
    float4 best;
    int bi[4] = {0, 0, 0, 0};
    if (iy0 >= 0 && iy0 + 2 < g.ih && ix0 >= 0 && ix0 + 2 < g.iw) {
      const float* p0 = img + (iy0 * g.iw + ix0) * g.c;
      float4 v[9];
#pragma unroll
      for (int w = 0; w < 9; ++w) v[w] = *reinterpret_cast<const float4*>(p0 + ((w / 3) * g.iw + (w % 3)) * g.c);
      best = v[0];
#pragma unroll
      for (int w = 1; w < 9; ++w) {
        if (v[w].x > best.x) { best.x = v[w].x; bi[0] = w; }
        if (v[w].y > best.y) { best.y = v[w].y; bi[1] = w; }
        if (v[w].z > best.z) { best.z = v[w].z; bi[2] = w; }
        if (v[w].w > best.w) { best.w = v[w].w; bi[3] = w; }
      }
    } else {
      best = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const int iy = iy0 + ky;
        if (iy < 0 || iy >= g.ih) continue;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int ix = ix0 + kx;
          if (ix < 0 || ix >= g.iw) continue;
          const float4 v = *reinterpret_cast<const float4*>(img + (iy * g.iw + ix) * g.c);
          const int w = ky * 3 + kx;
          if (v.x > best.x) { best.x = v.x; bi[0] = w; }
          if (v.y > best.y) { best.y = v.y; bi[1] = w; }
          if (v.z > best.z) { best.z = v.z; bi[2] = w; }
          if (v.w > best.w) { best.w = v.w; bi[3] = w; }
        }
      }
    }
    reinterpret_cast<float4*>(y)[i] = best;
    reinterpret_cast<uchar4*>(arg)[i] = make_uchar4((uint8_t)bi[0], (uint8_t)bi[1], (uint8_t)bi[2], (uint8_t)bi[3]);
    // the ReLU mask of y as bytes: bit q of byte [pixel][quad] = y[pixel][4 quad + q] > 0 (wsx.h)
    if (bits) bits[i] = (uint8_t)((best.x > 0.f ? 1 : 0) | (best.y > 0.f ? 2 : 0) | (best.z > 0.f ? 4 : 0) | (best.w > 0.f ? 8 : 0));
  }
}

// One thread owns 4 channels of one INPUT pixel and looks at the <= 4 windows that cover it (1, 2 or 4 by the parity
// of its coordinates): the candidates' argmax bytes and gradients are requested together, then compared.
__global__ void __launch_bounds__(256)
maxpool_bwd_kernel(PoolGeom g, const float* __restrict__ dy, const uint8_t* __restrict__ arg, float* __restrict__ dx) {
  const uint32_t c4 = (uint32_t)g.c >> 2;
  const uint32_t total = (uint32_t)g.n * g.ih * g.iw * c4;
  const uint32_t stride = gridDim.x * blockDim.x;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    uint32_t r, r2, cq, ix, iy, n;
    g.d_c4.divmod(i, r, cq);
    g.d_iw.divmod(r, r2, ix);
    g.d_ih.divmod(r2, n, iy);
    // windows oy with 2*oy - pt <= iy <= 2*oy - pt + 2: oy in {(y0-1)>>1, y0>>1} (one window when y0 is odd)
    const int y0 = (int)iy + g.pt, x0 = (int)ix + g.pl;
    const int oya = (y0 - 1) >> 1, oyb = y0 >> 1, oxa = (x0 - 1) >> 1, oxb = x0 >> 1;
    const int oys[2] = {oya, oyb}, oxs[2] = {oxa, oxb};
    const bool yok[2] = {oya >= 0 && oya < g.oh, oyb != oya && oyb < g.oh};
    const bool xok[2] = {oxa >= 0 && oxa < g.ow, oxb != oxa && oxb < g.ow};
    const uint32_t obase = n * (uint32_t)(g.oh * g.ow) * c4 + cq;
    uint32_t av[4];
    float4 dv[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int a = t >> 1, b = t & 1;
      const bool ok = yok[a] && xok[b];
      const uint32_t o = ok ? obase + (uint32_t)(oys[a] * g.ow + oxs[b]) * c4 : obase;    // masked: any mapped element
      av[t] = reinterpret_cast<const uint32_t*>(arg)[o];
      dv[t] = reinterpret_cast<const float4*>(dy)[o];
      if (!ok) av[t] = 0xFFFFFFFFu;                        // code 255 matches no tap
    }
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int a = t >> 1, b = t & 1;
      const uint32_t w = (uint32_t)((y0 - 2 * oys[a]) * 3 + (x0 - 2 * oxs[b]));
      if ((av[t] & 255u) == w) acc.x += dv[t].x;
      if (((av[t] >> 8) & 255u) == w) acc.y += dv[t].y;
      if (((av[t] >> 16) & 255u) == w) acc.z += dv[t].z;
      if ((av[t] >> 24) == w) acc.w += dv[t].w;
    }
    reinterpret_cast<float4*>(dx)[i] = acc;
  }
}

// Even maps (pt = pl = 0, iw even: the ImpalaDeep stacks): one thread owns 4 channels of TWO horizontally adjacent input
// pixels (2m, 2m + 1).  Their windows share the pooled columns {m - 1, m} (the odd pixel lies in column m only), so a
// pair costs 4 (even rows) or 2 (odd rows: one pooled row) window requests instead of the 8 the kernel above issues
// for two pixels -- 1.5 per pixel instead of 4; its terms are added in that kernel's order (bit-identical results).
__global__ void __launch_bounds__(256)
maxpool_bwd_pair_kernel(PoolGeom g, seedhip::FastDiv d_hw, const float* __restrict__ dy, const uint8_t* __restrict__ arg,
                        float* __restrict__ dx) {
  const uint32_t c4 = (uint32_t)g.c >> 2, hw = (uint32_t)g.iw >> 1;
  const uint32_t total = (uint32_t)g.n * g.ih * hw * c4;
  const uint32_t stride = gridDim.x * blockDim.x;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    uint32_t r, r2, cq, m, iy, n;
    g.d_c4.divmod(i, r, cq);
    d_hw.divmod(r, r2, m);
    g.d_ih.divmod(r2, n, iy);
    const int k = (int)iy >> 1;
    const bool yodd = iy & 1u;
    // pooled rows: odd iy -> k (tap row 1); even iy -> k - 1 (tap row 2, if it exists) then k (tap row 0)
    const int oys[2] = {yodd ? k : k - 1, k};
    const bool yok[2] = {yodd || k >= 1, !yodd};
    const int kys[2] = {yodd ? 1 : 2, 0};
    const bool xa = m >= 1;                                // pooled column m - 1 exists
    const uint32_t obase = n * (uint32_t)(g.oh * g.ow) * c4 + cq;
    uint32_t av[2][2];
    float4 dv[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        av[a][b] = 0xFFFFFFFFu;                            // code 255 matches no tap
        dv[a][b] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (yok[a] && (b == 1 || xa)) {
          const uint32_t o = obase + (uint32_t)(oys[a] * g.ow + (int)m - 1 + b) * c4;
          av[a][b] = reinterpret_cast<const uint32_t*>(arg)[o];
          dv[a][b] = reinterpret_cast<const float4*>(dy)[o];
        }
      }
    float4 e = make_float4(0.f, 0.f, 0.f, 0.f), o4 = e;    // even pixel 2m: columns m - 1 (tap column 2), m (0); odd: m (1)
#pragma unroll
    for (int a = 0; a < 2; ++a) {
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const uint32_t w = (uint32_t)(kys[a] * 3 + (b == 0 ? 2 : 0));
        if ((av[a][b] & 255u) == w) e.x += dv[a][b].x;
        if (((av[a][b] >> 8) & 255u) == w) e.y += dv[a][b].y;
        if (((av[a][b] >> 16) & 255u) == w) e.z += dv[a][b].z;
        if ((av[a][b] >> 24) == w) e.w += dv[a][b].w;
      }
      const uint32_t w1 = (uint32_t)(kys[a] * 3 + 1);
      if ((av[a][1] & 255u) == w1) o4.x += dv[a][1].x;
      if (((av[a][1] >> 8) & 255u) == w1) o4.y += dv[a][1].y;
      if (((av[a][1] >> 16) & 255u) == w1) o4.z += dv[a][1].z;
      if ((av[a][1] >> 24) == w1) o4.w += dv[a][1].w;
    }
    float4* dst = reinterpret_cast<float4*>(dx) + ((n * (uint32_t)g.ih + iy) * (uint32_t)g.iw + 2u * m) * c4 + cq;
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    __builtin_nontemporal_store(f32x4{e.x, e.y, e.z, e.w}, reinterpret_cast<f32x4*>(dst));          // written once, read by
    __builtin_nontemporal_store(f32x4{o4.x, o4.y, o4.z, o4.w}, reinterpret_cast<f32x4*>(dst + c4)); // later kernels only
  }
}

int make_geom(int n, int ih, int iw, int c, PoolGeom* g, const char* what) {
  SEEDHIP_REQUIRE(n >= 1 && ih >= 1 && iw >= 1 && c >= 4 && c % 4 == 0, "%s: need n,ih,iw >= 1 and c %% 4 == 0", what);
  g->n = n; g->ih = ih; g->iw = iw; g->c = c;
  g->oh = (ih + 1) / 2; g->ow = (iw + 1) / 2;
  const int ph = (g->oh - 1) * 2 + 3 - ih, pw = (g->ow - 1) * 2 + 3 - iw;
  g->pt = (ph > 0 ? ph : 0) / 2; g->pl = (pw > 0 ? pw : 0) / 2;
  SEEDHIP_REQUIRE((long long)n * ih * iw * c < (1LL << 31), "%s: tensor of 2^31 elements or more", what);
  g->d_c4.init(c / 4); g->d_ow.init(g->ow); g->d_oh.init(g->oh); g->d_iw.init(iw); g->d_ih.init(ih);
  return SEEDHIP_OK;
}
// (one item per thread up to 2^18 workgroups: a streaming fill of 1.2 GB runs at 6.1 TB/s from 65 536 workgroups against 4.5
// from 8 192 looping ones -- tools/probes/write_bw_probe.hip)
int grid_for(long long n) { long long b = (n + 255) / 256; return (int)(b > (1 << 18) ? (1 << 18) : (b < 1 ? 1 : b)); }

}  // namespace

extern "C" int seedhip_maxpool3x3s2_same_fwd(int n, int ih, int iw, int c, const float* x, float* y, uint8_t* argmax,
                                             void* stream) {
  return seedhip_maxpool3x3s2_same_fwd_bits(n, ih, iw, c, x, y, argmax, nullptr, stream);
}

// The same, also writing the sign of y as bytes [pixel][c / 4] (y_bits may be NULL)
extern "C" int seedhip_maxpool3x3s2_same_fwd_bits(int n, int ih, int iw, int c, const float* x, float* y, uint8_t* argmax,
                                                  uint8_t* y_bits, void* stream) {
  PoolGeom g;
  int rc = make_geom(n, ih, iw, c, &g, "maxpool_fwd"); if (rc) return rc;
  SEEDHIP_REQUIRE(x && y && argmax, "maxpool_fwd: null pointer");
  hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(grid_for((long long)n * g.oh * g.ow * (c / 4))), dim3(256), 0,
                     (hipStream_t)stream, g, x, y, argmax, y_bits);
  return seedhip::check_launch("maxpool_fwd_kernel");
}

extern "C" int seedhip_maxpool3x3s2_same_bwd(int n, int ih, int iw, int c, const float* dy, const uint8_t* argmax,
                                             float* dx, void* stream) {
  PoolGeom g;
  int rc = make_geom(n, ih, iw, c, &g, "maxpool_bwd"); if (rc) return rc;
  SEEDHIP_REQUIRE(dy && dx && argmax, "maxpool_bwd: null pointer");
  if (g.pt == 0 && g.pl == 0 && iw % 2 == 0) {
    seedhip::FastDiv d_hw; d_hw.init(iw / 2);
    hipLaunchKernelGGL(maxpool_bwd_pair_kernel, dim3(grid_for((long long)n * ih * (iw / 2) * (c / 4))), dim3(256), 0,
                       (hipStream_t)stream, g, d_hw, dy, argmax, dx);
    return seedhip::check_launch("maxpool_bwd_pair_kernel");
  }
  hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(grid_for((long long)n * ih * iw * (c / 4))), dim3(256), 0,
                     (hipStream_t)stream, g, dy, argmax, dx);
  return seedhip::check_launch("maxpool_bwd_kernel");
}

// Host-side element-level executor for the implicit-GEMM problem accessors
// (seed_rl_amd/csrc/conv_problems.h).  TEST INFRASTRUCTURE: compiled with g++ and
// driven from tests/test_host_emul.py; it runs the SAME accessor code the GPU
// kernel runs (index decode, padding, parity classes, split-K, permutations) with
// plain loops instead of MFMA tiles, so layout bugs are caught without a GPU.
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../../seed_rl_amd/csrc/conv_problems.h"
#include "../../seed_rl_amd/csrc/gemm_geom.h"
#include "../../seed_rl_amd/csrc/wsgemm_geom.h"

using namespace seedhip;

template <class P>
static void run_problem(const P& p, int slices) {
  for (int z = 0; z < slices; ++z) {
    int k0, k1;
    p.k_range(z, k0, k1);
    std::vector<double> colsum(p.N, 0.0);
    for (int m = 0; m < p.M; ++m) {
      for (int n = 0; n < p.N; ++n) {
        double acc = 0.0;
        for (int k = k0; k < k1; ++k) {
          float a, b;
          if (P::kAVecK) a = f4_get(p.load_a(p.a_row(m, z), k0 + ((k - k0) & ~3), z), (k - k0) & 3);
          else a = f4_get(p.load_a(p.a_row(m & ~3, z), k, z), m & 3);
          if (P::kBVecN) b = f4_get(p.load_b(p.b_col(n & ~3, z), k, z), n & 3);
          else b = f4_get(p.load_b(p.b_col(n, z), k0 + ((k - k0) & ~3), z), (k - k0) & 3);
          acc += (double)a * (double)b;
          if (P::kColSumB && m == 0) colsum[n] += b;
        }
        p.store(m, n, (float)acc, z);
      }
    }
    if (P::kColSumB) for (int n = 0; n < p.N; ++n) p.store_colsum(n, (float)colsum[n], z);
  }
}

static void reduce_slices(const float* partial, int slices, long long n, float* out) {
  for (long long i = 0; i < n; ++i) {
    double a = 0; for (int z = 0; z < slices; ++z) a += partial[(long long)z * n + i];
    out[i] = (float)a;
  }
}

extern "C" {

struct geom_c { int n_img, ih, iw, cin, oh, ow, kh, kw, stride, pad_t, pad_l, cout, ld_in, ld_out; };
struct sgeom_c { int T, B, ih, iw, oh, ow, kh, kw, stride, cout, ld_out; };

static ConvGeom cg(const geom_c* g) {
  ConvGeom c; memcpy(&c, g, sizeof(c)); return c;
}
static StackGeom sg(const sgeom_c* g) { StackGeom s; memcpy(&s, g, sizeof(s)); return s; }

int emul_fastdiv_check(uint32_t d, uint32_t xmax, uint32_t step) {
  FastDiv f; f.init(d);
  for (uint64_t x = 0; x <= xmax; x += step) {
    uint32_t q, r; f.divmod((uint32_t)x, q, r);
    if (q != (uint32_t)x / d || r != (uint32_t)x % d) return 0;
  }
  return 1;
}

void emul_conv_fwd(const geom_c* g, const void* in, int in_dtype, int in_relu, const float* w, const float* bias,
                   float* out, int out_relu, const float* residual) {
  ConvFwd p; p.in = in; p.in_dtype = in_dtype; p.in_relu = in_relu; p.w = w; p.bias = bias; p.out = out;
  p.out_relu = out_relu; p.residual = residual; p.init(cg(g));
  run_problem(p, 1);
}
void emul_conv_dgrad(const geom_c* g, const float* dy, const float* w, float* dx, const float* mask, const float* add) {
  ConvDgrad p; p.dy = dy; p.w = w; p.dx = dx; p.mask = mask; p.add = add; p.init(cg(g));
  run_problem(p, p.slices());
}
void emul_conv_wgrad(const geom_c* g, const void* in, int in_dtype, int in_relu, const float* dy, float* dw,
                     float* dbias, int k_per_slice) {
  ConvWgrad p; p.in = in; p.in_dtype = in_dtype; p.in_relu = in_relu; p.dy = dy; p.init(cg(g), k_per_slice);
  const int s = p.slices();
  std::vector<float> pw((size_t)s * p.M * p.N), pb((size_t)s * p.N);
  p.partial_w = pw.data(); p.partial_b = pb.data();
  run_problem(p, s);
  reduce_slices(pw.data(), s, (long long)p.M * p.N, dw);
  reduce_slices(pb.data(), s, p.N, dbias);
}
void emul_stack_fwd(const sgeom_c* g, const uint8_t* frames_ext, const uint8_t* nvalid, const float* w,
                    const float* bias, float* out, int out_relu) {
  ConvStackFwd p; p.frames_ext = frames_ext; p.nvalid = nvalid; p.w = w; p.bias = bias; p.out = out;
  p.out_relu = out_relu; p.init(sg(g));
  run_problem(p, 1);
}
void emul_stack_wgrad(const sgeom_c* g, const uint8_t* frames_ext, const uint8_t* nvalid, const float* dy, float* dw,
                      float* dbias, int k_per_slice) {
  ConvStackWgrad p; p.frames_ext = frames_ext; p.nvalid = nvalid; p.dy = dy; p.init(sg(g), k_per_slice);
  const int s = p.slices();
  std::vector<float> pw((size_t)s * p.M * p.N), pb((size_t)s * p.N);
  p.partial_w = pw.data(); p.partial_b = pb.data();
  run_problem(p, s);
  reduce_slices(pw.data(), s, (long long)p.M * p.N, dw);
  reduce_slices(pb.data(), s, p.N, dbias);
}


// ---- gather-GEMM convolutions (gemm.h): the SAME row / tap / border / scatter arithmetic the GPU stagers call ---- //
static seedhip_conv_geom to_abi(const geom_c* g) {
  seedhip_conv_geom a;
  a.n_img = g->n_img; a.ih = g->ih; a.iw = g->iw; a.cin = g->cin; a.oh = g->oh; a.ow = g->ow; a.kh = g->kh; a.kw = g->kw;
  a.stride = g->stride; a.pad_t = g->pad_t; a.pad_l = g->pad_l; a.cout = g->cout; a.ld_in = g->ld_in; a.ld_out = g->ld_out;
  return a;
}
// element (x, k) of a gathered operand, zero outside
static float gathered(const float* base, const gemm::Gather& g, int x, int k, int relu) {
  long long off; int y0, x0, toff, dy, dx; bool ok;
  gemm::gather_row(g, x, off, y0, x0);
  gemm::gather_tap(g, k, toff, dy, dx, ok);
  if (!ok || !gemm::gather_inside(g, y0 + dy, x0 + dx)) return 0.f;
  const float v = base[off + toff];
  return (relu && v < 0.f) ? 0.f : v;
}
int emul_gather_fwd(const geom_c* g, const float* in, int in_relu, const float* w, const float* bias, float* out,
                    int out_relu) {
  const seedhip_conv_geom a = to_abi(g);
  gemm::Params p;
  if (!gemm::conv_fwd_setup(p, &a)) return 0;
  for (int m = 0; m < p.M; ++m)
    for (int n = 0; n < p.N; ++n) {
      double acc = 0.0;
      for (int k = 0; k < p.K; ++k) acc += (double)gathered(in, p.ga, m, k, in_relu) * w[(long long)k * p.ldb + n];
      float v = (float)acc + (bias ? bias[n] : 0.f);
      if (out_relu && v < 0.f) v = 0.f;
      out[(long long)m * p.ldc + n] = v;
    }
  return 1;
}
int emul_gather_dgrad(const geom_c* g, const float* dy, const float* w, float* dx, const float* mask, const float* add) {
  const seedhip_conv_geom a = to_abi(g);
  gemm::Params p;
  if (!gemm::conv_dgrad_setup(p, &a)) return 0;
  for (int m = 0; m < p.M; ++m)
    for (int n = 0; n < p.N; ++n) {
      long long at;
      if (!gemm::scatter_addr(p.ga, p.gb, p.es, p.eih, p.eiw, p.ldc, m, n, at)) continue;
      double acc = 0.0;
      for (int k = 0; k < p.K; ++k) acc += (double)gathered(dy, p.ga, m, k, 0) * gathered(w, p.gb, n, k, 0);
      float v = (float)acc;
      if (mask && !(mask[at] > 0.f)) v = 0.f;
      if (add) v += add[at];
      dx[at] = v;
    }
  return 1;
}
int emul_gather_wgrad(const geom_c* g, const float* in, int in_relu, const float* dy, float* dw, float* dbias) {
  const seedhip_conv_geom a = to_abi(g);
  gemm::Params p;
  if (!gemm::conv_wgrad_setup(p, &a)) return 0;
  // the transposed im2col operand: reduction index = output pixel (the Gather's row), GEMM row = (tap, channel)
  for (int x = 0; x < p.M; ++x)
    for (int n = 0; n < p.N; ++n) {
      double acc = 0.0;
      for (int pix = 0; pix < p.K; ++pix) acc += (double)gathered(in, p.ga, pix, x, in_relu) * dy[(long long)pix * p.ldb + n];
      dw[(long long)x * p.N + n] = (float)acc;
    }
  // the stepped 32-bit form of the same index math (what the GPU stager executes): every pixel reached by stepping
  // from a decoded start must give the element gather_row / gather_tap / gather_inside give
  const gemm::Gather& ga = p.ga;
  if (!ga.coord_uv && ga.extent > 0 && ga.extent < (1LL << 29)) {
    const uint32_t rows_v = ga.d2.div(ga.d1.d);
    for (uint32_t step : {1u, 4u, 8u}) {
      if (step > ga.d2.d) continue;
      for (uint32_t start = 0; start < step && (int)start < p.K; ++start) {
        uint32_t u, v, w;
        gemm::gather_decode(ga, start, u, v, w);
        for (uint32_t pix = start; (int)pix < p.K; pix += step) {
          if (pix != start) gemm::gather_step(ga, rows_v, step, u, v, w);
          for (int x = 0; x < p.M; x += 4) {
            int toff, tdy, tdx; bool tap_ok;
            gemm::gather_tap(ga, x, toff, tdy, tdx, tap_ok);
            unsigned byte_off;
            const bool inside = gemm::gather_elem32(ga, u, v, w, toff, tdy, tdx, byte_off) && tap_ok;
            long long off; int y0, x0;
            gemm::gather_row(ga, (int)pix, off, y0, x0);
            const bool ref_inside = tap_ok && gemm::gather_inside(ga, y0 + tdy, x0 + tdx);
            if (inside != ref_inside) return 0;
            if (inside && (long long)byte_off != (off + toff) * 4) return 0;
          }
        }
      }
    }
  }
  for (int n = 0; n < p.N; ++n) {
    double sacc = 0.0;
    for (int pix = 0; pix < p.K; ++pix) sacc += dy[(long long)pix * p.ldb + n];
    dbias[n] = (float)sacc;
  }
  return 1;
}

}  // extern "C"

// ---- weight-stationary conv kernels (wsgemm.h ws_fast_kernel): the tile walk of the GPU kernel with the SAME helpers
// (wsgemm_geom.h): row decode by one division + stepping, A byte offsets and tap validity (out-of-range = zero),
// W' re-indexing, dX scatter offsets.  Returns 0 when the shape is outside the kernel's range. ----
extern "C" int emul_ws_fwd(const geom_c* g, const float* in, const float* w, const float* bias, float* out, int out_relu) {
  const seedhip_conv_geom a = to_abi(g);
  wsgemm::Params p;
  wsgemm::Plan pl = wsgemm::plan_fwd(p, &a);
  if (!pl.ok || p.gw < 8) return 0;
  const int N = p.N, ntiles = (p.M + 15) / 16;
  for (int tile = 0; tile < ntiles; ++tile)
    for (int srow = 0; srow < 8; ++srow) {                 // the two staging rows of a lane: srow and srow + 8 (stepped)
      uint32_t img, ya, xb;
      wsgemm::ws_locate(p, (uint32_t)tile * 16u + srow, img, ya, xb);
      for (int i = 0; i < 2; ++i) {
        if (i) wsgemm::ws_advance(p, 8u, img, ya, xb);
        const uint32_t m = (uint32_t)tile * 16u + srow + 8u * i;
        if (m >= (uint32_t)p.M) continue;
        for (int n = 0; n < N; ++n) {
          double acc = 0.0;
          for (int t = 0; t < p.nkt; ++t)
            for (int kc = 0; kc < wsgemm::BK; ++kc) {
              const unsigned byte = wsgemm::ws_row_byte(p, img, ya, xb, kc) + 4u * (unsigned)p.tile_off[t];
              if ((long long)byte + 4 > p.a_bytes) return 0;                       // a 'valid' conv never leaves the map
              acc += (double)in[byte / 4] * w[(long long)(t * wsgemm::BK + kc) * N + n];
            }
          float v = (float)acc + (bias ? bias[n] : 0.f);
          if (out_relu && v < 0.f) v = 0.f;
          out[(long long)m * p.ldc + n] = v;
        }
      }
    }
  return 1;
}
extern "C" int emul_ws_dgrad(const geom_c* g, const float* dy, const float* w, float* dx, const float* mask, const float* add) {
  const seedhip_conv_geom a = to_abi(g);
  wsgemm::Params p;
  wsgemm::Plan pl = wsgemm::plan_dgrad(p, &a);
  if (!pl.ok || p.gw < 8) return 0;
  const int N = p.N, ntiles = (p.M + 15) / 16;
  const bool exact = p.gh * p.s == p.ih && p.gw * p.s == p.iw;
  int minoff = 0;
  for (int t = 0; t < p.nkt; ++t) minoff = p.tile_off[t] < minoff ? p.tile_off[t] : minoff;
  std::vector<float> wp((size_t)p.K * N);
  for (int k = 0; k < p.K; ++k) for (int n = 0; n < N; ++n) wp[(size_t)k * N + n] = w[wsgemm::ws_wprime_src(p, k, n)];
  for (int tile = 0; tile < ntiles; ++tile)
    for (int kq = 0; kq < 4; ++kq) {                       // epilogue rows of a lane group: 4 kq + r, stepped from r = 0
      uint32_t img, ya, xb;
      wsgemm::ws_locate(p, (uint32_t)tile * 16u + 4u * kq, img, ya, xb);
      for (int r = 0; r < 4; ++r) {
        if (r) wsgemm::ws_advance(p, 1u, img, ya, xb);
        const uint32_t m = (uint32_t)tile * 16u + 4u * kq + r;
        if (m >= (uint32_t)p.M) continue;
        // the same row through the load cursor's decode (division at the tile's first staging row + step of 8)
        { uint32_t i2, a2, b2; const uint32_t lr = (4u * kq + r) & 7u, hi = (4u * kq + r) >> 3;
          wsgemm::ws_locate(p, (uint32_t)tile * 16u + lr, i2, a2, b2);
          if (hi) wsgemm::ws_advance(p, 8u, i2, a2, b2);
          if (i2 != img || a2 != ya || b2 != xb) return 0; }
        for (int n = 0; n < N; ++n) {
          double acc = 0.0;
          for (int t = 0; t < p.nkt; ++t) {
            if (!wsgemm::ws_tap_ok(p, ya, xb, p.tile_dy[t], p.tile_dx[t])) continue;          // hardware zero fill
            for (int kc = 0; kc < wsgemm::BK; ++kc) {
              // the kernel's view starts at A + minoff: byte offset relative to it, k-tile offset rebased
              const long long rel = (long long)wsgemm::ws_row_byte(p, img, ya, xb, kc) + 4LL * (p.tile_off[t] - minoff);
              const long long abs_f = rel / 4 + minoff;
              if (abs_f < 0 || abs_f * 4 + 4 > p.a_bytes) return 0;                          // a valid tap is inside dY
              acc += (double)dy[abs_f] * wp[(size_t)(t * wsgemm::BK + kc) * N + n];
            }
          }
          int py, px;
          const unsigned e_const = wsgemm::ws_dgrad_col(p, n, py, px);
          const unsigned at = wsgemm::ws_dgrad_at(p, img, ya, xb, e_const, py, px, exact);
          if (at == 0xffffffffu) continue;
          float v = (float)acc;
          if (mask && !(mask[at] > 0.f)) v = 0.f;
          if (add) v += add[at];
          dx[at] = v;
        }
      }
    }
  return 1;
}

// C-ABI entry points for Conv2D / Dense forward, data-gradient and weight-gradient (fp32 MFMA).
//
// Dispatch per layer shape (all paths produce the same results up to fp32 summation order):
//   * Dense (1x1 on a 1x1 image)            -> Dense{Fwd,Dgrad,Wgrad} accessors on the implicit-GEMM core
//   * small kernels (<= 4x4, stride <= 2)   -> halo kernels: input band staged once in LDS
//       forward  : halo_fwd.h   (stride-1 layers where it beats the core; see halo_wins below)
//       data grad: halo_fwd.h   (all stride-parity classes in one launch)
//       weight grad: halo_wgrad.h
//   * everything else                        -> Conv{Fwd,Dgrad,Wgrad} on the implicit-GEMM core
//     (igemm.h + conv_problems.h; accessor index math unit-tested on the CPU, tests/host/emul.cpp)
// The first Atari conv fused with frame stacking lives in stackconv.hip.
#include "common.h"
#include "conv_problems.h"
#include "conv_launch.h"
#include "halo_wgrad.h"
#include "halo_fwd.h"
#include "gemm.h"
#include "xgemm.h"
#include "xgemm8.h"
#include "wsgemm.h"
#include "wfx.h"
#include "wdx.h"
#include "wsx.h"
#include "wsy.h"
#include "wgx_api.h"
#include "fgx_api.h"
#include "cgx_api.h"
#include "../../include/seedhip.h"

using namespace seedhip;

namespace {

ConvGeom to_geom(const seedhip_conv_geom* g) {
  ConvGeom c;
  c.n_img = g->n_img; c.ih = g->ih; c.iw = g->iw; c.cin = g->cin; c.oh = g->oh; c.ow = g->ow;
  c.kh = g->kh; c.kw = g->kw; c.stride = g->stride; c.pad_t = g->pad_t; c.pad_l = g->pad_l; c.cout = g->cout;
  c.ld_in = g->ld_in; c.ld_out = g->ld_out;
  return c;
}

// Split-K epilogue of the dense layers: out[m, n] = act(sum_z partial[z][m, n] + bias[n] + residual) (forward) or
// dx = mask(sum_z partial) + add (data gradient).  Slices summed in order: deterministic.
__global__ void __launch_bounds__(256)
dense_epilogue_kernel(const float* __restrict__ partial, int slices, int M, int N, const float* __restrict__ bias,
                      const float* __restrict__ residual, int out_relu, const float* __restrict__ mask,
                      const float* __restrict__ add, float* __restrict__ out, int ld) {
  const long long total = (long long)M * N;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int m = (int)(i / N), n = (int)(i - (long long)m * N);
    float v = partial[i];
    for (int z = 1; z < slices; ++z) v += partial[(long long)z * total + i];
    const long long o = (long long)m * ld + n;
    if (bias) v += bias[n];
    if (residual) v += residual[o];
    if (out_relu && v < 0.f) v = 0.f;
    if (mask && !(mask[o] > 0.f)) v = 0.f;
    if (add) v += add[o];
    out[o] = v;
  }
}

// The same, four columns per thread (N % 4 == 0, 16-byte aligned rows): the x6 / x8 paths' second pass.
__global__ void __launch_bounds__(256)
dense_epilogue4_kernel(const float* __restrict__ partial, int slices, int M, int N, const float* __restrict__ bias,
                       const float* __restrict__ residual, int out_relu, const float* __restrict__ mask,
                       const float* __restrict__ add, float* __restrict__ out, int ld) {
  const int nq = N >> 2;
  const long long total4 = (long long)M * nq, total = (long long)M * N;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += stride) {
    const int m = (int)(i / nq), n = 4 * (int)(i - (long long)m * nq);
    const float* src = partial + (long long)m * N + n;
    float4 v = *reinterpret_cast<const float4*>(src);
    for (int z = 1; z < slices; ++z) {
      const float4 t = *reinterpret_cast<const float4*>(src + (long long)z * total);
      v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
    }
    const long long o = (long long)m * ld + n;
    if (bias) { const float4 t = *reinterpret_cast<const float4*>(bias + n); v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w; }
    if (residual) { const float4 t = *reinterpret_cast<const float4*>(residual + o); v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w; }
    if (out_relu) { if (v.x < 0.f) v.x = 0.f; if (v.y < 0.f) v.y = 0.f; if (v.z < 0.f) v.z = 0.f; if (v.w < 0.f) v.w = 0.f; }
    if (mask) {
      const float4 t = *reinterpret_cast<const float4*>(mask + o);
      if (!(t.x > 0.f)) v.x = 0.f; if (!(t.y > 0.f)) v.y = 0.f; if (!(t.z > 0.f)) v.z = 0.f; if (!(t.w > 0.f)) v.w = 0.f;
    }
    if (add) { const float4 t = *reinterpret_cast<const float4*>(add + o); v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w; }
    *reinterpret_cast<float4*>(out + o) = v;
  }
}

// Slices for a dense GEMM [M x N x K]: enough workgroups (64x64 tiles) to give every CU a couple, at least
// 8 k-tiles of 16 per slice.  1 = no split.
int dense_slices(int M, int N, int K) {
  const long long tiles = (long long)((M + 63) / 64) * ((N + 63) / 64);
  if (tiles >= 256) return 1;
  long long s = (512 + tiles - 1) / tiles;
  const long long max_by_k = K / 128 > 0 ? K / 128 : 1;
  if (s > max_by_k) s = max_by_k;
  if (s > 32) s = 32;
  return (int)(s < 1 ? 1 : s);
}
int slice_k(int K, int slices) { int per = (K + slices - 1) / slices; return (per + 15) / 16 * 16; }

// Dense layers on gemm.h (see there).  mode bits (SEEDHIP_GEMM, default 255): 1 forward, 2 data gradient, 4 weight
// gradient (8 / 16: wsgemm.h conv forward / data gradient; 32 / 64 / 128: gather-GEMM conv forward / data gradient / weight gradient); a cleared bit falls back to the Dense accessors of the implicit-GEMM core (A/B measurements).
constexpr int conv_min_n() { return 64; }
constexpr int gemm_mode() { return 255; }    // (r1-r3: bit mask of the GEMM-core dispatch for A/B runs; every bit is on)
bool al16(const void* q) { return (((uintptr_t)q) & 15) == 0; }
bool gemm_fwd_ok(const seedhip_conv_geom* g) {
  // rows may carry up to 3 pad columns (ld_in >= cin rounded up to 4; pads finite): the last k-vector of a row is
  // read whole and meets zero-filled B rows
  return (gemm_mode() & 1) && g->ld_in % 4 == 0 && g->ld_in >= (g->cin + 3) / 4 * 4 && g->cout % 4 == 0;
}
// 32 -> 32 3x3 layers on maps of >= 400 pixels (ImpalaDeep's five @18x24): bit 0 = forward through the halo kernel,
// bit 1 = data gradient through the halo kernel, instead of the gather-GEMM.  Measured in the cfg3 step (HIP graph, ms
// per step): r02d before the second tiling budget of halo_fwd.h: 0 -> 25.0, 1 -> 24.3, 2 -> 25.4, 3 -> 24.7; with it:
// 1 -> 21.25, 3 -> 21.17.  Stand-alone the gather-GEMM forward used to be the faster one (0.52 vs 0.56 ms) -- with the
// ReLU'd input and the residual add of the real layers it was not; the halo forward now takes 0.47 ms.
constexpr int halo_all() { return 3; }
bool gemm_dgrad_ok(const seedhip_conv_geom* g) { return (gemm_mode() & 2) && g->cout % 4 == 0 && g->ld_out % 4 == 0; }
bool gemm_wgrad_ok(const seedhip_conv_geom* g) {
  return (gemm_mode() & 4) && g->ld_in % 4 == 0 && g->ld_in >= (g->cin + 3) / 4 * 4 && g->cout % 4 == 0 && g->ld_out % 4 == 0;
}
size_t gemm_partial_bytes(int M, int N, int K) {
  const gemm::Plan pl = gemm::plan(M, N, K);
  return pl.slices > 1 ? (size_t)pl.slices * M * N * sizeof(float) : 0;
}

bool is_dense(const seedhip_conv_geom* g);
// convs with >= 64 output channels: weight gradient as a gather-GEMM over output pixels (gemm.h, bit 128)
bool conv_wgrad_gemm_ok(const seedhip_conv_geom* g) {
  gemm::Params tmp;
  // >= 64 output channels always; 32 channels only for unpadded layers (the second Atari conv: 0.237 -> 0.211 ms on
  // 128x32 tiles with ~500 pixel slices; the padded 3x3 ImpalaDeep layers are 1.8x slower here than on halo_wgrad.h)
  const bool wide = g->cout >= conv_min_n() || (g->cout == 32 && g->pad_t == 0 && g->pad_l == 0 && g->cin % 16 == 0);
  return (gemm_mode() & 128) && !is_dense(g) && wide && gemm::conv_wgrad_setup(tmp, g);
}
gemm::Plan conv_wgrad_plan(const seedhip_conv_geom* g) {
  const int M = g->kh * g->kw * g->cin, N = g->cout;
  const long long pixels = (long long)g->n_img * g->oh * g->ow;
  gemm::Plan pl = gemm::plan(M, N, (int)(pixels < (1LL << 30) ? pixels : (1LL << 30)), 0.8);
  if (pl.slices < 2) {                                    // always through the partial buffer: >= 2 slices
    pl.slices = 2; pl.k_per_slice = (int)(((pixels + 1) / 2 + gemm::BK - 1) / gemm::BK * gemm::BK);
  }
  return pl;
}

bool is_dense(const seedhip_conv_geom* g) {
  return g->kh == 1 && g->kw == 1 && g->ih == 1 && g->iw == 1 && g->oh == 1 && g->ow == 1 && g->stride == 1 &&
         g->pad_t == 0 && g->pad_l == 0;
}

// Dense layers on the bf16 pipe through the exact three-way split (xgemm.h): the plans (ok = served) as functions of
// the geometry alone, so that the workspace queries and the launches agree.  Row thresholds (r4, inference batches
// through FusedInferenceState): at 256 / 1024 rows the fp32-MFMA kernels of gemm.h are FASTER (92 / 163 us per inference
// call against 104 / 164 with xgemm.h and 111 / 172 with xgemm8.h, whose pre-split pass and 128-row tiles need a
// training-sized batch to pay)
constexpr int kX6MinRows = 2048, kX8MinRows = 4096;
xg::Plan x6_fwd_plan(const seedhip_conv_geom* g) {
  if (!is_dense(g) || g->n_img < kX6MinRows || g->ld_in % 4 || g->cin % 4 || g->ld_out % 4) return xg::Plan{false, 1, 0, {0, 0, 0}};
  return xg::plan(g->n_img, g->cout, g->cin, (long long)g->n_img * g->ld_in * 4, (long long)g->cin * g->cout * 4);
}
xg::Plan x6_dgrad_plan(const seedhip_conv_geom* g) {
  if (!is_dense(g) || g->n_img < kX6MinRows || g->ld_out % 4 || g->cout % 4 || g->ld_in % 4 || g->cin % 4) return xg::Plan{false, 1, 0, {0, 0, 0}};
  return xg::plan(g->n_img, g->cin, g->cout, (long long)g->n_img * g->ld_out * 4, (long long)g->cin * g->cout * 4);
}
xg::Plan x6_wgrad_plan(const seedhip_conv_geom* g) {
  if (!is_dense(g) || g->n_img < kX6MinRows || g->ld_in % 4 || g->cin % 4 || g->ld_out % 4 || g->cout % 4) return xg::Plan{false, 1, 0, {0, 0, 0}};
  return xg::plan(g->cin, g->cout, g->n_img, (long long)g->n_img * g->ld_in * 4, (long long)g->n_img * g->ld_out * 4);
}

// The 8-wave structure with the small operand pre-split (xgemm8.h); workspace = [partial sums][slabs]
xg8::Plan x8_fwd_plan(const seedhip_conv_geom* g) {
  if (!is_dense(g) || g->n_img < kX8MinRows || g->ld_in % 4 || g->cin % 4 || g->ld_out % 4) return xg8::Plan{false, 0, 0, 0, 1, 0, 0, 0, 0};
  return xg8::plan(g->n_img, g->cout, g->cin, (long long)g->n_img * g->ld_in * 4, false, false);
}
xg8::Plan x8_dgrad_plan(const seedhip_conv_geom* g) {
  if (!is_dense(g) || g->n_img < kX8MinRows || g->ld_out % 4 || g->cout % 4 || g->ld_in % 4 || g->cin % 4) return xg8::Plan{false, 0, 0, 0, 1, 0, 0, 0, 0};
  return xg8::plan(g->n_img, g->cin, g->cout, 0, true, false);
}
xg8::Plan x8_wgrad_plan(const seedhip_conv_geom* g) {
  if (!is_dense(g) || g->n_img < kX8MinRows || g->ld_in % 4 || g->cin % 4 || g->ld_out % 4 || g->cout % 4) return xg8::Plan{false, 0, 0, 0, 1, 0, 0, 0, 0};
  // the PRE-SPLIT operand is dY [rows x cout]: worth it only while it is the smaller of the two (an LSTM input projection,
  // 532 -> 2048, would spend more on splitting 170 MB of dY than the kernel saves: 413 vs 328 us, r4)
  if (g->cout > g->cin) return xg8::Plan{false, 0, 0, 0, 1, 0, 0, 0, 0};
  return xg8::plan(g->cin, g->cout, g->n_img, (long long)g->n_img * g->ld_in * 4, false, true);
}
size_t x8_ws(const xg8::Plan& xp, int M, int N, bool colsum_partials) {
  if (!xp.ok) return 0;
  size_t part = xg8::partial_bytes(M, N, xp);
  if (colsum_partials) part = (size_t)xp.slices * ((size_t)M * N + N) * sizeof(float);
  return xg8::al256(part) + xg8::planes_bytes(xp);
}

int check_geom(const seedhip_conv_geom* g, const char* what) {
  SEEDHIP_REQUIRE(g, "%s: null geometry", what);
  SEEDHIP_REQUIRE(g->n_img >= 1 && g->ih >= 1 && g->iw >= 1 && g->cin >= 1 && g->oh >= 1 && g->ow >= 1 &&
                  g->kh >= 1 && g->kw >= 1 && g->stride >= 1 && g->cout >= 1 && g->pad_t >= 0 && g->pad_l >= 0,
                  "%s: non-positive geometry field", what);
  SEEDHIP_REQUIRE(g->ld_in >= g->cin && g->ld_out >= g->cout, "%s: ld_in/ld_out smaller than channels", what);
  SEEDHIP_REQUIRE((long long)g->n_img * g->oh * g->ow < (1LL << 31) && (long long)g->n_img * g->ih * g->iw < (1LL << 31),
                  "%s: more than 2^31 pixels", what);
  SEEDHIP_REQUIRE((g->oh - 1) * g->stride - g->pad_t < g->ih && (g->ow - 1) * g->stride - g->pad_l < g->iw,
                  "%s: output extent does not fit the padded input", what);
  return SEEDHIP_OK;
}

}  // namespace

// Which matrix pipe serves this geometry (pass 0 forward, 1 data gradient, 2 weight gradient): 1 = fp32 MFMA
// (v_mfma_f32_16x16x4_f32), 6 = bf16 MFMA through the exact three-way split of both operands (xgemm.h: six bf16 MACs per
// algorithmic MAC).  What the bench prices a kernel's roofline with; 0 = unknown pass / null geometry.
// ONE switch for the conv layers' bf16x6 kernels (wfx / wdx / wsx / wsy / wgx): SEEDHIP_CONV_BF16X6=0 puts those layers back
// on the fp32-MFMA kernels (same-box A/B of the two pipes; tests/test_gpu_kernels.py runs the parity cases under it).
// Bits: 1 forward, 2 data gradient, 4 weight gradient; default 7.
static int conv_x6() { static const int on = getenv("SEEDHIP_CONV_BF16X6") ? atoi(getenv("SEEDHIP_CONV_BF16X6")) : 7; return on; }
static int wsx_enabled(int pass) { return conv_x6() & (1 << pass); }
static int wdx_enabled() { return conv_x6() & 2; }
static int wgx_enabled() { return conv_x6() & 4; }
static int wfx_enabled() { return conv_x6() & 1; }

extern "C" int seedhip_conv2d_pipe(const seedhip_conv_geom* g, int pass) {
  if (!g || pass < 0 || pass > 2) return 0;
  if (pass == 0 && wfx_enabled()) { wfx::Params xp; if (wfx::plan(xp, g)) return 6; }
  if (pass == 1 && wdx_enabled()) { wdx::Params dp; if (wdx::plan(dp, g)) return 6; }
  if (pass <= 1 && wsx_enabled(pass) && (wsx::geometry(g) || wsy::geometry(g) || fgx::plan(g) || cgx::plan(g) || (pass == 0 && cgx::plan_fwd2(g)) || (pass == 1 && cgx::plan_dgrad2(g)))) return 6;
  if (pass == 2 && wgx_enabled() && wgx::plan(g)) return 6;
  if (xg8::mode() & (1 << pass)) {
    const xg8::Plan x8 = pass == 0 ? x8_fwd_plan(g) : pass == 1 ? x8_dgrad_plan(g) : x8_wgrad_plan(g);
    if (x8.ok) return 6;
  }
  if (!(xg::mode() & (1 << pass))) return 1;
  const xg::Plan xp = pass == 0 ? x6_fwd_plan(g) : pass == 1 ? x6_dgrad_plan(g) : x6_wgrad_plan(g);
  return xp.ok ? 6 : 1;
}

extern "C" size_t seedhip_conv2d_fwd_workspace_bytes(const seedhip_conv_geom* g) {
  if (!g || !is_dense(g)) return 0;
  const int sl = dense_slices(g->n_img, g->cout, g->cin);
  const size_t core = sl > 1 ? (size_t)sl * g->n_img * g->cout * sizeof(float) : 0;
  size_t mm = gemm_fwd_ok(g) ? gemm_partial_bytes(g->n_img, g->cout, g->cin) : 0;
  if (xg::mode() & 1) {
    const xg::Plan xp = x6_fwd_plan(g);
    if (xp.ok && xg::partial_bytes(g->n_img, g->cout, xp) > mm) mm = xg::partial_bytes(g->n_img, g->cout, xp);
  }
  if (xg8::mode() & 1) { const size_t w8 = x8_ws(x8_fwd_plan(g), g->n_img, g->cout, false); if (w8 > mm) mm = w8; }
  return mm > core ? mm : core;
}
extern "C" size_t seedhip_conv2d_bwd_data_workspace_bytes(const seedhip_conv_geom* g) {
  if (!g || !is_dense(g)) return 0;
  const int sl = dense_slices(g->n_img, g->cin, g->cout);
  const size_t core = sl > 1 ? (size_t)sl * g->n_img * g->cin * sizeof(float) : 0;
  size_t mm = gemm_dgrad_ok(g) ? gemm_partial_bytes(g->n_img, g->cin, g->cout) : 0;
  if (xg::mode() & 2) {
    const xg::Plan xp = x6_dgrad_plan(g);
    if (xp.ok && xg::partial_bytes(g->n_img, g->cin, xp) > mm) mm = xg::partial_bytes(g->n_img, g->cin, xp);
  }
  if (xg8::mode() & 2) { const size_t w8 = x8_ws(x8_dgrad_plan(g), g->n_img, g->cin, false); if (w8 > mm) mm = w8; }
  return mm > core ? mm : core;
}

extern "C" int seedhip_conv2d_fwd(const seedhip_conv_geom* geom, const void* in, int in_dtype, int in_relu,
                                  const float* w, const float* bias, float* out, int out_relu,
                                  const float* residual, void* stream) {
  return seedhip_conv2d_fwd_ws(geom, in, in_dtype, in_relu, w, bias, out, out_relu, residual, nullptr, 0, stream);
}

// out = relu(conv(relu?(in)) + bias) AND its ReLU mask as bytes [pixel][cout / 4] (bit r of byte q = out[pixel][4 q + r] > 0):
// what seedhip_conv2d_bwd_data_bits of the NEXT layer reads instead of `out` (r5; served by wfx.h for its one geometry).
extern "C" int seedhip_conv2d_fwd_bits_supported(const seedhip_conv_geom* geom) {
  if (!geom || check_geom(geom, "conv2d_fwd_bits_supported") || !wfx_enabled()) return 0;
  wfx::Params xp;
  return wfx::plan(xp, geom) ? 1 : 0;
}
extern "C" int seedhip_conv2d_fwd_bits(const seedhip_conv_geom* geom, const void* in, int in_dtype, int in_relu,
                                       const float* w, const float* bias, float* out, uint8_t* relu_bits, void* stream) {
  int rc = check_geom(geom, "conv2d_fwd_bits"); if (rc) return rc;
  SEEDHIP_REQUIRE(in && w && out && relu_bits, "conv2d_fwd_bits: null pointer");
  wfx::Params xp;
  if (!(wfx_enabled() && in_dtype == kInF32 && al16(in) && al16(w) && al16(out) && al16(bias) && wfx::plan(xp, geom)))
    return fail(SEEDHIP_ERR_UNSUPPORTED, "conv2d_fwd_bits: geometry / alignment not served (ask seedhip_conv2d_fwd_bits_supported)");
  xp.X = (const float*)in; xp.W = w; xp.bias = bias; xp.Y = out; xp.in_relu = in_relu; xp.out_relu = 1; xp.bits = relu_bits;
  return wfx::launch(xp, (hipStream_t)stream);
}

// ImpalaDeep's residual-block layers (wsx.h / wsy.h): out = conv(relu?(in)) + bias (+ residual) AND the sign of `out` as
// bytes [pixel][cout / 4] -- the ReLU mask of the data gradient of the layer that reads `out` through a ReLU
// (seedhip_conv2d_bwd_data_bits_add) -- written from the epilogue's registers next to the output itself.
extern "C" int seedhip_conv2d_fwd_outbits_supported(const seedhip_conv_geom* geom) {
  if (!geom || check_geom(geom, "conv2d_fwd_outbits_supported") || !wsx_enabled(0) || !wsx_enabled(1)) return 0;
  return (wsx::geometry(geom) || wsy::geometry(geom)) ? 1 : 0;
}
extern "C" int seedhip_conv2d_fwd_outbits(const seedhip_conv_geom* geom, const float* in, int in_relu, const float* w,
                                          const float* bias, float* out, const float* residual, uint8_t* out_bits,
                                          void* stream) {
  int rc = check_geom(geom, "conv2d_fwd_outbits"); if (rc) return rc;
  SEEDHIP_REQUIRE(in && w && out && out_bits, "conv2d_fwd_outbits: null pointer");
  if (!(seedhip_conv2d_fwd_outbits_supported(geom) && al16(in) && al16(w) && al16(out) && al16(bias) && al16(residual)))
    return fail(SEEDHIP_ERR_UNSUPPORTED, "conv2d_fwd_outbits: geometry / alignment not served (ask seedhip_conv2d_fwd_outbits_supported)");
  wsx::Params sp;
  memset(&sp, 0, sizeof(sp));
  sp.X = in; sp.Wt = w; sp.bias = bias; sp.A = residual; sp.Y = out; sp.n_img = geom->n_img;
  sp.in_relu = in_relu; sp.out_relu = 0; sp.out_bits = out_bits;
  const int geo = wsx::geometry(geom);
  rc = geo ? wsx::launch(geo, false, sp, (hipStream_t)stream) : wsy::launch(wsy::geometry(geom), false, sp, (hipStream_t)stream);
  return rc >= 0 ? rc : fail(SEEDHIP_ERR_LAUNCH, "conv2d_fwd_outbits: launch");
}

extern "C" int seedhip_conv2d_fwd_ws(const seedhip_conv_geom* geom, const void* in, int in_dtype, int in_relu,
                                     const float* w, const float* bias, float* out, int out_relu,
                                     const float* residual, void* workspace, size_t workspace_bytes, void* stream) {
  int rc = check_geom(geom, "conv2d_fwd"); if (rc) return rc;
  SEEDHIP_REQUIRE(in && w && out, "conv2d_fwd: null pointer");
  SEEDHIP_REQUIRE(in_dtype == kInF32 || in_dtype == kInU8Div255, "conv2d_fwd: bad in_dtype %d", in_dtype);
  {
    // the second Atari conv at training batch sizes on the bf16 matrix pipe (wfx.h: exact three-way split, six products)
    wfx::Params xp;
    if (wfx_enabled() && in_dtype == kInF32 && !residual && al16(in) && al16(w) && al16(out) && al16(bias) && wfx::plan(xp, geom)) {
      xp.X = (const float*)in; xp.W = w; xp.bias = bias; xp.Y = out; xp.in_relu = in_relu; xp.out_relu = out_relu;
      const int rc2 = wfx::launch(xp, (hipStream_t)stream);
      if (rc2 >= 0) return rc2;
    }
    // the DQN torso's 3 x 3 64 -> 64 layer on 9 x 9 maps (cgx.h): bias + ReLU
    if (wsx_enabled(0) && in_dtype == kInF32 && !in_relu && !residual && al16(in) && al16(w) && al16(out) && al16(bias) &&
        (cgx::plan(geom) || cgx::plan_fwd2(geom)))
      return cgx::launch_fwd(geom, (const float*)in, w, bias, out, out_relu, (hipStream_t)stream);
  }
  if ((gemm_mode() & 8) && in_dtype == kInF32 && al16(in) && al16(w) && al16(out) && al16(bias) && al16(residual)) {
    // whole kernel resident in LDS, A rows gathered as 128-byte segments (wsgemm.h): the second Atari conv
    wsgemm::Params wp;
    wsgemm::Plan pl = wsgemm::plan_fwd(wp, geom);
    if (pl.ok) {
      wp.A = (const float*)in; wp.a_relu = in_relu; wp.W = w; wp.C = out; wp.bias = bias; wp.out_relu = out_relu;
      wp.residual = residual;
      return wsgemm::launch(wp, pl, (hipStream_t)stream);
    }
  }
  if ((gemm_mode() & 32) && !is_dense(geom) && in_dtype == kInF32 && al16(in) && al16(w) && al16(out) &&
      (geom->cout >= conv_min_n() || (!(halo_all() & 1) && geom->cout == 32 && geom->cin == 32 && geom->stride == 1 && geom->oh * geom->ow >= 400))) {
    // convs with >= 64 output channels on the GEMM core with a gathered A operand (gemm.h); the 16 / 32-channel
    // layers are faster on the halo kernels (see halo_all() for the 32 -> 32 layers on large maps)
    gemm::Params gp;
    if (gemm::conv_fwd_setup(gp, geom)) {
      gp.A = (const float*)in; gp.a_relu = in_relu; gp.B = w; gp.C = out; gp.bias = bias; gp.residual = residual;
      gp.out_relu = out_relu;
      gemm::Plan pl = gemm::plan(gp.M, gp.N, gp.K);
      pl.slices = 1; pl.k_per_slice = gp.K;
      gemm::launch<true, false, true, false, false>(gp, pl, (hipStream_t)stream);
      return check_launch("conv2d_fwd(gather gemm)");
    }
  }
  {
    // the 32 -> 32 3x3 'same' layers of ImpalaDeep on the bf16 matrix pipe (wsx.h)
    const int geo = wsx_enabled(0) && in_dtype == kInF32 ? wsx::geometry(geom) : 0;
    if (geo && al16(in) && al16(w) && al16(out) && al16(bias) && al16(residual)) {
      wsx::Params sp;
      memset(&sp, 0, sizeof(sp));
      sp.X = (const float*)in; sp.Wt = w; sp.bias = bias; sp.A = residual; sp.Y = out; sp.n_img = geom->n_img;
      sp.in_relu = in_relu; sp.out_relu = out_relu;
      const int rc2 = wsx::launch(geo, false, sp, (hipStream_t)stream);
      if (rc2 >= 0) return rc2;
    }
    const int geo16 = wsx_enabled(0) && in_dtype == kInF32 ? wsy::geometry(geom) : 0;      // the 16 -> 16 layers (wsy.h)
    if (geo16 && al16(in) && al16(w) && al16(out) && al16(bias) && al16(residual)) {
      wsx::Params sp;
      memset(&sp, 0, sizeof(sp));
      sp.X = (const float*)in; sp.Wt = w; sp.bias = bias; sp.A = residual; sp.Y = out; sp.n_img = geom->n_img;
      sp.in_relu = in_relu; sp.out_relu = out_relu;
      const int rc2 = wsy::launch(geo16, false, sp, (hipStream_t)stream);
      if (rc2 >= 0) return rc2;
    }
    // ImpalaDeep's 16 -> 32 stack-entry layer on the 36 x 48 map (fgx.h): plain convolution + bias
    if (wsx_enabled(0) && in_dtype == kInF32 && !in_relu && !out_relu && !residual && al16(in) && al16(w) && al16(out) &&
        al16(bias) && fgx::plan(geom))
      return fgx::launch_fwd(geom, (const float*)in, w, bias, out, (hipStream_t)stream);
  }
  {
    // small-kernel layers: input band staged once in LDS (halo_fwd.h).  Measured on MI355X
    // (tools/bench_kernels.py, cfg3 step): the halo forward wins for stride-1 layers; the implicit-GEMM core stays
    // ahead for stride 2.
    const bool halo_wins = geom->stride == 1 && geom->kh * geom->kw > 1 && ((halo_all() & 1) || !(geom->cin >= 32 && geom->oh * geom->ow >= 400));
    auto al16 = [](const void* q) { return (((uintptr_t)q) & 15) == 0; };
    if (halo_wins && al16(out) && al16(bias) && al16(residual) && (in_dtype == kInU8Div255 || al16(in))) {
      halo::FwdParams hp;
      memset(&hp, 0, sizeof(hp));
      hp.in = in; hp.in_dtype = in_dtype; hp.in_relu = in_relu; hp.w = w; hp.wmode = 0;
      hp.w_kh = geom->kh; hp.w_kw = geom->kw; hp.w_cin = geom->cin; hp.w_cout = geom->cout;
      hp.n_img = geom->n_img; hp.ih = geom->ih; hp.iw = geom->iw; hp.cin = geom->cin; hp.stride = geom->stride;
      hp.cout = geom->cout; hp.out = out; hp.OH = geom->oh; hp.OW = geom->ow; hp.ld_out = geom->ld_out; hp.so = 1;
      hp.bias = bias; hp.out_relu = out_relu; hp.residual = residual; hp.ld_in = geom->ld_in;
      halo::ClassSpec cs = {geom->kh, geom->kw, geom->pad_t, geom->pad_l, geom->oh, geom->ow, 0, 0, 0, 0};
      const halo::FwdPlan pl = halo::plan_fwd(hp, &cs, 1, in_dtype == kInU8Div255);
      if (pl.ok) return halo::launch_fwd_kernel(hp, pl, (hipStream_t)stream);
    }
  }
  if ((xg8::mode() & 1) && is_dense(geom) && in_dtype == kInF32 && al16(in) && al16(w) && al16(out) && workspace &&
      al16(workspace) && al16(bias) && al16(residual)) {
    const xg8::Plan xp = x8_fwd_plan(geom);
    const int M = geom->n_img, N = geom->cout, K = geom->cin;
    if (xp.ok && workspace_bytes >= x8_ws(xp, M, N, false)) {
      hipStream_t s = (hipStream_t)stream;
      unsigned char* ws = (unsigned char*)workspace;
      float* partial = xp.slices > 1 ? (float*)ws : nullptr;
      void* Bp = ws + xg8::al256(xg8::partial_bytes(M, N, xp));
      xg8::launch_split<256>(w, N, N, K, false, false, Bp, nullptr, s);        // B(k, n) = w[k][n]
      gemm::Params gp;
      memset(&gp, 0, sizeof(gp));
      gp.A = (const float*)in; gp.lda = geom->ld_in; gp.a_relu = in_relu; gp.B = w; gp.ldb = N;
      gp.M = M; gp.N = N; gp.K = K; gp.C = out; gp.ldc = geom->ld_out;
      gp.bias = bias; gp.residual = residual; gp.out_relu = out_relu; gp.partial = partial;
      if (xg8::launch<0>(gp, xp, nullptr, Bp, nullptr, s)) {
        if (xp.slices > 1) {
          int blocks = cdiv((long long)M * N / 4, 256); if (blocks > 2048) blocks = 2048;
          hipLaunchKernelGGL(dense_epilogue4_kernel, dim3(blocks), dim3(256), 0, s, partial, xp.slices, M, N, bias,
                             residual, out_relu, (const float*)nullptr, (const float*)nullptr, out, geom->ld_out);
        }
        return check_launch("conv2d_fwd(dense, bf16x6 / 8 waves)");
      }
    }
  }
  if ((xg::mode() & 1) && is_dense(geom) && in_dtype == kInF32 && al16(in) && al16(w) && al16(out) && al16(workspace) &&
      al16(bias) && al16(residual)) {
    const xg::Plan xp = x6_fwd_plan(geom);
    const int M = geom->n_img, N = geom->cout, K = geom->cin;
    if (xp.ok && (xp.slices == 1 || (workspace && workspace_bytes >= xg::partial_bytes(M, N, xp)))) {
      hipStream_t s = (hipStream_t)stream;
      gemm::Params gp;
      memset(&gp, 0, sizeof(gp));
      gp.A = (const float*)in; gp.lda = geom->ld_in; gp.a_relu = in_relu; gp.B = w; gp.ldb = N;
      gp.M = M; gp.N = N; gp.K = K; gp.C = out; gp.ldc = geom->ld_out;
      gp.bias = bias; gp.residual = residual; gp.out_relu = out_relu;
      if (xp.slices > 1) gp.partial = (float*)workspace;
      xg::launch<true, false>(gp, xp, s);
      if (xp.slices > 1) {
        int blocks = cdiv((long long)M * N, 256); if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(dense_epilogue_kernel, dim3(blocks), dim3(256), 0, s, gp.partial, xp.slices, M, N, bias,
                           residual, out_relu, (const float*)nullptr, (const float*)nullptr, out, geom->ld_out);
      }
      return check_launch("conv2d_fwd(dense, bf16x6)");
    }
  }
  if (is_dense(geom) && in_dtype == kInF32 && geom->ld_in % 4 == 0 && (((uintptr_t)in) & 15) == 0) {
    if (gemm_fwd_ok(geom) && al16(w) && al16(out) && al16(workspace)) {
      const int M = geom->n_img, N = geom->cout, K = geom->cin;
      const gemm::Plan pl = gemm::plan(M, N, K);
      if (pl.slices == 1 || (workspace && workspace_bytes >= (size_t)pl.slices * M * N * sizeof(float))) {
        hipStream_t s = (hipStream_t)stream;
        gemm::Params gp;
        memset(&gp, 0, sizeof(gp));
        gp.A = (const float*)in; gp.lda = geom->ld_in; gp.a_relu = in_relu; gp.B = w; gp.ldb = N;
        gp.M = M; gp.N = N; gp.K = K; gp.k_per_slice = pl.k_per_slice; gp.C = out; gp.ldc = geom->ld_out;
        gp.bias = bias; gp.residual = residual; gp.out_relu = out_relu;
        if (pl.slices > 1) gp.partial = (float*)workspace;
        gemm::launch<true, false>(gp, pl, s);
        if (pl.slices > 1) {
          int blocks = cdiv((long long)M * N, 256); if (blocks > 2048) blocks = 2048;
          hipLaunchKernelGGL(dense_epilogue_kernel, dim3(blocks), dim3(256), 0, s, gp.partial, pl.slices, M, N, bias,
                             residual, out_relu, (const float*)nullptr, (const float*)nullptr, out, geom->ld_out);
        }
        return check_launch("conv2d_fwd(dense, gemm)");
      }
    }
    if (geom->cin % 4 == 0) {
    DenseFwd d;
    d.in = (const float*)in; d.in_relu = in_relu; d.w = w; d.bias = bias; d.out = out; d.out_relu = out_relu;
    d.residual = residual;
    d.init(to_geom(geom));
    const int sl = dense_slices(d.M, d.N, d.K);
    if (sl > 1 && workspace && workspace_bytes >= (size_t)sl * d.M * d.N * sizeof(float)) {
      // under-filled grid (inference batches, recurrent steps): split K, reduce + epilogue in a second launch
      d.k_per_slice = slice_k(d.K, sl); d.partial = (float*)workspace;
      const int slices = (d.K + d.k_per_slice - 1) / d.k_per_slice;
      launch_igemm_auto(d, slices, (hipStream_t)stream);
      int blocks = cdiv((long long)d.M * d.N, 256); if (blocks > 2048) blocks = 2048;
      hipLaunchKernelGGL(dense_epilogue_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, d.partial, slices, d.M,
                         d.N, bias, residual, out_relu, (const float*)nullptr, (const float*)nullptr, out, d.ld_out);
      return check_launch("conv2d_fwd(dense, split-K)");
    }
    launch_igemm_auto(d, 1, (hipStream_t)stream);
    return check_launch("conv2d_fwd(dense)");
    }
  }
  ConvFwd p;
  p.in = in; p.in_dtype = in_dtype; p.in_relu = in_relu; p.w = w; p.bias = bias; p.out = out;
  p.out_relu = out_relu; p.residual = residual;
  p.init(to_geom(geom));
  launch_igemm_auto(p, 1, (hipStream_t)stream);
  return check_launch("conv2d_fwd");
}

// Dense forward WITHOUT its epilogue: partial[z][m][n] = sum over the k of slice z of in[m][k] w[k][n], z < *slices (>= 1).
// The caller's next kernel sums the slices in order, adds the bias and applies the activation while it loads its own
// operand (servestep.hip: the Dense layer of central inference feeds the heads) -- the reduce + epilogue launch of the
// split-K paths and the activation's round trip through memory disappear.  Same kernels, plans and slice order as
// seedhip_conv2d_fwd_ws takes for the shape (bf16x6 xgemm.h from 2048 rows, gemm.h below): summing the slices in order
// + bias (+ ReLU) reproduces its output bit for bit.
extern "C" size_t seedhip_dense_fwd_partial_workspace_bytes(const seedhip_conv_geom* g) {
  if (!g || !is_dense(g)) return 0;
  int sl = 1;
  if (xg::mode() & 1) { const xg::Plan xp = x6_fwd_plan(g); if (xp.ok) sl = xp.slices; }
  const gemm::Plan pl = gemm::plan(g->n_img, g->cout, g->cin);
  if (pl.slices > sl) sl = pl.slices;
  return (size_t)sl * g->n_img * g->cout * sizeof(float);
}

extern "C" int seedhip_dense_fwd_partial(const seedhip_conv_geom* geom, const float* in, int in_relu, const float* w,
                                         void* workspace, size_t workspace_bytes, int* slices, void* stream) {
  SEEDHIP_REQUIRE(geom && in && w && workspace && slices, "dense_fwd_partial: null pointer");
  SEEDHIP_REQUIRE(is_dense(geom) && gemm_fwd_ok(geom) && al16(in) && al16(w) && al16(workspace),
                  "dense_fwd_partial: needs a Dense geometry with ld_in %% 4 == 0, cout %% 4 == 0 and 16-byte aligned buffers");
  SEEDHIP_REQUIRE(workspace_bytes >= seedhip_dense_fwd_partial_workspace_bytes(geom), "dense_fwd_partial: workspace too small");
  const int M = geom->n_img, N = geom->cout, K = geom->cin;
  hipStream_t s = (hipStream_t)stream;
  gemm::Params gp;
  memset(&gp, 0, sizeof(gp));
  gp.A = in; gp.lda = geom->ld_in; gp.a_relu = in_relu; gp.B = w; gp.ldb = N;
  gp.M = M; gp.N = N; gp.K = K; gp.C = (float*)workspace; gp.ldc = N; gp.partial = (float*)workspace;
  if (xg::mode() & 1) {
    const xg::Plan xp = x6_fwd_plan(geom);
    if (xp.ok) {
      xg::launch<true, false>(gp, xp, s);
      *slices = xp.slices;
      return check_launch("dense_fwd_partial(bf16x6)");
    }
  }
  const gemm::Plan pl = gemm::plan(M, N, K);
  gp.k_per_slice = pl.k_per_slice;
  gemm::launch<true, false>(gp, pl, s);
  *slices = pl.slices;
  return check_launch("dense_fwd_partial(gemm)");
}

extern "C" int seedhip_conv2d_bwd_data(const seedhip_conv_geom* geom, const float* dy, const float* w, float* dx,
                                       const float* relu_mask, const float* add, void* stream) {
  return seedhip_conv2d_bwd_data_ws(geom, dy, w, dx, relu_mask, add, nullptr, 0, stream);
}

// Data gradient with the ReLU mask as bytes (one per four input channels, written by seedhip_conv2d_stack_fwd_bits).
// Dense data gradient with the byte mask: the bf16x6 kernel of xgemm.h, unsplit reduction only (the split-K epilogue
// kernel reads the fp32 mask)
static bool dense_bits_ok(const seedhip_conv_geom* g) {
  // the mask is indexed [row][cin / 4]: a padded input row (ld_in > cin, the LSTM-input layers) would read it with the
  // wrong pitch
  if (!(xg::mode() & 2) || !is_dense(g) || g->ld_in != g->cin) return false;
  const xg::Plan xp = x6_dgrad_plan(g);
  return xp.ok && xp.slices == 1;
}

// dX = mask ? dgrad : 0 (+ add) with the mask as bytes, on wsx.h / wsy.h; -1: geometry not served there
static int wsx_dgrad_bits(const seedhip_conv_geom* geom, const float* dy, const float* w, float* dx, const uint8_t* relu_bits,
                          const float* add, void* stream) {
  if (!wsx_enabled(1) || !al16(dy) || !al16(w) || !al16(dx) || !al16(add)) return -1;
  const int geo = wsx::geometry(geom), geo16 = geo ? 0 : wsy::geometry(geom);
  if (!geo && !geo16) return -1;
  wsx::Params sp;
  memset(&sp, 0, sizeof(sp));
  sp.X = dy; sp.Wt = w; sp.B = add; sp.Y = dx; sp.n_img = geom->n_img; sp.mask_bits = relu_bits;
  return geo ? wsx::launch(geo, true, sp, (hipStream_t)stream) : wsy::launch(geo16, true, sp, (hipStream_t)stream);
}

extern "C" int seedhip_conv2d_bwd_data_bits_supported(const seedhip_conv_geom* geom) {
  if (!geom || check_geom(geom, "conv2d_bwd_data_bits_supported")) return 0;
  if (dense_bits_ok(geom)) return 1;
  if (wsx_enabled(1) && (wsx::geometry(geom) || wsy::geometry(geom))) return 1;
  if (!(gemm_mode() & 16)) return 0;
  wsgemm::Params wp;
  wsgemm::Plan pl = wsgemm::plan_dgrad(wp, geom);
  if (!pl.ok) return 0;
  wp.mask_bits = reinterpret_cast<const unsigned char*>(16);          // any non-null value: nothing is launched
  return wsgemm::launch(wp, pl, nullptr, true) == SEEDHIP_OK;
}

extern "C" int seedhip_conv2d_bwd_data_bits(const seedhip_conv_geom* geom, const float* dy, const float* w, float* dx,
                                            const uint8_t* relu_bits, void* stream) {
  int rc = check_geom(geom, "conv2d_bwd_data_bits"); if (rc) return rc;
  SEEDHIP_REQUIRE(dy && w && dx && relu_bits, "conv2d_bwd_data_bits: null pointer");
  SEEDHIP_REQUIRE(al16(dy) && al16(w) && al16(dx), "conv2d_bwd_data_bits: operands must be 16-byte aligned");
  if (dense_bits_ok(geom)) {
    const xg::Plan xp = x6_dgrad_plan(geom);
    const int M = geom->n_img, N = geom->cin, K = geom->cout;
    gemm::Params gp;
    memset(&gp, 0, sizeof(gp));
    gp.A = dy; gp.lda = geom->ld_out; gp.B = w; gp.ldb = K; gp.M = M; gp.N = N; gp.K = K;
    gp.C = dx; gp.ldc = geom->ld_in; gp.mask_bits = relu_bits;
    xg::launch<true, true>(gp, xp, (hipStream_t)stream);
    return check_launch("conv2d_bwd_data_bits(dense, bf16x6)");
  }
  {
    const int rc2 = wsx_dgrad_bits(geom, dy, w, dx, relu_bits, nullptr, stream);
    if (rc2 >= 0) return rc2;
  }
  SEEDHIP_REQUIRE(gemm_mode() & 16, "conv2d_bwd_data_bits: not served (ask seedhip_conv2d_bwd_data_bits_supported)");
  {
    // the second Atari conv at training batch sizes on the bf16 matrix pipe, the mask byte per 16 output bytes (wdx.h)
    wdx::Params dp;
    if (wdx_enabled() && wdx::plan(dp, geom)) {
      dp.dY = dy; dp.W = w; dp.X = nullptr; dp.bits = relu_bits; dp.dX = dx;
      const int rc2 = wdx::launch(dp, (hipStream_t)stream);
      if (rc2 >= 0) return rc2;
    }
  }
  wsgemm::Params wp;
  wsgemm::Plan pl = wsgemm::plan_dgrad(wp, geom);
  if (!pl.ok) return fail(SEEDHIP_ERR_UNSUPPORTED, "conv2d_bwd_data_bits: geometry not served (ask seedhip_conv2d_bwd_data_bits_supported)");
  wp.A = dy; wp.W = w; wp.C = dx; wp.mask_bits = relu_bits;
  return wsgemm::launch(wp, pl, (hipStream_t)stream);
}

// The same with the skip path's gradient added behind the mask (dX = (bit ? dgrad : 0) + add): the first layer of a
// residual block.  Served where seedhip_conv2d_fwd_outbits is; add == NULL: seedhip_conv2d_bwd_data_bits.
extern "C" int seedhip_conv2d_bwd_data_bits_add(const seedhip_conv_geom* geom, const float* dy, const float* w, float* dx,
                                                const uint8_t* relu_bits, const float* add, void* stream) {
  if (!add) return seedhip_conv2d_bwd_data_bits(geom, dy, w, dx, relu_bits, stream);
  int rc = check_geom(geom, "conv2d_bwd_data_bits_add"); if (rc) return rc;
  SEEDHIP_REQUIRE(dy && w && dx && relu_bits, "conv2d_bwd_data_bits_add: null pointer");
  rc = wsx_dgrad_bits(geom, dy, w, dx, relu_bits, add, stream);
  return rc >= 0 ? rc : fail(SEEDHIP_ERR_UNSUPPORTED, "conv2d_bwd_data_bits_add: geometry / alignment not served (ask seedhip_conv2d_fwd_outbits_supported)");
}

// Data gradient of a convolution whose output went through MaxPool2D(3, 2, 'same') (ImpalaDeep's stack-entry layer,
// dmlab/networks.py:31-37), from the gradient of the POOLED map and the pool's argmax bytes: the max-pool backward runs in
// the loader (fgx.h) and the pre-pool gradient is also written to d_prepool [n, oh, ow, cout] for the weight gradient --
// results bit-identical to seedhip_maxpool3x3s2_same_bwd followed by seedhip_conv2d_bwd_data.
extern "C" int seedhip_conv2d_bwd_data_pool_supported(const seedhip_conv_geom* geom) {
  if (!geom || check_geom(geom, "conv2d_bwd_data_pool_supported") || !wsx_enabled(1)) return 0;
  return fgx::plan(geom) ? 1 : 0;
}
extern "C" int seedhip_conv2d_bwd_data_pool(const seedhip_conv_geom* geom, const float* dpooled, const uint8_t* argmax,
                                            const float* w, float* dx, float* d_prepool, void* stream) {
  int rc = check_geom(geom, "conv2d_bwd_data_pool"); if (rc) return rc;
  SEEDHIP_REQUIRE(dpooled && argmax && w && dx && d_prepool, "conv2d_bwd_data_pool: null pointer");
  if (!(seedhip_conv2d_bwd_data_pool_supported(geom) && al16(dpooled) && al16(w) && al16(dx) && al16(d_prepool) &&
        (((uintptr_t)argmax) & 3) == 0))
    return fail(SEEDHIP_ERR_UNSUPPORTED, "conv2d_bwd_data_pool: geometry / alignment not served (ask seedhip_conv2d_bwd_data_pool_supported)");
  return fgx::launch_dgrad_pool(geom, dpooled, argmax, w, dx, d_prepool, (hipStream_t)stream);
}

extern "C" int seedhip_conv2d_bwd_data_ws(const seedhip_conv_geom* geom, const float* dy, const float* w, float* dx,
                                          const float* relu_mask, const float* add, void* workspace,
                                          size_t workspace_bytes, void* stream) {
  int rc = check_geom(geom, "conv2d_bwd_data"); if (rc) return rc;
  SEEDHIP_REQUIRE(dy && w && dx, "conv2d_bwd_data: null pointer");
  {
    // the second Atari conv at training batch sizes on the bf16 matrix pipe (wdx.h)
    wdx::Params dp;
    if (wdx_enabled() && !add && al16(dy) && al16(w) && al16(dx) && al16(relu_mask) && wdx::plan(dp, geom)) {
      dp.dY = dy; dp.W = w; dp.X = relu_mask; dp.dX = dx;
      const int rc2 = wdx::launch(dp, (hipStream_t)stream);
      if (rc2 >= 0) return rc2;
    }
    // the DQN torso's 3 x 3 64 -> 64 layer (cgx.h)
    if (wsx_enabled(1) && !add && al16(dy) && al16(w) && al16(dx) && al16(relu_mask) && cgx::plan(geom))
      return cgx::launch_dgrad(geom, dy, w, dx, relu_mask, (hipStream_t)stream);
    if (wsx_enabled(1) && !add && al16(dy) && al16(w) && al16(dx) && al16(relu_mask) && cgx::plan_dgrad2(geom))
      return cgx::launch_dgrad2(geom, dy, w, dx, relu_mask, (hipStream_t)stream);
  }
  if ((gemm_mode() & 16) && al16(dy) && al16(w) && al16(dx) && al16(relu_mask) && al16(add)) {
    // all stride-parity classes as ONE weight-stationary GEMM over super-pixels (wsgemm.h)
    wsgemm::Params wp;
    wsgemm::Plan pl = wsgemm::plan_dgrad(wp, geom);
    if (pl.ok) {
      wp.A = dy; wp.W = w; wp.C = dx; wp.mask = relu_mask; wp.add = add;
      return wsgemm::launch(wp, pl, (hipStream_t)stream);
    }
  }
  // measured (ImpalaDeep 32 -> 32 @18x24: 3.74 -> 3.05 ms): with the 4x1-wave 32-column tiles the data gradient of
  // 32-channel layers on maps of >= 400 pixels also goes to the gather-GEMM; forward and weight gradient of such
  // layers stay on the halo kernels (equal / 2.7x slower there)
  const int dgrad_n = geom->stride * geom->stride * geom->cin;
  if ((gemm_mode() & 64) && !is_dense(geom) && al16(dy) && al16(w) &&
      (dgrad_n >= conv_min_n() || (!(halo_all() & 2) && dgrad_n == 32 && geom->ih * geom->iw >= 400 && geom->stride == 1))) {
    gemm::Params gp;
    if (gemm::conv_dgrad_setup(gp, geom)) {
      gp.A = dy; gp.B = w; gp.C = dx; gp.mask = relu_mask; gp.add = add;
      gemm::Plan pl = gemm::plan(gp.M, gp.N, gp.K);
      pl.slices = 1; pl.k_per_slice = gp.K;
      // 'valid' convs with several taps: rows in image-block x position order, so that a workgroup tile shares one
      // super-pixel and skips the taps that fall outside dY (gemm_geom.h Gather::blk; A/B: SEEDHIP_DGRAD_POS=0)
      constexpr int pos_major = 1;
      const int bm = (4 / pl.wn) * pl.mr * 16, nkt = (gp.K + gemm::BK - 1) / gemm::BK;
      const long long m_blk = (long long)((geom->n_img + bm - 1) / bm) * gp.ga.d1.d * bm;
      if (pos_major && geom->pad_t == 0 && geom->pad_l == 0 && gp.ga.ntaps > 1 && nkt <= 32 && geom->cout % gemm::BK == 0 &&
          m_blk < (1LL << 31)) {
        gp.ga.blk = bm; gp.ga.n_img = geom->n_img; gp.ga.d_blk.init(bm);
        gp.M = (int)m_blk;
      }
      gemm::launch<true, true, true, true, true>(gp, pl, (hipStream_t)stream);
      return check_launch("conv2d_bwd_data(gather gemm)");
    }
  }
  {
    // the same layers' data gradient: wsx.h with the weights flipped and transposed
    const int geo = wsx_enabled(1) ? wsx::geometry(geom) : 0;
    if (geo && al16(dy) && al16(w) && al16(dx) && al16(relu_mask) && al16(add)) {
      wsx::Params sp;
      memset(&sp, 0, sizeof(sp));
      sp.X = dy; sp.Wt = w; sp.A = relu_mask; sp.B = add; sp.Y = dx; sp.n_img = geom->n_img;
      const int rc2 = wsx::launch(geo, true, sp, (hipStream_t)stream);
      if (rc2 >= 0) return rc2;
    }
    const int geo16 = wsx_enabled(1) ? wsy::geometry(geom) : 0;
    if (geo16 && al16(dy) && al16(w) && al16(dx) && al16(relu_mask) && al16(add)) {
      wsx::Params sp;
      memset(&sp, 0, sizeof(sp));
      sp.X = dy; sp.Wt = w; sp.A = relu_mask; sp.B = add; sp.Y = dx; sp.n_img = geom->n_img;
      const int rc2 = wsy::launch(geo16, true, sp, (hipStream_t)stream);
      if (rc2 >= 0) return rc2;
    }
    if (wsx_enabled(1) && !relu_mask && !add && al16(dy) && al16(w) && al16(dx) && fgx::plan(geom))
      return fgx::launch_dgrad(geom, dy, w, dx, (hipStream_t)stream);
  }
  {
    // Data gradient as stride-1 halo convolutions of dY, one per stride-parity class of the input pixel, all in
    // ONE launch sharing the dY tile (halo_fwd.h): dX[q*s + py - pad] = sum_j dY[q - j] W[py + s*j].
    const seedhip_conv_geom* g = geom;
    const int s = g->stride;
    if (g->kh * g->kw > 1 && s <= 2 && g->kh >= s && g->kw >= s &&
        (((uintptr_t)dy | (uintptr_t)dx | (uintptr_t)relu_mask | (uintptr_t)add) & 15) == 0) {
      halo::ClassSpec cs[4];
      int ncls = 0;
      bool ok = true;
      for (int py = 0; py < s; ++py) {
        for (int px = 0; px < s; ++px) {
          const int JH = (g->kh - py + s - 1) / s, JW = (g->kw - px + s - 1) / s;
          const int q0y = py >= g->pad_t ? 0 : (g->pad_t - py + s - 1) / s;
          const int q0x = px >= g->pad_l ? 0 : (g->pad_l - px + s - 1) / s;
          const int QH = (g->ih - 1 + g->pad_t - py) / s - q0y + 1, QW = (g->iw - 1 + g->pad_l - px) / s - q0x + 1;
          if (QH < 1 || QW < 1) { ok = false; continue; }
          cs[ncls++] = halo::ClassSpec{JH, JW, JH - 1 - q0y, JW - 1 - q0x, QH, QW,
                                       q0y * s + py - g->pad_t, q0x * s + px - g->pad_l, py, px};
        }
      }
      if (ok && ncls == s * s) {
        halo::FwdParams hp;
        memset(&hp, 0, sizeof(hp));
        hp.in = dy; hp.in_dtype = kInF32; hp.in_relu = 0; hp.w = w; hp.wmode = 1;
        hp.w_kh = g->kh; hp.w_kw = g->kw; hp.w_cin = g->cin; hp.w_cout = g->cout; hp.w_s = s;
        hp.n_img = g->n_img; hp.ih = g->oh; hp.iw = g->ow; hp.cin = g->cout; hp.stride = 1; hp.cout = g->cin;
        hp.out = dx; hp.OH = g->ih; hp.OW = g->iw; hp.ld_out = g->ld_in; hp.so = s;
        hp.mask = relu_mask; hp.add = add; hp.ld_in = g->ld_out;
        const halo::FwdPlan pl = halo::plan_fwd(hp, cs, ncls, false);
        if (pl.ok) return halo::launch_fwd_kernel(hp, pl, (hipStream_t)stream);
      }
    }
  }
  if ((xg8::mode() & 2) && is_dense(geom) && al16(dy) && al16(w) && al16(dx) && al16(relu_mask) && al16(add) && workspace &&
      al16(workspace)) {
    const xg8::Plan xp = x8_dgrad_plan(geom);
    const int M = geom->n_img, N = geom->cin, K = geom->cout;
    if (xp.ok && workspace_bytes >= x8_ws(xp, M, N, false)) {
      hipStream_t s = (hipStream_t)stream;
      unsigned char* ws = (unsigned char*)workspace;
      float* partial = xp.slices > 1 ? (float*)ws : nullptr;
      unsigned char* Ap = ws + xg8::al256(xg8::partial_bytes(M, N, xp));
      unsigned char* Bp = Ap + xg8::al256(xp.a_planes);
      xg8::launch_split<128>(dy, geom->ld_out, M, K, true, false, Ap, nullptr, s);   // A(m, k) = dy[m][k]
      xg8::launch_split<256>(w, K, N, K, true, false, Bp, nullptr, s);               // B(k, n) = w[n][k]
      gemm::Params gp;
      memset(&gp, 0, sizeof(gp));
      gp.A = dy; gp.lda = geom->ld_out; gp.B = w; gp.ldb = K; gp.M = M; gp.N = N; gp.K = K;
      gp.C = dx; gp.ldc = geom->ld_in; gp.mask = relu_mask; gp.add = add; gp.partial = partial;
      if (xg8::launch<2>(gp, xp, Ap, Bp, nullptr, s)) {
        if (xp.slices > 1) {
          int blocks = cdiv((long long)M * N / 4, 256); if (blocks > 2048) blocks = 2048;
          hipLaunchKernelGGL(dense_epilogue4_kernel, dim3(blocks), dim3(256), 0, s, partial, xp.slices, M, N,
                             (const float*)nullptr, (const float*)nullptr, 0, relu_mask, add, dx, geom->ld_in);
        }
        return check_launch("conv2d_bwd_data(dense, bf16x6 / 8 waves)");
      }
    }
  }
  if ((xg::mode() & 2) && is_dense(geom) && al16(dy) && al16(w) && al16(dx) && al16(relu_mask) && al16(add) && al16(workspace)) {
    const xg::Plan xp = x6_dgrad_plan(geom);
    const int M = geom->n_img, N = geom->cin, K = geom->cout;
    if (xp.ok && (xp.slices == 1 || (workspace && workspace_bytes >= xg::partial_bytes(M, N, xp)))) {
      hipStream_t s = (hipStream_t)stream;
      gemm::Params gp;
      memset(&gp, 0, sizeof(gp));
      gp.A = dy; gp.lda = geom->ld_out; gp.B = w; gp.ldb = K; gp.M = M; gp.N = N; gp.K = K;
      gp.C = dx; gp.ldc = geom->ld_in; gp.mask = relu_mask; gp.add = add;
      if (xp.slices > 1) gp.partial = (float*)workspace;
      xg::launch<true, true>(gp, xp, s);
      if (xp.slices > 1) {
        int blocks = cdiv((long long)M * N, 256); if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(dense_epilogue_kernel, dim3(blocks), dim3(256), 0, s, gp.partial, xp.slices, M, N,
                           (const float*)nullptr, (const float*)nullptr, 0, relu_mask, add, dx, geom->ld_in);
      }
      return check_launch("conv2d_bwd_data(dense, bf16x6)");
    }
  }
  if (is_dense(geom) && geom->cout % 4 == 0 && geom->ld_out % 4 == 0 && (((uintptr_t)dy | (uintptr_t)w) & 15) == 0) {
    if (gemm_dgrad_ok(geom)) {
      const int M = geom->n_img, N = geom->cin, K = geom->cout;
      const gemm::Plan pl = gemm::plan(M, N, K);
      if (pl.slices == 1 || (workspace && workspace_bytes >= (size_t)pl.slices * M * N * sizeof(float))) {
        hipStream_t s = (hipStream_t)stream;
        gemm::Params gp;
        memset(&gp, 0, sizeof(gp));
        gp.A = dy; gp.lda = geom->ld_out; gp.B = w; gp.ldb = K; gp.M = M; gp.N = N; gp.K = K;
        gp.k_per_slice = pl.k_per_slice; gp.C = dx; gp.ldc = geom->ld_in; gp.mask = relu_mask; gp.add = add;
        if (pl.slices > 1) gp.partial = (float*)workspace;
        gemm::launch<true, true>(gp, pl, s);
        if (pl.slices > 1) {
          int blocks = cdiv((long long)M * N, 256); if (blocks > 2048) blocks = 2048;
          hipLaunchKernelGGL(dense_epilogue_kernel, dim3(blocks), dim3(256), 0, s, gp.partial, pl.slices, M, N,
                             (const float*)nullptr, (const float*)nullptr, 0, relu_mask, add, dx, geom->ld_in);
        }
        return check_launch("conv2d_bwd_data(dense, gemm)");
      }
    }
    DenseDgrad d;
    d.dy = dy; d.w = w; d.dx = dx; d.mask = relu_mask; d.add = add;
    d.init(to_geom(geom));
    const int sl = dense_slices(d.M, d.N, d.K);
    if (sl > 1 && workspace && workspace_bytes >= (size_t)sl * d.M * d.N * sizeof(float)) {
      d.k_per_slice = slice_k(d.K, sl); d.partial = (float*)workspace;
      const int slices = (d.K + d.k_per_slice - 1) / d.k_per_slice;
      launch_igemm_auto(d, slices, (hipStream_t)stream);
      int blocks = cdiv((long long)d.M * d.N, 256); if (blocks > 2048) blocks = 2048;
      hipLaunchKernelGGL(dense_epilogue_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, d.partial, slices, d.M,
                         d.N, (const float*)nullptr, (const float*)nullptr, 0, relu_mask, add, dx, d.ld_in);
      return check_launch("conv2d_bwd_data(dense, split-K)");
    }
    launch_igemm_auto(d, 1, (hipStream_t)stream);
    return check_launch("conv2d_bwd_data(dense)");
  }
  ConvDgrad p;
  p.dy = dy; p.w = w; p.dx = dx; p.mask = relu_mask; p.add = add;
  p.init(to_geom(geom));
  launch_igemm_auto(p, p.slices(), (hipStream_t)stream);
  return check_launch("conv2d_bwd_data");
}

extern "C" size_t seedhip_conv2d_bwd_weight_workspace_bytes(const seedhip_conv_geom* g) {
  if (!g) return 0;
  const int M = g->kh * g->kw * g->cin, N = g->cout;
  const long long pixels = (long long)g->n_img * g->oh * g->ow;
  const int per = pick_k_per_slice(pixels, tiles_for(M, N));
  const long long slices = (pixels + per - 1) / per;
  const size_t generic = (size_t)slices * ((size_t)M * N + N) * sizeof(float);
  const halo::WgradPlan pl = halo::plan_wgrad(g);
  size_t need = (pl.ok && pl.ws_bytes > generic) ? pl.ws_bytes : generic;
  if (is_dense(g) && gemm_wgrad_ok(g)) {
    const gemm::Plan gpl = gemm::plan(g->cin, g->cout, g->n_img, 0.8);
    const size_t mm = (size_t)gpl.slices * ((size_t)g->cin * g->cout + g->cout) * sizeof(float);
    if (mm > need) need = mm;
  }
  if (conv_wgrad_gemm_ok(g)) {
    const gemm::Plan gpl = conv_wgrad_plan(g);
    const size_t mm = (size_t)gpl.slices * ((size_t)M * N + N) * sizeof(float);
    if (mm > need) need = mm;
  }
  if (xg::mode() & 4) {
    const xg::Plan xp = x6_wgrad_plan(g);
    const size_t mm = xp.ok ? (size_t)xp.slices * ((size_t)M * N + N) * sizeof(float) : 0;
    if (mm > need) need = mm;
  }
  if (xg8::mode() & 4) { const size_t w8 = x8_ws(x8_wgrad_plan(g), M, N, true); if (w8 > need) need = w8; }
  if (const int k = wgx::plan(g)) {
    const size_t mm = (size_t)wgx::grid_for(k, g->n_img) * ((size_t)M * N + N) * sizeof(float);
    if (mm > need) need = mm;
  }
  return need;
}

extern "C" int seedhip_conv2d_bwd_weight(const seedhip_conv_geom* geom, const void* in, int in_dtype, int in_relu,
                                         const float* dy, float* dw, float* dbias, void* workspace,
                                         size_t workspace_bytes, void* stream) {
  int rc = check_geom(geom, "conv2d_bwd_weight"); if (rc) return rc;
  SEEDHIP_REQUIRE(in && dy && dw && workspace, "conv2d_bwd_weight: null pointer");
  SEEDHIP_REQUIRE(in_dtype == kInF32 || in_dtype == kInU8Div255, "conv2d_bwd_weight: bad in_dtype %d", in_dtype);
  SEEDHIP_REQUIRE(workspace_bytes >= seedhip_conv2d_bwd_weight_workspace_bytes(geom),
                  "conv2d_bwd_weight: workspace too small");
  if (wgx_enabled() && in_dtype == kInF32 && al16(in) && al16(dy) && al16(workspace)) {
    // bf16x6 through transposing LDS reads (wgx.h): the second Atari conv and ImpalaDeep's 3 x 3 layers
    if (const int k = wgx::plan(geom)) {
      const int M = geom->kh * geom->kw * geom->cin, N = geom->cout;
      hipStream_t s = (hipStream_t)stream;
      float* pw = (float*)workspace;
      float* pb = pw + (size_t)wgx::grid_for(k, geom->n_img) * M * N;
      int slices = 0;
      rc = wgx::launch(k, geom, (const float*)in, in_relu, dy, pw, dbias ? pb : nullptr, &slices, s); if (rc) return rc;
      reduce_slices2(pw, (long long)M * N, dw, pb, N, dbias, slices, s);
      return check_launch("conv2d_bwd_weight(wgx)");
    }
  }
  if (in_dtype == kInF32 && conv_wgrad_gemm_ok(geom) && al16(in) && al16(dy) && al16(workspace)) {
    gemm::Params gp;
    gemm::conv_wgrad_setup(gp, geom);
    const gemm::Plan pl = conv_wgrad_plan(geom);
    const int M = gp.M, N = gp.N;
    hipStream_t s = (hipStream_t)stream;
    gp.A = (const float*)in; gp.a_relu = in_relu; gp.B = dy; gp.k_per_slice = pl.k_per_slice;
    float* pw = (float*)workspace;
    float* pb = pw + (size_t)pl.slices * M * N;
    gp.partial = pw; gp.partial_colsum = dbias ? pb : nullptr;
    gemm::launch<false, false, true, false, false>(gp, pl, s);
    reduce_slices2(pw, (long long)M * N, dw, pb, N, dbias, pl.slices, s);
    return check_launch("conv2d_bwd_weight(gather gemm)");
  }
  {
    // 3x3 stride-1 layers: input band + halo staged once per tile in LDS (halo_wgrad.h)
    const halo::WgradPlan pl = halo::plan_wgrad(geom);
    if (pl.ok && (((uintptr_t)in) & 15) == 0 && (((uintptr_t)dy) & 15) == 0)
      return halo::launch_wgrad(geom, pl, in, in_dtype, in_relu, dy, dw, dbias, workspace, (hipStream_t)stream);
  }
  if ((xg8::mode() & 4) && is_dense(geom) && in_dtype == kInF32 && al16(in) && al16(dy) && al16(dw) && al16(dbias) && workspace &&
      al16(workspace)) {
    const xg8::Plan xp = x8_wgrad_plan(geom);
    const int M = geom->cin, N = geom->cout, K = geom->n_img;
    if (xp.ok && workspace_bytes >= x8_ws(xp, M, N, true)) {
      hipStream_t s = (hipStream_t)stream;
      unsigned char* ws = (unsigned char*)workspace;
      float* pw = (float*)ws;
      float* pb = pw + (size_t)xp.slices * M * N;
      unsigned char* Bp = ws + xg8::al256((size_t)xp.slices * ((size_t)M * N + N) * sizeof(float));
      float* cs = dbias ? (float*)(Bp + xg8::al256(xp.b_planes)) : nullptr;
      xg8::launch_split<256>(dy, geom->ld_out, N, K, false, false, Bp, cs, s);        // B(k, n) = dy[k][n]
      gemm::Params gp;
      memset(&gp, 0, sizeof(gp));
      gp.A = (const float*)in; gp.lda = geom->ld_in; gp.a_relu = in_relu; gp.B = dy; gp.ldb = geom->ld_out;
      gp.M = M; gp.N = N; gp.K = K;
      gp.partial = xp.slices > 1 ? pw : dw;                      // one slice: the raw sums are the result
      gp.partial_colsum = dbias ? (xp.slices > 1 ? pb : dbias) : nullptr;
      if (xg8::launch<1>(gp, xp, nullptr, Bp, cs, s)) {
        if (xp.slices > 1) reduce_slices2(pw, (long long)M * N, dw, pb, N, dbias, xp.slices, s);
        return check_launch("conv2d_bwd_weight(dense, bf16x6 / 8 waves)");
      }
    }
  }
  if ((xg::mode() & 4) && is_dense(geom) && in_dtype == kInF32 && al16(in) && al16(dy) && al16(dw) && al16(dbias) && al16(workspace)) {
    const xg::Plan xp = x6_wgrad_plan(geom);
    if (xp.ok) {
      const int M = geom->cin, N = geom->cout, K = geom->n_img;
      hipStream_t s = (hipStream_t)stream;
      gemm::Params gp;
      memset(&gp, 0, sizeof(gp));
      gp.A = (const float*)in; gp.lda = geom->ld_in; gp.a_relu = in_relu; gp.B = dy; gp.ldb = geom->ld_out;
      gp.M = M; gp.N = N; gp.K = K;
      float* pw = (float*)workspace;
      float* pb = pw + (size_t)xp.slices * M * N;
      gp.partial = xp.slices > 1 ? pw : dw;                      // one slice: the raw sums are the result
      gp.partial_colsum = dbias ? (xp.slices > 1 ? pb : dbias) : nullptr;
      xg::launch<false, false>(gp, xp, s);
      if (xp.slices > 1) reduce_slices2(pw, (long long)M * N, dw, pb, N, dbias, xp.slices, s);
      return check_launch("conv2d_bwd_weight(dense, bf16x6)");
    }
  }
  if (is_dense(geom) && in_dtype == kInF32 && geom->ld_in % 4 == 0 && (((uintptr_t)in) & 15) == 0) {
    if (gemm_wgrad_ok(geom) && al16(dy) && al16(workspace)) {
      const int M = geom->cin, N = geom->cout, K = geom->n_img;
      const gemm::Plan pl = gemm::plan(M, N, K, 0.8);
      hipStream_t s = (hipStream_t)stream;
      gemm::Params gp;
      memset(&gp, 0, sizeof(gp));
      gp.A = (const float*)in; gp.lda = geom->ld_in; gp.a_relu = in_relu; gp.B = dy; gp.ldb = geom->ld_out;
      gp.M = M; gp.N = N; gp.K = K; gp.k_per_slice = pl.k_per_slice;
      float* pw = (float*)workspace;
      float* pb = pw + (size_t)pl.slices * M * N;
      gp.partial = pl.slices > 1 ? pw : dw;                      // one slice: the raw sums are the result
      gp.partial_colsum = dbias ? (pl.slices > 1 ? pb : dbias) : nullptr;
      gemm::launch<false, false>(gp, pl, s);
      if (pl.slices > 1) {
        reduce_slices2(pw, (long long)M * N, dw, pb, N, dbias, pl.slices, s);
      }
      return check_launch("conv2d_bwd_weight(dense, gemm)");
    }
    if (geom->cin % 4 == 0) {
    DenseWgrad d;
    d.in = (const float*)in; d.in_relu = in_relu; d.dy = dy;
    const int M = geom->cin, N = geom->cout;
    d.init(to_geom(geom), pick_k_per_slice(geom->n_img, tiles_for(M, N)));
    const int slices = d.slices();
    d.partial_w = (float*)workspace;
    d.partial_b = dbias ? (float*)workspace + (size_t)slices * M * N : nullptr;
    hipStream_t s = (hipStream_t)stream;
    launch_igemm_auto(d, slices, s);
    reduce_slices2(d.partial_w, (long long)M * N, dw, d.partial_b, N, dbias, slices, s);
    return check_launch("conv2d_bwd_weight(dense)");
    }
  }
  ConvWgrad p;
  p.in = in; p.in_dtype = in_dtype; p.in_relu = in_relu; p.dy = dy;
  const int M = geom->kh * geom->kw * geom->cin, N = geom->cout;
  const long long pixels = (long long)geom->n_img * geom->oh * geom->ow;
  p.init(to_geom(geom), pick_k_per_slice(pixels, tiles_for(M, N)));
  const int slices = p.slices();
  p.partial_w = (float*)workspace;
  p.partial_b = dbias ? (float*)workspace + (size_t)slices * M * N : nullptr;
  hipStream_t s = (hipStream_t)stream;
  launch_igemm_auto(p, slices, s);
  reduce_slices2(p.partial_w, (long long)M * N, dw, p.partial_b, N, dbias, slices, s);
  return check_launch("conv2d_bwd_weight");
}

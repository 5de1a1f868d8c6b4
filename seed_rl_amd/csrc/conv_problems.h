// Problem accessors for the implicit-GEMM core (igemm.h): how each dense op of the
// reference agents maps onto C[M,N] = sum_k A[m,k] B[k,n].
//
//   ConvFwd        Conv2D / Dense forward            dmlab/networks.py:31-60,84-89,116-118
//                                                    atari/networks.py:233-251
//   ConvDgrad      gradient wrt the layer input      (TF autodiff of the above)
//   ConvWgrad      gradient wrt kernel (+ bias)      (TF autodiff of the above)
//   ConvStackFwd / ConvStackWgrad
//                  first Atari conv fused with learner-side frame stacking
//                  (atari/networks.py:57-173,330): reads the uint8 frames
//                  directly; the fp32 [T,B,84,84,4] tensor is never built.
//
// Tensors are NHWC fp32 (TF default), kernels [kh,kw,cin,cout] row-major ==
// a [K, Cout] matrix, Dense == 1x1 conv on a 1x1 image.  All functions are
// HOST+DEVICE: tests/host/emul.cpp runs them on the CPU against the oracle.
#pragma once
#include "igemm.h"

namespace seedhip {

struct ConvGeom {
  int n_img, ih, iw, cin, oh, ow, kh, kw, stride, pad_t, pad_l, cout;
  int ld_in;    // input pixel stride (elements), >= cin
  int ld_out;   // output pixel stride (elements), >= cout
};

enum InDtype { kInF32 = 0, kInU8Div255 = 1 };

SH_HD float4 ld4(const float* p) {
#if defined(__HIP_DEVICE_COMPILE__)
  return *reinterpret_cast<const float4*>(p);
#else
  return make_float4(p[0], p[1], p[2], p[3]);
#endif
}
SH_HD float4 f4_zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
SH_HD float4 f4_relu(float4 v) {
  return make_float4(v.x < 0.f ? 0.f : v.x, v.y < 0.f ? 0.f : v.y, v.z < 0.f ? 0.f : v.z, v.w < 0.f ? 0.f : v.w);
}
SH_HD void f4_set(float4& v, int i, float x) { if (i == 0) v.x = x; else if (i == 1) v.y = x; else if (i == 2) v.z = x; else v.w = x; }

// --------------------------------------------------------------------------- //
// Forward: out[m, co] = act( sum_k in[m-th patch][k] * W[k, co] + bias[co] (+ residual) )
// --------------------------------------------------------------------------- //
struct ConvFwd {
  static constexpr bool kAVecK = true, kBVecN = true, kColSumB = false;
  ConvGeom g;
  int M, N, K;
  const void* in; int in_dtype; int in_relu;
  const float* w; const float* bias;
  float* out; int out_relu; const float* residual;
  FastDiv d_ohow, d_ow, d_cin, d_kw;
  int vec_a, vec_b;

  void init(const ConvGeom& geom) {
    g = geom;
    M = g.n_img * g.oh * g.ow; N = g.cout; K = g.kh * g.kw * g.cin;
    d_ohow.init(g.oh * g.ow); d_ow.init(g.ow); d_cin.init(g.cin); d_kw.init(g.kw);
    vec_a = (in_dtype == kInF32) && (g.cin % 4 == 0) && (g.ld_in % 4 == 0);
    vec_b = (g.cout % 4 == 0);
  }
  SH_HD void k_range(int, int& k0, int& k1) const { k0 = 0; k1 = K; }

  struct ARow { long long base; int iy0, ix0, valid; };
  SH_HD ARow a_row(int m, int) const {
    ARow r; r.valid = m < M;
    uint32_t n, rem, oy, ox;
    d_ohow.divmod((uint32_t)(r.valid ? m : 0), n, rem);
    d_ow.divmod(rem, oy, ox);
    r.iy0 = (int)oy * g.stride - g.pad_t; r.ix0 = (int)ox * g.stride - g.pad_l;
    r.base = (long long)n * g.ih * g.iw * g.ld_in;
    return r;
  }
  SH_HD float load_a1(const ARow& r, int k) const {
    if (k >= K) return 0.f;
    uint32_t tap, c, ky, kx;
    d_cin.divmod((uint32_t)k, tap, c);
    d_kw.divmod(tap, ky, kx);
    const int iy = r.iy0 + (int)ky, ix = r.ix0 + (int)kx;
    if (iy < 0 || iy >= g.ih || ix < 0 || ix >= g.iw) return 0.f;
    const long long off = r.base + ((long long)iy * g.iw + ix) * g.ld_in + c;
    float v;
    if (in_dtype == kInU8Div255) v = (float)((const uint8_t*)in)[off] / 255.0f;
    else v = ((const float*)in)[off];
    return relu_if(v, in_relu);
  }
  SH_HD float4 load_a(const ARow& r, int k, int) const {
    if (!r.valid) return f4_zero();
    if (vec_a) {
      uint32_t tap, c, ky, kx;
      d_cin.divmod((uint32_t)k, tap, c);
      d_kw.divmod(tap, ky, kx);
      const int iy = r.iy0 + (int)ky, ix = r.ix0 + (int)kx;
      if (iy < 0 || iy >= g.ih || ix < 0 || ix >= g.iw) return f4_zero();
      float4 v = ld4((const float*)in + r.base + ((long long)iy * g.iw + ix) * g.ld_in + c);
      return in_relu ? f4_relu(v) : v;
    }
    return make_float4(load_a1(r, k), load_a1(r, k + 1), load_a1(r, k + 2), load_a1(r, k + 3));
  }
  struct BCol { int n; };
  SH_HD BCol b_col(int n, int) const { return BCol{n}; }
  SH_HD float4 load_b(const BCol& c, int k, int) const {
    if (c.n >= N) return f4_zero();
    const float* p = w + (long long)k * g.cout + c.n;
    if (vec_b) return ld4(p);
    return make_float4(p[0], c.n + 1 < N ? p[1] : 0.f, c.n + 2 < N ? p[2] : 0.f, c.n + 3 < N ? p[3] : 0.f);
  }
  SH_HD void store(int m, int n, float v, int) const {
    if (bias) v += bias[n];
    const long long o = (long long)m * g.ld_out + n;
    if (residual) v += residual[o];
    out[o] = relu_if(v, out_relu);
  }
  SH_HD void store_colsum(int, float, int) const {}
};

// --------------------------------------------------------------------------- //
// Data gradient: dX[n,iy,ix,ci] = sum_{ky,kx,co} dY[n,oy,ox,co] W[ky,kx,ci,co],
// oy = (iy + pad_t - ky) / stride.  Slice z = stride-parity class (py,px): only
// taps ky = py + stride*j contribute, so strided convs waste no MFMA work on
// structural zeros.  Epilogue: optional ReLU mask (mask[idx] > 0) and optional
// accumulate (residual skip path).
// --------------------------------------------------------------------------- //
struct ConvDgrad {
  static constexpr bool kAVecK = true, kBVecN = false, kColSumB = false;
  ConvGeom g;
  int M, N, K;          // M = n_img*QH*QW rows per class, N = cin, K = JH*JW*cout
  int QH, QW, JH, JW;
  const float* dy; const float* w;
  float* dx; const float* mask; const float* add;
  FastDiv d_q, d_qw, d_cout, d_jw, d_s;
  int vec;

  void init(const ConvGeom& geom) {
    g = geom;
    QH = (g.ih - 1 + g.pad_t) / g.stride + 1; QW = (g.iw - 1 + g.pad_l) / g.stride + 1;
    JH = (g.kh + g.stride - 1) / g.stride; JW = (g.kw + g.stride - 1) / g.stride;
    M = g.n_img * QH * QW; N = g.cin; K = JH * JW * g.cout;
    d_q.init(QH * QW); d_qw.init(QW); d_cout.init(g.cout); d_jw.init(JW); d_s.init(g.stride);
    vec = (g.cout % 4 == 0) && (g.ld_out % 4 == 0);
  }
  int slices() const { return g.stride * g.stride; }
  SH_HD void k_range(int, int& k0, int& k1) const { k0 = 0; k1 = K; }

  struct ARow { long long base; int qy, qx, valid; };
  SH_HD ARow a_row(int m, int) const {
    ARow r; r.valid = m < M;
    uint32_t n, rem, qy, qx;
    d_q.divmod((uint32_t)(r.valid ? m : 0), n, rem);
    d_qw.divmod(rem, qy, qx);
    r.qy = (int)qy; r.qx = (int)qx;
    r.base = (long long)n * g.oh * g.ow * g.ld_out;
    return r;
  }
  // tap decode for reduction index k (multiple of 4 in the vector path)
  SH_HD bool tap(int k, int z, int& ky, int& kx, int& j, int& i, int& co) const {
    uint32_t t, c, jj, ii, py, px;
    d_cout.divmod((uint32_t)k, t, c);
    d_jw.divmod(t, jj, ii);
    d_s.divmod((uint32_t)z, py, px);
    ky = (int)py + g.stride * (int)jj; kx = (int)px + g.stride * (int)ii;
    j = (int)jj; i = (int)ii; co = (int)c;
    return ky < g.kh && kx < g.kw;
  }
  SH_HD float load_a1(const ARow& r, int k, int z) const {
    if (k >= K) return 0.f;
    int ky, kx, j, i, co;
    if (!tap(k, z, ky, kx, j, i, co)) return 0.f;
    const int oy = r.qy - j, ox = r.qx - i;
    if (oy < 0 || oy >= g.oh || ox < 0 || ox >= g.ow) return 0.f;
    return dy[r.base + ((long long)oy * g.ow + ox) * g.ld_out + co];
  }
  SH_HD float4 load_a(const ARow& r, int k, int z) const {
    if (!r.valid) return f4_zero();
    if (vec) {
      int ky, kx, j, i, co;
      if (!tap(k, z, ky, kx, j, i, co)) return f4_zero();
      const int oy = r.qy - j, ox = r.qx - i;
      if (oy < 0 || oy >= g.oh || ox < 0 || ox >= g.ow) return f4_zero();
      return ld4(dy + r.base + ((long long)oy * g.ow + ox) * g.ld_out + co);
    }
    return make_float4(load_a1(r, k, z), load_a1(r, k + 1, z), load_a1(r, k + 2, z), load_a1(r, k + 3, z));
  }
  struct BCol { int n; };
  SH_HD BCol b_col(int n, int) const { return BCol{n}; }
  SH_HD float load_b1(const BCol& c, int k, int z) const {
    if (k >= K || c.n >= N) return 0.f;
    int ky, kx, j, i, co;
    if (!tap(k, z, ky, kx, j, i, co)) return 0.f;
    return w[((long long)(ky * g.kw + kx) * g.cin + c.n) * g.cout + co];
  }
  SH_HD float4 load_b(const BCol& c, int k, int z) const {   // (k..k+3, n): contiguous along co
    if (c.n >= N) return f4_zero();
    if (g.cout % 4 == 0) {
      int ky, kx, j, i, co;
      if (!tap(k, z, ky, kx, j, i, co)) return f4_zero();
      return ld4(w + ((long long)(ky * g.kw + kx) * g.cin + c.n) * g.cout + co);
    }
    return make_float4(load_b1(c, k, z), load_b1(c, k + 1, z), load_b1(c, k + 2, z), load_b1(c, k + 3, z));
  }
  SH_HD void store(int m, int n, float v, int z) const {
    uint32_t ni, rem, qy, qx, py, px;
    d_q.divmod((uint32_t)m, ni, rem);
    d_qw.divmod(rem, qy, qx);
    d_s.divmod((uint32_t)z, py, px);
    const int iy = (int)qy * g.stride + (int)py - g.pad_t, ix = (int)qx * g.stride + (int)px - g.pad_l;
    if (iy < 0 || iy >= g.ih || ix < 0 || ix >= g.iw) return;
    const long long o = (((long long)ni * g.ih + iy) * g.iw + ix) * g.ld_in + n;
    if (mask && !(mask[o] > 0.f)) v = 0.f;
    if (add) v += add[o];
    dx[o] = v;
  }
  SH_HD void store_colsum(int, float, int) const {}
};

// --------------------------------------------------------------------------- //
// Weight gradient: dW[i, co] = sum_pix im2col[pix, i] * dY[pix, co]; the bias
// gradient is the column sum of dY, accumulated for free while the dY tile is
// staged (kColSumB).  Slice z = split over output pixels; partial results go to
// a workspace [slices][K*cout (+cout)] reduced by reduce_slices (deterministic).
// --------------------------------------------------------------------------- //
struct ConvWgrad {
  static constexpr bool kAVecK = false, kBVecN = true, kColSumB = true;
  ConvGeom g;
  int M, N, K;          // M = kh*kw*cin (rows of dW), N = cout, K = n_img*oh*ow (pixels)
  int k_per_slice;
  const void* in; int in_dtype; int in_relu;
  const float* dy;
  float* partial_w;     // [slices][M*N]
  float* partial_b;     // [slices][N]
  FastDiv d_ohow, d_ow, d_cin, d_kw;
  int vec_a, vec_b;

  void init(const ConvGeom& geom, int k_per_slice_) {
    g = geom;
    M = g.kh * g.kw * g.cin; N = g.cout; K = g.n_img * g.oh * g.ow;
    k_per_slice = k_per_slice_;
    d_ohow.init(g.oh * g.ow); d_ow.init(g.ow); d_cin.init(g.cin); d_kw.init(g.kw);
    vec_a = (in_dtype == kInF32) && (g.cin % 4 == 0) && (g.ld_in % 4 == 0);
    vec_b = (g.cout % 4 == 0) && (g.ld_out % 4 == 0);
  }
  int slices() const { return (K + k_per_slice - 1) / k_per_slice; }
  SH_HD void k_range(int z, int& k0, int& k1) const {
    k0 = z * k_per_slice; k1 = k0 + k_per_slice; if (k1 > K) k1 = K;
  }
  struct ARow { int ky[4], kx[4], c[4], valid[4]; };
  SH_HD ARow a_row(int m, int) const {     // 4 consecutive im2col columns m..m+3
    ARow r;
    for (int q = 0; q < 4; ++q) {
      const int i = m + q;
      r.valid[q] = i < M;
      uint32_t tap, c, ky, kx;
      d_cin.divmod((uint32_t)(r.valid[q] ? i : 0), tap, c);
      d_kw.divmod(tap, ky, kx);
      r.ky[q] = (int)ky; r.kx[q] = (int)kx; r.c[q] = (int)c;
    }
    return r;
  }
  SH_HD float4 load_a(const ARow& r, int k, int) const {    // (m..m+3, pixel k)
    uint32_t n, rem, oy, ox;
    d_ohow.divmod((uint32_t)k, n, rem);
    d_ow.divmod(rem, oy, ox);
    const int iy0 = (int)oy * g.stride - g.pad_t, ix0 = (int)ox * g.stride - g.pad_l;
    const long long base = (long long)n * g.ih * g.iw * g.ld_in;
    if (vec_a) {                                   // same tap for all 4 (cin % 4 == 0)
      if (!r.valid[0]) return f4_zero();
      const int iy = iy0 + r.ky[0], ix = ix0 + r.kx[0];
      if (iy < 0 || iy >= g.ih || ix < 0 || ix >= g.iw) return f4_zero();
      float4 v = ld4((const float*)in + base + ((long long)iy * g.iw + ix) * g.ld_in + r.c[0]);
      return in_relu ? f4_relu(v) : v;
    }
    float4 v = f4_zero();
    for (int q = 0; q < 4; ++q) {
      if (!r.valid[q]) continue;
      const int iy = iy0 + r.ky[q], ix = ix0 + r.kx[q];
      if (iy < 0 || iy >= g.ih || ix < 0 || ix >= g.iw) continue;
      const long long off = base + ((long long)iy * g.iw + ix) * g.ld_in + r.c[q];
      float x;
      if (in_dtype == kInU8Div255) x = (float)((const uint8_t*)in)[off] / 255.0f;
      else x = ((const float*)in)[off];
      f4_set(v, q, relu_if(x, in_relu));
    }
    return v;
  }
  struct BCol { int n; };
  SH_HD BCol b_col(int n, int) const { return BCol{n}; }
  SH_HD float4 load_b(const BCol& c, int k, int) const {    // (pixel k, n..n+3)
    if (c.n >= N) return f4_zero();
    const float* p = dy + (long long)k * g.ld_out + c.n;
    if (vec_b) return ld4(p);
    return make_float4(p[0], c.n + 1 < N ? p[1] : 0.f, c.n + 2 < N ? p[2] : 0.f, c.n + 3 < N ? p[3] : 0.f);
  }
  SH_HD void store(int m, int n, float v, int z) const { partial_w[((long long)z * M + m) * N + n] = v; }
  SH_HD void store_colsum(int n, float v, int z) const { if (partial_b) partial_b[(long long)z * N + n] = v; }
};


// --------------------------------------------------------------------------- //
// Dense layers (1x1 kernel on a 1x1 image = plain GEMM): the same three problems with the
// im2col index arithmetic removed -- the generic accessors spend ~4 VALU instructions per MFMA
// on div/mod and bounds checks, which a single wave per SIMD cannot hide.
//   y[r, n] = act( sum_k relu?(x[r*ld_in + k]) * W[k*cout + n] + bias[n] (+ residual) )
// Requires cin % 4 == 0, ld_in % 4 == 0 (A rows are read as float4).
// --------------------------------------------------------------------------- //
struct DenseFwd {
  static constexpr bool kAVecK = true, kBVecN = true, kColSumB = false;
  int M, N, K, ld_in, ld_out, cout, in_relu, out_relu, vec_b;
  int k_per_slice;          // split-K: slice z reduces [z*k_per_slice, ...) into partial[z] (epilogue in dense_epilogue)
  float* partial;           // [slices][M*N] or null (single slice, fused epilogue)
  const float* in; const float* w; const float* bias; float* out; const float* residual;
  void init(const ConvGeom& g) {
    M = g.n_img; N = g.cout; K = g.cin; ld_in = g.ld_in; ld_out = g.ld_out; cout = g.cout;
    vec_b = (g.cout % 4 == 0);
    k_per_slice = K; partial = nullptr;
  }
  SH_HD void k_range(int z, int& k0, int& k1) const { k0 = z * k_per_slice; k1 = k0 + k_per_slice; if (k1 > K) k1 = K; }
  struct ARow { const float* p; };
  SH_HD ARow a_row(int m, int) const { return ARow{m < M ? in + (long long)m * ld_in : nullptr}; }
  SH_HD float4 load_a(const ARow& r, int k, int) const {
    if (!r.p) return f4_zero();
    const float4 v = ld4(r.p + k);
    return in_relu ? f4_relu(v) : v;
  }
  struct BCol { int n; };
  SH_HD BCol b_col(int n, int) const { return BCol{n}; }
  SH_HD float4 load_b(const BCol& c, int k, int) const {
    if (c.n >= N) return f4_zero();
    const float* p = w + (long long)k * cout + c.n;
    if (vec_b) return ld4(p);
    return make_float4(p[0], c.n + 1 < N ? p[1] : 0.f, c.n + 2 < N ? p[2] : 0.f, c.n + 3 < N ? p[3] : 0.f);
  }
  SH_HD void store(int m, int n, float v, int z) const {
    if (partial) { partial[((long long)z * M + m) * N + n] = v; return; }
    if (bias) v += bias[n];
    const long long o = (long long)m * ld_out + n;
    if (residual) v += residual[o];
    out[o] = relu_if(v, out_relu);
  }
  SH_HD void store_colsum(int, float, int) const {}
};

// dx[r, ci] = sum_co dy[r*ld_out + co] * W[ci*cout + co]; then mask / accumulate.  cout % 4 == 0, ld_out % 4 == 0.
struct DenseDgrad {
  static constexpr bool kAVecK = true, kBVecN = false, kColSumB = false;
  int M, N, K, ld_in, ld_out, cout;
  int k_per_slice; float* partial;
  const float* dy; const float* w; float* dx; const float* mask; const float* add;
  void init(const ConvGeom& g) {
    M = g.n_img; N = g.cin; K = g.cout; ld_in = g.ld_in; ld_out = g.ld_out; cout = g.cout;
    k_per_slice = K; partial = nullptr;
  }
  SH_HD void k_range(int z, int& k0, int& k1) const { k0 = z * k_per_slice; k1 = k0 + k_per_slice; if (k1 > K) k1 = K; }
  struct ARow { const float* p; };
  SH_HD ARow a_row(int m, int) const { return ARow{m < M ? dy + (long long)m * ld_out : nullptr}; }
  SH_HD float4 load_a(const ARow& r, int k, int) const { return r.p ? ld4(r.p + k) : f4_zero(); }
  struct BCol { const float* p; };
  SH_HD BCol b_col(int n, int) const { return BCol{n < N ? w + (long long)n * cout : nullptr}; }
  SH_HD float4 load_b(const BCol& c, int k, int) const { return c.p ? ld4(c.p + k) : f4_zero(); }
  SH_HD void store(int m, int n, float v, int z) const {
    if (partial) { partial[((long long)z * M + m) * N + n] = v; return; }
    const long long o = (long long)m * ld_in + n;
    if (mask && !(mask[o] > 0.f)) v = 0.f;
    if (add) v += add[o];
    dx[o] = v;
  }
  SH_HD void store_colsum(int, float, int) const {}
};

// dW[ci, co] = sum_r relu?(x[r*ld_in + ci]) * dy[r*ld_out + co]; db = column sums of dy.  Split over rows.
struct DenseWgrad {
  static constexpr bool kAVecK = false, kBVecN = true, kColSumB = true;
  int M, N, K, ld_in, ld_out, k_per_slice, in_relu, vec_b;
  const float* in; const float* dy; float* partial_w; float* partial_b;
  void init(const ConvGeom& g, int k_per_slice_) {
    M = g.cin; N = g.cout; K = g.n_img; ld_in = g.ld_in; ld_out = g.ld_out; k_per_slice = k_per_slice_;
    vec_b = (g.cout % 4 == 0) && (g.ld_out % 4 == 0);
  }
  int slices() const { return (K + k_per_slice - 1) / k_per_slice; }
  SH_HD void k_range(int z, int& k0, int& k1) const {
    k0 = z * k_per_slice; k1 = k0 + k_per_slice; if (k1 > K) k1 = K;
  }
  struct ARow { int m; };
  SH_HD ARow a_row(int m, int) const { return ARow{m}; }
  SH_HD float4 load_a(const ARow& r, int k, int) const {      // (m..m+3, row k); M % 4 == 0
    if (r.m >= M) return f4_zero();
    const float4 v = ld4(in + (long long)k * ld_in + r.m);
    return in_relu ? f4_relu(v) : v;
  }
  struct BCol { int n; };
  SH_HD BCol b_col(int n, int) const { return BCol{n}; }
  SH_HD float4 load_b(const BCol& c, int k, int) const {
    if (c.n >= N) return f4_zero();
    const float* p = dy + (long long)k * ld_out + c.n;
    if (vec_b) return ld4(p);
    return make_float4(p[0], c.n + 1 < N ? p[1] : 0.f, c.n + 2 < N ? p[2] : 0.f, c.n + 3 < N ? p[3] : 0.f);
  }
  SH_HD void store(int m, int n, float v, int z) const { partial_w[((long long)z * M + m) * N + n] = v; }
  SH_HD void store_colsum(int n, float v, int z) const { if (partial_b) partial_b[(long long)z * N + n] = v; }
};

// --------------------------------------------------------------------------- //
// Frame-stacked first conv (Atari): image n = t*B + b of an unroll; stack channel c
// (0 = newest) of step t is uint8 frame (t - c) of frames_ext (see frames.hip),
// valid iff c < nvalid[t,b]; value = u8 / 255.  K is ordered (c, ky, kx) so that
// 4 consecutive k are 4 horizontally adjacent pixels of ONE frame = one 32-bit
// load; the weight accessor applies the matching permutation of the
// [kh,kw,4,cout] kernel.  VALID padding only (atari/networks.py:234-239).
// --------------------------------------------------------------------------- //
struct StackGeom { int T, B, ih, iw, oh, ow, kh, kw, stride, cout, ld_out; };

SH_HD float4 u8x4_div255(const uint8_t* p, int aligned) {
#if defined(__HIP_DEVICE_COMPILE__)
  if (aligned) {
    const uint32_t u = *reinterpret_cast<const uint32_t*>(p);
    return make_float4((float)(u & 0xFF) / 255.0f, (float)((u >> 8) & 0xFF) / 255.0f,
                       (float)((u >> 16) & 0xFF) / 255.0f, (float)(u >> 24) / 255.0f);
  }
#else
  (void)aligned;
#endif
  return make_float4((float)p[0] / 255.0f, (float)p[1] / 255.0f, (float)p[2] / 255.0f, (float)p[3] / 255.0f);
}

struct ConvStackFwd {
  static constexpr bool kAVecK = true, kBVecN = true, kColSumB = false;
  StackGeom g;
  int M, N, K;
  const uint8_t* frames_ext; const uint8_t* nvalid;
  const float* w; const float* bias; float* out; int out_relu;
  FastDiv d_ohow, d_ow, d_khkw, d_kw;
  long long hw, bhw;
  int aligned, vec_b;

  void init(const StackGeom& geom) {
    g = geom;
    M = g.T * g.B * g.oh * g.ow; N = g.cout; K = 4 * g.kh * g.kw;
    hw = (long long)g.ih * g.iw; bhw = hw * g.B;
    d_ohow.init(g.oh * g.ow); d_ow.init(g.ow); d_khkw.init(g.kh * g.kw); d_kw.init(g.kw);
    aligned = (g.iw % 4 == 0) && (g.stride % 4 == 0) && (g.kw % 4 == 0) && ((((uintptr_t)frames_ext) & 3) == 0);
    vec_b = (g.cout % 4 == 0);
  }
  SH_HD void k_range(int, int& k0, int& k1) const { k0 = 0; k1 = K; }
  struct ARow { long long base; int nv, valid; };
  SH_HD ARow a_row(int m, int) const {
    ARow r; r.valid = m < M;
    uint32_t n, rem, oy, ox;
    d_ohow.divmod((uint32_t)(r.valid ? m : 0), n, rem);
    d_ow.divmod(rem, oy, ox);
    r.nv = nvalid[n];
    r.base = ((long long)n + 3LL * g.B) * hw + (long long)oy * g.stride * g.iw + ox * g.stride;
    return r;
  }
  SH_HD float4 load_a(const ARow& r, int k, int) const {
    if (!r.valid) return f4_zero();
    uint32_t c, rem, ky, kx;
    d_khkw.divmod((uint32_t)k, c, rem);
    d_kw.divmod(rem, ky, kx);
    if ((int)c >= r.nv) return f4_zero();
    const uint8_t* p = frames_ext + r.base - (long long)c * bhw + (long long)ky * g.iw + kx;
    if (g.kw % 4 == 0) return u8x4_div255(p, aligned);
    float4 v = f4_zero();                        // generic: element-wise (k+q may wrap rows / channels)
    for (int q = 0; q < 4; ++q) {
      const int kq = k + q;
      if (kq >= K) break;
      uint32_t c2, rem2, ky2, kx2;
      d_khkw.divmod((uint32_t)kq, c2, rem2);
      d_kw.divmod(rem2, ky2, kx2);
      if ((int)c2 >= r.nv) continue;
      f4_set(v, q, (float)frames_ext[r.base - (long long)c2 * bhw + (long long)ky2 * g.iw + kx2] / 255.0f);
    }
    return v;
  }
  SH_HD long long w_row(int k) const {            // (c,ky,kx) -> row of the [kh,kw,4,cout] kernel
    uint32_t c, rem;
    d_khkw.divmod((uint32_t)k, c, rem);
    return ((long long)rem * 4 + c) * g.cout;
  }
  struct BCol { int n; };
  SH_HD BCol b_col(int n, int) const { return BCol{n}; }
  SH_HD float4 load_b(const BCol& c, int k, int) const {
    if (c.n >= N) return f4_zero();
    const float* p = w + w_row(k) + c.n;
    if (vec_b) return ld4(p);
    return make_float4(p[0], c.n + 1 < N ? p[1] : 0.f, c.n + 2 < N ? p[2] : 0.f, c.n + 3 < N ? p[3] : 0.f);
  }
  SH_HD void store(int m, int n, float v, int) const {
    if (bias) v += bias[n];
    out[(long long)m * g.ld_out + n] = relu_if(v, out_relu);
  }
  SH_HD void store_colsum(int, float, int) const {}
};

struct ConvStackWgrad {
  static constexpr bool kAVecK = false, kBVecN = true, kColSumB = true;
  StackGeom g;
  int M, N, K;          // M = 4*kh*kw (permuted (c,ky,kx) rows), N = cout, K = T*B*oh*ow pixels
  int k_per_slice;
  const uint8_t* frames_ext; const uint8_t* nvalid;
  const float* dy;
  float* partial_w; float* partial_b;
  FastDiv d_ohow, d_ow, d_khkw, d_kw;
  long long hw, bhw;
  int aligned, vec_b;

  void init(const StackGeom& geom, int k_per_slice_) {
    g = geom;
    M = 4 * g.kh * g.kw; N = g.cout; K = g.T * g.B * g.oh * g.ow;
    k_per_slice = k_per_slice_;
    hw = (long long)g.ih * g.iw; bhw = hw * g.B;
    d_ohow.init(g.oh * g.ow); d_ow.init(g.ow); d_khkw.init(g.kh * g.kw); d_kw.init(g.kw);
    aligned = (g.iw % 4 == 0) && (g.stride % 4 == 0) && (g.kw % 4 == 0) && ((((uintptr_t)frames_ext) & 3) == 0);
    vec_b = (g.cout % 4 == 0) && (g.ld_out % 4 == 0);
  }
  int slices() const { return (K + k_per_slice - 1) / k_per_slice; }
  SH_HD void k_range(int z, int& k0, int& k1) const {
    k0 = z * k_per_slice; k1 = k0 + k_per_slice; if (k1 > K) k1 = K;
  }
  struct ARow { int c[4], off[4], valid[4]; };
  SH_HD ARow a_row(int m, int) const {
    ARow r;
    for (int q = 0; q < 4; ++q) {
      const int i = m + q;
      r.valid[q] = i < M;
      uint32_t c, rem, ky, kx;
      d_khkw.divmod((uint32_t)(r.valid[q] ? i : 0), c, rem);
      d_kw.divmod(rem, ky, kx);
      r.c[q] = (int)c; r.off[q] = (int)ky * g.iw + (int)kx;
    }
    return r;
  }
  SH_HD float4 load_a(const ARow& r, int k, int) const {
    uint32_t n, rem, oy, ox;
    d_ohow.divmod((uint32_t)k, n, rem);
    d_ow.divmod(rem, oy, ox);
    const int nv = nvalid[n];
    const long long base = ((long long)n + 3LL * g.B) * hw + (long long)oy * g.stride * g.iw + ox * g.stride;
    if (g.kw % 4 == 0) {                           // 4 rows = 4 adjacent pixels of one frame
      if (!r.valid[0] || r.c[0] >= nv) return f4_zero();
      return u8x4_div255(frames_ext + base - (long long)r.c[0] * bhw + r.off[0], aligned);
    }
    float4 v = f4_zero();
    for (int q = 0; q < 4; ++q) {
      if (!r.valid[q] || r.c[q] >= nv) continue;
      f4_set(v, q, (float)frames_ext[base - (long long)r.c[q] * bhw + r.off[q]] / 255.0f);
    }
    return v;
  }
  struct BCol { int n; };
  SH_HD BCol b_col(int n, int) const { return BCol{n}; }
  SH_HD float4 load_b(const BCol& c, int k, int) const {
    if (c.n >= N) return f4_zero();
    const float* p = dy + (long long)k * g.ld_out + c.n;
    if (vec_b) return ld4(p);
    return make_float4(p[0], c.n + 1 < N ? p[1] : 0.f, c.n + 2 < N ? p[2] : 0.f, c.n + 3 < N ? p[3] : 0.f);
  }
  SH_HD void store(int m, int n, float v, int z) const {       // un-permute (c,ky,kx) -> (ky,kx,c)
    uint32_t c, rem;
    d_khkw.divmod((uint32_t)m, c, rem);
    partial_w[((long long)z * M + ((long long)rem * 4 + c)) * N + n] = v;
  }
  SH_HD void store_colsum(int n, float v, int z) const { if (partial_b) partial_b[(long long)z * N + n] = v; }
};

}  // namespace seedhip

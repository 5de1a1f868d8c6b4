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
// Two generations live here.  The kernels that run (convpool_fwd_mfma_kernel, convpool_bwd_mfma_kernel, further down) put
// the conv on the matrix pipes -- exact bf16x3 forward, fp32 MFMA weight gradient fed by a deterministic scatter of the
// pool gradient -- at 0.43 + 0.80 ms (T=20, B=256).  The round-1 kernels below them in history and above them in this
// file (SEEDHIP_CONVPOOL_MFMA=0, kept for A/B runs: 1.09 + 1.69 ms) are plain fp32 FMA on the vector ALU
// (v_pk_fma_f32: two channels per instruction; FMA chain in (ky, kx, c) order):
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
// ReLU mask of the pooled tensor as bytes: bit q of byte [pixel][quad] = pooled[pixel][4 quad + q] > 0 (wsx.h)
__device__ __forceinline__ uint8_t sign_nibble(const float4& v) {
  return (uint8_t)((v.x > 0.f ? 1 : 0) | (v.y > 0.f ? 2 : 0) | (v.z > 0.f ? 4 : 0) | (v.w > 0.f ? 8 : 0));
}

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
__device__ __forceinline__ void fetch_input(const Geom& g, const uint8_t* __restrict__ x, int n, int r0, InPrefetch& pre, int tid,
                                            int nrows = kInRows, int nthreads = 256) {
  const int wp = g.iw + 2;
#pragma unroll
  for (int u = 0; u < kInPre; ++u) {
    const int idx = tid + u * nthreads;
    uint32_t v = 0;
    if (idx < nrows * wp) {
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

// 3x3 / 2 max-pool (TF 'SAME' windows, first maximum wins) of one band out of the LDS conv buffer
// [quad][kConvRows][iw] x float4: item = (pooled pixel, channel quad).
__device__ __forceinline__ void pool_band(const Geom& g, const float* cbuf, int n, int i0, int cy0,
                                          float* __restrict__ pooled, uint8_t* __restrict__ argmax, uint8_t* __restrict__ bits,
                                          int tid) {
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
    if (bits) bits[o] = sign_nibble(best);
  }
}

__global__ void __launch_bounds__(256)
convpool_fwd_kernel(const Geom g, const uint8_t* __restrict__ x, const float* __restrict__ w,
                    const float* __restrict__ bias, float* __restrict__ pooled, uint8_t* __restrict__ argmax,
                         uint8_t* __restrict__ bits) {
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
    pool_band(g, cbuf, n, i0, cy0, pooled, argmax, bits, tid);
  }
}

// ---- forward on the bf16 matrix pipe, exact ("bf16x3", as stackconv.hip) ---------------------------------------- //
// uint8 pixels are exact in bf16 and an fp32 weight is the exact sum of three bf16 numbers (hi + mid + lo by
// truncation), so x * (w/255) = x*hi + x*mid + x*lo with every product exact and fp32 accumulation inside
// v_mfma_f32_16x16x32_bf16.  The VALU kernel above spends 0.57 of its 1.09 ms (T=20, B=256) in the 108 packed FMAs per
// pixel that each of its four waves issues; here a 16-pixel group costs six MFMAs (96 matrix-pipe cycles).
//   input band in LDS: bf16 [row][col + 1][4] (8 bytes per pixel, 4th channel zero), two zero columns on the right
//   D = A x B, rows = the 16 output channels (A = weights, in registers), columns = 16 consecutive pixels (B):
//   lane (j = pixel, kq) supplies 8 k = two horizontally adjacent pixels x 4 channels = 16 contiguous bytes:
//     MFMA 0: kq 0..2 -> window row kq, columns 0..1;  kq 3 -> row 0, column 2 (+ column 3 against zero weights)
//     MFMA 1: kq 0..1 -> rows 1..2, column 2 (+ 3);    kq 2..3 -> zero weights
//   D: lane (j, kq) holds channels 4kq..4kq+3 of pixel j -> one 16-byte store into the conv buffer of the pool phase.
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
union Frag8 { uint4 u; bf16x8_t v; };

// LDS geometry of the MFMA forward: bf16 input band [kInRows][xh_cols] x 8 bytes (column = ix + 2: the first map pixel
// of a row is 16-byte aligned; one zero pixel left of the map, up to three right of it) TWICE, the second copy shifted
// by one pixel -- a lane's operand is two adjacent pixels starting at any column, and ds_read_b128 (4 LDS cycles; the
// unaligned alternative ds_read2_b64 costs 16) needs 16-byte alignment: odd columns read the shifted copy; conv band
// [quad][kConvRows][iw + 2] x float4 with -inf guard columns and a quad stride of 1 mod 16 slots (the pool reads
// four quads of neighbouring pixels per 16 lanes: spreads them over the banks).
__host__ __device__ inline int xh_cols(int iw) { return (iw + 6) & ~1; }                          // even: rows stay 16-byte aligned
__host__ __device__ inline int xh_floats(int iw, int pbf) { return ((2 * pbf + 3) * xh_cols(iw) * 2 + 3) & ~3; }   // bf16x4 per pixel = 2 floats
__host__ __device__ inline int cq_stride(int iw, int pbf) { return (((2 * pbf + 1) * (iw + 2) + 14) & ~15) + 1; }   // float4 slots
__host__ __device__ inline int cbuf2_floats(int iw, int pbf) { return (4 * cq_stride(iw, pbf) + 64) * 4; }   // + 64 scratch slots
constexpr int kMaxGroups = 65;                          // 16-pixel groups per tile: 9 rows * 114 / 16
constexpr int kMaxPoolItems = 1024;                     // (pooled pixel, quad) items per tile: 4 rows * 57 * 4 <= 1024

// three pixel bytes (low 24 bits of v) -> bf16x4 (exact: float(n), n < 256, has a zero low half)
__device__ __forceinline__ uint2 bf16_pixel(uint32_t v) {
  const uint32_t f0 = __float_as_uint((float)(v & 255u)), f1 = __float_as_uint((float)((v >> 8) & 255u));
  const uint32_t f2b = __float_as_uint((float)((v >> 16) & 255u));
  return make_uint2((f0 >> 16) | (f1 & 0xFFFF0000u), f2b >> 16);
}

// ROW8: iw % 8 == 0 and x 8-byte aligned -- a thread stages 8 pixels of one row: 24 contiguous bytes, two loads
template <bool ROW8, int PBF, int NT>                  // PBF pooled rows per band, NT threads per workgroup
__global__ void __launch_bounds__(NT, NT / 128)         // (threads, waves per SIMD): two workgroups per CU whatever NT
convpool_fwd_mfma_kernel(const Geom g, const uint8_t* __restrict__ x, const float* __restrict__ w,
                         const float* __restrict__ bias, float* __restrict__ pooled, uint8_t* __restrict__ argmax,
                         uint8_t* __restrict__ bits) {
  constexpr int kConvRowsF = 2 * PBF + 1, kInRowsF = 2 * PBF + 3;
  constexpr int NWV = NT / 64, kIts = (kMaxGroups + 2 * NWV - 1) / (2 * NWV), kItems = kMaxPoolItems / NT;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  uint2* xh = reinterpret_cast<uint2*>(smem);
  float4* cbuf = reinterpret_cast<float4*>(smem + 2 * xh_floats(g.iw, PBF));
  const int xh_b = xh_floats(g.iw, PBF) / 2;              // pixel index of the shifted copy: xh[xh_b + c] = xh[c + 1]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kq = lane >> 4, j = lane & 15;
  const int wp = g.iw + 2, wph = xh_cols(g.iw), cw = g.iw + 2, sq = cq_stride(g.iw, PBF);
  const int ntiles = g.ntiles;                            // the host filled bands / ntiles / d_bands for PBF rows per band
  for (int idx = tid; idx < 2 * xh_b; idx += NT) xh[idx] = make_uint2(0u, 0u);
  for (int idx = tid; idx < 4 * sq; idx += NT) cbuf[idx] = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);

  // ---- W/255 -> three bf16 parts per MFMA, in registers (row = channel j) ----
  Frag8 wA[2][3];
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    uint32_t part[3][4];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = e & 3, px = e >> 2;
      int ky, kx;
      if (m == 0) { ky = kq < 3 ? kq : 0; kx = kq < 3 ? px : 2 + px; }
      else { ky = kq + 1; kx = kq < 2 ? 2 + px : 3; }
      float wv = 0.f;
      if (kx < 3 && c < CIN) wv = w[((ky * 3 + kx) * CIN + c) * COUT + j] / 255.0f;
      const uint32_t hi = __float_as_uint(wv) >> 16;                  // exact split by truncation
      const float r1 = wv - __uint_as_float(hi << 16);
      const uint32_t mid = __float_as_uint(r1) >> 16;
      const uint32_t lo = __float_as_uint(r1 - __uint_as_float(mid << 16)) >> 16;
      const uint32_t v[3] = {hi, mid, lo};
#pragma unroll
      for (int s3 = 0; s3 < 3; ++s3) {
        if (e & 1) part[s3][e >> 1] |= v[s3] << 16; else part[s3][e >> 1] = v[s3];
      }
    }
#pragma unroll
    for (int s3 = 0; s3 < 3; ++s3) wA[m][s3].u = make_uint4(part[s3][0], part[s3][1], part[s3][2], part[s3][3]);
  }
  f32x4_t bias4 = {0.f, 0.f, 0.f, 0.f};
  if (bias) {
    const float4 bv = *reinterpret_cast<const float4*>(bias + 4 * kq);
    bias4 = f32x4_t{bv.x, bv.y, bv.z, bv.w};
  }
  // window of conv pixel (r, xc): input rows r..r+2, LDS columns xc+1..xc+3; offsets in pixels from (r, xc + 1)
  const int off0 = kq < 3 ? kq * wph : 2;
  const int off1 = kq < 2 ? (kq + 1) * wph + 2 : 2;

  // ---- pool items of this thread: item = (pooled pixel of the band) * 4 + quad, the same in every tile ----
  int pool_src[kItems];                               // cbuf slot of the window's first tap
#pragma unroll
  for (int u = 0; u < kItems; ++u) {
    const int item = tid + u * NT, cq = item & 3;
    uint32_t pi, pj;
    g.d_pw.divmod((uint32_t)(item >> 2), pi, pj);
    pool_src[u] = cq * sq + 2 * (int)pi * cw + 2 * (int)pj - g.pl + 1;
  }

  // ---- conv groups of this lane (pixel j of 16-pixel group wave + 4 (2 it + h)), the same in every tile ----
  const int npix = kConvRowsF * g.iw, ngroups = (npix + 15) >> 4;
  const int nit = (ngroups - __builtin_amdgcn_readfirstlane(wave) + 2 * NWV - 1) / (2 * NWV);     // iterations of this wave (scalar)
  int grp_src[kIts][2];                                 // conv row << 20 | xh pixel index of the window's (row 0, column 0)
  int grp_dst[kIts][2];                                 // cbuf slot of this lane's 4 channels (a scratch slot: no such pixel)
#pragma unroll
  for (int it = 0; it < kIts; ++it) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int pix = (it * 2 * NWV + wave + NWV * h) * 16 + j;
      uint32_t ur, ux;
      g.d_iw.divmod((uint32_t)(pix < npix ? pix : npix - 1), ur, ux);
      const int c0 = (int)ux + 1;                           // first LDS column of the window; odd: the shifted copy
      grp_src[it][h] = (int)(ur << 20) | ((c0 & 1) ? xh_b + (int)ur * wph + c0 - 1 : (int)ur * wph + c0);
      grp_dst[it][h] = pix < npix ? kq * sq + (int)ur * cw + (int)ux + 1 : 4 * sq + lane;   // dead lanes: scratch slots
    }
  }

  // ---- input staging ----
  InPrefetch pre;                                         // !ROW8: 3 bytes + flag per pixel
  uint4 ra = make_uint4(0u, 0u, 0u, 0u); uint2 rb = make_uint2(0u, 0u);   // ROW8: 24 bytes = 8 pixels of one row
  const int row8 = ROW8 ? tid / (g.iw >> 3) : 0, col8 = ROW8 ? (tid - row8 * (g.iw >> 3)) * 8 : 0;
  auto fetch = [&](int t) {
    uint32_t n, band;
    g.d_bands.divmod((uint32_t)t, n, band);
    const int r0 = 2 * (int)band * PBF - g.pt - 1;
    if constexpr (ROW8) {
      ra = make_uint4(0u, 0u, 0u, 0u); rb = make_uint2(0u, 0u);
      const int iy = r0 + row8;
      if (row8 < kInRowsF && iy >= 0 && iy < g.ih) {
        const uint8_t* s = x + (((long long)n * g.ih + iy) * g.iw + col8) * CIN;
        const uint2 lo = *reinterpret_cast<const uint2*>(s), mid = *reinterpret_cast<const uint2*>(s + 8);
        ra = make_uint4(lo.x, lo.y, mid.x, mid.y);
        rb = *reinterpret_cast<const uint2*>(s + 16);
      }
    } else {
      fetch_input(g, x, (int)n, r0, pre, tid, kInRowsF, NT);
    }
  };
  auto commit = [&]() {
    if constexpr (ROW8) {
      if (row8 < kInRowsF) {
        const uint32_t d[6] = {ra.x, ra.y, ra.z, ra.w, rb.x, rb.y};
        uint2 px[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {                     // pixel q = bytes 3q .. 3q+2 of the 24
          const int b0 = 3 * q, wd = b0 >> 2, sh = (b0 & 3) * 8;
          const uint32_t v = sh == 0 ? d[wd] : (sh <= 8 ? d[wd] >> sh : __builtin_amdgcn_alignbyte(d[wd + 1 < 6 ? wd + 1 : 5], d[wd], sh >> 3));
          px[q] = bf16_pixel(v);
        }
        uint4* dst = reinterpret_cast<uint4*>(xh + row8 * wph + col8 + 2);
#pragma unroll
        for (int q = 0; q < 4; ++q) dst[q] = make_uint4(px[2 * q].x, px[2 * q].y, px[2 * q + 1].x, px[2 * q + 1].y);
        uint2* dsh = xh + xh_b + row8 * wph + col8 + 1;
#pragma unroll
        for (int q = 0; q < 8; ++q) dsh[q] = px[q];
      }
    } else {
#pragma unroll
      for (int u = 0; u < kInPre; ++u) {
        const int idx = tid + u * NT;
        if (idx < kInRowsF * wp) {
          uint32_t r, c;
          g.d_wp.divmod((uint32_t)idx, r, c);
          const uint2 pxl = bf16_pixel(pre.v[u]);         // out-of-map pixels hold 0 = the 'same' zero padding
          xh[r * wph + c + 1] = pxl;
          xh[xh_b + r * wph + c] = pxl;
        }
      }
    }
  };

  if ((int)blockIdx.x < ntiles) fetch(blockIdx.x);
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    uint32_t un, uband;
    g.d_bands.divmod((uint32_t)tile, un, uband);
    const int n = (int)un, band = (int)uband;
    const int i0 = band * PBF;
    const int cy0 = 2 * i0 - g.pt;
    __syncthreads();                                      // previous tile's pool phase is done with cbuf / conv with xh
    commit();
    __syncthreads();
    if (tile + (int)gridDim.x < ntiles) fetch(tile + gridDim.x);   // flies under this tile's conv + pool
    // ---- conv rows cy0 .. cy0 + kConvRows - 1: two independent 16-pixel groups per iteration, the operands of the next
    //      iteration in flight under the MFMAs of this one (two register sets, ping-pong).  The phase is issue-bound
    //      (2 waves per SIMD): every instruction beside the 12 MFMAs of an iteration counts.  Rows outside the map
    //      become -inf (the pool's 'SAME' windows ignore them): first / last band only ----
    {
      const bool edge = cy0 < 0 || cy0 + kConvRowsF > g.ih;
      Frag8 pa0[2], pa1[2], pb0[2], pb1[2];
      auto load_b = [&](int it, Frag8 (&x0)[2], Frag8 (&x1)[2]) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const uint2* bp = xh + (grp_src[it][h] & 0xFFFFF);
          x0[h].u = *reinterpret_cast<const uint4*>(bp + off0);
          x1[h].u = *reinterpret_cast<const uint4*>(bp + off1);
        }
      };
      auto compute = [&](int it, const Frag8 (&x0)[2], const Frag8 (&x1)[2]) {
        f32x4_t acc[2] = {bias4, bias4};
#pragma unroll
        for (int s3 = 0; s3 < 3; ++s3) {
#pragma unroll
          for (int h = 0; h < 2; ++h) acc[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wA[0][s3].v, x0[h].v, acc[h], 0, 0, 0);
#pragma unroll
          for (int h = 0; h < 2; ++h) acc[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wA[1][s3].v, x1[h].v, acc[h], 0, 0, 0);
        }
        if (edge) {
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            if ((unsigned)(cy0 + (grp_src[it][h] >> 20)) >= (unsigned)g.ih) acc[h] = f32x4_t{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
          }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) *reinterpret_cast<f32x4_t*>(cbuf + grp_dst[it][h]) = acc[h];   // no exec juggling
      };
      load_b(0, pa0, pa1);
#pragma unroll
      for (int it = 0; it < kIts; it += 2) {
        if (it >= nit) break;
        if (it + 1 < kIts && it + 1 < nit) load_b(it + 1, pb0, pb1);
        compute(it, pa0, pa1);
        if (it + 1 >= kIts || it + 1 >= nit) break;
        if (it + 2 < kIts && it + 2 < nit) load_b(it + 2, pa0, pa1);
        compute(it + 1, pb0, pb1);
      }
    }
    __syncthreads();
    // ---- 3x3 / 2 max-pool out of LDS, first maximum wins (v > best in (ky, kx) order): no border tests, the guard
    //      columns / rows hold -inf; the band's output is contiguous: item-th float4 / uchar4 of the band ----
    {
      const int rows = (i0 + PBF <= g.ph) ? PBF : g.ph - i0;
      const long long obase = ((long long)n * g.ph + i0) * g.pw * 4;
#pragma unroll
      for (int u = 0; u < kItems; ++u) {
        const int item = tid + u * NT;
        if (item < rows * g.pw * 4) {
          const float4* src = cbuf + pool_src[u];
          float4 best = src[0];
          int bi0 = 0, bi1 = 0, bi2 = 0, bi3 = 0;
#pragma unroll
          for (int code = 1; code < 9; ++code) {
            const float4 v = src[(code / 3) * cw + (code % 3)];
            if (v.x > best.x) { best.x = v.x; bi0 = code; }
            if (v.y > best.y) { best.y = v.y; bi1 = code; }
            if (v.z > best.z) { best.z = v.z; bi2 = code; }
            if (v.w > best.w) { best.w = v.w; bi3 = code; }
          }
          reinterpret_cast<float4*>(pooled)[obase + item] = best;
          reinterpret_cast<uchar4*>(argmax)[obase + item] = make_uchar4((uint8_t)bi0, (uint8_t)bi1, (uint8_t)bi2, (uint8_t)bi3);
          if (bits) bits[obase + item] = sign_nibble(best);
        }
      }
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

// ---- backward on the fp32 matrix pipe ------------------------------------------------------------------------------ //
// dW[k][co] = sum_pixels x[pixel; k] * G[pixel][co] / 255 with G = the pre-pool gradient.  The VALU kernel above
// GATHERS G per conv pixel (it examines up to 4 windows x 4 channels per pixel and quad: 2.25 checks for every pooled
// element that exists) and then issues 108 packed FMAs per pixel in each of its four waves.  Here:
//   * G is SCATTERED: every pooled (pixel, channel) adds its gradient to the ONE conv pixel its argmax names, into an
//     fp32 [conv pixel][16] tile in LDS.  Windows of equal (row, column) parity are disjoint, so four passes -- one per
//     parity class, a barrier in between -- need no atomics and add in a fixed order: deterministic;
//   * dW and db come from v_mfma_f32_16x16x4_f32: rows = the 27 (tap, channel) rows of dW (two tiles, the rows >= 27 read
//     the zero 4th channel), columns = 16 output channels, 4 pixels reduced per instruction; A = the input band in
//     LDS as fp32 pixel VALUES (0..255; the 1/255 is applied once to the sums), B = G;
//   * staging as the forward: 8 pixels per thread (ROW8), all per-thread coordinates decoded once.
constexpr int kOwnRows = 2 * PB;                          // conv rows a band owns
__host__ __device__ inline int iw_pad4(int iw) { return (iw + 3) & ~3; }
__host__ __device__ inline int xf_floats(int iw) { return kInRows * (iw_pad4(iw) + 2) * 4; }
__host__ __device__ inline int gt_floats(int iw) { return kOwnRows * iw_pad4(iw) * COUT; }

template <bool ROW8>
__global__ void __launch_bounds__(256)
convpool_bwd_mfma_kernel(const Geom g, const uint8_t* __restrict__ x, const float* __restrict__ dpooled,
                         const uint8_t* __restrict__ argmax, float* __restrict__ partial_w, float* __restrict__ partial_b) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* xf = smem;                                       // [kInRows][iwp + 2][4]: column = ix + 1
  float* gt = smem + xf_floats(g.iw);                     // [kOwnRows][iwp][16]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kq = lane >> 4, j = lane & 15;
  const int iwp = iw_pad4(g.iw), wpx = iwp + 2, wp = g.iw + 2;
  for (int idx = tid; idx < xf_floats(g.iw); idx += 256) xf[idx] = 0.f;

  // ---- this thread's pooled items, per parity class p = (row parity, column parity) and round: pooled row
  //      prow = 0..PB of the PB + 1 rows i0-1 .. i0+PB-1 that touch the band, column pcol, channel quad q ----
  constexpr int kRounds = 2;
  int it_goff[4][kRounds];                                // gt float offset of the window's tap (0, 0), channel 4q (may be < 0)
  int it_src[4][kRounds];                                 // float4 / uchar4 index inside the band's PB + 1 pooled rows, -1: none
  int it_r0[4][kRounds];                                  // conv row (relative to cy0) of the window's first row
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int pr = p >> 1, pc = p & 1;
    const int nrow = (PB + 1 - pr + 1) >> 1, ncol = (g.pw - pc + 1) >> 1;
#pragma unroll
    for (int rd = 0; rd < kRounds; ++rd) {
      const int it = tid + rd * 256;
      const int q = it & 3, rest = it >> 2;
      const int ri = ncol > 0 ? rest / ncol : 0, ci = rest - ri * ncol;
      const int prow = 2 * ri + pr, pcol = 2 * ci + pc;
      const bool ok = ncol > 0 && ri < nrow;
      it_src[p][rd] = ok ? (prow * g.pw + pcol) * 4 + q : -1;
      it_r0[p][rd] = 2 * prow - 2;
      it_goff[p][rd] = ((2 * prow - 2) * iwp + 2 * pcol - g.pl) * COUT + 4 * q;
    }
  }
  // ---- MFMA operand offsets ----
  int a_off[2];                                           // xf float offset of k-row 16 mt + j relative to the window's (0, 0) pixel
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const int k = 16 * mt + j;
    const int tap = k / 3, c = k - 3 * tap, ky = tap / 3, kx = tap - 3 * ky;
    a_off[mt] = k < 27 ? (ky * wpx + kx) * 4 + c : 3;     // rows 27..31: the always-zero 4th channel
  }
  typedef float f32x4_t __attribute__((ext_vector_type(4)));
  f32x4_t acc[2] = {f32x4_t{0.f, 0.f, 0.f, 0.f}, f32x4_t{0.f, 0.f, 0.f, 0.f}};
  float bsum = 0.f;

  // ---- prefetch: input bytes (as the forward) + the thread's pooled items ----
  InPrefetch pre;
  uint4 ra = make_uint4(0u, 0u, 0u, 0u); uint2 rb = make_uint2(0u, 0u);
  const int row8 = ROW8 ? tid / (g.iw >> 3) : 0, col8 = ROW8 ? (tid - row8 * (g.iw >> 3)) * 8 : 0;
  float4 pd[4][kRounds];
  uint32_t pa[4][kRounds];
  auto fetch = [&](int t) {
    uint32_t un, uband;
    g.d_bands.divmod((uint32_t)t, un, uband);
    const int n = (int)un, i0 = (int)uband * PB;
    const int r0 = 2 * i0 - g.pt - 1;
    if constexpr (ROW8) {
      ra = make_uint4(0u, 0u, 0u, 0u); rb = make_uint2(0u, 0u);
      const int iy = r0 + row8;
      if (row8 < kInRows && iy >= 0 && iy < g.ih) {
        const uint8_t* s = x + (((long long)n * g.ih + iy) * g.iw + col8) * CIN;
        const uint2 lo = *reinterpret_cast<const uint2*>(s), mid = *reinterpret_cast<const uint2*>(s + 8);
        ra = make_uint4(lo.x, lo.y, mid.x, mid.y);
        rb = *reinterpret_cast<const uint2*>(s + 16);
      }
    } else {
      fetch_input(g, x, n, r0, pre, tid);
    }
    // pooled rows i0-1 .. i0+PB-1; rows outside the map: no item (code 255 matches nothing)
    const long long pbase = ((long long)n * g.ph + i0 - 1) * g.pw * 4;
    const int lo_item = i0 == 0 ? g.pw * 4 : 0;                                  // first band: row i0-1 does not exist
    const int hi_item = (g.ph - (i0 - 1) < PB + 1 ? g.ph - (i0 - 1) : PB + 1) * g.pw * 4;
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
      for (int rd = 0; rd < kRounds; ++rd) {
        float4 d = make_float4(0.f, 0.f, 0.f, 0.f);
        uint32_t am = 0xFFFFFFFFu;
        const int si = it_src[p][rd];
        if (si >= lo_item && si < hi_item) {
          d = reinterpret_cast<const float4*>(dpooled)[pbase + si];
          am = reinterpret_cast<const uint32_t*>(argmax)[pbase + si];
        }
        pd[p][rd] = d; pa[p][rd] = am;
      }
  };
  auto commit = [&]() {
    if constexpr (ROW8) {
      if (row8 < kInRows) {
        const uint32_t d[6] = {ra.x, ra.y, ra.z, ra.w, rb.x, rb.y};
        float4* dst = reinterpret_cast<float4*>(xf) + row8 * wpx + col8 + 1;
#pragma unroll
        for (int q = 0; q < 8; ++q) {                     // pixel q = bytes 3q .. 3q+2 of the 24
          const int b0 = 3 * q, wd = b0 >> 2, sh = (b0 & 3) * 8;
          const uint32_t v = sh == 0 ? d[wd] : (sh <= 8 ? d[wd] >> sh : __builtin_amdgcn_alignbyte(d[wd + 1 < 6 ? wd + 1 : 5], d[wd], sh >> 3));
          dst[q] = make_float4((float)(v & 255u), (float)((v >> 8) & 255u), (float)((v >> 16) & 255u), 0.f);
        }
      }
    } else {
#pragma unroll
      for (int u = 0; u < kInPre; ++u) {
        const int idx = tid + u * 256;
        if (idx < kInRows * wp) {
          uint32_t r, c;
          g.d_wp.divmod((uint32_t)idx, r, c);
          const uint32_t v = pre.v[u];                    // out-of-map pixels hold 0 = the 'same' zero padding
          reinterpret_cast<float4*>(xf)[r * wpx + c] = make_float4((float)(v & 255u), (float)((v >> 8) & 255u), (float)((v >> 16) & 255u), 0.f);
        }
      }
    }
  };

  if ((int)blockIdx.x < g.ntiles) fetch(blockIdx.x);
  for (int tile = blockIdx.x; tile < g.ntiles; tile += gridDim.x) {
    uint32_t un, uband;
    g.d_bands.divmod((uint32_t)tile, un, uband);
    const int band = (int)uband;
    const int i0 = band * PB;
    const int cy0 = 2 * i0 - g.pt;                        // first conv row owned by the band
    int crow = kOwnRows;                                  // conv rows owned: the last band takes what is left
    if (band == g.bands - 1) crow = g.ih - cy0;
    const int rlo = cy0 < 0 ? -cy0 : 0;                   // owned rows are [rlo, crow) relative to cy0
    __syncthreads();                                      // previous tile's MFMA phase is done with xf / gt
    commit();
    for (int idx = tid; idx < kOwnRows * iwp * (COUT / 4); idx += 256) reinterpret_cast<float4*>(gt)[idx] = make_float4(0.f, 0.f, 0.f, 0.f);
    // the items leave the registers before the next tile's are requested
    float4 cd[4][kRounds]; uint32_t ca[4][kRounds];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
      for (int rd = 0; rd < kRounds; ++rd) { cd[p][rd] = pd[p][rd]; ca[p][rd] = pa[p][rd]; }
    __syncthreads();
    if (tile + (int)gridDim.x < g.ntiles) fetch(tile + gridDim.x);
    // ---- scatter: four parity classes, disjoint windows inside a class ----
#pragma unroll
    for (int p = 0; p < 4; ++p) {
#pragma unroll
      for (int rd = 0; rd < kRounds; ++rd) {
        const float dv[4] = {cd[p][rd].x, cd[p][rd].y, cd[p][rd].z, cd[p][rd].w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const uint32_t code = (ca[p][rd] >> (8 * c)) & 255u;
          const uint32_t ky = (code * 11u) >> 5, kx = code - 3u * ky;
          const int r = it_r0[p][rd] + (int)ky;
          if (code < 9u && (unsigned)(r - rlo) < (unsigned)(crow - rlo)) {
            float* dst = gt + it_goff[p][rd] + ((int)ky * iwp + (int)kx) * COUT + c;
            *dst += dv[c];
          }
        }
      }
      __syncthreads();
    }
    // ---- dW += X^T G, db += sum G: wave w takes the 4-pixel groups w, w+4, ... of every owned row ----
    for (int r = rlo; r < crow; ++r) {
      const float* xrow = xf + (r * wpx) * 4 + kq * 4;     // window (0, 0) of pixel xc = 4 xg + kq is xf pixel (r, xc)
      const float* grow = gt + (r * iwp + kq) * COUT + j;
      for (int xg = wave; xg < (iwp >> 2); xg += 4) {
        const float b = grow[xg * 4 * COUT];
        const float a0 = xrow[xg * 16 + a_off[0]], a1 = xrow[xg * 16 + a_off[1]];
        bsum += b;
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b, acc[1], 0, 0, 0);
      }
    }
  }

  // ---- four waves -> one slice per workgroup (fixed order), scaled by 1/255 ----
  __syncthreads();
  float* red = smem;                                      // [4 waves][32 rows][16] + [4][16]
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[(wave * 32 + 16 * mt + 4 * kq + r) * COUT + j] = acc[mt][r];
  {
    float sb = bsum;
    sb += __shfl_xor(sb, 16, 64);
    sb += __shfl_xor(sb, 32, 64);
    if (lane < 16) red[4 * 32 * COUT + wave * COUT + lane] = sb;
  }
  __syncthreads();
  for (int idx = tid; idx < 27 * COUT; idx += 256) {
    float v = red[idx];
    for (int w = 1; w < 4; ++w) v += red[w * 32 * COUT + idx];
    partial_w[(long long)blockIdx.x * 27 * COUT + idx] = v / 255.0f;
  }
  if (partial_b && tid < COUT) {
    float v = red[4 * 32 * COUT + tid];
    for (int w = 1; w < 4; ++w) v += red[4 * 32 * COUT + w * COUT + tid];
    partial_b[(long long)blockIdx.x * COUT + tid] = v;
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
  return seedhip_conv3x3_u8_pool_fwd_bits(x, n, ih, iw, cin, w, bias, cout, pooled, argmax, nullptr, stream);
}

// The same, also writing the sign of the pooled tensor as bytes [pixel][cout / 4] (pooled_bits may be NULL): the ReLU mask
// of the first residual block's data gradient (seedhip_conv2d_bwd_data_bits_add).
extern "C" int seedhip_conv3x3_u8_pool_fwd_bits(const uint8_t* x, int n, int ih, int iw, int cin, const float* w,
                                                const float* bias, int cout, float* pooled, uint8_t* argmax,
                                                uint8_t* pooled_bits, void* stream) {
  Geom g;
  int rc = make_geom(n, ih, iw, cin, cout, &g, "conv3x3_u8_pool_fwd"); if (rc) return rc;
  SEEDHIP_REQUIRE(x && w && pooled && argmax, "conv3x3_u8_pool_fwd: null pointer");
  SEEDHIP_REQUIRE(((((uintptr_t)w) | ((uintptr_t)bias) | ((uintptr_t)pooled)) & 15) == 0 && (((uintptr_t)argmax) & 3) == 0,
                  "conv3x3_u8_pool_fwd: w / bias / pooled must be 16-byte aligned, argmax 4-byte aligned");
  static const int use_mfma = getenv("SEEDHIP_CONVPOOL_MFMA") ? atoi(getenv("SEEDHIP_CONVPOOL_MFMA")) : 1;
  if (use_mfma && 4 * g.pw * 4 <= kMaxPoolItems && 9 * iw <= kMaxGroups * 16) {
    constexpr int nthr = 256;
    const bool row8 = iw % 8 == 0 && (2 * 4 + 3) * (iw / 8) <= 256 && (((uintptr_t)x) & 7) == 0;
    hipStream_t s = (hipStream_t)stream;
#define SEEDHIP_CPF(R8_, NT_)                                                                                     \
    if (row8 == R8_ && nthr == NT_) {                                                                             \
      const size_t lds = (size_t)(2 * xh_floats(iw, 4) + cbuf2_floats(iw, 4)) * sizeof(float);                    \
      int per_cu = (int)((160 * 1024) / (lds + 256)); if (per_cu > 2048 / NT_) per_cu = 2048 / NT_; if (per_cu < 1) per_cu = 1; \
      const int grid = g.ntiles < 256 * per_cu ? g.ntiles : 256 * per_cu;                                         \
      if (lds > 64 * 1024)                                                                                        \
        (void)hipFuncSetAttribute((const void*)convpool_fwd_mfma_kernel<R8_, 4, NT_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
      hipLaunchKernelGGL((convpool_fwd_mfma_kernel<R8_, 4, NT_>), dim3(grid), dim3(NT_), lds, s, g, x, w, bias, pooled, argmax, pooled_bits); \
      return seedhip::check_launch("convpool_fwd_mfma_kernel");                                                   \
    }
    SEEDHIP_CPF(true, 256) SEEDHIP_CPF(true, 512) SEEDHIP_CPF(false, 256) SEEDHIP_CPF(false, 512)
#undef SEEDHIP_CPF
  }
  const size_t lds = (size_t)(xin_floats(iw) + cbuf_floats(iw)) * sizeof(float);
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute((const void*)convpool_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(convpool_fwd_kernel, dim3(grid_for(g)), dim3(256), lds, (hipStream_t)stream, g, x, w, bias, pooled,
                     argmax, pooled_bits);
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
  static const int use_mfma = getenv("SEEDHIP_CONVPOOL_MFMA") ? atoi(getenv("SEEDHIP_CONVPOOL_MFMA")) : 1;
  const int ncol_max = (g.pw + 1) / 2, nrow_max = (PB + 2) / 2;
  if (use_mfma && nrow_max * ncol_max * 4 <= 2 * 256) {
    size_t lds = (size_t)(xf_floats(iw) + gt_floats(iw)) * sizeof(float);
    if (lds < (4 * 32 * COUT + 4 * COUT) * sizeof(float)) lds = (4 * 32 * COUT + 4 * COUT) * sizeof(float);   // the final cross-wave sum
    const bool row8 = iw % 8 == 0 && kInRows * (iw / 8) <= 256 && (((uintptr_t)x) & 7) == 0;
    hipStream_t s = (hipStream_t)stream;
    if (row8) {
      if (lds > 64 * 1024)
        (void)hipFuncSetAttribute((const void*)convpool_bwd_mfma_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL(convpool_bwd_mfma_kernel<true>, dim3(grid), dim3(256), lds, s, g, x, dpooled, argmax, pw, pb);
    } else {
      if (lds > 64 * 1024)
        (void)hipFuncSetAttribute((const void*)convpool_bwd_mfma_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL(convpool_bwd_mfma_kernel<false>, dim3(grid), dim3(256), lds, s, g, x, dpooled, argmax, pw, pb);
    }
    rc = seedhip::check_launch("convpool_bwd_mfma_kernel"); if (rc) return rc;
    seedhip::reduce_slices2(pw, 27LL * COUT, dw, pb, COUT, dbias, grid, s);
    return seedhip::check_launch("conv3x3_u8_pool_bwd");
  }
  const size_t lds = (size_t)(xin_floats(iw) + dyp_floats(g.pw)) * sizeof(float) + (size_t)(PB + 1) * g.pw * COUT;
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute((const void*)convpool_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(convpool_bwd_kernel, dim3(grid), dim3(256), lds, s, g, x, dpooled, argmax, pw, pb);
  rc = seedhip::check_launch("convpool_bwd_kernel"); if (rc) return rc;
  seedhip::reduce_slices2(pw, 27LL * COUT, dw, pb, COUT, dbias, grid, s);
  return seedhip::check_launch("conv3x3_u8_pool_bwd");
}

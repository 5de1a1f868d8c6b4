// First Atari conv (8x8 stride 4, VALID) fused with learner-side frame stacking
// and the /255 normalisation: forward and weight gradient, specialised for CDNA4.
//
// Replaces /root/reference/atari/networks.py:57-173 (stack_frames) + :330 (/255) +
// the first Conv2D of the torso (:234; IMPALA-paper shallow torso: Conv 8x8/4 x16) and
// the TF autodiff of that conv wrt its kernel/bias.
//
// Why a dedicated kernel: this layer is 45% of the learner step's flops, its GEMM is
// skinny (N = 16 output channels) and its A operand is an im2col gather over uint8
// frames.  The generic implicit-GEMM core spends its time on gather index math; here
//   * a workgroup owns ONE batch column b and walks the unroll in time; each of its 5 waves
//     keeps ITS 20-row band of the last 4 raw uint8 frames in a private LDS ring (6.7 KB):
//     frame bytes are read from HBM/L2 about once (1 B per pixel per frame is the algorithmic
//     minimum; bands overlap by 4 rows) and serve the 4 stack positions they appear in; the
//     next band is prefetched into registers while the current step computes, and the time
//     loop contains NO workgroup barrier (waves are autonomous);
//   * MFMA operands are built straight from the LDS bytes: one ds_read_b32 yields 4
//     horizontally adjacent pixels = 4 consecutive k of the reduction, converted with
//     v_cvt_f32_ubyte{0..3}; K is ordered so that a lane's (pixel, k-quad) address
//     is a fixed per-lane offset plus compile-time row offsets -- no div/mod in the loop;
//   * 1/255 is folded into the LDS copy of the weights (forward) / applied once to the
//     reduced dW (backward) instead of once per input element;
//   * cumulative-OR done masking (networks.py:131-157) = "channels c >= nvalid[t,b]
//     are zero" => those k-steps are skipped (fwd) or fed zero words (wgrad);
//   * forward computes out^T tiles (rows = channels, cols = pixels) so that each lane
//     ends up with 4 consecutive channels of one pixel: 16-byte fully coalesced stores.
// 5 waves per workgroup: 400 output pixels = 5 waves x 5 MFMA tiles (fwd) /
// 5 waves x 20 four-pixel groups (wgrad) -- perfectly balanced for 84x84 frames.
//
// The description above is the fp32-MFMA pair of kernels (v_mfma_f32_16x16x4_f32, 2*400*256*Cout flops per frame),
// kept for A/B runs (SEEDHIP_STACK_BF16=0).  The DEFAULT kernels further down evaluate the same fp32 arithmetic on
// the bf16 matrix pipe through an exact three-way operand split ("bf16x3": uint8 pixels are exact in bf16, an fp32
// weight / gradient is the exact sum of three bf16 numbers): forward with the same band-per-wave organisation and
// the ring held in bf16 -- since r6 on EIGHT waves over two batch columns (stackconv_fwd_w8_kernel: a five-wave
// workgroup puts two of its waves on one SIMD; the five-wave stackconv_fwd_bf16r_kernel stays for tensors above 2 GB
// and SEEDHIP_STACK_W8=0) --, weight gradient through transposing LDS reads with a (stack channel, k-step parity)
// wave organisation (stackconv_wgrad_tr_kernel), central inference's rows forward (stackconv_rows_w8_kernel).
// Requests that live across a loop iteration are asm statements with hand-counted waits in all three:
// tools/isa_inflight.py --cfg (tests/test_isa_structure.py) follows each of them through the compiled control flow.
#include "common.h"
#include <cstdlib>
#include <type_traits>
#include "conv_problems.h"
#include "conv_launch.h"
#include "../../include/seedhip.h"

namespace seedhip {
namespace stackconv {

typedef float f32x4_t __attribute__((ext_vector_type(4)));

typedef unsigned su32x4_t __attribute__((ext_vector_type(4)));
constexpr int kWaves = 5;
constexpr int kThreads = kWaves * 64;
constexpr int kMT = 5;         // MFMA tiles in flight per wave (forward)
constexpr int kWFloats = 64 * 64;   // one 16-channel slice of the kernel: [64 k-steps][64 lanes]

struct Params {
  const uint8_t* frames_ext;   // u8 [3+T1, B, fsz]
  const uint8_t* nvalid;       // u8 [T1, B]
  const float* w;              // [8,8,4,cout]
  const float* bias;           // [cout] or null
  float* out;                  // fwd: [T1*B, oh*ow, ld_out]
  const float* dy;             // wgrad: same layout as out
  float* partial_w;            // wgrad: [gridDim.x][256*cout]
  float* partial_b;            // wgrad: [gridDim.x][cout] or null
  int T1, B, ih, iw, oh, ow, cout, ld_out, out_relu;
  int fsz;                     // ih*iw bytes, multiple of 16, <= 2*kThreads*16
  int spc, items;              // steps per chunk; items = B * nchunks
  int buf32;                   // fwd (bf16x3 kernel): frames_ext and out are < 2 GB: the time loop addresses them as buffers (uniform 32-bit step offsets)
  unsigned char* relu_bits;    // fwd (bf16x3 kernel), optional: [T1*B*oh*ow, ld_out / 4] bytes, bit r of byte q = out[.., 4q + r] > 0
};

__device__ __forceinline__ float ubyte(uint32_t w, int q) { return (float)((w >> (8 * q)) & 0xFFu); }

// nvalid[idx] through the SCALAR data cache (constant address space, uniform address -> s_load_dword): a vector load
// here is waited for with s_waitcnt vmcnt(0) at the head of the time loop, and that counter also holds the previous
// step's output stores and the band prefetch -- every wave would sit out their latency before its first MFMA.
__device__ __forceinline__ int nvalid_at(const uint8_t* nvalid, long long idx) {
  typedef __attribute__((address_space(4))) const uint32_t cu32_t;
  const uintptr_t a = reinterpret_cast<uintptr_t>(nvalid) + (uintptr_t)idx;
  const uint32_t w = *reinterpret_cast<cu32_t*>(a & ~(uintptr_t)3);      // the aligned word around the byte
  return (int)((w >> (8 * (unsigned)(a & 3))) & 0xFFu);
}

// Geometry of the specialised path: 84x84 frames, 8x8 stride-4 VALID -> 20x20 outputs.  Wave w of a workgroup owns
// output rows 4w..4w+3 (80 pixels = 5 MFMA tiles / 20 four-pixel groups), which depend on input rows 16w..16w+19
// only.  Each wave keeps ITS band of the last 4 frames in its own LDS ring and walks time on its own: the t loop has
// NO workgroup barrier (a per-step barrier costs the fp32 matrix pipe 13-20%: tools/probes/mfma_probe4.hip).
constexpr int kIW = 84, kIH = 84, kOW = 20, kBandRows = 20, kBandBytes = kBandRows * kIW;   // 1680 B = 105 uint4
constexpr int kBandVec = kBandBytes / 16;
constexpr int kSlots = 4;
constexpr int kWaveRing = kSlots * kBandBytes;                                             // 6720 B per wave

struct BandPrefetch { uint4 v0, v1; };

__device__ __forceinline__ const uint4* band_src(const Params& p, int e, int b, int wave) {
  return reinterpret_cast<const uint4*>(p.frames_ext + ((long long)e * p.B + b) * p.fsz + wave * 16 * kIW);
}
__device__ __forceinline__ BandPrefetch band_load(const uint4* src, int lane) {
  BandPrefetch r;
  r.v0 = src[lane];
  r.v1 = lane + 64 < kBandVec ? src[lane + 64] : make_uint4(0, 0, 0, 0);
  return r;
}
__device__ __forceinline__ void band_store(unsigned char* slot, const BandPrefetch& r, int lane) {
  uint4* dst = reinterpret_cast<uint4*>(slot);
  dst[lane] = r.v0;
  if (lane + 64 < kBandVec) dst[lane + 64] = r.v1;
}
// LDS accesses of ONE wave execute in program order; this only stops the compiler from moving the wave's later
// reads above its stores and drains the LDS queue.
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// Band of ext rows t0..t0+3 -> the wave's ring (all 8 loads in flight before the first store).
__device__ __forceinline__ void band_prologue(const Params& p, unsigned char* myring, int b, int t0, int wave, int lane) {
  BandPrefetch f[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) f[e] = band_load(band_src(p, t0 + e, b, wave), lane);
#pragma unroll
  for (int e = 0; e < 4; ++e) band_store(myring + ((t0 + e) % kSlots) * kBandBytes, f[e], lane);
  wave_lds_fence();
}

// ------------------------------------------------------------------------------------ //
// Forward.  k-step ks = (c*4 + r)*4 + q ; lane (kq = lane>>4, j = lane&15) supplies
//   weights  W[ky = 2r + (kq>>1)][kx = 4(kq&1) + q][c][co0 + j] / 255      (MFMA A: rows = channels)
//   pixels   frame_{t-c}[(oy*4 + ky)*iw + ox*4 + kx],  pixel = tile*16 + j  (MFMA B: cols = pixels)
// D: lane holds channels co0 + 4*(lane>>4) + {0..3} of pixel tile*16 + (lane&15).
// ------------------------------------------------------------------------------------ //
__global__ void __launch_bounds__(kThreads)
stackconv_fwd_kernel(const Params p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* w_lds = reinterpret_cast<float*>(smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  unsigned char* myring = smem + kWFloats * sizeof(float) + wave * kWaveRing;
  const int kq = lane >> 4, j = lane & 15;
  const int co0 = blockIdx.z * 16;

  {
    constexpr int kPer = (kWFloats + kThreads - 1) / kThreads;
    float wv[kPer];
#pragma unroll
    for (int u = 0; u < kPer; ++u) {
      const int idx = tid + u * kThreads;                 // idx = ((c*4 + r)*64 + lane)*4 + q
      const int q = idx & 3, l = (idx >> 2) & 63, r = (idx >> 8) & 3, c = (idx >> 10) & 3;
      const int ky = 2 * r + ((l >> 4) >> 1), kx = 4 * ((l >> 4) & 1) + q;
      wv[u] = p.w[((ky * 8 + kx) * 4 + c) * p.cout + co0 + (l & 15)];
    }
#pragma unroll
    for (int u = 0; u < kPer; ++u) {
      const int idx = tid + u * kThreads;
      if (idx < kWFloats) w_lds[idx] = wv[u] / 255.0f;
    }
  }
  f32x4_t bias4 = {0.f, 0.f, 0.f, 0.f};
  if (p.bias) {
    const float4 bv = *reinterpret_cast<const float4*>(p.bias + co0 + 4 * kq);
    bias4 = f32x4_t{bv.x, bv.y, bv.z, bv.w};
  }
  // per-lane byte offset of (tile m, pixel j, k-quad kq) inside a band slot: fixed for the whole launch
  int aoff[kMT];
#pragma unroll
  for (int m = 0; m < kMT; ++m) {
    const int pix = m * 16 + j;                       // 0..79 within the band
    const int oy = pix / kOW, ox = pix - oy * kOW;
    aoff[m] = (oy * 4 + (kq >> 1)) * kIW + ox * 4 + 4 * (kq & 1);
  }
  __syncthreads();                                    // weights visible; the only workgroup barrier

  for (int item = blockIdx.x; item < p.items; item += gridDim.x) {
    const int b = item % p.B, chunk = item / p.B;
    const int t0 = chunk * p.spc;
    const int t1 = (t0 + p.spc < p.T1) ? t0 + p.spc : p.T1;
    band_prologue(p, myring, b, t0, wave, lane);
    for (int t = t0; t < t1; ++t) {
      // Prefetch the band of the frame step t+1 adds (ext row t+4) while this step computes.
      const bool more = t + 1 < t1;
      BandPrefetch pf;
      if (more) pf = band_load(band_src(p, t + 4, b, wave), lane);
      const int nv = p.nvalid[(long long)t * p.B + b];
      f32x4_t acc[kMT];
#pragma unroll
      for (int m = 0; m < kMT; ++m) acc[m] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      for (int c = 0; c < nv; ++c) {
        const unsigned char* base = myring + ((t + 3 - c) % kSlots) * kBandBytes;
        const float* wl = w_lds + (c * 4 * 64 + lane) * 4;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          uint32_t a[kMT];
#pragma unroll
          for (int m = 0; m < kMT; ++m)
            a[m] = *reinterpret_cast<const uint32_t*>(base + aoff[m] + r * 2 * kIW);
          const f32x4_t bw = *reinterpret_cast<const f32x4_t*>(wl + r * 256);   // the 4 q-steps: one ds_read_b128
#pragma unroll
          for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int m = 0; m < kMT; ++m)
              acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[q], ubyte(a[m], q), acc[m], 0, 0, 0);
          }
        }
      }
#pragma unroll
      for (int m = 0; m < kMT; ++m) {
        const int pix = wave * 80 + m * 16 + j;
        f32x4_t v = acc[m] + bias4;
        if (p.out_relu) {
          v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
        }
        float* o = p.out + (((long long)t * p.B + b) * 400 + pix) * p.ld_out + co0 + 4 * kq;
        *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
      }
      if (more) {
        wave_lds_fence();                              // this wave's reads of frame t are done
        band_store(myring + ((t + 4) % kSlots) * kBandBytes, pf, lane);
        wave_lds_fence();
      }
    }
  }
}

// ------------------------------------------------------------------------------------ //
// The bf16 matrix pipe, exact: "bf16x3" (forward kernel further down, weight gradient next).
// The layer's inputs are uint8 pixels: every value 0..255 is EXACT in bf16 (8 significant bits).  An fp32 weight
// is the EXACT sum of three bf16 numbers, w = hi + mid + lo (8 + 8 + 8 significant bits, same exponent range as
// fp32).  So x * w = x*hi + x*mid + x*lo with every product exact (8 x 8 bits) and the accumulation in fp32 inside
// v_mfma_f32_16x16x32_bf16: the same real-number sum the fp32 MFMA evaluates, up to fp32 summation order -- at
// 3/16 of its matrix-pipe time (the bf16 MFMA does 16x the MACs per cycle).  This is not a reduced-precision
// path: no operand is rounded (tests/test_gpu_kernels.py checks it at the fp32 tolerance, tests/test_bf16_split.py
// checks hi + mid + lo == w bit for bit).
// Measured on MI355X (T=20, B=512): forward 0.43 -> 0.15 ms, weight gradient 0.43 -> 0.24 ms against the fp32-MFMA
// kernels above (kept: SEEDHIP_STACK_BF16=0 selects them for A/B runs).
// ------------------------------------------------------------------------------------ //
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
union Frag8 { uint4 u; bf16x8_t v; };
constexpr int kGroups = 8;

// two bytes (a, b) of w -> packed bf16 pair (low half = byte a): exact, float(n) of n < 256 has zero low mantissa
template <int A, int B>
__device__ __forceinline__ uint32_t bf16_pair(uint32_t w) {
  const uint32_t f0 = __float_as_uint(ubyte(w, A)), f1 = __float_as_uint(ubyte(w, B));
  return __builtin_amdgcn_perm(f1, f0, 0x07060302u);
}

// ------------------------------------------------------------------------------------ //
// Weight gradient.  dW[k][co] = sum_pixels X[pixel][k] * dY[pixel][co]; one MFMA reduces
// over 4 pixels (a "group").  Lane (kq = lane>>4, i = lane&15):
//   MFMA A (rows = 16 k-rows of an m-tile): X[pixel 4g+kq][c][ky = i>>1][kx = 4(i&1) + q]
//          -> ONE ds_read_b32 per channel c yields the bytes of the 4 m-tiles (c, q=0..3)
//   MFMA B (cols = channels):               dY[pixel 4g+kq][co0 + i]   (256-B coalesced global load)
// 16 accumulators (m-tile = c*4 + q); D: lane holds k-rows 4*(lane>>4)+{0..3}, channel co0 + (lane&15).
// Wave w reduces its band's 20 groups per step; accumulators live in registers across all the
// (column, time-chunk) items of the persistent workgroup, then the 5 waves are summed through LDS in a
// fixed order and ONE partial slice per workgroup is written.
// ------------------------------------------------------------------------------------ //
__global__ void __launch_bounds__(kThreads, 4)
stackconv_wgrad_kernel(const Params p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* red = reinterpret_cast<float*>(smem);                       // [256 k_mem][16]
  float* redb = red + kWFloats;                                      // [kWaves][16]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  unsigned char* myring = smem + (kWFloats + kWaves * 16) * sizeof(float) + wave * kWaveRing;
  const int kq = lane >> 4, i = lane & 15;
  const int co0 = blockIdx.z * 16;
  constexpr int P = 400;

  f32x4_t acc[16];
#pragma unroll
  for (int m = 0; m < 16; ++m) acc[m] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f;
  const int wrap = 4 * kIW - 4 * kOW;                                // next output row: +4 input rows, back 20 pixels

  for (int item = blockIdx.x; item < p.items; item += gridDim.x) {
    const int b = item % p.B, chunk = item / p.B;
    const int t0 = chunk * p.spc;
    const int t1 = (t0 + p.spc < p.T1) ? t0 + p.spc : p.T1;
    band_prologue(p, myring, b, t0, wave, lane);
    // B operands of the first batch of the first step
    float dyn[4];
    {
      const float* d0 = p.dy + (((long long)t0 * p.B + b) * P + wave * 80) * p.ld_out + co0 + i;
#pragma unroll
      for (int u = 0; u < 4; ++u) dyn[u] = d0[(long long)(4 * u + kq) * p.ld_out];
    }
    for (int t = t0; t < t1; ++t) {
      const bool more = t + 1 < t1;
      BandPrefetch pf;
      if (more) pf = band_load(band_src(p, t + 4, b, wave), lane);
      const int nv = p.nvalid[(long long)t * p.B + b];
      const float* dy_band = p.dy + (((long long)t * p.B + b) * P + wave * 80) * p.ld_out + co0 + i;
      const unsigned char* slot[4];
      uint32_t cmask[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        slot[c] = myring + ((t + 3 - c) % kSlots) * kBandBytes;
        cmask[c] = c < nv ? 0xFFFFFFFFu : 0u;              // cumulative-OR done mask: channel c is zero
      }
      // lane's pixel walks pk = 4g + kq inside the band; its byte offset advances incrementally
      int ox = kq;
      int aoff = (i >> 1) * kIW + ox * 4 + 4 * (i & 1);
      for (int g0 = 0; g0 < 20; g0 += 4) {
        float dyv[4];
        uint32_t word[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          dyv[u] = dyn[u];
#pragma unroll
          for (int c = 0; c < 4; ++c)
            word[u][c] = *reinterpret_cast<const uint32_t*>(slot[c] + aoff) & cmask[c];
          ox += 4; aoff += 16;
          if (ox >= kOW) { ox -= kOW; aoff += wrap; }
        }
        // B operands of the next batch (next step's first batch at the end of a step) fly under the 64 MFMAs
        {
          const bool last = g0 + 4 >= 20;
          const float* src = last ? dy_band + (long long)p.B * P * p.ld_out : dy_band;
          const int gb = last ? 0 : g0 + 4;
          if (!last || more) {
#pragma unroll
            for (int u = 0; u < 4; ++u) dyn[u] = src[(long long)(4 * (gb + u) + kq) * p.ld_out];
          }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          bsum += dyv[u];
#pragma unroll
          for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int q = 0; q < 4; ++q)
              acc[c * 4 + q] = __builtin_amdgcn_mfma_f32_16x16x4f32(ubyte(word[u][c], q), dyv[u], acc[c * 4 + q], 0, 0, 0);
        }
      }
      if (more) {
        wave_lds_fence();
        band_store(myring + ((t + 4) % kSlots) * kBandBytes, pf, lane);
        wave_lds_fence();
      }
    }
  }

  // Cross-wave reduction in wave order (deterministic), then one partial slice per workgroup.
  bsum += __shfl_xor(bsum, 16, 64);
  bsum += __shfl_xor(bsum, 32, 64);
  if (lane < 16) redb[wave * 16 + lane] = bsum;
  for (int w = 0; w < kWaves; ++w) {
    if (wave == w) {
#pragma unroll
      for (int m = 0; m < 16; ++m) {
        const int c = m >> 2, q = m & 3;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 4 * kq + r;                       // k-row within the m-tile
          const int ky = row >> 1, kx = 4 * (row & 1) + q;
          const int idx = ((ky * 8 + kx) * 4 + c) * 16 + i;
          red[idx] = (w == 0) ? acc[m][r] : red[idx] + acc[m][r];
        }
      }
    }
    __syncthreads();
  }
  float* pw = p.partial_w + (long long)blockIdx.x * 256 * p.cout;
  for (int idx = tid; idx < kWFloats; idx += kThreads)
    pw[(idx >> 4) * p.cout + co0 + (idx & 15)] = red[idx] / 255.0f;
  if (p.partial_b && tid < 16) {
    float s = 0.f;
    for (int w = 0; w < kWaves; ++w) s += redb[w * 16 + tid];
    p.partial_b[(long long)blockIdx.x * p.cout + co0 + tid] = s;
  }
}

// ------------------------------------------------------------------------------------ //
// Weight gradient on the bf16 matrix pipe, exact ("bf16x3"): X is uint8 (exact in bf16), each dY value is split into
// three bf16 parts whose sum is the fp32 value, products are exact, accumulation is fp32 (kernel: channel-per-wave
// decomposition further down).  v_mfma_f32_16x16x32_bf16 reduces 32 pixels per instruction: lane group kq owns an
// 8-pixel chunk = 2 output rows x 4 pixels ((2rp + a, 4xc + b), element e = 4a + b).
//   A (rows = 16 k-rows of m-tile (c, q); row i = (ky = i>>1, kx = 4(i&1) + q)): frames are held in LDS as bf16;
//     pixel (a, b) contributes halfword q of the 8 bytes at element (8rp + 4a + ky)*84 + 16xc + 4b + 4(i&1):
//     8 ds_read_b64 serve the four m-tiles q = 0..3, one v_perm joins each pair of pixels
//   B (cols = 16 channels): dY[pixel e of chunk][co0 + j], 8 global dwords per lane and group.
// ------------------------------------------------------------------------------------ //
// Exact split by TRUNCATION (cheaper than rounding, equally exact): hi = the top 8 significant bits of v, mid = the
// top 8 of what is left, lo = the remaining <= 8 bits; hi + mid + lo == v bit for bit (tests/test_bf16_split.py).
__device__ __forceinline__ void split3_pack(const float (&v)[8], Frag8 (&out)[3]) {
  uint32_t h[8], m[8], l[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    h[e] = __float_as_uint(v[e]) & 0xFFFF0000u;
    const float r1 = v[e] - __uint_as_float(h[e]);
    m[e] = __float_as_uint(r1) & 0xFFFF0000u;
    l[e] = __float_as_uint(r1 - __uint_as_float(m[e]));            // <= 8 significant bits: low half is zero
  }
  uint32_t part[3][4];
#pragma unroll
  for (int e2 = 0; e2 < 4; ++e2) {                                  // pack high halves of elements (2*e2, 2*e2 + 1)
    part[0][e2] = __builtin_amdgcn_perm(h[2 * e2 + 1], h[2 * e2], 0x07060302u);
    part[1][e2] = __builtin_amdgcn_perm(m[2 * e2 + 1], m[2 * e2], 0x07060302u);
    part[2][e2] = __builtin_amdgcn_perm(l[2 * e2 + 1], l[2 * e2], 0x07060302u);
  }
#pragma unroll
  for (int s3 = 0; s3 < 3; ++s3) out[s3].u = make_uint4(part[s3][0], part[s3][1], part[s3][2], part[s3][3]);
}

// ------------------------------------------------------------------------------------ //
// bf16x3 forward.  The wave's band ring holds bf16: each frame byte is converted ONCE, when its band is staged
// (u8 -> bf16 is exact), instead of at each of its ~16 uses (4 stack positions x 2 x 2 overlapping windows).
//   k-group G = (c, half): 32 k = lane group kq (ky = 4*half + kq) x 8 horizontally adjacent pixels (kx = 0..7)
//   B operand (cols = 16 pixels): 8 consecutive bf16 of the band = two ds_read_b64, no conversion in the loop
//   A operand (rows = 16 channels): W/255 split; hi and mid parts live in registers (8 groups x 2 x 4 VGPRs), the lo
//     parts in LDS (8 KB, one ds_read_b128 per group) -- all three in registers spill at 3 waves per SIMD
//   3 MFMAs per (G, pixel tile) instead of 8 fp32 MFMAs of twice the duration.
// ------------------------------------------------------------------------------------ //
constexpr int kBand16 = kBandBytes * 2;                  // 3360 B: one band slot in bf16
constexpr int kWaveRing16 = kSlots * kBand16;            // 13440 B per wave

// 16 frame bytes -> 16 bf16 (two uint4)
__device__ __forceinline__ void cvt16(const uint4& v, uint4& lo, uint4& hi) {
  lo = make_uint4(bf16_pair<0, 1>(v.x), bf16_pair<2, 3>(v.x), bf16_pair<0, 1>(v.y), bf16_pair<2, 3>(v.y));
  hi = make_uint4(bf16_pair<0, 1>(v.z), bf16_pair<2, 3>(v.z), bf16_pair<0, 1>(v.w), bf16_pair<2, 3>(v.w));
}
__device__ __forceinline__ void band_store16(unsigned char* slot, const BandPrefetch& r, int lane) {
  uint4* dst = reinterpret_cast<uint4*>(slot);
  uint4 a, b;
  cvt16(r.v0, a, b);
  dst[2 * lane] = a; dst[2 * lane + 1] = b;
  if (lane + 64 < kBandVec) {
    cvt16(r.v1, a, b);
    dst[2 * (lane + 64)] = a; dst[2 * (lane + 64) + 1] = b;
  }
}
__device__ __forceinline__ void band_prologue16(const Params& p, unsigned char* myring, int b, int t0, int wave, int lane) {
  BandPrefetch f[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) f[e] = band_load(band_src(p, t0 + e, b, wave), lane);
#pragma unroll
  for (int e = 0; e < 4; ++e) band_store16(myring + ((t0 + e) % kSlots) * kBand16, f[e], lane);
  wave_lds_fence();
}

// EXP (tools/probes/stack_probe.hip only; the library instantiates EXP = 0): leave one ingredient out -- 1 no MFMAs,
// 2 no global band prefetch after the prologue, 4 no output stores, 8 no LDS pixel reads (one fragment reused),
// 16 no band conversion / LDS store.
// BITS: also write the ReLU byte mask (Params::relu_bits).  RELU: the output activation as a template parameter -- a
// run-time `if (p.out_relu)` is a branch per output tile, and every branch in the epilogue is a basic-block boundary the
// scheduler cannot move the stores / the next frame's work across (an untaken one around the byte store cost 20 us).
// BUF (r4): the time loop reaches frames_ext and out through buffer resources -- a step's base is a UNIFORM 32-bit byte
// offset (SGPR soffset), the lane's part a loop-invariant 32-bit register.  With 64-bit per-lane pointers the kernel
// recomputed five output addresses per step (v_mad_u64 chains) from loop invariants that did not fit its 168 registers:
// four 64-bit values were SPILLED and reloaded at the head of every step, each reload behind an s_waitcnt vmcnt(0) --
// i.e. every step first waited for the previous step's five output stores to reach memory (found in the ISA:
// tools/isa_waits.py prints "spilled VGPRs 8"; the matrix pipe was 26 % busy).
template <int EXP, bool BITS = false, bool RELU = true, bool BUF = false>
__global__ void __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(3, 3)))
stackconv_fwd_bf16r_kernel(const Params p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  uint4* wlo_lds = reinterpret_cast<uint4*>(smem);                   // [G][lane]: the lo parts (8 KB); hi, mid in registers
  unsigned char* myring = smem + kGroups * 64 * 16 + wave * kWaveRing16;
  const int kq = lane >> 4, j = lane & 15;
  const int co0 = blockIdx.z * 16;

  // ---- W/255 -> three bf16 parts, in registers: wreg[G][part] = 8 k (kx = 0..7) of row ky = 4*half + kq ----
  Frag8 wreg[kGroups][2];
#pragma unroll
  for (int G = 0; G < kGroups; ++G) {
    const int c = G >> 1, ky = 4 * (G & 1) + kq;
    uint32_t part[3][4];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float w = p.w[((ky * 8 + e) * 4 + c) * p.cout + co0 + j] / 255.0f;
      const uint32_t hi = __float_as_uint(w) >> 16;                   // exact split by truncation (see split3_pack)
      const float r1 = w - __uint_as_float(hi << 16);
      const uint32_t mid = __float_as_uint(r1) >> 16;
      const uint32_t lo = __float_as_uint(r1 - __uint_as_float(mid << 16)) >> 16;
      const uint32_t v[3] = {hi, mid, lo};
#pragma unroll
      for (int s3 = 0; s3 < 3; ++s3) {
        if (e & 1) part[s3][e >> 1] |= v[s3] << 16; else part[s3][e >> 1] = v[s3];
      }
    }
#pragma unroll
    for (int s3 = 0; s3 < 2; ++s3) wreg[G][s3].u = make_uint4(part[s3][0], part[s3][1], part[s3][2], part[s3][3]);
    if (wave == 0) wlo_lds[G * 64 + lane] = make_uint4(part[2][0], part[2][1], part[2][2], part[2][3]);
  }
  f32x4_t bias4 = {0.f, 0.f, 0.f, 0.f};
  if (p.bias) {
    const float4 bv = *reinterpret_cast<const float4*>(p.bias + co0 + 4 * kq);
    bias4 = f32x4_t{bv.x, bv.y, bv.z, bv.w};
  }
  int aoff[kMT];                                      // byte offset of (tile m, pixel j, row kq) inside a bf16 band slot
#pragma unroll
  for (int m = 0; m < kMT; ++m) {
    const int pix = m * 16 + j;
    const int oy = pix / kOW, ox = pix - oy * kOW;
    aoff[m] = ((oy * 4 + kq) * kIW + ox * 4) * 2;
  }
  __syncthreads();                                    // lo parts visible; the only workgroup barrier

  // BUF: views of the two big tensors and the lane's loop-invariant byte offsets into a step's slice of them
  const __amdgpu_buffer_rsrc_t fview = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint8_t*>(p.frames_ext), 0, BUF ? (int)((long long)(3 + p.T1) * p.B * p.fsz) : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t oview = __builtin_amdgcn_make_buffer_rsrc(
      p.out, 0, BUF ? (int)((long long)p.T1 * p.B * 400 * p.ld_out * 4) : 0, 0x00020000);
  // BITS + BUF: the byte mask is indexed like out / 16 (one byte per lane's four channels)
  const __amdgpu_buffer_rsrc_t bview = __builtin_amdgcn_make_buffer_rsrc(
      p.relu_bits, 0, (BUF && BITS) ? (int)((long long)p.T1 * p.B * 400 * p.ld_out / 4) : 0, 0x00020000);
  const unsigned fv0 = 16u * (unsigned)lane, fv1 = lane + 64 < kBandVec ? 16u * (unsigned)(lane + 64) : 0x80000000u;
  unsigned ov[kMT];
#pragma unroll
  for (int m = 0; m < kMT; ++m) ov[m] = (unsigned)(((wave * 80 + m * 16 + j) * p.ld_out + co0 + 4 * kq) * 4);

  for (int item = blockIdx.x; item < p.items; item += gridDim.x) {
    const int b = item % p.B, chunk = item / p.B;
    const int t0 = chunk * p.spc;
    const int t1 = (t0 + p.spc < p.T1) ? t0 + p.spc : p.T1;
    band_prologue16(p, myring, b, t0, wave, lane);
    for (int t = t0; t < t1; ++t) {
      const bool more = t + 1 < t1;
      const int nv = nvalid_at(p.nvalid, (long long)t * p.B + b);
      BandPrefetch pf;
      if (EXP & 2) { pf.v0 = make_uint4(lane, t, 3, 4); pf.v1 = pf.v0; }
      else if (BUF) {
        if (more) {
          const unsigned so = __builtin_amdgcn_readfirstlane((unsigned)(((t + 4) * p.B + b) * p.fsz + wave * 16 * kIW));
          pf.v0 = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(fview, fv0, so, 0));
          pf.v1 = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(fview, fv1, so, 0));
        }
      }
      else if (more) pf = band_load(band_src(p, t + 4, b, wave), lane);
      f32x4_t acc[kMT];
#pragma unroll
      for (int m = 0; m < kMT; ++m) acc[m] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      auto group = [&](int G) {                       // G static after unrolling: weights stay in registers
        const unsigned char* base = myring + ((t + 3 - (G >> 1)) % kSlots) * kBand16 + (G & 1) * 4 * kIW * 2;
        Frag8 wlo;
        wlo.u = wlo_lds[G * 64 + lane];
        // the five pixel fragments first, then part-major MFMAs: consecutive instructions then write DIFFERENT
        // accumulators (a dependent 16x16x32 bf16 MFMA issued straight behind its producer waits out the producer's
        // passes: lo -> mid -> hi on one accumulator ran the matrix pipe at ~40 %, tools/probes/stack_probe.hip)
        Frag8 xf[kMT];
#pragma unroll
        for (int m = 0; m < kMT; ++m) {
          // two 8-byte reads with their own 16-bit immediate offsets: merged into one ds_read2_b64 (8-bit offsets in
          // units of 8 bytes) every fragment costs a v_add for its base -- 60 VALU instructions per 120 MFMAs
          typedef __attribute__((address_space(3))) const volatile unsigned long long lds_cv64_t;
          lds_cv64_t* src = (lds_cv64_t*)(base + ((EXP & 8) ? aoff[0] : aoff[m]));
          if ((EXP & 8) && (m > 0 || G > 0)) xf[m].u = make_uint4(t, G, m, lane);
          else {
            const unsigned long long x0 = src[0], x1 = src[1];
            xf[m].u = make_uint4((unsigned)x0, (unsigned)(x0 >> 32), (unsigned)x1, (unsigned)(x1 >> 32));
          }
        }
        if (EXP & 1) {
#pragma unroll
          for (int m = 0; m < kMT; ++m) asm volatile("" :: "v"(xf[m].u.x), "v"(xf[m].u.y), "v"(xf[m].u.z), "v"(xf[m].u.w), "v"(wlo.u.x), "v"(wlo.u.w));
          return;
        }
#pragma unroll
        for (int m = 0; m < kMT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wlo.v, xf[m].v, acc[m], 0, 0, 0);         // lo
#pragma unroll
        for (int m = 0; m < kMT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wreg[G][1].v, xf[m].v, acc[m], 0, 0, 0);  // mid
#pragma unroll
        for (int m = 0; m < kMT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wreg[G][0].v, xf[m].v, acc[m], 0, 0, 0);  // hi
      };
      if (nv == 4) {                                   // the common case: one straight-line block of 8 k-groups
#pragma unroll
        for (int G = 0; G < kGroups; ++G) group(G);
      } else {
#pragma unroll
        for (int G = 0; G < kGroups; ++G) if (G < 2 * nv) group(G);
      }
      // The next band goes to LDS BEFORE this step's output stores are issued: waiting for the prefetch with stores in
      // flight is s_waitcnt vmcnt(0) (loads and stores share the counter and do not retire in order with each other), i.e.
      // the wave would sit out the write latency of stores it has just issued, every step.  Issued after it, the stores
      // have the whole next step to complete.  (Same-box A/B: 129.8 -> 126.7 us alone, nothing on the step: the other
      // waves of the SIMD cover most of that wait.)
      if (more) {
        wave_lds_fence();                              // this wave's reads of frame t are done (the MFMA operands above)
        if (EXP & 16) asm volatile("" :: "v"(pf.v0.x), "v"(pf.v0.y), "v"(pf.v0.z), "v"(pf.v0.w), "v"(pf.v1.x), "v"(pf.v1.y), "v"(pf.v1.z), "v"(pf.v1.w));
        else band_store16(myring + ((t + 4) % kSlots) * kBand16, pf, lane);
        wave_lds_fence();
      }
      const unsigned oso = __builtin_amdgcn_readfirstlane((unsigned)((t * p.B + b) * 400 * p.ld_out) * 4u);   // BUF: this step's slice of out
      // Every output is FINISHED before the first store (pinned): a 16-byte store reads its data registers over several
      // cycles, and hipcc put the next tile's `v_pk_add_f32` -- which reuses them -- into the very next issue slot
      // behind `buffer_store_dwordx4` (no wait state: the stored quad's upper half came out as the next tile's;
      // tools/isa_store_hazard.py finds the pattern in a compiled object, tests/test_isa_structure.py runs it)
      f32x4_t vout[kMT];
#pragma unroll
      for (int m = 0; m < kMT; ++m) {
        vout[m] = acc[m] + bias4;
        if (RELU) {                                    // one v_med3 each (fmaxf is two: it first quiets its operand)
#pragma unroll
          for (int r = 0; r < 4; ++r) vout[m][r] = __builtin_amdgcn_fmed3f(vout[m][r], 0.f, __builtin_inff());
        }
      }
#pragma unroll
      for (int m = 0; m < kMT; ++m) asm volatile("" : "+v"(vout[m]));
#pragma unroll
      for (int m = 0; m < kMT; ++m) {
        const int pix = wave * 80 + m * 16 + j;
        const f32x4_t v = vout[m];
        float* o = p.out + (((long long)t * p.B + b) * 400 + pix) * p.ld_out + co0 + 4 * kq;
        if (EXP & 4) asm volatile("" :: "v"(v));
        else if (BUF) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(su32x4_t, v), oview, ov[m], oso, 0);
        else *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
        // the ReLU mask the next layer's data gradient needs, as one byte per lane (its four channels): that kernel then
        // reads 1 byte where it read the 16 of the activation (wsgemm.h, ws_tab_kernel<.., BITS>)
        if (BITS) {
          // v is post-ReLU (v_med3 with +0 and +inf never returns -0): v > 0 <=> its bit pattern != 0; seven VALU
          // instructions per quad (4 v_min_u32 + 3 v_lshl_or_b32) where compares + selects + ors were eleven
          const su32x4_t bu = __builtin_bit_cast(su32x4_t, v);
          const unsigned m01 = ((bu[1] < 1u ? bu[1] : 1u) << 1) | (bu[0] < 1u ? bu[0] : 1u);
          const unsigned m23 = ((bu[3] < 1u ? bu[3] : 1u) << 1) | (bu[2] < 1u ? bu[2] : 1u);
          const unsigned char mk = (unsigned char)((m23 << 2) | m01);
          if (BUF) __builtin_amdgcn_raw_buffer_store_b8(mk, bview, ov[m] >> 4, oso >> 4, 0);
          else p.relu_bits[(o - p.out) >> 2] = mk;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------ //
// bf16x3 forward, EIGHT waves (r6).  The five-wave kernel above cannot balance a CU: a workgroup's waves go to the four
// SIMDs in a fixed cyclic order, so waves 0 and 4 of every workgroup share one -- with two workgroups per CU one SIMD
// runs FOUR waves and the others two.  tools/probes/stack_probe.hip says so directly: the MFMA-only build takes 95.5 us
// with two workgroups per CU and 90.2 us with ONE (= 10 080 MFMAs of 17.9 cycles on the loaded SIMD either way, the matrix
// pipes of the other three half idle), and the full kernel 135-147 against 139.
// Here one 512-thread workgroup per CU walks TWO batch columns; wave w (column w >> 2) owns a run of 16-pixel tiles of
// its column's 400 output pixels -- 7 + 6 + 6 + 6 tiles for column 0, 6 + 6 + 6 + 7 for column 1 --, and waves w and
// w + 4 share a SIMD: 13 / 12 / 12 / 13 tiles per SIMD.  A run of <= 7 tiles touches <= 7 output rows = a band of 28 input
// rows (4 704 B in bf16, four ring slots per wave: 150.5 KB of LDS); two waves per SIMD have 256 registers each, so all
// three planes of W / 255 stay in registers (96) and nothing but pixels is read from LDS.  Operands, MFMA order per
// accumulator and epilogue are those of the five-wave kernel: bit-identical outputs.
// ------------------------------------------------------------------------------------ //
constexpr int kW8Waves = 8, kW8Threads = kW8Waves * 64;
constexpr int kW8Rows = 28, kW8Band8 = kW8Rows * kIW, kW8Band16 = 2 * kW8Band8;       // 2 352 B of pixels, 4 704 B as bf16
constexpr int kW8Vec = kW8Band8 / 16;                                               // 147 uint4 per band
constexpr int kW8Ring = kSlots * kW8Band16;                                         // 18 816 B per wave
constexpr int kW8Rings = kW8Waves * kW8Ring;                                        // 150 528 B
constexpr int kW8Lds = kW8Rings + kW8Waves * 512;                                   // + the mask scratch of each wave

struct Band3 { uint4 v[3]; };

template <int NT, int EXP, bool BITS, bool RELU, int MODE>
__device__ __forceinline__ void w8_run(const Params& p, unsigned char* myring, unsigned char* scratch, const Frag8 (&wreg)[kGroups][3],
                                       const f32x4_t bias4, int col, int p0, int lane) {
  const int kq = lane >> 4, j = lane & 15;
  const int co0 = blockIdx.z * 16;
  const int row0 = 4 * (p0 / kOW);                     // first input row of the band
  int aoff[NT];                                        // byte offset of (tile m, pixel j, row kq) inside a bf16 band slot
  unsigned ov[NT];
#pragma unroll
  for (int m = 0; m < NT; ++m) {
    const int pix = p0 + m * 16 + j;
    const int oy = pix / kOW, ox = pix - oy * kOW;
    aoff[m] = ((oy * 4 - row0 + kq) * kIW + ox * 4) * 2;
    ov[m] = (unsigned)((pix * p.ld_out + co0 + 4 * kq) * 4);
  }
  const __amdgpu_buffer_rsrc_t fview = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint8_t*>(p.frames_ext), 0, (int)((long long)(3 + p.T1) * p.B * p.fsz), 0x00020000);
  const __amdgpu_buffer_rsrc_t oview = __builtin_amdgcn_make_buffer_rsrc(
      p.out, 0, (int)((long long)p.T1 * p.B * 400 * p.ld_out * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t bview = __builtin_amdgcn_make_buffer_rsrc(
      p.relu_bits, 0, BITS ? (int)((long long)p.T1 * p.B * 400 * p.ld_out / 4) : 0, 0x00020000);
  // the band's 147 16-byte pieces: lanes 0..63 take pieces l, l + 64 and (l < 19) l + 128; a band that reaches past
  // its frame (the last wave's 28 rows from row 56 / 60 on) reads into the next frame or, at the tensor's end, zeros:
  // those rows belong to no pixel of the run
  const unsigned fv[3] = {16u * (unsigned)lane, 16u * (unsigned)(lane + 64), lane + 128 < kW8Vec ? 16u * (unsigned)(lane + 128) : 0x80000000u};
  auto band_get = [&](int e, int b) -> Band3 {
    const unsigned so = __builtin_amdgcn_readfirstlane((unsigned)((e * p.B + b) * p.fsz + row0 * kIW));
    Band3 r;
#pragma unroll
    for (int u = 0; u < 3; ++u) r.v[u] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(fview, fv[u], so, 0));
    return r;
  };
  auto band_put = [&](unsigned char* slot, const Band3& r) {
    uint4* dst = reinterpret_cast<uint4*>(slot);
    uint4 a, b;
    cvt16(r.v[0], a, b); dst[2 * lane] = a; dst[2 * lane + 1] = b;
    cvt16(r.v[1], a, b); dst[2 * (lane + 64)] = a; dst[2 * (lane + 64) + 1] = b;
    if (lane + 128 < kW8Vec) { cvt16(r.v[2], a, b); dst[2 * (lane + 128)] = a; dst[2 * (lane + 128) + 1] = b; }
  };
  // MODE 2: the band requests are asm statements hipcc's s_waitcnt bookkeeping knows nothing about, and the wait for them
  // counts the stores issued behind them (vector memory operations retire in order: hipcc itself waits vmcnt(1) for a
  // load that one store follows); the registers are read behind the wait only (take_band)
  typedef unsigned sgpr128_t __attribute__((ext_vector_type(4)));
  const uint64_t fbase = reinterpret_cast<uint64_t>(p.frames_ext);
  const sgpr128_t fwords = {(unsigned)__builtin_amdgcn_readfirstlane((unsigned)fbase),
                            (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(fbase >> 32)) & 0xFFFFu,
                            (unsigned)__builtin_amdgcn_readfirstlane((int)((long long)(3 + p.T1) * p.B * p.fsz)), 0x00020000u};
  auto band_req = [&](int e, int b, f32x4_t (&r)[3]) {
    const unsigned so = __builtin_amdgcn_readfirstlane((unsigned)((e * p.B + b) * p.fsz + row0 * kIW));
    asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %3, %6, %7 offen\n\tbuffer_load_dwordx4 %1, %4, %6, %7 offen\n\t"
                 "buffer_load_dwordx4 %2, %5, %6, %7 offen"
                 : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]) : "v"(fv[0]), "v"(fv[1]), "v"(fv[2]), "s"(fwords), "s"(so)
                 : "memory");                          // (no store of the step may be moved in front of the request: the wait counts them)
  };
  constexpr int kStoresPerStep = (EXP & 4) ? 0 : (BITS ? NT + 2 : NT);
  // BITS: byte offset of pixel lane + 64 h's mask dword inside a step's slice of relu_bits (pixels past the run: out of range)
  const unsigned bvo[2] = {(unsigned)(((p0 + lane) * p.ld_out + co0) >> 2),
                           lane + 64 < 16 * NT ? (unsigned)(((p0 + lane + 64) * p.ld_out + co0) >> 2) : 0x80000000u};
  auto take_band = [&](const f32x4_t (&r)[3], bool first) -> Band3 {   // first: nothing was issued behind the request
    if (first) asm volatile("s_waitcnt vmcnt(0)");
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    f32x2_t o[6];
    asm volatile("s_waitcnt vmcnt(%12)\n\tv_mov_b64 %0, %6\n\tv_mov_b64 %1, %7\n\tv_mov_b64 %2, %8\n\tv_mov_b64 %3, %9\n\t"
                 "v_mov_b64 %4, %10\n\tv_mov_b64 %5, %11"
                 : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5])
                 : "v"(__builtin_shufflevector(r[0], r[0], 0, 1)), "v"(__builtin_shufflevector(r[0], r[0], 2, 3)),
                   "v"(__builtin_shufflevector(r[1], r[1], 0, 1)), "v"(__builtin_shufflevector(r[1], r[1], 2, 3)),
                   "v"(__builtin_shufflevector(r[2], r[2], 0, 1)), "v"(__builtin_shufflevector(r[2], r[2], 2, 3)), "n"(kStoresPerStep));
    Band3 b3;
#pragma unroll
    for (int u = 0; u < 3; ++u)
      b3.v[u] = make_uint4(__float_as_uint(o[2 * u][0]), __float_as_uint(o[2 * u][1]), __float_as_uint(o[2 * u + 1][0]), __float_as_uint(o[2 * u + 1][1]));
    return b3;
  };
  // EXP & 32 (probe): s_memtime stamps of workgroup 0's waves, five per step, into partial_w as [wave][step][8] words
  unsigned* stamps = reinterpret_cast<unsigned*>(p.partial_w) + (size_t)(threadIdx.x >> 6) * 64 * 8;
  auto stamp = [&](int t, int k) {
    if ((EXP & 32) && blockIdx.x == 0 && t < 64) {
      const unsigned long long now = __builtin_readcyclecounter();
      if (lane == 0) stamps[t * 8 + k] = (unsigned)now;
    }
  };
  const int pairs = (p.B + 1) >> 1;
  for (int item = blockIdx.x; item < p.items; item += gridDim.x) {
    const int b = 2 * (item % pairs) + col, chunk = item / pairs;
    if (b >= p.B) continue;                            // odd B: the last pair has one column (wave-uniform)
    const int t0 = chunk * p.spc;
    const int t1 = (t0 + p.spc < p.T1) ? t0 + p.spc : p.T1;
    {
      Band3 f[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) f[e] = band_get(t0 + e, b);
#pragma unroll
      for (int e = 0; e < 4; ++e) band_put(myring + ((t0 + e) % kSlots) * kW8Band16, f[e]);
      wave_lds_fence();
    }
    Band3 pf;
    f32x4_t pfr[3];
    if (MODE >= 2) {                                   // the band of step t0 + 1 is on its way before the first step
      if (EXP & 2) { pf.v[0] = make_uint4(lane, t0, 3, 4); pf.v[1] = pf.v[0]; pf.v[2] = pf.v[0]; }
      else if (t0 + 1 < t1) band_req(t0 + 4, b, pfr);
    }
    for (int t = t0; t < t1; ++t) {
      const bool more = t + 1 < t1;
      const int nv = nvalid_at(p.nvalid, (long long)t * p.B + b);
      if (MODE < 2) {
        if (EXP & 2) { pf.v[0] = make_uint4(lane, t, 3, 4); pf.v[1] = pf.v[0]; pf.v[2] = pf.v[0]; }
        else if (more) pf = band_get(t + 4, b);
      }
      f32x4_t acc[NT];
#pragma unroll
      for (int m = 0; m < NT; ++m) acc[m] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      // the MFMAs of tiles [LO, HI) over the valid k-groups; the outputs of tiles [LO, HI): bias, activation, stores
      auto mma = [&](auto lo_c, auto hi_c) {
        constexpr int LO = decltype(lo_c)::value, HI = decltype(hi_c)::value;
        auto group = [&](int G) {                     // G static after unrolling
          const unsigned char* base = myring + ((t + 3 - (G >> 1)) % kSlots) * kW8Band16 + (G & 1) * 4 * kIW * 2;
          Frag8 xf[NT];
#pragma unroll
          for (int m = LO; m < HI; ++m) {
            typedef __attribute__((address_space(3))) const volatile unsigned long long lds_cv64_t;
            lds_cv64_t* src = (lds_cv64_t*)(base + ((EXP & 8) ? aoff[0] : aoff[m]));
            if ((EXP & 8) && (m > 0 || G > 0)) xf[m].u = make_uint4(t, G, m, lane);
            else {
              const unsigned long long x0 = src[0], x1 = src[1];
              xf[m].u = make_uint4((unsigned)x0, (unsigned)(x0 >> 32), (unsigned)x1, (unsigned)(x1 >> 32));
            }
          }
          if (EXP & 1) {
#pragma unroll
            for (int m = LO; m < HI; ++m) asm volatile("" :: "v"(xf[m].u.x), "v"(xf[m].u.y), "v"(xf[m].u.z), "v"(xf[m].u.w));
            return;
          }
#pragma unroll
          for (int s3 = 2; s3 >= 0; --s3)             // lo, mid, hi: consecutive MFMAs write different accumulators
#pragma unroll
            for (int m = LO; m < HI; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wreg[G][s3].v, xf[m].v, acc[m], 0, 0, 0);
        };
        if (nv == 4) {
#pragma unroll
          for (int G = 0; G < kGroups; ++G) group(G);
        } else {
#pragma unroll
          for (int G = 0; G < kGroups; ++G) if (G < 2 * nv) group(G);
        }
      };
      const unsigned oso = __builtin_amdgcn_readfirstlane((unsigned)((t * p.B + b) * 400 * p.ld_out) * 4u);
      auto emit = [&](auto lo_c, auto hi_c) {
        constexpr int LO = decltype(lo_c)::value, HI = decltype(hi_c)::value;
        f32x4_t vout[NT];
#pragma unroll
        for (int m = LO; m < HI; ++m) {
          vout[m] = acc[m] + bias4;
          if (RELU) {
#pragma unroll
            for (int r = 0; r < 4; ++r) vout[m][r] = __builtin_amdgcn_fmed3f(vout[m][r], 0.f, __builtin_inff());
          }
        }
#pragma unroll
        for (int m = LO; m < HI; ++m) asm volatile("" : "+v"(vout[m]));  // every output finished before the first store (see above)
#pragma unroll
        for (int m = LO; m < HI; ++m) {
          const f32x4_t v = vout[m];
          if (EXP & 4) asm volatile("" :: "v"(v));
          else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(su32x4_t, v), oview, ov[m], oso, 0);
          if (BITS) {
            // the ReLU mask: the lane's byte (its four channels of tile m's pixel j) goes to the wave's LDS scratch as
            // [pixel][kq]; one dword per PIXEL is stored from there -- two store instructions per run where every tile
            // had its own byte store (a vector memory instruction costs its issue slot in the CU's queue whatever it carries:
            // 7 byte stores per wave and step were 10 us of the kernel)
            const su32x4_t bu = __builtin_bit_cast(su32x4_t, v);
            const unsigned m01 = ((bu[1] < 1u ? bu[1] : 1u) << 1) | (bu[0] < 1u ? bu[0] : 1u);
            const unsigned m23 = ((bu[3] < 1u ? bu[3] : 1u) << 1) | (bu[2] < 1u ? bu[2] : 1u);
            scratch[m * 64 + j * 4 + kq] = (unsigned char)((m23 << 2) | m01);
          }
        }
        if (BITS && HI == NT) {
          wave_lds_fence();
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const unsigned w = reinterpret_cast<const unsigned*>(scratch)[(lane + 64 * h < 16 * NT) ? lane + 64 * h : 0];
            if (!(EXP & 4)) __builtin_amdgcn_raw_buffer_store_b32(w, bview, bvo[h], oso >> 4, 0);
          }
          wave_lds_fence();
        }
      };
      auto stage = [&]() {                             // the next band goes to LDS before the stores behind it are issued (see above)
        if (more) {
          if (MODE == 2 && !(EXP & 2)) pf = take_band(pfr, t == t0);
          if (MODE == 0 && (EXP & 32)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          stamp(t, 2);
          wave_lds_fence();
          if (EXP & 16) asm volatile("" :: "v"(pf.v[0].x), "v"(pf.v[0].y), "v"(pf.v[1].x), "v"(pf.v[1].y), "v"(pf.v[2].x), "v"(pf.v[2].y));
          else band_put(myring + ((t + 4) % kSlots) * kW8Band16, pf);
          wave_lds_fence();
        }
      };
      typedef std::integral_constant<int, 0> c0_t;
      typedef std::integral_constant<int, NT> cn_t;
      if (MODE == 0) {
        stamp(t, 0);
        mma(c0_t(), cn_t());
        stamp(t, 1);
        stage();
        stamp(t, 3);
        emit(c0_t(), cn_t());
        stamp(t, 4);
      } else if (MODE == 2) {
        // the request for step t + 2's band is issued IN FRONT of this step's stores: a CU's vector memory pipeline is a
        // queue, and a load behind the 57 KB the eight waves store per step waits until they have drained; the wait for
        // it, one step later, is vmcnt(number of younger stores), not vmcnt(0)
        stamp(t, 0);
        mma(c0_t(), cn_t());
        stamp(t, 1);
        stage();
        stamp(t, 3);
        if (!(EXP & 2) && t + 2 < t1) band_req(t + 5, b, pfr);
        emit(c0_t(), cn_t());
        stamp(t, 4);
      }
    }
  }
}

template <int EXP, bool BITS = false, bool RELU = true, int MODE = 0>
__global__ void __launch_bounds__(kW8Threads) __attribute__((amdgpu_waves_per_eu(2, 2)))
stackconv_fwd_w8_kernel(const Params p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  unsigned char* myring = smem + wave * kW8Ring;
  const int kq = lane >> 4, j = lane & 15;
  const int co0 = blockIdx.z * 16;
  // W / 255 -> three bf16 parts in registers: wreg[G][part] = 8 k (kx = 0..7) of row ky = 4 * half + kq (the five-wave kernel's split)
  Frag8 wreg[kGroups][3];
#pragma unroll
  for (int G = 0; G < kGroups; ++G) {
    const int c = G >> 1, ky = 4 * (G & 1) + kq;
    uint32_t part[3][4];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float w = p.w[((ky * 8 + e) * 4 + c) * p.cout + co0 + j] / 255.0f;
      const uint32_t hi = __float_as_uint(w) >> 16;
      const float r1 = w - __uint_as_float(hi << 16);
      const uint32_t mid = __float_as_uint(r1) >> 16;
      const uint32_t lo = __float_as_uint(r1 - __uint_as_float(mid << 16)) >> 16;
      const uint32_t v[3] = {hi, mid, lo};
#pragma unroll
      for (int s3 = 0; s3 < 3; ++s3) {
        if (e & 1) part[s3][e >> 1] |= v[s3] << 16; else part[s3][e >> 1] = v[s3];
      }
    }
#pragma unroll
    for (int s3 = 0; s3 < 3; ++s3) wreg[G][s3].u = make_uint4(part[s3][0], part[s3][1], part[s3][2], part[s3][3]);
  }
  f32x4_t bias4 = {0.f, 0.f, 0.f, 0.f};
  if (p.bias) {
    const float4 bv = *reinterpret_cast<const float4*>(p.bias + co0 + 4 * kq);
    bias4 = f32x4_t{bv.x, bv.y, bv.z, bv.w};
  }
  // runs of tiles: column 0 = 7 + 6 + 6 + 6 from pixel 0 / 112 / 208 / 304, column 1 = 6 + 6 + 6 + 7 from 0 / 96 / 192 / 288
  const int col = wave >> 2, q = wave & 3;
  const int p0 = 96 * q + ((col == 0 && q > 0) ? 16 : 0);
  if (wave == 0 || wave == 7) w8_run<7, EXP, BITS, RELU, MODE>(p, myring, smem + kW8Rings + wave * 512, wreg, bias4, col, p0, lane);
  else w8_run<6, EXP, BITS, RELU, MODE>(p, myring, smem + kW8Rings + wave * 512, wreg, bias4, col, p0, lane);
}

// ------------------------------------------------------------------------------------ //
// The same forward for CENTRAL INFERENCE (r6; servestep.hip): one step of n independent environments.  Row b's stack is
// its request frame obs[b] plus the three frames the unroll store already holds for that env (store_obs rows
// hist_rows[4b + c], c = 1..3; see servestep.hip), nvalid[b] of them inside the episode -- the bit-packed per-env
// stacking state, its unpack pass before and its re-pack pass after the conv do not exist here.  Same operands, same
// MFMA order per accumulator as stackconv_fwd_bf16r_kernel: bit-identical outputs.  The W / 255 planes arrive
// pre-split (w_split: serve_begin or seedhip_serve_split_conv0, the training kernel's own prologue arithmetic): a
// workgroup lives for two or three rows, not for an unroll, and the 64 divisions + splits per lane were a third of its time.
// Rows are software-pipelined through the ring BY STACK CHANNEL: the k-groups of channel c are the only readers of ring
// slot 3 - c, so the next row's frame c is staged into it as soon as they are done, from one of two prefetch registers
// (all four frames of the next row held across the MFMA phase spilled: 32 registers over the 170 of three waves per SIMD).
// ------------------------------------------------------------------------------------ //
struct RowsParams {
  const uint8_t* obs;          // u8 [B, fsz]
  const uint8_t* store_obs;    // u8 [rows, fsz]: the observation field of the unroll store (history frames)
  const long long* hist_rows;  // [B][4]
  const uint8_t* nvalid;       // [B]
  const float* bias; const uint4* w_split;
  float* out; int B, cout, ld_out, fsz;
};

template <bool RELU>
__global__ void __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(3, 3)))
stackconv_rows_kernel(const RowsParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  uint4* wlo_lds = reinterpret_cast<uint4*>(smem);
  unsigned char* myring = smem + kGroups * 64 * 16 + wave * kWaveRing16;
  const int kq = lane >> 4, j = lane & 15;
  const int co0 = blockIdx.z * 16;
  const unsigned band0 = (unsigned)__builtin_amdgcn_readfirstlane(wave * 16 * kIW);
  const unsigned fv0 = 16u * (unsigned)lane;

  // Frame c of row b as a buffer of its own: the row index comes through the scalar cache, the base is uniform, the
  // lane's part two loop-invariant 32-bit offsets (64-bit per-lane pointers cost registers this kernel does not have).
  // Channels outside the episode (c >= nvalid[b]) are loaded and staged like the others -- their history rows are always
  // in range (serve_begin) and their k-groups are skipped -- so that the staging code has no branches.
  auto band_of = [&](int b, int c) -> BandPrefetch {
    typedef __attribute__((address_space(4))) const unsigned cu32_t;
    long long row = b;
    if (c) {
      cu32_t* q = reinterpret_cast<cu32_t*>(reinterpret_cast<uintptr_t>(p.hist_rows + 4 * (long long)b + c));
      row = (long long)(((unsigned long long)q[1] << 32) | q[0]);
    }
    const uint8_t* fr = (c == 0 ? p.obs : p.store_obs) + row * p.fsz;
    const __amdgpu_buffer_rsrc_t v = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(fr), 0, p.fsz, 0x00020000);
    BandPrefetch r;
    r.v0 = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(v, fv0, band0, 0));
    // lanes >= 41 read past the band (or, in the last wave, past the frame: zeros); band_store16 does not store them
    r.v1 = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(v, fv0, band0 + 1024u, 0));
    return r;
  };
  // the first row's frames are requested before the weights, and ALL of those before the first use: one memory round
  // trip for the prologue (staged one by one, every LDS store of a lo part waited for its load: eight round trips)
  int b = __builtin_amdgcn_readfirstlane(blockIdx.x);  // grid <= B
  int nv = nvalid_at(p.nvalid, b);
  BandPrefetch f0[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) f0[c] = band_of(b, c);
  Frag8 wreg[kGroups][2];
  {
    const uint4* img = p.w_split + (long long)blockIdx.z * kGroups * 3 * 64 + lane;
    if (wave == 0) {                                   // the lo parts go memory -> LDS without passing through registers
      typedef __attribute__((address_space(3))) void lds_void_t;
      const __amdgpu_buffer_rsrc_t wview = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<uint4*>(p.w_split + (long long)blockIdx.z * kGroups * 3 * 64), 0, kGroups * 3 * 1024, 0x00020000);
#pragma unroll
      for (int G = 0; G < kGroups; ++G)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wview, (lds_void_t*)(smem + G * 1024), 16, 16u * (unsigned)lane, (G * 3 + 2) * 1024, 0, 0);
    }
#pragma unroll
    for (int G = 0; G < kGroups; ++G) {
      wreg[G][0].u = img[(G * 3 + 0) * 64];
      wreg[G][1].u = img[(G * 3 + 1) * 64];
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) band_store16(myring + (3 - c) * kBand16, f0[c], lane);   // stack channel c sits in ring slot 3 - c
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the LDS-DMA writes are through before the barrier
  }
  // the slice's 16 biases sit in LDS behind the rings (four registers the MFMA phase does not have)
  float* bias_lds = reinterpret_cast<float*>(smem + kGroups * 64 * 16 + kWaves * kWaveRing16);
  if (tid < 16) bias_lds[tid] = p.bias ? p.bias[co0 + tid] : 0.f;
  int aoff[kMT];
#pragma unroll
  for (int m = 0; m < kMT; ++m) {
    const int pix = m * 16 + j;
    const int oy = pix / kOW, ox = pix - oy * kOW;
    aoff[m] = ((oy * 4 + kq) * kIW + ox * 4) * 2;
  }
  // the lane's byte offset inside a row's [400][ld_out] output block (tile m: + a uniform m * 16 pixels)
  const unsigned ov0 = (unsigned)(((wave * 80 + j) * p.ld_out + co0 + 4 * kq) * 4);
  const int row_bytes = 400 * p.ld_out * 4, tile_bytes = 16 * p.ld_out * 4;
  __syncthreads();                                    // lo parts visible (and every wave's ring staged); the only workgroup barrier

  for (;;) {
    const int bn = __builtin_amdgcn_readfirstlane(b + (int)gridDim.x);
    const bool more = bn < p.B;
    f32x4_t acc[kMT];
#pragma unroll
    for (int m = 0; m < kMT; ++m) acc[m] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    auto group = [&](int G) {
      const unsigned char* base = myring + (3 - (G >> 1)) * kBand16 + (G & 1) * 4 * kIW * 2;
      Frag8 wlo;
      wlo.u = wlo_lds[G * 64 + lane];
      Frag8 xf[kMT];
#pragma unroll
      for (int m = 0; m < kMT; ++m) {
        typedef __attribute__((address_space(3))) const volatile unsigned long long lds_cv64_t;
        lds_cv64_t* src = (lds_cv64_t*)(base + aoff[m]);
        const unsigned long long x0 = src[0], x1 = src[1];
        xf[m].u = make_uint4((unsigned)x0, (unsigned)(x0 >> 32), (unsigned)x1, (unsigned)(x1 >> 32));
      }
#pragma unroll
      for (int m = 0; m < kMT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wlo.v, xf[m].v, acc[m], 0, 0, 0);
#pragma unroll
      for (int m = 0; m < kMT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wreg[G][1].v, xf[m].v, acc[m], 0, 0, 0);
#pragma unroll
      for (int m = 0; m < kMT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wreg[G][0].v, xf[m].v, acc[m], 0, 0, 0);
    };
    int nvn = 0;
    if (nv == 4 && more) {                             // the common case, straight-line: channel c's groups, then the next row's frame c
      nvn = nvalid_at(p.nvalid, bn);
      BandPrefetch pa = band_of(bn, 0), pb = band_of(bn, 1);
      // sched_barrier: the frame's conversion (plain VALU on the loaded registers) may not be scheduled above its
      // channel's k-groups -- hipcc hoisted it into the first one, i.e. the wave waited for the prefetch it had just issued
      group(0); group(1);
      __builtin_amdgcn_sched_barrier(0);
      wave_lds_fence(); band_store16(myring + 3 * kBand16, pa, lane);
      pa = band_of(bn, 2);
      group(2); group(3);
      __builtin_amdgcn_sched_barrier(0);
      wave_lds_fence(); band_store16(myring + 2 * kBand16, pb, lane);
      pb = band_of(bn, 3);
      group(4); group(5);
      __builtin_amdgcn_sched_barrier(0);
      wave_lds_fence(); band_store16(myring + 1 * kBand16, pa, lane);
      group(6); group(7);
      __builtin_amdgcn_sched_barrier(0);
      wave_lds_fence(); band_store16(myring + 0 * kBand16, pb, lane);
    } else {
      BandPrefetch pa, pb;
      if (more) {
        nvn = nvalid_at(p.nvalid, bn);
        pa = band_of(bn, 0); pb = band_of(bn, 1);
      }
#pragma unroll
      for (int G = 0; G < kGroups; ++G) if (G < 2 * nv) group(G);
      wave_lds_fence();
      if (more) {
        band_store16(myring + 3 * kBand16, pa, lane); band_store16(myring + 2 * kBand16, pb, lane);
        pa = band_of(bn, 2); pb = band_of(bn, 3);
        band_store16(myring + 1 * kBand16, pa, lane); band_store16(myring + 0 * kBand16, pb, lane);
      }
    }
    wave_lds_fence();
    const f32x4_t bias4 = *reinterpret_cast<const f32x4_t*>(bias_lds + 4 * kq);
    f32x4_t vout[kMT];
#pragma unroll
    for (int m = 0; m < kMT; ++m) {
      vout[m] = acc[m] + bias4;
      if (RELU) {
#pragma unroll
        for (int r = 0; r < 4; ++r) vout[m][r] = __builtin_amdgcn_fmed3f(vout[m][r], 0.f, __builtin_inff());
      }
    }
#pragma unroll
    for (int m = 0; m < kMT; ++m) asm volatile("" : "+v"(vout[m]));   // every output finished before the first store (see above)
    // the row's output block as a buffer: uniform base, the lane's loop-invariant 32-bit offsets (64-bit per-lane
    // pointers were spilled, and every store then waited for the one before it behind its address reload)
    const __amdgpu_buffer_rsrc_t oview = __builtin_amdgcn_make_buffer_rsrc(p.out + (long long)b * (row_bytes / 4), 0, row_bytes, 0x00020000);
#pragma unroll
    for (int m = 0; m < kMT; ++m)
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(su32x4_t, vout[m]), oview, ov0, m * tile_bytes, 0);
    if (!more) break;
    b = bn; nv = nvn;
  }
}

// ------------------------------------------------------------------------------------ //
// The rows forward on EIGHT waves (r6): stackconv_fwd_w8_kernel's organisation -- one 512-thread workgroup per CU, two
// rows at a time (waves 0-3 row 2 i, waves 4-7 row 2 i + 1), runs of 7 + 6 + 6 + 6 / 6 + 6 + 6 + 7 tiles, 28-row bands, all
// three planes of W / 255 in registers -- with the five-wave rows kernel's pipeline by stack channel (channel c's k-groups
// are the only readers of ring slot 3 - c: the next row's frame c is staged behind them from two sets of prefetch
// registers).  Same operands and MFMA order per accumulator: bit-identical outputs.
// ------------------------------------------------------------------------------------ //
template <int NT, bool RELU>
__device__ __forceinline__ void rows8_run(const RowsParams& p, unsigned char* myring, const float* bias_lds, const Frag8 (&wreg)[kGroups][3],
                                          int col, int p0, int lane) {
  const int kq = lane >> 4, j = lane & 15;
  const int co0 = blockIdx.z * 16;
  const int row0 = 4 * (p0 / kOW);
  const unsigned band0 = (unsigned)__builtin_amdgcn_readfirstlane(row0 * kIW);
  int aoff[NT];
  unsigned ov[NT];
#pragma unroll
  for (int m = 0; m < NT; ++m) {
    const int pix = p0 + m * 16 + j;
    const int oy = pix / kOW, ox = pix - oy * kOW;
    aoff[m] = ((oy * 4 - row0 + kq) * kIW + ox * 4) * 2;
    ov[m] = (unsigned)((pix * p.ld_out + co0 + 4 * kq) * 4);
  }
  // frame c of row b as a buffer of its own (see stackconv_rows_kernel); a band that reaches past the frame reads zeros
  const unsigned fv[3] = {16u * (unsigned)lane, 16u * (unsigned)(lane + 64), lane + 128 < kW8Vec ? 16u * (unsigned)(lane + 128) : 0x80000000u};
  auto band_of = [&](int b, int c) -> Band3 {
    typedef __attribute__((address_space(4))) const unsigned cu32_t;
    long long row = b;
    if (c) {
      cu32_t* q = reinterpret_cast<cu32_t*>(reinterpret_cast<uintptr_t>(p.hist_rows + 4 * (long long)b + c));
      row = (long long)(((unsigned long long)q[1] << 32) | q[0]);
    }
    const uint8_t* fr = (c == 0 ? p.obs : p.store_obs) + row * p.fsz;
    const __amdgpu_buffer_rsrc_t v = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(fr), 0, p.fsz, 0x00020000);
    Band3 r;
#pragma unroll
    for (int u = 0; u < 3; ++u) r.v[u] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(v, fv[u], band0, 0));
    return r;
  };
  auto band_put = [&](unsigned char* slot, const Band3& r) {
    uint4* dst = reinterpret_cast<uint4*>(slot);
    uint4 a, b;
    cvt16(r.v[0], a, b); dst[2 * lane] = a; dst[2 * lane + 1] = b;
    cvt16(r.v[1], a, b); dst[2 * (lane + 64)] = a; dst[2 * (lane + 64) + 1] = b;
    if (lane + 128 < kW8Vec) { cvt16(r.v[2], a, b); dst[2 * (lane + 128)] = a; dst[2 * (lane + 128) + 1] = b; }
  };
  const int row_bytes = 400 * p.ld_out * 4;
  const int stride = 2 * (int)gridDim.x;
  int b = __builtin_amdgcn_readfirstlane(2 * (int)blockIdx.x + col);
  if (b >= p.B) return;                               // odd B: the last pair has one row (wave-uniform; no barrier follows)
  int nv = nvalid_at(p.nvalid, b);
  {
    Band3 f0[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) f0[c] = band_of(b, c);
#pragma unroll
    for (int c = 0; c < 4; ++c) band_put(myring + (3 - c) * kW8Band16, f0[c]);     // stack channel c sits in ring slot 3 - c
    wave_lds_fence();
  }
  for (;;) {
    const int bn = __builtin_amdgcn_readfirstlane(b + stride);
    const bool more = bn < p.B;
    f32x4_t acc[NT];
#pragma unroll
    for (int m = 0; m < NT; ++m) acc[m] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    auto group = [&](int G) {
      const unsigned char* base = myring + (3 - (G >> 1)) * kW8Band16 + (G & 1) * 4 * kIW * 2;
      Frag8 xf[NT];
#pragma unroll
      for (int m = 0; m < NT; ++m) {
        typedef __attribute__((address_space(3))) const volatile unsigned long long lds_cv64_t;
        lds_cv64_t* src = (lds_cv64_t*)(base + aoff[m]);
        const unsigned long long x0 = src[0], x1 = src[1];
        xf[m].u = make_uint4((unsigned)x0, (unsigned)(x0 >> 32), (unsigned)x1, (unsigned)(x1 >> 32));
      }
#pragma unroll
      for (int s3 = 2; s3 >= 0; --s3)
#pragma unroll
        for (int m = 0; m < NT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wreg[G][s3].v, xf[m].v, acc[m], 0, 0, 0);
    };
    int nvn = 0;
    if (nv == 4 && more) {                             // the common case, straight-line: channel c's groups, then the next row's frame c
      nvn = nvalid_at(p.nvalid, bn);
      Band3 pa = band_of(bn, 0), pb = band_of(bn, 1);
      group(0); group(1);
      __builtin_amdgcn_sched_barrier(0);               // (the conversion may not be scheduled above its channel's k-groups: see the five-wave kernel)
      wave_lds_fence(); band_put(myring + 3 * kW8Band16, pa);
      pa = band_of(bn, 2);
      group(2); group(3);
      __builtin_amdgcn_sched_barrier(0);
      wave_lds_fence(); band_put(myring + 2 * kW8Band16, pb);
      pb = band_of(bn, 3);
      group(4); group(5);
      __builtin_amdgcn_sched_barrier(0);
      wave_lds_fence(); band_put(myring + 1 * kW8Band16, pa);
      group(6); group(7);
      __builtin_amdgcn_sched_barrier(0);
      wave_lds_fence(); band_put(myring + 0 * kW8Band16, pb);
    } else {
      Band3 pa, pb;
      if (more) {
        nvn = nvalid_at(p.nvalid, bn);
        pa = band_of(bn, 0); pb = band_of(bn, 1);
      }
#pragma unroll
      for (int G = 0; G < kGroups; ++G) if (G < 2 * nv) group(G);
      wave_lds_fence();
      if (more) {
        band_put(myring + 3 * kW8Band16, pa); band_put(myring + 2 * kW8Band16, pb);
        pa = band_of(bn, 2); pb = band_of(bn, 3);
        band_put(myring + 1 * kW8Band16, pa); band_put(myring + 0 * kW8Band16, pb);
      }
    }
    wave_lds_fence();
    const f32x4_t bias4 = *reinterpret_cast<const f32x4_t*>(bias_lds + 4 * kq);
    f32x4_t vout[NT];
#pragma unroll
    for (int m = 0; m < NT; ++m) {
      vout[m] = acc[m] + bias4;
      if (RELU) {
#pragma unroll
        for (int r = 0; r < 4; ++r) vout[m][r] = __builtin_amdgcn_fmed3f(vout[m][r], 0.f, __builtin_inff());
      }
    }
#pragma unroll
    for (int m = 0; m < NT; ++m) asm volatile("" : "+v"(vout[m]));   // every output finished before the first store
    const __amdgpu_buffer_rsrc_t oview = __builtin_amdgcn_make_buffer_rsrc(p.out + (long long)b * (row_bytes / 4), 0, row_bytes, 0x00020000);
#pragma unroll
    for (int m = 0; m < NT; ++m)
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(su32x4_t, vout[m]), oview, ov[m], 0, 0);
    if (!more) break;
    b = bn; nv = nvn;
  }
}

template <bool RELU>
__global__ void __launch_bounds__(kW8Threads) __attribute__((amdgpu_waves_per_eu(2, 2)))
stackconv_rows_w8_kernel(const RowsParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int co0 = blockIdx.z * 16;
  Frag8 wreg[kGroups][3];
  {
    const uint4* img = p.w_split + (long long)blockIdx.z * kGroups * 3 * 64 + lane;
#pragma unroll
    for (int G = 0; G < kGroups; ++G)
#pragma unroll
      for (int s3 = 0; s3 < 3; ++s3) wreg[G][s3].u = img[(G * 3 + s3) * 64];
  }
  float* bias_lds = reinterpret_cast<float*>(smem + kW8Rings);
  if (tid < 16) bias_lds[tid] = p.bias ? p.bias[co0 + tid] : 0.f;
  __syncthreads();                                    // the biases; the only workgroup barrier
  const int col = wave >> 2, q = wave & 3;
  const int p0 = 96 * q + ((col == 0 && q > 0) ? 16 : 0);
  unsigned char* myring = smem + wave * kW8Ring;
  if (wave == 0 || wave == 7) rows8_run<7, RELU>(p, myring, bias_lds, wreg, col, p0, lane);
  else rows8_run<6, RELU>(p, myring, bias_lds, wreg, col, p0, lane);
}

// (r1-r4: the channel-per-wave / channel-pair weight-gradient kernels lived here; r5: the transposing-read kernel below)
constexpr int kFrame16 = kIH * kIW * 2;                  // 14112 B: one frame in bf16
constexpr int kFrameSlots = 5;
__device__ __forceinline__ void frame_store16(unsigned char* slot, const uint4& v, int idx) {   // 16 bytes -> 16 bf16
  uint4 a, b;
  cvt16(v, a, b);
  reinterpret_cast<uint4*>(slot)[2 * idx] = a;
  reinterpret_cast<uint4*>(slot)[2 * idx + 1] = b;
}

// ------------------------------------------------------------------------------------ //
// bf16x3 weight gradient through TRANSPOSING LDS reads (r5).
// The channel-pair kernel of r2-r4 built every MFMA A operand -- eight pixels of one kernel position -- from sixteen
// 8-byte LDS reads and sixteen v_perm per channel and group, and loads dY with 4-byte global loads: ~6 VALU per MFMA
// on a matrix pipe that was 0.36 busy (125 us in rocprofv3; this kernel: 123.5 us at 2.9 VALU per MFMA, matrix pipe 0.38 busy,
// and 134-138 us against 140-145 inside the cfg2 step, where its 16-byte dY loads contend less).  ds_read_b64_tr_b16 (wgx.h) hands a lane the four k-values (pixels) of one
// column of a [4 pixels][16 columns] block whose rows are addressed per lane: with the frames as bf16 rows in LDS
// (what the ring already holds), the 16 columns of a block are kernel positions (ky, kx = 0..7), (ky + 1, kx = 0..7)
// -- two runs of 16 contiguous bytes of two frame rows -- and the rows are output pixels (their windows start 4 input
// pixels = 8 bytes apart: neighbouring pixels' reads overlap and broadcast).  No permutes, no gathers:
//   * dW[(ky, kx, c), co] = sum over steps and pixels of X[pixel; ky, kx, c] dY[pixel, co]: MFMA 16 x 16 x 32, rows =
//     16 kernel positions of one stack channel (tile = (c, ky pair)), columns = 16 output channels, reduction = 32
//     pixels; pixels are exact in bf16, dY is split ONCE per element into three bf16 planes [pixel][16] in LDS (16-byte
//     items from registers requested a step ahead) and read through the same transposing reads;
//   * 8 waves: wave (c = w & 3, wk = w >> 2) owns the four tiles of stack channel c (16 accumulator registers for the
//     whole launch) and the k-steps of parity wk; per k-step 6 reads for dY's planes, 8 for the frames, 12 MFMAs;
//   * double-buffered dY planes: the planes of step t + 1 are written (and frame t + 4 converted into the ring) in the
//     same barrier interval in which step t is multiplied -- waves 0-3 prepare first and multiply second, waves 4-7
//     the other way round, so a SIMD's two waves want different pipes; ONE barrier per step;
//   * a stack channel outside the episode (c >= nvalid[t, b]) skips its wave's MFMAs of that step.
// LDS: 5 frame slots (70.6 KB) + 2 x 39.9 KB of dY planes: one 8-wave workgroup per CU.
// ------------------------------------------------------------------------------------ //
constexpr int kTrKS = 13;                                 // 32-pixel k-steps of a 400-pixel frame (416: 16 zero pixels)
constexpr int kTrYPlane = kTrKS * 32 * 32;                // bytes of one bf16 plane of dY: [416 pixels][16 channels]
constexpr int kTrYBuf = 3 * kTrYPlane;                    // 39 936
constexpr int kTrLds = kFrameSlots * kFrame16 + 2 * kTrYBuf;   // 150 432

typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef short s16x8_t __attribute__((ext_vector_type(8)));
__device__ __forceinline__ bf16x8_t tr_operand(const unsigned char* base, unsigned o0, unsigned o1) {
  typedef __attribute__((address_space(3))) s16x4_t lds_s16x4_t;
  const s16x4_t a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(base + o0));
  const s16x4_t b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(base + o1));
  const s16x8_t v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  return __builtin_bit_cast(bf16x8_t, v);
}

template <int LD, int EXP = 0>                           // row stride of dY (floats) when it is 16 or 32, else 0 = run time
__global__ void __launch_bounds__(512, 2)                // EXP (tools/probes/stack_wgrad_probe.hip only): 32 = s_memtime stamps
stackconv_wgrad_tr_kernel(const Params p) {
  const int ld_out = LD ? LD : p.ld_out;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* ybuf = smem + kFrameSlots * kFrame16;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = wave & 3, wk = wave >> 2;                 // stack channel; parity of this wave's k-steps
  const int g = lane >> 4, c16 = lane & 15, jrow = c16 >> 2, q = c16 & 3;
  const int co0 = blockIdx.z * 16;
  constexpr int P = 400, kVec = kIH * kIW / 16;           // 441 uint4 per uint8 frame
  constexpr int kKPW = (kTrKS + 1) / 2;                   // k-steps per wave (7 / 6)

  for (int i = tid * 16; i < 2 * kTrYBuf; i += 512 * 16) *reinterpret_cast<uint4*>(ybuf + i) = make_uint4(0, 0, 0, 0);

  // per-lane operand offsets of this wave's k-steps (step independent): pixel of element j of read rd in 16-lane group
  // g is 16 (g >> 1) + 8 rd + 4 (g & 1) + j (a half-wave covers 8 consecutive pixels: no bank conflicts)
  unsigned xb[kKPW][2], yb[kKPW][2];
#pragma unroll
  for (int jj = 0; jj < kKPW; ++jj)
#pragma unroll
    for (int rd = 0; rd < 2; ++rd) {
      const int pix = (wk + 2 * jj) * 32 + 16 * (g >> 1) + 8 * rd + 4 * (g & 1) + jrow;
      const int pc = pix < P ? pix : P - 1;               // (padding pixels: dY's zero rows; X at the last real pixel)
      const int oy = pc / kOW, ox = pc - oy * kOW;
      // lanes q = 0, 1 supply kernel row 2 kyp (columns kx = 0..3 / 4..7), q = 2, 3 the row below
      xb[jj][rd] = (unsigned)(((4 * oy + (q >> 1)) * kIW + 4 * ox + 4 * (q & 1)) * 2);
      yb[jj][rd] = (unsigned)(pix * 32 + q * 8);
    }

  f32x4_t acc[4];
#pragma unroll
  for (int t4 = 0; t4 < 4; ++t4) acc[t4] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  f32x4_t bsum = {0.f, 0.f, 0.f, 0.f};

  // dY of one step: 1600 16-byte items (pixel = item >> 2, channel quad = item & 3), four per thread (the last partly)
  // Requests are asm statements (hipcc's s_waitcnt bookkeeping drained the queue at every step head and waited for a
  // step's own requests inside the step: s_memtime stamps, tools/probes/stack_wgrad_probe.hip); the waits count the
  // requests issued behind the one that is needed (vector memory operations retire in order) and the registers are read
  // behind the wait only (take4).  dY and the frames are buffers: a uniform 32-bit step offset + a loop-invariant lane
  // offset; a request that is not needed (past the chunk) goes out of range and costs no traffic.
  typedef unsigned sgpr128_t __attribute__((ext_vector_type(4)));
  auto words_of = [](const void* ptr, unsigned long long bytes) -> sgpr128_t {
    const uint64_t ab = reinterpret_cast<uint64_t>(ptr);
    return sgpr128_t{(unsigned)__builtin_amdgcn_readfirstlane((unsigned)ab), (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(ab >> 32)) & 0xFFFFu,
                     (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(bytes > 0xFFFFFFFFull ? 0xFFFFFFFFull : bytes)), 0x00020000u};
  };
  const sgpr128_t ywords = words_of(p.dy, (unsigned long long)p.T1 * p.B * P * ld_out * 4ull);
  const sgpr128_t fwords = words_of(p.frames_ext, (unsigned long long)(3 + p.T1) * p.B * p.fsz);
  constexpr unsigned kOob = 0xFFFFFFFFu;                  // >= any num_records: the request returns zeros without touching memory
  unsigned yoff[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int it = tid + 512 * j;
    yoff[j] = (j < 3 || it < 4 * P) ? (unsigned)(((it >> 2) * ld_out + 4 * (it & 3) + co0) * 4) : kOob;
  }
  const unsigned foff = tid < kVec ? 16u * (unsigned)tid : kOob;
  f32x4_t ly[4], lf;
  auto req_dy = [&](int t, int b, bool valid) {
    const unsigned so = __builtin_amdgcn_readfirstlane((unsigned)(((long long)t * p.B + b) * P * ld_out) * 4u);
    const unsigned kill = valid ? 0u : kOob;
    asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %4, %8, %9 offen\n\tbuffer_load_dwordx4 %1, %5, %8, %9 offen\n\t"
                 "buffer_load_dwordx4 %2, %6, %8, %9 offen\n\tbuffer_load_dwordx4 %3, %7, %8, %9 offen"
                 : "=&v"(ly[0]), "=&v"(ly[1]), "=&v"(ly[2]), "=&v"(ly[3])
                 : "v"(yoff[0] | kill), "v"(yoff[1] | kill), "v"(yoff[2] | kill), "v"(yoff[3] | kill), "s"(ywords), "s"(so)
                 : "memory");
  };
  auto req_frame = [&](int e, int b, bool valid) {
    const unsigned so = __builtin_amdgcn_readfirstlane((unsigned)(((long long)e * p.B + b) * p.fsz));
    asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2, %3 offen" : "=&v"(lf) : "v"(foff | (valid ? 0u : kOob)), "s"(fwords), "s"(so) : "memory");
  };
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  auto take4 = [&](const f32x4_t& r, auto n_c) -> f32x4_t {   // wait until at most N younger requests are outstanding, then copy
    constexpr int N = decltype(n_c)::value;
    f32x2_t lo, hi;
    asm volatile("s_waitcnt vmcnt(%4)\n\tv_mov_b64 %0, %2\n\tv_mov_b64 %1, %3" : "=&v"(lo), "=&v"(hi)
                 : "v"(__builtin_shufflevector(r, r, 0, 1)), "v"(__builtin_shufflevector(r, r, 2, 3)), "n"(N));
    return f32x4_t{lo[0], lo[1], hi[0], hi[1]};
  };
  f32x4_t ty[4];                                          // dY items behind their wait (take_dy), split by put_dy
  auto take_dy = [&](auto n_c) {                          // N = requests issued behind the four dY requests
    ty[0] = take4(ly[0], n_c);
#pragma unroll
    for (int j = 1; j < 4; ++j) ty[j] = take4(ly[j], n_c);
  };
  auto put_dy = [&](unsigned char* dst) {                 // registers -> three planes (split by truncation, exact)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int it = tid + 512 * j;
      if (j < 3 || it < 4 * P) {
        const f32x4_t v = ty[j];
        bsum += v;
        uint32_t h[4], m[4], l[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          h[e] = __float_as_uint(v[e]) & 0xFFFF0000u;
          const float r1 = v[e] - __uint_as_float(h[e]);
          m[e] = __float_as_uint(r1) & 0xFFFF0000u;
          l[e] = __float_as_uint(r1 - __uint_as_float(m[e]));
        }
        unsigned char* d = dst + it * 8;
        *reinterpret_cast<uint2*>(d) = make_uint2(__builtin_amdgcn_perm(h[1], h[0], 0x07060302u), __builtin_amdgcn_perm(h[3], h[2], 0x07060302u));
        *reinterpret_cast<uint2*>(d + kTrYPlane) = make_uint2(__builtin_amdgcn_perm(m[1], m[0], 0x07060302u), __builtin_amdgcn_perm(m[3], m[2], 0x07060302u));
        *reinterpret_cast<uint2*>(d + 2 * kTrYPlane) = make_uint2(__builtin_amdgcn_perm(l[1], l[0], 0x07060302u), __builtin_amdgcn_perm(l[3], l[2], 0x07060302u));
      }
    }
  };
  // the MFMAs of one step: frames of ext rows t .. t + 3 in the ring, dY planes in `yp`; the operands of k-step jj + 1 are
  // requested before the twelve MFMAs of k-step jj (pinned)
  auto multiply = [&](auto wkc, int t, const unsigned char* yp) {
    constexpr int WK = decltype(wkc)::value;              // (compile time: the k-step list is static, no branches inside)
    const unsigned char* frame = smem + ((t + 3 - c) % kFrameSlots) * kFrame16;
    bf16x8_t bf[2][3], xa[2][4];
    auto fetch = [&](int jj, bf16x8_t (&b3)[3], bf16x8_t (&a4)[4]) {
#pragma unroll
      for (int s3 = 0; s3 < 3; ++s3) b3[s3] = tr_operand(yp, yb[jj][0] + s3 * kTrYPlane, yb[jj][1] + s3 * kTrYPlane);
#pragma unroll
      for (int kyp = 0; kyp < 4; ++kyp) a4[kyp] = tr_operand(frame, xb[jj][0] + kyp * (2 * kIW * 2), xb[jj][1] + kyp * (2 * kIW * 2));
    };
    fetch(0, bf[0], xa[0]);
#pragma unroll
    for (int jj = 0; jj < kKPW; ++jj) {
      if (WK + 2 * jj < kTrKS) {
        if (jj + 1 < kKPW && WK + 2 * (jj + 1) < kTrKS) fetch(jj + 1, bf[(jj + 1) & 1], xa[(jj + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s3 = 2; s3 >= 0; --s3)                   // lo, mid, hi; the four accumulators alternate
#pragma unroll
          for (int kyp = 0; kyp < 4; ++kyp)
            acc[kyp] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa[jj & 1][kyp], bf[jj & 1][s3], acc[kyp], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };

  // EXP & 32: stamps of workgroup 0's waves into partial_b as [wave][step][8] words (the probe passes a buffer of its own)
  unsigned* stamps = reinterpret_cast<unsigned*>(p.partial_b) + (size_t)wave * 64 * 8;
  auto stamp = [&](int t, int k) {
    if ((EXP & 32) && blockIdx.x == 0 && t < 64) {
      const unsigned long long now = __builtin_readcyclecounter();
      if (lane == 0) stamps[t * 8 + k] = (unsigned)now;
    }
  };
  auto run = [&](auto wkc) {
    constexpr int WK = decltype(wkc)::value;
    for (int item = blockIdx.x; item < p.items; item += gridDim.x) {
      const int b = item % p.B, chunk = item / p.B;
      const int t0 = chunk * p.spc;
      const int t1 = (t0 + p.spc < p.T1) ? t0 + p.spc : p.T1;
      __syncthreads();                                      // previous item's last step is done with the ring and the planes
      for (int e = 0; e < 4; ++e) {                         // ext rows t0 .. t0 + 3 -> slots
        const uint4* src = reinterpret_cast<const uint4*>(p.frames_ext + ((long long)(t0 + e) * p.B + b) * p.fsz);
        for (int idx = tid; idx < kVec; idx += 512) frame_store16(smem + ((t0 + e) % kFrameSlots) * kFrame16, src[idx], idx);
      }
      typedef std::integral_constant<int, 0> n0_t;
      typedef std::integral_constant<int, 1> n1_t;
      typedef std::integral_constant<int, 4> n4_t;
      req_dy(t0, b, true);
      take_dy(n0_t());
      put_dy(ybuf + (t0 & 1) * kTrYBuf);
      req_dy(t0 + 1, b, t0 + 1 < t1);
      __syncthreads();
      for (int t = t0; t < t1; ++t) {
        const bool more = t + 1 < t1;
        const int nv = nvalid_at(p.nvalid, (long long)t * p.B + b);
        const unsigned char* yp = ybuf + (t & 1) * kTrYBuf;
        // Per step every wave issues ONE frame request (ext row t + 4) and four dY requests (step t + 2) -- out of range behind
        // the chunk, where they cost no traffic -- and takes one of each: every count below is unconditional (no wait depends
        // on a branch taken earlier; tools/isa_inflight.py --cfg follows every path of the compiled loop), no wait of a step
        // covers a request of that step, and the LAST dY request is taken behind the loop: a request left in flight at the
        // loop's exit lands in registers hipcc has handed to the epilogue by then (an out-of-range one writes zeros: a
        // pointer of the slice stores went to nil once in seven runs of the first build, "Memory access fault ... (nil)").
        //   waves 0-3: take dY(t + 1) [vmcnt(0): requested before the previous step's MFMAs], planes, request frame + dY,
        //              MFMAs, take the frame [vmcnt(4): the dY requests behind it stay in flight across the barrier]
        //   waves 4-7: request the frame, MFMAs, take dY(t + 1) [vmcnt(1): requested at the end of the previous step], planes,
        //              take the frame [vmcnt(0)], request dY
        stamp(t, 0);
        if (WK == 0 || (EXP & 64)) {                        // EXP & 64 (probe): every wave prepares first
          take_dy(n0_t());
          if (more) put_dy(ybuf + ((t + 1) & 1) * kTrYBuf);
          req_frame(t + 4, b, more);
          req_dy(t + 2, b, t + 2 < t1);
          stamp(t, 1);
          if (c < nv) multiply(wkc, t, yp);
          stamp(t, 2);
          const f32x4_t fv = take4(lf, n4_t());
          if (more && tid < kVec) frame_store16(smem + ((t + 4) % kFrameSlots) * kFrame16, __builtin_bit_cast(uint4, fv), tid);   // a slot no wave reads in step t
        } else {
          req_frame(t + 4, b, more);
          if (c < nv) multiply(wkc, t, yp);
          stamp(t, 1);
          take_dy(n1_t());
          if (more) put_dy(ybuf + ((t + 1) & 1) * kTrYBuf);
          stamp(t, 2);
          const f32x4_t fv = take4(lf, n0_t());
          if (more && tid < kVec) frame_store16(smem + ((t + 4) % kFrameSlots) * kFrame16, __builtin_bit_cast(uint4, fv), tid);
          req_dy(t + 2, b, t + 2 < t1);
        }
        stamp(t, 3);
        // planes / frame of step t + 1 visible, every wave done with step t.  NOT __syncthreads(): its fence would drain the
        // requests in flight; the LDS writes of this wave are what the others need
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        stamp(t, 4);
      }
      take_dy(std::integral_constant<int, 0>());            // the out-of-range request of the last step: nothing stays in flight
    }
  };
  if (wk == 0) run(std::integral_constant<int, 0>()); else run(std::integral_constant<int, 1>());
  // (belt and braces: nothing is in flight here -- see the step loop --, and the request registers stay live up to this point)
  asm volatile("s_waitcnt vmcnt(0)" :: "v"(ly[0]), "v"(ly[1]), "v"(ly[2]), "v"(ly[3]), "v"(lf) : "memory");

  // ---- the two k-step parities of a channel -> one tile set per channel (parity 1 through LDS), straight into the slice
  float* red = reinterpret_cast<float*>(smem);            // [c][tile][lane][4]
  if (wk == 1) {
#pragma unroll
    for (int kyp = 0; kyp < 4; ++kyp) *reinterpret_cast<f32x4_t*>(red + ((c * 4 + kyp) * 64 + lane) * 4) = acc[kyp];
  }
  __syncthreads();
  if (wk == 0) {
    float* pw = p.partial_w + (long long)blockIdx.x * 256 * p.cout;
#pragma unroll
    for (int kyp = 0; kyp < 4; ++kyp) {
      const f32x4_t o = acc[kyp] + *reinterpret_cast<const f32x4_t*>(red + ((c * 4 + kyp) * 64 + lane) * 4);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = 4 * g + r;                          // row of the tile: kernel position (2 kyp + m / 8, m % 8)
        const int ky = 2 * kyp + (m >> 3), kx = m & 7;
        pw[((ky * 8 + kx) * 4 + c) * p.cout + co0 + c16] = o[r] / 255.0f;
      }
    }
  }
  if (p.partial_b && !(EXP & 32)) {                       // thread t summed output channels 4 (t & 3) .. + 3 of its dY items
    __syncthreads();
    f32x4_t* redb = reinterpret_cast<f32x4_t*>(smem + 16384);
    redb[tid] = bsum;
    __syncthreads();
    if (tid < 16) {
      float sacc = 0.f;
      for (int u = tid >> 2; u < 512; u += 4) sacc += reinterpret_cast<const float*>(redb + u)[tid & 3];
      p.partial_b[(long long)blockIdx.x * p.cout + co0 + tid] = sacc;
    }
  }
}

// ------------------------------------------------------------------------------------ //
// Host side: eligibility, work decomposition, launch.
// ------------------------------------------------------------------------------------ //
bool eligible(const seedhip_stack_conv_geom* g, const void* frames_ext, const void* io) {
  const long long fsz = (long long)g->ih * g->iw;
  (void)fsz;
  return g->kh == 8 && g->kw == 8 && g->stride == 4 && g->ih == kIH && g->iw == kIW && g->oh == 20 && g->ow == 20 &&
         g->cout % 16 == 0 && g->ld_out % 4 == 0 && (((uintptr_t)frames_ext) & 15) == 0 && (((uintptr_t)io) & 15) == 0 &&
         (long long)g->T * g->B * g->oh * g->ow * g->ld_out < (1LL << 30);      // 32-bit BYTE offsets into the conv output
}

// Chooses the time chunking so that the persistent grid is evenly loaded.
void decompose(int T1, int B, int max_grid, int* spc, int* items, int* grid) {
  double best = -1.0;
  int best_n = 1;
  for (int n = 1; n <= T1; ++n) {
    const int s = (T1 + n - 1) / n;
    if ((T1 + s - 1) / s != n) continue;                   // n must be the real chunk count for this s
    const long long it = (long long)B * n;
    const long long gr = it < max_grid ? it : max_grid;
    const double rounds = (double)((it + gr - 1) / gr);
    const double balance = (double)it / (rounds * gr);
    const double fill = (double)gr / max_grid;             // prefer using the whole chip
    const double overhead = (double)s / (s + 0.75);        // 3 extra frame loads per chunk
    const double score = balance * (fill < 1.0 ? fill : 1.0) * overhead;
    if (score > best + 1e-9) { best = score; best_n = n; }
  }
  *spc = (T1 + best_n - 1) / best_n;
  const long long it = (long long)B * best_n;
  *items = (int)it;
  *grid = (int)(it < max_grid ? it : max_grid);
}

// MI355X: 256 CUs (this library targets gfx950 only); a fixed number keeps the workspace
// query a pure function of the geometry.
int max_grid_for(int wgs_per_cu) { return 256 * wgs_per_cu; }

Params make_params(const seedhip_stack_conv_geom* g, const uint8_t* frames_ext, const uint8_t* nvalid) {
  Params p;
  memset(&p, 0, sizeof(p));
  p.frames_ext = frames_ext; p.nvalid = nvalid;
  p.T1 = g->T; p.B = g->B; p.ih = g->ih; p.iw = g->iw; p.oh = g->oh; p.ow = g->ow; p.cout = g->cout;
  p.ld_out = g->ld_out; p.fsz = g->ih * g->iw;
  return p;
}

bool bf16x3_enabled() {
  static const int bf16x3 = getenv("SEEDHIP_STACK_BF16") ? atoi(getenv("SEEDHIP_STACK_BF16")) : 1;
  return bf16x3 != 0;
}

bool w8_enabled() {                                    // SEEDHIP_STACK_W8=0: the five-wave forward (A/B)
  static const int w8 = getenv("SEEDHIP_STACK_W8") ? atoi(getenv("SEEDHIP_STACK_W8")) : 1;
  return w8 != 0;
}

int launch_fwd(const seedhip_stack_conv_geom* g, const uint8_t* frames_ext, const uint8_t* nvalid, const float* w,
               const float* bias, float* out, int out_relu, hipStream_t s, unsigned char* relu_bits = nullptr) {
  Params p = make_params(g, frames_ext, nvalid);
  p.w = w; p.bias = bias; p.out = out; p.out_relu = out_relu; p.relu_bits = relu_bits;
  const bool bf16x3 = bf16x3_enabled();
  if (relu_bits && !bf16x3) return fail(SEEDHIP_ERR_UNSUPPORTED, "conv2d_stack_fwd_bits: only the bf16x3 kernel writes the byte mask");
  if (bf16x3) {
    const size_t lds = (size_t)kGroups * 64 * 16 + (size_t)kWaves * kWaveRing16;
    int grid;
    decompose(p.T1, p.B, max_grid_for(2), &p.spc, &p.items, &grid);
#define SEEDHIP_SCF(BITS_, RELU_, BUF_)                                                                           \
    {                                                                                                             \
      (void)hipFuncSetAttribute((const void*)stackconv_fwd_bf16r_kernel<0, BITS_, RELU_, BUF_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
      hipLaunchKernelGGL((stackconv_fwd_bf16r_kernel<0, BITS_, RELU_, BUF_>), dim3(grid, 1, g->cout / 16), dim3(kThreads), lds, s, p); \
      return check_launch("stackconv_fwd_bf16r_kernel");                                                          \
    }
    constexpr int buf_on = 1;
    const long long lim = (1LL << 31) - (1 << 20);
    p.buf32 = buf_on && (long long)(3 + p.T1) * p.B * p.fsz < lim && (long long)p.T1 * p.B * 400 * p.ld_out * 4 < lim;
    if (p.buf32 && w8_enabled()) {                     // eight waves, two columns per workgroup, one workgroup per CU
      int grid8;
      decompose(p.T1, (p.B + 1) / 2, max_grid_for(1), &p.spc, &p.items, &grid8);
#define SEEDHIP_SCF8(BITS_, RELU_)                                                                                 \
      {                                                                                                           \
        (void)hipFuncSetAttribute((const void*)stackconv_fwd_w8_kernel<0, BITS_, RELU_, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, kW8Lds); \
        hipLaunchKernelGGL((stackconv_fwd_w8_kernel<0, BITS_, RELU_, 2>), dim3(grid8, 1, g->cout / 16), dim3(kW8Threads), kW8Lds, s, p); \
        return check_launch("stackconv_fwd_w8_kernel");                                                           \
      }
      if (relu_bits) SEEDHIP_SCF8(true, true)
      if (out_relu) SEEDHIP_SCF8(false, true)
      SEEDHIP_SCF8(false, false)
#undef SEEDHIP_SCF8
    }
    if (relu_bits) { if (p.buf32) SEEDHIP_SCF(true, true, true) else SEEDHIP_SCF(true, true, false) }
    if (p.buf32) { if (out_relu) SEEDHIP_SCF(false, true, true) else SEEDHIP_SCF(false, false, true) }
    if (out_relu) SEEDHIP_SCF(false, true, false)
    SEEDHIP_SCF(false, false, false)
#undef SEEDHIP_SCF
  }
  const size_t lds = kWFloats * sizeof(float) + (size_t)kWaves * kWaveRing;
  const int per_cu = (int)((160 * 1024) / lds) < 3 ? (int)((160 * 1024) / lds) : 3;
  int grid;
  decompose(p.T1, p.B, max_grid_for(per_cu < 1 ? 1 : per_cu), &p.spc, &p.items, &grid);
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute((const void*)stackconv_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(stackconv_fwd_kernel, dim3(grid, 1, g->cout / 16), dim3(kThreads), lds, s, p);
  return check_launch("stackconv_fwd_kernel");
}

size_t wgrad_lds(int fsz) { (void)fsz; return (kWFloats + kWaves * 16) * sizeof(float) + (size_t)kWaves * kWaveRing; }

bool wgrad_tr_fits(const seedhip_stack_conv_geom* g) {
  const long long lim = (1LL << 32) - (1 << 20);
  return (long long)(3 + g->T) * g->B * g->ih * g->iw < lim && (long long)g->T * g->B * g->oh * g->ow * g->ld_out * 4 < lim;
}

int wgrad_grid(const seedhip_stack_conv_geom* g, int* spc, int* items) {
  const size_t lds = wgrad_lds(g->ih * g->iw);
  int per_cu = (int)((160 * 1024) / lds);
  if (per_cu > 2) per_cu = 2;                     // measured: 2 workgroups per CU beat 3 (0.43 vs 0.48 ms at cfg2)
  if (per_cu < 1) per_cu = 1;
  static const int bf16x3 = getenv("SEEDHIP_STACK_BF16") ? atoi(getenv("SEEDHIP_STACK_BF16")) : 1;
  if (bf16x3 && wgrad_tr_fits(g)) per_cu = 1;         // transposing-read kernel: 150 KB of LDS, one 8-wave workgroup per CU
  int grid;
  decompose(g->T, g->B, max_grid_for(per_cu), spc, items, &grid);
  return grid;
}

}  // namespace stackconv
}  // namespace seedhip

using namespace seedhip;

namespace {
int check_stack(const seedhip_stack_conv_geom* g, const char* what) {
  SEEDHIP_REQUIRE(g, "%s: null geometry", what);
  SEEDHIP_REQUIRE(g->T >= 1 && g->B >= 1 && g->ih >= 1 && g->iw >= 1 && g->kh >= 1 && g->kw >= 1 && g->stride >= 1 &&
                  g->cout >= 1, "%s: non-positive geometry field", what);
  SEEDHIP_REQUIRE(g->oh == (g->ih - g->kh) / g->stride + 1 && g->ow == (g->iw - g->kw) / g->stride + 1,
                  "%s: oh/ow must be the VALID-padding output size", what);
  SEEDHIP_REQUIRE(g->ld_out >= g->cout, "%s: ld_out < cout", what);
  SEEDHIP_REQUIRE((long long)g->T * g->B * g->oh * g->ow < (1LL << 31), "%s: more than 2^31 pixels", what);
  return SEEDHIP_OK;
}
StackGeom to_stack(const seedhip_stack_conv_geom* g) {
  StackGeom s; s.T = g->T; s.B = g->B; s.ih = g->ih; s.iw = g->iw; s.oh = g->oh; s.ow = g->ow; s.kh = g->kh;
  s.kw = g->kw; s.stride = g->stride; s.cout = g->cout; s.ld_out = g->ld_out;
  return s;
}
size_t generic_wgrad_ws(const seedhip_stack_conv_geom* g) {
  const int M = 4 * g->kh * g->kw, N = g->cout;
  const long long pixels = (long long)g->T * g->B * g->oh * g->ow;
  const int per = pick_k_per_slice(pixels, tiles_for(M, N));
  const long long slices = (pixels + per - 1) / per;
  return (size_t)slices * ((size_t)M * N + N) * sizeof(float);
}
size_t fast_wgrad_ws(const seedhip_stack_conv_geom* g) {
  // geometry-only eligibility (pointer alignment is checked at launch)
  const long long fsz = (long long)g->ih * g->iw;
  (void)fsz;
  if (!(g->kh == 8 && g->kw == 8 && g->stride == 4 && g->ih == stackconv::kIH && g->iw == stackconv::kIW &&
        g->cout % 16 == 0)) return 0;
  int spc, items;
  const int grid = stackconv::wgrad_grid(g, &spc, &items);
  return (size_t)grid * ((size_t)256 * g->cout + g->cout) * sizeof(float);
}
}  // namespace

extern "C" int seedhip_conv2d_stack_fwd(const seedhip_stack_conv_geom* geom, const uint8_t* frames_ext,
                                        const uint8_t* nvalid, const float* w, const float* bias, float* out,
                                        int out_relu, void* stream) {
  int rc = check_stack(geom, "conv2d_stack_fwd"); if (rc) return rc;
  SEEDHIP_REQUIRE(frames_ext && nvalid && w && out, "conv2d_stack_fwd: null pointer");
  if (stackconv::eligible(geom, frames_ext, out) && (!bias || (((uintptr_t)bias) & 15) == 0))
    return stackconv::launch_fwd(geom, frames_ext, nvalid, w, bias, out, out_relu, (hipStream_t)stream);
  ConvStackFwd p;                                   // generic geometry: implicit-GEMM core
  p.frames_ext = frames_ext; p.nvalid = nvalid; p.w = w; p.bias = bias; p.out = out; p.out_relu = out_relu;
  p.init(to_stack(geom));
  launch_igemm_auto(p, 1, (hipStream_t)stream);
  return check_launch("conv2d_stack_fwd");
}

// The same forward, also writing the ReLU mask of its output as one byte per four channels (seedhip.h).
extern "C" int seedhip_conv2d_stack_fwd_bits_supported(const seedhip_stack_conv_geom* g) {
  return g && g->kh == 8 && g->kw == 8 && g->stride == 4 && g->ih == stackconv::kIH && g->iw == stackconv::kIW &&
         g->oh == 20 && g->ow == 20 && g->cout % 16 == 0 && g->ld_out % 4 == 0 && stackconv::bf16x3_enabled() &&
         (long long)g->T * g->B * g->oh * g->ow * g->ld_out < (1LL << 30);
}

extern "C" int seedhip_conv2d_stack_fwd_bits(const seedhip_stack_conv_geom* geom, const uint8_t* frames_ext,
                                             const uint8_t* nvalid, const float* w, const float* bias, float* out,
                                             uint8_t* relu_bits, void* stream) {
  int rc = check_stack(geom, "conv2d_stack_fwd_bits"); if (rc) return rc;
  SEEDHIP_REQUIRE(frames_ext && nvalid && w && out && relu_bits, "conv2d_stack_fwd_bits: null pointer");
  if (!(seedhip_conv2d_stack_fwd_bits_supported(geom) && stackconv::eligible(geom, frames_ext, out) &&
        (!bias || (((uintptr_t)bias) & 15) == 0)))
    return fail(SEEDHIP_ERR_UNSUPPORTED, "conv2d_stack_fwd_bits: geometry / alignment not served (ask seedhip_conv2d_stack_fwd_bits_supported)");
  return stackconv::launch_fwd(geom, frames_ext, nvalid, w, bias, out, 1, (hipStream_t)stream, relu_bits);
}

// Central inference: the first conv over n = geom->B independent rows (geom->T == 1) whose stacks live in the unroll
// store (see stackconv_rows_kernel / servestep.hip).
extern "C" int seedhip_conv2d_stack_fwd_rows_supported(const seedhip_stack_conv_geom* g) {
  return g && g->T == 1 && g->kh == 8 && g->kw == 8 && g->stride == 4 && g->ih == stackconv::kIH && g->iw == stackconv::kIW &&
         g->oh == 20 && g->ow == 20 && g->cout % 16 == 0 && g->ld_out % 4 == 0 && stackconv::bf16x3_enabled();
}

extern "C" int seedhip_conv2d_stack_fwd_rows(const seedhip_stack_conv_geom* geom, const uint8_t* obs, const uint8_t* store_obs,
                                             const long long* hist_rows, const uint8_t* nvalid, const void* w_split,
                                             const float* bias, float* out, int out_relu, void* stream) {
  int rc = check_stack(geom, "conv2d_stack_fwd_rows"); if (rc) return rc;
  SEEDHIP_REQUIRE(obs && store_obs && hist_rows && nvalid && w_split && out, "conv2d_stack_fwd_rows: null pointer");
  SEEDHIP_REQUIRE(seedhip_conv2d_stack_fwd_rows_supported(geom), "conv2d_stack_fwd_rows: geometry not served (ask seedhip_conv2d_stack_fwd_rows_supported)");
  SEEDHIP_REQUIRE(((((uintptr_t)obs) | ((uintptr_t)store_obs) | ((uintptr_t)out) | ((uintptr_t)bias) | ((uintptr_t)w_split)) & 15) == 0,
                  "conv2d_stack_fwd_rows: 16-byte aligned buffers");
  stackconv::RowsParams p;
  p.obs = obs; p.store_obs = store_obs; p.hist_rows = hist_rows; p.nvalid = nvalid;
  p.bias = bias; p.w_split = (const uint4*)w_split; p.out = out; p.B = geom->B; p.cout = geom->cout;
  p.ld_out = geom->ld_out; p.fsz = geom->ih * geom->iw;
  hipStream_t s = (hipStream_t)stream;
  if (stackconv::w8_enabled()) {                      // eight waves, two rows per workgroup, one workgroup per CU
    const int pairs = (p.B + 1) / 2;
    const int grid8 = pairs < stackconv::max_grid_for(1) ? pairs : stackconv::max_grid_for(1);
    const int lds8 = stackconv::kW8Rings + 64;
    if (out_relu) {
      (void)hipFuncSetAttribute((const void*)stackconv::stackconv_rows_w8_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds8);
      hipLaunchKernelGGL(stackconv::stackconv_rows_w8_kernel<true>, dim3(grid8, 1, geom->cout / 16), dim3(stackconv::kW8Threads), lds8, s, p);
    } else {
      (void)hipFuncSetAttribute((const void*)stackconv::stackconv_rows_w8_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds8);
      hipLaunchKernelGGL(stackconv::stackconv_rows_w8_kernel<false>, dim3(grid8, 1, geom->cout / 16), dim3(stackconv::kW8Threads), lds8, s, p);
    }
    return check_launch("stackconv_rows_w8_kernel");
  }
  const size_t lds = (size_t)stackconv::kGroups * 64 * 16 + (size_t)stackconv::kWaves * stackconv::kWaveRing16 + 64;
  const int grid = p.B < stackconv::max_grid_for(2) ? p.B : stackconv::max_grid_for(2);
  if (out_relu) {
    (void)hipFuncSetAttribute((const void*)stackconv::stackconv_rows_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(stackconv::stackconv_rows_kernel<true>, dim3(grid, 1, geom->cout / 16), dim3(stackconv::kThreads), lds, s, p);
  } else {
    (void)hipFuncSetAttribute((const void*)stackconv::stackconv_rows_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(stackconv::stackconv_rows_kernel<false>, dim3(grid, 1, geom->cout / 16), dim3(stackconv::kThreads), lds, s, p);
  }
  return check_launch("stackconv_rows_kernel");
}

extern "C" size_t seedhip_conv2d_stack_bwd_weight_workspace_bytes(const seedhip_stack_conv_geom* g) {
  if (!g) return 0;
  const size_t a = generic_wgrad_ws(g), b = fast_wgrad_ws(g);
  return a > b ? a : b;
}

extern "C" int seedhip_conv2d_stack_bwd_weight(const seedhip_stack_conv_geom* geom, const uint8_t* frames_ext,
                                               const uint8_t* nvalid, const float* dy, float* dw, float* dbias,
                                               void* workspace, size_t workspace_bytes, void* stream) {
  int rc = check_stack(geom, "conv2d_stack_bwd_weight"); if (rc) return rc;
  SEEDHIP_REQUIRE(frames_ext && nvalid && dy && dw && workspace, "conv2d_stack_bwd_weight: null pointer");
  SEEDHIP_REQUIRE(workspace_bytes >= seedhip_conv2d_stack_bwd_weight_workspace_bytes(geom),
                  "conv2d_stack_bwd_weight: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  const int M = 4 * geom->kh * geom->kw, N = geom->cout;
  if (stackconv::eligible(geom, frames_ext, dy)) {
    stackconv::Params p = stackconv::make_params(geom, frames_ext, nvalid);
    const int grid = stackconv::wgrad_grid(geom, &p.spc, &p.items);
    p.dy = dy;
    p.partial_w = (float*)workspace;
    p.partial_b = dbias ? (float*)workspace + (size_t)grid * M * N : nullptr;
    const size_t lds = stackconv::wgrad_lds(p.fsz);
    if (lds > 64 * 1024)
      (void)hipFuncSetAttribute((const void*)stackconv::stackconv_wgrad_kernel,
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    static const int bf16x3_env = getenv("SEEDHIP_STACK_BF16") ? atoi(getenv("SEEDHIP_STACK_BF16")) : 1;
    // the transposing-read kernel addresses the frames and dY as buffers with 32-bit offsets (r6)
    const bool bf16x3 = bf16x3_env && stackconv::wgrad_tr_fits(geom);
    if (bf16x3) {
#define SEEDHIP_TR(LD_)                                                                                            \
      {                                                                                                           \
        (void)hipFuncSetAttribute((const void*)stackconv::stackconv_wgrad_tr_kernel<LD_>,                          \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)stackconv::kTrLds);            \
        hipLaunchKernelGGL(stackconv::stackconv_wgrad_tr_kernel<LD_>, dim3(grid, 1, N / 16), dim3(512), stackconv::kTrLds, s, p); \
      }
      if (geom->ld_out == 16) SEEDHIP_TR(16) else if (geom->ld_out == 32) SEEDHIP_TR(32) else SEEDHIP_TR(0)
#undef SEEDHIP_TR
    }
    else
      hipLaunchKernelGGL(stackconv::stackconv_wgrad_kernel, dim3(grid, 1, N / 16), dim3(stackconv::kThreads), lds, s, p);
    rc = check_launch("stackconv_wgrad_kernel"); if (rc) return rc;
    reduce_slices2(p.partial_w, (long long)M * N, dw, p.partial_b, N, dbias, grid, s);
    return check_launch("conv2d_stack_bwd_weight");
  }
  ConvStackWgrad p;
  p.frames_ext = frames_ext; p.nvalid = nvalid; p.dy = dy;
  const long long pixels = (long long)geom->T * geom->B * geom->oh * geom->ow;
  p.init(to_stack(geom), pick_k_per_slice(pixels, tiles_for(M, N)));
  const int slices = p.slices();
  p.partial_w = (float*)workspace;
  p.partial_b = dbias ? (float*)workspace + (size_t)slices * M * N : nullptr;
  launch_igemm_auto(p, slices, s);
  reduce_slices2(p.partial_w, (long long)M * N, dw, p.partial_b, N, dbias, slices, s);
  return check_launch("conv2d_stack_bwd_weight");
}

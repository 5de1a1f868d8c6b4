// Streaming weight gradient for small strided 'valid' convolutions whose tap rows are 64 floats wide and that have 32
// output channels -- the second Atari conv, Conv2D(32, 4, 2) on the 20x20x16 output of the first
// (/root/reference/atari/networks.py:236): dW[(ky, kx, ci), co] = sum_pixels X[pixel; ky, kx, ci] * dY[pixel, co].
//
// The gather-GEMM of gemm.h stages both operands through LDS with two workgroup barriers per 32-pixel k-tile and pays
// 4.4 VALU + 1.8 SALU instructions per MFMA for the gathered operand (SQ counters, profiles/r02a_cfg2_mfma.csv:
// matrix pipe 56 % busy, 735 MB moved for 387 MB).  Here NOTHING is staged:
//   * wave w of a workgroup owns kernel row ky = w: its 64 dW rows (kx, ci) x 32 columns are 8 accumulator tiles that
//     live in registers for the whole launch;
//   * the MFMA reduction index is the PIXEL: lane (lx, kq) feeds pixel 4q + kq of the image.  Its A operand is ONE
//     16-byte load straight from global memory -- floats 4 lx .. 4 lx + 3 of the pixel's 256-byte tap row, i.e. rows
//     {4 lx + e} of the four x-interleaved tiles e -- and its B operand ONE 8-byte load, dY[pixel][2 lx, 2 lx + 1]:
//     two loads feed eight MFMAs, no LDS, no barrier, no transposition;
//   * per-image geometry (pixel -> byte offset of its window) is a 4*Q-entry LDS table read once per pixel quad; the
//     image base is a uniform soffset; pixels past the end of an image read through an out-of-range offset (zeros);
//   * loads run D pixel quads ahead through a static register ring across image boundaries.
// Every workgroup writes one partial slice (deterministic second-pass reduction, as for the other weight gradients);
// the bias gradient is wave 0's running sum of the dY values it loads anyway.
#pragma once
#include "common.h"
#include "igemm.h"
#include "../../include/seedhip.h"
#include <cstdlib>

namespace seedhip {
namespace wsw {

typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));

struct Params {
  const float* X; const float* dY; float* partial_w; float* partial_b;
  int n_img, P, Q;                       // pixels per image, pixel quads per image (Q = ceil(P / 4))
  int ow, s, iw, cin, kh;                // geometry for the table
  unsigned x_img_bytes, y_img_bytes, x_row_bytes;   // bytes per image of X / dY; bytes per input row (iw * cin * 4)
  long long x_bytes, y_bytes;            // buffer extents
  int in_relu;
};

constexpr int kDepthDefault = 8;         // pixel quads in flight per wave
constexpr unsigned kOut = 0x80000000u;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t view(const void* q, long long bytes) {
  const uint64_t qb = reinterpret_cast<uint64_t>(q);
  const uint64_t sq = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((unsigned)(qb >> 32)) << 32) |
                      (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((unsigned)qb);
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(sq), 0, __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000);
}

// EXP (tools/probes/wsw_probe.hip only): 1 no MFMAs, 2 no loads after the first ring fill.
template <bool RELU, int kDepth = kDepthDefault, int EXP = 0>
__global__ void __launch_bounds__(256)
wsw_kernel(const Params p) {
  extern __shared__ __attribute__((aligned(16))) unsigned tab[];        // [4 Q]: window byte offset of pixel p in its image
  const int tid = threadIdx.x, lane = tid & 63, ky = tid >> 6, lx = lane & 15, kq = lane >> 4;
  for (int i = tid; i < 4 * p.Q; i += blockDim.x) {
    const int oy = i / p.ow, ox = i - oy * p.ow;
    tab[i] = i < p.P ? (unsigned)((oy * p.s * p.iw + ox * p.s) * p.cin) * 4u : kOut;
  }
  __syncthreads();
  if (ky >= p.kh) return;
  const __amdgpu_buffer_rsrc_t xr = view(p.X, p.x_bytes), yr = view(p.dY, p.y_bytes);
  const unsigned xlane = (unsigned)ky * p.x_row_bytes + 16u * (unsigned)lx;      // this lane's part of a window row
  const unsigned ylane = 8u * (unsigned)lx;

  f32x4_t acc[4][2];
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[e][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  float bsum0 = 0.f, bsum1 = 0.f;

  // load cursor (uniform): image and quad of the next request
  int l_img = blockIdx.x, l_q = 0;
  u32x4_t rx[kDepth];
  u32x2_t ry[kDepth];
  auto request = [&](int d) {
    // past the last image the soffsets leave the buffers: zeros come back and are accumulated harmlessly
    const unsigned xs = __builtin_amdgcn_readfirstlane((unsigned)l_img * p.x_img_bytes);
    const unsigned ys = __builtin_amdgcn_readfirstlane((unsigned)l_img * p.y_img_bytes);
    const int pix = 4 * l_q + kq;
    const unsigned t = tab[pix];
    const unsigned yo = pix < p.P ? (unsigned)pix * (p.y_img_bytes / (unsigned)p.P) + ylane : kOut;
    const bool live = l_img < p.n_img;
    rx[d] = __builtin_amdgcn_raw_buffer_load_b128(xr, live ? t + xlane : kOut, xs, 0);
    ry[d] = __builtin_amdgcn_raw_buffer_load_b64(yr, live ? yo : kOut, ys, 0);
    if (++l_q == p.Q) { l_q = 0; l_img += gridDim.x; }
  };
  const int my_images = (p.n_img - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  const int total = my_images * p.Q;
#pragma unroll
  for (int d = 0; d < kDepth; ++d) request(d);
  for (int g0 = 0; g0 < total; g0 += kDepth) {
#pragma unroll
    for (int d = 0; d < kDepth; ++d) {
      f32x4_t x = __builtin_bit_cast(f32x4_t, rx[d]);
      const f32x2_t y = __builtin_bit_cast(f32x2_t, ry[d]);
      if (!(EXP & 2)) request(d);                            // the quad kDepth steps ahead, into the registers just read
      if (RELU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) x[e] = fmaxf(x[e], 0.f);
      }
      if (ky == 0) { bsum0 += y[0]; bsum1 += y[1]; }
      if (EXP & 1) { asm volatile("" :: "v"(x), "v"(y)); continue; }
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[e][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[e], y[j], acc[e][j], 0, 0, 0);
    }
  }

  // ---- partial slice of this workgroup: acc[e][j][r] = dW[row = 16 kq + 4 r + e][col = 2 lx + j] of kernel row ky ----
  const int N = 32, rows_per_ky = 64;
  float* pw = p.partial_w + (long long)blockIdx.x * (p.kh * rows_per_ky * N) + (long long)ky * rows_per_ky * N;
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 16 * kq + 4 * r + e;
      *reinterpret_cast<f32x2_t*>(pw + row * N + 2 * lx) = f32x2_t{acc[e][0][r], acc[e][1][r]};
    }
  if (p.partial_b && ky == 0) {
    bsum0 += __shfl_xor(bsum0, 16, 64); bsum0 += __shfl_xor(bsum0, 32, 64);
    bsum1 += __shfl_xor(bsum1, 16, 64); bsum1 += __shfl_xor(bsum1, 32, 64);
    if (kq == 0) *reinterpret_cast<f32x2_t*>(p.partial_b + (long long)blockIdx.x * N + 2 * lx) = f32x2_t{bsum0, bsum1};
  }
}

// ---- the same weight gradient with every byte requested ONCE (r4) --------------------------------------------------- //
// The streaming kernel above issues 1.34 GB of L2 load requests for 387 MB of operands (every input row is requested by
// two kernel rows x two overlapping windows, dY by all four waves), and its memory side alone (90-100 us) ADDS to the
// matrix side (111 us) instead of hiding under it: 167-181 us.  Here a workgroup stages each of its images ONCE:
//   * LDS holds two slots of {X image (ih x iw x cin fp32, 25 KB), dY of the image (P x 32 fp32 + zero rows up to a
//     multiple of four pixels, 11 KB)}; the slot of image n + 1 is filled by LDS-DMA (`buffer_load_dwordx4 ... lds`:
//     1 KB per wave instruction, no VGPRs, no ds_write, contiguous 16-byte pieces -- 36 instructions per image, nine per
//     wave) while image n is multiplied; bytes past the end of an image's dY come back as zeros (out-of-range offsets),
//     which is what pads the last pixel quad;
//   * the operands are then LDS reads with the SAME indexing as the global loads above: A = one ds_read_b128 at the
//     pixel's window offset (kernel row ky of wave ky, floats 4 lx .. 4 lx + 3), B = one ds_read_b64 of dY[pixel][2 lx,
//     2 lx + 1] -- two LDS reads per eight MFMAs; the per-quad window offsets live in registers (Q of them);
//   * one workgroup barrier per image (5 400 MFMA cycles), behind a wait for the wave's own DMA pieces.
// PF: pixel quads whose operands are requested ahead of their MFMAs (pinned with sched_barrier; 0: the compiler's order)
template <bool RELU, int Q, int PF = 0>
__global__ void __launch_bounds__(256, 2)
wsw_lds_kernel(const Params p, const int xs, const int ys) {           // xs / ys: bytes of the X / dY part of a slot (KB multiples)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, lx = lane & 15, kq = lane >> 4;
  const int ky = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int slot = xs + ys;
  const __amdgpu_buffer_rsrc_t xr = view(p.X, p.x_bytes), yr = view(p.dY, p.y_bytes);

  // window offsets of this lane's pixel of every quad (same for every image)
  unsigned xo[Q], yo[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    const int pix = 4 * q + kq;
    const int pp = pix < p.P ? pix : 0;                      // a pad pixel multiplies X by the zero rows of dY
    const int oy = pp / p.ow, ox = pp - oy * p.ow;
    xo[q] = (unsigned)((oy * p.s * p.iw + ox * p.s) * p.cin) * 4u + (unsigned)ky * p.x_row_bytes + 16u * (unsigned)lx;
    yo[q] = (unsigned)xs + (unsigned)pix * 128u + 8u * (unsigned)lx;
  }
  const int nx = xs >> 10, npieces = (xs + ys) >> 10;      // 1 KB pieces of a slot; piece j is issued by wave j % 4
  auto stage = [&](int img, int s) {                         // image img -> slot s (asynchronous)
    const unsigned xb = __builtin_amdgcn_readfirstlane((unsigned)img * p.x_img_bytes);
    const unsigned yb = __builtin_amdgcn_readfirstlane((unsigned)img * p.y_img_bytes);
    for (int j = ky; j < npieces; j += 4) {
      const bool is_x = j < nx;
      const unsigned off = (unsigned)(is_x ? j : j - nx) * 1024u + 16u * (unsigned)lane;
      const unsigned lim = is_x ? p.x_img_bytes : p.y_img_bytes;
      typedef __attribute__((address_space(3))) void lds_void_t;
      lds_void_t* dst = (lds_void_t*)(smem + s * slot + j * 1024);
      if (is_x) __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, dst, 16, off < lim ? off : kOut, xb, 0, 0);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(yr, dst, 16, off < lim ? off : kOut, yb, 0, 0);
    }
  };

  f32x4_t acc[4][2];
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[e][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  float bsum0 = 0.f, bsum1 = 0.f;

  const int my_images = (p.n_img - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  if (my_images > 0) stage(blockIdx.x, 0);
  __syncthreads();                                           // (waits for this wave's pieces: vmcnt(0), then the barrier)
  for (int n = 0; n < my_images; ++n) {
    if (n + 1 < my_images) stage(blockIdx.x + (n + 1) * gridDim.x, (n + 1) & 1);
    const unsigned so = (unsigned)((n & 1) * slot);
    if (ky < p.kh) {
      f32x4_t xr_[PF > 0 ? PF : 1]; f32x2_t yr_[PF > 0 ? PF : 1];
#pragma unroll
      for (int q = 0; q < PF; ++q) {
        xr_[q] = *reinterpret_cast<const f32x4_t*>(smem + so + xo[q]);
        yr_[q] = *reinterpret_cast<const f32x2_t*>(smem + so + yo[q]);
      }
#pragma unroll
      for (int q = 0; q < Q; ++q) {
        f32x4_t x; f32x2_t y;
        if (PF > 0) {
          x = xr_[q % PF]; y = yr_[q % PF];
          if (q + PF < Q) {
            xr_[q % PF] = *reinterpret_cast<const f32x4_t*>(smem + so + xo[q + PF]);
            yr_[q % PF] = *reinterpret_cast<const f32x2_t*>(smem + so + yo[q + PF]);
          }
          __builtin_amdgcn_sched_barrier(0);
        } else {
          x = *reinterpret_cast<const f32x4_t*>(smem + so + xo[q]);
          y = *reinterpret_cast<const f32x2_t*>(smem + so + yo[q]);
        }
        if (RELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) x[e] = fmaxf(x[e], 0.f);
        }
        if (ky == 0) { bsum0 += y[0]; bsum1 += y[1]; }
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[e][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[e], y[j], acc[e][j], 0, 0, 0);
        if (PF > 0) __builtin_amdgcn_sched_barrier(0);
      }
    }
    __syncthreads();                                         // image n consumed by all four waves; image n + 1 landed
  }
  if (ky >= p.kh) return;

  // ---- partial slice of this workgroup (as wsw_kernel) ----
  const int N = 32, rows_per_ky = 64;
  float* pw = p.partial_w + (long long)blockIdx.x * (p.kh * rows_per_ky * N) + (long long)ky * rows_per_ky * N;
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 16 * kq + 4 * r + e;
      *reinterpret_cast<f32x2_t*>(pw + row * N + 2 * lx) = f32x2_t{acc[e][0][r], acc[e][1][r]};
    }
  if (p.partial_b && ky == 0) {
    bsum0 += __shfl_xor(bsum0, 16, 64); bsum0 += __shfl_xor(bsum0, 32, 64);
    bsum1 += __shfl_xor(bsum1, 16, 64); bsum1 += __shfl_xor(bsum1, 32, 64);
    if (kq == 0) *reinterpret_cast<f32x2_t*>(p.partial_b + (long long)blockIdx.x * N + 2 * lx) = f32x2_t{bsum0, bsum1};
  }
}

// Measured at cfg2 (tools/probes/wsw_probe.hip; us for the kernel alone): MFMAs alone 111, loads alone 90-100; depth 8 x
// 512 workgroups 193 (no overlap at all: too few bytes in flight), depth 12 x 1024 workgroups 153, depth 16 x 512 150.
// In the learner step (random data, the slice reduction included) every setting lands within 2 % of the others and ~4 %
// under the gather-GEMM (0.188 vs 0.195 ms).  Default: 512 workgroups (two per CU) x 16 quads.
inline int grid_for(int n_img) {
  static const int g = getenv("SEEDHIP_WSW_GRID") ? atoi(getenv("SEEDHIP_WSW_GRID")) : 512;
  return n_img < g ? n_img : g;
}

// Eligibility + geometry; returns false when the shape is outside this kernel's range.
inline bool plan(Params& p, const seedhip_conv_geom* g) {
  if (g->pad_t || g->pad_l || g->kw * g->cin != 64 || g->cout != 32 || g->kh < 1 || g->kh > 4 || g->ld_in != g->cin ||
      g->ld_out != g->cout)
    return false;
  const long long xb = (long long)g->n_img * g->ih * g->iw * g->cin * 4, yb = (long long)g->n_img * g->oh * g->ow * g->cout * 4;
  const int P = g->oh * g->ow;
  if (xb >= (1LL << 31) - (1 << 22) || yb >= (1LL << 31) - (1 << 22) || P > 1024) return false;
  memset(&p, 0, sizeof(p));
  p.n_img = g->n_img; p.P = P; p.Q = (P + 3) / 4; p.ow = g->ow; p.s = g->stride; p.iw = g->iw; p.cin = g->cin; p.kh = g->kh;
  p.x_img_bytes = (unsigned)(g->ih * g->iw * g->cin) * 4u; p.y_img_bytes = (unsigned)(P * g->cout) * 4u;
  p.x_row_bytes = (unsigned)(g->iw * g->cin) * 4u;
  p.x_bytes = xb; p.y_bytes = yb;
  return true;
}

inline int launch(const Params& p, hipStream_t s) {
  const int grid = grid_for(p.n_img);
  // the LDS-staged kernel: instantiated for the second Atari conv's 81 pixels (21 quads); SEEDHIP_WSW_LDS=0: streaming
  static const int lds_on = getenv("SEEDHIP_WSW_LDS") ? atoi(getenv("SEEDHIP_WSW_LDS")) : 1;
  if (lds_on && p.Q == 21 && (p.x_img_bytes & 15) == 0 && (p.y_img_bytes & 15) == 0 && p.y_img_bytes == (unsigned)p.P * 128u) {
    const int xs = (int)((p.x_img_bytes + 1023u) & ~1023u), ys = (int)(((unsigned)(4 * p.Q) * 128u + 1023u) & ~1023u);
    const int bytes = 2 * (xs + ys);
    if (bytes <= 80 * 1024) {
      static const int pf = getenv("SEEDHIP_WSW_PF") ? atoi(getenv("SEEDHIP_WSW_PF")) : 0;
#define SEEDHIP_WSWL(R_, P_) { \
        static const bool ok = hipFuncSetAttribute((const void*)wsw_lds_kernel<R_, 21, P_>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) == hipSuccess; \
        if (ok) { hipLaunchKernelGGL((wsw_lds_kernel<R_, 21, P_>), dim3(grid), dim3(256), bytes, s, p, xs, ys); return check_launch("wsw_lds_kernel"); } }
      if (p.in_relu) { if (pf == 2) SEEDHIP_WSWL(true, 2) else if (pf == 3) SEEDHIP_WSWL(true, 3) else SEEDHIP_WSWL(true, 0) }
      else { if (pf == 2) SEEDHIP_WSWL(false, 2) else if (pf == 3) SEEDHIP_WSWL(false, 3) else if (pf == 4) SEEDHIP_WSWL(false, 4) else SEEDHIP_WSWL(false, 0) }
#undef SEEDHIP_WSWL
    }
  }
  const size_t lds = (size_t)4 * p.Q * sizeof(unsigned);
  static const int depth = getenv("SEEDHIP_WSW_DEPTH") ? atoi(getenv("SEEDHIP_WSW_DEPTH")) : 16;
#define SEEDHIP_WSW_L(R_, D_) hipLaunchKernelGGL((wsw_kernel<R_, D_>), dim3(grid), dim3(256), lds, s, p)
  if (p.in_relu) { if (depth >= 16) SEEDHIP_WSW_L(true, 16); else if (depth >= 12) SEEDHIP_WSW_L(true, 12); else SEEDHIP_WSW_L(true, 8); }
  else { if (depth >= 16) SEEDHIP_WSW_L(false, 16); else if (depth >= 12) SEEDHIP_WSW_L(false, 12); else SEEDHIP_WSW_L(false, 8); }
#undef SEEDHIP_WSW_L
  return check_launch("wsw_kernel");
}

}  // namespace wsw
}  // namespace seedhip

// Image-resident forward of small strided 'valid' convolutions with 16 input and 32 output channels and a 4 x 4 kernel --
// the second Atari conv, Conv2D(32, 4, 2) on the 20 x 20 x 16 output of the first
// (/root/reference/atari/networks.py:236): Y[pixel, co] = act(bias[co] + sum_{ky, kx, ci} X[pixel; ky, kx, ci] W[ky, kx, ci, co]).
//
// wsgemm.h's weight-stationary kernel gathers every A fragment straight from global memory: each input element is
// requested by the (up to) four windows that contain it -- 892 MB of L2 load requests for 275 MB of input at cfg2 -- and
// its memory side alone (82 us) adds to the LDS + matrix side (117 us) instead of hiding under it (156 -> 145 us over
// two rounds of tuning).  Same remedy as wsw.h's LDS kernel (r4):
//   * a workgroup owns a contiguous run of images and stages each ONCE, by LDS-DMA (`buffer_load_dwordx4 ... lds`,
//     1 KB per wave instruction, no VGPRs, no ds_write), into a ring of three 25 KB slots, one round ahead of its use;
//   * the WEIGHTS live in registers -- lane (n, kq) holds W[tap][4 kq + e][n + 16 j] for all 16 taps: 128 VGPRs -- so the
//     only LDS traffic of the loop is the pixel operand: ONE ds_read_b128 (four channels of one tap of the lane's
//     pixel, immediate tap offsets) per EIGHT MFMAs;
//   * MFMA operands swapped as in ws_tab_kernel (W supplies the instruction's rows): a lane ends up with four
//     consecutive output channels of one pixel -> 16-byte stores with bias and ReLU fused;
//   * tiles are 16 consecutive pixels of the workgroup's run (they cross image boundaries: 81 pixels per image), one
//     tile per wave and round, a barrier per round (128 MFMAs = 4 096 matrix-pipe cycles per wave).
#pragma once
#include "common.h"
#include "igemm.h"
#include "wsw.h"
#include "../../include/seedhip.h"

namespace seedhip {
namespace wfw {

typedef float f32x4_t __attribute__((ext_vector_type(4)));

struct Params {
  const float* X; const float* W; const float* bias; float* Y;
  int n_img, P;                          // images, output pixels per image
  int ow, s, iw, ih;
  unsigned x_img_bytes;                  // ih * iw * 16 * 4
  long long x_bytes;
  int in_relu, out_relu;
  int per_wg;                            // images per workgroup
  FastDiv dP, dow;                       // division by P and by ow
};

constexpr int kSlots = 3;
constexpr unsigned kOut = 0x80000000u;

template <bool RELU_IN>
__global__ void __launch_bounds__(256, 2)
wfw_kernel(const Params p, const int xs) {                   // xs: bytes of a slot (KB multiple >= x_img_bytes)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, i = lane & 15, kq = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int img0 = blockIdx.x * p.per_wg;
  int nimg = p.n_img - img0; if (nimg > p.per_wg) nimg = p.per_wg;
  if (nimg <= 0) return;
  const __amdgpu_buffer_rsrc_t xr = wsw::view(p.X, p.x_bytes);

  // weights: wreg[tap][e][j] = W[tap][ci = 4 kq + e][co = 16 j + i]  (Keras layout [kh, kw, cin, cout])
  float wreg[16][4][2];
#pragma unroll
  for (int t = 0; t < 16; ++t)
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int j = 0; j < 2; ++j) wreg[t][e][j] = p.W[(t * 16 + 4 * kq + e) * 32 + 16 * j + i];
  f32x4_t bias4[2] = {f32x4_t{0.f, 0.f, 0.f, 0.f}, f32x4_t{0.f, 0.f, 0.f, 0.f}};
  if (p.bias) {
#pragma unroll
    for (int j = 0; j < 2; ++j) bias4[j] = *reinterpret_cast<const f32x4_t*>(p.bias + 16 * j + 4 * kq);
  }

  const int npieces = xs >> 10;                              // 1 KB pieces of an image; piece q is issued by wave q % 4
  auto stage = [&](int li) {                                 // local image li -> slot li % 3 (asynchronous)
    const unsigned xb = __builtin_amdgcn_readfirstlane((unsigned)(img0 + li) * p.x_img_bytes);
    const int slot = li % kSlots;
    for (int q = wave; q < npieces; q += 4) {
      const unsigned off = (unsigned)q * 1024u + 16u * (unsigned)lane;
      typedef __attribute__((address_space(3))) void lds_void_t;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_void_t*)(smem + slot * xs + q * 1024), 16,
                                               off < p.x_img_bytes ? off : kOut, xb, 0, 0);
    }
  };

  const int total = nimg * p.P;                              // pixels of this workgroup's run
  const int rounds = (total + 63) >> 6;
  // Round r reads images (64 r) / P .. (64 r + 63) / P.  The images of round r + 1 are requested at the START of round r
  // and are complete at the barrier that ends it.  Three slots suffice: a request for image s <= (64 r + 127) / P
  // overwrites image s - 3 <= (64 r) / P - 1 (P >= 64), which no round >= r reads, and every wave is past round r - 1.
  auto need = [&](int r) { int v = (int)p.dP.div((unsigned)(64 * r + 63)); return v < nimg - 1 ? v : nimg - 1; };
  int staged = 0;
  for (const int n0 = need(0); staged <= n0; ++staged) stage(staged);
  __syncthreads();                                           // (vmcnt(0) of this wave's pieces, then the barrier)

  const long long ybase = (long long)img0 * p.P * 32;
  for (int r = 0; r < rounds; ++r) {
    for (const int n1 = need(r + 1); staged <= n1; ++staged) stage(staged);
    const int P = 64 * r + 16 * wave + i;                    // this lane's pixel in the run
    const bool live = P < total;
    const unsigned Pc = (unsigned)(live ? P : total - 1);
    unsigned li, pix, oy, ox;
    p.dP.divmod(Pc, li, pix);
    p.dow.divmod(pix, oy, ox);
    const unsigned char* src = smem + (li % kSlots) * (unsigned)xs + ((oy * p.s * p.iw + ox * p.s) * 16 + 4 * kq) * 4;
    f32x4_t acc[2] = {bias4[0], bias4[1]};
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      f32x4_t x = *reinterpret_cast<const f32x4_t*>(src + ((t >> 2) * p.iw + (t & 3)) * 64);
      if (RELU_IN) {
#pragma unroll
        for (int e = 0; e < 4; ++e) x[e] = fmaxf(x[e], 0.f);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[t][e][j], x[e], acc[j], 0, 0, 0);
    }
    // acc[j][q] = Y[pixel P][16 j + 4 kq + q]
    if (live) {
      float* o = p.Y + ybase + (long long)P * 32 + 4 * kq;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        f32x4_t v = acc[j];
        if (p.out_relu) {
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q] = __builtin_amdgcn_fmed3f(v[q], 0.f, __builtin_inff());
        }
        *reinterpret_cast<f32x4_t*>(o + 16 * j) = v;
      }
    }
    __syncthreads();                                         // round r read by all waves; the staged images landed
  }
}

// Eligibility + geometry (the second Atari conv; enough images that a workgroup's prologue is amortised)
inline bool plan(Params& p, const seedhip_conv_geom* g) {
  if (g->pad_t || g->pad_l || g->kh != 4 || g->kw != 4 || g->cin != 16 || g->cout != 32 || g->ld_in != 16 || g->ld_out != 32)
    return false;
  const long long xb = (long long)g->n_img * g->ih * g->iw * 16 * 4;
  const int P = g->oh * g->ow;
  if (xb >= (1LL << 31) - (1 << 22) || P < 16 || g->n_img < 2048) return false;
  const unsigned img = (unsigned)(g->ih * g->iw * 16) * 4u;
  const unsigned xs = (img + 1023u) & ~1023u;
  if (kSlots * xs > 78 * 1024) return false;                // two workgroups per CU
  if (P < 64) return false;                                 // three slots cover a round of 64 pixels and its successor
  memset(&p, 0, sizeof(p));
  p.dP.init((uint32_t)P); p.dow.init((uint32_t)g->ow);
  p.n_img = g->n_img; p.P = P; p.ow = g->ow; p.s = g->stride; p.iw = g->iw; p.ih = g->ih;
  p.x_img_bytes = img; p.x_bytes = xb;
  return true;
}

inline int launch(Params& p, hipStream_t s) {
  static const int grid_want = getenv("SEEDHIP_WFW_GRID") ? atoi(getenv("SEEDHIP_WFW_GRID")) : 512;
  p.per_wg = (p.n_img + grid_want - 1) / grid_want;
  const int grid = (p.n_img + p.per_wg - 1) / p.per_wg;
  const int xs = (int)((p.x_img_bytes + 1023u) & ~1023u);
  const int bytes = kSlots * xs;
#define SEEDHIP_WFW_L(R_) { \
    static const bool ok = hipFuncSetAttribute((const void*)wfw_kernel<R_>, hipFuncAttributeMaxDynamicSharedMemorySize, 78 * 1024) == hipSuccess; \
    if (!ok) return -1; \
    hipLaunchKernelGGL((wfw_kernel<R_>), dim3(grid), dim3(256), bytes, s, p, xs); }
  if (p.in_relu) SEEDHIP_WFW_L(true) else SEEDHIP_WFW_L(false)
#undef SEEDHIP_WFW_L
  return check_launch("wfw_kernel");
}

}  // namespace wfw
}  // namespace seedhip

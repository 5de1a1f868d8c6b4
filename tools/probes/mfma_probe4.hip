// Probe 4: the stackconv forward inner structure, feature by feature (bisecting its ~50% MFMA utilisation).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float ubyte(uint32_t w, int q) { return (float)((w >> (8 * q)) & 0xFFu); }
constexpr int kFsz = 7056, kIw = 84;

// F bit0: A words from LDS ring; bit1: weights from LDS; bit2: per-step barrier; bit3: epilogue global stores;
//   bit4: frame prefetch global->regs->LDS per step; bit5: convert a whole r-slice first (asm barrier), MFMAs back to back
template <int F>
__global__ void __launch_bounds__(320) probe(const uint32_t* __restrict__ in, float* __restrict__ out, int steps, int nvp) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* w_lds = reinterpret_cast<float*>(smem);
  unsigned char* ring = smem + 4096 * 4;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kq = lane >> 4, j = lane & 15;
  for (int i = tid; i < 4096; i += 320) w_lds[i] = (float)(i & 255) * (1.0f / 255.0f);
  for (int i = tid; i < 5 * kFsz / 4; i += 320) reinterpret_cast<uint32_t*>(ring)[i] = in[i & 1023];
  __syncthreads();
  int aoff[5];
  for (int m = 0; m < 5; ++m) {
    const int pix = (wave * 5 + m) * 16 + j;
    const int oy = pix / 20, ox = pix - oy * 20;
    aoff[m] = (oy * 4 + (kq >> 1)) * kIw + ox * 4 + 4 * (kq & 1);
  }
  float total = 0.f;
  for (int t = 0; t < steps; ++t) {
    uint4 pf0 = make_uint4(0, 0, 0, 0), pf1 = pf0;
    if (F & 16) {
      const uint4* src = reinterpret_cast<const uint4*>(in) + (((blockIdx.x * 31 + t) & 63) * 441);
      if (tid < 441) pf0 = src[tid];
      if (tid + 320 < 441) pf1 = src[tid + 320];
    }
    f32x4 acc[5];
#pragma unroll
    for (int m = 0; m < 5; ++m) acc[m] = f32x4{0, 0, 0, 0};
    const int nv = nvp;
    for (int c = 0; c < nv; ++c) {
      const unsigned char* base = ring + ((t + 3 - c) % 5) * kFsz;
      const float* wl = w_lds + c * 16 * 64 + lane;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        uint32_t a[5];
#pragma unroll
        for (int m = 0; m < 5; ++m)
          a[m] = (F & 1) ? *reinterpret_cast<const uint32_t*>(base + aoff[m] + r * 2 * kIw) : (uint32_t)(aoff[m] + r + c);
        if (F & 32) {
          float af[4][5], bw[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            bw[q] = (F & 2) ? wl[(r * 4 + q) * 64] : 1.5f + q;
#pragma unroll
            for (int m = 0; m < 5; ++m) af[q][m] = ubyte(a[m], q);
          }
          asm volatile("" : "+v"(af[0][0]), "+v"(af[0][1]), "+v"(af[0][2]), "+v"(af[0][3]), "+v"(af[0][4]),
                            "+v"(af[1][0]), "+v"(af[1][1]), "+v"(af[1][2]), "+v"(af[1][3]), "+v"(af[1][4]),
                            "+v"(af[2][0]), "+v"(af[2][1]), "+v"(af[2][2]), "+v"(af[2][3]), "+v"(af[2][4]),
                            "+v"(af[3][0]), "+v"(af[3][1]), "+v"(af[3][2]), "+v"(af[3][3]), "+v"(af[3][4]),
                            "+v"(bw[0]), "+v"(bw[1]), "+v"(bw[2]), "+v"(bw[3]));
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int m = 0; m < 5; ++m)
              acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[q], af[q][m], acc[m], 0, 0, 0);
        } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float bw = (F & 2) ? wl[(r * 4 + q) * 64] : 1.5f + q;
#pragma unroll
          for (int m = 0; m < 5; ++m)
            acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(bw, ubyte(a[m], q), acc[m], 0, 0, 0);
        }
        }
      }
    }
    if (F & 8) {
#pragma unroll
      for (int m = 0; m < 5; ++m) {
        float* o = out + ((size_t)((blockIdx.x * 7 + t) & 1023) * 400 + (wave * 5 + m) * 16 + j) * 16 + 4 * kq;
        *reinterpret_cast<float4*>(o) = make_float4(acc[m][0], acc[m][1], acc[m][2], acc[m][3]);
      }
    } else {
#pragma unroll
      for (int m = 0; m < 5; ++m) total += acc[m][0] + acc[m][1] + acc[m][2] + acc[m][3];
    }
    if (F & 16) {
      uint4* dst = reinterpret_cast<uint4*>(ring + ((t + 4) % 5) * kFsz);
      if (tid < 441) dst[tid] = pf0;
      if (tid + 320 < 441) dst[tid + 320] = pf1;
    }
    if (F & 4) __syncthreads();
  }
  if (!(F & 8)) out[blockIdx.x * 320 + tid] = total;
}

template <int F>
void run(int grid, const uint32_t* in, float* out) {
  const int steps = 210;
  const size_t lds = 4096 * 4 + 5 * kFsz;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL((probe<F>), dim3(grid), dim3(320), lds, 0, in, out, 2, 4);
  hipEventRecord(a);
  hipLaunchKernelGGL((probe<F>), dim3(grid), dim3(320), lds, 0, in, out, steps, 4);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double flops = (double)grid * 5 * steps * 320 * 2048.0;
  printf("F=%2d (ldsA %d ldsW %d barrier %d stores %d prefetch %d) grid %4d  %8.3f ms  %7.1f TF/s\n", F, F & 1, (F >> 1) & 1,
         (F >> 2) & 1, (F >> 3) & 1, (F >> 4) & 1, grid, ms, flops / ms / 1e9);
}

int main() {
  uint32_t* in; float* out;
  hipMalloc(&in, 64 * 441 * 16 + 4096); hipMalloc(&out, (size_t)1024 * 400 * 16 * 4 + 4096 * 320 * 4);
  uint32_t* h = (uint32_t*)malloc(64 * 441 * 16);
  for (int i = 0; i < 64 * 441 * 4; ++i) h[i] = (uint32_t)rand() * 2654435761u;
  hipMemcpy(in, h, 64 * 441 * 16, hipMemcpyHostToDevice);
  const int grid = 768;
  run<0>(grid, in, out); run<1>(grid, in, out); run<2>(grid, in, out); run<3>(grid, in, out);
  run<7>(grid, in, out); run<15>(grid, in, out); run<31>(grid, in, out); run<27>(grid, in, out);
  run<32>(grid, in, out); run<35>(grid, in, out); run<63>(grid, in, out); run<59>(grid, in, out);
  return 0;
}

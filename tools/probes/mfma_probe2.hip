// Probe 2: how many LDS reads per 20 MFMAs can ride along before the fp32 MFMA pipe loses throughput?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int LR, int WIDE>
__global__ void __launch_bounds__(320) probe(const uint32_t* __restrict__ in, float* __restrict__ out, int iters) {
  __shared__ __attribute__((aligned(16))) uint32_t lds[8192];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 8192; i += 320) lds[i] = in[i & 1023];
  __syncthreads();
  f32x4 acc[5];
#pragma unroll
  for (int m = 0; m < 5; ++m) acc[m] = f32x4{0, 0, 0, 0};
  uint32_t w[12];
#pragma unroll
  for (int m = 0; m < 12; ++m) w[m] = in[lane + m];
  for (int it = 0; it < iters; ++it) {
    uint32_t wn[12];
#pragma unroll
    for (int m = 0; m < 12; ++m) wn[m] = w[m];
    if (WIDE == 0) {
#pragma unroll
      for (int m = 0; m < LR; ++m) wn[m] = lds[((it * 7 + m * 67) & 1023) + lane];
    } else {
#pragma unroll
      for (int m = 0; m < LR; ++m) {
        uint4 v = *reinterpret_cast<const uint4*>(&lds[(((it * 7 + m * 67) & 255) + lane) * 4]);
        wn[m] = v.x ^ v.y ^ v.z ^ v.w;
      }
    }
    float af[20];
#pragma unroll
    for (int k = 0; k < 20; ++k) af[k] = (float)((w[k % 5] >> (8 * (k / 5))) & 0xff);
    asm volatile("" : "+v"(af[0]), "+v"(af[1]), "+v"(af[2]), "+v"(af[3]), "+v"(af[4]), "+v"(af[5]), "+v"(af[6]),
                      "+v"(af[7]), "+v"(af[8]), "+v"(af[9]), "+v"(af[10]), "+v"(af[11]), "+v"(af[12]), "+v"(af[13]),
                      "+v"(af[14]), "+v"(af[15]), "+v"(af[16]), "+v"(af[17]), "+v"(af[18]), "+v"(af[19]));
#pragma unroll
    for (int k = 0; k < 20; ++k)
      acc[k % 5] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(w[5 + k / 5]), af[k], acc[k % 5], 0, 0, 0);
#pragma unroll
    for (int m = 0; m < 12; ++m) w[m] = wn[m];
  }
  float s = 0;
#pragma unroll
  for (int m = 0; m < 5; ++m) s += acc[m][0] + acc[m][1] + acc[m][2] + acc[m][3];
  out[blockIdx.x * 320 + tid] = s;
}

template <int LR, int WIDE>
void run(int grid, const uint32_t* in, float* out) {
  const int iters = 4000;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL((probe<LR, WIDE>), dim3(grid), dim3(320), 0, 0, in, out, 10);
  hipEventRecord(a);
  hipLaunchKernelGGL((probe<LR, WIDE>), dim3(grid), dim3(320), 0, 0, in, out, iters);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double flops = (double)grid * 5 * iters * 20 * 2048.0;
  printf("LR=%2d wide=%d grid %4d  %8.3f ms  %7.1f TF/s\n", LR, WIDE, grid, ms, flops / ms / 1e9);
}

int main() {
  uint32_t* in; float* out;
  hipMalloc(&in, 8192 * 4); hipMalloc(&out, 4096 * 320 * 4);
  uint32_t* h = (uint32_t*)malloc(8192 * 4);
  for (int i = 0; i < 8192; ++i) h[i] = (uint32_t)rand() * 2654435761u;
  hipMemcpy(in, h, 8192 * 4, hipMemcpyHostToDevice);
  for (int grid : {512, 768, 1024}) {
    run<0, 0>(grid, in, out); run<1, 0>(grid, in, out); run<2, 0>(grid, in, out); run<5, 0>(grid, in, out);
    run<9, 0>(grid, in, out); run<12, 0>(grid, in, out); run<2, 1>(grid, in, out); run<3, 1>(grid, in, out);
  }
  return 0;
}

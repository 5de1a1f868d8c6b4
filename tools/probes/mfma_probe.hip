// Ablation probe for the fp32 MFMA pipe (v_mfma_f32_16x16x4_f32) under the stackconv instruction mix.
// hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_probe tools/probes/mfma_probe.hip && /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int V, int NACC>
__global__ void __launch_bounds__(320) probe(const uint32_t* __restrict__ in, float* __restrict__ out, int iters) {
  __shared__ uint32_t lds[4096 + 2048];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 4096 + 2048; i += 320) lds[i] = in[i & 1023];
  __syncthreads();
  f32x4 acc[NACC];
#pragma unroll
  for (int m = 0; m < NACC; ++m) acc[m] = f32x4{0, 0, 0, 0};
  float bw = (float)lane, ax = 1.0f + lane;
  uint32_t w[NACC];
#pragma unroll
  for (int m = 0; m < NACC; ++m) w[m] = in[lane + m];
  for (int it = 0; it < iters; ++it) {
    if (V >= 2) {
#pragma unroll
      for (int m = 0; m < NACC; ++m) w[m] = lds[((it * 7 + m * 67) & 1023) + lane];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float b = bw;
      if (V >= 3) b = __uint_as_float(lds[4096 + ((it * 4 + q) & 31) * 64 + lane]);
#pragma unroll
      for (int m = 0; m < NACC; ++m) {
        float a = ax;
        if (V >= 1) a = (float)((w[m] >> (8 * q)) & 0xff);
        acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, acc[m], 0, 0, 0);
      }
    }
  }
  float s = 0;
#pragma unroll
  for (int m = 0; m < NACC; ++m) s += acc[m][0] + acc[m][1] + acc[m][2] + acc[m][3];
  out[blockIdx.x * 320 + tid] = s;
}

template <int V, int NACC>
void run(const char* name, int grid, const uint32_t* in, float* out) {
  const int iters = 4000;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL((probe<V, NACC>), dim3(grid), dim3(320), 0, 0, in, out, 10);
  hipEventRecord(a);
  hipLaunchKernelGGL((probe<V, NACC>), dim3(grid), dim3(320), 0, 0, in, out, iters);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double flops = (double)grid * 5 * iters * 4 * NACC * 2048.0;
  printf("%-34s grid %4d  %8.3f ms  %7.1f TF/s\n", name, grid, ms, flops / ms / 1e9);
}

int main() {
  uint32_t* in; float* out;
  hipMalloc(&in, 8192 * 4); hipMalloc(&out, 4096 * 320 * 4);
  hipMemset(in, 0x3c, 8192 * 4);
  for (int grid : {256, 512, 768}) {
    run<0, 5>("V0 pure mfma, 5 acc", grid, in, out);
    run<1, 5>("V1 + cvt_ubyte", grid, in, out);
    run<2, 5>("V2 + lds A reads", grid, in, out);
    run<3, 5>("V3 + lds B reads", grid, in, out);
    run<0, 4>("V0 pure mfma, 4 acc", grid, in, out);
    run<0, 8>("V0 pure mfma, 8 acc", grid, in, out);
  }
  return 0;
}

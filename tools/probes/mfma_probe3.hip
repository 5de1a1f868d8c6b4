// Probe 3: bisect what makes the stackconv fwd MFMA loop run at half rate.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float ubyte(uint32_t w, int q) { return (float)((w >> (8 * q)) & 0xFFu); }

// V: 0 = one long loop (like probe 1); 1 = outer step loop, inner c loop of 4 with acc re-init + epilogue sum
//    2 = V1 + dynamic LDS 51 KB;  3 = V1 + trip count from memory
template <int V>
__global__ void __launch_bounds__(320) probe(const uint32_t* __restrict__ in, float* __restrict__ out, int steps, int nvp) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  uint32_t a0[5];
#pragma unroll
  for (int m = 0; m < 5; ++m) a0[m] = in[lane + m];
  float total = 0.f;
  for (int t = 0; t < steps; ++t) {
    f32x4 acc[5];
#pragma unroll
    for (int m = 0; m < 5; ++m) acc[m] = f32x4{0, 0, 0, 0};
    const int nv = (V == 3) ? (int)in[t & 7] : nvp;
    for (int c = 0; c < nv; ++c) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        uint32_t a[5];
#pragma unroll
        for (int m = 0; m < 5; ++m) a[m] = a0[m] + r + c;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float bw = 1.5f + q;
#pragma unroll
          for (int m = 0; m < 5; ++m)
            acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(bw, ubyte(a[m], q), acc[m], 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int m = 0; m < 5; ++m) total += acc[m][0] + acc[m][1] + acc[m][2] + acc[m][3];
    if (V == 2) { smem[tid] = (unsigned char)total; }
  }
  out[blockIdx.x * 320 + tid] = total;
}

template <int V>
void run(const char* name, int grid, size_t lds, const uint32_t* in, float* out) {
  const int steps = 250;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipFuncSetAttribute((const void*)probe<V>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
  hipLaunchKernelGGL((probe<V>), dim3(grid), dim3(320), lds, 0, in, out, 2, 4);
  hipEventRecord(a);
  hipLaunchKernelGGL((probe<V>), dim3(grid), dim3(320), lds, 0, in, out, steps, 4);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double flops = (double)grid * 5 * steps * 320 * 2048.0;
  printf("%-40s grid %4d lds %6zu  %8.3f ms  %7.1f TF/s\n", name, grid, lds, ms, flops / ms / 1e9);
}

int main() {
  uint32_t* in; float* out;
  hipMalloc(&in, 8192 * 4); hipMalloc(&out, 4096 * 320 * 4);
  uint32_t h[8192]; for (int i = 0; i < 8192; ++i) h[i] = 4;
  hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  for (int grid : {512, 768}) {
    run<1>("V1 step loop, c loop(4), re-init", grid, 0, in, out);
    run<2>("V2 + 51KB dyn LDS", grid, 51664, in, out);
    run<2>("V2 + 35KB dyn LDS", grid, 35 * 1024, in, out);
    run<3>("V3 trip count from memory", grid, 0, in, out);
  }
  return 0;
}

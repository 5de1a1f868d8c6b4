// MFMA issue ORDER over a fixed set of accumulators (r4): rotating over NACC accumulators (a0 a1 .. a0 a1 ..) against
// chains of CH back-to-back MFMAs on one accumulator before moving on (a0 a0 a0 a1 a1 a1 ..).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/chain tools/probes/mfma_chain_probe.hip && /tmp/chain
// SHAPE 0: v_mfma_f32_16x16x32_bf16, 2: v_mfma_f32_16x16x4_f32.  RD: 16-byte LDS reads per NACC * CH MFMAs.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <type_traits>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE, int NACC, int CH, int RD>
__global__ void __launch_bounds__(256) probe(float* __restrict__ out, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  unsigned* lu = reinterpret_cast<unsigned*>(smem);
  for (int i = tid; i < 4096; i += 256) lu[i] = 0x3f803f80u + (unsigned)(i & 7);
  __syncthreads();
  f32x4 acc[NACC];
#pragma unroll
  for (int m = 0; m < NACC; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
  u32x4 a[4], b[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) { a[r] = u32x4{0x3f803f80u, 0x3f803f80u, (unsigned)lane + r, 0x3f803f80u}; b[r] = u32x4{0x3f003f00u, 0x3f003f00u + r, 0x3f003f00u, 0x3f003f00u}; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < RD; ++r) {
      if (r & 1) b[(r >> 1) & 3] = *reinterpret_cast<const u32x4*>(smem + ((it * 528 + r * 2048 + lane * 16) & 16383));
      else a[(r >> 1) & 3] = *reinterpret_cast<const u32x4*>(smem + ((it * 1040 + r * 2048 + lane * 16) & 16383));
    }
    // CH == 0: rotation (a0 a1 .. aN-1) repeated; CH > 0: CH MFMAs on a0, CH on a1, ...  Same MFMA count either way
    constexpr int total = NACC * (CH > 0 ? CH : 3);
#pragma unroll
    for (int q = 0; q < total; ++q) {
      const int m = CH > 0 ? q / CH : q % NACC;
      if constexpr (SHAPE == 0) {
        acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[q & 3]), __builtin_bit_cast(bf16x8, b[(q >> 1) & 3]), acc[m], 0, 0, 0);
      } else {
        acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a[q & 3][q & 1]), __uint_as_float(b[(q >> 1) & 3][q & 3]), acc[m], 0, 0, 0);
      }
    }
  }
  float s = 0;
#pragma unroll
  for (int m = 0; m < NACC; ++m) s += acc[m][0] + acc[m][1] + acc[m][2] + acc[m][3];
  out[blockIdx.x * 256 + tid] = s;
}

template <int SHAPE, int NACC, int CH, int RD>
void run(int k, float* out) {
  const int iters = 8000, grid = 256 * k;
  const int lds = (160 * 1024 / k) - 1024;
  hipFuncSetAttribute((const void*)probe<SHAPE, NACC, CH, RD>, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL((probe<SHAPE, NACC, CH, RD>), dim3(grid), dim3(256), lds, 0, out, 10);
  hipEventRecord(a);
  hipLaunchKernelGGL((probe<SHAPE, NACC, CH, RD>), dim3(grid), dim3(256), lds, 0, out, iters);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const int total = NACC * (CH > 0 ? CH : 3);
  const double flops = (double)grid * 4 * iters * total * (SHAPE == 0 ? 16384.0 : 2048.0);
  printf("%s  %d accumulators, %-22s  %2d LDS reads per %2d MFMAs  waves/SIMD %d  %8.1f TF/s\n",
         SHAPE == 0 ? "bf16 16x16x32" : "f32 16x16x4  ", NACC, CH == 0 ? "rotating" : (CH == 3 ? "chains of 3" : CH == 2 ? "chains of 2" : CH == 8 ? "chains of 8" : "chains of 24"),
         RD, total, k, flops / ms / 1e9);
}

int main() {
  float* out;
  hipMalloc(&out, 2048 * 256 * 4);
  for (int k : {2, 3, 4}) {
    run<0, 5, 0, 0>(k, out); run<0, 5, 3, 0>(k, out); run<0, 5, 24, 0>(k, out);
    run<0, 5, 0, 6>(k, out); run<0, 5, 3, 6>(k, out); run<0, 5, 24, 6>(k, out);
    run<0, 8, 0, 6>(k, out); run<0, 8, 3, 6>(k, out);
    run<2, 8, 0, 2>(k, out); run<2, 8, 2, 2>(k, out); run<2, 8, 8, 2>(k, out); run<2, 2, 0, 1>(k, out);
  }
  return 0;
}

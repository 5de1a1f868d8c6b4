// bf16 MFMA rate by shape, accumulators in rotation, waves per SIMD and LDS operand reads (r4).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/bf16_occ tools/probes/mfma_bf16_occ_probe.hip && /tmp/bf16_occ
// SHAPE 0: v_mfma_f32_16x16x32_bf16 (the first-conv kernels), 1: v_mfma_f32_32x32x16_bf16 (xgemm.h).
// NACC accumulators; RD: 16-byte LDS operand reads per 6 MFMAs (0, 2, 4).  One 256-thread workgroup = one wave per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE, int NACC, int RD>
__global__ void __launch_bounds__(256) probe(const float* __restrict__ in, float* __restrict__ out, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  unsigned* lu = reinterpret_cast<unsigned*>(smem);
  for (int i = tid; i < 4096; i += 256) lu[i] = 0x3f803f80u + (unsigned)(i & 7);
  __syncthreads();
  typedef typename std::conditional<SHAPE == 0, f32x4, f32x16>::type acc_t;
  acc_t acc[NACC];
#pragma unroll
  for (int m = 0; m < NACC; ++m)
#pragma unroll
    for (int e = 0; e < (SHAPE == 0 ? 4 : 16); ++e) acc[m][e] = 0.f;
  u32x4 a[2] = {u32x4{0x3f803f80u, 0x3f803f80u, (unsigned)lane, 0x3f803f80u}, u32x4{0x3f803f80u, 1u, 2u, 0x3f803f80u}};
  u32x4 b[2] = {u32x4{0x3f003f00u, 0x3f003f00u, 0x3f003f00u, 0x3f003f00u}, u32x4{0x3e803e80u, 3u, 4u, 0x3e803e80u}};
  for (int it = 0; it < iters; ++it) {
    if (RD >= 2) {
      a[0] = *reinterpret_cast<const u32x4*>(smem + ((it * 1040 + lane * 16) & 16383));
      b[0] = *reinterpret_cast<const u32x4*>(smem + ((it * 528 + 4096 + lane * 16) & 16383));
    }
    if (RD >= 4) {
      a[1] = *reinterpret_cast<const u32x4*>(smem + ((it * 272 + 8192 + lane * 16) & 16383));
      b[1] = *reinterpret_cast<const u32x4*>(smem + ((it * 144 + 12288 + lane * 16) & 16383));
    }
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      const int m = q % NACC;
      const bf16x8 av = __builtin_bit_cast(bf16x8, a[q & 1]), bv = __builtin_bit_cast(bf16x8, b[(q >> 1) & 1]);
      if constexpr (SHAPE == 0) acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, acc[m], 0, 0, 0);
      else acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[m], 0, 0, 0);
    }
  }
  float s = 0;
#pragma unroll
  for (int m = 0; m < NACC; ++m)
#pragma unroll
    for (int e = 0; e < (SHAPE == 0 ? 4 : 16); ++e) s += acc[m][e];
  out[blockIdx.x * 256 + tid] = s;
}

template <int SHAPE, int NACC, int RD>
void run(int k, const float* in, float* out) {
  const int iters = 20000, grid = 256 * k;
  const int lds = (160 * 1024 / k) - 1024;
  hipFuncSetAttribute((const void*)probe<SHAPE, NACC, RD>, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL((probe<SHAPE, NACC, RD>), dim3(grid), dim3(256), lds, 0, in, out, 10);
  hipEventRecord(a);
  hipLaunchKernelGGL((probe<SHAPE, NACC, RD>), dim3(grid), dim3(256), lds, 0, in, out, iters);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double flops = (double)grid * 4 * iters * 6 * (SHAPE == 0 ? 16384.0 : 32768.0);
  printf("%s  acc %d  16-B lds reads per 6 MFMAs %d  waves/SIMD %d   %8.3f ms  %7.1f TF/s\n",
         SHAPE == 0 ? "16x16x32" : "32x32x16", NACC, RD, k, ms, flops / ms / 1e9);
}

int main() {
  float* in; float* out;
  hipMalloc(&in, 8192 * 4); hipMalloc(&out, 2048 * 256 * 4);
  for (int k : {1, 2, 3, 4}) {
    run<0, 1, 0>(k, in, out); run<0, 2, 0>(k, in, out); run<0, 3, 0>(k, in, out); run<0, 6, 0>(k, in, out);
    run<0, 3, 2>(k, in, out); run<0, 3, 4>(k, in, out); run<0, 6, 4>(k, in, out);
    run<1, 1, 0>(k, in, out); run<1, 2, 0>(k, in, out); run<1, 3, 0>(k, in, out); run<1, 3, 2>(k, in, out); run<1, 3, 4>(k, in, out);
  }
  return 0;
}

// How many waves per SIMD does v_mfma_f32_16x16x4_f32 need, and what does an LDS operand read cost beside it? (r4)
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_occ tools/probes/mfma_occ_probe.hip && /tmp/mfma_occ
// One 256-thread workgroup = one wave per SIMD; grid = 256 k with `lds_pad` bytes of dynamic LDS so that exactly k
// workgroups fit a CU (160 KB / k).  NACC independent accumulators; RD ds_read_b128 per 8 MFMAs (0, 1, 2).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC, int RD, int NV = 0>
__global__ void __launch_bounds__(256) probe(const float* __restrict__ in, float* __restrict__ out, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  float* lf = reinterpret_cast<float*>(smem);
  // (the operand VALUES matter: the chip clocks to its power budget -- `in` is constant or pseudo-random, chosen by the host)
  for (int i = tid; i < 4096; i += 256) lf[i] = in[(i * 7 + blockIdx.x) & 8191];
  __syncthreads();
  f32x4 acc[NACC];
#pragma unroll
  for (int m = 0; m < NACC; ++m) acc[m] = f32x4{0, 0, 0, 0};
  f32x4 a = {1.0f + lane, 2.0f, 3.0f, 4.0f};
  f32x4 b = {0.5f, 0.25f, 0.125f, 1.0f};
  int junk[4] = {lane, tid, 3, 4};
  for (int it = 0; it < iters; ++it) {
    if (RD >= 1) a = *reinterpret_cast<const f32x4*>(smem + ((it * 1040 + lane * 16) & 16383));
    if (RD >= 2) b = *reinterpret_cast<const f32x4*>(smem + ((it * 528 + 4096 + lane * 16) & 16383));
#pragma unroll
    for (int v = 0; v < NV; ++v) asm volatile("v_add_u32 %0, %0, %1" : "+v"(junk[v & 3]) : "v"(lane));   // NV plain VALU per 8 MFMAs
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int m = (e * 2 + j) % NACC;
        acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[j], a[e], acc[m], 0, 0, 0);
      }
  }
  float s = 0;
#pragma unroll
  for (int m = 0; m < NACC; ++m) s += acc[m][0] + acc[m][1] + acc[m][2] + acc[m][3];
  out[blockIdx.x * 256 + tid] = s + (float)(junk[0] + junk[1] + junk[2] + junk[3]);
}

template <int NACC, int RD, int NV = 0>
void run(int k, const float* in, float* out) {
  const int iters = 20000, grid = 256 * k;
  const int lds = (160 * 1024 / k) - 1024;                   // k workgroups per CU, not k + 1
  hipFuncSetAttribute((const void*)probe<NACC, RD, NV>, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL((probe<NACC, RD, NV>), dim3(grid), dim3(256), lds, 0, in, out, 10);
  hipEventRecord(a);
  hipLaunchKernelGGL((probe<NACC, RD, NV>), dim3(grid), dim3(256), lds, 0, in, out, iters);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double flops = (double)grid * 4 * iters * 8 * 2048.0;
  printf("acc %d  lds reads per 8 MFMAs %d  VALU per 8 MFMAs %2d  waves/SIMD %d   %8.3f ms  %7.1f TF/s\n", NACC, RD, NV, k, ms, flops / ms / 1e9);
}

int main(int argc, char** argv) {
  float* in; float* out;
  hipMalloc(&in, 8192 * 4); hipMalloc(&out, 2048 * 256 * 4);
  hipMemset(in, 0x3c, 8192 * 4);
  if (argc > 1) {                                            // any argument: pseudo-random operands in (-1, 1)
    static float h[8192];
    unsigned s = 12345u;
    for (int i = 0; i < 8192; ++i) { s = s * 1664525u + 1013904223u; h[i] = ((int)(s >> 8) - (1 << 23)) / (float)(1 << 23); }
    hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    printf("operands: pseudo-random\n");
  } else printf("operands: constant\n");
  for (int k : {1, 2, 3, 4}) {
    run<8, 0>(k, in, out); run<2, 0>(k, in, out); run<8, 1>(k, in, out); run<8, 2>(k, in, out); run<2, 1>(k, in, out);
  }
  for (int k : {2, 4}) {
    run<4, 0>(k, in, out); run<2, 1, 2>(k, in, out); run<2, 1, 4>(k, in, out); run<2, 1, 8>(k, in, out); run<2, 1, 16>(k, in, out);
    run<8, 2, 4>(k, in, out); run<8, 2, 8>(k, in, out);
  }
  return 0;
}

// Which XCD does workgroup i of a 1-D grid land on?  (hipcc --offload-arch=gfx950 xcc_probe.hip -o xcc_probe)
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(int* out) {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  if (threadIdx.x == 0) out[blockIdx.x] = (int)(v & 15);
}
int main() {
  const int n = 256;
  int* d; (void)hipMalloc(&d, n * 4);
  (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024);
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(k, dim3(n), dim3(256), rep ? 120 * 1024 : 0, 0, d);
    int h[n]; (void)hipMemcpy(h, d, n * 4, hipMemcpyDeviceToHost);
    int mism = 0;
    for (int i = 0; i < n; ++i) mism += h[i] != h[i % 8];
    printf("lds %3d KB: first 16:", rep ? 120 : 0);
    for (int i = 0; i < 16; ++i) printf(" %d", h[i]);
    printf("  | blocks whose XCD differs from block (i %% 8)'s: %d\n", mism);
  }
  return 0;
}

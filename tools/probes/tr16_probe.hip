// Probe: which lane's address feeds which element of ds_read_b64_tr_b16 (gfx950)?
//   hipcc --offload-arch=gfx950 -O3 tr16_probe.hip -o /tmp/tr16 && /tmp/tr16
// Every lane reads 8 bytes at its own address (lane l -> halfwords 4 l .. 4 l + 3 of an LDS array holding its own
// index); the result names, per lane and element, the source halfword.  Expected (wgx.h relies on it): element j of
// lane l comes from lane 16 (l / 16) + 4 j + (l % 16) / 4, halfword (l % 4) of that lane's four.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void probe(short* out) {
  __shared__ __attribute__((aligned(16))) short lds[256];
  for (int i = threadIdx.x; i < 256; i += 64) lds[i] = (short)i;
  __syncthreads();
  const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + threadIdx.x * 4));
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = v[j];
}
int main() {
  short* d; hipMalloc(&d, 512);
  probe<<<1, 64>>>(d);
  short h[256]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l)
    for (int j = 0; j < 4; ++j) {
      const int want = 4 * (16 * (l / 16) + 4 * j + (l % 16) / 4) + (l % 4);
      if (h[4 * l + j] != want) { if (bad++ < 8) printf("lane %d elem %d: halfword %d, expected %d\n", l, j, h[4 * l + j], want); }
    }
  printf("tr16 probe: %s (%d mismatches); lane 5 reads halfwords %d %d %d %d\n", bad ? "DIFFERENT" : "as expected", bad, h[20], h[21], h[22], h[23]);
  return bad != 0;
}

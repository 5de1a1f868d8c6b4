// Probe: raw buffer loads on gfx950 -- do out-of-range voffsets return zeros, with an SGPR soffset, and with a
// descriptor base below the allocation?   hipcc --offload-arch=gfx950 -O3 buffer_oob_probe.hip -o /tmp/bo && /tmp/bo
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
__global__ void probe(const float* a, int nbytes, int minoff_floats, unsigned soff, unsigned* out) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)(a + minoff_floats), 0, nbytes - minoff_floats * 4, 0x00020000);
  unsigned voff = threadIdx.x * 16u;
  if (threadIdx.x & 1) voff = 0x80000000u;
  if (threadIdx.x == 2) voff = (unsigned)(nbytes - minoff_floats * 4);      // first byte past num_records
  u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
  out[threadIdx.x * 4 + 0] = v[0]; out[threadIdx.x * 4 + 1] = v[1]; out[threadIdx.x * 4 + 2] = v[2]; out[threadIdx.x * 4 + 3] = v[3];
}
int main() {
  const int n = 4096;
  std::vector<float> h(n); for (int i = 0; i < n; ++i) h[i] = 1.0f + i;
  float* d; unsigned* o; hipMalloc(&d, n * 4 + 65536); hipMalloc(&o, 64 * 16);
  float* a = d + 8192;                                    // leave room below for the negative-base case
  hipMemset(d, 0, n * 4 + 65536);
  hipMemcpy(a, h.data(), n * 4, hipMemcpyHostToDevice);
  for (int cfg = 0; cfg < 3; ++cfg) {
    const int minoff = cfg == 2 ? -320 : 0; const unsigned soff = cfg == 0 ? 0 : 1280 + (cfg == 2 ? 0 : 0);
    probe<<<1, 64>>>(a, 1024 * 4 /* only the first 1024 floats are "in range" */, minoff, soff, o);
    std::vector<unsigned> r(256); hipMemcpy(r.data(), o, 1024, hipMemcpyDeviceToHost);
    printf("cfg %d (minoff %d, soff %u): lane0 %.0f lane1(oob) %u lane2(just past) %u lane4 %.0f err %s\n", cfg, minoff, soff,
           *(float*)&r[0], r[4], r[8], *(float*)&r[16], hipGetErrorString(hipDeviceSynchronize()));
  }
  return 0;
}

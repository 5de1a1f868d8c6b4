// Probe: the streaming weight-gradient kernel (csrc/wsw.h) at the cfg2 second-conv shape: prefetch depth, grid size,
// and the kernel with its MFMAs / its loads left out.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iseed_rl_amd/csrc -Iinclude tools/probes/wsw_probe.hip \
//         seed_rl_amd/csrc/error.cpp -o tools/probes/wsw_probe.bin
#include "wsw.h"   // (moved here from csrc/ in r5; build with -I seed_rl_amd/csrc -I tools/probes)
#include <vector>
using namespace seedhip;
using namespace seedhip::wsw;

template <int DEPTH, int EXP>
static float run(Params p, int grid, int reps) {
  const size_t lds = (size_t)4 * p.Q * sizeof(unsigned);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((wsw_kernel<false, DEPTH, EXP>), dim3(grid), dim3(256), lds, 0, p);
  (void)hipEventRecord(e0, 0);
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((wsw_kernel<false, DEPTH, EXP>), dim3(grid), dim3(256), lds, 0, p);
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) printf("error: %s\n", hipGetErrorString(e));
  return ms / reps * 1e3f;
}

int main() {
  const int n_img = 21 * 512;
  seedhip_conv_geom g;
  memset(&g, 0, sizeof(g));
  g.n_img = n_img; g.ih = 20; g.iw = 20; g.cin = 16; g.oh = 9; g.ow = 9; g.kh = 4; g.kw = 4; g.stride = 2; g.cout = 32;
  g.ld_in = 16; g.ld_out = 32;
  const size_t nx = (size_t)n_img * 400 * 16, ny = (size_t)n_img * 81 * 32;
  float *x, *y, *pw, *pb;
  (void)hipMalloc(&x, nx * 4); (void)hipMalloc(&y, ny * 4); (void)hipMalloc(&pw, (size_t)2048 * 8192 * 4); (void)hipMalloc(&pb, 2048 * 32 * 4);
  std::vector<float> h(nx);
  for (size_t i = 0; i < nx; ++i) h[i] = (float)((i * 2654435761u >> 16) & 1023) / 1024.f - 0.3f;
  (void)hipMemcpy(x, h.data(), nx * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(y, h.data(), ny * 4, hipMemcpyHostToDevice);
  Params p;
  if (!plan(p, &g)) { printf("plan failed\n"); return 1; }
  p.X = x; p.dY = y; p.partial_w = pw; p.partial_b = pb; p.in_relu = 0;
  const double flops = 2.0 * n_img * 81 * 32 * 256;
#define R(D, E, G, what) { float us = run<D, E>(p, G, 20); printf("  depth %2d grid %4d %-28s %7.1f us  %6.1f TF/s-equivalent\n", D, G, what, us, flops / us / 1e6); }
  R(8, 0, 512, "full") R(8, 0, 256, "full") R(8, 0, 768, "full") R(8, 0, 1024, "full")
  R(4, 0, 512, "full") R(4, 0, 1024, "full") R(12, 0, 512, "full") R(12, 0, 1024, "full") R(16, 0, 512, "full")
  R(8, 1, 512, "no MFMA") R(8, 1, 1024, "no MFMA") R(8, 2, 512, "no loads (MFMA only)") R(8, 2, 1024, "no loads (MFMA only)")
  R(12, 1, 1024, "no MFMA")
  return 0;
}

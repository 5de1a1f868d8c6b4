// Probe: the table-driven weight-stationary conv kernel (csrc/wsgemm.h: ws_tab_kernel) at the cfg2 second-conv shape,
// with one ingredient left out at a time -- which of MFMA issue, global loads, stores, mask loads, LDS fragment reads,
// LDS staging writes sets the 0.18 ms of the data gradient / 0.14 ms of the forward?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iseed_rl_amd/csrc -Iinclude tools/probes/ws_probe.hip \
//         seed_rl_amd/csrc/error.cpp -o tools/probes/ws_probe.bin && tools/probes/ws_probe.bin
#include "wsgemm.h"
#include <vector>
using namespace seedhip;
using namespace seedhip::wsgemm;

template <int NT, int NKT, int MODE, int EXP>
static float run(Params p, int grid, size_t lds, int reps) {
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute((const void*)ws_tab_kernel<NT, NKT, MODE, EXP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((ws_tab_kernel<NT, NKT, MODE, EXP>), dim3(grid), dim3(512), lds, 0, p);
  hipEventRecord(e0, 0);
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((ws_tab_kernel<NT, NKT, MODE, EXP>), dim3(grid), dim3(512), lds, 0, p);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) printf("error: %s\n", hipGetErrorString(e));
  return ms / reps * 1e3f;
}

int main(int argc, char** argv) {
  const int n_img = argc > 1 ? atoi(argv[1]) : 21 * 512;
  int grid_override = argc > 2 ? atoi(argv[2]) : 0;
  seedhip_conv_geom g;
  memset(&g, 0, sizeof(g));
  g.n_img = n_img; g.ih = 20; g.iw = 20; g.cin = 16; g.oh = 9; g.ow = 9; g.kh = 4; g.kw = 4; g.stride = 2; g.cout = 32;
  g.ld_in = 16; g.ld_out = 32;
  const size_t nx = (size_t)n_img * 400 * 16, ny = (size_t)n_img * 81 * 32;
  float *x, *y, *w, *dx, *bias;
  hipMalloc(&x, nx * 4); hipMalloc(&y, ny * 4); hipMalloc(&dx, nx * 4); hipMalloc(&w, 4 * 4 * 16 * 32 * 4); hipMalloc(&bias, 128);
  std::vector<float> h(nx);
  for (size_t i = 0; i < nx; ++i) h[i] = (float)((i * 2654435761u >> 16) & 1023) / 1024.f - 0.3f;
  hipMemcpy(x, h.data(), nx * 4, hipMemcpyHostToDevice);
  hipMemcpy(y, h.data(), ny * 4, hipMemcpyHostToDevice);
  hipMemcpy(w, h.data(), 8192 * 4, hipMemcpyHostToDevice);
  hipMemcpy(bias, h.data(), 128, hipMemcpyHostToDevice);
  const double flops = 2.0 * n_img * 81 * 32 * 256;
  {
    Params p; Plan pl = plan_fwd(p, &g);
    p.A = x; p.W = w; p.C = y; p.bias = bias; p.out_relu = 1;
    p.ntiles = (p.M + 15) / 16;
    const int wgs = (p.ntiles + 7) / 8;
    int grid = wgs < 512 ? wgs : 512; if (grid_override) grid = grid_override;
    const size_t lds = ((size_t)p.N * (p.K + 8) + 8 * 16 * LDA + (size_t)p.gh * p.gw * 2) * 4;
    printf("forward  M=%d N=%d K=%d grid=%d lds=%zu pl.ok=%d\n", p.M, p.N, p.K, grid, lds, (int)pl.ok);
#define R(E, what) { float us = run<2, 8, 0, E>(p, grid, lds, 20); printf("  fwd   %-44s %7.1f us  %6.1f TF/s-equivalent\n", what, us, flops / us / 1e6); }
    R(0, "full") R(1, "no MFMA") R(2, "no global A loads") R(4, "no stores") R(16, "no LDS fragment reads") R(32, "no LDS staging writes")
    R(1 | 16 | 32, "memory only (no MFMA, no LDS)") R(2 | 4, "no global traffic (LDS + MFMA)") R(2 | 4 | 16 | 32, "MFMA only")
    R(1 | 2 | 4, "LDS only")
#undef R
  }
  {
    Params p; Plan pl = plan_dgrad(p, &g);
    p.A = y; p.W = w; p.C = dx; p.mask = x;
    p.ntiles = (p.M + 15) / 16;
    const int wgs = (p.ntiles + 7) / 8;
    int grid = wgs < 512 ? wgs : 512; if (grid_override) grid = grid_override;
    const size_t lds = ((size_t)p.N * (p.K + 8) + 8 * 16 * LDA + (size_t)p.gh * p.gw * (p.nkt + 1)) * 4;
    printf("data gradient  M=%d N=%d K=%d grid=%d lds=%zu pl.ok=%d\n", p.M, p.N, p.K, grid, lds, (int)pl.ok);
#define R(E, what) { float us = run<4, 4, 1, E>(p, grid, lds, 20); printf("  dgrad %-44s %7.1f us  %6.1f TF/s-equivalent\n", what, us, flops / us / 1e6); }
    R(0, "full") R(1, "no MFMA") R(2, "no global A loads") R(4, "no stores") R(8, "no mask loads") R(4 | 8, "no stores, no mask loads")
    R(16, "no LDS fragment reads") R(32, "no LDS staging writes")
    R(1 | 16 | 32, "memory only (no MFMA, no LDS)") R(2 | 4 | 8, "no global traffic (LDS + MFMA)") R(2 | 4 | 8 | 16 | 32, "MFMA only")
    R(1 | 2 | 4 | 8, "LDS only")
#undef R
  }
  return 0;
}

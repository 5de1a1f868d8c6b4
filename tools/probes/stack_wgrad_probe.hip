// Probe: the bf16x3 first-conv weight gradient (csrc/stackconv.hip: stackconv_wgrad_tr_kernel) at cfg2 (T1 = 21, B = 512):
// launch time and s_memtime stamps of workgroup 0's waves (where does a step go: prepare / multiply / frame / barrier).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iseed_rl_amd/csrc -Iinclude tools/probes/stack_wgrad_probe.hip \
//         seed_rl_amd/csrc/error.cpp -o tools/probes/stack_wgrad_probe.bin
#include "../../seed_rl_amd/csrc/stackconv.hip"
#include <vector>
#include <algorithm>
using namespace seedhip::stackconv;

template <int EXP>
static float run(Params p, int grid, int reps) {
  (void)hipFuncSetAttribute((const void*)stackconv_wgrad_tr_kernel<16, EXP>, hipFuncAttributeMaxDynamicSharedMemorySize, kTrLds);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((stackconv_wgrad_tr_kernel<16, EXP>), dim3(grid, 1, 1), dim3(512), kTrLds, 0, p);
  (void)hipEventRecord(e0, 0);
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((stackconv_wgrad_tr_kernel<16, EXP>), dim3(grid, 1, 1), dim3(512), kTrLds, 0, p);
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) printf("error: %s\n", hipGetErrorString(e));
  return ms / reps * 1e3f;
}

int main(int argc, char** argv) {
  const int T1 = 21, B = argc > 1 ? atoi(argv[1]) : 512;
  seedhip_stack_conv_geom g;
  memset(&g, 0, sizeof(g));
  g.T = T1; g.B = B; g.ih = 84; g.iw = 84; g.oh = 20; g.ow = 20; g.kh = 8; g.kw = 8; g.stride = 4; g.cout = 16; g.ld_out = 16;
  uint8_t *frames, *nvalid; float *dy, *pw; unsigned* st;
  const size_t nf = (size_t)(T1 + 3) * B * 7056, no = (size_t)T1 * B * 400 * 16;
  (void)hipMalloc(&frames, nf); (void)hipMalloc(&nvalid, T1 * B); (void)hipMalloc(&dy, no * 4);
  std::vector<uint8_t> h(nf);
  for (size_t i = 0; i < nf; ++i) h[i] = (uint8_t)((i * 2654435761u) >> 13);
  (void)hipMemcpy(frames, h.data(), nf, hipMemcpyHostToDevice);
  std::vector<uint8_t> nv(T1 * B, 4);
  (void)hipMemcpy(nvalid, nv.data(), nv.size(), hipMemcpyHostToDevice);
  std::vector<float> hd(no);
  for (size_t i = 0; i < no; ++i) hd[i] = (float)((i * 37) % 211) / 211.f - 0.5f;
  (void)hipMemcpy(dy, hd.data(), no * 4, hipMemcpyHostToDevice);
  Params p = make_params(&g, frames, nvalid);
  const int grid = wgrad_grid(&g, &p.spc, &p.items);
  (void)hipMalloc(&pw, (size_t)grid * (256 * 16 + 16) * 4); (void)hipMalloc(&st, 8 * 64 * 8 * 4);
  p.dy = dy; p.partial_w = pw; p.partial_b = pw + (size_t)grid * 256 * 16;
  printf("stack conv wgrad (transposing reads): T1=%d B=%d grid=%d spc=%d items=%d lds=%d\n", T1, B, grid, p.spc, p.items, kTrLds);
  std::vector<float> r, r2;
  for (int i = 0; i < 7; ++i) { r.push_back(run<0>(p, grid, 20)); r2.push_back(run<64>(p, grid, 20)); }
  printf("  full, us per launch, seven interleaved rounds:"); for (float v : r) printf(" %.1f", v); printf("\n");
  printf("  every wave prepares first:                   "); for (float v : r2) printf(" %.1f", v); printf("\n");
  std::sort(r.begin(), r.end()); std::sort(r2.begin(), r2.end());
  printf("  medians: %.1f / %.1f us\n", r[3], r2[3]);
  Params ps = p; ps.partial_b = (float*)st;
  (void)hipMemset(st, 0, 8 * 64 * 8 * 4);
  run<32>(ps, grid, 1);
  std::vector<unsigned> hs(8 * 64 * 8);
  (void)hipMemcpy(hs.data(), st, hs.size() * 4, hipMemcpyDeviceToHost);
  printf("  stamps, cycles, mean of steps 2..19: first half (waves 0-3 prepare, 4-7 multiply) | second half | frame -> ring | barrier | step\n");
  for (int w = 0; w < 8; ++w) {
    double d[5] = {0, 0, 0, 0, 0};
    for (int t = 2; t < 20; ++t) {
      const unsigned* q = &hs[(w * 64 + t) * 8];
      d[0] += (unsigned)(q[1] - q[0]); d[1] += (unsigned)(q[2] - q[1]); d[2] += (unsigned)(q[3] - q[2]); d[3] += (unsigned)(q[4] - q[3]);
      d[4] += (unsigned)(hs[(w * 64 + t + 1) * 8] - q[0]);
    }
    printf("    wave %d: %7.0f %7.0f %7.0f %7.0f %8.0f\n", w, d[0] / 18, d[1] / 18, d[2] / 18, d[3] / 18, d[4] / 18);
  }
  (void)hipMemset(st, 0, 8 * 64 * 8 * 4);
  run<32 | 64>(ps, grid, 1);
  (void)hipMemcpy(hs.data(), st, hs.size() * 4, hipMemcpyDeviceToHost);
  printf("  every wave prepares first: prepare + requests | multiply | frame -> ring | barrier | step\n");
  for (int w = 0; w < 8; ++w) {
    double d[5] = {0, 0, 0, 0, 0};
    for (int t = 2; t < 20; ++t) {
      const unsigned* q = &hs[(w * 64 + t) * 8];
      d[0] += (unsigned)(q[1] - q[0]); d[1] += (unsigned)(q[2] - q[1]); d[2] += (unsigned)(q[3] - q[2]); d[3] += (unsigned)(q[4] - q[3]);
      d[4] += (unsigned)(hs[(w * 64 + t + 1) * 8] - q[0]);
    }
    printf("    wave %d: %7.0f %7.0f %7.0f %7.0f %8.0f\n", w, d[0] / 18, d[1] / 18, d[2] / 18, d[3] / 18, d[4] / 18);
  }
  return 0;
}

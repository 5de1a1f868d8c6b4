// Probe: the bf16x3 first-conv forward (csrc/stackconv.hip: stackconv_fwd_bf16r_kernel) at cfg2 (T1 = 21, B = 512),
// one ingredient left out at a time.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iseed_rl_amd/csrc -Iinclude tools/probes/stack_probe.hip \
//         seed_rl_amd/csrc/error.cpp -o tools/probes/stack_probe.bin
#include "../../seed_rl_amd/csrc/stackconv.hip"
#include <vector>
#include <algorithm>
using namespace seedhip::stackconv;

template <int EXP>
static float run(Params p, int grid, size_t lds, int reps) {
  (void)hipFuncSetAttribute((const void*)stackconv_fwd_bf16r_kernel<EXP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(stackconv_fwd_bf16r_kernel<EXP>, dim3(grid, 1, 1), dim3(kThreads), lds, 0, p);
  (void)hipEventRecord(e0, 0);
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(stackconv_fwd_bf16r_kernel<EXP>, dim3(grid, 1, 1), dim3(kThreads), lds, 0, p);
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) printf("error: %s\n", hipGetErrorString(e));
  return ms / reps * 1e3f;
}

template <int EXP, bool BITS = false, int MODE = 0>
static float run8(Params p, int grid, int reps) {
#define stackconv_fwd_w8_kernel_ (stackconv_fwd_w8_kernel<EXP, BITS, true, MODE>)
  (void)hipFuncSetAttribute((const void*)stackconv_fwd_w8_kernel_, hipFuncAttributeMaxDynamicSharedMemorySize, kW8Lds);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(stackconv_fwd_w8_kernel_, dim3(grid, 1, 1), dim3(kW8Threads), kW8Lds, 0, p);
  (void)hipEventRecord(e0, 0);
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(stackconv_fwd_w8_kernel_, dim3(grid, 1, 1), dim3(kW8Threads), kW8Lds, 0, p);
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
  const int wgs_per_cu = argc > 2 ? atoi(argv[2]) : 2;
  seedhip_stack_conv_geom g;
  memset(&g, 0, sizeof(g));
  g.T = T1; g.B = B; g.ih = 84; g.iw = 84; g.oh = 20; g.ow = 20; g.kh = 8; g.kw = 8; g.stride = 4; g.cout = 16; g.ld_out = 16;
  uint8_t *frames, *nvalid; float *w, *bias, *out;
  const size_t nf = (size_t)(T1 + 3) * B * 7056, no = (size_t)T1 * B * 400 * 16;
  (void)hipMalloc(&frames, nf); (void)hipMalloc(&nvalid, T1 * B); (void)hipMalloc(&w, 4096 * 4); (void)hipMalloc(&bias, 64); (void)hipMalloc(&out, no * 4);
  std::vector<uint8_t> h(nf);
  for (size_t i = 0; i < nf; ++i) h[i] = (uint8_t)((i * 2654435761u) >> 13);
  (void)hipMemcpy(frames, h.data(), nf, hipMemcpyHostToDevice);
  std::vector<uint8_t> nv(T1 * B, 4);
  (void)hipMemcpy(nvalid, nv.data(), nv.size(), hipMemcpyHostToDevice);
  std::vector<float> hw(4096);
  for (int i = 0; i < 4096; ++i) hw[i] = (float)((i * 37) % 211) / 211.f - 0.5f;
  (void)hipMemcpy(w, hw.data(), 4096 * 4, hipMemcpyHostToDevice); (void)hipMemcpy(bias, hw.data(), 64, hipMemcpyHostToDevice);
  Params p = make_params(&g, frames, nvalid);
  p.w = w; p.bias = bias; p.out = out; p.out_relu = 1;
  int grid;
  decompose(p.T1, p.B, max_grid_for(wgs_per_cu), &p.spc, &p.items, &grid);
  const size_t lds = (size_t)kGroups * 64 * 16 + (size_t)kWaves * kWaveRing16;
  const double flops = 2.0 * T1 * B * 400 * 16 * 256;
  printf("stack conv fwd bf16x3: T1=%d B=%d grid=%d spc=%d items=%d lds=%zu\n", T1, B, grid, p.spc, p.items, lds);
#define R(E, what) { float us = run<E>(p, grid, lds, 20); printf("  %-46s %7.1f us  %6.1f algorithmic TF/s\n", what, us, flops / us / 1e6); }
  R(0, "full") R(1, "no MFMA") R(2, "no global band prefetch") R(4, "no stores") R(8, "no LDS pixel reads") R(16, "no band convert + LDS store")
  R(2 | 4, "no global traffic") R(1 | 8, "no MFMA, no LDS pixel reads") R(2 | 4 | 16, "LDS reads + MFMA only") R(2 | 4 | 8 | 16, "MFMA only (+ w lo reads)")
  R(1 | 8 | 16, "memory only") R(1 | 2 | 4 | 16, "LDS pixel reads only")
  {
    Params q = p;
    int grid8;
    decompose(q.T1, (q.B + 1) / 2, max_grid_for(1), &q.spc, &q.items, &grid8);
    printf("eight waves, two columns per workgroup: grid=%d spc=%d items=%d lds=%d\n", grid8, q.spc, q.items, kW8Lds);
    {   // the eight-wave kernel against the five-wave kernel: bit-identical outputs
      float* out2; (void)hipMalloc(&out2, no * 4);
      (void)hipMemset(out, 0, no * 4); (void)hipMemset(out2, 0xff, no * 4);
      (void)hipFuncSetAttribute((const void*)stackconv_fwd_bf16r_kernel<0, false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      Params p5 = p; p5.buf32 = 1;
      hipLaunchKernelGGL((stackconv_fwd_bf16r_kernel<0, false, true, true>), dim3(grid, 1, 1), dim3(kThreads), lds, 0, p5);
      Params q2 = q; q2.out = out2;
      (void)hipFuncSetAttribute((const void*)stackconv_fwd_w8_kernel<0, false, true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, kW8Lds);
      hipLaunchKernelGGL((stackconv_fwd_w8_kernel<0, false, true, 2>), dim3(grid8, 1, 1), dim3(kW8Threads), kW8Lds, 0, q2);
      std::vector<float> a(no), c(no);
      (void)hipMemcpy(a.data(), out, no * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(c.data(), out2, no * 4, hipMemcpyDeviceToHost);
      size_t bad = 0, first = 0;
      for (size_t i = 0; i < no; ++i) if (memcmp(&a[i], &c[i], 4)) { if (!bad) first = i; ++bad; }
      printf("  w8 vs five-wave outputs: %zu of %zu differ (first at %zu)\n", bad, no, first);
      // ... and the byte masks
      unsigned char *ba, *bb; (void)hipMalloc(&ba, no / 4); (void)hipMalloc(&bb, no / 4);
      (void)hipMemset(ba, 0, no / 4); (void)hipMemset(bb, 0xff, no / 4);
      p5.relu_bits = ba; q2.relu_bits = bb;
      (void)hipFuncSetAttribute((const void*)stackconv_fwd_bf16r_kernel<0, true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL((stackconv_fwd_bf16r_kernel<0, true, true, true>), dim3(grid, 1, 1), dim3(kThreads), lds, 0, p5);
      (void)hipFuncSetAttribute((const void*)stackconv_fwd_w8_kernel<0, true, true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, kW8Lds);
      hipLaunchKernelGGL((stackconv_fwd_w8_kernel<0, true, true, 2>), dim3(grid8, 1, 1), dim3(kW8Threads), kW8Lds, 0, q2);
      std::vector<unsigned char> ha(no / 4), hb(no / 4);
      (void)hipMemcpy(ha.data(), ba, no / 4, hipMemcpyDeviceToHost); (void)hipMemcpy(hb.data(), bb, no / 4, hipMemcpyDeviceToHost);
      (void)hipMemcpy(c.data(), out2, no * 4, hipMemcpyDeviceToHost);
      bad = 0;
      for (size_t i = 0; i < no / 4; ++i) if (ha[i] != hb[i]) { if (!bad) first = i; ++bad; }
      size_t bad2 = 0;
      for (size_t i = 0; i < no; ++i) if (memcmp(&a[i], &c[i], 4)) ++bad2;
      printf("  w8 vs five-wave byte masks: %zu of %zu differ (first at %zu); outputs of that launch: %zu differ\n", bad, no / 4, first, bad2);
    }
#define R8(E, what) { float us = run8<E>(q, grid8, 20); printf("  %-46s %7.1f us  %6.1f algorithmic TF/s\n", what, us, flops / us / 1e6); }
    unsigned char* bits; (void)hipMalloc(&bits, no / 4); q.relu_bits = bits;
#define R8X(E, B_, M_, what) { float us = run8<E, B_, M_>(q, grid8, 20); printf("  %-46s %7.1f us  %6.1f algorithmic TF/s\n", what, us, flops / us / 1e6); }
    R8X(0, true, 0, "full + byte mask")
    {
      unsigned* st; (void)hipMalloc(&st, 8 * 64 * 8 * 4);
      Params qs = q; qs.partial_w = (float*)st;
      auto show = [&](const char* what) {
        std::vector<unsigned> h(8 * 64 * 8);
        (void)hipMemcpy(h.data(), st, h.size() * 4, hipMemcpyDeviceToHost);
        printf("  stamps (%s): per wave, mean cycles over steps 2..19: mma | wait | stage | emit+issue | whole step\n", what);
        for (int w = 0; w < 8; ++w) {
          double d[5] = {0, 0, 0, 0, 0};
          for (int t = 2; t < 20; ++t) {
            const unsigned* r = &h[(w * 64 + t) * 8];
            d[0] += (unsigned)(r[1] - r[0]); d[1] += (unsigned)(r[2] - r[1]); d[2] += (unsigned)(r[3] - r[2]); d[3] += (unsigned)(r[4] - r[3]);
            d[4] += (unsigned)(h[(w * 64 + t + 1) * 8] - r[0]);
          }
          printf("    wave %d: %7.0f %7.0f %7.0f %7.0f %8.0f\n", w, d[0] / 18, d[1] / 18, d[2] / 18, d[3] / 18, d[4] / 18);
        }
      };
      (void)hipMemset(st, 0, 8 * 64 * 8 * 4); run8<32, false, 0>(qs, grid8, 1); show("mode 0");
      (void)hipMemset(st, 0, 8 * 64 * 8 * 4); run8<32, false, 2>(qs, grid8, 1); show("mode 2");
      (void)hipMemset(st, 0, 8 * 64 * 8 * 4); run8<32 | 4, false, 2>(qs, grid8, 1); show("mode 2, no stores");
    }
    {
      printf("  A/B, five interleaved rounds of 20 launches each, median us:\n");
      std::vector<float> r[6];
      for (int round = 0; round < 5; ++round) {
        r[0].push_back(run8<0, false, 0>(q, grid8, 20)); r[1].push_back(run8<0, false, 2>(q, grid8, 20));
        r[2].push_back(run8<0, true, 0>(q, grid8, 20)); r[3].push_back(run8<0, true, 2>(q, grid8, 20));
        r[4].push_back(run<0>(p, grid, lds, 20)); r[5].push_back(run<0>(p, grid, lds, 20));
      }
      const char* names[6] = {"w8 mode 0", "w8 mode 2 (loads before stores)", "w8 mode 0 + mask", "w8 mode 2 + mask", "five waves", "five waves (again)"};
      for (int k = 0; k < 6; ++k) { std::sort(r[k].begin(), r[k].end()); printf("    %-34s %7.1f  (min %.1f max %.1f)\n", names[k], r[k][2], r[k][0], r[k][4]); }
    }
    R8X(0, false, 2, "full, loads before stores") R8X(0, true, 2, "full + byte mask, loads before stores")
    R8X(4, false, 2, "no stores, loads before stores") R8X(2, false, 2, "no prefetch, loads before stores")
    R8(0, "full") R8(1, "no MFMA") R8(2, "no global band prefetch") R8(4, "no stores") R8(8, "no LDS pixel reads") R8(16, "no band convert + LDS store")
    R8(2 | 4, "no global traffic") R8(2 | 4 | 16, "LDS reads + MFMA only") R8(2 | 4 | 8 | 16, "MFMA only")
    R8(1 | 8 | 16, "memory only") R8(1 | 2 | 4 | 16, "LDS pixel reads only")
  }
  return 0;
}

// One LSTM time step in ONE launch: z_t = zx_t + h_{t-1} U (fp32 MFMA), cell non-linearities, done-reset of the next
// step's state -- the body of the `for t` loop of /root/reference/dmlab/networks.py:152-171 and
// /root/reference/atari/networks.py:176-218 (_unroll_cell) around tf.keras.layers.LSTMCell.
//
// Why: per step the recurrent GEMM is tiny ([B, H] x [H, 4H], 0.5 GFLOP at B = 256, H = 512).  As split-K GEMM +
// reduce/epilogue + gate kernel it cost 3 launches per step (2 600 launches per R2D2 learner step: T = 120, online +
// target network).  Measured (tools/bench_lstm_step.py, HIP-graph replay of 100 steps, B = 256): 17.3 vs 20.1 us
// per step at H = 512, 10.6 vs 12.5 at H = 256 -- the step stays latency-bound -- with a third of the launches.  Here a workgroup owns 32 batch rows x 16 units (all four gates): 256 workgroups at
// B = 256, H = 512, one per CU.
//   * U is re-laid out ONCE per forward pass as Up[k][4*unit + gate] (seedhip_lstm_permute_u), so a tile's 64 columns
//     are 256 contiguous bytes per k row and, with the x-interleaved fragments of gemm.h, every lane ends up holding
//     the FOUR GATES of one (row, unit): the cell update needs no cross-lane traffic;
//   * the four waves split K (each stages its own A / B k-tiles in a private LDS region: no workgroup barrier in the
//     k loop), then their accumulators are summed through LDS in wave order (deterministic);
//   * epilogue: + zx, store z (the backward recomputes the gates from it), c' = sig(f) c + sig(i) tanh(g),
//     h' = sig(o) tanh(c'), h_out, and the next step's inputs keep_next * (h', c').
#include "common.h"
#include "../../include/seedhip.h"

namespace {

typedef float f32x4_t __attribute__((ext_vector_type(4)));

constexpr int BK = 32, LDA = BK + 8, kRows = 32, kUnits = 16, kCols = 4 * kUnits;     // tile: 32 rows x 64 columns
constexpr int kWaveFloats = kRows * LDA + BK * kCols;                                   // A + B k-tile of one wave

__device__ __forceinline__ float sigm(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ void wave_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// U [H, 4H] (Keras: gate-major columns i | f | g | o) -> Up [H, 4H] with column 4*u + gate.
__global__ void __launch_bounds__(256)
lstm_permute_u_kernel(const float* __restrict__ u, int H, float* __restrict__ up) {
  const long long total = (long long)H * 4 * H;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int k = (int)(i / (4 * H)), c = (int)(i - (long long)k * 4 * H);
    const int unit = c >> 2, gate = c & 3;
    up[i] = u[(long long)k * 4 * H + gate * H + unit];
  }
}

struct StepParams {
  const float* hin; const float* up; const float* zx; const float* cin; const uint8_t* done_next;
  int B, H;
  float* z; float* h_out; int ld_h; float* hin_next; float* cin_next;
};

constexpr int kWaves = 4;                                 // K is split over the waves: H / 4 per wave

__global__ void __launch_bounds__(64 * kWaves)
lstm_step_fwd_kernel(const StepParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];       // kWaves * kWaveFloats floats
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lx = lane & 15, kq = lane >> 4;
  // Workgroup -> tile: consecutive workgroup ids go round-robin to the 8 XCDs (separate L2s), so give each XCD its own
  // set of column tiles: all row tiles of a column tile then share ONE L2 copy of that 128 KB slice of U instead of
  // every XCD streaming the whole 4 MB matrix each step.
  const int nrow = (p.B + kRows - 1) / kRows, ncol = p.H / kUnits;
  int rt, ct;
  if ((ncol & 7) == 0) {
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3, per = ncol >> 3;     // j in [0, per * nrow)
    ct = xcd * per + j / nrow; rt = j - (j / nrow) * nrow;
  } else {
    ct = blockIdx.x / nrow; rt = blockIdx.x - ct * nrow;
  }
  const int m0 = rt * kRows, u0 = ct * kUnits;
  const int H = p.H, ld_u = 4 * H;
  float* As = smem + wave * kWaveFloats;                  // [32 rows][LDA]
  float* Bs = As + kRows * LDA;                           // [32 k][64 cols]

  // ---- epilogue operands of this thread's items (row, unit), requested first: their latency hides under the GEMM ----
  constexpr int kItems = kRows * kUnits / (64 * kWaves);  // 512 (row, unit) pairs over the workgroup's threads
  const int e_ul = tid & 15;
  const int e_unit = u0 + e_ul;
  float zx0[kItems], zx1[kItems], zx2[kItems], zx3[kItems], cp[kItems], keep[kItems];
#pragma unroll
  for (int it = 0; it < kItems; ++it) {
    const int e_b = m0 + (tid >> 4) + it * (4 * kWaves);
    zx0[it] = zx1[it] = zx2[it] = zx3[it] = cp[it] = 0.f; keep[it] = 1.f;
    if (e_b < p.B) {
      const long long zo = (long long)e_b * 4 * H + e_unit;
      zx0[it] = p.zx[zo]; zx1[it] = p.zx[zo + H]; zx2[it] = p.zx[zo + 2 * H]; zx3[it] = p.zx[zo + 3 * H];
      cp[it] = p.cin[(long long)e_b * H + e_unit];
      keep[it] = (p.done_next && p.done_next[e_b]) ? 0.f : 1.f;
    }
  }

  // this wave's slice of K
  const int kper = H / kWaves, k0 = wave * kper, nkt = kper / BK;
  // staging: A 32 x 32 floats = 256 float4 -> 4 per lane (row = v >> 3, kc = (v & 7) * 4); B 32 x 64 = 512 float4 -> 8
  const float* a_src[4]; int a_lds[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int v = lane + 64 * i, row = v >> 3, kc = (v & 7) * 4;
    a_lds[i] = row * LDA + kc;
    a_src[i] = (m0 + row < p.B) ? p.hin + (long long)(m0 + row) * H + k0 + kc : nullptr;
  }
  const float* b_src = p.up + (long long)k0 * ld_u + 4 * u0;     // + (k row) * ld_u + col
  float4 ra0[4], rb0[8], ra1[4], rb1[8];                  // two register stages: both k-tiles of H <= 512 fly at once
  auto load = [&](int kt, float4 (&ra)[4], float4 (&rb)[8]) {
    if (kt >= nkt) return;
#pragma unroll
    for (int i = 0; i < 4; ++i) ra[i] = a_src[i] ? *reinterpret_cast<const float4*>(a_src[i] + kt * BK) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int v = lane + 64 * i, kr = v >> 4, c4 = (v & 15) * 4;
      rb[i] = *reinterpret_cast<const float4*>(b_src + (long long)(kt * BK + kr) * ld_u + c4);
    }
  };
  f32x4_t acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const float* a_frag = As + lx * LDA + 4 * kq;
  const float* b_frag = Bs + (4 * kq) * kCols + 4 * lx;
  load(0, ra0, rb0);
  load(1, ra1, rb1);
  auto step = [&](int kt, float4 (&ra)[4], float4 (&rb)[8]) {
    wave_fence();                                         // fragment reads of the previous k-tile are issued
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<float4*>(As + a_lds[i]) = ra[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int v = lane + 64 * i;
      *reinterpret_cast<float4*>(Bs + (v >> 4) * kCols + (v & 15) * 4) = rb[i];
    }
    wave_fence();
    load(kt + 2, ra, rb);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      f32x4_t af[2], bf[4];
#pragma unroll
      for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const f32x4_t*>(a_frag + i * 16 * LDA + h * 16);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) bf[kk] = *reinterpret_cast<const f32x4_t*>(b_frag + (h * 16 + kk) * kCols);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][kk], bf[kk][j], acc[i][j], 0, 0, 0);
    }
  };
  for (int kt = 0; kt < nkt; kt += 2) {
    step(kt, ra0, rb0);
    if (kt + 1 < nkt) step(kt + 1, ra1, rb1);
  }

  // ---- sum the waves' partial tiles through LDS (wave order), then the cell update ----
  // lane holds rows 16 i + 4 kq + r, unit lx, gates j: partial layout [wave][row][unit] x float4(gates)
  __syncthreads();                                        // every wave is done with its staging region
  float4* part = reinterpret_cast<float4*>(smem);        // kWaves x 32 x 16 float4 = 64 KB <= the staging regions
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 16 * i + 4 * kq + r;
      part[(wave * kRows + row) * kUnits + lx] = make_float4(acc[i][0][r], acc[i][1][r], acc[i][2][r], acc[i][3][r]);
    }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < kItems; ++it) {
    const int e_row = (tid >> 4) + it * (4 * kWaves), e_b = m0 + e_row;
    if (e_b >= p.B) continue;
    float4 s = part[e_row * kUnits + e_ul];
#pragma unroll
    for (int w = 1; w < kWaves; ++w) {
      const float4 t = part[(w * kRows + e_row) * kUnits + e_ul];
      s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
    }
    const long long zo = (long long)e_b * 4 * H + e_unit;
    const float zi = zx0[it] + s.x, zf = zx1[it] + s.y, zg = zx2[it] + s.z, zoo = zx3[it] + s.w;
    p.z[zo] = zi; p.z[zo + H] = zf; p.z[zo + 2 * H] = zg; p.z[zo + 3 * H] = zoo;
    const float ig = sigm(zi), fg = sigm(zf), gg = tanhf(zg), og = sigm(zoo);
    const float c = fg * cp[it] + ig * gg;
    const float hh = og * tanhf(c);
    p.h_out[(long long)e_b * p.ld_h + e_unit] = hh;
    p.hin_next[(long long)e_b * H + e_unit] = hh * keep[it];
    p.cin_next[(long long)e_b * H + e_unit] = c * keep[it];
  }
}

}  // namespace

extern "C" int seedhip_lstm_permute_u(const float* u, int H, float* up, void* stream) {
  SEEDHIP_REQUIRE(u && up && H >= 4 && H % 4 == 0, "lstm_permute_u: need H %% 4 == 0");
  long long blocks = ((long long)H * 4 * H + 255) / 256; if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(lstm_permute_u_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, u, H, up);
  return seedhip::check_launch("lstm_permute_u_kernel");
}

extern "C" int seedhip_lstm_step_supported(int B, int H) { return B >= 1 && H >= 32 * kWaves && H % (32 * kWaves) == 0; }

extern "C" int seedhip_lstm_step_fwd(const float* hin, const float* up, const float* zx, const float* cin,
                                     const uint8_t* done_next, int B, int H, float* z, float* h_out, int ld_h,
                                     float* hin_next, float* cin_next, void* stream) {
  SEEDHIP_REQUIRE(hin && up && zx && cin && z && h_out && hin_next && cin_next, "lstm_step_fwd: null pointer");
  SEEDHIP_REQUIRE(seedhip_lstm_step_supported(B, H), "lstm_step_fwd: need H %% 128 == 0 (H = %d)", H);
  SEEDHIP_REQUIRE(ld_h >= H, "lstm_step_fwd: ld_h < H");
  SEEDHIP_REQUIRE(((((uintptr_t)hin) | ((uintptr_t)up)) & 15) == 0, "lstm_step_fwd: hin / up must be 16-byte aligned");
  StepParams p;
  p.hin = hin; p.up = up; p.zx = zx; p.cin = cin; p.done_next = done_next; p.B = B; p.H = H;
  p.z = z; p.h_out = h_out; p.ld_h = ld_h; p.hin_next = hin_next; p.cin_next = cin_next;
  const size_t lds = (size_t)kWaves * kWaveFloats * sizeof(float);                     // 104 KB: one workgroup per CU
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)lstm_step_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  hipLaunchKernelGGL(lstm_step_fwd_kernel, dim3(((B + kRows - 1) / kRows) * (H / kUnits)), dim3(64 * kWaves), lds,
                     (hipStream_t)stream, p);
  return seedhip::check_launch("lstm_step_fwd_kernel");
}

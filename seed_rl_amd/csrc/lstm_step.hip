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

// ---- the whole unroll in ONE launch ---------------------------------------------------------------------------------
// Same tiling and the same arithmetic (bit-identical results) as lstm_step_fwd_kernel, but the time loop runs inside
// the kernel: every workgroup is resident (one per CU), keeps ITS 16 units' slice of U in LDS for the whole sequence
// (H x 64 floats: 128 KB at H = 512) and its cell state in registers, and the steps are separated by a grid barrier
// (monotonic counter in global memory: agent-scope release of h_t, one lane polls with agent-scope acquire loads).
// Per step a workgroup then only reads its 32 rows of h_{t-1} and zx_t; no launch, no U traffic.  Every spin is
// bounded: a workgroup that waits too long (the grid was not co-resident) raises the abort flag and all leave.
// Arms an exchange buffer with the sentinel and clears the 8-byte sync word, as ONE ordinary kernel (a captured
// hipMemsetAsync node misbehaved under HIP-graph replay on ROCm 7.2: see DESIGN.md).
// hs (may be null): the XCD handshake words of the sequence kernels, `nrow` groups of `ncol` words `hs_stride` apart.
__global__ void __launch_bounds__(256)
seq_arm_kernel(uint4* __restrict__ buf, long long n16, unsigned* __restrict__ sync_ws, unsigned* __restrict__ hs = nullptr,
               int nrow = 0, int ncol = 0, long long hs_stride = 0) {
  const uint4 s = make_uint4(0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long long)gridDim.x * 256) buf[i] = s;
  if (blockIdx.x == 0 && threadIdx.x < 2) sync_ws[threadIdx.x] = 0u;
  if (hs && blockIdx.x == 0)
    for (int i = threadIdx.x; i < nrow * ncol; i += 256) hs[(long long)(i / ncol) * hs_stride + (i % ncol)] = 0xffffffffu;
}

struct SeqParams {
  const float* up; const float* zx; const uint8_t* done; int T1, B, H;
  float* z; float* h_out; int ld_h; float* hin; float* cin;          // hin / cin [T1 + 1, B, H]; slot 0 = initial state
  int* abort_flag; int fault;
  int* sticky;                                               // set together with abort_flag, never cleared by a launch (may be null)
  int xcd_local;                                             // try the same-XCD exchange (plain producer stores)
};

constexpr unsigned kMaxSpins = 200000;                    // x (one agent-scope round trip): a few hundred ms
constexpr unsigned kSentinel = 0xffffffffu;               // "not written yet" (a NaN no arithmetic produces)

__global__ void __launch_bounds__(64 * kWaves)
lstm_seq_fwd_kernel(const SeqParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];       // U slabs [H][64] | staging / partial sums (24 KB)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lx = lane & 15, kq = lane >> 4;
  const int nrow = (p.B + kRows - 1) / kRows, ncol = p.H / kUnits;
  // Row tile = blockIdx % nrow: workgroup i runs on XCD i % 8 (tools/probes/xcc_probe.hip), so with 8 row tiles
  // (B = 256) the ncol workgroups that exchange h among themselves -- those of ONE row tile -- share an XCD and its
  // L2.  (U lives in LDS for the whole sequence: which XCD reads it once does not matter.)
  const int ct = blockIdx.x / nrow, rt = blockIdx.x - ct * nrow;
  const int m0 = rt * kRows, u0 = ct * kUnits;
  const int H = p.H, ld_u = 4 * H, kper = H / kWaves, k0 = wave * kper, nkt = kper / BK;    // nkt <= 4 (H <= 512)
  const long long BH = (long long)p.B * H;
  float* Us = smem + wave * kper * kCols;                 // this wave's K slice of the tile's U columns: [kper][64]
  float* stage = smem + H * kCols;
  float* As = stage + wave * kRows * LDA;                 // [32 rows][LDA], wave-private
  float4* part = reinterpret_cast<float4*>(stage);       // [owner 4][source 3][2][64 lanes] float4, overlays the A stages

  {
    const float* b_src = p.up + (long long)k0 * ld_u + 4 * u0;
#pragma unroll 8
    for (int v = lane; v < kper * 16; v += 64) {
      const int kr = v >> 4, c4 = (v & 15) * 4;
      *reinterpret_cast<float4*>(Us + kr * kCols + c4) = *reinterpret_cast<const float4*>(b_src + (long long)kr * ld_u + c4);
    }
  }
  // this thread's two (row, unit) items: the quarter of the tile wave `wave` reduces -- rows 16 i + 4 kq + r0 + {0, 1}
  const int own_i = wave >> 1, own_r0 = 2 * (wave & 1);
  const int e_unit = u0 + lx;
  int e_b[2]; float cst[2];
#pragma unroll
  for (int rr = 0; rr < 2; ++rr) {
    e_b[rr] = m0 + 16 * own_i + 4 * kq + own_r0 + rr;
    cst[rr] = e_b[rr] < p.B ? p.cin[(long long)e_b[rr] * H + e_unit] : 0.f;
  }
  const float* a_src[4]; int a_lds[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int v = lane + 64 * i, row = v >> 3, kc = (v & 7) * 4;
    a_lds[i] = row * LDA + kc;
    a_src[i] = (m0 + row < p.B) ? p.hin + (long long)(m0 + row) * H + k0 + kc : nullptr;
  }
  const float* a_frag = As + lx * LDA + 4 * kq;
  const float* b_frag = Us + (4 * kq) * kCols + 4 * lx;

  bool dead = false;                                      // a wait timed out: stop polling, finish with garbage
  // ---- same-XCD exchange?  Every workgroup publishes the XCD it runs on (agent-scope store into a word of z's LAST
  //      step, which nobody touches before step T1 - 1; armed with the sentinel by seq_arm_kernel) and reads the words
  //      of its row tile's ncol workgroups.  If they all agree, h can travel through the shared L2: PLAIN producer
  //      stores keep the line there (an sc1 store would drop it: MI355X_MICROARCH.md), the consumers' sc1 loads bypass
  //      only their L1.  Placement is verified, never assumed: any disagreement (or a timeout) keeps agent-scope stores.
  bool xcd_local = false;
  if (p.xcd_local) {
    unsigned my_xcd;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(my_xcd));
    my_xcd &= 15u;
    unsigned* hs = reinterpret_cast<unsigned*>(p.z + ((long long)(p.T1 - 1) * p.B + m0) * 4 * H);
    int* flag = reinterpret_cast<int*>(stage);
    if (tid == 0) __hip_atomic_store(hs + ct, my_xcd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (wave == 0) {
      unsigned v = my_xcd;
      bool late = false;
      if (lane < ncol) {
        unsigned spins = 0;
        while ((v = __hip_atomic_load(hs + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == kSentinel) {
          if (++spins > kMaxSpins) { late = true; break; }
          __builtin_amdgcn_s_sleep(1);
        }
      }
      const bool same = __all(!late && v == my_xcd);
      if (lane == 0) *flag = same ? 1 : 0;
    }
    __syncthreads();
    xcd_local = *flag != 0;
    __syncthreads();                                      // the flag word is the A stage of wave 0 again
  }
  for (int t = 0; t < p.T1; ++t) {
    // epilogue operands first: their latency hides under the wait / the GEMM
    float zxv[2][4], keep[2];
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      keep[rr] = 1.f;
#pragma unroll
      for (int g = 0; g < 4; ++g) zxv[rr][g] = 0.f;
      if (e_b[rr] < p.B) {
        const float* zp = p.zx + ((long long)t * p.B + e_b[rr]) * 4 * H + e_unit;
#pragma unroll
        for (int g = 0; g < 4; ++g) zxv[rr][g] = zp[g * H];
        if (t + 1 < p.T1 && p.done[(long long)(t + 1) * p.B + e_b[rr]]) keep[rr] = 0.f;
      }
    }
    // h_{t-1}: this wave's [32 rows] x [its K quarter], straight into registers with agent-scope (sc1) loads.  The
    // data is its own ready flag: slot t of hin was filled with kSentinel before the launch and every producer
    // stores each word exactly once, so the wave re-reads until no word is the sentinel.  No counter, no fence, and
    // no workgroup barrier here -- each wave waits only for the 8 workgroups that produce its columns.
    f32x4_t ra[4][4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int i = 0; i < 4; ++i) ra[kt][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    for (unsigned spins = 0;;) {
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
        if (kt < nkt)
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (a_src[i]) {
              const float* q = a_src[i] + t * BH + kt * BK;
              asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(ra[kt][i]) : "v"(q) : "memory");
            }
      asm volatile("s_waitcnt vmcnt(0)"
                   : "+v"(ra[0][0]), "+v"(ra[0][1]), "+v"(ra[0][2]), "+v"(ra[0][3]), "+v"(ra[1][0]), "+v"(ra[1][1]),
                     "+v"(ra[1][2]), "+v"(ra[1][3]), "+v"(ra[2][0]), "+v"(ra[2][1]), "+v"(ra[2][2]), "+v"(ra[2][3]),
                     "+v"(ra[3][0]), "+v"(ra[3][1]), "+v"(ra[3][2]), "+v"(ra[3][3])
                   :: "memory");
      if (t == 0 || dead) break;                          // slot 0 was written by the previous kernel
      bool stale = false;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int e = 0; e < 4; ++e) stale |= __float_as_uint(ra[kt][i][e]) == kSentinel;
      if (!__any(stale)) break;
      ++spins;
      if (spins > kMaxSpins || ((spins & 31) == 0 && __hip_atomic_load(p.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
        if (lane == 0) {
          __hip_atomic_store(p.abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (p.sticky) __hip_atomic_store(p.sticky, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        dead = true;
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
    f32x4_t acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      if (kt >= nkt) break;
      wave_fence();
#pragma unroll
      for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4_t*>(As + a_lds[i]) = ra[kt][i];
      wave_fence();
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        f32x4_t af[2], bf[4];
#pragma unroll
        for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const f32x4_t*>(a_frag + i * 16 * LDA + h * 16);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) bf[kk] = *reinterpret_cast<const f32x4_t*>(b_frag + (kt * BK + h * 16 + kk) * kCols);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][kk], bf[kk][j], acc[i][j], 0, 0, 0);
      }
    }
    // ---- cross-wave sum: wave q reduces quarter q = (i, r0); the others hand it their partials through LDS ----
    __syncthreads();                                      // every wave is done with its A stage (part overlays them)
    float own[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int i = q >> 1, r0 = 2 * (q & 1);
      const bool mine = q == wave;
#pragma unroll
      for (int rr = 0; rr < 2; ++rr) {
#pragma unroll
        for (int g = 0; g < 4; ++g) own[rr][g] = mine ? acc[i][g][r0 + rr] : own[rr][g];
        if (!mine)
          part[((q * 3 + (wave < q ? wave : wave - 1)) * 2 + rr) * 64 + lane] =
              make_float4(acc[i][0][r0 + rr], acc[i][1][r0 + rr], acc[i][2][r0 + rr], acc[i][3][r0 + rr]);
      }
    }
    __syncthreads();
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int w = 0; w < kWaves; ++w) {                  // wave order, as lstm_step_fwd_kernel sums
        const float4 v = (w == wave) ? make_float4(own[rr][0], own[rr][1], own[rr][2], own[rr][3]) : part[((wave * 3 + (w < wave ? w : w - 1)) * 2 + rr) * 64 + lane];
        if (w == 0) s = v; else { s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
      }
      if (e_b[rr] >= p.B) continue;
      const long long row = (long long)t * p.B + e_b[rr];
      float* zo = p.z + row * 4 * H + e_unit;
      const float zi = zxv[rr][0] + s.x, zf = zxv[rr][1] + s.y, zg = zxv[rr][2] + s.z, zoo = zxv[rr][3] + s.w;
      zo[0] = zi; zo[H] = zf; zo[2 * H] = zg; zo[3 * H] = zoo;
      const float ig = sigm(zi), fg = sigm(zf), gg = tanhf(zg), og = sigm(zoo);
      const float c = fg * cst[rr] + ig * gg;
      const float hh = og * tanhf(c);
      p.h_out[row * p.ld_h + e_unit] = hh;
      float hn = hh * keep[rr];
      if (__float_as_uint(hn) == kSentinel) hn = __uint_as_float(0x7fc00000u);       // a NaN that happens to be the sentinel
      if (!((p.fault & 1) && blockIdx.x == 0 && t == 1)) { // test hook: a producer that never delivers
        float* hp = p.hin + (t + 1) * BH + (long long)e_b[rr] * H + e_unit;
        if (xcd_local) *hp = hn;                           // stays in the XCD's L2, where all its readers are
        else __hip_atomic_store(hp, hn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      cst[rr] = c * keep[rr];
      p.cin[(t + 1) * BH + (long long)e_b[rr] * H + e_unit] = cst[rr];
    }
    __syncthreads();                                      // partial-sum reads are done before the next A stage
  }
}

// ---- the whole BACKWARD recurrence in one launch --------------------------------------------------------------------
// Per step (t = T1-1 .. 0): dz_t = cell'(z_t, c_{t-1}; dh_out_t + keep * dh_rec, keep * dc_rec), dh_rec = dz_t U^T --
// lstm_gates_bwd_kernel + one [B, 4H] x [4H, H] GEMM (+ its split-K reduce) per step as separate launches.  Here the
// workgroup that owns (32 rows, 16 units) in the forward kernel owns them again: it keeps the SAME 64 columns of Up in
// LDS, computes dz for its tile (cell-state gradient in registers), and multiplies that [32 x 64] slice of dz with its
// [64 x H] slice of U^T: a PARTIAL dh_rec for every unit, K-split over the column tiles.  The partials are exchanged
// through a three-slot ring in global memory -- chunk (producer, consumer) = 16 units x 32 rows, agent-scope stores and
// loads, the data is its own ready flag (sentinel as in the forward kernel; the consumer re-arms a chunk right after
// reading it) -- and summed by the consumer in producer order: deterministic, 64 KB in + 64 KB out per workgroup and
// step instead of a full re-read of dz.
struct SeqBwdParams {
  const float* up; const float* z; const float* cin; const float* dh_out; int ld_dh; const uint8_t* done;
  int T1, B, H;
  float* dz; float* ring; int* abort_flag; int fault;
  int* sticky;                                               // as SeqParams::sticky
};

constexpr int LDT = kCols + 4;                            // dz tile row stride (bank spread for the b128 fragment reads)
constexpr int LDR = kRows + 4;                            // half-sum row stride

template <int NT>                                         // output column tiles per wave: H = 64 NT
__global__ void __launch_bounds__(64 * kWaves)
lstm_seq_bwd_kernel(const SeqBwdParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];       // Up slab [H][64] (16-byte chunks XOR-swizzled) | dz tile | half sums
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lx = lane & 15, kq = lane >> 4;
  constexpr int H = 64 * NT, ncol = H / kUnits, NP = ncol / 2;
  const int nrow = (p.B + kRows - 1) / kRows;
  int rt, ct;
  if ((ncol & 7) == 0) {
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3, per = ncol >> 3;
    ct = xcd * per + j / nrow; rt = j - (j / nrow) * nrow;
  } else {
    ct = blockIdx.x / nrow; rt = blockIdx.x - ct * nrow;
  }
  const int m0 = rt * kRows, u0 = ct * kUnits;
  float* slab = smem;
  float* At = smem + H * kCols;                           // [32][LDT]: row, then 4 * unit + gate
  float* red = At + kRows * LDT;                          // [2 halves][16 units][LDR]: sums of the producers' partials
  for (int v = tid; v < H * 16; v += 64 * kWaves) {
    const int j = v >> 4, c = v & 15;
    *reinterpret_cast<float4*>(slab + j * kCols + ((c ^ (j & 15)) << 2)) =
        *reinterpret_cast<const float4*>(p.up + (long long)j * 4 * H + 4 * u0 + 4 * c);
  }
  // gate items of this thread: (row, unit) x 2
  const int ul = tid & 15, unit = u0 + ul;
  int row_l[2], e_b[2]; float dcst[2];
#pragma unroll
  for (int it = 0; it < 2; ++it) { row_l[it] = (tid >> 4) + 16 * it; e_b[it] = m0 + row_l[it]; dcst[it] = 0.f; }
  // partial-sum reader role: (half of the producers, row, unit quad)
  const int half = wave >> 1, rd_unit = (tid & 127) >> 3, rd_rq = tid & 7;        // chunk layout: [unit][row]
  const long long chunk = kRows * kUnits;                 // floats per (producer, consumer) chunk
  const long long slot_floats = (long long)nrow * ncol * ncol * chunk;
  bool dead = false;
  const f32x4_t sentinel4 = {__uint_as_float(kSentinel), __uint_as_float(kSentinel), __uint_as_float(kSentinel), __uint_as_float(kSentinel)};
  __syncthreads();

  for (int t = p.T1 - 1; t >= 0; --t) {
    float zv[2][4], cp[2], dho[2], keep[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      zv[it][0] = zv[it][1] = zv[it][2] = zv[it][3] = cp[it] = dho[it] = 0.f; keep[it] = 1.f;
      if (e_b[it] < p.B) {
        const long long row = (long long)t * p.B + e_b[it];
#pragma unroll
        for (int g = 0; g < 4; ++g) zv[it][g] = p.z[row * 4 * H + g * H + unit];
        cp[it] = p.cin[row * H + unit];
        dho[it] = p.dh_out[row * p.ld_dh + unit];
        if (t + 1 < p.T1 && p.done[(long long)(t + 1) * p.B + e_b[it]]) keep[it] = 0.f;
      }
    }
    float rec[2] = {0.f, 0.f};
    if (t + 1 < p.T1) {                                   // dh_rec of step t + 1: sum the producers' partials
      float* src = p.ring + ((t + 1) % 3) * slot_floats + (((long long)rt * ncol + half * NP) * ncol + ct) * chunk +
                   rd_unit * kRows + rd_rq * 4;
      f32x4_t v[NP];
      for (unsigned spins = 0;;) {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
          const float* q = src + (long long)i * ncol * chunk;
          asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v[i]) : "v"(q) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        bool stale = false;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
          asm volatile("" : "+v"(v[i]));                  // the loads above have landed: keep uses below the wait
#pragma unroll
          for (int e = 0; e < 4; ++e) stale |= __float_as_uint(v[i][e]) == kSentinel;
        }
        if (dead || !__any(stale) || (p.fault & 8)) break;
        ++spins;
        if (spins > kMaxSpins || ((spins & 31) == 0 && __hip_atomic_load(p.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
          if (lane == 0) {
            __hip_atomic_store(p.abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (p.sticky) __hip_atomic_store(p.sticky, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          dead = true;
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
#pragma unroll
      for (int i = 0; i < NP; ++i) {                      // re-arm the chunks: this slot is written again at step t - 2
        float* q = src + (long long)i * ncol * chunk;
        asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(q), "v"(sentinel4) : "memory");
      }
      f32x4_t sum = v[0];
#pragma unroll
      for (int i = 1; i < NP; ++i) sum += v[i];
      *reinterpret_cast<f32x4_t*>(red + (half * kUnits + rd_unit) * LDR + rd_rq * 4) = sum;
      __syncthreads();
#pragma unroll
      for (int it = 0; it < 2; ++it) rec[it] = red[ul * LDR + row_l[it]] + red[(kUnits + ul) * LDR + row_l[it]];
    }
    // cell backward (the arithmetic of lstm_gates_bwd_kernel), dz to global (gate-major, for the weight gradients)
    // and to the LDS tile (unit-major: the k order of the Up slab)
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const float dh = dho[it] + keep[it] * rec[it], dc = keep[it] * dcst[it];
      const float ig = sigm(zv[it][0]), fg = sigm(zv[it][1]), gg = tanhf(zv[it][2]), og = sigm(zv[it][3]);
      const float c = fg * cp[it] + ig * gg; const float tc = tanhf(c);
      const float d_o = dh * tc; const float d_c = dc + dh * og * (1.0f - tc * tc);
      float4 d;
      d.x = d_c * gg * ig * (1.0f - ig); d.y = d_c * cp[it] * fg * (1.0f - fg);
      d.z = d_c * ig * (1.0f - gg * gg); d.w = d_o * og * (1.0f - og);
      dcst[it] = d_c * fg;
      if (e_b[it] < p.B) {
        float* dr = p.dz + ((long long)t * p.B + e_b[it]) * 4 * H + unit;
        dr[0] = d.x; dr[H] = d.y; dr[2 * H] = d.z; dr[3 * H] = d.w;
      } else {
        d = make_float4(0.f, 0.f, 0.f, 0.f);
      }
      *reinterpret_cast<float4*>(At + row_l[it] * LDT + ul * 4) = d;
    }
    // Ring depth 3: the chunks re-armed at step t (slot (t + 1) % 3) are next written by their producers at step
    // t - 2, i.e. after those have consumed what this workgroup publishes at the end of step t - 1 -- and every wave
    // here passes the s_waitcnt vmcnt(0) of step t - 1's wait loop (its re-arming stores are acknowledged) and the
    // barrier after it before any wave publishes.  With two slots the acknowledgement would have to be waited for
    // right here, on the critical path.
    __syncthreads();
    if (t > 0) {                                          // partial dh_rec = dz tile [32 x 64] x Up slab^T [64 x H]
      f32x4_t acc[2][NT];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int h = 0; h < 4; ++h) {
        if (p.fault & 2) break;
        f32x4_t af[2], bf[NT];
#pragma unroll
        for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const f32x4_t*>(At + (16 * i + lx) * LDT + (4 * h + kq) * 4);
#pragma unroll
        for (int j = 0; j < NT; ++j)
          bf[j] = *reinterpret_cast<const f32x4_t*>(slab + ((wave * NT + j) * 16 + lx) * kCols + (((4 * h + kq) ^ lx) << 2));
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][kk], bf[j][kk], acc[i][j], 0, 0, 0);
      }
      if (!((p.fault & 1) && blockIdx.x == 0 && t == p.T1 - 2) && !(p.fault & 4)) {                 // test hook: a producer that never delivers
        float* dst = p.ring + (t % 3) * slot_floats + (((long long)rt * ncol + ct) * ncol + wave * NT) * chunk +
                     lx * kRows + 4 * kq;                 // lane: unit lx, rows 16 i + 4 kq + (0..3) -- 16-byte stores
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            f32x4_t v = acc[i][j];
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (__float_as_uint(v[r]) == kSentinel) v[r] = __uint_as_float(0x7fc00000u);
            float* q = dst + j * chunk + 16 * i;
            // s_nop: a >64-bit store reads its data registers a cycle after issue, and the hazard recogniser cannot
            // see into inline asm (without it the next tile's v_mov landed in v[0] first)
            asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(q), "v"(v) : "memory");
          }
      }
    }
    __syncthreads();                                      // dz tile / half sums are reused by the next step
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
  // once per process, thread-safe (C++11 static initialisation): the ABI is called from the inference pool and the
  // training thread concurrently
  static const hipError_t attr_rc =
      hipFuncSetAttribute((const void*)lstm_step_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  (void)attr_rc;
  hipLaunchKernelGGL(lstm_step_fwd_kernel, dim3(((B + kRows - 1) / kRows) * (H / kUnits)), dim3(64 * kWaves), lds,
                     (hipStream_t)stream, p);
  return seedhip::check_launch("lstm_step_fwd_kernel");
}


namespace {
int seq_resident_limit(size_t lds) {                       // workgroups that are surely co-resident on this device
  int dev = 0, cus = 0, lds_cu = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
  if (hipDeviceGetAttribute(&lds_cu, hipDeviceAttributeMaxSharedMemoryPerMultiprocessor, dev) != hipSuccess || lds_cu <= 0)
    lds_cu = 160 * 1024;
  long long per = (long long)lds_cu / (long long)(lds + 64);
  if (per > 2) per = 2;                                    // 256 threads x <= 128 VGPRs: registers allow at least this
  return (int)(cus * per);
}
size_t seq_lds_bytes(int H) { return (size_t)H * kCols * sizeof(float) + 24 * 1024; }
}  // namespace

extern "C" int seedhip_lstm_seq_supported(int T1, int B, int H) {
  if (!(T1 >= 2 && seedhip_lstm_step_supported(B, H) && H <= 512)) return 0;
  const int grid = ((B + kRows - 1) / kRows) * (H / kUnits);
  return grid <= seq_resident_limit(seq_lds_bytes(H));
}

extern "C" int seedhip_lstm_seq_fwd(const float* up, const float* zx, const uint8_t* done, int T1, int B, int H,
                                    float* z, float* h_out, int ld_h, float* hin, float* cin, void* sync_ws,
                                    void* stream) {
  return seedhip_lstm_seq_fwd_sticky(up, zx, done, T1, B, H, z, h_out, ld_h, hin, cin, sync_ws, nullptr, stream);
}

extern "C" int seedhip_lstm_seq_fwd_sticky(const float* up, const float* zx, const uint8_t* done, int T1, int B, int H,
                                           float* z, float* h_out, int ld_h, float* hin, float* cin, void* sync_ws,
                                           int* sticky_abort, void* stream) {
  SEEDHIP_REQUIRE(up && zx && done && z && h_out && hin && cin && sync_ws, "lstm_seq_fwd: null pointer");
  SEEDHIP_REQUIRE(seedhip_lstm_seq_supported(T1, B, H),
                  "lstm_seq_fwd: unsupported (T1 = %d, B = %d, H = %d): need T1 >= 2, H %% 128 == 0, H <= 512 and a co-resident grid",
                  T1, B, H);
  SEEDHIP_REQUIRE(ld_h >= H, "lstm_seq_fwd: ld_h < H");
  SEEDHIP_REQUIRE(((((uintptr_t)hin) | ((uintptr_t)up)) & 15) == 0, "lstm_seq_fwd: hin / up must be 16-byte aligned");
  SeqParams p;
  p.up = up; p.zx = zx; p.done = done; p.T1 = T1; p.B = B; p.H = H; p.z = z; p.h_out = h_out; p.ld_h = ld_h;
  p.hin = hin; p.cin = cin; p.abort_flag = (int*)sync_ws + 1; p.sticky = sticky_abort;
  { const char* e = getenv("SEEDHIP_LSTM_SEQ_FAULT"); p.fault = e ? atoi(e) : 0; }   // tests: exercise the bounded wait (read per call on purpose: tests flip it)
  p.xcd_local = 1;                                      // (plain stores where a row tile's workgroups share an XCD: verified at run time)
  const size_t lds = seq_lds_bytes(H);
  static const hipError_t attr_rc = hipFuncSetAttribute(
      (const void*)lstm_seq_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)seq_lds_bytes(512));
  (void)attr_rc;
  {
    const long long n16 = (long long)T1 * B * H / 4;       // H % 128 == 0: whole 16-byte words
    long long blocks = (n16 + 255) / 256; if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(seq_arm_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<uint4*>(hin + (size_t)B * H), n16, (unsigned*)sync_ws,
                       reinterpret_cast<unsigned*>(z + (size_t)(T1 - 1) * B * 4 * H), (B + kRows - 1) / kRows, H / kUnits,
                       (long long)kRows * 4 * H);
    if (int rc = seedhip::check_launch("seq_arm_kernel")) return rc;
  }
  hipLaunchKernelGGL(lstm_seq_fwd_kernel, dim3(((B + kRows - 1) / kRows) * (H / kUnits)), dim3(64 * kWaves), lds,
                     (hipStream_t)stream, p);
  return seedhip::check_launch("lstm_seq_fwd_kernel");
}


namespace {
size_t seq_bwd_lds_bytes(int H) { return ((size_t)H * kCols + kRows * LDT + 2 * kUnits * LDR) * sizeof(float); }
size_t seq_bwd_ring_bytes(int B, int H) {
  const size_t nrow = (B + kRows - 1) / kRows, ncol = H / kUnits;
  return 3 * nrow * ncol * ncol * kRows * kUnits * sizeof(float);
}
template <int NT>
int launch_seq_bwd(const SeqBwdParams& p, int grid, hipStream_t stream) {
  const size_t lds = seq_bwd_lds_bytes(64 * NT);
  static const hipError_t attr_rc =        // one static per NT instantiation
      hipFuncSetAttribute((const void*)lstm_seq_bwd_kernel<NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  (void)attr_rc;
  hipLaunchKernelGGL(lstm_seq_bwd_kernel<NT>, dim3(grid), dim3(64 * kWaves), lds, stream, p);
  return seedhip::check_launch("lstm_seq_bwd_kernel");
}
}  // namespace

extern "C" size_t seedhip_lstm_seq_bwd_workspace_bytes(int B, int H) {
  return (B >= 1 && H >= 128 && H % 128 == 0) ? seq_bwd_ring_bytes(B, H) : 0;
}

extern "C" int seedhip_lstm_seq_bwd(const float* up, const float* z, const float* cin, const float* dh_out, int ld_dh,
                                    const uint8_t* done, int T1, int B, int H, float* dz, void* ring_ws, void* sync_ws,
                                    void* stream) {
  return seedhip_lstm_seq_bwd_sticky(up, z, cin, dh_out, ld_dh, done, T1, B, H, dz, ring_ws, sync_ws, nullptr, stream);
}

extern "C" int seedhip_lstm_seq_bwd_sticky(const float* up, const float* z, const float* cin, const float* dh_out, int ld_dh,
                                           const uint8_t* done, int T1, int B, int H, float* dz, void* ring_ws,
                                           void* sync_ws, int* sticky_abort, void* stream) {
  SEEDHIP_REQUIRE(up && z && cin && dh_out && done && dz && ring_ws && sync_ws, "lstm_seq_bwd: null pointer");
  SEEDHIP_REQUIRE(seedhip_lstm_seq_supported(T1, B, H),
                  "lstm_seq_bwd: unsupported (T1 = %d, B = %d, H = %d): need T1 >= 2, H %% 128 == 0, H <= 512 and a co-resident grid",
                  T1, B, H);
  SEEDHIP_REQUIRE(ld_dh >= H, "lstm_seq_bwd: ld_dh < H");
  SEEDHIP_REQUIRE((((uintptr_t)up) | ((uintptr_t)ring_ws)) % 16 == 0, "lstm_seq_bwd: up / ring_ws must be 16-byte aligned");
  SeqBwdParams p;
  p.up = up; p.z = z; p.cin = cin; p.dh_out = dh_out; p.ld_dh = ld_dh; p.done = done; p.T1 = T1; p.B = B; p.H = H;
  p.dz = dz; p.ring = (float*)ring_ws; p.abort_flag = (int*)sync_ws + 1; p.sticky = sticky_abort;
  // bit 0: tests, exercise the bounded wait; bits 1-3: timing attribution only (tools/bench_lstm_step.py SEQ_DBG):
  // 2 = no MFMA, 4 = no partial stores, 8 = do not wait for the partials -- results are garbage with any of them
  { const char* e = getenv("SEEDHIP_LSTM_SEQ_FAULT"); p.fault = e ? atoi(e) : 0; }
  {
    const long long n16 = (long long)(seq_bwd_ring_bytes(B, H) / 16);
    long long blocks = (n16 + 255) / 256; if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(seq_arm_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<uint4*>(ring_ws), n16, (unsigned*)sync_ws);
    if (int rc = seedhip::check_launch("seq_arm_kernel")) return rc;
  }
  const int grid = ((B + kRows - 1) / kRows) * (H / kUnits);
  switch (H / 64) {
    case 2: return launch_seq_bwd<2>(p, grid, (hipStream_t)stream);
    case 4: return launch_seq_bwd<4>(p, grid, (hipStream_t)stream);
    case 6: return launch_seq_bwd<6>(p, grid, (hipStream_t)stream);
    default: return launch_seq_bwd<8>(p, grid, (hipStream_t)stream);
  }
}

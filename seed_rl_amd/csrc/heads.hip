// Policy + baseline heads FORWARD as one skinny GEMM.
//
// Replaces, for the packed head matrix W [feat, ldh] = [policy_logits | baseline | pad] (ldh = round4(A + 1) <= 32):
//   y = x W + b                      dmlab/networks.py:116-124 (Dense policy_logits, Dense baseline)
// The general GEMM core (gemm.h) served it through its cost model: a 64x64 tile is three quarters empty at N = 20, so
// it split K over two slices and added an epilogue launch (18 us per cfg2 step, 13 + 5 us per 1024-row inference
// step).  Here N is the whole problem: a wave owns 16 data rows x all ldh columns, W (<= 64 KB) sits in LDS for the
// life of the workgroup, the data rows come straight from global memory into MFMA operands (v_mfma_f32_16x16x4_f32;
// 16-byte loads through the k-permutation: lane (i, kq) of a 16-k block holds k = 4 kq + s for the block's MFMA s):
// one launch, 15 us at 10 752 rows, ~5 us at 1 024.
// (The BACKWARD of the heads stays on gemm.h: a one-pass kernel for dx / dW / db was built the same way in round 3 and
// measured 89 us against the core's 27 -- its 4-byte operand loads serialise four memory round trips per 16-row tile;
// beating the core there needs LDS-staged, double-buffered x tiles, for at most ~15 us of a 1.28 ms step.)
#include "common.h"
#include "../../include/seedhip.h"

namespace {

typedef float f32x4_t __attribute__((ext_vector_type(4)));
constexpr int kWaves = 4, kThreads = 64 * kWaves, kMaxFeat = 512, kMaxN = 32;

struct FwdParams {
  const float* x; int ldx; const float* w; const float* bias; long long rows; int feat, ldh; float* y;
};

// y[rows, ldh] = x[rows, feat] W[feat, ldh] + b.  LDS: Wt[32][feat + 4] (transposed, zero rows for n >= ldh).
__global__ void __launch_bounds__(kThreads)
heads_fwd_kernel(const FwdParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int F = p.feat, LDW = F + 4;
  float* Wt = smem;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, kq = lane >> 4;
  // W [feat][ldh] is read the way it lies in memory (coalesced) and scattered into the transposed LDS tile; the rows
  // of the pad columns n >= ldh are zeroed
  for (int idx = tid; idx < F * p.ldh; idx += kThreads) {
    const int k = idx / p.ldh, n = idx - k * p.ldh;
    Wt[n * LDW + k] = p.w[idx];
  }
  for (int idx = tid; idx < (kMaxN - p.ldh) * F; idx += kThreads) {
    const int n = p.ldh + idx / F, k = idx % F;
    Wt[n * LDW + k] = 0.f;
  }
  __syncthreads();
  const int ntiles = p.ldh > 16 ? 2 : 1;
  float b0 = 0.f, b1 = 0.f;
  if (p.bias) { b0 = j < p.ldh ? p.bias[j] : 0.f; b1 = 16 + j < p.ldh ? p.bias[16 + j] : 0.f; }
  const long long tiles = (p.rows + 15) / 16;
  for (long long tile = (long long)blockIdx.x * kWaves + wave; tile < tiles; tile += (long long)gridDim.x * kWaves) {
    const long long row0 = tile * 16;
    long long r = row0 + j;                                  // A operand: data row j of the tile (clamped at the end)
    if (r >= p.rows) r = p.rows - 1;
    const float* xr = p.x + r * p.ldx + 4 * kq;
    f32x4_t acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    const float* w0 = Wt + j * LDW + 4 * kq;
    const float* w1 = Wt + (16 + j) * LDW + 4 * kq;
#pragma unroll 4
    for (int blk = 0; blk < F / 16; ++blk) {
      const f32x4_t a = *reinterpret_cast<const f32x4_t*>(xr + 16 * blk);
      const f32x4_t bw0 = *reinterpret_cast<const f32x4_t*>(w0 + 16 * blk);
#pragma unroll
      for (int s = 0; s < 4; ++s) acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], bw0[s], acc0, 0, 0, 0);
      if (ntiles == 2) {
        const f32x4_t bw1 = *reinterpret_cast<const f32x4_t*>(w1 + 16 * blk);
#pragma unroll
        for (int s = 0; s < 4; ++s) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], bw1[s], acc1, 0, 0, 0);
      }
    }
    // D: lane (j, kq) holds rows 4 kq + r of column j (tile 0) / 16 + j (tile 1)
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const long long row = row0 + 4 * kq + rr;
      if (row >= p.rows) continue;
      float* yr = p.y + row * p.ldh;
      if (j < p.ldh) yr[j] = acc0[rr] + b0;
      if (ntiles == 2 && 16 + j < p.ldh) yr[16 + j] = acc1[rr] + b1;
    }
  }
}

}  // namespace

extern "C" int seedhip_heads_supported(int feat, int ldh) {
  return feat >= 64 && feat <= kMaxFeat && feat % 64 == 0 && ldh >= 4 && ldh <= kMaxN && ldh % 4 == 0;
}

extern "C" int seedhip_heads_fwd(const float* x, int ldx, const float* w, const float* bias, long long rows, int feat,
                                 int ldh, float* y, void* stream) {
  SEEDHIP_REQUIRE(seedhip_heads_supported(feat, ldh), "heads_fwd: need feat %% 64 == 0, feat <= 512, ldh %% 4 == 0, ldh <= 32 (feat = %d, ldh = %d)", feat, ldh);
  SEEDHIP_REQUIRE(rows >= 0 && ldx >= feat, "heads_fwd: bad rows / ldx");
  if (rows == 0) return SEEDHIP_OK;
  SEEDHIP_REQUIRE(x && w && y, "heads_fwd: null pointer");
  SEEDHIP_REQUIRE(((((uintptr_t)x) & 15) == 0) && ldx % 4 == 0, "heads_fwd: x must be 16-byte aligned with ldx %% 4 == 0");
  FwdParams p{x, ldx, w, bias, rows, feat, ldh, y};
  const size_t lds = (size_t)kMaxN * (feat + 4) * sizeof(float);
  (void)hipFuncSetAttribute((const void*)heads_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const long long tiles = (rows + 15) / 16;
  long long grid = (tiles + kWaves - 1) / kWaves;
  if (grid > 1024) grid = 1024;
  hipLaunchKernelGGL(heads_fwd_kernel, dim3((int)grid), dim3(kThreads), lds, (hipStream_t)stream, p);
  return seedhip::check_launch("heads_fwd_kernel");
}

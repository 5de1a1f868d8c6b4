// Weight-stationary gather-GEMM for small strided 'valid' convolutions whose whole kernel fits LDS -- the second
// Atari conv, Conv2D(32, 4, 2) on the 20x20x16 output of the first (/root/reference/atari/networks.py:236), forward
// and data gradient.
//
//   C[m, n] = sum_k A(m, k) * W'(k, n)        rows m = flat (image, grid-y, grid-x), persistent workgroups over m-tiles
//
//   forward        m = output pixel (oy, ox);  k = (ky, kx, ci): each ky contributes kw*cin CONTIGUOUS floats of the
//                  NHWC input row, so a 32-deep k-tile is one 128-byte segment per row;  n = co;  W' = W as stored.
//   data gradient  m = "super-pixel" (a, b) = the s x s block of input pixels (s*a+py, s*b+px);  n = (py, px, ci);
//                  k = (jy, jx, co):  dX[s*a+py, s*b+px, ci] = sum dY[a-jy, b-jx, co] * W[py+s*jy, px+s*jx, ci, co]
//                  -- ONE GEMM for all stride-parity classes, A rows are again contiguous 128-byte segments of dY
//                  (zero-filled where a-jy / b-jx leave the map), no structural zeros multiplied.
// W' [K][N] (<= 40 KB) is written to LDS once per workgroup and stays; only A streams: global -> registers (float4)
// -> LDS [row][k] (stride 40 floats) -> ds_read_b128 fragments through the k-permutation of gemm.h, while the W'
// fragments are x-interleaved reads (lane owns NR consecutive n), so the epilogue stores 8/16 bytes per lane.
// Waves are autonomous after the W' load (each covers all N columns of its own rows): no workgroup barriers in
// the loop, and the register prefetch (two k-tiles deep) runs across m-tile boundaries.
// Measured on MI355X (T=20, B=512: 10752 images): forward 0.19 ms (implicit-GEMM core: 0.234), data gradient
// 0.24 ms (stride-parity halo kernel: 0.377).  Both are within ~2x of their HBM time (386 MB / 661 MB of compulsory
// traffic: input, output, ReLU mask), so what matters is overlapping many waves' load / MFMA / store phases; a
// variant holding W' in 128 VGPRs with A fragments loaded straight from global (no LDS at all) was slower at its
// 2 waves per SIMD.
#pragma once
#include "common.h"
#include "igemm.h"
#include "wsgemm_geom.h"
#include "../../include/seedhip.h"
#include <cstdlib>

namespace seedhip {
namespace wsgemm {

typedef float f32x4_t __attribute__((ext_vector_type(4)));
template <int R> struct Vec;
template <> struct Vec<4> { typedef f32x4_t type; };
template <> struct Vec<2> { typedef float type __attribute__((ext_vector_type(2))); };

// LDS traffic of one wave is ordered; this only stops the compiler from moving LDS accesses across the point.
__device__ __forceinline__ void wave_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int MR, int NR, int WAVES>
__global__ void __launch_bounds__(64 * WAVES)
ws_kernel(const Params p) {
  constexpr int N = 16 * NR, LDB = (NR == 2) ? N + 8 : N;     // b64 fragment rows 4 apart: +32 banks
  constexpr int kThreads = 64 * WAVES;
  constexpr int VA = 2 * MR;                  // float4 per lane per k-tile of the wave's 16*MR rows
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Bs = smem;                           // [K][LDB]
  float* As = smem + p.K * LDB;               // [BM][LDA]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lx = lane & 15, kq = lane >> 4;

  // ---- W' -> LDS, once ----
  if (p.mode == 0) {
    for (int idx = tid; idx < p.K * N / 4; idx += kThreads) {
      const int e = idx * 4, k = e / N, n = e - k * N;
      *reinterpret_cast<float4*>(Bs + k * LDB + n) = *reinterpret_cast<const float4*>(p.W + e);
    }
  } else {
    const int jw = p.kw / p.s;
    for (int idx = tid; idx < p.K * N; idx += kThreads) {       // idx = k*N + n: conflict-free LDS writes
      const int k = idx / N, n = idx - k * N;
      const int co = k % p.cout, tap = k / p.cout, jy = tap / jw, jx = tap - jy * jw;
      const int ci = n % p.cin, cls = n / p.cin, py = cls / p.s, px = cls - py * p.s;
      Bs[k * LDB + n] = p.W[(((py + p.s * jy) * p.kw + px + p.s * jx) * p.cin + ci) * p.cout + co];
    }
  }

  __syncthreads();                              // the only workgroup barrier: W' visible to all four waves

  // ---- wave-autonomous from here: every wave covers all N columns, so the A rows of its 16*MR-row tile are
  // private to it.  Each wave stages its own rows into its own LDS region (in-order LDS per wave: a compiler
  // fence is all the synchronisation needed) and walks its own sequence of m-tiles; the four waves drift apart
  // and fill each other's load / epilogue phases on the matrix pipe.
  // vector i of this lane = row (lane >> 3) + 8 i of the wave tile, floats kc .. kc+3 of the k-tile
  float* Aw = As + wave * (MR * 16 * LDA);
  const int kc = (lane & 7) * 4;
  unsigned rowbase[VA], vmask[VA];
  float4 r0[VA], r1[VA];                       // two register stages: k-tiles q+1 and q+2 of this wave's sequence
  auto setup_tile = [&](int wt) {
#pragma unroll
    for (int i = 0; i < VA; ++i) {
      const int m = wt * (MR * 16) + (lane >> 3) + 8 * i;
      unsigned mask = 0, base = 0;
      if (m < p.M) {
        uint32_t img, rem, a, b;
        p.d_g.divmod((uint32_t)m, img, rem);
        p.d_gw.divmod(rem, a, b);
        base = img * p.a_img_stride + a * p.a_row_stride + b * p.a_col_stride + kc;
        if (p.mode == 0) mask = 0xffffffffu;                  // forward: every tap of a 'valid' conv is inside
        else for (int t = 0; t < p.nkt; ++t) {
          const int y = (int)a + p.tile_dy[t], x = (int)b + p.tile_dx[t];
          if (y >= 0 && y < p.vh && x >= 0 && x < p.vw) mask |= 1u << t;
        }
      }
      rowbase[i] = base; vmask[i] = mask;
    }
  };
  const int wstride = gridDim.x * WAVES;
  int ltile = blockIdx.x * WAVES + wave, lkt = 0;            // load cursor: runs two k-tiles ahead, across m-tiles
  auto advance_load = [&](float4 (&r)[VA]) {
    if (ltile >= p.ntiles) return;
    if (lkt == 0) setup_tile(ltile);
    const int off = p.tile_off[lkt];
#pragma unroll
    for (int i = 0; i < VA; ++i) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if ((vmask[i] >> lkt) & 1u) {
        v = *reinterpret_cast<const float4*>(p.A + (long long)rowbase[i] + off);
        if (p.a_relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      }
      r[i] = v;
    }
    if (++lkt == p.nkt) { lkt = 0; ltile += wstride; }
  };

  const float* a_frag = Aw + lx * LDA + 4 * kq;
  const float* b_frag = Bs + (4 * kq) * LDB + NR * lx;
  typedef typename Vec<NR>::type bvec_t;

  advance_load(r0);
  advance_load(r1);
  for (int tile = blockIdx.x * WAVES + wave; tile < p.ntiles; tile += wstride) {
    f32x4_t acc[MR][NR];
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
      for (int j = 0; j < NR; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // data gradient: output offsets of this lane's MR x 4 rows, and the ReLU-mask values, fetched now so that their
    // latency hides under the tile's MFMAs instead of stalling the epilogue
    const int n = NR * lx;
    unsigned out_at[MR][4];
    bvec_t mpre[MR][4];
    if (p.mode == 1) {
      const int cls = n / p.cin, ci = n - cls * p.cin, py = cls / p.s, px = cls - py * p.s;
#pragma unroll
      for (int i = 0; i < MR; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = tile * (MR * 16) + 16 * i + 4 * kq + r;
          unsigned at = 0xffffffffu;
          bvec_t mv;
#pragma unroll
          for (int j = 0; j < NR; ++j) mv[j] = 1.f;
          if (m < p.M) {
            uint32_t img, rem, a, b;
            p.d_g.divmod((uint32_t)m, img, rem);
            p.d_gw.divmod(rem, a, b);
            const int oy = (int)a * p.s + py, ox = (int)b * p.s + px;
            if (oy < p.ih && ox < p.iw) {
              at = ((img * p.ih + oy) * p.iw + ox) * p.ld_in + ci;
              if (p.mask) mv = *reinterpret_cast<const bvec_t*>(p.mask + at);
            }
          }
          out_at[i][r] = at; mpre[i][r] = mv;
        }
      }
    }

    auto step = [&](int kt, float4 (&r)[VA]) {
      wave_fence();                                         // fragment reads of the previous k-tile are issued
#pragma unroll
      for (int i = 0; i < VA; ++i) *reinterpret_cast<float4*>(Aw + ((lane >> 3) + 8 * i) * LDA + kc) = r[i];
      wave_fence();
      advance_load(r);                                      // the k-tile two steps ahead, into the set just drained
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        f32x4_t a_kc[MR];
        bvec_t b_oc[4];
#pragma unroll
        for (int i = 0; i < MR; ++i) a_kc[i] = *reinterpret_cast<const f32x4_t*>(a_frag + i * 16 * LDA + h * 16);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) b_oc[kk] = *reinterpret_cast<const bvec_t*>(b_frag + (kt * BK + h * 16 + kk) * LDB);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
          for (int i = 0; i < MR; ++i)
#pragma unroll
            for (int j = 0; j < NR; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_kc[i][kk], b_oc[kk][j], acc[i][j], 0, 0, 0);
      }
    };
    for (int kt = 0; kt < p.nkt; kt += 2) {                 // nkt is even (plan): register sets alternate statically
      step(kt, r0);
      step(kt + 1, r1);
    }

    // ---- epilogue: lane holds columns n = NR*lx .. +NR-1 of rows 16 i + 4 kq + r ----
#pragma unroll
    for (int i = 0; i < MR; ++i) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        bvec_t v;
#pragma unroll
        for (int j = 0; j < NR; ++j) v[j] = acc[i][j][r];
        if (p.mode == 0) {
          const int m = tile * (MR * 16) + 16 * i + 4 * kq + r;
          if (m >= p.M) continue;
          const long long at = (long long)m * p.ldc + n;
          if (p.bias) { const bvec_t bv = *reinterpret_cast<const bvec_t*>(p.bias + n); v += bv; }
          if (p.residual) { const bvec_t rv = *reinterpret_cast<const bvec_t*>(p.residual + at); v += rv; }
          if (p.out_relu) {
#pragma unroll
            for (int j = 0; j < NR; ++j) v[j] = fmaxf(v[j], 0.f);
          }
          *reinterpret_cast<bvec_t*>(p.C + at) = v;
        } else {
          const unsigned at = out_at[i][r];
          if (at == 0xffffffffu) continue;
#pragma unroll
          for (int j = 0; j < NR; ++j) if (!(mpre[i][r][j] > 0.f)) v[j] = 0.f;
          if (p.add) { const bvec_t av = *reinterpret_cast<const bvec_t*>(p.add + at); v += av; }
          *reinterpret_cast<bvec_t*>(p.C + at) = v;
        }
      }
    }
  }
}


// Same algorithm on a VALU diet.  On gfx950 a SIMD's VALU instructions and MFMAs do not overlap (SQ counters on the
// cfg2 step: SQ_VALU_MFMA_COEXEC_CYCLES = 0; kernel times follow 32 cycles per fp32 MFMA + ~4 per VALU instruction),
// and the generic kernel above issues 3.7 (forward) to 5 (data gradient) VALU instructions per MFMA: run-time k-tile
// cursor, per-load branches / selects, two divisions per row, 64-bit addresses, a ReLU select that is never taken.
// Here everything the shape fixes is static (NKT k-tiles unrolled, MR = 1, 8 waves, no input ReLU) and:
//   * A is read through a buffer resource: per load ONE byte offset register (set to an out-of-range value for
//     border taps / tail rows: the hardware returns zeros) and the k-tile offset in an SGPR -- no address arithmetic,
//     no select on the data;
//   * rows after the first of a lane are stepped instead of divided; bounds checks that the shape makes vacuous
//     (full tiles, map sizes that are multiples of the stride) are skipped by uniform branches;
//   * loads run ONE WHOLE TILE ahead (see the tile loop).
template <int NR, int NKT, int MODE>
__global__ void __launch_bounds__(512)
ws_fast_kernel(const Params p) {
  constexpr int WAVES = 8, N = 16 * NR, LDB = (NR == 2) ? N + 8 : N, kThreads = 64 * WAVES;
  constexpr unsigned kOOB = 0xC0800000u;                      // beyond num_records of any buffer we bind (< 2 GB); the bit
                                                              // pattern of -4.0f: an inline constant, free as an operand
  static_assert(NKT <= kMaxTiles, "k-tile tables");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Bs = smem;                           // [K][LDB]
  float* As = smem + p.K * LDB;               // [WAVES][16][LDA]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lx = lane & 15, kq = lane >> 4;

  if (MODE == 0) {
    for (int idx = tid; idx < p.K * N / 4; idx += kThreads) {
      const int e = idx * 4, k = e / N, n = e - k * N;
      *reinterpret_cast<float4*>(Bs + k * LDB + n) = *reinterpret_cast<const float4*>(p.W + e);
    }
  } else {
    // run-time integer divisions cost ~20 VALU each and this loop runs in EVERY workgroup: with power-of-two channel
    // counts / stride the re-indexing (ws_wprime_src) is shifts and masks
    for (int idx = tid; idx < p.K * N; idx += kThreads) {
      const int k = idx / N, n = idx - k * N;                 // N is a compile-time power of two
      Bs[k * LDB + n] = p.W[ws_wprime_src(p, k, n)];
    }
  }
  __syncthreads();

  float* Aw = As + wave * (16 * LDA);
  const int kc = (lane & 7) * 4, srow = lane >> 3;            // this lane stages rows srow and srow + 8, floats kc..kc+3
  // buffer view of A starting at the most negative k-tile offset, so that every k-tile offset is an unsigned SGPR
  int minoff = 0;
#pragma unroll
  for (int t = 0; t < NKT; ++t) minoff = p.tile_off[t] < minoff ? p.tile_off[t] : minoff;
  unsigned soff[NKT]; int tdy[NKT], tdx[NKT];                 // uniform: SGPRs
#pragma unroll
  for (int t = 0; t < NKT; ++t) {
    soff[t] = __builtin_amdgcn_readfirstlane((unsigned)(p.tile_off[t] - minoff) * 4u);
    tdy[t] = p.tile_dy[t]; tdx[t] = p.tile_dx[t];
  }
  // (readfirstlane: the descriptor and the offsets are uniform by construction; saying so keeps them in SGPRs and the
  // loads free of waterfall loops)
  const uint64_t abase = reinterpret_cast<uint64_t>(p.A + minoff);
  const uint64_t sbase = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((unsigned)(abase >> 32)) << 32) |
                         (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((unsigned)abase);   // (the builtin returns int)
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      reinterpret_cast<void*>(sbase), 0, __builtin_amdgcn_readfirstlane((int)p.a_bytes - minoff * 4), 0x00020000);
  // the epilogue's tensors (indexed alike) as buffer views too: 32-bit byte offsets, no 64-bit address arithmetic
  auto view = [&](const void* q) {
    const uint64_t qb = reinterpret_cast<uint64_t>(q);
    const uint64_t sq = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((unsigned)(qb >> 32)) << 32) |
                        (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((unsigned)qb);
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(sq), 0, __builtin_amdgcn_readfirstlane((int)p.c_bytes), 0x00020000);
  };
  const __amdgpu_buffer_rsrc_t c_rsrc = view(p.C), m_rsrc = view(MODE == 1 ? (const void*)p.mask : (const void*)p.residual),
                               add_rsrc = view(p.add);
  auto locate = [&](uint32_t m, uint32_t& img, uint32_t& a, uint32_t& b) { ws_locate(p, m, img, a, b); };
  auto advance = [&](uint32_t step, uint32_t& img, uint32_t& a, uint32_t& b) { ws_advance(p, step, img, a, b); };      // step < gw

  // load cursor: byte offset of (row, k-tile) or kOOB -- forward: one per row (every tap of a 'valid' conv is inside)
  constexpr int NV = MODE == 0 ? 1 : NKT;
  unsigned voff[2][NV];
  auto setup = [&](int wt) {
    uint32_t img, a, b;
    const uint32_t m0 = (uint32_t)wt * 16u + (uint32_t)srow;
    locate(m0, img, a, b);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (i) advance(8u, img, a, b);
      const unsigned byte = ws_row_byte(p, img, a, b, kc);
      const bool row_ok = m0 + 8u * i < (uint32_t)p.M;
      if (MODE == 0) voff[i][0] = row_ok ? byte : kOOB;
      else {
#pragma unroll
        for (int t = 0; t < NKT; ++t) {
          const bool ok = row_ok && ws_tap_ok(p, a, b, tdy[t], tdx[t]);
          voff[i][t] = ok ? byte : kOOB;
        }
      }
    }
  };
  typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
  u32x4_t rg[NKT][2];                                         // one register stage per k-tile x two rows: a whole tile ahead
  auto fetch = [&](int kt) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
      rg[kt][i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff[i][MODE == 0 ? 0 : kt], soff[kt], 0);
  };

  const float* a_frag = Aw + lx * LDA + 4 * kq;
  const float* b_frag = Bs + (4 * kq) * LDB + NR * lx;
  typedef typename Vec<NR>::type bvec_t;
  typedef unsigned uvec_t __attribute__((ext_vector_type(NR)));
  auto ld_vec = [&](const __amdgpu_buffer_rsrc_t& r, unsigned byte_off) -> bvec_t {
    if constexpr (NR == 4) return __builtin_bit_cast(bvec_t, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0));
    else return __builtin_bit_cast(bvec_t, __builtin_amdgcn_raw_buffer_load_b64(r, byte_off, 0, 0));
  };
  auto st_vec = [&](const bvec_t& v, unsigned byte_off) {
    if constexpr (NR == 4) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uvec_t, v), c_rsrc, byte_off, 0, 0);
    else __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(uvec_t, v), c_rsrc, byte_off, 0, 0);
  };
  const int wstride = gridDim.x * WAVES;
  const int n = NR * lx;
  // data gradient: dx offset of super-pixel (img, a, b), class (py, px), channel ci is linear in (img, a, b)
  unsigned e_const = 0;
  int e_py = 0, e_px = 0;
  if (MODE == 1) e_const = ws_dgrad_col(p, n, e_py, e_px);
  const bool exact = MODE == 1 && p.gh * p.s == p.ih && p.gw * p.s == p.iw;   // no super-pixel hangs over the map
  bvec_t bias_v;
#pragma unroll
  for (int j = 0; j < NR; ++j) bias_v[j] = 0.f;
  if (MODE == 0 && p.bias) bias_v = *reinterpret_cast<const bvec_t*>(p.bias + n);

  // Loads run ONE WHOLE TILE ahead: k-tile kt of the next tile is requested right after k-tile kt of this one went to
  // LDS, i.e. before this tile's epilogue stores.  A wave's memory operations retire in issue order, so a load issued
  // behind those stores would also wait for their acknowledgement.
  int tile = blockIdx.x * WAVES + wave;
  setup(tile);
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt) fetch(kt);
  for (; tile < p.ntiles; tile += wstride) {
    setup(tile + wstride);                                    // the load cursor: this wave's next tile
    f32x4_t acc[NR];                                          // first written by the first MFMA of the tile (C = 0)

    // epilogue addresses (and the ReLU-mask values: their latency hides under the tile's MFMAs)
    const uint32_t m0 = (uint32_t)tile * 16u + 4u * (uint32_t)kq;
    const bool full = (uint32_t)tile * 16u + 16u <= (uint32_t)p.M;              // uniform
    unsigned out_at[4];
    bvec_t mpre[4];
    if (MODE == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) out_at[r] = (full || m0 + r < (uint32_t)p.M) ? (m0 + r) * (unsigned)p.ldc + n : 0xffffffffu;
    } else {
      uint32_t img, a, b;
      locate(m0, img, a, b);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (r) advance(1u, img, a, b);
        unsigned at = ws_dgrad_at(p, img, a, b, e_const, e_py, e_px, exact);
        if (!full && m0 + r >= (uint32_t)p.M) at = 0xffffffffu;
        out_at[r] = at;
        if (p.mask) mpre[r] = ld_vec(m_rsrc, at == 0xffffffffu ? kOOB : at * 4u);    // (an invalid row is never stored)
      }
    }

#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      wave_fence();                                           // fragment reads of the previous k-tile are issued
#pragma unroll
      for (int i = 0; i < 2; ++i) *reinterpret_cast<u32x4_t*>(Aw + (srow + 8 * i) * LDA + kc) = rg[kt][i];
      wave_fence();
      fetch(kt);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const f32x4_t a_kc = *reinterpret_cast<const f32x4_t*>(a_frag + h * 16);
        bvec_t b_oc[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) b_oc[kk] = *reinterpret_cast<const bvec_t*>(b_frag + (kt * BK + h * 16 + kk) * LDB);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
          for (int j = 0; j < NR; ++j)
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_kc[kk], b_oc[kk][j],
                                                          (kt == 0 && h == 0 && kk == 0) ? f32x4_t{0.f, 0.f, 0.f, 0.f} : acc[j], 0, 0, 0);
      }
    }

#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const unsigned at = out_at[r];
      if (at == 0xffffffffu) continue;
      bvec_t v;
#pragma unroll
      for (int j = 0; j < NR; ++j) v[j] = acc[j][r];
      if (MODE == 0) {
        v += bias_v;
        if (p.residual) v += ld_vec(m_rsrc, at * 4u);
        if (p.out_relu) {
#pragma unroll
          for (int j = 0; j < NR; ++j) v[j] = fmaxf(v[j], 0.f);
        }
      } else {
        if (p.mask) {
#pragma unroll
          for (int j = 0; j < NR; ++j) v[j] = mpre[r][j] > 0.f ? v[j] : 0.f;
        }
        if (p.add) v += ld_vec(add_rsrc, at * 4u);
      }
      st_vec(v, at * 4u);
    }
  }
}


// Third generation: the same weight-stationary GEMM with (almost) NO per-tile integer work.  The SQ counters of the cfg2
// step say a SIMD does not run VALU work under another wave's MFMAs, so the ~470 VALU / ~280 SALU instructions and ~20
// branches that ws_fast_kernel spends per 128-MFMA tile on row decoding, tap validity and scatter addresses are matrix
// time lost one-for-one.  Here:
//   * the geometry of ONE image (row -> per-k-tile byte offset or "outside the map", row -> output byte offset) is
//     tabulated in LDS once per workgroup (G = gh * gw rows: 81 / 100 entries); a tile row costs one magic division by
//     G, one ds_read of its table line and one add per load offset.  Invalid taps are the offset 0x80000000: beyond
//     num_records of the buffer view, the hardware returns zeros; rows past M fall beyond the buffers by themselves
//     (loads return zeros, stores are dropped): no tail code, no branch in the tile loop;
//   * the MFMA operands are SWAPPED (W' supplies the 16 "rows" of the instruction, the data rows its 16 "columns"):
//     a lane then holds four CONSECUTIVE output channels of ONE data row per accumulator, which is a 16-byte store as it
//     stands -- no register transposition, one output address per lane and tile instead of four, and the stride-parity
//     class / n-tile offset of the store is a uniform soffset.
// Per 128-MFMA tile: ~20 (forward) / ~60 (data gradient, 32 of them the ReLU-mask selects) VALU instructions.
// EXP (tools/probes/ws_probe.hip only; the library instantiates EXP = 0): leave one ingredient out to see what the
// others cost -- 1 no MFMAs, 2 no global A loads after the first tile, 4 no stores, 8 no mask loads, 16 no LDS fragment
// reads, 32 no LDS staging writes.
// MPF (data gradient with the fp32 mask): a tile's ReLU-mask vectors are requested one tile AHEAD.  Requested at the head
// of their own tile they are the youngest vector-memory operations in flight when the tile's first operand wait comes
// (vmcnt retires in order; with stores pending that wait is vmcnt(0) anyway): every tile began with a full memory
// round trip that no other work of the wave could cover.
template <int NT, int NKT, int MODE, int EXP = 0, bool BITS = false, bool MPF = false>
__global__ void __launch_bounds__(512)
ws_tab_kernel(const Params p) {
  static_assert(!BITS || MODE == 1, "the byte mask belongs to the data gradient");
  constexpr int WAVES = 8, N = 16 * NT, K = NKT * BK, LDW = K + 8, kThreads = 64 * WAVES;
  constexpr int NV = MODE == 0 ? 1 : NKT;                     // load offsets per row
  constexpr unsigned kOut = 0x80000000u;                      // "outside": + any row base (< 2^31) stays beyond num_records
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Wt = smem;                                           // [N][LDW]   W' transposed: k contiguous
  float* As = Wt + N * LDW;                                   // [WAVES][16][LDA]
  unsigned* tab_ld = reinterpret_cast<unsigned*>(As + WAVES * 16 * LDA);   // [G][NV]
  const int G = p.gh * p.gw;
  unsigned* tab_out = tab_ld + G * NV;                        // [G]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lx = lane & 15, kq = lane >> 4;

  // ---- once per workgroup: W' -> LDS (transposed), geometry tables ----
  for (int idx = tid; idx < K * N; idx += kThreads) {
    const int k = idx / N, n = idx - k * N;                   // N is a compile-time power of two
    Wt[n * LDW + k] = MODE == 0 ? p.W[idx] : p.W[ws_wprime_src(p, k, n)];
  }
  for (int r = tid; r < G; r += kThreads) {
    const int a = r / p.gw, b = r - a * p.gw;
    const unsigned byte0 = ((unsigned)a * p.a_row_stride + (unsigned)b * p.a_col_stride) * 4u;
    if (MODE == 0) {
      tab_ld[r] = byte0;
      tab_out[r] = (unsigned)r * (unsigned)p.ldc * 4u;
    } else {
#pragma unroll
      for (int t = 0; t < NKT; ++t) tab_ld[r * NKT + t] = ws_tap_ok(p, a, b, p.tile_dy[t], p.tile_dx[t]) ? byte0 : kOut;
      tab_out[r] = ((unsigned)a * (unsigned)(p.s * p.iw * p.ld_in) + (unsigned)b * (unsigned)(p.s * p.ld_in)) * 4u;
    }
  }
  __syncthreads();

  float* Aw = As + wave * (16 * LDA);
  const int kc = (lane & 7) * 4, srow = lane >> 3;            // this lane stages rows srow and srow + 8, floats kc..kc+3
  int minoff = 0;
#pragma unroll
  for (int t = 0; t < NKT; ++t) minoff = p.tile_off[t] < minoff ? p.tile_off[t] : minoff;
  unsigned soff[NKT];                                         // uniform: SGPRs
#pragma unroll
  for (int t = 0; t < NKT; ++t) soff[t] = __builtin_amdgcn_readfirstlane((unsigned)(p.tile_off[t] - minoff) * 4u);
  auto view = [&](const void* q, long long bytes) {
    const uint64_t qb = reinterpret_cast<uint64_t>(q);
    const uint64_t sq = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((unsigned)(qb >> 32)) << 32) |
                        (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((unsigned)qb);   // (the builtin returns int)
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(sq), 0, __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000);
  };
  const __amdgpu_buffer_rsrc_t rsrc = view(p.A + minoff, p.a_bytes - (long long)minoff * 4),
                               c_rsrc = view(p.C, p.c_bytes),
                               m_rsrc = BITS ? view(p.mask_bits, p.c_bytes >> 4) : view(p.mask, p.c_bytes);
  const unsigned a_img_bytes = p.a_img_stride * 4u;
  const unsigned c_img_bytes = MODE == 0 ? (unsigned)G * (unsigned)p.ldc * 4u : (unsigned)(p.ih * p.iw * p.ld_in) * 4u;
  // uniform store offsets of the NT accumulators: forward n = 16 t + 4 kq + r; data gradient n = (class, ci), cin % 16 == 0
  unsigned coff[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    if (MODE == 0) coff[t] = 64u * t;
    else {
      const int cls = (16 * t) / p.cin, ci0 = 16 * t - cls * p.cin, py = cls / p.s, px = cls - py * p.s;
      coff[t] = (unsigned)((py * p.iw + px) * p.ld_in + ci0) * 4u;
    }
    coff[t] = __builtin_amdgcn_readfirstlane(coff[t]);
  }
  unsigned cbit[NT];                                          // the same offsets into the byte mask (one byte per 16 of C)
#pragma unroll
  for (int t = 0; t < NT; ++t) cbit[t] = __builtin_amdgcn_readfirstlane(coff[t] >> 4);

  typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
  unsigned voff[2][NV];
  auto setup = [&](int wt) {                                  // load offsets of wave tile wt (may lie past the end)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const uint32_t m = (uint32_t)wt * 16u + (uint32_t)(srow + 8 * i);
      uint32_t img, rem;
      p.d_g.divmod(m, img, rem);
      const unsigned base = img * a_img_bytes + (unsigned)(kc * 4);
      if (MODE == 0) voff[i][0] = tab_ld[rem] + base;
      else {
#pragma unroll
        for (int t4 = 0; t4 < NKT; t4 += 4) {
          const u32x4_t tv = *reinterpret_cast<const u32x4_t*>(tab_ld + rem * NKT + t4);
#pragma unroll
          for (int j = 0; j < 4; ++j) voff[i][t4 + j] = tv[j] + base;
        }
      }
    }
  };
  u32x4_t rg[NKT][2];                                         // one register stage per k-tile x two rows: a whole tile ahead
  bool first_tile = true;
  auto fetch = [&](int kt) {
    if ((EXP & 2) && !first_tile) return;
#pragma unroll
    for (int i = 0; i < 2; ++i)
      rg[kt][i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff[i][MODE == 0 ? 0 : kt], soff[kt], 0);
  };
  const float* a_frag = Aw + lx * LDA + 4 * kq;               // MFMA "B" operand: data row lx, k = 4 kq .. 4 kq + 3
  const float* w_frag = Wt + lx * LDW + 4 * kq;               // MFMA "A" operand: output channel 16 t + lx
  f32x4_t bias_v[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    bias_v[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    if (MODE == 0 && p.bias) bias_v[t] = *reinterpret_cast<const f32x4_t*>(p.bias + 16 * t + 4 * kq);
  }
  const int wstride = gridDim.x * WAVES;
  const bool relu = p.out_relu != 0, has_mask = BITS || p.mask != nullptr;

  int tile = blockIdx.x * WAVES + wave;
  setup(tile);
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt) fetch(kt);
  first_tile = false;
  f32x4_t av_keep = f32x4_t{1.f, 2.f, 3.f, 4.f}, wv_keep[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) wv_keep[t] = f32x4_t{0.5f, 0.25f, 0.125f, 1.f};
  auto out_offset = [&](int tl) {                             // (rows >= M land beyond c_bytes: stores dropped, mask loads zero)
    const uint32_t m = (uint32_t)tl * 16u + (uint32_t)lx;
    if (MODE == 0) return m * (unsigned)(p.ldc * 4) + (unsigned)(kq * 16);
    uint32_t img, rem;
    p.d_g.divmod(m, img, rem);
    return img * c_img_bytes + tab_out[rem] + (unsigned)(kq * 16);
  };
  u32x4_t mnext[NT];                                          // MPF: the next tile's mask vectors
  unsigned at_next = 0;
  if (MPF) {
    at_next = out_offset(tile);
#pragma unroll
    for (int t = 0; t < NT; ++t) mnext[t] = __builtin_amdgcn_raw_buffer_load_b128(m_rsrc, at_next, coff[t], 0);
  }
  for (; tile < p.ntiles; tile += wstride) {
    setup(tile + wstride);
    // output address of this lane's data row m = 16 tile + lx (columns 4 kq .. of every n-tile)
    const unsigned at = MPF ? at_next : out_offset(tile);
    u32x4_t mpre[NT];
    int mbit[NT];
    if (MPF) {
#pragma unroll
      for (int t = 0; t < NT; ++t) mpre[t] = mnext[t];
    } else if (BITS) {                                               // one byte per accumulator: this lane's four channels
#pragma unroll
      for (int t = 0; t < NT; ++t) mbit[t] = (int)__builtin_amdgcn_raw_buffer_load_b8(m_rsrc, at >> 4, cbit[t], 0);
    } else if (MODE == 1 && has_mask) {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        if (EXP & 8) mpre[t] = u32x4_t{0x3f800000u, 0x3f800000u, 0u, 0x3f800000u};
        else mpre[t] = __builtin_amdgcn_raw_buffer_load_b128(m_rsrc, at, coff[t], 0);
      }
    }
    f32x4_t acc[NT];
    if (EXP & 1) {
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      wave_fence();                                           // fragment reads of the previous k-tile are issued
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        if (EXP & 32) asm volatile("" :: "v"(rg[kt][i]));
        else *reinterpret_cast<u32x4_t*>(Aw + (srow + 8 * i) * LDA + kc) = rg[kt][i];
      }
      wave_fence();
      if (MPF && kt == 0) {                                   // behind the tile's operand wait: in flight for a whole tile
        at_next = out_offset(tile + wstride);
#pragma unroll
        for (int t = 0; t < NT; ++t) mnext[t] = __builtin_amdgcn_raw_buffer_load_b128(m_rsrc, at_next, coff[t], 0);
      }
      fetch(kt);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        f32x4_t av, wv[NT];
        if (EXP & 16) {
          av = av_keep;
#pragma unroll
          for (int t = 0; t < NT; ++t) wv[t] = wv_keep[t];
        } else {
          av = *reinterpret_cast<const f32x4_t*>(a_frag + h * 16);
#pragma unroll
          for (int t = 0; t < NT; ++t) wv[t] = *reinterpret_cast<const f32x4_t*>(w_frag + t * 16 * LDW + kt * BK + h * 16);
        }
        if (EXP & 1) {                                        // operands stay live, no matrix work
          asm volatile("" :: "v"(av));
#pragma unroll
          for (int t = 0; t < NT; ++t) asm volatile("" :: "v"(wv[t]));
          continue;
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
          for (int t = 0; t < NT; ++t)
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[t][kk], av[kk],
                                                          (kt == 0 && h == 0 && kk == 0) ? f32x4_t{0.f, 0.f, 0.f, 0.f} : acc[t], 0, 0, 0);
      }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      f32x4_t v = acc[t];
      if (MODE == 0) {
        v += bias_v[t];
        if (relu) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
        }
      } else if (BITS) {                                       // bit r sign-extended to a 0 / ~0 word, ANDed in
#pragma unroll
        for (int r = 0; r < 4; ++r)
          v[r] = __uint_as_float(__float_as_uint(v[r]) & (unsigned)((mbit[t] << (31 - r)) >> 31));
      } else if (has_mask) {
        const f32x4_t mv = __builtin_bit_cast(f32x4_t, mpre[t]);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = mv[r] > 0.f ? v[r] : 0.f;
      }
      if (EXP & 4) asm volatile("" :: "v"(v));
      else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), c_rsrc, at, coff[t], 0);
    }
  }
}

// `dry`: no launch -- SEEDHIP_OK iff the call would be served (the byte-mask query of conv.hip).
inline int launch(Params& p, Plan& pl, hipStream_t s, bool dry = false) {
  pl.mr = 1;                                                 // measured (cfg2 step): 16-row wave tiles, 8 waves per workgroup
  constexpr int waves = 8;
  const int ldb = pl.nr == 2 ? p.N + 8 : p.N;
  pl.lds = ((size_t)p.K * ldb + (size_t)waves * 16 * pl.mr * LDA) * sizeof(float);
  p.ntiles = (p.M + 16 * pl.mr - 1) / (16 * pl.mr);          // wave tiles
  int per_cu = (int)((160 * 1024) / (pl.lds + 512)); if (per_cu > 32 / waves) per_cu = 32 / waves; if (per_cu < 1) per_cu = 1;
  const int wgs = (p.ntiles + waves - 1) / waves;
  pl.grid = wgs < 256 * per_cu ? wgs : 256 * per_cu;
  // the specialised kernel: MR = 1, 8 waves, static k-tile count
  if (pl.mr == 1 && waves == 8 && p.gw >= 8 && !p.a_relu && p.a_bytes < (1LL << 31) - (1 << 20) && p.c_bytes < (1LL << 31) &&
      (p.mode == 1 || (long long)p.M * p.ldc < (1LL << 32) - 64)) {
    // 100-128 VGPRs: four waves per SIMD, i.e. two 8-wave workgroups per CU are resident whatever LDS allows; a
    // persistent grid of exactly the resident workgroups avoids a second, ragged round (data gradient 0.196 -> 0.189 ms)
    if (wgs > 256 * 2) pl.grid = 256 * 2;
    // table-driven variant (ws_tab_kernel): dense output rows, no residual / add, data gradient with whole super-pixels
    // and 16-channel-aligned classes
    const bool tab_ok = !p.residual && !p.add && p.gh * p.gw <= 1024 &&
        (p.mode == 0 ? (p.ldc == p.N && p.ldc % 4 == 0)
                     : (p.cin % 16 == 0 && p.gh * p.s == p.ih && p.gw * p.s == p.iw && p.c_bytes < (1LL << 31) - (1 << 20)));
    if (tab_ok) {
      const size_t lds = ((size_t)p.N * (p.K + 8) + 8 * 16 * LDA + (size_t)p.gh * p.gw * ((p.mode == 0 ? 1 : p.nkt) + 1)) * sizeof(float);
      if (p.mask_bits) {                                     // byte ReLU mask: the 16-channel, four-tap data gradient only
        if (!(pl.nr == 4 && p.nkt == 4 && p.mode == 1 && p.cin == 16 && p.ld_in == 16 && lds <= 72 * 1024))
          return dry ? SEEDHIP_ERR_UNSUPPORTED : fail(SEEDHIP_ERR_UNSUPPORTED, "wsgemm: no byte-mask kernel for this data gradient");
        if (dry) return SEEDHIP_OK;
        if (lds > 64 * 1024)
          (void)hipFuncSetAttribute((const void*)ws_tab_kernel<4, 4, 1, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((ws_tab_kernel<4, 4, 1, 0, true>), dim3(pl.grid), dim3(512), lds, s, p);
        return check_launch("ws_tab_kernel(byte mask)");
      }
      if (dry) return SEEDHIP_OK;
      if (p.mask && pl.nr == 4 && p.nkt == 4 && p.mode == 1 && lds <= 72 * 1024) {
        if (lds > 64 * 1024)
          (void)hipFuncSetAttribute((const void*)ws_tab_kernel<4, 4, 1, 0, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((ws_tab_kernel<4, 4, 1, 0, false, true>), dim3(pl.grid), dim3(512), lds, s, p);
        return check_launch("ws_tab_kernel(mask a tile ahead)");
      }
#define SEEDHIP_WST(NT_, NKT_, MODE_)                                                                             \
      if (pl.nr == NT_ && p.nkt == NKT_ && p.mode == MODE_ && lds <= 72 * 1024) {                                 \
        if (lds > 64 * 1024)                                                                                      \
          (void)hipFuncSetAttribute((const void*)ws_tab_kernel<NT_, NKT_, MODE_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((ws_tab_kernel<NT_, NKT_, MODE_>), dim3(pl.grid), dim3(512), lds, s, p);               \
        return check_launch("ws_tab_kernel");                                                                     \
      }
      SEEDHIP_WST(2, 8, 0) SEEDHIP_WST(4, 8, 0) SEEDHIP_WST(4, 4, 1) SEEDHIP_WST(2, 4, 1) SEEDHIP_WST(4, 8, 1) SEEDHIP_WST(2, 8, 1)
#undef SEEDHIP_WST
    }
    // (the byte mask and the dry run end here: the kernels below take the fp32 mask only)
    if (p.mask_bits || dry)
      return dry ? (p.mask_bits ? SEEDHIP_ERR_UNSUPPORTED : SEEDHIP_OK)
                 : fail(SEEDHIP_ERR_UNSUPPORTED, "wsgemm: the byte ReLU mask needs ws_tab_kernel");
#define SEEDHIP_WSF(NR_, NKT_, MODE_)                                                                             \
    if (pl.nr == NR_ && p.nkt == NKT_ && p.mode == MODE_) {                                                       \
      if (pl.lds > 64 * 1024)                                                                                     \
        (void)hipFuncSetAttribute((const void*)ws_fast_kernel<NR_, NKT_, MODE_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.lds); \
      hipLaunchKernelGGL((ws_fast_kernel<NR_, NKT_, MODE_>), dim3(pl.grid), dim3(512), pl.lds, s, p);             \
      return check_launch("ws_fast_kernel");                                                                      \
    }
    SEEDHIP_WSF(2, 8, 0) SEEDHIP_WSF(4, 8, 0) SEEDHIP_WSF(4, 4, 1) SEEDHIP_WSF(2, 4, 1) SEEDHIP_WSF(4, 8, 1) SEEDHIP_WSF(2, 8, 1)
#undef SEEDHIP_WSF
  }
  if (p.mask_bits || dry)
    return dry ? (p.mask_bits ? SEEDHIP_ERR_UNSUPPORTED : SEEDHIP_OK)
               : fail(SEEDHIP_ERR_UNSUPPORTED, "wsgemm: the byte ReLU mask needs ws_tab_kernel");
#define SEEDHIP_WS(MR_, NR_, W_)                                                                                  \
  if (pl.mr == MR_ && pl.nr == NR_ && waves == W_) {                                                              \
    if (pl.lds > 64 * 1024)                                                                                       \
      (void)hipFuncSetAttribute((const void*)ws_kernel<MR_, NR_, W_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.lds); \
    hipLaunchKernelGGL((ws_kernel<MR_, NR_, W_>), dim3(pl.grid), dim3(64 * W_), pl.lds, s, p);                    \
    return check_launch("ws_kernel");                                                                             \
  }
  SEEDHIP_WS(1, 2, 8) SEEDHIP_WS(1, 4, 8)     // (r2 swept MR 1 / 2 / 4 and 4 / 8 / 16 waves: these won and are the only ones built)
#undef SEEDHIP_WS
  return fail(SEEDHIP_ERR_UNSUPPORTED, "wsgemm: no kernel for MR=%d NR=%d waves=%d", pl.mr, pl.nr, waves);
}

}  // namespace wsgemm
}  // namespace seedhip

// Weight gradient of small-kernel convolutions (3x3/1, 4x4/2, ...) with the input staged ONCE per spatial tile.
//
// Used by seedhip_conv2d_bwd_weight for the ResNet stacks of
// /root/reference/dmlab/networks.py:31-60 (TF autodiff of Conv2D wrt kernel / bias).
//
//   dW[ky,kx,c,co] = sum_{n,oy,ox} X[n, oy*s+ky-pad, ox*s+kx-pad, c] * dY[n,oy,ox,co],  db[co] = sum dY
//
// The implicit-GEMM formulation gathers every input element 9 times from global memory with
// per-element index arithmetic (VALU-bound: 4-16% of the fp32 MFMA peak on these shapes).  Here
// a workgroup walks (image, row-band) tiles: the input band + halo (after ReLU / u8->/255
// conversion) and the dY band are copied to LDS once with plain coalesced float4 copies, and
// all 9 taps read them from LDS with addresses of the form  pixel_base + per-lane tap offset --
// no div/mod per operand.  MFMA roles (v_mfma_f32_16x16x4_f32, 4 pixels reduced per issue):
//   A: rows = 16 consecutive rows of dW viewed as [9*cin, cout] (row = tap*cin + c),
//      lane (i, kq) supplies X[pixel 4g+kq shifted by the row's tap][c];
//   B: cols = 16 output channels, lane (kq, j) supplies dY[pixel 4g+kq][co].
// Fragments are x-interleaved where the layout allows (fp32 input, cin % 4 == 0; see gemm.h): a lane reads 4
// consecutive channels of its pixel as ONE ds_read_b128 and feeds 4 row-tiles (tile e of a quad owns rows 4*lane + e),
// and NT consecutive output channels of dY as one b64 / b128 -- 4 LDS reads per 18 MFMAs for a 3x3 layer with 32
// output channels instead of 11 (every ds_read costs the fp32 matrix pipe ~14 cycles).
// The dW rows are split over MSPLIT waves (each MTW row-tiles), pixel groups over the other
// 4/MSPLIT waves; accumulators stay in registers across all tiles of the persistent workgroup,
// then waves are summed through LDS in a fixed order and ONE partial slice per workgroup goes
// to the workspace (reduced by reduce_slices: deterministic, no atomics).
#pragma once
#include "common.h"
#include "igemm.h"
#include "conv_launch.h"
#include <cstdlib>
#include "../../include/seedhip.h"

namespace seedhip {
namespace halo {

typedef float f32x4_t __attribute__((ext_vector_type(4)));

struct WgradParams {
  const void* in; int in_dtype, in_relu;
  const float* dy;
  float* partial_w;            // [grid][rows*cout]
  float* partial_b;            // [grid][cout] or null
  int n_img, ih, iw, cin, oh, ow, cout, ld_in, ld_out, pad_t, pad_l, kh, kw, stride;
  int TH;                      // output rows per tile
  int bands;                   // ceil(oh / TH)
  int ntiles;                  // n_img * bands
  int rows;                    // kh * kw * cin
  int xs;                      // LDS pixel stride of the X tile (floats)
  int ilv;                     // x-interleaved fragments (fp32 input, cin % 4 == 0)
  int twp;                     // tile width incl. halo = (ow - 1) * stride + kw
  int thp;                     // tile rows incl. halo for a full band = (TH - 1) * stride + kh
  FastDiv d_ow;
};

template <int MTW, int NT, int MSPLIT, bool ILV>         // ILV = p.ilv, compile-time: no branches inside the pixel loop
__global__ void __launch_bounds__(256)
halo_wgrad_kernel(const WgradParams p) {
  constexpr int PP = 4 / MSPLIT;                      // pixel partitions
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int x_floats = (p.thp * p.twp * p.xs + 3) & ~3;
  float* xs_lds = smem;
  float* dy_lds = smem + x_floats;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kq = lane >> 4, i = lane & 15;
  const int ms = wave % MSPLIT, pp = wave / MSPLIT;
  const int coutp = NT * 16;

  // per-lane LDS offset of each of this wave's dW rows (tap shift + channel).  Interleaved mode (p.ilv): the first
  // 4*NQ row-tiles form quads, tile e of quad Q owns rows base + 64 Q + 4 i + e and lane i reads them as one float4;
  // lane_off[4Q] is that quad's offset.  The remaining tiles (and everything when !p.ilv) own rows base + 16 mt + i.
  constexpr int NQ = MTW / 4;
  constexpr bool ilv = ILV;
  int lane_off[MTW];
#pragma unroll
  for (int mt = 0; mt < MTW; ++mt) {
    const bool quad = ilv && mt < 4 * NQ;
    const int R = quad ? (ms * MTW + (mt & ~3)) * 16 + 4 * i + (mt & 3) : (ms * MTW + mt) * 16 + i;
    const int Rc = R < p.rows ? R : 0;
    const int tap = Rc / p.cin, c = Rc - tap * p.cin;
    lane_off[mt] = ((tap / p.kw) * p.twp + (tap % p.kw)) * p.xs + c;
  }
  f32x4_t acc[MTW][NT];
#pragma unroll
  for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  float bsum[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) bsum[nt] = 0.f;

  // Software pipeline over tiles: the global loads of tile i+1 (float4 per lane, kept in registers)
  // are in flight while tile i is reduced out of LDS.
  constexpr int kXV = 7, kDV = NT <= 2 ? 6 : 3;        // float4 registers per thread for the X / dY tile (2 workgroups per
                                                       // CU: the budget is 256 registers, 144 of them accumulators at NT = 4)
  const bool vec = p.in_dtype == 0 && (p.cin & 3) == 0 && (p.ld_in & 3) == 0;
  float4 xr[kXV], dr[kDV];
  auto band_of = [&](int tile, int& n, int& y0, int& th) {
    n = tile / p.bands;
    const int band = tile - n * p.bands;
    y0 = band * p.TH;
    th = (y0 + p.TH <= p.oh) ? p.TH : p.oh - y0;
  };
  // A thread's staging vectors have the same tile coordinates in every tile: decode them once (the integer divisions
  // used to run in every load and every store -- cfg3's SQ counters, r02c: ~5 VALU per MFMA, matrix pipe 41 % busy).
  int st_row[kXV], st_goff[kXV], st_lds[kXV];          // tile row; image offset relative to the band's first input row
  int dy_goff[kDV];                                    // (-1: column outside the map); LDS offset; dY offset in the band
  {
    const int c4 = p.cin >> 2, per_row = p.twp * (c4 > 0 ? c4 : 1), rowf = p.twp * p.xs;
#pragma unroll
    for (int u = 0; u < kXV; ++u) {
      const int v = tid + u * 256;
      const int r = v / per_row, rem = v - r * per_row;
      const int xcol = rem / (c4 > 0 ? c4 : 1), cq = rem - xcol * c4;
      const int ix = xcol - p.pad_l;
      st_row[u] = r;
      st_goff[u] = (ix >= 0 && ix < p.iw) ? ((r - p.pad_t) * p.iw + ix) * p.ld_in + 4 * cq : -1;
      st_lds[u] = r * rowf + xcol * p.xs + 4 * cq;
    }
    const int d4 = coutp >> 2;
#pragma unroll
    for (int u = 0; u < kDV; ++u) {
      const int v = tid + u * 256;
      const int pix = v / d4, cq = v - pix * d4;
      dy_goff[u] = pix * p.ld_out + 4 * cq;
    }
  }
  // The tile prefetch goes through BUFFER loads whose offset is 0x80000000 for cells outside the image / band (the
  // hardware returns zeros, the tensors are < 2 GB: `bufok`): one unconditional load per register.  The conditional
  // global loads they replace (`val = 0; if (inside) val = load;`) came out of the compiler as predicated blocks with
  // register copies behind the load -- and an s_waitcnt vmcnt(0) in front of the copies: the whole prefetch was waited
  // for where it was issued, in front of the MFMA loop it was meant to hide under.
  const long long x_bytes = (long long)p.n_img * p.ih * p.iw * p.ld_in * 4, d_bytes = (long long)p.n_img * p.oh * p.ow * p.ld_out * 4;
  const bool bufok = vec && x_bytes < (1LL << 31) && d_bytes < (1LL << 31);
  auto mk_view = [](const void* q, long long bytes) {
    const uint64_t ab = reinterpret_cast<uint64_t>(q);
    const uint64_t sb = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((unsigned)(ab >> 32)) << 32) |
                        (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((unsigned)ab);
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(sb), 0, __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000);
  };
  const __amdgpu_buffer_rsrc_t x_view = mk_view(p.in, bufok ? x_bytes : 0), d_view = mk_view(p.dy, bufok ? d_bytes : 0);
  auto view_ld = [](const __amdgpu_buffer_rsrc_t& r, unsigned byte_off) {
    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
    const u32x4_t t = __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0);
    return make_float4(__uint_as_float(t[0]), __uint_as_float(t[1]), __uint_as_float(t[2]), __uint_as_float(t[3]));
  };
  auto load_tile = [&](int tile) {
    int n, y0, th; band_of(tile, n, y0, th);
    if (bufok) {
      const int nrows = (th - 1) * p.stride + p.kh;
      const int iy0 = y0 * p.stride - p.pad_t;
      const unsigned xb = (unsigned)((((long long)n * p.ih + y0 * p.stride) * p.iw * p.ld_in) * 4);
#pragma unroll
      for (int u = 0; u < kXV; ++u) {
        const int iy = iy0 + st_row[u];
        const bool ok = st_row[u] < nrows && st_goff[u] != -1 && iy >= 0 && iy < p.ih;
        xr[u] = view_ld(x_view, ok ? xb + (unsigned)(st_goff[u] * 4) : 0x80000000u);
      }
      const int nd = th * p.ow * (coutp >> 2);
      const unsigned db = (unsigned)((((long long)n * p.oh + y0) * p.ow * p.ld_out) * 4);
#pragma unroll
      for (int u = 0; u < kDV; ++u)
        dr[u] = view_ld(d_view, tid + u * 256 < nd ? db + (unsigned)(dy_goff[u] * 4) : 0x80000000u);
      return;
    }
    if (vec) {
      const int nrows = (th - 1) * p.stride + p.kh;
      const int iy0 = y0 * p.stride - p.pad_t;
      const float* xsrc = (const float*)p.in + ((long long)n * p.ih + y0 * p.stride) * p.iw * p.ld_in;
#pragma unroll
      for (int u = 0; u < kXV; ++u) {
        float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
        const int iy = iy0 + st_row[u];
        if (st_row[u] < nrows && st_goff[u] != -1 && iy >= 0 && iy < p.ih)
          val = *reinterpret_cast<const float4*>(xsrc + st_goff[u]);
        xr[u] = val;
      }
    }
    const int c4 = coutp >> 2;
    const int nd = th * p.ow * c4;
    const float* src = p.dy + ((long long)n * p.oh + y0) * p.ow * p.ld_out;
#pragma unroll
    for (int u = 0; u < kDV; ++u) {
      const int v = tid + u * 256;
      float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
      if (v < nd) val = *reinterpret_cast<const float4*>(src + dy_goff[u]);
      dr[u] = val;
    }
  };
  auto store_tile = [&](int tile) {
    int n, y0, th; band_of(tile, n, y0, th);
    const int rowf = p.twp * p.xs;                     // floats per LDS row
    const int nrows = (th - 1) * p.stride + p.kh;
    if (vec) {
#pragma unroll
      for (int u = 0; u < kXV; ++u) {
        if (st_row[u] < nrows) {
          float4 val = xr[u];
          if (p.in_relu) { val.x = fmaxf(val.x, 0.f); val.y = fmaxf(val.y, 0.f); val.z = fmaxf(val.z, 0.f); val.w = fmaxf(val.w, 0.f); }
          *reinterpret_cast<float4*>(xs_lds + st_lds[u]) = val;
        }
      }
    } else {                                           // first layer (u8 / odd channel counts): direct, synchronous
      const int per_row = p.twp * p.cin;
      for (int v = tid; v < nrows * per_row; v += 256) {
        const int r = v / per_row, rem = v - r * per_row;
        const int xcol = rem / p.cin, c = rem - xcol * p.cin;
        const int iy = y0 * p.stride - p.pad_t + r, ix = xcol - p.pad_l;
        float val = 0.f;
        if (iy >= 0 && iy < p.ih && ix >= 0 && ix < p.iw) {
          const long long off = (((long long)n * p.ih + iy) * p.iw + ix) * p.ld_in + c;
          val = p.in_dtype == 1 ? (float)((const uint8_t*)p.in)[off] / 255.0f : ((const float*)p.in)[off];
          if (p.in_relu) val = fmaxf(val, 0.f);
        }
        xs_lds[r * rowf + xcol * p.xs + c] = val;
      }
    }
    const int c4 = coutp >> 2;
    const int nd = th * p.ow * c4;
#pragma unroll
    for (int u = 0; u < kDV; ++u) {
      const int v = tid + u * 256;
      if (v < nd) *reinterpret_cast<float4*>(dy_lds + 4 * v) = dr[u];
    }
  };

  if ((int)blockIdx.x < p.ntiles) load_tile(blockIdx.x);
  for (int tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
    int n, y0, th; band_of(tile, n, y0, th);
    __syncthreads();                                   // previous tile fully consumed
    store_tile(tile);
    __syncthreads();
    if (tile + (int)gridDim.x < p.ntiles) load_tile(tile + gridDim.x);
    // ---- MFMA over this wave's pixel groups (a two-register-set prefetch of the next group's operands was measured
    //      slower, r02d: 32->32 @18x24 0.46 -> 0.61 ms) ----
    const int npix = th * p.ow;
    const int G = (npix + 3) >> 2;
    auto load_ab = [&](int g, float (&a)[MTW], float (&b)[NT]) {
      const int pix = 4 * g + kq;
      const bool pv = pix < npix;                      // tail group of a band whose pixel count is not 4k
      uint32_t py, px;
      p.d_ow.divmod((uint32_t)(pv ? pix : 0), py, px);
      const float* xb = xs_lds + ((int)py * p.stride * p.twp + (int)px * p.stride) * p.xs;
      if (ilv) {                                         // NT consecutive channels NT*i .. : one read
        if constexpr (NT == 1) {
          b[0] = pv ? dy_lds[pix * coutp + i] : 0.f;
        } else {
          typedef float bvec_t __attribute__((ext_vector_type(NT)));
          const bvec_t bv = *reinterpret_cast<const bvec_t*>(dy_lds + (pv ? pix : 0) * coutp + NT * i);
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) b[nt] = pv ? bv[nt] : 0.f;
        }
      } else {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) b[nt] = pv ? dy_lds[pix * coutp + nt * 16 + i] : 0.f;
      }
      // rows beyond p.rows (partial last tile of the first layer) read row 0's operand: finite values into
      // accumulators that are never written out -- no predicate, no exec juggling around the LDS reads
      if (ilv) {
#pragma unroll
        for (int Q = 0; Q < NQ; ++Q) {
          const float4 v = *reinterpret_cast<const float4*>(xb + lane_off[4 * Q]);
          a[4 * Q] = v.x; a[4 * Q + 1] = v.y; a[4 * Q + 2] = v.z; a[4 * Q + 3] = v.w;
        }
#pragma unroll
        for (int mt = 4 * NQ; mt < MTW; ++mt) a[mt] = xb[lane_off[mt]];
      } else {
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt) a[mt] = xb[lane_off[mt]];
      }
    };
    auto mma = [&](const float (&a)[MTW], const float (&b)[NT]) {
      if (ms == 0) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bsum[nt] += b[nt];
      }
#pragma unroll
      for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt], b[nt], acc[mt][nt], 0, 0, 0);
    };
    for (int g = pp; g < G; g += PP) {
      float a[MTW], b[NT];
      load_ab(g, a, b);
      mma(a, b);
    }
  }

  // ---- reduce the PP pixel partitions through LDS (fixed order), write one slice ----
  __syncthreads();
  float* red = smem;                                   // [MSPLIT*MTW*16 rows][coutp]
  for (int q = 0; q < PP; ++q) {
    if (pp == q) {
#pragma unroll
      for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int lr = 4 * kq + r;                                  // MFMA row of the tile = the A lane index
            const bool quad = ilv && mt < 4 * NQ;
            const int row = quad ? (ms * MTW + (mt & ~3)) * 16 + 4 * lr + (mt & 3) : (ms * MTW + mt) * 16 + lr;
            const int col = ilv ? NT * i + nt : nt * 16 + i;
            float* d = red + row * coutp + col;
            *d = (q == 0) ? acc[mt][nt][r] : *d + acc[mt][nt][r];
          }
    }
    __syncthreads();
  }
  float* pw = p.partial_w + (long long)blockIdx.x * p.rows * p.cout;
  for (int v = tid; v < p.rows * coutp; v += 256) {
    const int row = v / coutp, co = v - row * coutp;
    if (co < p.cout) pw[row * p.cout + co] = red[v];
  }
  if (p.partial_b) {
    __syncthreads();
    float* rb = smem;                                  // [4 waves][coutp]
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      float s = bsum[nt];
      s += __shfl_xor(s, 16, 64);
      s += __shfl_xor(s, 32, 64);
      if (lane < 16) rb[wave * coutp + (ilv ? NT * lane + nt : nt * 16 + lane)] = s;
    }
    __syncthreads();
    if (tid < p.cout) {
      float s = 0.f;
      for (int w = 0; w < 4; ++w) s += rb[w * coutp + tid];     // waves with ms != 0 hold zeros
      p.partial_b[(long long)blockIdx.x * p.cout + tid] = s;
    }
  }
}

// ---- host side ------------------------------------------------------------------------------ //
struct WgradPlan { bool ok; int MTW, NT, MSPLIT, TH, grid; size_t lds; size_t ws_bytes; };

inline WgradPlan plan_wgrad(const seedhip_conv_geom* g) {
  WgradPlan pl; memset(&pl, 0, sizeof(pl));
  if (g->kh > 4 || g->kw > 4 || g->stride > 2 || g->kh < g->stride || g->kw < g->stride) return pl;
  if (g->cout % 16 != 0 || g->cout > 64 || g->ld_out % 4 != 0 || g->oh * g->ow < 16) return pl;
  const int rows = g->kh * g->kw * g->cin;
  const int MT = (rows + 15) / 16;
  // row-tiles per wave: 2 (tiny first layer), 8 or 9; split the rows over 1, 2 or 4 waves
  int msplit = 1, mtw = 0;
  if (MT <= 2) { mtw = 2; }
  else {
    for (msplit = 1; msplit <= 4; msplit *= 2) {
      const int per = (MT + msplit - 1) / msplit;
      if (per <= 9) { mtw = per <= 8 ? 8 : 9; break; }
    }
    if (!mtw) return pl;
  }
  pl.MSPLIT = msplit; pl.MTW = mtw; pl.NT = g->cout / 16;
  if (pl.MTW == 2 && pl.NT != 1) return pl;
  if (pl.NT == 3) return pl;
  if (pl.MTW * pl.NT > 36) return pl;                  // accumulator registers
  const int twp = (g->ow - 1) * g->stride + g->kw;
  int th = 256 / g->ow; if (th < 1) th = 1; if (th > g->oh) th = g->oh;
  size_t red_b = (size_t)msplit * pl.MTW * 16 * g->cout * 4;
  auto need_of = [&](int t, bool& fits_regs) {
    const size_t x_b = (size_t)((t - 1) * g->stride + g->kh) * twp * g->cin * 4, dy_b = (size_t)t * g->ow * g->cout * 4;
    size_t need = x_b + dy_b + 16; if (need < red_b) need = red_b;
    fits_regs = (g->cin % 4 != 0 || x_b <= 7 * 256 * 16) && dy_b <= (size_t)(pl.NT <= 2 ? 6 : 3) * 256 * 16;   // kXV / kDV
    return need;
  };
  for (;; --th) {
    bool fits_regs;
    const size_t need = need_of(th, fits_regs);
    if ((need <= 64 * 1024 && fits_regs) || th == 1) { if (!fits_regs) return pl; break; }
  }
  // th is the tallest band that fits; among the heights down to 70 % of it take the one that wastes the fewest rows in
  // the last band (36 rows: 9 x 4, not 7 x 5 + 1 -- measured 0.55 vs 0.60 ms; 9 rows: one band of 9, not 8 + 1)
  {
    int best = th, best_waste = ((g->oh + th - 1) / th) * th - g->oh;
    for (int t = th - 1; t >= 1 && 10 * t >= 7 * th; --t) {
      const int waste = ((g->oh + t - 1) / t) * t - g->oh;
      if (waste < best_waste) { best = t; best_waste = waste; }
    }
    th = best;
    bool fits_regs;
    pl.lds = need_of(th, fits_regs);
  }
  if (pl.lds > 150 * 1024) return pl;
  pl.TH = th;
  const int bands = (g->oh + th - 1) / th;
  const long long ntiles = (long long)g->n_img * bands;
  constexpr int cap = 2;
  int per_cu = (int)((160 * 1024) / pl.lds); if (per_cu > cap) per_cu = cap; if (per_cu < 1) per_cu = 1;
  const long long mg = 256LL * per_cu;
  pl.grid = (int)(ntiles < mg ? ntiles : mg);
  pl.ws_bytes = (size_t)pl.grid * ((size_t)rows * g->cout + g->cout) * sizeof(float);
  pl.ok = true;
  return pl;
}

inline int launch_wgrad(const seedhip_conv_geom* g, const WgradPlan& pl, const void* in, int in_dtype, int in_relu,
                        const float* dy, float* dw, float* dbias, void* workspace, hipStream_t s) {
  WgradParams p;
  p.in = in; p.in_dtype = in_dtype; p.in_relu = in_relu; p.dy = dy;
  p.n_img = g->n_img; p.ih = g->ih; p.iw = g->iw; p.cin = g->cin; p.oh = g->oh; p.ow = g->ow; p.cout = g->cout;
  p.ld_in = g->ld_in; p.ld_out = g->ld_out; p.pad_t = g->pad_t; p.pad_l = g->pad_l;
  p.kh = g->kh; p.kw = g->kw; p.stride = g->stride;
  p.TH = pl.TH; p.bands = (g->oh + pl.TH - 1) / pl.TH; p.ntiles = g->n_img * p.bands;
  p.rows = g->kh * g->kw * g->cin; p.xs = g->cin;
  p.ilv = (in_dtype == 0 && g->cin % 4 == 0 && g->ld_in % 4 == 0) ? 1 : 0;
  p.twp = (g->ow - 1) * g->stride + g->kw; p.thp = (pl.TH - 1) * g->stride + g->kh;
  p.d_ow.init(g->ow);
  p.partial_w = (float*)workspace;
  p.partial_b = dbias ? (float*)workspace + (size_t)pl.grid * p.rows * g->cout : nullptr;
#define SEEDHIP_HALO_LAUNCH(MTW_, NT_, MS_)                                                                        \
  do {                                                                                                             \
    if (pl.lds > 64 * 1024)                                                                                        \
      (void)hipFuncSetAttribute(p.ilv ? (const void*)halo_wgrad_kernel<MTW_, NT_, MS_, true>                       \
                                      : (const void*)halo_wgrad_kernel<MTW_, NT_, MS_, false>,                     \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.lds);                          \
    if (p.ilv) hipLaunchKernelGGL((halo_wgrad_kernel<MTW_, NT_, MS_, true>), dim3(pl.grid), dim3(256), pl.lds, s, p);  \
    else hipLaunchKernelGGL((halo_wgrad_kernel<MTW_, NT_, MS_, false>), dim3(pl.grid), dim3(256), pl.lds, s, p);   \
  } while (0)
#define SEEDHIP_HALO_CASE(MTW_, NT_, MS_) if (pl.MTW == MTW_ && pl.NT == NT_ && pl.MSPLIT == MS_) { SEEDHIP_HALO_LAUNCH(MTW_, NT_, MS_); launched = true; }
  bool launched = false;
  SEEDHIP_HALO_CASE(2, 1, 1)
  SEEDHIP_HALO_CASE(9, 1, 1) SEEDHIP_HALO_CASE(9, 2, 1) SEEDHIP_HALO_CASE(9, 1, 2) SEEDHIP_HALO_CASE(9, 2, 2)
  SEEDHIP_HALO_CASE(9, 4, 4) SEEDHIP_HALO_CASE(9, 4, 2) SEEDHIP_HALO_CASE(9, 2, 4) SEEDHIP_HALO_CASE(9, 1, 4)
  SEEDHIP_HALO_CASE(8, 1, 1) SEEDHIP_HALO_CASE(8, 2, 1) SEEDHIP_HALO_CASE(8, 1, 2) SEEDHIP_HALO_CASE(8, 2, 2)
  SEEDHIP_HALO_CASE(8, 4, 4) SEEDHIP_HALO_CASE(8, 4, 2) SEEDHIP_HALO_CASE(8, 2, 4) SEEDHIP_HALO_CASE(8, 1, 4)
  SEEDHIP_HALO_CASE(8, 4, 1) SEEDHIP_HALO_CASE(9, 4, 1)
#undef SEEDHIP_HALO_CASE
  if (!launched) return fail(SEEDHIP_ERR_UNSUPPORTED, "halo_wgrad: no kernel for MTW=%d NT=%d MSPLIT=%d", pl.MTW, pl.NT, pl.MSPLIT);
#undef SEEDHIP_HALO_LAUNCH
  int rc = check_launch("halo_wgrad_kernel"); if (rc) return rc;
  reduce_slices2(p.partial_w, (long long)p.rows * g->cout, dw, p.partial_b, g->cout, dbias, pl.grid, s);
  return check_launch("halo_wgrad reduce");
}

}  // namespace halo
}  // namespace seedhip

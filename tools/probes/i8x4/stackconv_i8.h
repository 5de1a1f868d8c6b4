// NOT part of the library.  The first Atari conv's forward on the int8 matrix pipe ("i8x4"), built and measured in round 6
// and left out: 148 us against the bf16x3 kernel's 141 us on the same box (cfg2, one launch of 10 752 frames), i.e. 80
// MFMAs + 40 LDS fragment reads per new frame and wave instead of 120 + 88 bought nothing -- the kernel is bound by the
// SIMD's instruction issue (the int32 -> fp32 epilogue of this formulation costs as many VALU instructions as the u8 ->
// bf16 staging it removes), not by MFMA count or LDS volume (profiles/r06_stackconv_ab.txt).  Its first build also failed
// test_stack_conv_parity; it was not debugged further.  Kept as the record of the A/B (it was spliced into
// csrc/stackconv.hip in front of stackconv_rows_kernel; the launcher hunk is at the end).
// ------------------------------------------------------------------------------------ //
// "i8x4" forward (r6).  The bf16x3 kernel above sits at a third of both of its roofs because per new
// frame it issues 120 MFMAs and 88 LDS fragment reads per wave, for a layer whose one operand is EIGHT-bit data.  Here
// the pixels stay bytes and the products run on the int8 matrix pipe (v_mfma_i32_16x16x64_i8: 64 k per instruction at
// the bf16 rate for 32): ONE instruction reduces the whole 8 x 8 window of a stack channel.
//   * W / 255 of an output channel is put on a per-channel binary grid: Wq = rint(w / 255 * 2^s), s such that
//     max |Wq| is in [2^29, 2^30), and Wq is written in FOUR balanced base-256 digits (signed bytes): the grid step is
//     2^-30 of the channel's largest weight -- finer than the fp32 representation of every weight within 2^-6 of that
//     maximum, and the only approximation of the kernel: pixels are exact, every product and every sum is an EXACT
//     integer (no accumulation rounding at all, in any order), one int32 accumulator per digit plane (|acc| < 2^24);
//   * x = (x - 128) + 128: the ring holds the bytes XOR 0x80 (signed), the 128 * sum of the digits of the VALID stack
//     channels is the accumulators' initial value (C operand of the first MFMA: no instruction);
//   * out = ldexp(acc3 2^24 + acc2 2^16 + acc1 2^8 + acc0, -s) + bias: four exact conversions, three fp32 FMAs.
// Per new frame and wave: 80 MFMAs (was 120), 40 LDS fragment reads (was 88: the weights' lo parts are gone, all four
// planes live in 64 registers), no u8 -> bf16 conversion, half the ring.  Accuracy against an fp64 evaluation:
// tests/test_gpu_kernels.py::test_stack_conv_fwd_fp32_accuracy (the same 2 x torch-fp32 bound as before; measured below
// torch's own error).  A channel with a non-finite weight returns NaN in every pixel (fp32 would return NaN or inf).
// ------------------------------------------------------------------------------------ //
typedef int i32x4_t __attribute__((ext_vector_type(4)));
constexpr int kRingI8 = kSlots * kBandBytes;             // 6720 B per wave: raw (sign-flipped) bytes
constexpr int kI8Tab = 4 * 4 * 16 * 4;                   // acc init [nv][plane][16 channels] int32
constexpr int kI8Lds = kWaves * kRingI8 + kI8Tab + 16 * 4 + 16 * 4;   // + ldexp exponents [16] + bias (NaN-poisoned) [16]

// The workgroup quantizes its slice TOGETHER (every wave needs the same 64 registers of digits; computed per lane they
// were 128 LDS reads + 64 fp64 conversions each and spilled 180 registers): weights -> LDS, per-channel maximum,
// 4 096 (k, channel) items over the 320 threads -> digit bytes in the MFMA A-operand image
//   img[(c * 4 + plane) * 64 + lane] (16 bytes: byte e = window row 2 (lane >> 4) + (e >> 3), column e & 7; channel lane & 15),
// digit sums -> the accumulators' initial values.  `smem` = the rings' bytes, not yet in use (32 KB of their 33.6).
__device__ __forceinline__ void i8_quantize_slice(const Params& p, int co0, unsigned char* smem, int* tab, int* sexp_lds,
                                                  float* bias_lds, i32x4_t (&wq)[4][4]) {
  const int tid = threadIdx.x, lane = tid & 63;
  float* wl = reinterpret_cast<float*>(smem);                          // [256 k][16]
  unsigned char* img = smem + 16384;                                   // [c][plane][lane][16]
  __shared__ float s_part[16][16];
  __shared__ int s_exp[16], s_bad[16];
  for (int idx = tid; idx < 1024; idx += kThreads)
    *reinterpret_cast<float4*>(wl + 4 * idx) = *reinterpret_cast<const float4*>(p.w + (idx >> 2) * p.cout + co0 + 4 * (idx & 3));
  __syncthreads();
  if (tid < 256) {                                                     // (channel, sixteenth of k): partial maxima
    const int co = tid & 15, part = tid >> 4;
    float amax = 0.f;
    bool bad = false;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float a = __builtin_fabsf(wl[(part * 16 + i) * 16 + co]);
      bad = bad || !(a <= 3.402823466e38f);
      amax = a > amax ? a : amax;
    }
    s_part[part][co] = bad ? __builtin_nanf("") : amax;
  }
  __syncthreads();
  if (tid < 16) {
    float amax = 0.f;
    bool bad = false;
#pragma unroll
    for (int i = 0; i < 16; ++i) { const float a = s_part[i][tid]; bad = bad || !(a == a); amax = a > amax ? a : amax; }
    const bool live = amax > 0.f && !bad;
    // amax / 255 = m 2^ex with m in [0.5, 1): s = 30 - ex puts the largest |Wq| into [2^29, 2^30)
    const int sexp = live ? 30 - __builtin_amdgcn_frexp_exp((double)amax / 255.0) : 0;
    s_exp[tid] = live ? sexp : -100000;                                // (marker: every digit 0)
    s_bad[tid] = bad;
    sexp_lds[tid] = -sexp;
    bias_lds[tid] = bad ? __builtin_nanf("") : (p.bias ? p.bias[co0 + tid] : 0.f);
  }
  __syncthreads();
  for (int it = tid; it < 4096; it += kThreads) {
    const int co = it & 15, k = it >> 4;                               // k = (ky * 8 + kx) * 4 + c
    const int c = k & 3, kx = (k >> 2) & 7, ky = k >> 5;
    const int sexp = s_exp[co];
    int q = 0;
    if (sexp > -100000) q = (int)__builtin_rint((double)wl[k * 16 + co] * __builtin_ldexp(1.0 / 255.0, sexp));   // |q| <= 2^30
    unsigned char* dst = img + ((c * 4) * 64 + (ky >> 1) * 16 + co) * 16 + (ky & 1) * 8 + kx;
#pragma unroll
    for (int pl = 0; pl < 4; ++pl) {
      const int d = pl == 3 ? q : (((q + 128) & 255) - 128);           // balanced digit in [-128, 127]
      q = (q - d) >> 8;
      dst[pl * 64 * 16] = (unsigned char)d;
    }
  }
  __syncthreads();
  if (tid < 64) {                                                      // (plane, channel): 128 x digit sums, cumulative over c
    const int co = tid & 15, pl = tid >> 4;
    int run = 0;
    for (int c = 0; c < 4; ++c) {
      int sum = 0;
      for (int kq = 0; kq < 4; ++kq) {
        const uint4 v = *reinterpret_cast<const uint4*>(img + ((c * 4 + pl) * 64 + kq * 16 + co) * 16);
        const unsigned wds[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4)
#pragma unroll
          for (int b = 0; b < 4; ++b) sum += (int)(signed char)((wds[q4] >> (8 * b)) & 0xFFu);
      }
      run += 128 * sum;
      tab[(c * 4 + pl) * 16 + co] = run;                               // nv = c + 1
    }
  }
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int pl = 0; pl < 4; ++pl)
      wq[c][pl] = *reinterpret_cast<const i32x4_t*>(img + ((c * 4 + pl) * 64 + lane) * 16);
  __syncthreads();                                                     // image consumed, tables written: the rings may be filled
}

// sign-flipped band -> the wave's ring slot (two 16-byte stores per lane)
__device__ __forceinline__ void band_store_i8(unsigned char* slot, const BandPrefetch& r, int lane) {
  uint4* dst = reinterpret_cast<uint4*>(slot);
  const unsigned f = 0x80808080u;
  dst[lane] = make_uint4(r.v0.x ^ f, r.v0.y ^ f, r.v0.z ^ f, r.v0.w ^ f);
  if (lane + 64 < kBandVec) dst[lane + 64] = make_uint4(r.v1.x ^ f, r.v1.y ^ f, r.v1.z ^ f, r.v1.w ^ f);
}

template <bool BITS, bool RELU>
__global__ void __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(3, 3)))
stackconv_fwd_i8_kernel(const Params p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  unsigned char* myring = smem + wave * kRingI8;
  int* tab = reinterpret_cast<int*>(smem + kWaves * kRingI8);          // [nv - 1][plane][16]
  int* sexp_lds = tab + 4 * 4 * 16;
  float* bias_lds = reinterpret_cast<float*>(sexp_lds + 16);
  const int kq = lane >> 4, j = lane & 15;
  const int co0 = blockIdx.z * 16;

  i32x4_t wq[4][4];
  i8_quantize_slice(p, co0, smem, tab, sexp_lds, bias_lds, wq);
  int aoff[kMT];                                      // byte offset of (tile m, pixel j, window rows 2 kq, 2 kq + 1) in a band slot
#pragma unroll
  for (int m = 0; m < kMT; ++m) {
    const int pix = m * 16 + j;
    const int oy = pix / kOW, ox = pix - oy * kOW;
    aoff[m] = (oy * 4 + 2 * kq) * kIW + ox * 4;
  }
  const i32x4_t nsexp = *reinterpret_cast<const i32x4_t*>(sexp_lds + 4 * kq);   // (no workgroup barrier from here on)
  const f32x4_t bias4 = *reinterpret_cast<const f32x4_t*>(bias_lds + 4 * kq);

  const __amdgpu_buffer_rsrc_t fview = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint8_t*>(p.frames_ext), 0, (int)((long long)(3 + p.T1) * p.B * p.fsz), 0x00020000);
  const __amdgpu_buffer_rsrc_t oview = __builtin_amdgcn_make_buffer_rsrc(
      p.out, 0, (int)((long long)p.T1 * p.B * 400 * p.ld_out * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t bview = __builtin_amdgcn_make_buffer_rsrc(
      p.relu_bits, 0, BITS ? (int)((long long)p.T1 * p.B * 400 * p.ld_out / 4) : 0, 0x00020000);
  const unsigned fv0 = 16u * (unsigned)lane, fv1 = lane + 64 < kBandVec ? 16u * (unsigned)(lane + 64) : 0x80000000u;
  const unsigned ov0 = (unsigned)(((wave * 80 + j) * p.ld_out + co0 + 4 * kq) * 4);
  const int tile_bytes = 16 * p.ld_out * 4;
  auto band_at = [&](int e, int b) {                   // ext row e of column b, this wave's band
    const unsigned so = __builtin_amdgcn_readfirstlane((unsigned)((e * p.B + b) * p.fsz + wave * 16 * kIW));
    BandPrefetch r;
    r.v0 = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(fview, fv0, so, 0));
    r.v1 = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(fview, fv1, so, 0));
    return r;
  };

  for (int item = blockIdx.x; item < p.items; item += gridDim.x) {
    const int b = item % p.B, chunk = item / p.B;
    const int t0 = chunk * p.spc;
    const int t1 = (t0 + p.spc < p.T1) ? t0 + p.spc : p.T1;
    {
      BandPrefetch f[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) f[e] = band_at(t0 + e, b);
#pragma unroll
      for (int e = 0; e < 4; ++e) band_store_i8(myring + ((t0 + e) % kSlots) * kBandBytes, f[e], lane);
      wave_lds_fence();
    }
    for (int t = t0; t < t1; ++t) {
      const bool more = t + 1 < t1;
      const int nv = nvalid_at(p.nvalid, (long long)t * p.B + b);
      BandPrefetch pf;
      if (more) pf = band_at(t + 4, b);
      i32x4_t init[4];
#pragma unroll
      for (int pl = 0; pl < 4; ++pl) init[pl] = *reinterpret_cast<const i32x4_t*>(tab + ((nv - 1) * 4 + pl) * 16 + 4 * kq);
      const unsigned oso = __builtin_amdgcn_readfirstlane((unsigned)((t * p.B + b) * 400 * p.ld_out) * 4u);
      auto tile = [&](int m, auto full) {
        constexpr bool kFull = decltype(full)::value;
        i32x4_t acc[4] = {init[0], init[1], init[2], init[3]};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          if (!kFull && c >= nv) break;
          typedef __attribute__((address_space(3))) const volatile unsigned lds_cv32_t;
          lds_cv32_t* src = (lds_cv32_t*)(myring + ((t + 3 - c) % kSlots) * kBandBytes + aoff[m]);
          const i32x4_t xf = {(int)src[0], (int)src[1], (int)src[kIW / 4], (int)src[kIW / 4 + 1]};
#pragma unroll
          for (int pl = 0; pl < 4; ++pl) acc[pl] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wq[c][pl], xf, acc[pl], 0, 0, 0);
        }
        f32x4_t v;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float f = __builtin_fmaf((float)acc[1][r], 256.0f, (float)acc[0][r]);
          f = __builtin_fmaf((float)acc[2][r], 65536.0f, f);
          f = __builtin_fmaf((float)acc[3][r], 16777216.0f, f);
          v[r] = ldexpf(f, nsexp[r]) + bias4[r];
          if (RELU) v[r] = __builtin_amdgcn_fmed3f(v[r], 0.f, __builtin_inff());
        }
        asm volatile("" : "+v"(v));                    // finished before its store is issued (see the bf16x3 kernel)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(su32x4_t, v), oview, ov0, oso + m * tile_bytes, 0);
        if (BITS) {
          const su32x4_t bu = __builtin_bit_cast(su32x4_t, v);
          const unsigned m01 = ((bu[1] < 1u ? bu[1] : 1u) << 1) | (bu[0] < 1u ? bu[0] : 1u);
          const unsigned m23 = ((bu[3] < 1u ? bu[3] : 1u) << 1) | (bu[2] < 1u ? bu[2] : 1u);
          __builtin_amdgcn_raw_buffer_store_b8((unsigned char)((m23 << 2) | m01), bview, ov0 >> 4, (oso + m * tile_bytes) >> 4, 0);
        }
      };
      if (nv == 4) {
#pragma unroll
        for (int m = 0; m < kMT; ++m) tile(m, std::true_type());
      } else {
#pragma unroll
        for (int m = 0; m < kMT; ++m) tile(m, std::false_type());
      }
      if (more) {
        wave_lds_fence();                              // this wave's reads of frame t are done
        band_store_i8(myring + ((t + 4) % kSlots) * kBandBytes, pf, lane);
        wave_lds_fence();
      }
    }
  }
}


// ---- launcher hunk (launch_fwd) ----
  {
    // the int8-pipe forward (SEEDHIP_STACK_I8=0: the bf16x3 kernel): buffer addressing only (tensors below 2 GB)
    static const int i8 = getenv("SEEDHIP_STACK_I8") ? atoi(getenv("SEEDHIP_STACK_I8")) : 1;
    const long long lim = (1LL << 31) - (1 << 20);
    if (bf16x3 && i8 && (long long)(3 + p.T1) * p.B * p.fsz < lim && (long long)p.T1 * p.B * 400 * p.ld_out * 4 < lim) {
      int grid;
      decompose(p.T1, p.B, max_grid_for(i8 > 1 ? i8 : 2), &p.spc, &p.items, &grid);
      p.buf32 = 1;
#define SEEDHIP_SCI(BITS_, RELU_)                                                                                  \
      {                                                                                                           \
        hipLaunchKernelGGL((stackconv_fwd_i8_kernel<BITS_, RELU_>), dim3(grid, 1, g->cout / 16), dim3(kThreads), kI8Lds, s, p); \
        return check_launch("stackconv_fwd_i8_kernel");                                                           \
      }
      if (relu_bits) SEEDHIP_SCI(true, true)
      if (out_relu) SEEDHIP_SCI(false, true)
      SEEDHIP_SCI(false, false)
#undef SEEDHIP_SCI
    }
  }

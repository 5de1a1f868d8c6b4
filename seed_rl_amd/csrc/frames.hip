// Learner-side frame stacking.
//
// Replaces /root/reference/atari/networks.py:57-173 (stack_frames): unpack of
// the bit-packed int32 stacking state, newest->oldest stacking with the
// cumulative-OR done mask, and re-packing of the last 3 frames.
//
// The cumulative-OR mask is monotone in the stack index, so the whole mask
// collapses to one byte per (t,b):  nvalid[t,b] = 1 if done[t], 2 if done[t-1],
// 3 if done[t-2], else 4; channel c (0 = newest) of step t is frame[t-c] when
// c < nvalid[t,b] and 0 otherwise (frames from before the unroll come from the
// state, which the previous unroll already zeroed past its own episode ends).
//
// HBM layout chosen for the fused conv1 loader: an EXTENDED uint8 frame buffer
// frames_ext[3+T, B, HW] whose first 3 time rows hold the unpacked state
// (row 2 = most recent), so that "frame at time t-c" is always frames_ext[t-c+3]
// and the 16x-larger fp32 stacked tensor of the reference is never materialised.
//
// Kernels (all HBM-bound byte work, 1 B per pixel per frame read once):
//   stack_prepare : state -> frames_ext[0:3], done -> nvalid
//   stack_frames  : reference-shaped fp32 [T,B,HW,4] output (parity / generic use)
//   stack_pack    : new int32 state from the last step
#include "common.h"
#include "../../include/seedhip.h"

namespace {

__global__ void __launch_bounds__(256)
stack_prepare_kernel(const int* __restrict__ state, const uint8_t* __restrict__ done, int T, int B, long long HW,
                     uint8_t* __restrict__ frames_ext, uint8_t* __restrict__ nvalid) {
  const long long n = (long long)B * HW;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int s = state[i];
    frames_ext[i] = (uint8_t)(s & 0xFF);                  // oldest   (time -3)
    frames_ext[n + i] = (uint8_t)((s >> 8) & 0xFF);       //          (time -2)
    frames_ext[2 * n + i] = (uint8_t)((s >> 16) & 0xFF);  // newest   (time -1)
  }
  const long long tb = (long long)T * B;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < tb; i += stride) {
    const long long t = i / B;
    int nv = 4;
    if (done[i]) nv = 1;
    else if (t >= 1 && done[i - B]) nv = 2;
    else if (t >= 2 && done[i - 2 * (long long)B]) nv = 3;
    nvalid[i] = (uint8_t)nv;
  }
}

__global__ void __launch_bounds__(256)
stack_frames_kernel(const uint8_t* __restrict__ frames_ext, const uint8_t* __restrict__ nvalid, int T, int B,
                    long long HW, float* __restrict__ out) {
  // one thread per (t,b,pixel); writes a float4 (newest -> oldest).
  const long long n = (long long)T * B * HW;
  const long long bhw = (long long)B * HW;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const long long tb = i / HW;
    const int nv = nvalid[tb];
    const uint8_t* p = frames_ext + i + 3 * bhw;          // frame at time t
    float4 o;
    o.x = (float)p[0];
    o.y = nv > 1 ? (float)p[-bhw] : 0.f;
    o.z = nv > 2 ? (float)p[-2 * bhw] : 0.f;
    o.w = nv > 3 ? (float)p[-3 * bhw] : 0.f;
    reinterpret_cast<float4*>(out)[i] = o;
  }
}

__global__ void __launch_bounds__(256)
stack_pack_kernel(const uint8_t* __restrict__ frames_ext, const uint8_t* __restrict__ nvalid, int T, int B,
                  long long HW, int* __restrict__ new_state) {
  const long long n = (long long)B * HW;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const long long b = i / HW;
    const int nv = nvalid[(long long)(T - 1) * B + b];
    const uint8_t* p = frames_ext + (long long)(T - 1 + 3) * n + i;   // newest frame
    const int f0 = p[0];
    const int f1 = nv > 1 ? p[-n] : 0;
    const int f2 = nv > 2 ? p[-2 * n] : 0;
    new_state[i] = (f0 << 16) | (f1 << 8) | f2;            // networks.py:164-169 (MSB = newest)
  }
}

// ---- the same two steps against a per-environment state TABLE (central inference, learner.py:381-403) ------------- //
// `state_table[rows[b]]` is column b's packed state: stack_prepare reads it in place (zero_mask[b] != 0: the actor
// restarted, its state counts as zeros) and stack_pack writes the new state back in place (valid_mask[b] == 0: skip
// the row) -- the gather of n x 28 KB into a scratch and the scatter back (2 x 58 MB of traffic per inference batch of
// 1024 Atari envs) disappear.  Four pixels per thread: one 16-byte state access, three / one 4-byte frame accesses.
__global__ void __launch_bounds__(256)
stack_prepare_rows_kernel(const int* __restrict__ table, const long long* __restrict__ rows,
                          const uint8_t* __restrict__ zero_mask, const uint8_t* __restrict__ done, int T, int B,
                          long long HW, uint8_t* __restrict__ frames_ext, uint8_t* __restrict__ nvalid) {
  const long long q = HW >> 2, n4 = (long long)B * q, n = (long long)B * HW;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const long long b = i / q, e = i - b * q;
    uint4 s = make_uint4(0, 0, 0, 0);
    if (!(zero_mask && zero_mask[b]))
      s = reinterpret_cast<const uint4*>(table + (rows ? rows[b] : b) * HW)[e];
    const unsigned v[4] = {s.x, s.y, s.z, s.w};
    unsigned f[3] = {0, 0, 0};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      f[0] |= (v[k] & 0xFFu) << (8 * k);                      // oldest   (time -3)
      f[1] |= ((v[k] >> 8) & 0xFFu) << (8 * k);               //          (time -2)
      f[2] |= ((v[k] >> 16) & 0xFFu) << (8 * k);              // newest   (time -1)
    }
    const long long at = b * HW + 4 * e;
#pragma unroll
    for (int r = 0; r < 3; ++r) *reinterpret_cast<unsigned*>(frames_ext + r * n + at) = f[r];
  }
  const long long tb = (long long)T * B;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < tb; i += stride) {
    const long long t = i / B;
    int nv = 4;
    if (done[i]) nv = 1;
    else if (t >= 1 && done[i - B]) nv = 2;
    else if (t >= 2 && done[i - 2 * (long long)B]) nv = 3;
    nvalid[i] = (uint8_t)nv;
  }
}

__global__ void __launch_bounds__(256)
stack_pack_rows_kernel(const uint8_t* __restrict__ frames_ext, const uint8_t* __restrict__ nvalid, int T, int B,
                       long long HW, int* __restrict__ table, const long long* __restrict__ rows,
                       const uint8_t* __restrict__ valid_mask) {
  const long long q = HW >> 2, n4 = (long long)B * q, n = (long long)B * HW;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const long long b = i / q, e = i - b * q;
    if (valid_mask && !valid_mask[b]) continue;
    const int nv = nvalid[(long long)(T - 1) * B + b];
    const uint8_t* p = frames_ext + (long long)(T - 1 + 3) * n + b * HW + 4 * e;      // newest frame, 4 pixels
    const unsigned f0 = *reinterpret_cast<const unsigned*>(p);
    const unsigned f1 = nv > 1 ? *reinterpret_cast<const unsigned*>(p - n) : 0u;
    const unsigned f2 = nv > 2 ? *reinterpret_cast<const unsigned*>(p - 2 * n) : 0u;
    unsigned o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)                               // networks.py:164-169 (MSB = newest)
      o[k] = (((f0 >> (8 * k)) & 0xFFu) << 16) | (((f1 >> (8 * k)) & 0xFFu) << 8) | ((f2 >> (8 * k)) & 0xFFu);
    reinterpret_cast<uint4*>(table + (rows ? rows[b] : b) * HW)[e] = make_uint4(o[0], o[1], o[2], o[3]);
  }
}

int grid_for(long long n) { int g = seedhip::cdiv(n, 256); return g > 4096 ? 4096 : (g < 1 ? 1 : g); }
}  // namespace

extern "C" int seedhip_stack_prepare_indexed(const int* state_table, const long long* rows, const uint8_t* zero_mask,
                                             const uint8_t* done, int T, int B, long long HW, uint8_t* frames_ext,
                                             uint8_t* nvalid, void* stream) {
  SEEDHIP_REQUIRE(T >= 1 && B >= 1 && HW >= 4 && HW % 4 == 0, "stack_prepare_indexed: need T, B >= 1 and HW %% 4 == 0");
  SEEDHIP_REQUIRE(state_table && done && frames_ext && nvalid, "stack_prepare_indexed: null pointer");
  SEEDHIP_REQUIRE(((((uintptr_t)state_table) & 15) | (((uintptr_t)frames_ext) & 3)) == 0,
                  "stack_prepare_indexed: state table must be 16-byte aligned, frames 4-byte aligned");
  const long long n = (long long)B * (HW / 4) > (long long)T * B ? (long long)B * (HW / 4) : (long long)T * B;
  hipLaunchKernelGGL(stack_prepare_rows_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, state_table, rows,
                     zero_mask, done, T, B, HW, frames_ext, nvalid);
  return seedhip::check_launch("stack_prepare_rows_kernel");
}

extern "C" int seedhip_stack_pack_state_indexed(const uint8_t* frames_ext, const uint8_t* nvalid, int T, int B,
                                                long long HW, int* state_table, const long long* rows,
                                                const uint8_t* valid_mask, void* stream) {
  SEEDHIP_REQUIRE(T >= 1 && B >= 1 && HW >= 4 && HW % 4 == 0, "stack_pack_indexed: need T, B >= 1 and HW %% 4 == 0");
  SEEDHIP_REQUIRE(frames_ext && nvalid && state_table, "stack_pack_indexed: null pointer");
  SEEDHIP_REQUIRE(((((uintptr_t)state_table) & 15) | (((uintptr_t)frames_ext) & 3)) == 0,
                  "stack_pack_indexed: state table must be 16-byte aligned, frames 4-byte aligned");
  hipLaunchKernelGGL(stack_pack_rows_kernel, dim3(grid_for((long long)B * (HW / 4))), dim3(256), 0, (hipStream_t)stream,
                     frames_ext, nvalid, T, B, HW, state_table, rows, valid_mask);
  return seedhip::check_launch("stack_pack_rows_kernel");
}

extern "C" int seedhip_stack_prepare(const int* frame_stacking_state, const uint8_t* done, int T, int B,
                                     long long HW, uint8_t* frames_ext, uint8_t* nvalid, void* stream) {
  SEEDHIP_REQUIRE(T >= 1 && B >= 1 && HW >= 1, "stack_prepare: bad T/B/HW");
  SEEDHIP_REQUIRE(frame_stacking_state && done && frames_ext && nvalid, "stack_prepare: null pointer");
  const long long n = (long long)B * HW > (long long)T * B ? (long long)B * HW : (long long)T * B;
  hipLaunchKernelGGL(stack_prepare_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream,
                     frame_stacking_state, done, T, B, HW, frames_ext, nvalid);
  return seedhip::check_launch("stack_prepare_kernel");
}

extern "C" int seedhip_stack_frames_f32(const uint8_t* frames_ext, const uint8_t* nvalid, int T, int B, long long HW,
                                        float* stacked, void* stream) {
  SEEDHIP_REQUIRE(T >= 1 && B >= 1 && HW >= 1, "stack_frames: bad T/B/HW");
  SEEDHIP_REQUIRE(frames_ext && nvalid && stacked, "stack_frames: null pointer");
  SEEDHIP_REQUIRE((((uintptr_t)stacked) & 15) == 0, "stack_frames: output must be 16-byte aligned");
  hipLaunchKernelGGL(stack_frames_kernel, dim3(grid_for((long long)T * B * HW)), dim3(256), 0, (hipStream_t)stream,
                     frames_ext, nvalid, T, B, HW, stacked);
  return seedhip::check_launch("stack_frames_kernel");
}

extern "C" int seedhip_stack_pack_state(const uint8_t* frames_ext, const uint8_t* nvalid, int T, int B, long long HW,
                                        int* new_state, void* stream) {
  SEEDHIP_REQUIRE(T >= 1 && B >= 1 && HW >= 1, "stack_pack: bad T/B/HW");
  SEEDHIP_REQUIRE(frames_ext && nvalid && new_state, "stack_pack: null pointer");
  hipLaunchKernelGGL(stack_pack_kernel, dim3(grid_for((long long)B * HW)), dim3(256), 0, (hipStream_t)stream,
                     frames_ext, nvalid, T, B, HW, new_state);
  return seedhip::check_launch("stack_pack_kernel");
}

// ---- football observations: packed bit planes -> uint8 0 / 255 (football/observation.py:48-63) ---- //
// unpackbits: every uint16 word holds 16 binary channels in the order 2^7..2^0, 2^15..2^8 (the byte order np.packbits
// leaves in a little-endian uint16); channel j of word w becomes out[16 w + j] = 255 if the bit is set.  The first conv
// then reads the bytes with its /255 fused (dmlab-style uint8 input): the float tensor the reference materialises (and
// its XLA-compiled cast) never exists.  HBM-bound byte work: 2 B read + 16 B written per word, one 16-byte store per lane.
namespace {
__global__ void __launch_bounds__(256)
unpackbits_u16_kernel(const uint16_t* __restrict__ in, long long n_words, uint4* __restrict__ out) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long w = (long long)blockIdx.x * blockDim.x + threadIdx.x; w < n_words; w += stride) {
    const unsigned v = in[w];
    unsigned o[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      unsigned word = 0;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int j = 4 * q + b;                               // output channel
        const int bit = j < 8 ? 7 - j : 23 - j;
        word |= ((v >> bit) & 1u) ? (0xFFu << (8 * b)) : 0u;
      }
      o[q] = word;
    }
    out[w] = make_uint4(o[0], o[1], o[2], o[3]);
  }
}
}  // namespace

extern "C" int seedhip_unpackbits_u16(const uint16_t* packed, long long n_words, uint8_t* out, void* stream) {
  SEEDHIP_REQUIRE(n_words >= 0, "unpackbits: negative size");
  if (n_words == 0) return SEEDHIP_OK;
  SEEDHIP_REQUIRE(packed && out, "unpackbits: null pointer");
  SEEDHIP_REQUIRE((((uintptr_t)out) & 15) == 0, "unpackbits: output must be 16-byte aligned");
  long long blocks = (n_words + 255) / 256; if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(unpackbits_u16_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, packed, n_words,
                     reinterpret_cast<uint4*>(out));
  return seedhip::check_launch("unpackbits_u16_kernel");
}

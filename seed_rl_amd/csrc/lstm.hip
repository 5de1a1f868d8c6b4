// LSTM core of the agents, time-unrolled with done-reset.
//
// Replaces the Python `for t` loop of /root/reference/dmlab/networks.py:152-171 and
// /root/reference/atari/networks.py:176-218 (_unroll_cell) around
// tf.keras.layers.LSTMCell (implementation 2):
//     state = where(done_t, 0, state)
//     z = x_t W + h U + b ; i,f,g,o = split(z) ; c' = sig(f) c + sig(i) tanh(g) ; h' = sig(o) tanh(c')
// and its TF autodiff (back-propagation through time).
//
// Decomposition (the dense contractions run on the fp32-MFMA GEMM core, conv.hip):
//   * x_t W + b for ALL steps is one GEMM [T*B, in] x [in, 4H] (done by the caller);
//   * per step, h_in U is one small GEMM whose epilogue adds the pre-computed x-projection
//     (`residual`), then `lstm_gates_fwd` applies the cell non-linearities AND the done-reset
//     for the NEXT step (it writes keep_{t+1} * state into the next step's input slot), so
//     the reset costs no extra pass;
//   * backward per step: `lstm_gates_bwd` (recomputes the gates from the stored pre-activations,
//     produces dz_t and the cell-state gradient) then one GEMM dz_t U^T; the weight gradients
//     dW, dU, db are three large GEMMs over all steps at the end (done by the caller).
// Elementwise kernels: HBM/L2-bound, float4 per thread.
#include "common.h"
#include "../../include/seedhip.h"

namespace {

__device__ __forceinline__ float sigm(float x) { return 1.0f / (1.0f + expf(-x)); }

// X[n, feat] = reward (optionally clipped to [-1,1]); X[n, feat+1+a] = one_hot(prev_action)[a]; pad = 0.
__global__ void __launch_bounds__(256)
lstm_assemble_kernel(float* __restrict__ x, int ldx, int feat, int num_actions, const float* __restrict__ reward,
                     const void* __restrict__ prev_actions, int action_elem_size, int clip_reward, long long rows) {
  const int extra = ldx - feat;
  const long long total = rows * extra;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const long long n = i / extra;
    const int e = (int)(i - n * extra);
    float v = 0.f;
    if (e == 0) {
      v = reward[n];
      if (clip_reward) v = fminf(fmaxf(v, -1.0f), 1.0f);
    } else if (e <= num_actions) {
      const long long a = action_elem_size == 8 ? ((const long long*)prev_actions)[n]
                                                : (long long)((const int*)prev_actions)[n];
      v = (a == e - 1) ? 1.0f : 0.f;
    }
    x[n * ldx + feat + e] = v;
  }
}

// dst = keep * src for the initial state (keep = !done[0]).
__global__ void __launch_bounds__(256)
lstm_mask_state_kernel(const float* __restrict__ h0, const float* __restrict__ c0, const uint8_t* __restrict__ done0,
                       int B, int H, float* __restrict__ hin, float* __restrict__ cin) {
  const int total = B * H;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const float keep = done0[i / H] ? 0.f : 1.f;
    hin[i] = h0[i] * keep;
    cin[i] = c0[i] * keep;
  }
}

// One step of the cell non-linearities.  z [B,4H] (i,f,g,o), cin [B,H] (already reset).
// Writes h_out [B,H] (row stride ld_h), and the NEXT step's inputs hin_next/cin_next =
// keep_next * (h', c') (keep_next = !done_next[b]; done_next == null -> no reset: final state).
__global__ void __launch_bounds__(256)
lstm_gates_fwd_kernel(const float* __restrict__ z, const float* __restrict__ cin, const uint8_t* __restrict__ done_next,
                      int B, int H, float* __restrict__ h_out, int ld_h, float* __restrict__ hin_next,
                      float* __restrict__ cin_next) {
  const int h4 = H >> 2;
  const int total = B * h4;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int b = i / h4, j = (i - b * h4) * 4;
    const float* zr = z + (long long)b * 4 * H + j;
    const float4 zi = *reinterpret_cast<const float4*>(zr);
    const float4 zf = *reinterpret_cast<const float4*>(zr + H);
    const float4 zg = *reinterpret_cast<const float4*>(zr + 2 * H);
    const float4 zo = *reinterpret_cast<const float4*>(zr + 3 * H);
    const float4 cp = *reinterpret_cast<const float4*>(cin + (long long)b * H + j);
    float4 c, h;
#define CELL(k) { const float ig = sigm(zi.k), fg = sigm(zf.k), gg = tanhf(zg.k), og = sigm(zo.k); \
                  c.k = fg * cp.k + ig * gg; h.k = og * tanhf(c.k); }
    CELL(x) CELL(y) CELL(z) CELL(w)
#undef CELL
    *reinterpret_cast<float4*>(h_out + (long long)b * ld_h + j) = h;
    const float keep = (done_next && done_next[b]) ? 0.f : 1.f;
    *reinterpret_cast<float4*>(hin_next + (long long)b * H + j) = make_float4(h.x * keep, h.y * keep, h.z * keep, h.w * keep);
    *reinterpret_cast<float4*>(cin_next + (long long)b * H + j) = make_float4(c.x * keep, c.y * keep, c.z * keep, c.w * keep);
  }
}

// Backward of one step.  dh_out [B,H] (row stride ld_dh): gradient arriving at h_t from the layers above;
// dh_rec / dc_rec [B,H]: RAW recurrent gradients produced by step t+1 (null at the last step),
// masked here by keep_next = !done_next[b].  Outputs dz [B,4H] and dc_prev [B,H] (raw: the mask of THIS
// step's reset is applied by step t-1's call).
__global__ void __launch_bounds__(256)
lstm_gates_bwd_kernel(const float* __restrict__ z, const float* __restrict__ cin, const float* __restrict__ dh_out,
                      int ld_dh, const float* __restrict__ dh_rec, const float* __restrict__ dc_rec,
                      const uint8_t* __restrict__ done_next, int B, int H, float* __restrict__ dz,
                      float* __restrict__ dc_prev) {
  const int h4 = H >> 2;
  const int total = B * h4;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int b = i / h4, j = (i - b * h4) * 4;
    const float* zr = z + (long long)b * 4 * H + j;
    const float4 zi = *reinterpret_cast<const float4*>(zr);
    const float4 zf = *reinterpret_cast<const float4*>(zr + H);
    const float4 zg = *reinterpret_cast<const float4*>(zr + 2 * H);
    const float4 zo = *reinterpret_cast<const float4*>(zr + 3 * H);
    const float4 cp = *reinterpret_cast<const float4*>(cin + (long long)b * H + j);
    float4 dh = *reinterpret_cast<const float4*>(dh_out + (long long)b * ld_dh + j);
    float4 dc = make_float4(0.f, 0.f, 0.f, 0.f);
    const float keep = (done_next && done_next[b]) ? 0.f : 1.f;
    if (dh_rec) {
      const float4 r = *reinterpret_cast<const float4*>(dh_rec + (long long)b * H + j);
      dh.x += keep * r.x; dh.y += keep * r.y; dh.z += keep * r.z; dh.w += keep * r.w;
    }
    if (dc_rec) {
      const float4 r = *reinterpret_cast<const float4*>(dc_rec + (long long)b * H + j);
      dc = make_float4(keep * r.x, keep * r.y, keep * r.z, keep * r.w);
    }
    float4 di, df, dg, dO, dcp;
#define CELL(k) { const float ig = sigm(zi.k), fg = sigm(zf.k), gg = tanhf(zg.k), og = sigm(zo.k); \
                  const float c = fg * cp.k + ig * gg; const float tc = tanhf(c);                   \
                  const float d_o = dh.k * tc; const float d_c = dc.k + dh.k * og * (1.0f - tc * tc); \
                  di.k = d_c * gg * ig * (1.0f - ig); df.k = d_c * cp.k * fg * (1.0f - fg);          \
                  dg.k = d_c * ig * (1.0f - gg * gg); dO.k = d_o * og * (1.0f - og); dcp.k = d_c * fg; }
    CELL(x) CELL(y) CELL(z) CELL(w)
#undef CELL
    float* dr = dz + (long long)b * 4 * H + j;
    *reinterpret_cast<float4*>(dr) = di;
    *reinterpret_cast<float4*>(dr + H) = df;
    *reinterpret_cast<float4*>(dr + 2 * H) = dg;
    *reinterpret_cast<float4*>(dr + 3 * H) = dO;
    *reinterpret_cast<float4*>(dc_prev + (long long)b * H + j) = dcp;
  }
}

int grid_for(long long n) { long long b = (n + 255) / 256; return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b)); }

}  // namespace

extern "C" int seedhip_lstm_assemble_inputs(float* x, int ldx, int feat, int num_actions, const float* reward,
                                            const void* prev_actions, int action_elem_size, int clip_reward,
                                            long long rows, void* stream) {
  SEEDHIP_REQUIRE(x && reward && prev_actions, "lstm_assemble_inputs: null pointer");
  SEEDHIP_REQUIRE(rows >= 1 && feat >= 0 && num_actions >= 1 && ldx >= feat + 1 + num_actions,
                  "lstm_assemble_inputs: ldx must hold feat + 1 + num_actions columns");
  SEEDHIP_REQUIRE(action_elem_size == 4 || action_elem_size == 8, "lstm_assemble_inputs: action_elem_size must be 4 or 8");
  hipLaunchKernelGGL(lstm_assemble_kernel, dim3(grid_for(rows * (ldx - feat))), dim3(256), 0, (hipStream_t)stream, x,
                     ldx, feat, num_actions, reward, prev_actions, action_elem_size, clip_reward, rows);
  return seedhip::check_launch("lstm_assemble_kernel");
}

extern "C" int seedhip_lstm_mask_state(const float* h0, const float* c0, const uint8_t* done0, int B, int H, float* hin,
                                       float* cin, void* stream) {
  SEEDHIP_REQUIRE(h0 && c0 && done0 && hin && cin && B >= 1 && H >= 1, "lstm_mask_state: bad argument");
  hipLaunchKernelGGL(lstm_mask_state_kernel, dim3(grid_for((long long)B * H)), dim3(256), 0, (hipStream_t)stream, h0,
                     c0, done0, B, H, hin, cin);
  return seedhip::check_launch("lstm_mask_state_kernel");
}

extern "C" int seedhip_lstm_gates_fwd(const float* z, const float* cin, const uint8_t* done_next, int B, int H,
                                      float* h_out, int ld_h, float* hin_next, float* cin_next, void* stream) {
  SEEDHIP_REQUIRE(z && cin && h_out && hin_next && cin_next, "lstm_gates_fwd: null pointer");
  SEEDHIP_REQUIRE(B >= 1 && H >= 4 && H % 4 == 0 && ld_h >= H && ld_h % 4 == 0, "lstm_gates_fwd: need H %% 4 == 0, ld_h >= H");
  hipLaunchKernelGGL(lstm_gates_fwd_kernel, dim3(grid_for((long long)B * H / 4)), dim3(256), 0, (hipStream_t)stream, z,
                     cin, done_next, B, H, h_out, ld_h, hin_next, cin_next);
  return seedhip::check_launch("lstm_gates_fwd_kernel");
}

extern "C" int seedhip_lstm_gates_bwd(const float* z, const float* cin, const float* dh_out, int ld_dh,
                                      const float* dh_rec, const float* dc_rec, const uint8_t* done_next, int B, int H,
                                      float* dz, float* dc_prev, void* stream) {
  SEEDHIP_REQUIRE(z && cin && dh_out && dz && dc_prev, "lstm_gates_bwd: null pointer");
  SEEDHIP_REQUIRE(B >= 1 && H >= 4 && H % 4 == 0 && ld_dh >= H && ld_dh % 4 == 0, "lstm_gates_bwd: need H %% 4 == 0, ld_dh >= H");
  hipLaunchKernelGGL(lstm_gates_bwd_kernel, dim3(grid_for((long long)B * H / 4)), dim3(256), 0, (hipStream_t)stream, z,
                     cin, dh_out, ld_dh, dh_rec, dc_rec, done_next, B, H, dz, dc_prev);
  return seedhip::check_launch("lstm_gates_bwd_kernel");
}

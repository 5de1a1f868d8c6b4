/* seedhip.h -- C ABI of libseedhip.so: the MI355X (gfx950) SEED-RL learner hot path.
 *
 * The reference hot path is pure Python/TensorFlow graph code with no FFI of its
 * own; each entry point below replaces one Python-level function
 * (or the TF op sequence it expands to) of /root/reference and is what a binding
 * for that function would call (see INTEGRATION.md for the ctypes stubs).
 *
 * Conventions:
 *   - all pointers are DEVICE pointers to caller-allocated HBM, never retained past
 *     the call; no hidden allocations (workspaces are passed in, sized by the
 *     matching *_workspace_bytes query);
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it;
 *   - returns 0 on success, <0 on error (never throws / aborts);
 *     seedhip_last_error() gives the thread-local message;
 *   - re-entrant, no global mutable state; reductions are deterministic
 *     (no floating-point atomics);
 *   - tensors are contiguous row-major, time-major [T, B, ...], NHWC, fp32 unless
 *     stated; Keras kernel layouts ([kh,kw,cin,cout], Dense [in,out]).
 */
#ifndef SEEDHIP_H_
#define SEEDHIP_H_
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Bumped whenever an EXISTING entry point changes its signature or a *_workspace_bytes() contract changes (new entry
 * points alone do not bump it: a missing symbol already fails at load).  History: 1 = round 1; 2 = round 2
 * (seedhip_adam_flat*, seedhip_inference_pre/post signatures, impala-loss workspace size); 3 = round 3; 4 = round 4;
 * 5 = round 5 (seedhip_conv2d_stack_bwd_weight_fused* REMOVED: the fused pair lost to the two-call path).
 * Bindings must compare seedhip_abi_version() with the version they were written against before the first call
 * (seed_rl_amd/_lib.py does): a stale library would otherwise be called with shifted arguments. */
#define SEEDHIP_ABI_VERSION 6

const char* seedhip_last_error(void);
int seedhip_abi_version(void);

/* ---- V-trace ------------------------------------------------------------------
 * Replaces common/vtrace.py:34-148 `from_importance_weights` (arithmetic :84-148).
 * Inputs [T,B] (+[B] bootstrap), outputs vs / pg_advantages [T,B].  Trailing dims of
 * the reference ([T,B,C], vtrace.py:49-51) are folded into B by the caller.
 * clip_*_threshold < 0 means None (no clipping, vtrace.py:111-114,138-142). */
int seedhip_vtrace_from_importance_weights(
    const float* target_action_log_probs, const float* behaviour_action_log_probs,
    const float* discounts, const float* rewards, const float* values,
    const float* bootstrap_value, float clip_rho_threshold, float clip_pg_rho_threshold,
    float lambda_, int T, long long B, float* vs, float* pg_advantages, void* stream);

/* ---- categorical distribution ---------------------------------------------------
 * Replaces common/parametric_distribution.py:69-74 log_prob / entropy for
 * categorical_distribution (:83-97; tfd.Categorical).  logits [rows, A]; actions
 * int32 (elem_size 4) or int64 (8); either output may be NULL. */
int seedhip_categorical_log_prob_entropy(const float* logits, const void* actions, int action_elem_size,
                                         long long rows, int A, float* log_prob, float* entropy,
                                         void* stream);

/* ---- IMPALA loss head, forward + backward -------------------------------------------
 * Replaces agents/vtrace/learner.py:82-157 (compute_loss after the agent unroll)
 * and its autodiff wrt the learner outputs.  All inputs have T+1 time steps.
 * logits rows may be strided (row stride in floats) so the head GEMM output
 * [rows, ld] can be consumed / its gradient produced in place:
 *   learner_policy_logits[(t*B+b)*logits_ld + a], learner_baseline[(t*B+b)*baseline_ld],
 *   d_policy_logits / d_baseline use the same strides.
 * mean_denominator = N of the reduce_means (T*B; global T*B for data-parallel mean).
 * vs / pg_advantages ([T,B], optional) are the V-trace outputs for parity checks.
 * scalars[SEEDHIP_LOSS_NUM] receives the losses and the logged values. */
enum {
  SEEDHIP_LOSS_TOTAL = 0,       /* learner.py:134-135 */
  SEEDHIP_LOSS_POLICY = 1,      /* :111-112 */
  SEEDHIP_LOSS_V = 2,           /* :115-116 */
  SEEDHIP_LOSS_ENTROPY = 3,     /* :121 */
  SEEDHIP_LOSS_KL = 4,          /* :124-125 */
  SEEDHIP_LOSS_ENTROPY_MEAN = 5,/* :119-120, logged :154 */
  SEEDHIP_LOSS_KL_MEAN = 6,     /* logged :156 */
  SEEDHIP_LOSS_VALUE_MEAN = 7,  /* logged :139-140 */
  SEEDHIP_LOSS_V_L2_ERROR = 8,  /* logged :141 */
  SEEDHIP_LOSS_MAX_ACTION_ABS = 9, /* logged :152-153 */
  SEEDHIP_LOSS_ENTROPY_COST = 10,  /* logged :155 (agent.entropy_cost()) */
  SEEDHIP_LOSS_ENTROPY_ADJUSTMENT = 11, /* :128-132 */
  SEEDHIP_LOSS_NUM = 16
};
size_t seedhip_impala_loss_workspace_bytes(int T, int B);
int seedhip_impala_loss_fwd_bwd(
    const float* learner_policy_logits, int logits_ld, const float* learner_baseline, int baseline_ld,
    const float* behaviour_policy_logits, const void* actions, int action_elem_size,
    const float* rewards, const uint8_t* done, int T, int B, int A,
    float entropy_cost, float baseline_cost, float kl_cost, float discounting, float lambda_,
    float max_abs_reward, float clip_rho_threshold, float clip_pg_rho_threshold,
    float mean_denominator, float* d_policy_logits, float* d_baseline, float* vs, float* pg_advantages,
    float* scalars, void* workspace, size_t workspace_bytes, void* stream);
/* The same head with the learner's LEARNABLE entropy cost (agents/vtrace/learner.py:225-234: the learner attaches
 * entropy_cost_param = log(FLAGS.entropy_cost) / speed to every agent without an entropy_cost() of its own, and
 * entropy_cost = exp(speed * param)) and the Lagrange-style adjustment loss of :127-135:
 *   has_target_entropy: total += cost * stop_gradient(mean(H) - target_entropy),
 *                       d_entropy_cost_param[0] = speed * cost * (mean(H) - target_entropy);
 *   otherwise           the term is 0 * cost and the gradient 0 (never "None", :131-132).
 * entropy_cost_param / d_entropy_cost_param are device scalars (the parameter lives in the flat parameter buffer, its
 * gradient in the flat gradient buffer).  For the data-parallel 'mean' reduction pass target_entropy / world (each
 * replica's mean(H) is already its share of the global mean). */
int seedhip_impala_loss_fwd_bwd_adaptive(
    const float* learner_policy_logits, int logits_ld, const float* learner_baseline, int baseline_ld,
    const float* behaviour_policy_logits, const void* actions, int action_elem_size,
    const float* rewards, const uint8_t* done, int T, int B, int A,
    const float* entropy_cost_param, float entropy_cost_adjustment_speed, int has_target_entropy,
    float target_entropy, float* d_entropy_cost_param,
    float baseline_cost, float kl_cost, float discounting, float lambda_,
    float max_abs_reward, float clip_rho_threshold, float clip_pg_rho_threshold,
    float mean_denominator, float* d_policy_logits, float* d_baseline, float* vs, float* pg_advantages,
    float* scalars, void* workspace, size_t workspace_bytes, void* stream);

/* CRC32C (Castagnoli) of a HOST buffer, continuing from `crc` (0 to start): the checksum TensorFlow's checkpoint
 * bundles carry (tensorflow/core/lib/hash/crc32c.h); lets tf_checkpoint.py read / write tf.train.Checkpoint files of the
 * reference (agents/vtrace/learner.py:286-296) at memory speed.  Host code; no stream. */
unsigned int seedhip_crc32c(const void* data, size_t n, unsigned int crc);

/* A HIP stream restricted to a subset of the compute units (r6): mask = `words` 32-bit words, bit i of word w = compute
 * unit 32 w + i.  The closed serving loop runs central inference and the train step on disjoint CU sets
 * (learner_server.LearnerServer); the reference places inference and training on different TPU cores
 * (agents/vtrace/learner.py:199-214, 406-414).  *stream is a hipStream_t; destroy it with seedhip_stream_destroy. */
int seedhip_stream_create_cu_mask(const unsigned int* mask, int words, void** stream);
int seedhip_stream_destroy(void* stream);

/* ---- optimizer --------------------------------------------------------------------
 * Replaces the Keras Adam apply_gradients of agents/vtrace/learner.py:272-275
 * (dmlab/vtrace_main.py:46-51) over ONE flat parameter buffer.  lr_t already
 * includes the bias correction (host, fp64).  grad_scale multiplies the gradient
 * first (1/world for data-parallel mean; 1 for the reference cross-replica SUM).
 * clamp_index >= 0: params[clamp_index] carries a Keras variable constraint (the entropy-cost parameter's
 * clip_by_value(v, -20/speed, 20/speed), learner.py:228-230), applied after its update; -1: none. */
int seedhip_adam_flat(float* params, const float* grads, float* m, float* v, long long n,
                      float lr_t, float beta_1, float beta_2, float epsilon, float grad_scale,
                      long long clamp_index, float clamp_lo, float clamp_hi, void* stream);
/* Same update with lr_t read from a device scalar, so that a whole train step (whose only per-step host
 * value is the bias-corrected learning rate) can be captured once in a HIP graph and replayed. */
int seedhip_adam_flat_dev_lr(float* params, const float* grads, float* m, float* v, long long n,
                             const float* lr_t_device, float beta_1, float beta_2, float epsilon,
                             float grad_scale, long long clamp_index, float clamp_lo, float clamp_hi, void* stream);
/* Both of the above (lr_t_device NULL: lr_t is used) with a GUARD: when int32 *skip_if_nonzero (device memory, may be
 * NULL) is non-zero at launch time on the device, the update is skipped altogether -- parameters and moments stay as
 * they are.  The word is the sticky abort flag of the LSTM sequence kernels: a step whose gradients are invalid is
 * dropped instead of applied, inside a replayed HIP graph too (no host decision on the step's critical path). */
int seedhip_adam_flat_guarded(float* params, const float* grads, float* m, float* v, long long n,
                              float lr_t, const float* lr_t_device, float beta_1, float beta_2, float epsilon,
                              float grad_scale, long long clamp_index, float clamp_lo, float clamp_hi,
                              const int* skip_if_nonzero, void* stream);
/* tf.clip_by_global_norm(grads, clip_norm) of agents/r2d2/learner.py:606-609; sumsq_out[0] = |g|^2.
 * clip_norm <= 0: only compute the norm. */
size_t seedhip_global_norm_workspace_bytes(void);
int seedhip_clip_by_global_norm(float* grads, long long n, float clip_norm, float* sumsq_out,
                                void* workspace, size_t workspace_bytes, void* stream);

/* ---- frame stacking -------------------------------------------------------------------
 * Replaces atari/networks.py:57-173 `stack_frames` (bit-packed int32 state, :32-54).
 * frames_ext is uint8 [3+T, B, HW]: rows 3.. hold the unroll's frames (written by the
 * caller / the unroll store), rows 0..2 are filled by seedhip_stack_prepare from
 * the packed state; nvalid is uint8 [T,B] (number of valid stack channels). */
/* football/observation.py:48-63 unpackbits: packed uint16 bit planes [n_words] -> uint8 [n_words * 16], 255 where the
 * bit is set, channel order 2^7..2^0, 2^15..2^8 per word (what GFootball._torso feeds its first conv after / 255). */
int seedhip_unpackbits_u16(const uint16_t* packed, long long n_words, uint8_t* out, void* stream);
int seedhip_stack_prepare(const int* frame_stacking_state, const uint8_t* done, int T, int B, long long HW,
                          uint8_t* frames_ext, uint8_t* nvalid, void* stream);
int seedhip_stack_frames_f32(const uint8_t* frames_ext, const uint8_t* nvalid, int T, int B, long long HW,
                             float* stacked /* [T,B,HW,4] newest->oldest, range 0..255 */, void* stream);
int seedhip_stack_pack_state(const uint8_t* frames_ext, const uint8_t* nvalid, int T, int B, long long HW,
                             int* new_state /* [B,HW] */, void* stream);
/* The same two steps against a per-environment state TABLE int32[num_envs, HW] (central inference keeps one packed
 * state per env, learner.py:381-403): column b's state is state_table[rows[b]] (rows NULL: row b), read in place --
 * zero_mask[b] != 0 (may be NULL): the actor restarted, its state counts as zeros (:363-365) -- and written back in
 * place (valid_mask[b] == 0: row skipped).  Replaces a gather of n rows into a scratch and a scatter back. HW % 4 == 0. */
int seedhip_stack_prepare_indexed(const int* state_table, const long long* rows, const uint8_t* zero_mask,
                                  const uint8_t* done, int T, int B, long long HW, uint8_t* frames_ext, uint8_t* nvalid,
                                  void* stream);
int seedhip_stack_pack_state_indexed(const uint8_t* frames_ext, const uint8_t* nvalid, int T, int B, long long HW,
                                     int* state_table, const long long* rows, const uint8_t* valid_mask, void* stream);

/* ---- Conv2D / Dense on fp32 MFMA -------------------------------------------------------
 * Replace the Keras Conv2D / Dense forward of dmlab/networks.py:31-60,84-89,116-118 and
 * atari/networks.py:233-251 and their TF autodiff.  NHWC fp32; kernel [kh,kw,cin,cout];
 * Dense == 1x1 conv on a 1x1 image (n_img = rows).  ld_in / ld_out = pixel strides. */
typedef struct {
  int n_img, ih, iw, cin, oh, ow, kh, kw, stride, pad_t, pad_l, cout;
  int ld_in, ld_out;
} seedhip_conv_geom;
/* in_dtype: 0 = fp32, 1 = uint8 scaled by 1/255 (dmlab/networks.py:98-100).
 * out = act(conv(relu?(in)) + bias (+ residual)); bias / residual may be NULL. */
int seedhip_conv2d_fwd(const seedhip_conv_geom* geom, const void* in, int in_dtype, int in_relu,
                       const float* w, const float* bias, float* out, int out_relu,
                       const float* residual, void* stream);
/* Same, with a caller workspace (seedhip_conv2d_fwd_workspace_bytes, 0 for most shapes): dense layers whose grid
 * would leave the chip under-filled (inference batches, per-step recurrent GEMMs) split the reduction over
 * workgroups and finish in a deterministic reduce + epilogue launch.  Without workspace they run unsplit. */
size_t seedhip_conv2d_fwd_workspace_bytes(const seedhip_conv_geom* geom);
int seedhip_conv2d_fwd_ws(const seedhip_conv_geom* geom, const void* in, int in_dtype, int in_relu,
                          const float* w, const float* bias, float* out, int out_relu, const float* residual,
                          void* workspace, size_t workspace_bytes, void* stream);
/* dx = conv_transpose(dy, w); then dx *= (relu_mask > 0) if relu_mask; dx += add if add. */
int seedhip_conv2d_bwd_data(const seedhip_conv_geom* geom, const float* dy, const float* w, float* dx,
                            const float* relu_mask, const float* add, void* stream);
size_t seedhip_conv2d_bwd_data_workspace_bytes(const seedhip_conv_geom* geom);
int seedhip_conv2d_bwd_data_ws(const seedhip_conv_geom* geom, const float* dy, const float* w, float* dx,
                               const float* relu_mask, const float* add, void* workspace, size_t workspace_bytes,
                               void* stream);
/* Which matrix pipe serves this geometry; pass 0 forward, 1 data gradient, 2 weight gradient.  Returns 1 = fp32 MFMA,
 * 6 = bf16 MFMA through the exact three-way split of both fp32 operands (six bf16 MACs per algorithmic MAC: ceiling
 * 2500 / 6 algorithmic TFLOP/s); 0 on a null geometry / unknown pass.  Reporting only (bench.py's roofline). */
int seedhip_conv2d_pipe(const seedhip_conv_geom* geom, int pass);
size_t seedhip_conv2d_bwd_weight_workspace_bytes(const seedhip_conv_geom* geom);
/* dw[kh,kw,cin,cout] and dbias[cout] (dbias may be NULL) are OVERWRITTEN. */
int seedhip_conv2d_bwd_weight(const seedhip_conv_geom* geom, const void* in, int in_dtype, int in_relu,
                              const float* dy, float* dw, float* dbias, void* workspace,
                              size_t workspace_bytes, void* stream);

/* First Atari conv fused with frame stacking + /255 (atari/networks.py:57-173,234,330):
 * consumes frames_ext / nvalid directly, VALID padding, 4 stacked channels.  nvalid [T, B] (bytes) is read through the
 * scalar cache in aligned 4-byte words: the allocation must be readable up to the next 4-byte boundary on both sides of
 * the array (any hipMalloc / torch allocation is). */
typedef struct { int T, B, ih, iw, oh, ow, kh, kw, stride, cout, ld_out; } seedhip_stack_conv_geom;
int seedhip_conv2d_stack_fwd(const seedhip_stack_conv_geom* geom, const uint8_t* frames_ext,
                             const uint8_t* nvalid, const float* w, const float* bias, float* out,
                             int out_relu, void* stream);
/* The ReLU mask as BYTES (r3): seedhip_conv2d_stack_fwd_bits is seedhip_conv2d_stack_fwd with out_relu = 1 that
 * also writes relu_bits [T * B * oh * ow, ld_out / 4] -- bit r of byte [pixel][q] = out[pixel][4 q + r] > 0 --, and
 * seedhip_conv2d_bwd_data_bits is seedhip_conv2d_bwd_data (no `add`) that reads those bytes (indexed like dx / 4 floats)
 * instead of the fp32 activation: 1/16 of the mask traffic of the next layer's data gradient
 * (atari torso: Conv 8x8/4 x16 -> ReLU -> Conv 4x4/2; TF autodiff's ReluGrad).  Each is served by ONE specialised kernel:
 * ask the *_supported query (1 / 0, geometry only; operands must be 16-byte aligned) and use the fp32-mask entry points
 * otherwise -- the *_bits calls return SEEDHIP_ERR_UNSUPPORTED rather than fall back. */
int seedhip_conv2d_stack_fwd_bits_supported(const seedhip_stack_conv_geom* geom);
int seedhip_conv2d_stack_fwd_bits(const seedhip_stack_conv_geom* geom, const uint8_t* frames_ext,
                                  const uint8_t* nvalid, const float* w, const float* bias, float* out,
                                  uint8_t* relu_bits, void* stream);
/* r5: the same pair one layer up -- seedhip_conv2d_fwd_bits is seedhip_conv2d_fwd with out_relu = 1 (no residual) that
 * also writes relu_bits [n_img * oh * ow, cout / 4] for the layer that consumes `out` (the shallow torso's Dense layer:
 * seedhip_conv2d_bwd_data_bits also serves Dense geometries, relu_bits indexed like dx / 4 floats). */
int seedhip_conv2d_fwd_bits_supported(const seedhip_conv_geom* geom);
int seedhip_conv2d_fwd_bits(const seedhip_conv_geom* geom, const void* in, int in_dtype, int in_relu, const float* w,
                            const float* bias, float* out, uint8_t* relu_bits, void* stream);
int seedhip_conv2d_bwd_data_bits_supported(const seedhip_conv_geom* geom);
int seedhip_conv2d_bwd_data_bits(const seedhip_conv_geom* geom, const float* dy, const float* w, float* dx,
                                 const uint8_t* relu_bits, void* stream);
/* r5, ImpalaDeep's residual blocks (dmlab/networks.py:41-51: relu -> conv -> relu -> conv -> += skip): every tensor that
 * is read through a ReLU gets its mask written by the kernel that PRODUCES it, from the epilogue's registers.
 * seedhip_conv2d_fwd_outbits is seedhip_conv2d_fwd with out_relu = 0 (bias / residual may be NULL) that also writes
 * out_bits [n_img * oh * ow, cout / 4] -- bit r of byte [pixel][q] = out[pixel][4 q + r] > 0;
 * seedhip_maxpool3x3s2_same_fwd_bits / seedhip_conv3x3_u8_pool_fwd_bits (below) do the same for the pooled tensor that
 * enters a stack's first block.  seedhip_conv2d_bwd_data_bits_add is seedhip_conv2d_bwd_data with those bytes for
 * relu_mask and the skip path's gradient `add` (may be NULL) added behind the mask.  Served for the 3x3 'same' 16->16
 * (36x48) and 32->32 (18x24, 9x12) layers at training batch sizes: ask seedhip_conv2d_fwd_outbits_supported (it answers
 * for both calls); no fallback (SEEDHIP_ERR_UNSUPPORTED). */
int seedhip_conv2d_fwd_outbits_supported(const seedhip_conv_geom* geom);
int seedhip_conv2d_fwd_outbits(const seedhip_conv_geom* geom, const float* in, int in_relu, const float* w,
                               const float* bias, float* out, const float* residual, uint8_t* out_bits, void* stream);
int seedhip_conv2d_bwd_data_bits_add(const seedhip_conv_geom* geom, const float* dy, const float* w, float* dx,
                                     const uint8_t* relu_bits, const float* add, void* stream);
/* r5: the data gradient of a convolution whose output went through MaxPool2D(3, 2, 'same') (ImpalaDeep's stack-entry
 * layers, dmlab/networks.py:31-37; TF autodiff: MaxPoolGrad -> Conv2DBackpropInput) from the gradient of the POOLED map
 * dpooled [n, oh / 2, ow / 2, cout] and the pool's argmax bytes: the max-pool backward runs in the kernel's loader, and the
 * pre-pool gradient is also written to d_prepool [n, oh, ow, cout] (the weight gradient's operand).  Bit-identical to
 * seedhip_maxpool3x3s2_same_bwd + seedhip_conv2d_bwd_data.  Served for the 16 -> 32 layer on 36 x 48 maps at training
 * batch sizes (ask _supported); no fallback. */
int seedhip_conv2d_bwd_data_pool_supported(const seedhip_conv_geom* geom);
int seedhip_conv2d_bwd_data_pool(const seedhip_conv_geom* geom, const float* dpooled, const uint8_t* argmax,
                                 const float* w, float* dx, float* d_prepool, void* stream);
size_t seedhip_conv2d_stack_bwd_weight_workspace_bytes(const seedhip_stack_conv_geom* geom);
int seedhip_conv2d_stack_bwd_weight(const seedhip_stack_conv_geom* geom, const uint8_t* frames_ext,
                                    const uint8_t* nvalid, const float* dy, float* dw, float* dbias,
                                    void* workspace, size_t workspace_bytes, void* stream);

/* ---- MaxPool2D(3, strides 2, 'same') -------------------------------------------------------
 * Replaces the Keras MaxPool2D of dmlab/networks.py:36-37 and its autodiff (TF 'SAME' padding:
 * 0 before / 1 after for even sizes).  NHWC fp32, c % 4 == 0.  y [n, ceil(ih/2), ceil(iw/2), c];
 * argmax (uint8, same shape as y) is written by fwd and consumed by bwd. */
int seedhip_maxpool3x3s2_same_fwd(int n, int ih, int iw, int c, const float* x, float* y, uint8_t* argmax,
                                  void* stream);
int seedhip_maxpool3x3s2_same_bwd(int n, int ih, int iw, int c, const float* dy, const uint8_t* argmax, float* dx,
                                  void* stream);
/* fwd that also writes the ReLU mask of y as bytes y_bits [n * oh * ow, c / 4] (seedhip_conv2d_fwd_outbits; NULL: plain) */
int seedhip_maxpool3x3s2_same_fwd_bits(int n, int ih, int iw, int c, const float* x, float* y, uint8_t* argmax,
                                       uint8_t* y_bits, void* stream);

/* ---- first ImpalaDeep stage fused: Conv2D(16, 3, 'same') on uint8 frames (x/255) + MaxPool2D(3, 2, 'same') ----
 * Replaces dmlab/networks.py:31-37 for stack 0 (with the x/255 of :98-100) and the autodiff of that pair wrt the conv
 * kernel and bias; the 72x96x16 pre-pool activation (2.4 GB at T=20, B=256) and its gradient are never written.
 * x u8 [n, ih, iw, 3]; w [3,3,3,16] (Keras layout); pooled fp32 / argmax u8 [n, ceil(ih/2), ceil(iw/2), 16] with the
 * argmax byte code of seedhip_maxpool3x3s2_same_fwd.  cin must be 3, cout 16, iw <= 114.
 * bwd: dpooled = gradient wrt `pooled`; dw [3,3,3,16], dbias [16] or null; workspace from the _workspace_bytes call. */
int seedhip_conv3x3_u8_pool_fwd(const uint8_t* x, int n, int ih, int iw, int cin, const float* w, const float* bias,
                                int cout, float* pooled, uint8_t* argmax, void* stream);
/* fwd that also writes the ReLU mask of `pooled` as bytes [n * ph * pw, 4] (seedhip_conv2d_fwd_outbits; NULL: plain) */
int seedhip_conv3x3_u8_pool_fwd_bits(const uint8_t* x, int n, int ih, int iw, int cin, const float* w, const float* bias,
                                     int cout, float* pooled, uint8_t* argmax, uint8_t* pooled_bits, void* stream);
size_t seedhip_conv3x3_u8_pool_bwd_workspace_bytes(int n, int ih, int iw);
int seedhip_conv3x3_u8_pool_bwd(const uint8_t* x, int n, int ih, int iw, int cin, const float* dpooled,
                                const uint8_t* argmax, int cout, float* dw, float* dbias, void* workspace,
                                size_t workspace_bytes, void* stream);

/* ---- LSTM core with done-reset ---------------------------------------------------------------
 * Replace the time loop of dmlab/networks.py:152-171 / atari/networks.py:176-218 (_unroll_cell)
 * around tf.keras.layers.LSTMCell (gate order i,f,g,o; one bias) and its autodiff; the dense
 * contractions (x W + b for all steps, h U per step, dz U^T per step, dW/dU/db at the end) are
 * seedhip_conv2d_* calls on 1x1 geometries -- see seed_rl_amd/networks.py:_LstmCore.
 *   assemble_inputs: x[n, feat] = reward (clipped to [-1,1] if clip_reward: dmlab/networks.py:112;
 *                    raw for R2D2: atari/networks.py:263-271), x[n, feat+1+a] = one_hot(prev_action),
 *                    remaining pad columns up to ldx = 0.
 *   mask_state:      hin/cin = where(done0, 0, h0/c0)
 *   gates_fwd:       z [B,4H] pre-activations, cin [B,H] (already reset) -> h_out [B,H] (stride ld_h),
 *                    hin_next/cin_next = where(done_next, 0, (h', c')); done_next NULL = no reset.
 *   gates_bwd:       dh_out (stride ld_dh) + keep_next * dh_rec -> dz [B,4H], dc_prev [B,H];
 *                    dh_rec / dc_rec may be NULL (last step). */
int seedhip_lstm_assemble_inputs(float* x, int ldx, int feat, int num_actions, const float* reward,
                                 const void* prev_actions, int action_elem_size, int clip_reward, long long rows,
                                 void* stream);
int seedhip_lstm_mask_state(const float* h0, const float* c0, const uint8_t* done0, int B, int H, float* hin,
                            float* cin, void* stream);
/* One whole LSTM step in one launch (recurrent GEMM + gates + done-reset): z = zx + hin * U, then as
 * seedhip_lstm_gates_fwd.  `up` is U [H, 4H] re-laid out by seedhip_lstm_permute_u (column 4*unit + gate), once per
 * forward pass.  Needs H % 128 == 0 (seedhip_lstm_step_supported); hin / up 16-byte aligned.  zx, z [B, 4H]. */
int seedhip_lstm_permute_u(const float* u, int H, float* up, void* stream);
int seedhip_lstm_step_supported(int B, int H);
int seedhip_lstm_step_fwd(const float* hin, const float* up, const float* zx, const float* cin,
                          const uint8_t* done_next, int B, int H, float* z, float* h_out, int ld_h, float* hin_next,
                          float* cin_next, void* stream);
/* The whole unroll of T1 steps in ONE launch (the `for t` loop of dmlab/networks.py:152-171 itself): workgroups stay
 * resident, keep their slice of U in LDS and the cell state in registers; h_t is exchanged through hin with
 * agent-scope stores / loads and is its own ready flag (the call pre-fills hin[1..T1] with a sentinel bit pattern and
 * consumers re-read until none is left).  Bit-identical to T1 calls of seedhip_lstm_step_fwd.  zx, z [T1, B, 4H];
 * done [T1, B] (done[0] is NOT applied here: hin[0] / cin[0] hold the already-masked initial state,
 * seedhip_lstm_mask_state); hin, cin [T1 + 1, B, H] (slot t + 1 = state after step t, reset where done[t + 1]);
 * h_out [T1 * B, ld_h].  sync_ws: 8 bytes of device memory, zeroed by the call; every wait is bounded, and after the
 * launch completes int32 sync_ws[1] != 0 means one timed out (the workgroups were not co-resident) and the outputs
 * are invalid.  seedhip_lstm_seq_supported: T1 >= 2, H % 128 == 0, H <= 512, and the grid of ceil(B / 32) * H / 16
 * workgroups fits the device's CUs (needs a current HIP device).  (z's last step doubles as scratch for the kernel's
 * start-up XCD handshake before step T1 - 1 overwrites it: do not read z concurrently with the call.) */
int seedhip_lstm_seq_supported(int T1, int B, int H);
int seedhip_lstm_seq_fwd(const float* up, const float* zx, const uint8_t* done, int T1, int B, int H, float* z,
                         float* h_out, int ld_h, float* hin, float* cin, void* sync_ws, void* stream);
/* The same with a STICKY abort word: when a wait times out, int32 *sticky_abort (device memory, may be NULL) is set
 * to 1 together with sync_ws[1] and -- unlike sync_ws, which every launch zeroes -- is never cleared by the library:
 * it tells seedhip_adam_flat_guarded to drop the step whose gradients came from an aborted sequence kernel, and the
 * host (which clears it) to fall back to seedhip_lstm_step_fwd. */
int seedhip_lstm_seq_fwd_sticky(const float* up, const float* zx, const uint8_t* done, int T1, int B, int H, float* z,
                                float* h_out, int ld_h, float* hin, float* cin, void* sync_ws, int* sticky_abort,
                                void* stream);
/* The whole BACKWARD recurrence in one launch: for t = T1-1 .. 0, dz_t = cell backward of step t (as
 * seedhip_lstm_gates_bwd: dh = dh_out_t + keep_{t+1} * dh_rec, dc = keep_{t+1} * dc_rec) and dh_rec = dz_t U^T (the
 * per-step seedhip_conv2d_bwd_data of the dense geometry [B, H] -> [B, 4H]).  Same residency / tiling as
 * seedhip_lstm_seq_fwd (and the same `up`); partial dh_rec sums travel through ring_ws
 * (seedhip_lstm_seq_bwd_workspace_bytes, 16-byte aligned, filled by the call) and are added in a fixed order:
 * deterministic, equal to the per-step path up to fp32 summation order.  z, dz [T1, B, 4H] (gate-major columns);
 * cin [>= T1, B, H] (slot t = cell state entering step t); dh_out [T1 * B, ld_dh]; done [T1, B].  sync_ws as in
 * seedhip_lstm_seq_fwd.  Supported exactly when seedhip_lstm_seq_supported(T1, B, H). */
size_t seedhip_lstm_seq_bwd_workspace_bytes(int B, int H);
int seedhip_lstm_seq_bwd(const float* up, const float* z, const float* cin, const float* dh_out, int ld_dh,
                         const uint8_t* done, int T1, int B, int H, float* dz, void* ring_ws, void* sync_ws,
                         void* stream);
int seedhip_lstm_seq_bwd_sticky(const float* up, const float* z, const float* cin, const float* dh_out, int ld_dh,
                                const uint8_t* done, int T1, int B, int H, float* dz, void* ring_ws, void* sync_ws,
                                int* sticky_abort, void* stream);   /* sticky_abort as in seedhip_lstm_seq_fwd_sticky */
int seedhip_lstm_gates_fwd(const float* z, const float* cin, const uint8_t* done_next, int B, int H, float* h_out,
                           int ld_h, float* hin_next, float* cin_next, void* stream);
int seedhip_lstm_gates_bwd(const float* z, const float* cin, const float* dh_out, int ld_dh, const float* dh_rec,
                           const float* dc_rec, const uint8_t* done_next, int B, int H, float* dz, float* dc_prev,
                           void* stream);

/* ---- R2D2: dueling head + n-step double-Q loss ---------------------------------------------------
 * dueling_fwd/bwd replace atari/networks.py:273-285 (_head): va [rows, ld] holds the advantage head
 * output in columns 0..A-1 and the value head output in column A; q = value + adv - mean(adv),
 * action = argmax(q) (int32, may be NULL).  bwd: d_va from dq (same layout, pad columns zeroed).
 * r2d2_loss_fwd_bwd replaces agents/r2d2/learner.py:258-330 + the `reduce_mean(loss * importance_weights)`
 * of :604 and its gradient wrt training_q: inputs are the post-burn-in [T,B,...] tensors; actions int32;
 * n-step double-Q Bellman target with value rescaling (epsilon = --value_function_rescaling_epsilon);
 * outputs loss_per_sequence [B], priorities [B], d_training_q [T,B,A], total_loss [1]
 * (= sum_b loss_b * w_b / mean_denominator; importance_weights NULL = 1). */
int seedhip_dueling_fwd(const float* va, int ld, long long rows, int A, float* q, int* action, void* stream);
int seedhip_dueling_bwd(const float* dq, long long rows, int A, float* d_va, int ld, void* stream);
size_t seedhip_r2d2_loss_workspace_bytes(int T, int B, int n_steps);
int seedhip_r2d2_loss_fwd_bwd(const float* training_q, const float* target_q, const int* actions,
                              const float* rewards, const uint8_t* done, const float* importance_weights,
                              int T, int B, int A, float gamma, int n_steps, float eta, float epsilon,
                              float mean_denominator, float* loss_per_sequence, float* priorities,
                              float* d_training_q, float* total_loss, void* workspace, size_t workspace_bytes,
                              void* stream);

/* ---- trajectory store row mover -------------------------------------------------------------------
 * The one data-movement primitive behind the device-resident UnrollStore / Aggregator
 * (common/utils.py:155-257, 461-543: scatter_nd_update / sparse_read / gather_nd) and the direct
 * time-major batch assembly that replaces make_time_major (utils.py:735-761):
 *   dst[dst_rows[i]] = src[src_rows[i]],  i < n, rows of row_bytes bytes.
 * dst_rows / src_rows NULL = identity (row i); src NULL = zero fill.  Rows written must be distinct and
 * must not alias rows read. */
int seedhip_rows_move(void* dst, const long long* dst_rows, const void* src, const long long* src_rows,
                      long long n, long long row_bytes, void* stream);

/* ---- batched inference bookkeeping (no host synchronisation, HIP-graph capturable) -----------------------
 * The small-tensor part of the `inference` function of agents/vtrace/learner.py:350-405 on the device store.
 * n = inference batch size (<= 65536: one workgroup walks it in 1024-row chunks), ids int64, unique within a call.
 *  inference_pre : run-id compare / resets (:353-366), episode statistics (:373-378; finished episodes are
 *                  appended to episode_stats[stats_capacity][3] = (frames, return, raw_return) through *stats_count),
 *                  previous actions (:381).  Outputs reset_mask u8[n], prev_actions i64[n].
 *  inference_post: advances the unroll-store index (utils.py:187-194), detects completed unrolls (:229-233), assigns
 *                  them consecutive columns of a time-major training batch of `batch_capacity` columns
 *                  (*batch_count += number completed) and emits the row-index lists the row mover needs:
 *                  append_rows[n], complete u8[n], batch_cols[n], gather_src/dst/mask [full_length*n],
 *                  last_rows[n] (the step carried to slot 0, utils.py:237-252; overlap 0).  actions_table[e] = action.
 *  error_flag bits: 1 id out of range, 2 duplicate ids, 4 store overflow, 8 training batch overflow.
 *  rows_move_masked: seedhip_rows_move with a per-row mask: zero_where_masked = 0 moves only rows with mask != 0;
 *                  1 moves every row but writes zeros where mask != 0. */
int seedhip_rows_move_masked(void* dst, const long long* dst_rows, const void* src, const long long* src_rows,
                             long long n, long long row_bytes, const uint8_t* row_mask, int zero_where_masked,
                             void* stream);
/* The same move for up to 16 fields (host arrays of per-field dst / src / row_bytes) in ONE launch; row_mask may be
 * NULL (plain move). */
int seedhip_rows_move_multi(int nfields, void* const* dst, const void* const* src, const long long* row_bytes,
                            const long long* dst_rows, const long long* src_rows, long long n,
                            const uint8_t* row_mask, int zero_where_masked, void* stream);
/* inference_pre also validates the ids: out-of-range (flag 1) and duplicate (flag 2, found in O(1) through
 * stamp_table[num_envs] / *call_counter, both zero-initialised by the caller and owned by these kernels) rows get
 * valid[i] = 0 and are skipped by every table access of the step; ids_safe[i] is the id clamped into range for the
 * gathers that run unmasked.  (The reference raises on both: common/utils.py:173-176.)
 * will_complete u8[n] (optional; ABI 3): 1 where the env's unroll completes with this step (store index, after a
 * possible reset, + 1 == full_length) -- known before the agent runs, so the caller can set aside the previous agent
 * state of exactly those envs (first_agent_states.replace(completed, agent_states.read(completed)), learner.py:398-399)
 * and let the agent update its state tables in place. */
int seedhip_inference_pre(const long long* env_ids, const long long* run_ids, const float* reward,
                          const float* raw_reward, const uint8_t* done, int n, int num_envs, int num_action_repeats,
                          long long* run_ids_table, long long* info_frames, float* info_return,
                          float* info_raw_return, long long* actions_table, long long* store_index,
                          uint8_t* reset_mask, long long* prev_actions, float* episode_stats, int stats_capacity,
                          int* stats_count, int* error_flag, long long* ids_safe, uint8_t* valid, int* stamp_table,
                          int* call_counter, uint8_t* will_complete /* may be NULL */, int full_length, void* stream);
/* inference_post: env_ids = ids_safe, valid = the mask of inference_pre (NULL: every in-range row).  With
 * policy_logits != NULL the actions are SAMPLED here from the head rows policy_logits[i*logits_ld + a] (Gumbel-max
 * over Philox4x32-10 randoms keyed by rng_state[0] = seed, rng_state[1] = call counter, advanced by this launch:
 * the categorical sample of dmlab/networks.py:122 without an eager op) and written to actions[n]; otherwise actions[n]
 * is an input.  carry u8[n] marks every completed unroll (its last step is carried to slot 0, utils.py:237-252), complete
 * u8[n] those that also got a training-batch column (flag 8 when the batch is full: the unroll is dropped, the env's
 * store stays consistent); emit_env / emit_col [n] + *emit_count: the same completions as a compact list in batch-column
 * order, for seedhip_emit_unrolls.  The training batch is a RING of batch_capacity columns: *batch_start (device
 * scalar, NULL = 0) is its head, *batch_count its fill; the r-th accepted completion gets column (start + fill + r) %
 * capacity, so the consumer takes columns from the head and advances it -- nothing is ever compacted. */
int seedhip_inference_post(const long long* env_ids, const uint8_t* valid, long long* actions,
                           const float* policy_logits, int logits_ld, int num_actions, unsigned long long* rng_state,
                           int n, int num_envs, int full_length, int batch_capacity, long long* store_index,
                           long long* actions_table, int* batch_count, long long* append_rows, uint8_t* complete,
                           uint8_t* carry, long long* batch_cols, long long* emit_env, long long* emit_col,
                           int* emit_count, long long* last_rows, int* error_flag,
                           const int* batch_start /* may be NULL = 0 */, void* stream);
/* Completed unrolls -> training batch (unroll_queue.enqueue_many + dequeue + make_time_major of learner.py:396-397,
 * 418-432) from the COMPACT list inference_post leaves on the device (ABI 3): emit_env[r] / emit_col[r], r < *emit_count,
 * are the env and the batch column of the r-th completed unroll; for every field f and step t < full_length, store row
 * t * num_envs + env_r (row_bytes[f] bytes) moves to batch row t * batch_capacity + col_r.  Up to 16 fields (host arrays
 * of device pointers); max_unrolls (= the inference batch size) only bounds the grid -- the kernel reads the count. */
int seedhip_emit_unrolls(int nfields, void* const* dst, const void* const* src, const long long* row_bytes,
                         const long long* emit_env, const long long* emit_col, const int* emit_count, int max_unrolls,
                         int full_length, int num_envs, int batch_capacity, void* stream);
/* The agents' action sampling (tfd.Categorical(logits).sample(), common/parametric_distribution.py:94-95 as used by
 * dmlab/networks.py:122): actions[r] ~ Categorical(logits[r*ld .. r*ld + num_actions)), int64.  Same generator and
 * per-row function as inference_post: equal (seed, counter) give equal actions. */
int seedhip_categorical_sample(const float* logits, int ld, long long rows, int num_actions,
                               unsigned long long* rng_state, long long* actions, void* stream);
/* Up to 32 INDEPENDENT row moves in one launch (no operation may read rows another one writes):
 *   dst[dst_rows[i] * dst_pitch ...] = src[src_rows[i] * src_pitch ...], row_bytes bytes, i < n;
 * pitches in bytes (0 = row_bytes: dense rows), so strided sources such as the logits columns of a head-GEMM output are
 * read in place; row_mask / zero_where_masked as in seedhip_rows_move_masked; src NULL = zero fill. */
typedef struct seedhip_row_op {
  void* dst; const void* src; long long row_bytes; long long dst_pitch; long long src_pitch;
  const long long* dst_rows; const long long* src_rows; long long n; const uint8_t* row_mask; int zero_where_masked;
} seedhip_row_op;
int seedhip_rows_move_ops(int nops, const seedhip_row_op* ops, void* stream);

/* ---- central inference in six launches for the frame-stacked Atari agents (r6; csrc/servestep.hip) ----------
 * The same `inference` function (agents/vtrace/learner.py:350-405) for agents whose only recurrent state is the frame
 * stack (atari/networks.py:57-173), with the stack kept WHERE THE UNROLL STORE ALREADY HOLDS IT: stack channel c >= 1 of
 * env e at store slot idx is the observation at slot idx - c (slots below 0 wrap to full_length - 1 + (idx - c): slot 0
 * is the carried copy of the last slot, common/utils.py:237-255), valid iff c <= stack_valid[e] and the step is not
 * `done`; stack_valid' = min(3, done ? 1 : stack_valid + 1) -- the reference's cumulative-OR done masks
 * (networks.py:131-157) as a counter.  Needs full_length >= 5.  One step =
 *   seedhip_serve_begin              every piece of bookkeeping that does not depend on the network (what
 *                                    seedhip_inference_pre + _post do, incl. batch columns in env_ids order), and the first
 *                                    conv's W / 255 as three bf16 planes (conv0_split, seedhip_serve_conv0_split_bytes)
 *   seedhip_conv2d_stack_fwd_rows    first conv from the request frames + the store's history
 *   seedhip_conv2d_fwd ...           the torso's other convs (library kernels)
 *   seedhip_dense_fwd_partial        the Dense layer's split-K partial sums (no reduce / epilogue launch)
 *   seedhip_serve_finish             sum of the slices + bias + ReLU -> packed heads -> action sampling
 *                                    (dmlab/networks.py:122; same (seed, call, row) function as seedhip_categorical_sample)
 *                                    -> the step's fields (scalars and frames) appended to the store, action table (:403)
 *   seedhip_serve_emit               completed unrolls -> time-major training batch ring (:396-397, 418-432), last
 *                                    step carried to slot 0, first agent state handed over and the next one packed
 *                                    (:398-399; bit order of networks.py:164-169) from the store's frames.
 * Tables are per env [num_envs], scratch per row [n]; all device memory, caller-owned, zero-initialised once.
 * error_flag bits as seedhip_inference_pre / _post. */
typedef struct seedhip_serve_step {
  /* the request batch */
  const long long* env_ids; const long long* run_ids; const float* reward; const float* raw_reward;
  const uint8_t* done; const uint8_t* abandoned /* may be NULL */; const int* episode_step /* may be NULL */;
  int n, num_envs, num_action_repeats, full_length, batch_capacity;
  /* per-env tables */
  long long* run_ids_table; long long* info_frames; float* info_return; float* info_raw_return;
  long long* actions_table; long long* store_index;
  uint8_t* stack_valid;         /* frames of the stack in front of the next step that are inside the episode, 0..3 */
  uint8_t* first_zero;          /* 1: the env's current unroll starts from the initial (zero) agent state */
  int* stamp_table; int* call_counter;
  float* episode_stats; int stats_capacity; int* stats_count; int* error_flag;
  int* batch_count; const int* batch_start /* may be NULL = 0 */;
  unsigned long long* rng_state; /* [2] seed, call counter (advanced by serve_begin) */
  /* per-call scratch */
  long long* ids_safe; uint8_t* valid; long long* prev_actions;
  long long* append_rows;       /* [n] store row (slot * num_envs + env) of this step, -1 for masked rows */
  long long* hist_rows;         /* [n][4] store rows of stack channels 0..3 */
  uint8_t* nvalid;              /* [n] stack channels inside the episode at this step, 1..4 */
  uint8_t* prev_valid;          /* [n] stack_valid before this step */
  long long* emit_env; long long* emit_col /* -1: dropped, batch full */; int* emit_row; int* emit_count;
  unsigned long long* rng_snapshot; /* [2] the (seed, call) this step samples with */
} seedhip_serve_step;
/* the store's scalar fields, time-major [full_length, num_envs] (policy_logits [.., num_actions]) */
typedef struct seedhip_serve_fields {
  long long* prev_actions; float* reward; uint8_t* done; uint8_t* abandoned; int* episode_step;
  long long* action; float* policy_logits; float* baseline;
} seedhip_serve_fields;
size_t seedhip_serve_conv0_split_bytes(int cout);
/* the weight planes alone (what serve_begin's extra workgroups do): conv0_w [8,8,4,cout] fp32 -> conv0_split */
int seedhip_serve_split_conv0(const float* conv0_w, int conv0_cout, void* conv0_split, void* stream);
/* heads_image (may be NULL; seedhip_serve_heads_image_bytes(feat) bytes): the packed heads W [feat, ldh]
 * (seedhip_heads_supported) as the B-operand register image serve_finish multiplies with. */
size_t seedhip_serve_heads_image_bytes(int feat);
int seedhip_serve_begin(const seedhip_serve_step* step, const float* conv0_w /* [8,8,4,cout] */, int conv0_cout,
                        void* conv0_split /* may be NULL */, const float* heads_w, int feat, int ldh,
                        void* heads_image, void* stream);
/* geom->T == 1, geom->B == n.  obs u8 [n, ih*iw]; store_obs u8 [full_length * num_envs, ih*iw] (read only: history rows
 * hist_rows[4 b + c], 1 <= c < nvalid[b]; the request frames are appended by seedhip_serve_finish); w_split: the W / 255
 * planes of seedhip_serve_begin / seedhip_serve_split_conv0. */
int seedhip_conv2d_stack_fwd_rows_supported(const seedhip_stack_conv_geom* geom);
int seedhip_conv2d_stack_fwd_rows(const seedhip_stack_conv_geom* geom, const uint8_t* obs, const uint8_t* store_obs,
                                  const long long* hist_rows, const uint8_t* nvalid, const void* w_split,
                                  const float* bias, float* out, int out_relu, void* stream);
/* Dense forward without its epilogue: partial[z][m][n], z < *slices, in `workspace`; summing the slices in order + bias
 * (+ ReLU) equals seedhip_conv2d_fwd_ws's output for the same geometry below 4096 rows bit for bit. */
size_t seedhip_dense_fwd_partial_workspace_bytes(const seedhip_conv_geom* geom);
int seedhip_dense_fwd_partial(const seedhip_conv_geom* geom, const float* in, int in_relu, const float* w,
                              void* workspace, size_t workspace_bytes, int* slices, void* stream);
/* fc_partial [slices][n][feat]; heads_image: serve_begin's image of the packed heads [feat, ldh]; actions int64 [n] out.
 * obs u8 [n, hw] -> store_obs rows append_rows[b] (the append of the largest field, common/utils.py:187-194; by the waves
 * that are idle while one wave of each workgroup multiplies the heads and samples); hw % 16 == 0. */
int seedhip_serve_finish(const seedhip_serve_step* step, const seedhip_serve_fields* store_fields,
                         const float* fc_partial, int slices, const float* fc_bias, int feat, const void* heads_image,
                         const float* heads_b, int ldh, int num_actions, long long* actions, const uint8_t* obs,
                         uint8_t* store_obs, long long hw, void* stream);
/* nfields <= 16 fields (host arrays of device pointers; rows of row_bytes[f]): store [full_length, num_envs] -> batch
 * [full_length, batch_capacity]; first_table int32 [num_envs, hw], batch_first int32 [batch_capacity, hw], store_obs as
 * above (hw = ih * iw bytes per frame, % 16 == 0). */
int seedhip_serve_emit(const seedhip_serve_step* step, int nfields, void* const* batch, void* const* store,
                       const long long* row_bytes, int* first_table, int* batch_first, const uint8_t* store_obs,
                       long long hw, void* stream);

/* ---- prioritized replay sampling --------------------------------------------------------------------
 * Replaces PrioritizedReplay.sample of common/utils.py:309-357 for priority_exponent > 0: categorical sampling
 * (with replacement) proportional to priorities[i]^priority_exponent over the first `limit` slots, driven by
 * caller-supplied uniforms in [0,1) (one per sample), and the normalised importance weights
 * ((1/limit)/prob)^importance_sampling_exponent / max.  indices int64[num_samples], weights f32[num_samples]. */
size_t seedhip_replay_sample_workspace_bytes(long long limit);
int seedhip_replay_sample(const float* priorities, long long limit, float priority_exponent,
                          float importance_sampling_exponent, const float* uniforms, int num_samples,
                          long long* indices, float* weights, void* workspace, size_t workspace_bytes, void* stream);

/* ---- policy + baseline heads forward as one skinny GEMM ---------------------------------------------------
 * The two Dense heads of dmlab/networks.py:116-124 (policy_logits [feat, A], baseline [feat, 1]) packed as ONE matrix
 * w [feat, ldh] = [logits | baseline | zero pad], ldh = round4(A + 1):
 *   y [rows, ldh] = x [rows, feat] (row stride ldx) w + bias.
 * Supported (seedhip_heads_supported): feat % 64 == 0, feat <= 512, ldh % 4 == 0, ldh <= 32; anything else (and the
 * backward of the heads) goes through seedhip_conv2d_* with a dense geometry. */
int seedhip_heads_supported(int feat, int ldh);
int seedhip_heads_fwd(const float* x, int ldx, const float* w, const float* bias, long long rows, int feat, int ldh,
                      float* y, void* stream);

/* ---- R2D2 actor-side exploration and replay <-> time-major batch -----------------------------------
 * seedhip_epsilon_greedy replaces apply_epsilon_greedy of agents/r2d2/learner.py:147-177: actions[i] (int64, in
 * place) becomes a uniform random action in [0, num_actions) with probability epsilons[env_ids[i]] (the caller's
 * per-environment table = get_envs_epsilon, :129-145: 0.4^linspace(1, 8, num_training_envs) ++ eval_epsilon); ids out
 * of [0, num_envs) keep their action.  Randoms: Philox4x32-10 keyed by rng_state[0] = seed, rng_state[1] = call
 * counter (advanced by this call); replaced u8[n] (may be NULL) records which rows were replaced. */
int seedhip_epsilon_greedy(long long* actions, const long long* env_ids, const float* epsilons, int n, int num_envs,
                           int num_actions, unsigned long long* rng_state, uint8_t* replaced, void* stream);
/* Row indices for moving unrolls between replay rows [slot][t] and a TIME-MAJOR batch [t][column] with ONE row move
 * per field (utils.make_time_major of agents/r2d2/learner.py:453-457 folded into PrioritizedReplay's gather /
 * scatter): for k = t * num_unrolls + b: replay_rows[k] = slots[b] * steps + t, batch_rows[k] = k. */
int seedhip_replay_time_rows(const long long* slots, int num_unrolls, int steps, long long* replay_rows,
                             long long* batch_rows, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SEEDHIP_H_ */

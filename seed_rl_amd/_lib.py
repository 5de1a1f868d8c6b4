"""ctypes binding of libseedhip.so (the C ABI declared in include/seedhip.h).

The library is built in-tree by `python -m seed_rl_amd.build` (or
`__graft_entry__.build()`); there is no fallback -- if it is missing or a
symbol is absent this module raises at import/use.
"""
import ctypes
import os

# torch MUST be imported before libseedhip.so is loaded: torch ships its own
# libamdhip64 and the kernels share torch's HIP runtime (streams, device pointers).
# Loading ours first would bind it to a second, separate runtime.
import torch  # noqa: F401  pylint: disable=unused-import

_HERE = os.path.dirname(os.path.abspath(__file__))
# SEEDHIP_LIB: another build of the same library (same-box A/B runs of two builds; nothing else changes: no fallback)
LIB_PATH = os.environ.get('SEEDHIP_LIB') or os.path.join(_HERE, 'lib', 'libseedhip.so')
ABI_VERSION = 6          # include/seedhip.h SEEDHIP_ABI_VERSION this binding was written against

c_int, c_ll, c_float, c_size_t, c_void_p = (
    ctypes.c_int, ctypes.c_longlong, ctypes.c_float, ctypes.c_size_t, ctypes.c_void_p)
P = c_void_p


class ConvGeom(ctypes.Structure):
  """seedhip_conv_geom."""
  _fields_ = [(n, c_int) for n in
              'n_img ih iw cin oh ow kh kw stride pad_t pad_l cout ld_in ld_out'.split()]


class StackConvGeom(ctypes.Structure):
  """seedhip_stack_conv_geom."""
  _fields_ = [(n, c_int) for n in 'T B ih iw oh ow kh kw stride cout ld_out'.split()]


class RowOp(ctypes.Structure):
  """seedhip_row_op."""
  _fields_ = [('dst', c_void_p), ('src', c_void_p), ('row_bytes', c_ll), ('dst_pitch', c_ll), ('src_pitch', c_ll),
              ('dst_rows', c_void_p), ('src_rows', c_void_p), ('n', c_ll), ('row_mask', c_void_p),
              ('zero_where_masked', c_int)]


class ServeStep(ctypes.Structure):
  """seedhip_serve_step (field order of include/seedhip.h)."""
  _fields_ = ([(n, c_void_p) for n in 'env_ids run_ids reward raw_reward done abandoned episode_step'.split()] +
              [(n, c_int) for n in 'n num_envs num_action_repeats full_length batch_capacity'.split()] +
              [(n, c_void_p) for n in ('run_ids_table info_frames info_return info_raw_return actions_table store_index '
                                       'stack_valid first_zero stamp_table call_counter episode_stats').split()] +
              [('stats_capacity', c_int)] +
              [(n, c_void_p) for n in ('stats_count error_flag batch_count batch_start rng_state ids_safe valid '
                                       'prev_actions append_rows hist_rows nvalid prev_valid emit_env emit_col emit_row '
                                       'emit_count rng_snapshot').split()])


class ServeFields(ctypes.Structure):
  """seedhip_serve_fields."""
  _fields_ = [(n, c_void_p) for n in 'prev_actions reward done abandoned episode_step action policy_logits baseline'.split()]


# name -> (restype, argtypes); every symbol include/seedhip.h declares.
SIGNATURES = {
    'seedhip_last_error': (ctypes.c_char_p, []),
    'seedhip_abi_version': (c_int, []),
    'seedhip_crc32c': (ctypes.c_uint, [P, c_size_t, ctypes.c_uint]),
    'seedhip_stream_create_cu_mask': (c_int, [P, c_int, P]),
    'seedhip_stream_destroy': (c_int, [P]),
    'seedhip_vtrace_from_importance_weights':
        (c_int, [P, P, P, P, P, P, c_float, c_float, c_float, c_int, c_ll, P, P, P]),
    'seedhip_categorical_log_prob_entropy': (c_int, [P, P, c_int, c_ll, c_int, P, P, P]),
    'seedhip_impala_loss_workspace_bytes': (c_size_t, [c_int, c_int]),
    'seedhip_impala_loss_fwd_bwd':
        (c_int, [P, c_int, P, c_int, P, P, c_int, P, P, c_int, c_int, c_int,
                 c_float, c_float, c_float, c_float, c_float, c_float, c_float, c_float, c_float,
                 P, P, P, P, P, P, c_size_t, P]),
    'seedhip_impala_loss_fwd_bwd_adaptive':
        (c_int, [P, c_int, P, c_int, P, P, c_int, P, P, c_int, c_int, c_int,
                 P, c_float, c_int, c_float, P,
                 c_float, c_float, c_float, c_float, c_float, c_float, c_float, c_float,
                 P, P, P, P, P, P, c_size_t, P]),
    'seedhip_adam_flat': (c_int, [P, P, P, P, c_ll, c_float, c_float, c_float, c_float, c_float,
                                  c_ll, c_float, c_float, P]),
    'seedhip_adam_flat_dev_lr': (c_int, [P, P, P, P, c_ll, P, c_float, c_float, c_float, c_float,
                                         c_ll, c_float, c_float, P]),
    'seedhip_adam_flat_guarded': (c_int, [P, P, P, P, c_ll, c_float, P, c_float, c_float, c_float, c_float,
                                          c_ll, c_float, c_float, P, P]),
    'seedhip_global_norm_workspace_bytes': (c_size_t, []),
    'seedhip_clip_by_global_norm': (c_int, [P, c_ll, c_float, P, P, c_size_t, P]),
    'seedhip_unpackbits_u16': (c_int, [P, c_ll, P, P]),
    'seedhip_stack_prepare': (c_int, [P, P, c_int, c_int, c_ll, P, P, P]),
    'seedhip_stack_frames_f32': (c_int, [P, P, c_int, c_int, c_ll, P, P]),
    'seedhip_stack_pack_state': (c_int, [P, P, c_int, c_int, c_ll, P, P]),
    'seedhip_stack_prepare_indexed': (c_int, [P, P, P, P, c_int, c_int, c_ll, P, P, P]),
    'seedhip_stack_pack_state_indexed': (c_int, [P, P, c_int, c_int, c_ll, P, P, P, P]),
    'seedhip_conv2d_fwd': (c_int, [ctypes.POINTER(ConvGeom), P, c_int, c_int, P, P, P, c_int, P, P]),
    'seedhip_conv2d_bwd_data': (c_int, [ctypes.POINTER(ConvGeom), P, P, P, P, P, P]),
    'seedhip_conv2d_pipe': (c_int, [ctypes.POINTER(ConvGeom), c_int]),
    'seedhip_conv2d_fwd_workspace_bytes': (c_size_t, [ctypes.POINTER(ConvGeom)]),
    'seedhip_conv2d_fwd_ws': (c_int, [ctypes.POINTER(ConvGeom), P, c_int, c_int, P, P, P, c_int, P, P, c_size_t, P]),
    'seedhip_conv2d_bwd_data_workspace_bytes': (c_size_t, [ctypes.POINTER(ConvGeom)]),
    'seedhip_conv2d_bwd_data_ws': (c_int, [ctypes.POINTER(ConvGeom), P, P, P, P, P, P, c_size_t, P]),
    'seedhip_conv2d_bwd_weight_workspace_bytes': (c_size_t, [ctypes.POINTER(ConvGeom)]),
    'seedhip_conv2d_bwd_weight':
        (c_int, [ctypes.POINTER(ConvGeom), P, c_int, c_int, P, P, P, P, c_size_t, P]),
    'seedhip_conv2d_stack_fwd': (c_int, [ctypes.POINTER(StackConvGeom), P, P, P, P, P, c_int, P]),
    'seedhip_conv2d_stack_fwd_bits_supported': (c_int, [ctypes.POINTER(StackConvGeom)]),
    'seedhip_conv2d_stack_fwd_bits': (c_int, [ctypes.POINTER(StackConvGeom), P, P, P, P, P, P, P]),
    'seedhip_conv2d_fwd_bits_supported': (c_int, [ctypes.POINTER(ConvGeom)]),
    'seedhip_conv2d_fwd_bits': (c_int, [ctypes.POINTER(ConvGeom), P, c_int, c_int, P, P, P, P, P]),
    'seedhip_conv2d_bwd_data_bits_supported': (c_int, [ctypes.POINTER(ConvGeom)]),
    'seedhip_conv2d_bwd_data_bits': (c_int, [ctypes.POINTER(ConvGeom), P, P, P, P, P]),
    'seedhip_conv2d_fwd_outbits_supported': (c_int, [ctypes.POINTER(ConvGeom)]),
    'seedhip_conv2d_fwd_outbits': (c_int, [ctypes.POINTER(ConvGeom), P, c_int, P, P, P, P, P, P]),
    'seedhip_conv2d_bwd_data_bits_add': (c_int, [ctypes.POINTER(ConvGeom), P, P, P, P, P, P]),
    'seedhip_conv2d_bwd_data_pool_supported': (c_int, [ctypes.POINTER(ConvGeom)]),
    'seedhip_conv2d_bwd_data_pool': (c_int, [ctypes.POINTER(ConvGeom), P, P, P, P, P, P]),
    'seedhip_conv2d_stack_bwd_weight_workspace_bytes': (c_size_t, [ctypes.POINTER(StackConvGeom)]),
    'seedhip_conv2d_stack_bwd_weight':
        (c_int, [ctypes.POINTER(StackConvGeom), P, P, P, P, P, P, c_size_t, P]),
    'seedhip_maxpool3x3s2_same_fwd': (c_int, [c_int, c_int, c_int, c_int, P, P, P, P]),
    'seedhip_maxpool3x3s2_same_fwd_bits': (c_int, [c_int, c_int, c_int, c_int, P, P, P, P, P]),
    'seedhip_maxpool3x3s2_same_bwd': (c_int, [c_int, c_int, c_int, c_int, P, P, P, P]),
    'seedhip_conv3x3_u8_pool_fwd': (c_int, [P, c_int, c_int, c_int, c_int, P, P, c_int, P, P, P]),
    'seedhip_conv3x3_u8_pool_fwd_bits': (c_int, [P, c_int, c_int, c_int, c_int, P, P, c_int, P, P, P, P]),
    'seedhip_conv3x3_u8_pool_bwd_workspace_bytes': (c_size_t, [c_int, c_int, c_int]),
    'seedhip_conv3x3_u8_pool_bwd': (c_int, [P, c_int, c_int, c_int, c_int, P, P, c_int, P, P, P, c_size_t, P]),
    'seedhip_lstm_assemble_inputs': (c_int, [P, c_int, c_int, c_int, P, P, c_int, c_int, c_ll, P]),
    'seedhip_lstm_mask_state': (c_int, [P, P, P, c_int, c_int, P, P, P]),
    'seedhip_lstm_gates_fwd': (c_int, [P, P, P, c_int, c_int, P, c_int, P, P, P]),
    'seedhip_lstm_permute_u': (c_int, [P, c_int, P, P]),
    'seedhip_lstm_step_supported': (c_int, [c_int, c_int]),
    'seedhip_lstm_step_fwd': (c_int, [P, P, P, P, P, c_int, c_int, P, P, c_int, P, P, P]),
    'seedhip_lstm_seq_supported': (c_int, [c_int, c_int, c_int]),
    'seedhip_lstm_seq_fwd': (c_int, [P, P, P, c_int, c_int, c_int, P, P, c_int, P, P, P, P]),
    'seedhip_lstm_seq_fwd_sticky': (c_int, [P, P, P, c_int, c_int, c_int, P, P, c_int, P, P, P, P, P]),
    'seedhip_lstm_seq_bwd_workspace_bytes': (c_size_t, [c_int, c_int]),
    'seedhip_lstm_seq_bwd': (c_int, [P, P, P, P, c_int, P, c_int, c_int, c_int, P, P, P, P]),
    'seedhip_lstm_seq_bwd_sticky': (c_int, [P, P, P, P, c_int, P, c_int, c_int, c_int, P, P, P, P, P]),
    'seedhip_lstm_gates_bwd': (c_int, [P, P, P, c_int, P, P, P, c_int, c_int, P, P, P]),
    'seedhip_rows_move': (c_int, [P, P, P, P, c_ll, c_ll, P]),
    'seedhip_rows_move_masked': (c_int, [P, P, P, P, c_ll, c_ll, P, c_int, P]),
    'seedhip_rows_move_multi': (c_int, [c_int, P, P, P, P, P, c_ll, P, c_int, P]),
    'seedhip_inference_pre': (c_int, [P, P, P, P, P, c_int, c_int, c_int, P, P, P, P, P, P, P, P, P, c_int, P, P,
                                      P, P, P, P, P, c_int, P]),
    'seedhip_inference_post': (c_int, [P, P, P, P, c_int, c_int, P, c_int, c_int, c_int, c_int, P, P, P, P, P, P, P, P,
                                       P, P, P, P, P, P]),
    'seedhip_emit_unrolls': (c_int, [c_int, P, P, P, P, P, P, c_int, c_int, c_int, c_int, P]),
    'seedhip_categorical_sample': (c_int, [P, c_int, c_ll, c_int, P, P, P]),
    'seedhip_rows_move_ops': (c_int, [c_int, P, P]),
    'seedhip_serve_conv0_split_bytes': (c_size_t, [c_int]),
    'seedhip_serve_split_conv0': (c_int, [P, c_int, P, P]),
    'seedhip_serve_heads_image_bytes': (c_size_t, [c_int]),
    'seedhip_serve_begin': (c_int, [ctypes.POINTER(ServeStep), P, c_int, P, P, c_int, c_int, P, P]),
    'seedhip_conv2d_stack_fwd_rows_supported': (c_int, [ctypes.POINTER(StackConvGeom)]),
    'seedhip_conv2d_stack_fwd_rows': (c_int, [ctypes.POINTER(StackConvGeom), P, P, P, P, P, P, P, c_int, P]),
    'seedhip_dense_fwd_partial_workspace_bytes': (c_size_t, [ctypes.POINTER(ConvGeom)]),
    'seedhip_dense_fwd_partial': (c_int, [ctypes.POINTER(ConvGeom), P, c_int, P, P, c_size_t, ctypes.POINTER(c_int), P]),
    'seedhip_serve_finish': (c_int, [ctypes.POINTER(ServeStep), ctypes.POINTER(ServeFields), P, c_int, P, c_int, P, P,
                                     c_int, c_int, P, P, P, c_ll, P]),
    'seedhip_serve_emit': (c_int, [ctypes.POINTER(ServeStep), c_int, P, P, P, P, P, P, c_ll, P]),
    'seedhip_replay_sample_workspace_bytes': (c_size_t, [c_ll]),
    'seedhip_replay_sample': (c_int, [P, c_ll, c_float, c_float, P, c_int, P, P, P, c_size_t, P]),
    'seedhip_heads_supported': (c_int, [c_int, c_int]),
    'seedhip_heads_fwd': (c_int, [P, c_int, P, P, c_ll, c_int, c_int, P, P]),
    'seedhip_epsilon_greedy': (c_int, [P, P, P, c_int, c_int, c_int, P, P, P]),
    'seedhip_replay_time_rows': (c_int, [P, c_int, c_int, P, P, P]),
    'seedhip_dueling_fwd': (c_int, [P, c_int, c_ll, c_int, P, P, P]),
    'seedhip_dueling_bwd': (c_int, [P, c_ll, c_int, P, c_int, P]),
    'seedhip_r2d2_loss_workspace_bytes': (c_size_t, [c_int, c_int, c_int]),
    'seedhip_r2d2_loss_fwd_bwd':
        (c_int, [P, P, P, P, P, P, c_int, c_int, c_int, c_float, c_int, c_float, c_float, c_float,
                 P, P, P, P, P, c_size_t, P]),
}


class SeedHipError(RuntimeError):
  pass


_lib = None


def lib():
  """Loads libseedhip.so once; raises if it has not been built."""
  global _lib
  if _lib is None:
    if not os.path.exists(LIB_PATH):
      raise SeedHipError(
          'libseedhip.so not built (%s). Run `python -m seed_rl_amd.build`. '
          'There is no CPU fallback for the HIP hot path.' % LIB_PATH)
    l = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
      fn = getattr(l, name)            # AttributeError if the symbol is missing
      fn.restype = res
      fn.argtypes = args
    have = l.seedhip_abi_version()
    if have != ABI_VERSION:
      raise SeedHipError('%s reports ABI version %d but this binding was written against %d: the library is stale '
                         '(run `python -m seed_rl_amd.build`); calling it would pass shifted arguments.'
                         % (LIB_PATH, have, ABI_VERSION))
    _lib = l
  return _lib


def check(rc, what=''):
  if rc != 0:
    msg = lib().seedhip_last_error()
    raise SeedHipError('%s failed (rc=%d): %s' % (what, rc, msg.decode() if msg else ''))


def ptr(t):
  """Device pointer of a torch tensor (None -> NULL)."""
  return None if t is None else c_void_p(t.data_ptr())


def stream():
  """The current torch HIP stream as a hipStream_t (void*)."""
  return c_void_p(torch.cuda.current_stream().cuda_stream)


def require_cuda(*tensors):
  for t in tensors:
    if t is not None and not t.is_cuda:
      raise SeedHipError('seed_rl_amd kernels need device tensors (got a %s tensor); '
                         'there is no CPU fallback.' % t.device)
    if t is not None and not t.is_contiguous():
      raise SeedHipError('seed_rl_amd kernels need contiguous tensors')

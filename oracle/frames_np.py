"""NumPy restatement of learner-side frame stacking and the done-aware RNN
unroll (oracle; test infrastructure only).

Follows /root/reference/atari/networks.py:32-54 (initial state), :57-173
(stack_frames, bit-packed int32 state, LSB = oldest) and :176-218
(_unroll_cell).  Pinned by atari/networks_test.py:119-247.
"""
import numpy as np


def initial_frame_stacking_state(stack_size, batch_size, observation_shape):
  """networks.py:32-54."""
  if stack_size == 1:
    return ()
  return np.zeros([batch_size, int(np.prod(observation_shape))], np.int32)


def stack_frames(frames, frame_stacking_state, done, stack_size):
  """networks.py:57-173.

  frames: [T,B,*obs,1] un-normalised (any numeric dtype; values 0..255).
  frame_stacking_state: int32[B, prod(obs)] (or () when stack_size == 1).
  done: bool[T,B].
  Returns (float32[T,B,*obs,stack_size] newest->oldest, new int32 state).
  """
  frames = np.asarray(frames)
  done = np.asarray(done).astype(bool)
  if frames.shape[0:2] != done.shape[0:2]:
    raise ValueError('Expected same first 2 dims for frames and dones. '
                     'Got {} vs {}.'.format(frames.shape[0:2], done.shape[0:2]))
  if stack_size > 4:
    raise ValueError('Only up to stack size 4 is supported due to bit-packing.')
  if stack_size > 1 and frames.shape[-1] != 1:
    raise ValueError('Due to frame stacking, we require last observation '
                     'dimension to be 1. Got {}'.format(frames.shape[-1]))
  if stack_size == 1:
    return frames.astype(np.float32), ()
  state = np.asarray(frame_stacking_state)
  if state.dtype != np.int32:
    raise ValueError('Expected dtype int32 got {}'.format(state.dtype))
  T, batch_size = frames.shape[0:2]
  obs_shape = frames.shape[2:-1]
  state = state.reshape((batch_size,) + tuple(obs_shape))

  unstacked = [((state >> (8 * i)) & 0xFF).astype(np.float32)     # :102-108
               for i in range(stack_size - 1)]
  extended = np.concatenate(
      [u.reshape((1,) + u.shape + (1,)) for u in unstacked] +
      [frames.astype(np.float32)], axis=0)                        # :113-117
  stacked = np.concatenate(
      [extended[stack_size - 1 - i:extended.shape[0] - i]
       for i in range(stack_size)], axis=-1)                      # :123-126

  row_shape = (T, batch_size) + (1,) * (frames.ndim - 2)
  done_masks = [np.zeros(row_shape, bool), done.reshape(row_shape)]  # :131-135
  while len(done_masks) < stack_size:
    prev = done_masks[-1]
    shifted = np.concatenate([np.zeros_like(prev[:1]), prev[:-1]], axis=0)
    done_masks.append(np.logical_or(prev, shifted))               # :136-143
  stacked_done = np.concatenate(done_masks, axis=-1)
  stacked = np.where(stacked_done, np.float32(0), stacked)        # :154-157

  last = stacked[-1, ..., :-1].astype(np.int32)                   # :164-169
  shifts = np.array([8 * i for i in range(stack_size - 2, -1, -1)], np.int32)
  new_state = np.sum(last << shifts, axis=-1, dtype=np.int32)
  new_state = new_state.reshape(batch_size, int(np.prod(obs_shape)))
  return stacked.astype(np.float32), new_state


def unroll_cell(inputs, done, start_state, zero_state, recurrent_cell):
  """networks.py:176-218 (state = tuple of [B,...] arrays)."""
  inputs = np.asarray(inputs)
  done = np.asarray(done).astype(bool)
  assert inputs.shape[0] == done.shape[0]
  state = tuple(np.asarray(s) for s in start_state)
  zero_state = tuple(np.asarray(z) for z in zero_state)
  outs = []
  for t in range(inputs.shape[0]):
    d = done[t]
    state = tuple(
        np.where(d.reshape((d.shape[0],) + (1,) * (y.ndim - 1)), x, y)
        for x, y in zip(zero_state, state))
    o, state = recurrent_cell(inputs[t], state)
    outs.append(o)
  return np.stack(outs), state

"""Argument checks of the host wrappers that run before any kernel is launched (no GPU): the byte-mask variants of the
conv calls refuse combinations the kernels do not implement instead of silently ignoring an operand."""
import pytest
import torch


def _g():
  from seed_rl_amd import ops
  return ops, ops.conv_geom(512, 18, 24, 32, 3, 3, 1, 'same', 32)


def test_conv_fwd_byte_mask_arguments():
  ops, g = _g()
  x = torch.zeros((512, 18, 24, 32)); w = torch.zeros((3, 3, 32, 32)); out = torch.zeros_like(x)
  bits = torch.zeros((512, 18, 24, 8), dtype=torch.uint8)
  with pytest.raises(ValueError):                            # out_bits: the sign of a pre-activation output
    ops.conv2d_fwd(g, x, w, None, out, in_relu=True, out_relu=True, out_bits=bits)
  with pytest.raises(ValueError):                            # relu_bits and out_bits are different kernels
    ops.conv2d_fwd(g, x, w, None, out, out_relu=True, relu_bits=bits, out_bits=bits)
  with pytest.raises(ValueError):                            # relu_bits: ReLU'd output, no residual
    ops.conv2d_fwd(g, x, w, None, out, out_relu=False, relu_bits=bits)
  with pytest.raises(ValueError):
    ops.conv2d_fwd(g, x, w, None, out, out_relu=True, residual=x, relu_bits=bits)
  with pytest.raises(ValueError):                            # uint8 input is the first layer's business
    ops.conv2d_fwd(g, x, w, None, out, in_dtype=ops.IN_U8_DIV255, out_bits=bits)


def test_conv_dgrad_byte_mask_arguments():
  ops, g = _g()
  dy = torch.zeros((512, 18, 24, 32)); w = torch.zeros((3, 3, 32, 32)); dx = torch.zeros_like(dy)
  bits = torch.zeros((512, 18, 24, 8), dtype=torch.uint8)
  with pytest.raises(ValueError):                            # one mask, not two
    ops.conv2d_bwd_data(g, dy, w, dx, relu_mask=dy, relu_bits=bits)

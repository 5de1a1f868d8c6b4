"""How fast are the library fp32 GEMMs (rocBLAS / hipBLASLt through torch.mm) at the cfg2 / cfg3 / cfg5 Dense shapes?
A yardstick for csrc/gemm.h (not a product path)."""
import torch
torch.backends.cuda.matmul.allow_tf32 = False
dev = torch.device('cuda')


def t(fn, reps=20):
  for _ in range(3):
    fn()
  s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  s.record()
  for _ in range(reps):
    fn()
  e.record(); torch.cuda.synchronize()
  return s.elapsed_time(e) / reps * 1e3


for name, M, K, N in [('atari fc', 10752, 2592, 256), ('r2d2 fc', 121 * 256, 3136, 512), ('deep fc', 21 * 256, 3456, 256),
                      ('r2d2 lstm-x', 121 * 256, 532, 2048)]:
  x = torch.randn(M, K, device=dev); w = torch.randn(K, N, device=dev); dy = torch.randn(M, N, device=dev)
  fl = 2.0 * M * K * N
  for what, fn in (('fwd  x@W', lambda: torch.mm(x, w)), ('dgrad dy@W^T', lambda: torch.mm(dy, w.t())),
                   ('wgrad x^T@dy', lambda: torch.mm(x.t(), dy))):
    us = t(fn)
    print('%-12s %-14s M=%d K=%d N=%d  %8.1f us  %6.1f TF/s' % (name, what, M, K, N, us, fl / us / 1e6))

import sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tools')
from seed_rl_amd import ops
from bench_kernels import timeit
dev = torch.device('cuda')
n = 10752
g = ops.conv_geom(n, 20, 20, 16, 4, 4, 2, 'valid', 32)
for name, mk in (('randn', torch.randn), ('zeros', lambda *a, **k: torch.zeros(*a, **k)), ('ones', lambda *a, **k: torch.ones(*a, **k))):
  x = mk((n, 20, 20, 16), device=dev); w = mk((4, 4, 16, 32), device=dev); b = mk((32,), device=dev)
  out = torch.empty((n, 9, 9, 32), device=dev); dy = mk((n, 9, 9, 32), device=dev)
  dw = torch.empty_like(w); db = torch.empty_like(b); dx = torch.empty_like(x)
  ws = torch.empty(ops.conv2d_bwd_weight_workspace_bytes(g) // 4 + 4, device=dev)
  fl = 2.0 * n * 81 * 256 * 32
  t1 = timeit(lambda: ops.conv2d_fwd(g, x, w, b, out, out_relu=True))
  t2 = timeit(lambda: ops.conv2d_bwd_weight(g, x, dy, dw, db, ws))
  t3 = timeit(lambda: ops.conv2d_bwd_data(g, dy, w, dx, relu_mask=x))
  print('%-6s fwd %.1f us (%.0f TF)  wgrad %.1f us (%.0f TF)  dgrad %.1f us (%.0f TF)' % (name, t1 * 1e3, fl / t1 / 1e9, t2 * 1e3, fl / t2 / 1e9, t3 * 1e3, fl / t3 / 1e9))

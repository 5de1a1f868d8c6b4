import os, sys, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from seed_rl_amd import _lib
dev = torch.device('cuda')
for B in (1 << 17, 1 << 18, 1 << 19, 1 << 20, 1 << 21, 1 << 22):
  T = 20
  t = [torch.rand((T, B), device=dev) for _ in range(5)]
  boot = torch.rand(B, device=dev)
  vs, pg = torch.empty_like(t[0]), torch.empty_like(t[0])
  def run():
    _lib.lib().seedhip_vtrace_from_importance_weights(_lib.ptr(t[0]), _lib.ptr(t[1]), _lib.ptr(t[2]), _lib.ptr(t[3]), _lib.ptr(t[4]), _lib.ptr(boot), 1.0, 1.0, 1.0, T, B, _lib.ptr(vs), _lib.ptr(pg), _lib.stream())
  for _ in range(5): run()
  s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  s.record()
  for _ in range(30): run()
  e.record(); torch.cuda.synchronize()
  us = s.elapsed_time(e) * 1e3 / 30
  print(os.environ.get('SEEDHIP_VTRACE_VARIANT', '-'), os.environ.get('SEEDHIP_VTRACE_CHUNK', '-'), B, round(us, 1), 'us', round((T * B * 28 + B * 4) / us / 1e3), 'GB/s')

"""The bench line's bookkeeping that needs no GPU: the whole-step roofline (VERDICT r3 task 4) and the kernel-source
digest that decides whether a committed PMC profile still describes this tree's kernels."""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_step_roofline_sums_per_kernel_floors():
  bench = importlib.import_module('bench')
  kern = {
      # 100 MB at 6.3 TB/s = 15.9 us > 1 GF at 157.3 TF/s = 6.4 us: HBM-bound, launched twice
      'a': dict(calls=2, total_ms=0.08, avg_ms=0.04, flops=1e9, bytes=100e6, pipe='f32'),
      # 14.27 GF on the fp32 pipe = 90.7 us > 30 MB
      'b': dict(calls=1, total_ms=0.15, avg_ms=0.15, flops=14.27e9, bytes=30e6, pipe='f32'),
      # the same flops on bf16x6: 14.27 / 416.7 = 34.2 us
      'c': dict(calls=1, total_ms=0.09, avg_ms=0.09, flops=14.27e9, bytes=30e6, pipe='bf16x6'),
      # bf16x3: 35.2 GF / 833 TF/s = 42.3 us < 351 MB / 6.3 TB/s = 55.7 us
      'd': dict(calls=1, total_ms=0.13, avg_ms=0.13, flops=35.2e9, bytes=351e6, pipe='bf16x3'),
  }
  r = bench.step_roofline(kern, 0.5)
  want = 2 * 100e6 / 6.3e12 * 1e3 + 14.27e9 / 157.3e12 * 1e3 + 14.27e9 / (2500e12 / 6) * 1e3 + 351e6 / 6.3e12 * 1e3
  assert abs(r['floor_ms'] - want) < 1e-3, (r['floor_ms'], want)
  assert abs(r['frac'] - want / 0.5) < 1e-3
  assert r['kernels']['a']['bound'] == 'hbm' and r['kernels']['b']['bound'] == 'mfma' and r['kernels']['d']['bound'] == 'hbm'


def test_kernel_source_digest_tracks_the_sources(tmp_path, monkeypatch):
  from seed_rl_amd import build
  d0 = build.csrc_digest()
  assert len(d0) == 64 and d0 == build.csrc_digest()          # stable
  # a changed byte in any kernel source changes the digest (checked on a copy of the directory)
  import shutil
  src = os.path.join(ROOT, 'seed_rl_amd', 'csrc')
  dst = tmp_path / 'pkg' / 'csrc'
  shutil.copytree(src, dst, ignore=shutil.ignore_patterns('serve'))
  monkeypatch.setattr(build, '_HERE', str(tmp_path / 'pkg'))
  d1 = build.csrc_digest()
  with open(dst / 'adam.hip', 'a') as f:
    f.write('// touched\n')
  assert build.csrc_digest() != d1

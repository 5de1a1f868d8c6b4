"""The C-ABI library loads (no GPU needed) and exports every symbol include/seedhip.h
declares; the ctypes signature table covers exactly that set."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
  src = open(os.path.join(ROOT, 'include', 'seedhip.h')).read()
  src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
  return sorted(set(re.findall(r'\b(seedhip_[a-z0-9_]+)\s*\(', src)))


@pytest.fixture(scope='module')
def built_lib():
  from seed_rl_amd import build, _lib
  build.build()
  return ctypes.CDLL(_lib.LIB_PATH)


def test_header_symbols_exported(built_lib):
  syms = declared_symbols()
  assert len(syms) >= 15
  for s in syms:
    assert hasattr(built_lib, s), 'libseedhip.so does not export %s' % s


def test_signature_table_matches_header(built_lib):
  from seed_rl_amd import _lib
  assert sorted(_lib.SIGNATURES) == declared_symbols()
  l = _lib.lib()
  assert l.seedhip_abi_version() == _lib.ABI_VERSION == 6
  assert l.seedhip_impala_loss_workspace_bytes(20, 512) == (512 // 2) * 8 * 4
  assert l.seedhip_global_norm_workspace_bytes() > 0


def test_argument_validation_without_gpu(built_lib):
  """Shape validation happens before any launch, so it is testable on CPU."""
  from seed_rl_amd import _lib
  l = _lib.lib()
  rc = l.seedhip_vtrace_from_importance_weights(None, None, None, None, None, None, 1.0, 1.0, 1.0, 5, 4, None, None, None)
  assert rc == -1 and b'null pointer' in l.seedhip_last_error()
  rc = l.seedhip_vtrace_from_importance_weights(None, None, None, None, None, None, 1.0, 1.0, 1.0, -1, 4, None, None, None)
  assert rc == -1 and b'negative' in l.seedhip_last_error()
  assert l.seedhip_vtrace_from_importance_weights(None, None, None, None, None, None, 1.0, 1.0, 1.0, 0, 4, None, None, None) == 0
  g = _lib.ConvGeom(1, 4, 4, 4, 4, 4, 3, 3, 1, 1, 1, 8, 2, 8)   # ld_in < cin
  rc = l.seedhip_conv2d_fwd(ctypes.byref(g), None, 0, 0, None, None, None, 0, None, None)
  assert rc == -1 and b'ld_in' in l.seedhip_last_error()


def test_product_does_not_import_oracle():
  """The shipped package must never route through the CPU oracle."""
  pkg = os.path.join(ROOT, 'seed_rl_amd')
  for dp, _, fs in os.walk(pkg):
    for f in fs:
      if f.endswith(('.py', '.hip', '.h', '.cpp')):
        txt = open(os.path.join(dp, f)).read()
        assert 'import oracle' not in txt and 'from oracle' not in txt, f


def test_no_cpu_fallback():
  import torch
  from seed_rl_amd import _lib, vtrace
  z = torch.zeros(3, 2)
  with pytest.raises(_lib.SeedHipError):
    vtrace.from_importance_weights(z, z, z, z, z, torch.zeros(2))

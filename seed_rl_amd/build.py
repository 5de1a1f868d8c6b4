"""In-tree build of libseedhip.so for gfx950 (hipcc cross-compiles without a GPU)."""
import os
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))


def build(verbose=False, jobs=None):
  jobs = jobs or os.cpu_count() or 4
  cmd = ['make', '-C', os.path.join(_HERE, 'csrc'), '-j%d' % jobs]
  res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
  if verbose or res.returncode != 0:
    sys.stdout.write(res.stdout)
  if res.returncode != 0:
    raise RuntimeError('libseedhip.so build failed')
  return os.path.join(_HERE, 'lib', 'libseedhip.so')


if __name__ == '__main__':
  print(build(verbose=True))

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


def csrc_digest():
  """sha256 over the HIP sources libseedhip.so is built from (names + contents, sorted): what a committed hardware
  profile is stamped with, so that bench.py can tell whether `profiles/*_traffic.json` still describes HEAD's kernels."""
  import hashlib
  h = hashlib.sha256()
  d = os.path.join(_HERE, 'csrc')
  for name in sorted(os.listdir(d)):
    if name.endswith(('.hip', '.h', '.cpp')):
      h.update(name.encode())
      with open(os.path.join(d, name), 'rb') as f:
        h.update(f.read())
  return h.hexdigest()


if __name__ == '__main__':
  if len(sys.argv) > 1 and sys.argv[1] == 'digest':
    print(csrc_digest())
  else:
    print(build(verbose=True))

"""`python bench.py --gpus N` must start N ranks by itself, run them on the same launch mode as N=1 (HIP graphs; for
N > 1 graph segments with the gradient exchange between them, reference semantics agents/vtrace/learner.py:249-275)
and refuse to print an N-rank line from fewer devices.  The RCCL path itself needs N GPUs; what
runs here is the dry run over gloo on the one device, which walks the same code: self-launch under
torch.distributed.run, sharded columns, GraphedStep's segmented capture, overlapped range exchange, update graph,
per-rank times, exposed-exchange measurement."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*args, timeout=600):
  env = dict(os.environ)
  env.pop('RANK', None); env.pop('WORLD_SIZE', None); env.pop('LOCAL_RANK', None)
  return subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + list(args), cwd=ROOT, env=env,
                        stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout, universal_newlines=True)


def _json_lines(out):
  return [json.loads(l) for l in out.splitlines() if l.startswith('{') and '"metric"' in l]


def test_bench_two_ranks_gloo_dry_run(device):
  r = _run('--gpus', '2', '--backend', 'gloo', '--steps', '3', '--warmup', '3', '--quick', '--batch', '32')
  assert r.returncode == 0, r.stderr[-2000:]
  lines = _json_lines(r.stdout)
  assert len(lines) == 1, r.stdout[-2000:]                      # rank 0 prints ONE line
  line = lines[0]
  ex = line['exchange']
  assert ex['ranks'] == 2 and ex['backend'] == 'gloo'
  assert len(ex['per_rank_ms_per_step']) == 2 and all(t > 0 for t in ex['per_rank_ms_per_step'])
  assert line['config']['launch'].startswith('hip-graph x2 segments'), line['config']['launch']
  assert line['config']['global_batch'] == 64 and line['config']['parallelism'] == 'dp2'
  # the Dense + heads range (the tail of the bucket) is exchanged under the conv backward
  (lo, hi), = ex['overlapped_ranges']
  assert 0 < lo < hi and hi * 4 == ex['bucket_bytes']
  assert ex['exposed_ms_per_step'] is not None
  assert line['value'] > 0 and line['ms_per_step'] >= max(ex['per_rank_ms_per_step']) - 1e-3
  if torch.cuda.device_count() < 2:
    assert line['n_gpus'] == 1 and 'dry_run' in line


def test_bench_refuses_more_ranks_than_devices(device):
  n = torch.cuda.device_count() + 1
  r = _run('--gpus', str(n), '--quick', '--steps', '2', '--warmup', '1', timeout=120)
  assert r.returncode != 0
  assert 'GPU' in r.stderr and not _json_lines(r.stdout)

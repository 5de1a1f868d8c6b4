#!/bin/bash
# A/B of the in-kernel split's h part: truncation (lib/) against round-to-nearest-even (lib_rne/, -DSEEDHIP_SPLIT_H_RNE=1),
# on the per-tensor fp64 gate of the full-size cfg3 step and on the step times.  Run through gpurun from the repo root.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/ab_rne; mkdir -p $OUT
for v in rne trunc; do
  if [ $v = rne ]; then export SEEDHIP_LIB=$R/seed_rl_amd/lib_rne/libseedhip.so; else unset SEEDHIP_LIB; fi
  rm -f $R/gpurun_out/fullsize_parity.jsonl
  python -m pytest tests/test_gpu_fullsize.py -q -k "cfg3_dmlab_T20_B256" > $OUT/$v.pytest.log 2>&1
  tail -3 $OUT/$v.pytest.log
  cp $R/gpurun_out/fullsize_parity.jsonl $OUT/$v.parity.jsonl 2>/dev/null
  python bench.py --quick 2>/dev/null | tail -1 > $OUT/$v.cfg2.json
  python bench.py --quick --config dmlab 2>/dev/null | tail -1 > $OUT/$v.cfg3.json
  python bench.py --quick --config r2d2 2>/dev/null | tail -1 > $OUT/$v.cfg5.json
done
python - <<PY
import json
for v in ('rne', 'trunc'):
  r = json.loads(open('$OUT/%s.parity.jsonl' % v).read().strip().split('\n')[-1])
  print(v, {k: r.get(k) for k in ('grad_q99_gate_vs_fp64', 'grad_q99_gate_bias_vs_fp64', 'grad_q99_rel_err_vs_fp64', 'oracle_grad_q99_rel_err_vs_fp64', 'grad_max_rel_err_vs_fp64')})
  for c in ('cfg2', 'cfg3', 'cfg5'):
    d = json.loads(open('$OUT/%s.%s.json' % (v, c)).read())
    print('   ', c, d['ms_per_step'], d['windows']['median'], d['windows']['min'])
PY

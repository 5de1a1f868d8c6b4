#!/bin/bash
# rocprofv3 kernel trace of the Dense A/B tool (run through gpurun from the repo root): which launches a Dense call is made of
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_x8
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
export PYTHONPATH=$R
rocprofv3 --kernel-trace --stats -d $OUT -o x8 --output-format csv -- python $R/tools/bench_x6.py --child ${1:-atari} > $OUT/x8.log 2>&1
python - <<P
import csv
rows=list(csv.DictReader(open('$OUT/x8_kernel_stats.csv')))
for r in rows[:14]:
  print('%-120s calls %5s avg %9.1f us' % (r['Name'][:120], r['Calls'], float(r['AverageNs'])/1e3))
P

#!/bin/bash
# Per-kernel time of the central inference step (HIP-graph replay of one batch): rocprofv3 --kernel-trace --stats.
#   tools/prof_inference.sh <tag> [n] [envs] [extra env assignments...]
TAG=${1:-inf}; N=${2:-1024}; ENVS=${3:-4096}
export TMPDIR=/tmp
OUT=gpurun_out/prof/$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o $TAG -- python tools/bench_inference.py --n $N --envs $ENVS --mode packed --calls 400 > $OUT/$TAG.log 2>&1
grep "packed" $OUT/$TAG.log
F=$(find $OUT -name "*kernel_stats.csv" | head -1)
python - "$F" <<PY
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:24]:
  print("%-100s %6s %9.1f us %5s%%" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY

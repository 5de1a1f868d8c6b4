#!/bin/bash
# Kernel-trace profile of the cfg3 (and cfg5) steps + the unprofiled bench lines of the same build:
#   gpurun -- 'bash tools/prof_cfg3.sh r02d'   then   python tools/prof_summary.py gpurun_out/prof r02d_cfg3 profiles/r02d_cfg3
set -u
TAG=${1:-r02x}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/prof; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --steps 5 --warmup 3 --quick --graph 0"
rocprofv3 --kernel-trace --stats -d $OUT -o ${TAG}_cfg3 --output-format csv -- $B --config dmlab > $OUT/${TAG}_cfg3.log 2>&1
rocprofv3 --kernel-trace --stats -d $OUT -o ${TAG}_cfg5 --output-format csv -- $B --config r2d2 > $OUT/${TAG}_cfg5.log 2>&1
python $R/bench.py 2>/dev/null | tail -1 > $OUT/${TAG}_cfg2_bench.json
python $R/bench.py --quick --config dmlab 2>/dev/null | tail -1 > $OUT/${TAG}_cfg3_bench.json
python $R/bench.py --quick --config r2d2 2>/dev/null | tail -1 > $OUT/${TAG}_cfg5_bench.json
ls $OUT | grep ${TAG}

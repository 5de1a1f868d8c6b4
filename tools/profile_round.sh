#!/bin/bash
# One profiling round on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh r03 [nobench]
# Writes rocprofv3 output under gpurun_out/prof; condense afterwards (here, CPU side) with
#   python tools/prof_summary.py gpurun_out/prof <tag>_cfg2 profiles/<tag>_cfg2   (+ cfg3, cfg5)
#   python tools/make_traffic.py profiles/<tag>_cfg2_pmc.csv profiles/<tag>_cfg2_traffic.json
# PMC passes are separate runs with --kernel-trace only (no sys/hip/hsa tracing next to --pmc).
set -u
TAG=${1:-r01x}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof
mkdir -p $OUT
python -m seed_rl_amd.build digest > $OUT/${TAG}_cfg2_csrc.sha256     # which kernels this profile describes
cd /tmp; export TMPDIR=/tmp
export PYTHONPATH=$R
B="python $R/bench.py --steps 5 --warmup 3 --quick --graph 0"
rocprofv3 --kernel-trace --stats -d $OUT -o ${TAG}_cfg2 --output-format csv -- $B > $OUT/${TAG}_cfg2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT -o ${TAG}_cfg2_fetch --output-format csv -- $B > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT -o ${TAG}_cfg2_write --output-format csv -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d $OUT -o ${TAG}_cfg3 --output-format csv -- $B --config dmlab > $OUT/${TAG}_cfg3.log 2>&1
cp $OUT/${TAG}_cfg2_csrc.sha256 $OUT/${TAG}_cfg3_csrc.sha256
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT -o ${TAG}_cfg3_fetch --output-format csv -- $B --config dmlab > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT -o ${TAG}_cfg3_write --output-format csv -- $B --config dmlab > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d $OUT -o ${TAG}_cfg5 --output-format csv -- $B --config r2d2 > $OUT/${TAG}_cfg5.log 2>&1
if [ "${2:-}" != "nobench" ]; then
  # the unprofiled bench lines of the same build (HIP-graph launch, default K / W)
  python $R/bench.py 2>/dev/null | tail -1 > $OUT/${TAG}_cfg2_bench.json
  python $R/bench.py --quick --config dmlab 2>/dev/null | tail -1 > $OUT/${TAG}_cfg3_bench.json
  python $R/bench.py --quick --config r2d2 2>/dev/null | tail -1 > $OUT/${TAG}_cfg5_bench.json
fi
ls -la $OUT | head -40

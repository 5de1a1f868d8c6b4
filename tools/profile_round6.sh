#!/bin/bash
# Round-6 profiling round in ONE gpurun (the committed kernel stats and the bench line of the same box):
#   tools/profile_round6.sh r06
# 1. the default bench line (HIP-graph launch, six windows) -> <tag>_cfg2_bench.json, cfg3 / cfg5 quick lines
# 2. rocprofv3 --kernel-trace --stats of cfg2 / cfg3 / cfg5 (eager launches: every kernel its own dispatch), PMC passes of
#    cfg2 (FETCH_SIZE, WRITE_SIZE, MFMA busy, instruction mix) in separate runs with --kernel-trace only
# 3. central inference: kernel stats of one 1 024-row call; the closed loop's kernel timeline (summary only)
set -u
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof; PMC=$R/gpurun_out/pmc
mkdir -p $OUT $PMC
python -m seed_rl_amd.build digest > $OUT/${TAG}_cfg2_csrc.sha256
cp $OUT/${TAG}_cfg2_csrc.sha256 $OUT/${TAG}_cfg3_csrc.sha256
python $R/bench.py 2>$OUT/${TAG}_bench.err | tail -1 > $OUT/${TAG}_cfg2_bench.json
python $R/bench.py --quick --config dmlab 2>/dev/null | tail -1 > $OUT/${TAG}_cfg3_bench.json
python $R/bench.py --quick --config r2d2 2>/dev/null | tail -1 > $OUT/${TAG}_cfg5_bench.json
cd /tmp; export TMPDIR=/tmp; export PYTHONPATH=$R
B="python $R/bench.py --steps 5 --warmup 3 --quick --graph 0"
rocprofv3 --kernel-trace --stats -d $OUT -o ${TAG}_cfg2 --output-format csv -- $B > $OUT/${TAG}_cfg2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT -o ${TAG}_cfg2_fetch --output-format csv -- $B > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT -o ${TAG}_cfg2_write --output-format csv -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d $OUT -o ${TAG}_cfg3 --output-format csv -- $B --config dmlab > $OUT/${TAG}_cfg3.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT -o ${TAG}_cfg3_fetch --output-format csv -- $B --config dmlab > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT -o ${TAG}_cfg3_write --output-format csv -- $B --config dmlab > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d $OUT -o ${TAG}_cfg5 --output-format csv -- $B --config r2d2 > $OUT/${TAG}_cfg5.log 2>&1
B3="python $R/bench.py --steps 3 --warmup 2 --quick --graph 0"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 -d $PMC -o ${TAG}_mfma --kernel-trace --output-format csv -- $B3 > $PMC/${TAG}_mfma.log 2>&1
rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES -d $PMC -o ${TAG}_mix --kernel-trace --output-format csv -- $B3 > $PMC/${TAG}_mix.log 2>&1
rocprofv3 --kernel-trace --stats -d $OUT -o ${TAG}_inf1024 --output-format csv -- python $R/tools/bench_inference.py --n 1024 --envs 4096 --mode packed --calls 400 > $OUT/${TAG}_inf1024.log 2>&1
mkdir -p $OUT/serve
rocprofv3 --kernel-trace --output-format csv -d $OUT/serve -o serve -- python $R/tools/bench_serving.py --mode inprocess --n 4096 --envs 16384 --seconds 2 > $OUT/${TAG}_serving.log 2>&1
python $R/tools/serving_timeline.py $(find $OUT/serve -name "*kernel_trace.csv" | head -1) > $OUT/${TAG}_serving_timeline.txt 2>&1
tail -1 $OUT/${TAG}_serving.log | cut -c1-1500 >> $OUT/${TAG}_serving_timeline.txt
rm -rf $OUT/serve
# the big traces are not needed on the way back (the condensed files are): keep the stats and counter tables only
find $OUT -name "*_kernel_trace.csv" -size +20M -delete
ls -la $OUT | head -60

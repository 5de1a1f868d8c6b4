#!/bin/bash
# MFMA-pipe utilisation per kernel: one rocprofv3 --pmc pass (kernel-trace only) over the cfg2 bench step.
#   tools/pmc_mfma.sh <tag>    ->  gpurun_out/pmc/<tag>_mfma_counter_collection.csv ; condense with tools/pmc_mfma.py
set -u
TAG=${1:-r02x}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --steps 3 --warmup 2 --quick --graph 0"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 -d $OUT -o ${TAG}_mfma --kernel-trace --output-format csv -- $B > $OUT/${TAG}_mfma.log 2>&1
rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES -d $OUT -o ${TAG}_mix --kernel-trace --output-format csv -- $B > $OUT/${TAG}_mix.log 2>&1
ls -la $OUT | grep $TAG

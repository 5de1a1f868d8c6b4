#!/bin/bash
# SQ counters of the kernels one bench_kernels.py invocation launches (two rocprofv3 --pmc passes, kernel-trace only).
#   tools/pmc_kernel.sh <tag> <bench_kernels.py arguments...>  ->  gpurun_out/pmc/<tag>_{lds,mix}_counter_collection.csv
#   condense with: python tools/pmc_kernel.py gpurun_out/pmc <tag> [kernel-name substring]
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
B="python $R/tools/bench_kernels.py $*"
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS -d $OUT -o ${TAG}_lds --kernel-trace --output-format csv -- $B > $OUT/${TAG}_lds.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU -d $OUT -o ${TAG}_mix --kernel-trace --output-format csv -- $B > $OUT/${TAG}_mix.log 2>&1
ls $OUT | grep $TAG

#!/bin/bash
# SQ / TCC counter passes over the Dense A/B child (xgemm.h kernels): tools/pmc_x6.sh <shape> ; then tools/pmc_dump.py xgemm gpurun_out/pmcx/*.csv
set -u
SHAPE=${1:-r2d2}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmcx
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
export SEEDHIP_X6=7
B="python $R/tools/bench_x6.py --child $SHAPE"
SETS=("SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LEVEL_WAVES"
      "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"
      "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_VMEM_TA_ADDR_FIFO_FULL SQ_INSTS_BRANCH"
      "FETCH_SIZE TCC_HIT_sum"
      "WRITE_SIZE TCC_MISS_sum")
for i in 0 1 2 3 4; do
  rocprofv3 --pmc ${SETS[$i]} --kernel-trace -d $OUT -o q$i --output-format csv -- $B > $OUT/q$i.log 2>&1
done
python $R/tools/pmc_dump.py xgemm $OUT/q*_counter_collection.csv

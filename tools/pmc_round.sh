#!/bin/bash
# SQ counter passes over the cfg2 bench (separate --pmc runs, kernel-trace only); condense with tools/pmc_sum.py or
# tools/pmc_valu.py.  Usage: tools/pmc_round.sh [pass ...]   (default: all three)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --steps 3 --warmup 2 --quick --graph 0"
SETS=("SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_LEVEL_WAVES"
      "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_INSTS_BRANCH SQ_WAIT_INST_LDS"
      "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_VMEM_TA_ADDR_FIFO_FULL SQ_LDS_BANK_CONFLICT")
PASSES=${@:-1 2 3}
for i in $PASSES; do
  rocprofv3 --pmc ${SETS[$((i-1))]} --kernel-trace -d $OUT -o p$i --output-format csv -- $B > $OUT/p$i.log 2>&1
done
ls $OUT | head

#!/bin/bash
# SQ counter passes over the first-conv micro-benchmark only (separate --pmc runs, kernel-trace only)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_stack
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp; export PYTHONPATH=$R
B="python $R/tools/bench_kernels.py stack"
SETS=("SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LEVEL_WAVES GRBM_GUI_ACTIVE"
      "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_INSTS_SMEM SQ_WAIT_INST_LDS"
      "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT")
for i in 1 2 3; do
  rocprofv3 --pmc ${SETS[$((i-1))]} --kernel-trace -d $OUT -o p$i --output-format csv -- $B > $OUT/p$i.log 2>&1
done
python - <<P
import csv, collections
for i in (1,2,3):
  d = collections.defaultdict(lambda: collections.defaultdict(list))
  try:
    rows = list(csv.DictReader(open('$OUT/p%d_counter_collection.csv' % i)))
  except Exception as e:
    print('pass', i, 'missing', e); continue
  for r in rows:
    if 'stackconv' in r['Kernel_Name']:
      d[r['Kernel_Name'][:70]][r['Counter_Name']].append(float(r['Counter_Value']))
  for k, c in d.items():
    print(k)
    for n, v in c.items():
      print('   %-28s %14.0f  (n=%d)' % (n, sum(v) / len(v), len(v)))
P

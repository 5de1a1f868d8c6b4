set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --steps 2 --warmup 2 --quick --graph 0 --config r2d2"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT -o cfg5_mfma --kernel-trace --output-format csv -- $B > $OUT/cfg5_mfma.log 2>&1
rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES -d $OUT -o cfg5_mix --kernel-trace --output-format csv -- $B > $OUT/cfg5_mix.log 2>&1

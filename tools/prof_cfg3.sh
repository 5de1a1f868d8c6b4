R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/prof; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT -o r02d_cfg3 --output-format csv -- python $R/bench.py --steps 5 --warmup 3 --quick --graph 0 --config dmlab > $OUT/r02d_cfg3.log 2>&1
ls $OUT | grep r02d

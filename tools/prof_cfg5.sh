R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/prof; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT -o r02f_cfg5 --output-format csv -- python $R/bench.py --steps 5 --warmup 3 --quick --graph 0 --config r2d2 > $OUT/r02f_cfg5.log 2>&1
python $R/bench.py --quick --config r2d2 2>/dev/null | tail -1 > $OUT/r02f_cfg5_bench.json

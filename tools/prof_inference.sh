R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/prof; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT -o r02e_inf256 --output-format csv -- python $R/tools/bench_inference.py --n 256 --envs 512 --mode graph --calls 200 > $OUT/r02e_inf256.log 2>&1
tail -3 $OUT/r02e_inf256.log

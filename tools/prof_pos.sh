R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/prof; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for v in 0 1; do
SEEDHIP_DGRAD_POS=$v rocprofv3 --kernel-trace --stats -d $OUT -o pos${v}_cfg5 --output-format csv -- python $R/bench.py --steps 5 --warmup 3 --quick --graph 0 --config r2d2 > $OUT/pos${v}.log 2>&1
done
ls $OUT | grep pos

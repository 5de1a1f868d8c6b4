# rocprofv3 kernel trace of inference and training running TOGETHER (tools/bench_serving.py, in-process feeder).
#   bash tools/prof_serving.sh [tag]     then: python tools/serving_overlap.py gpurun_out/prof/<tag>_kernel_trace.csv
TAG=${1:-r03_serving}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/prof; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT -o $TAG --output-format csv -- python $R/tools/bench_serving.py --mode inprocess --seconds 1.5 > $OUT/$TAG.log 2>&1
tail -1 $OUT/$TAG.log | cut -c 1-400

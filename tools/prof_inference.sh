# rocprofv3 kernel trace of the captured central-inference step (tools/bench_inference.py --mode graph) at one batch size.
#   bash tools/prof_inference.sh [n] [envs] [tag]
N=${1:-1024}; E=${2:-4096}; TAG=${3:-r03_inf$N}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/prof; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT -o $TAG --output-format csv -- python $R/tools/bench_inference.py --n $N --envs $E --mode graph --calls 200 > $OUT/$TAG.log 2>&1
tail -3 $OUT/$TAG.log

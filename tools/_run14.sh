cd /root/repo
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "stack_conv" > gpurun_out/t14.log 2>&1
echo "pytest rc $?" >> gpurun_out/t14.log
(python tools/bench_kernels.py stack; SEEDHIP_STACK_TR=0 python tools/bench_kernels.py stack) > gpurun_out/bk14.log 2>&1
for v in 1 0 1 0; do echo "--- atari TR=$v"; SEEDHIP_STACK_TR=$v python bench.py --quick 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value']); k=d['kernels_ms_per_step']; print({n:v for n,v in k.items() if 'stack' in n})"; done > gpurun_out/b14.log 2>&1
tail -3 gpurun_out/t14.log | cut -c1-300; grep stack gpurun_out/bk14.log; cat gpurun_out/b14.log

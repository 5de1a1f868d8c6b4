#!/bin/bash
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "cgx" 2>&1 | tail -8
timeout 200 python tools/bench_kernels.py r2d2conv 2>&1 | grep -v amdgpu
SEEDHIP_CONV_BF16X6=4 timeout 200 python tools/bench_kernels.py r2d2conv 2>&1 | grep -v amdgpu

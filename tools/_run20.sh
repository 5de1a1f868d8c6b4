#!/bin/bash
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_deep.py -x -q -m gpu -k "padded_kernel or train_step_parity" 2>&1 | tail -5
timeout 600 python bench.py --config r2d2 --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-200
SEEDHIP_LSTM_PADK=0 timeout 600 python bench.py --config r2d2 --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-200
timeout 300 python bench.py --config dmlab --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-200

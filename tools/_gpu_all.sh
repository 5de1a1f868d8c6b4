#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
timeout 2400 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -15 > gpurun_out/gpu_all.log
cat gpurun_out/gpu_all.log

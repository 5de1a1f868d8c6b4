cd /root/repo
export SEEDHIP_SKIP_FP64=0
timeout 2400 python -m pytest tests -q -m gpu --deselect tests/test_gpu_fullsize.py > gpurun_out/final_tests.log 2>&1
echo "pytest rc $?" >> gpurun_out/final_tests.log
timeout 1200 python -m pytest tests/test_gpu_fullsize.py -q -s > gpurun_out/final_fullsize.log 2>&1
echo "pytest rc $?" >> gpurun_out/final_fullsize.log
bash tools/profile_round.sh r05 > gpurun_out/profile_round.log 2>&1
bash tools/pmc_mfma.sh r05 > gpurun_out/pmc_mfma.log 2>&1
bash tools/pmc_cfg3.sh > gpurun_out/pmc_cfg3.log 2>&1
tail -3 gpurun_out/final_tests.log | cut -c1-300; tail -3 gpurun_out/final_fullsize.log | cut -c1-300; ls gpurun_out/prof | grep r05 | head -30

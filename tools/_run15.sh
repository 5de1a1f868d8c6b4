cd /root/repo
bash tools/pmc_kernel.sh w15 stack > /dev/null 2>&1
python tools/pmc_kernel.py gpurun_out/pmc w15 stackconv > gpurun_out/pmc_w15.log 2>&1
SEEDHIP_STACK_TR=0 bash tools/pmc_kernel.sh w15b stack > /dev/null 2>&1
python tools/pmc_kernel.py gpurun_out/pmc w15b stackconv >> gpurun_out/pmc_w15.log 2>&1
cat gpurun_out/pmc_w15.log

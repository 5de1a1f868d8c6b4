#!/bin/bash
# Same-box A/B of two builds of libseedhip.so on the cfg2 step: tools/ab_quick.sh <other-lib.so> [rounds]
# (bench.py --quick, alternating; prints ms_per_step of the six windows and the attributed kernel times)
other=$1; rounds=${2:-2}
for r in $(seq 1 $rounds); do
  for lib in "" "$other"; do
    SEEDHIP_LIB=$lib python bench.py --quick 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
k=d['kernels_ms_per_step']
print('${lib:-HEAD-tree}'[-40:], 'median %.4f' % d['windows']['median'], 'min %.4f' % d['windows']['min'], ' fwd %.4f wgrad %.4f' % (k['stack_conv_fwd'], k['stack_conv_wgrad']))
"
  done
done

import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['kernels_ms_per_step']
print(d['ms_per_step'], 'wgrad', k.get('conv_wgrad[4x4/2 16->32 @20x20]'), 'dgrad', k.get('conv_dgrad[4x4/2 16->32 @20x20]'), 'fwd', k.get('conv_fwd[4x4/2 16->32 @20x20]'))

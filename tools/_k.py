import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['kernels_ms_per_step']
print(d['ms_per_step'], 'dgrad', k.get('conv_dgrad[4x4/2 16->32 @20x20]'), 'fwd', k.get('conv_fwd[4x4/2 16->32 @20x20]'), 'roofline', d['roofline']['kernel'], d['roofline']['avg_kernel_ms'], d['roofline']['frac'])

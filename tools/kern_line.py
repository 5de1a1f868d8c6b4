import json,sys
d=json.loads(sys.stdin.readlines()[-1])
k=d['kernels_ms_per_step']
print(d['ms_per_step'], {n.replace('conv_','').split('[')[0]+('1' if '4x4' in n else 'fc' if '2592' in n else ''):v for n,v in k.items() if '4x4' in n or 'stack_conv' in n or '2592' in n})

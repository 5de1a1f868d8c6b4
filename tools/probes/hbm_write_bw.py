"""HBM write / copy / read rates of plain torch kernels at the second conv's output size and beyond (MI355X: fill 88 MB in
14 us = 6.1 TB/s, copy 7.0 TB/s read + write, sum 3.8 TB/s): the 33 us that wfx.h's output path takes for 87.5 MB are not
a write-bandwidth limit.  python tools/probes/hbm_write_bw.py"""
import torch
def t(fn,reps=20):
    for _ in range(3): fn()
    s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e)/reps
for mb in (87.5, 350, 1400):
    n=int(mb*1e6/4); a=torch.empty(n,device='cuda'); b=torch.empty(n,device='cuda')
    ms=t(lambda: a.fill_(1.0)); print('fill %.0f MB: %.1f us  %.2f TB/s'%(mb,ms*1e3,mb/1e6/ms*1e3))
    ms=t(lambda: b.copy_(a)); print('copy %.0f MB: %.1f us  %.2f TB/s (r+w)'%(mb,ms*1e3,2*mb/1e6/ms*1e3))
    ms=t(lambda: a.sum()); print('sum  %.0f MB: %.1f us  %.2f TB/s'%(mb,ms*1e3,mb/1e6/ms*1e3))

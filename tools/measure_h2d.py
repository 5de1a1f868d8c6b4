import torch, time
n = 21 * 512 * 7056
h = torch.empty(n, dtype=torch.uint8).pin_memory()
d = torch.empty(n, dtype=torch.uint8, device='cuda')
for _ in range(3): d.copy_(h, non_blocking=True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): d.copy_(h, non_blocking=True)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print('H2D %.1f MB pinned: %.3f ms  (%.1f GB/s)' % (n / 1e6, ms, n / ms / 1e6))

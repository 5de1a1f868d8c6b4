import torch, time
dev = torch.device('cuda')
for mb in (7.2, 14.4, 28.9):
  n = int(mb * 1e6)
  h = torch.zeros(n, dtype=torch.uint8).pin_memory()
  d = torch.zeros(n, dtype=torch.uint8, device=dev)
  s = torch.cuda.Stream()
  with torch.cuda.stream(s):
    for _ in range(5): d.copy_(h, non_blocking=True)
    s.synchronize()
    t0 = time.perf_counter()
    for _ in range(50): d.copy_(h, non_blocking=True)
    t1 = time.perf_counter()
    s.synchronize()
    t2 = time.perf_counter()
  print('%.1f MB: %.1f GB/s, host submit %.1f us per copy, total %.1f us per copy' % (mb, 50 * n / (t2 - t0) / 1e9, (t1 - t0) / 50 * 1e6, (t2 - t0) / 50 * 1e6))

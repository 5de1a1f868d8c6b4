#!/usr/bin/env python
"""Reads a rocprofv3 kernel trace of tools/bench_serving.py and says how busy the device was: the union of all kernel
intervals, per-queue busy time, and the largest kernels by total time, over the steady-state window.

  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof/serve -o serve -- python tools/bench_serving.py ...
  python tools/serving_timeline.py gpurun_out/prof/serve/*_kernel_trace.csv
"""
import collections
import csv
import sys


def main():
  rows = list(csv.DictReader(open(sys.argv[1])))
  ev = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Queue_Id'], r['Kernel_Name'][:70]) for r in rows]
  ev.sort()
  t_lo, t_hi = ev[0][0], ev[-1][1]
  w0, w1 = t_lo + (t_hi - t_lo) * 0.6, t_lo + (t_hi - t_lo) * 0.95          # steady state: the last part of the run
  ev = [e for e in ev if e[0] >= w0 and e[1] <= w1]
  span = w1 - w0
  busy, cur_s, cur_e = 0, None, None
  for s, e, _, _ in ev:
    if cur_e is None or s > cur_e:
      if cur_e is not None:
        busy += cur_e - cur_s
      cur_s, cur_e = s, e
    else:
      cur_e = max(cur_e, e)
  busy += (cur_e - cur_s) if cur_e else 0
  print('window %.1f ms, device busy (union of kernels) %.1f%%, sum of kernel durations / window %.2f'
        % (span / 1e6, 100.0 * busy / span, sum(e - s for s, e, _, _ in ev) / span))
  perq = collections.defaultdict(int)
  for s, e, q, _ in ev:
    perq[q] += e - s
  for q, t in sorted(perq.items(), key=lambda kv: -kv[1]):
    print('  queue %s: busy %.1f%%' % (q, 100.0 * t / span))
  perk = collections.defaultdict(lambda: [0, 0])
  for s, e, _, k in ev:
    perk[k][0] += e - s; perk[k][1] += 1
  for k, (t, c) in sorted(perk.items(), key=lambda kv: -kv[1][0])[:16]:
    print('  %-70s %6d calls %8.1f us avg %5.1f%% of window' % (k, c, t / c / 1e3, 100.0 * t / span))


if __name__ == '__main__':
  main()

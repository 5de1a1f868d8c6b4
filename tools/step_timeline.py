#!/usr/bin/env python
"""One replay of the train step's HIP graph as a timeline: kernel, start offset, duration, idle gap in front of it.

  rocprofv3 --kernel-trace --output-format csv -d <dir> -o t -- python bench.py --quick
  python tools/step_timeline.py <dir>/.../t_kernel_trace.csv [first-kernel-substring]

Takes the LAST complete run of kernels between two dispatches whose name contains the first-kernel substring (default:
stack_prepare, the first kernel of the cfg2 step); says what the kernels add up to and what the device idled."""
import csv
import sys


def main():
  path = sys.argv[1]
  first = sys.argv[2] if len(sys.argv) > 2 else 'stack_prepare'
  rows = []
  with open(path) as f:
    for r in csv.DictReader(f):
      rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
  rows.sort()
  starts = [i for i, r in enumerate(rows) if first in r[2]]
  if len(starts) < 3:
    print('fewer than three steps in the trace'); return 1
  a, b = starts[-3], starts[-2]                    # a replay in the middle of the timed windows
  step = rows[a:b]
  t0 = step[0][0]
  total = rows[b][0] - t0
  busy = 0
  prev_end = t0
  print('%-72s %9s %9s %8s' % ('kernel', 'start us', 'dur us', 'gap us'))
  for s, e, n in step:
    print('%-72s %9.1f %9.1f %8.1f' % (n[:72], (s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3))
    busy += e - s
    prev_end = max(prev_end, e)
  print('step %.1f us, kernels %.1f us, idle %.1f us (%d launches)' % (total / 1e3, busy / 1e3, (total - busy) / 1e3, len(step)))
  return 0


if __name__ == '__main__':
  sys.exit(main())

#!/usr/bin/env python
"""Per-kernel averages of every counter in rocprofv3 --pmc CSVs.  python tools/pmc_dump.py <substr> <csv> [<csv> ...]"""
import collections
import csv
import sys


def main():
  sub = sys.argv[1]
  d = collections.defaultdict(lambda: collections.defaultdict(list))
  for path in sys.argv[2:]:
    for r in csv.DictReader(open(path)):
      k = r['Kernel_Name']
      if sub not in k:
        continue
      k = k[:70]
      d[k][r['Counter_Name']].append(float(r['Counter_Value']))
      d[k]['__dur_us__' + path[-28:-23]].append((float(r['End_Timestamp']) - float(r['Start_Timestamp'])) / 1e3)
  for k, c in d.items():
    print(k)
    for n in sorted(c):
      v = c[n]
      print('   %-34s %16.1f  (n=%d)' % (n, sum(v) / len(v), len(v)))


if __name__ == '__main__':
  main()

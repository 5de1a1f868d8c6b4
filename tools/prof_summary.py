#!/usr/bin/env python
"""Condenses rocprofv3 output (gpurun_out/prof/*) into the small summaries committed under profiles/.

  python tools/prof_summary.py gpurun_out/prof r1 profiles/r01_cfg2

writes <out>_kernel_stats.csv (the --stats table) and <out>_pmc.csv (per-kernel FETCH_SIZE /
WRITE_SIZE averages in KB, from the separate --pmc passes <tag>_fetch / <tag>_write).
"""
import collections
import csv
import os
import sys


def main():
  src, tag, out = sys.argv[1], sys.argv[2], sys.argv[3]
  rows = list(csv.DictReader(open(os.path.join(src, tag + '_kernel_stats.csv'))))
  with open(out + '_kernel_stats.csv', 'w') as f:
    w = csv.writer(f)
    w.writerow(['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage', 'MinNs', 'MaxNs'])
    for r in rows:
      w.writerow([r['Name'][:160], r['Calls'], r['TotalDurationNs'], r['AverageNs'], r['Percentage'], r['MinNs'],
                  r['MaxNs']])
  dig = os.path.join(src, tag + '_csrc.sha256')               # kernel-source digest written on the GPU box
  if os.path.exists(dig):
    with open(out + '_csrc.sha256', 'w') as f:
      f.write(open(dig).read())
  pmc = collections.OrderedDict()
  for suffix, counter in (('_fetch', 'FETCH_SIZE'), ('_write', 'WRITE_SIZE')):
    p = os.path.join(src, tag + suffix + '_counter_collection.csv')
    if not os.path.exists(p):
      continue
    for r in csv.DictReader(open(p)):
      if r['Counter_Name'] != counter:
        continue
      key = (r['Kernel_Name'][:160], r['Grid_Size'])
      pmc.setdefault(key, collections.defaultdict(list))[counter].append(float(r['Counter_Value']))
  if pmc:
    with open(out + '_pmc.csv', 'w') as f:
      w = csv.writer(f)
      w.writerow(['Kernel', 'Grid_Size', 'dispatches',
                  'FETCH_SIZE_KB_avg(raw; x2 for wide coalesced reads on gfx950)', 'WRITE_SIZE_KB_avg'])
      for (k, g), d in pmc.items():
        fs, ws = d.get('FETCH_SIZE', []), d.get('WRITE_SIZE', [])
        w.writerow([k, g, max(len(fs), len(ws)), '%.1f' % (sum(fs) / len(fs)) if fs else '',
                    '%.1f' % (sum(ws) / len(ws)) if ws else ''])


if __name__ == '__main__':
  main()

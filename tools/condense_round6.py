#!/usr/bin/env python
"""Condenses what tools/profile_round6.sh left under gpurun_out/ into the committed profiles/<tag>_* files:

  python tools/condense_round6.py [r06]

kernel stats + PMC tables of cfg2 / cfg3 (tools/prof_summary.py), the MFMA counters (tools/pmc_mfma.py), the traffic
records bench.py reads (tools/make_traffic.py; stamped with the kernel-source digest written on the GPU box), the three
bench lines, kernel stats of cfg5 and of one 1 024-row inference call, the serving timeline."""
import csv
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(*a):
  subprocess.run([sys.executable] + list(a), cwd=ROOT, check=True, stdout=subprocess.DEVNULL)


def condense(src, dst):
  rows = list(csv.DictReader(open(src)))
  with open(dst, 'w') as f:
    w = csv.writer(f)
    w.writerow(['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage', 'MinNs', 'MaxNs'])
    for r in rows:
      w.writerow([r['Name'][:160], r['Calls'], r['TotalDurationNs'], r['AverageNs'], r['Percentage'], r['MinNs'], r['MaxNs']])


def main():
  tag = sys.argv[1] if len(sys.argv) > 1 else 'r06'
  prof, pmc, out = os.path.join(ROOT, 'gpurun_out', 'prof'), os.path.join(ROOT, 'gpurun_out', 'pmc'), os.path.join(ROOT, 'profiles')
  for cfg in ('cfg2', 'cfg3'):
    run('tools/prof_summary.py', prof, '%s_%s' % (tag, cfg), os.path.join(out, '%s_%s' % (tag, cfg)))
  run('tools/pmc_mfma.py', pmc, tag, os.path.join(out, '%s_cfg2_mfma.csv' % tag))
  run('tools/make_traffic.py', os.path.join(out, '%s_cfg2_pmc.csv' % tag), os.path.join(out, '%s_cfg2_traffic.json' % tag))
  run('tools/make_traffic.py', os.path.join(out, '%s_cfg3_pmc.csv' % tag), os.path.join(out, '%s_cfg3_traffic.json' % tag), 'cfg3')
  for cfg in ('cfg2', 'cfg3', 'cfg5'):
    shutil.copy(os.path.join(prof, '%s_%s_bench.json' % (tag, cfg)), os.path.join(out, '%s_%s_bench.json' % (tag, cfg)))
  shutil.copy(os.path.join(prof, '%s_serving_timeline.txt' % tag), os.path.join(out, '%s_serving_timeline.txt' % tag))
  condense(os.path.join(prof, '%s_cfg5_kernel_stats.csv' % tag), os.path.join(out, '%s_cfg5_kernel_stats.csv' % tag))
  condense(os.path.join(prof, '%s_inf1024_kernel_stats.csv' % tag), os.path.join(out, '%s_inf1024_kernel_stats.csv' % tag))
  print(open(os.path.join(out, '%s_cfg2_csrc.sha256' % tag)).read().strip(), '= kernel sources of the profiled build')


if __name__ == '__main__':
  main()

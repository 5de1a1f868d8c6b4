#!/usr/bin/env python
"""Condenses tools/pmc_kernel.sh output: per kernel, matrix-pipe utilisation, instruction mix per MFMA, LDS bank-conflict
share and where the waves' cycles go.
  python tools/pmc_kernel.py gpurun_out/pmc <tag> [kernel-name substring] [out.csv]"""
import collections
import csv
import os
import sys


def load(path):
  d = collections.defaultdict(lambda: collections.defaultdict(list))
  if os.path.exists(path):
    for r in csv.DictReader(open(path)):
      d[r['Kernel_Name'][:140]][r['Counter_Name']].append(float(r['Counter_Value']))
      d[r['Kernel_Name'][:140]]['__dur_ns__'].append(float(r['End_Timestamp']) - float(r['Start_Timestamp']))
  return d


def main():
  src, tag = sys.argv[1], sys.argv[2]
  sub = sys.argv[3] if len(sys.argv) > 3 else ''
  out = sys.argv[4] if len(sys.argv) > 4 else None
  a = load(os.path.join(src, tag + '_lds_counter_collection.csv'))
  b = load(os.path.join(src, tag + '_mix_counter_collection.csv'))
  avg = lambda c, n: (sum(c[n]) / len(c[n]) if c.get(n) else 0.0)
  rows = []
  for k, m in b.items():
    if sub not in k or avg(m, 'SQ_INSTS_MFMA') <= 0:
      continue
    l = a.get(k, {})
    gui, busy, mf = avg(m, 'GRBM_GUI_ACTIVE'), avg(m, 'SQ_VALU_MFMA_BUSY_CYCLES'), avg(m, 'SQ_INSTS_MFMA')
    wc = avg(l, 'SQ_WAVE_CYCLES') or 1.0
    rows.append(dict(kernel=k, dur_us=avg(m, '__dur_ns__') / 1e3, clock_GHz=gui / 8.0 / avg(m, '__dur_ns__') if gui else 0,
                     mfma_util=busy * 8.0 / (1024.0 * gui) if gui else 0, valu_per_mfma=avg(m, 'SQ_INSTS_VALU') / mf,
                     salu_per_mfma=avg(m, 'SQ_INSTS_SALU') / mf, lds_per_mfma=avg(l, 'SQ_INSTS_LDS') / mf,
                     lds_conflict_frac=avg(l, 'SQ_LDS_BANK_CONFLICT') / (avg(l, 'SQ_LDS_IDX_ACTIVE') or 1.0),
                     lds_active_per_gui=avg(l, 'SQ_LDS_IDX_ACTIVE') * 8.0 / (256.0 * gui) if gui else 0,
                     wait_any=avg(l, 'SQ_WAIT_ANY') / wc, wait_inst=avg(l, 'SQ_WAIT_INST_ANY') / wc,
                     wait_inst_lds=avg(l, 'SQ_WAIT_INST_LDS') / wc, issuing=avg(l, 'SQ_ACTIVE_INST_ANY') / wc))
  rows.sort(key=lambda r: -r['dur_us'])
  for r in rows:
    print('%-100s %7.1f us @%.2f GHz  MFMA %.3f  VALU/MFMA %.2f SALU %.2f LDS %.2f  LDS conflict %.2f active %.2f  '
          'parked %.2f issue-stall %.2f (lds %.2f) issuing %.2f' %
          (r['kernel'][:100], r['dur_us'], r['clock_GHz'], r['mfma_util'], r['valu_per_mfma'], r['salu_per_mfma'], r['lds_per_mfma'],
           r['lds_conflict_frac'], r['lds_active_per_gui'], r['wait_any'], r['wait_inst'], r['wait_inst_lds'], r['issuing']))
  if out and rows:
    with open(out, 'w') as f:
      w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
      w.writeheader()
      for r in rows:
        w.writerow({k: (('%.4f' % v) if isinstance(v, float) else v) for k, v in r.items()})


if __name__ == '__main__':
  main()

#!/usr/bin/env python
"""Condenses tools/pmc_mfma.sh output into profiles/<tag>_cfg2_mfma.csv: per kernel, the share of SIMD cycles the matrix
pipe was busy, and the instruction mix per MFMA.
  MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)
(SQ_VALU_MFMA_BUSY_CYCLES is summed over the chip: 32 cycles per v_mfma_f32_16x16x4_f32 -- checked against SQ_INSTS_MFMA;
GRBM_GUI_ACTIVE is summed over the 8 XCDs: GRBM_GUI_ACTIVE / 8 / kernel duration = the clock the kernel actually ran at,
2.30-2.35 GHz for these kernels against the 2.4 GHz the 157.3 TFLOP/s peak assumes).
  python tools/pmc_mfma.py gpurun_out/pmc r02a profiles/r02a_cfg2_mfma.csv"""
import collections
import csv
import os
import sys


def load(path):
  d = collections.defaultdict(lambda: collections.defaultdict(list))
  if os.path.exists(path):
    for r in csv.DictReader(open(path)):
      d[r['Kernel_Name'][:120]][r['Counter_Name']].append(float(r['Counter_Value']))
      d[r['Kernel_Name'][:120]]['__dur_ns__'].append(float(r['End_Timestamp']) - float(r['Start_Timestamp']))
  return d


def main():
  src, tag, out = sys.argv[1], sys.argv[2], sys.argv[3]
  a = load(os.path.join(src, tag + '_mfma_counter_collection.csv'))
  b = load(os.path.join(src, tag + '_mix_counter_collection.csv'))
  avg = lambda c, n: (sum(c[n]) / len(c[n]) if c.get(n) else 0.0)
  rows = []
  for k, c in a.items():
    busy, gui = avg(c, 'SQ_VALU_MFMA_BUSY_CYCLES'), avg(c, 'GRBM_GUI_ACTIVE')
    if busy < 1e6 or gui <= 0:
      continue
    m = b.get(k, {})
    mf = avg(m, 'SQ_INSTS_MFMA')
    rows.append((busy, k, gui, busy * 8.0 / (1024.0 * gui), gui / 8.0 / avg(c, '__dur_ns__'), avg(c, '__dur_ns__') / 1e3, mf,
                 avg(m, 'SQ_INSTS_VALU') / mf if mf else 0.0,
                 avg(m, 'SQ_INSTS_SALU') / mf if mf else 0.0, avg(m, 'SQ_INSTS_LDS') / mf if mf else 0.0,
                 avg(m, 'SQ_WAIT_INST_ANY') / avg(m, 'SQ_WAVE_CYCLES') if avg(m, 'SQ_WAVE_CYCLES') else 0.0))
  with open(out, 'w') as f:
    w = csv.writer(f)
    w.writerow(['Kernel', 'avg_duration_us', 'effective_clock_GHz', 'GRBM_GUI_ACTIVE_cycles(8 XCDs)', 'SQ_VALU_MFMA_BUSY_CYCLES',
                'mfma_utilisation', 'MFMA_insts', 'VALU_per_MFMA', 'SALU_per_MFMA', 'LDS_per_MFMA', 'wave_wait_frac'])
    for busy, k, gui, frac, clk, dur, mf, va, sa, lds, wait in sorted(rows, reverse=True):
      w.writerow([k, '%.1f' % dur, '%.3f' % clk, '%.0f' % gui, '%.0f' % busy, '%.3f' % frac, '%.0f' % mf, '%.2f' % va,
                  '%.2f' % sa, '%.2f' % lds, '%.2f' % wait])
      print('%-92s %6.1f us @%.2f GHz  MFMA util %.3f  VALU/MFMA %.2f SALU %.2f LDS %.2f wait %.2f'
            % (k[:92], dur, clk, frac, va, sa, lds, wait))


if __name__ == '__main__':
  main()

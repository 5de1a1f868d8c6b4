"""Per-kernel instruction mix from tools/pmc_round.sh pass 2 (+ pass 1 when present): VALU / SALU / LDS / VMEM per MFMA.
On gfx950 a SIMD's VALU instructions do not overlap its MFMAs, so VALU-per-MFMA is the number to drive down.
  python tools/pmc_valu.py gpurun_out/pmc"""
import csv, collections, os, sys
d = collections.defaultdict(lambda: collections.defaultdict(list))
for i in (1, 2, 3):
  f = os.path.join(sys.argv[1], 'p%d_counter_collection.csv' % i)
  if not os.path.exists(f): continue
  for r in csv.DictReader(open(f)):
    d[r['Kernel_Name'][:78]][r['Counter_Name']].append(float(r['Counter_Value']))
rows = []
for k, c in d.items():
  g = lambda n: (sum(c[n]) / len(c[n]) if c.get(n) else 0.0)
  mf = g('SQ_INSTS_MFMA')
  if mf < 1e5: continue
  rows.append((g('SQ_INSTS_VALU'), k, mf, g('SQ_INSTS_SALU'), g('SQ_INSTS_LDS'), g('SQ_INSTS_VMEM_RD') + g('SQ_INSTS_VMEM_WR'),
               g('SQ_INSTS_BRANCH'), g('SQ_VALU_MFMA_BUSY_CYCLES')))
for va, k, mf, sa, lds, vm, br, busy in sorted(rows, reverse=True):
  print('%-78s\n    MFMA %6.2fM  VALU %6.1fM (%.2f per MFMA)  SALU/MFMA %.2f  LDS/MFMA %.2f  VMEM/MFMA %.3f  BR/MFMA %.2f  MFMA busy/SIMD %.0fK cycles'
        % (k, mf / 1e6, va / 1e6, va / mf, sa / mf, lds / mf, vm / mf, br / mf, busy / 1024 / 1e3))

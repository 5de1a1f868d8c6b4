#!/usr/bin/env python
"""Static check of the compiled kernels for the wide-store write-after-read hazard (r4): a VMEM store of more than 8
bytes reads its data registers over several cycles; a VALU instruction issued in the very next slot that WRITES one of
those registers can corrupt the stored value (seen on gfx950 with `buffer_store_dwordx4` from
__builtin_amdgcn_raw_buffer_store_b128: no wait state was inserted, one test failed with the next tile's values in the
upper half of a stored quad).  Prints every `*_store_dwordx3/x4 v[a:b]` whose NEXT instruction writes into v[a:b].

  python tools/isa_store_hazard.py build/obj/*.o        (exit code 1 if anything is found)"""
import re
import subprocess
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import isa_waits

STORE = re.compile(r'^\s*(buffer_store_dwordx[34]|global_store_dwordx[34]|flat_store_dwordx[34]|scratch_store_dwordx[34])\s+(.*)$')
VREG = re.compile(r'v\[(\d+):(\d+)\]|v(\d+)')


def regs(tok):
  m = VREG.match(tok.strip())
  if not m:
    return None
  if m.group(1) is not None:
    return int(m.group(1)), int(m.group(2))
  return int(m.group(3)), int(m.group(3))


def data_regs(mn, ops):
  toks = [t.strip() for t in ops.split(',')]
  if mn.startswith('buffer_store'):
    return regs(toks[0])                       # vdata first
  if mn.startswith('global_store') or mn.startswith('flat_store'):
    return regs(toks[1]) if len(toks) > 1 else None      # vaddr, vdata
  if mn.startswith('scratch_store'):
    return regs(toks[1]) if len(toks) > 1 else None
  return None


def main():
  found = 0
  for obj in sys.argv[1:]:
    for co in isa_waits.device_code(obj):
      out = subprocess.run([os.path.join(isa_waits.LLVM, 'llvm-objdump'), '-d', co], capture_output=True, text=True).stdout
      kernel = None
      lines = out.splitlines()
      for i, ln in enumerate(lines):
        m = re.match(r'^[0-9a-f]+ <(.*)>:$', ln)
        if m:
          kernel = m.group(1)
          continue
        body = ln.split('//')[0]
        sm = STORE.match(body)
        if not sm:
          continue
        d = data_regs(sm.group(1), sm.group(2))
        if d is None:
          continue
        j = i + 1
        while j < len(lines) and not lines[j].split('//')[0].strip():
          j += 1
        if j >= len(lines):
          continue
        nxt = lines[j].split('//')[0].strip()
        mn = nxt.split()[0] if nxt else ''
        if not (mn.startswith('v_') or mn.startswith('ds_read') or mn.endswith('_load_dwordx4') or '_load_' in mn):
          continue
        if mn.startswith('v_cmp') and not mn.startswith('v_cmpx'):
          continue
        if mn.startswith('v_mfma'):
          continue                             # an MFMA writes its destination passes later (>= 16 cycles): not at issue
        ops = nxt[len(mn):].split(',')
        w = regs(ops[0]) if ops and ops[0].strip() else None
        if w and not (w[1] < d[0] or w[0] > d[1]):
          found += 1
          print('%s: %s\n    %s\n    %s' % (os.path.basename(obj), kernel, body.strip(), nxt))
  print('%d wide store(s) followed at once by a write of their data registers' % found)
  return 1 if found else 0


if __name__ == '__main__':
  sys.exit(main())

#!/usr/bin/env python
"""Where does a kernel WAIT?  Prints, for every kernel of a compiled object whose name matches, the order of its
vector-memory requests, s_waitcnt vmcnt(N), barriers and MFMA bursts, plus registers / spills -- the view that found
this round's stalls ("what the waits say"):

  * vector memory operations retire in order and s_waitcnt vmcnt(N) means "all but the N youngest": a load needed NOW
    behind a prefetch needed LATER waits for the prefetch (look for a small load followed by vmcnt(0) at a loop head);
  * loads and stores share the counter and do not retire in order with each other: with a store in flight every operand
    wait is vmcnt(0) (look for load / wait / store / load / wait ... chains in an epilogue);
  * the compiler sinks loads to their first use (a "prefetch" that shows up BEHIND the MFMA burst it was written in
    front of) and waits for a conditional load where it is issued (load immediately followed by vmcnt(0) and v_mov).

  python tools/isa_waits.py build/obj/conv.o ws_tab_kernel
  python tools/isa_waits.py build/obj/stackconv.o fwd_bf16r --from-mfma     # only from the first MFMA on

Needs the ROCm LLVM tools (clang-offload-bundler, llvm-objdump, llvm-readelf); no GPU."""
import argparse
import os
import re
import subprocess
import sys
import tempfile

LLVM = os.environ.get('ROCM_LLVM_BIN', '/opt/rocm/lib/llvm/bin')
KEEP = re.compile(r'^\s*(global_load|global_store|buffer_load|buffer_store|flat_load|flat_store|scratch_|s_load|'
                  r's_waitcnt|s_barrier|v_mfma|s_endpgm|s_cbranch_scc|s_cbranch_vcc)')
BURST = re.compile(r'^(v_mfma|global_load|global_store|buffer_load|buffer_store|scratch_load|scratch_store)')


def device_code(obj):
  """Paths of the gfx code objects inside a hipcc object / shared library (.hip_fatbin: one offload bundle per
  translation unit), or [obj] if the file already is a code object."""
  tmp = tempfile.mkdtemp(prefix='isa_waits_')
  fat = os.path.join(tmp, 'fatbin')
  r = subprocess.run([os.path.join(LLVM, 'llvm-objcopy'), '--dump-section', '.hip_fatbin=' + fat, obj], capture_output=True)
  if r.returncode != 0 or not os.path.exists(fat):
    return [obj]
  blob = open(fat, 'rb').read()
  magic = b'__CLANG_OFFLOAD_BUNDLE__'
  starts = [m.start() for m in re.finditer(re.escape(magic), blob)]
  cos = []
  for n, st in enumerate(starts):
    end = starts[n + 1] if n + 1 < len(starts) else len(blob)
    part = os.path.join(tmp, 'bundle%d' % n)
    open(part, 'wb').write(blob[st:end])
    listing = subprocess.run([os.path.join(LLVM, 'clang-offload-bundler'), '--list', '--type=o', '--input=' + part],
                             capture_output=True, text=True).stdout.split()
    for t in listing:
      if 'gfx' in t:
        co = os.path.join(tmp, 'code%d.co' % n)
        subprocess.run([os.path.join(LLVM, 'clang-offload-bundler'), '--unbundle', '--type=o', '--input=' + part,
                        '--targets=' + t, '--output=' + co], check=True, capture_output=True)
        cos.append(co)
  return cos or [obj]


def kernels(co, pattern):
  notes = subprocess.run([os.path.join(LLVM, 'llvm-readelf'), '--notes', co], capture_output=True, text=True).stdout
  info = {}
  # one YAML list item per kernel ("  - .agpr_count: ..." opens it: keys are sorted, .name comes in the middle)
  for item in re.split(r'\n  - (?=\.)', notes):
    m = re.search(r'\.name:\s+(\S+)', item)
    if not m:
      continue
    cur = info.setdefault(m.group(1), {})
    for key in ('.vgpr_count', '.agpr_count', '.sgpr_count', '.vgpr_spill_count', '.group_segment_fixed_size'):
      k = re.search(r'(?m)^\s*(?:- )?' + re.escape(key) + r':\s+(\d+)', item)
      if k:
        cur[key] = int(k.group(1))
  return {k: v for k, v in info.items() if pattern in k}


def trace(co, symbol, from_mfma):
  asm = subprocess.run([os.path.join(LLVM, 'llvm-objdump'), '-d', '--no-show-raw-insn', '--disassemble-symbols=' + symbol, co],
                       capture_output=True, text=True).stdout.splitlines()
  rows = []
  for n, line in enumerate(asm):
    if KEEP.match(line):
      text = line.split('//')[0].strip()
      op = text.split()[0]
      arg = text[len(op):].strip()
      rows.append((n, op, arg if op.startswith('s_waitcnt') else ''))
  if from_mfma:
    first = next((i for i, r in enumerate(rows) if r[1].startswith('v_mfma')), 0)
    rows = rows[max(first - 12, 0):]
  out, i = [], 0
  while i < len(rows):                                  # collapse bursts of the same request / MFMA
    n, op, arg = rows[i]
    j = i
    if BURST.match(op):
      while j + 1 < len(rows) and rows[j + 1][1] == op:
        j += 1
    out.append('%6d  %s %s%s' % (n, op, arg, '   x%d' % (j - i + 1) if j > i else ''))
    i = j + 1
  return out


def main():
  ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
  ap.add_argument('object', help='hipcc object, shared library or code object')
  ap.add_argument('pattern', help='substring of the (mangled) kernel names to show')
  ap.add_argument('--from-mfma', action='store_true', help='start a few lines before the first MFMA')
  ap.add_argument('--no-branches', action='store_true', help='leave the conditional branches out')
  a = ap.parse_args()
  shown = 0
  for co in device_code(a.object):
    for name, meta in kernels(co, a.pattern).items():
      shown += 1
      demangled = ''
      for filt in (os.path.join(LLVM, 'llvm-cxxfilt'), 'c++filt'):
        try:
          demangled = subprocess.run([filt, name], capture_output=True, text=True).stdout.strip()
          break
        except OSError:
          continue
      print('== %s' % (demangled or name))
      print('   VGPR %s  AGPR %s  SGPR %s  spilled VGPRs %s  LDS %s B' % tuple(
          meta.get(k, '?') for k in ('.vgpr_count', '.agpr_count', '.sgpr_count', '.vgpr_spill_count', '.group_segment_fixed_size')))
      for line in trace(co, name, a.from_mfma):
        if a.no_branches and 's_cbranch' in line:
          continue
        print(line)
  if not shown:
    sys.exit('no kernel matches %r in %s' % (a.pattern, a.object))


if __name__ == '__main__':
  main()

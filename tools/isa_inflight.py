#!/usr/bin/env python
"""Reads of registers a vector-memory load may still be writing.

hipcc places `s_waitcnt` for the loads IT knows; the conv kernels that keep a unit of input items in flight across a whole
compute phase (fgx.h, cgx.h) request them through `asm volatile("buffer_load_dword...")` and wait with a hand-counted
`s_waitcnt vmcnt(N)` -- which only works while the compiler never touches the destination registers in between (a copy
made in front of the wait reads whatever the registers held before: cgx.h's first build).  This tool walks every
matching kernel of an object in address order and reports, for each `buffer_load_dword[x2|x4] ... offen` with a literal 0
scalar offset (the form the asm requests use), the first later instruction that READS one of its destination registers
when no `s_waitcnt vmcnt(N)` with N <= the number of vector memory operations issued behind the request lies between the
two (a wait that lets MORE operations stay outstanding than were issued behind the request does not cover it).

  python tools/isa_inflight.py build/obj/fgx.o fgx_kernel          # prints offenders, exit code 1 if any
"""
import re
import subprocess
import sys

sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.abspath(__file__)))
import isa_waits

REG = re.compile(r'v\[(\d+):(\d+)\]|v(\d+)')
VMEM = ('buffer_load', 'buffer_store', 'buffer_atomic', 'global_load', 'global_store', 'global_atomic', 'flat_load', 'flat_store',
        'scratch_load', 'scratch_store')


def regs(tok):
  out = set()
  for m in REG.finditer(tok):
    if m.group(1) is not None:
      out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    else:
      out.add(int(m.group(3)))
  return out


def kernels_text(co):
  txt = subprocess.run([__import__('os').path.join(isa_waits.LLVM, 'llvm-objdump'), '-d', co], capture_output=True, text=True).stdout
  name, body = None, []
  for line in txt.splitlines():
    m = re.match(r'^[0-9a-f]+ <(.+)>:$', line)
    if m:
      if name:
        yield name, body
      name, body = m.group(1), []
    elif name and line.startswith('\t'):
      ins = line.split('//')[0].strip()
      if ins:
        body.append(ins)
  if name:
    yield name, body


def check(body, asm_only=True):
  """[(index of the load, index of the offending read, text)].  asm_only: only requests with a literal 0 scalar offset
  (fgx.h / cgx.h); otherwise every buffer load -- the compiler's own are followed by its waits and pass trivially.
  The scan follows the ADDRESS order from the request: it stops at an unconditional branch or the end of the program
  (the code behind belongs to another path), drops a destination register from the watch list when anything overwrites
  it, and ends when the list is empty or the first watched register is read."""
  bad = []
  for i, ins in enumerate(body):
    parts = ins.split(None, 1)
    if not parts[0].startswith(('buffer_load_dword', 'buffer_load_ubyte', 'buffer_load_ushort')) or 'offen' not in ins or ' lds' in ins:
      continue
    ops = [o.strip() for o in parts[1].split(',')]
    if len(ops) < 4 or (asm_only and not ops[3].startswith('0 ')):
      continue
    live = set(regs(ops[0]))
    waited = False
    younger = 0                                   # vector memory operations issued behind the request (address order)
    for j in range(i + 1, len(body)):
      t = body[j]
      if t.startswith(('s_endpgm', 's_branch', 's_setpc')):
        break
      if t.startswith('s_waitcnt') and 'vmcnt' in t:
        # vmcnt(N) lets the N youngest operations stay outstanding: it covers this request only if at most `younger`
        # operations were issued behind it on the way here (the COUNT, not only the presence of a wait).  A wait
        # reached through a loop's back edge is not seen in address order: such requests pass on the presence test alone.
        m = re.search(r'vmcnt\((\d+)\)', t)
        if m and int(m.group(1)) <= younger:
          waited = True
      elif t.startswith(VMEM):
        younger += 1
      p2 = t.split(None, 1)
      if len(p2) < 2:
        continue
      o2 = [o.strip() for o in p2[1].split(',')]
      mn = p2[0]
      stores = mn.startswith(('buffer_store', 'global_store', 'ds_write', 'scratch_store', 'flat_store'))
      reads = o2 if stores or mn.startswith(('v_cmp', 's_')) else o2[1:]
      if mn.startswith(('v_mad_u64_u32', 'v_mad_i64_i32')):
        # 32-bit address arithmetic widened to v_mad_u64_u32: the HIGH half of its 64-bit addend is undefined (only the low
        # result is used) and the register allocator names any register for it, in-flight ones included -- count the low half
        reads = o2[2:4] + ['v%d' % min(regs(o2[4]))] if len(o2) > 4 and regs(o2[4]) else o2[2:]
      writes = [] if stores else o2[:1]
      if any(regs(r) & live for r in reads):
        if not waited:
          bad.append((i, j, '%s   <-   %s' % (t, ins)))
        break
      if not mn.startswith('v_mfma'):
        for w in writes:
          live -= regs(w)
      if not live:
        break
  return bad


def kernels_cfg(co):
  """(name, [(text, branch target index or None)]) per kernel: llvm-objdump prints every instruction's address and a
  branch's target as <kernel+0xOFFSET>."""
  txt = subprocess.run([__import__('os').path.join(isa_waits.LLVM, 'llvm-objdump'), '-d', co], capture_output=True, text=True).stdout
  out, name, base, rows = [], None, 0, []
  def flush():
    if name:
      index = {a: i for i, (a, _, _) in enumerate(rows)}
      out.append((name, [(t, index.get(base + off) if off is not None else None) for _, t, off in rows]))
  for line in txt.splitlines():
    m = re.match(r'^([0-9a-f]+) <(.+)>:$', line)
    if m:
      flush()
      base, name, rows = int(m.group(1), 16), m.group(2), []
    elif name and line.startswith('\t'):
      parts = line.split('//')
      ins = parts[0].strip()
      ma = re.match(r'\s*([0-9A-F]+):', parts[1]) if len(parts) > 1 else None
      if not ins or not ma:
        continue
      mt = re.search(r'<[^>]*\+0x([0-9a-f]+)>\s*$', line) if ins.startswith(('s_branch', 's_cbranch')) else None
      rows.append((int(ma.group(1), 16), ins, int(mt.group(1), 16) if mt else None))
  flush()
  return out


def _reads_writes(t):
  p2 = t.split(None, 1)
  if len(p2) < 2:
    return [], []
  o2 = [o.strip() for o in p2[1].split(',')]
  mn = p2[0]
  stores = mn.startswith(('buffer_store', 'global_store', 'ds_write', 'scratch_store', 'flat_store'))
  reads = o2 if stores or mn.startswith(('v_cmp', 's_')) else o2[1:]
  if mn.startswith(('v_mad_u64_u32', 'v_mad_i64_i32')):
    reads = o2[2:4] + ['v%d' % min(regs(o2[4]))] if len(o2) > 4 and regs(o2[4]) else o2[2:]
  writes = [] if stores or mn.startswith('v_mfma') else o2[:1]
  return reads, writes


def check_cfg(rows, asm_only=False, overwrites=False):
  """The same question along the CONTROL FLOW (r6): from every request, every path -- both sides of a conditional branch,
  loops through their back edges -- is followed until the request's registers are read, overwritten, or the program
  ends; a read is an offender unless a `s_waitcnt vmcnt(N)` with N <= the number of vector memory operations issued
  behind the request ON THAT PATH lies in front of it.  States are (instruction, waited, live registers) with the
  smallest count seen: a loop is walked until its counts stop shrinking.  A WRITE of the request's registers without a
  covering wait in front of it is an offender as well with overwrites=True (the data lands behind the write): that is how
  a request left in flight at a loop's exit shows.  Off by default: the analysis is path-insensitive, and a loop whose
  first iteration hipcc peels merges "only one round" with "more rounds follow" behind the peeled copy -- the phi moves of
  that merge write the next round's request registers on a path no run takes (wfx.h, wdx.h, wsx.h, cgx.h).  A request whose
  registers are untouched up to s_endpgm is not reported."""
  # every instruction once: (kind, wait count, registers read, registers written, branch target)
  END, WAIT, MEM, BR, CBR, OTHER = range(6)
  pre = []
  for t, target in rows:
    reads, writes = _reads_writes(t)
    rd = frozenset().union(*[regs(r) for r in reads]) if reads else frozenset()
    wr = frozenset().union(*[regs(w) for w in writes]) if writes else frozenset()
    n = None
    if t.startswith(('s_endpgm', 's_setpc')):
      kind = END
    elif t.startswith('s_waitcnt') and 'vmcnt' in t:
      m = re.search(r'vmcnt\((\d+)\)', t)
      kind, n = WAIT, (int(m.group(1)) if m else None)
    elif t.startswith(VMEM):
      kind = MEM
    elif t.startswith('s_branch'):
      kind = BR
    elif t.startswith('s_cbranch') and target is not None:
      kind = CBR
    else:
      kind = OTHER
    pre.append((kind, n, rd, wr, target))
  bad = []
  for i, (ins, _) in enumerate(rows):
    parts = ins.split(None, 1)
    if not parts[0].startswith(('buffer_load_dword', 'buffer_load_ubyte', 'buffer_load_ushort')) or 'offen' not in ins or ' lds' in ins:
      continue
    ops = [o.strip() for o in parts[1].split(',')]
    if len(ops) < 4 or (asm_only and not ops[3].startswith('0 ')):
      continue
    dst0 = frozenset(regs(ops[0]))
    work = [(i + 1, 0, False, dst0, False)]
    best = {}
    found = None
    while work and found is None:
      pc, y, w, live, looped = work.pop()
      while pc is not None and pc < len(rows):
        seen = best.setdefault((pc, w, looped), [])               # (count, live registers) of earlier visits: a visit with a count
        if any(y0 <= y and live <= l0 for y0, l0 in seen):   # no larger and no fewer live registers has covered this one
          break
        seen[:] = [(y0, l0) for y0, l0 in seen if not (y <= y0 and l0 <= live)] + [(y, live)]
        kind, n, rd, wr, target = pre[pc]
        if kind == END:
          break
        if kind == WAIT and n is not None and n <= y:
          w = True
        if rd & live:
          if not w:
            found = (i, pc, '%s   <-   %s' % (rows[pc][0], ins))
          break
        if kind == MEM:
          y = min(y + 1, 64)
        if wr & live:
          # a WRITE of a register the request may still be filling: the value is lost when the data lands behind it -- how a
          # request left in flight at a loop's exit shows (the epilogue takes over its registers; r6, the first conv's weight
          # gradient).  hipcc waits in front of such a write for the loads IT tracks; another load into the same registers is
          # the next request of a register ring, not a hazard of this one.
          # (only on paths that have not gone through a back edge since the request: a loop unrolled by two whose second
          # half is conditional pairs "second half skipped" with "loop continues" here, which no run does, and the first
          # half's temporaries share registers with the second half's requests)
          if overwrites and not w and kind != MEM and not looped:
            found = (i, pc, '%s   OVERWRITES (in flight)   %s' % (rows[pc][0], ins))
            break
          live = live - wr
          if not live:
            break
        if kind == BR:
          looped = looped or target <= pc
          pc = target
          continue
        if kind == CBR:
          work.append((target, y, w, live, looped or target <= pc))
        pc += 1
    if found:
      bad.append(found)
  return bad


def main():
  obj, pat = sys.argv[1], sys.argv[2]
  if '--cfg' in sys.argv:
    n_bad = 0
    for co in isa_waits.device_code(obj):
      for name, rows in kernels_cfg(co):
        if pat not in name:
          continue
        bad = check_cfg(rows, asm_only='--all' not in sys.argv, overwrites='--overwrites' in sys.argv)
        print('%s: %d request(s) read before a covering wait on some path' % (name[:110], len(bad)))
        for i, j, t in bad[:10]:
          print('   load @%d read @%d: %s' % (i, j, t))
        n_bad += len(bad)
    return 1 if n_bad else 0
  n_bad = 0
  for co in isa_waits.device_code(obj):
    for name, body in kernels_text(co):
      if pat not in name:
        continue
      bad = check(body, asm_only='--all' not in sys.argv)
      print('%s: %d request(s) read before a wait' % (name[:110], len(bad)))
      for i, j, t in bad[:10]:
        print('   load @%d read @%d: %s' % (i, j, t))
      n_bad += len(bad)
  return 1 if n_bad else 0


if __name__ == '__main__':
  sys.exit(main())

"""Structure of the compiled gfx950 kernels that this round's speed-ups rest on, checked on the ISA (no GPU): the
properties below were found by reading `tools/isa_waits.py` output and each was worth 1-13 % of a kernel when it broke
("what the waits say").  A compiler or source change that silently brings a stall back -- a
vector load of `nvalid` at the head of the time loop, spills in the first conv, an unpinned prefetch -- fails here
instead of showing up as a slower bench line."""
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
LIB = os.path.join(ROOT, 'seed_rl_amd', 'lib', 'libseedhip.so')


@pytest.fixture(scope='module')
def code_objects():
  import isa_waits
  if not os.path.exists(LIB):
    pytest.skip('libseedhip.so is not built')
  for tool in ('llvm-objcopy', 'clang-offload-bundler', 'llvm-readelf', 'llvm-objdump'):
    if not os.path.exists(os.path.join(isa_waits.LLVM, tool)) and shutil.which(tool) is None:
      pytest.skip('ROCm LLVM tools not found')
  cos = isa_waits.device_code(LIB)
  assert cos and cos != [LIB], 'no gfx950 code object inside libseedhip.so'
  return isa_waits, cos


def _find(code_objects, pattern):
  isa_waits, cos = code_objects
  hits = [(co, name, meta) for co in cos for name, meta in isa_waits.kernels(co, pattern).items()]
  assert hits, 'kernel %r is not in the library' % pattern
  return hits


def _ops(code_objects, co, name):
  isa_waits, _ = code_objects
  return [line.split()[1] for line in isa_waits.trace(co, name, False)]


def test_first_conv_reads_nvalid_through_the_scalar_cache(code_objects):
  # a vector load of nvalid[t, b] behind the band prefetch made every wave wait for that prefetch before its first MFMA
  for pat in ('stackconv_fwd_bf16r_kernelILi0ELb0ELb1', 'stackconv_wgrad_tr_kernelILi16'):
    for co, name, meta in _find(code_objects, pat):
      ops = _ops(code_objects, co, name)
      assert 'global_load_ubyte' not in ops, (pat, 'nvalid is read with a vector load again')
      assert any(o.startswith('s_load_dword') for o in ops)


def test_register_budgets_of_the_hot_kernels(code_objects):
  budget = {                                             # pattern -> (max VGPRs, max spilled VGPRs)
      'stackconv_fwd_bf16r_kernelILi0ELb0ELb1ELb0': (168, 16),  # three waves per SIMD (64-bit pointers: tensors >= 2 GB)
      'stackconv_fwd_bf16r_kernelILi0ELb0ELb1ELb1': (168, 0),   # r4, buffer-addressed time loop: NO spill (8 spilled VGPRs were
                                                                 # reloaded behind s_waitcnt vmcnt(0) at the head of every step)
      'stackconv_fwd_bf16r_kernelILi0ELb0ELb0ELb1': (168, 0),
      # r5: the bf16x6 weight gradients (wgx.h), two 4-wave workgroups per CU: no spill in any geometry
      'wgx_kernelINS0_3GeoILi4ELi4ELi2ELi0ELi16ELi32': (256, 0),
      'wgx_kernelINS0_3GeoILi3ELi3ELi1ELi1ELi16ELi16': (256, 0),
      'wgx_kernelINS0_3GeoILi3ELi3ELi1ELi1ELi16ELi32': (256, 0),
      'wgx_kernelINS0_3GeoILi3ELi3ELi1ELi1ELi32ELi32ELi18': (256, 0),
      'wgx_kernelINS0_3GeoILi3ELi3ELi1ELi1ELi32ELi32ELi9': (256, 0),
      # r5: the 16 -> 32 stack-entry layer (fgx.h): 108 weight registers + one unit of input items, two workgroups per CU
      'fgx_kernelINS0_3GeoILi16ELi32': (256, 0),
      'fgx_kernelINS0_3GeoILi32ELi16ELi36ELi48ELi4ELi1ELb1EEELb0': (256, 0),
      'fgx_kernelINS0_3GeoILi32ELi16ELi36ELi48ELi4ELi1ELb1EEELb1': (256, 0),    # + the max-pool backward in its loader
      # r5: the DQN torso's second / third convolution (cgx.h): one 8-wave workgroup per CU, 96 / 108 weight registers
      'cgx_kernelINS0_3GeoILi4ELi32': (256, 0),
      'cgx_kernelINS0_3GeoILi3ELi64ELi9': (256, 0),
      'cgx_kernelINS0_3GeoILi3ELi64ELi7': (256, 0),
      'cgx_dg2_kernel': (256, 8),                              # (tile offsets: one scratch dword per tile)
      'wfx_kernelILb0ELb0ELi0ELb0': (256, 4),
      'wfx_kernelILb0ELb0ELi0ELb1': (256, 6),                   # r5: + the byte mask of its output (one scratch round trip per round)
      'wdx_kernelILi1ELi0': (256, 0),
      'wdx_kernelILi2ELi0': (256, 0),                          # r5: the byte-mask variant
      'wsx_kernelINS0_3GeoILi18ELi24EEELb0': (256, 0),         # r4: ImpalaDeep's 32 -> 32 3x3 layers, forward / data gradient
      'wsx_kernelINS0_3GeoILi18ELi24EEELb1': (256, 0),
      'wsy_kernelINS0_3GeoILi36ELi48EEELb0': (256, 0),         # r4: the 16 -> 16 layers
      'wsy_kernelINS0_3GeoILi36ELi48EEELb1': (256, 0),                           # r4: the same for the data gradient
                       # r4: one 8-wave workgroup per CU; the spilled registers are prologue-only
      'xg8_kernelILi0ELi0': (256, 0),                           # one 8-wave workgroup per CU: two waves per SIMD
      'xg8_kernelILi1ELi0': (256, 0),
      'stackconv_wgrad_tr_kernelILi16': (256, 0),              # r5: one 8-wave workgroup per CU (150 KB of LDS)
      'ws_tab_kernelILi4ELi4ELi1ELi0ELb0ELb1': (128, 0),       # data gradient with the mask a tile ahead: four waves per SIMD
      'ws_tab_kernelILi2ELi8ELi0ELi0ELb0ELb0': (128, 0),
  }
  for pat, (vmax, smax) in budget.items():
    for _, name, meta in _find(code_objects, pat):
      assert meta['.vgpr_count'] <= vmax, (pat, meta)
      assert meta.get('.vgpr_spill_count', 0) <= smax, (pat, meta)


def test_halo_epilogue_is_one_wait_then_stores(code_objects):
  # forward with residual (MT = NT = 2): operands requested in front of the k loop, ONE wait, then the stores back to back
  for co, name, _ in _find(code_objects, 'halo_fwd_kernelILi2ELi2ELb0ELi7'):
    ops = _ops(code_objects, co, name)
    assert not any(o.startswith('flat_load') for o in ops)
    last_mfma = max(i for i, o in enumerate(ops) if o.startswith('v_mfma'))
    tail = [o for o in ops[last_mfma:] if o.startswith(('global_load', 'global_store', 's_waitcnt'))]
    assert not any(o.startswith('global_load') for o in tail), 'epilogue operands are requested behind the k loop again'
    first_store = next(i for i, o in enumerate(tail) if o.startswith('global_store'))
    assert 's_waitcnt' not in tail[first_store:], 'a wait between the output stores'


def test_halo_wgrad_prefetch_is_unconditional(code_objects):
  for co, name, _ in _find(code_objects, 'halo_wgrad_kernelILi9ELi2ELi2ELb1'):
    assert any(o.startswith('buffer_load_dwordx4') for o in _ops(code_objects, co, name))


def test_no_wide_store_followed_by_a_write_of_its_data():
  """A 12/16-byte VMEM store reads its data registers over several cycles; hipcc (ROCm 7.2, gfx950) has been seen to put a
  VALU instruction that REWRITES them into the very next issue slot behind a `buffer_store_dwordx4` -- the stored quad then
  carries the next tile's values (r4: the first conv's forward without ReLU).  No compiled kernel may contain the pattern
  (tools/isa_store_hazard.py; outputs are finished and pinned before the first store where it appeared)."""
  import subprocess, sys
  lib = os.path.join(ROOT, 'seed_rl_amd', 'lib', 'libseedhip.so')
  if not os.path.exists(lib):
    pytest.skip('library not built')
  r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'isa_store_hazard.py'), lib], capture_output=True, text=True,
                     timeout=600)
  assert r.returncode == 0, r.stdout[-3000:]


def test_in_flight_load_registers_are_not_read_before_their_wait():
  """The conv kernels keep input items in flight across a compute phase through asm buffer loads the compiler knows
  nothing about and wait for them by hand.  A copy the compiler makes in front of that wait reads the registers before
  the data arrives -- cgx.h's first build did (a tied "+v" operand of the wait allocated elsewhere), and wsy.h had copied
  the next round's items in its loop pre-header since round 4 (a race its 1 000-cycle head start happened to win).
  tools/isa_inflight.py walks every kernel of the conv objects: no buffer load's destination may be read before a
  `s_waitcnt vmcnt` (xg::take_item / xg::move_item keep the only reads inside or behind the waiting statement)."""
  import isa_inflight, isa_waits
  obj_dir = os.path.join(ROOT, 'build', 'obj')
  seen = 0
  for obj in ('fgx.o', 'cgx.o', 'conv.o', 'wgx.o', 'stackconv.o'):
    path = os.path.join(obj_dir, obj)
    if not os.path.exists(path):
      pytest.skip('%s is not built' % obj)
    for co in isa_waits.device_code(path):
      for name, body in isa_inflight.kernels_text(co):
        seen += 1
        bad = isa_inflight.check(body, asm_only=False)
        assert not bad, (name, bad[:3])
  assert seen >= 20, seen


def test_in_flight_check_counts_the_wait():
  """tools/isa_inflight.py holds a wait to its COUNT: vmcnt(N) covers a request only when at most N vector
  memory operations were issued behind it."""
  import isa_inflight
  load = 'buffer_load_dwordx4 v[4:7], v1, s[0:3], 0 offen'
  store = 'buffer_store_dwordx4 v[8:11], v2, s[4:7], 0 offen'
  use = 'v_add_f32_e32 v12, v4, v4'
  assert not isa_inflight.check([load, store, store, 's_waitcnt vmcnt(2)', use])        # two younger operations: covered
  assert isa_inflight.check([load, store, 's_waitcnt vmcnt(2)', use])                   # one younger: vmcnt(2) may pass early
  assert isa_inflight.check([load, use])                                                # no wait at all
  assert not isa_inflight.check([load, store, 's_waitcnt vmcnt(2)', 's_waitcnt vmcnt(0)', use])


def test_in_flight_registers_along_the_control_flow():
  """tools/isa_inflight.py --cfg (r6): from every buffer request of the conv objects, every path of the compiled kernel --
  both sides of a conditional branch, loops through their back edges -- is followed to the first read of the request's
  registers; a `s_waitcnt vmcnt(N)` with N <= the number of vector memory operations issued behind the request on that
  path must lie in front of it.  This is what holds the first conv's kernels (requests that live across the step loop's
  back edge, waits counted by hand) to their counts.  (The analysis is path-insensitive: fgx.h's loaders count the
  previous unit's output stores, which a wave past the last band used to skip by returning early -- in the last unit only,
  behind which nothing is taken; it now issues them out of range and the kernel passes on every path.)"""
  import isa_inflight, isa_waits
  obj_dir = os.path.join(ROOT, 'build', 'obj')
  seen = 0
  for obj in ('stackconv.o', 'conv.o', 'wgx.o', 'cgx.o', 'fgx.o'):
    path = os.path.join(obj_dir, obj)
    if not os.path.exists(path):
      pytest.skip('%s is not built' % obj)
    for co in isa_waits.device_code(path):
      for name, rows in isa_inflight.kernels_cfg(co):
        # conv.o: the kernels with asm requests (the igemm / gemm.h / halo kernels leave every wait to hipcc: two minutes
        # of path walking for nothing)
        if obj == 'conv.o' and not any(ns in name for ns in ('3wfx', '3wdx', '3wsx', '3wsy', '2xg', '3xg8')):
          continue
        seen += 1
        # writes of registers a request may still be filling (a request left in flight at a loop's exit: the first conv's
        # weight gradient, first build) where the kernels' loops give the path-insensitive walk no infeasible merges
        bad = isa_inflight.check_cfg(rows, asm_only=False, overwrites=obj in ('stackconv.o', 'wgx.o', 'fgx.o'))
        assert not bad, (name, bad[:3])
  assert seen >= 20, seen


def test_control_flow_check_follows_a_loop():
  import isa_inflight
  # request at the loop's end, stores behind it, the take at the next iteration's head: vmcnt(2) covers it, vmcnt(3) does not
  def prog(n):
    return [('s_nop 0', None),
            ('s_waitcnt vmcnt(%d)' % n, None), ('v_add_f32_e32 v12, v4, v4', None),
            ('buffer_load_dwordx4 v[4:7], v1, s[0:3], 0 offen', None),
            ('buffer_store_dwordx4 v[8:11], v2, s[4:7], 0 offen', None), ('buffer_store_dwordx4 v[8:11], v2, s[4:7], 0 offen', None),
            ('s_cbranch_scc1 65530', 1), ('s_endpgm', None)]
  assert not isa_inflight.check_cfg(prog(2))
  assert isa_inflight.check_cfg(prog(3))
  # a request left in flight at the loop's exit: the epilogue writes its registers
  tail = [('buffer_load_dwordx4 v[4:7], v1, s[0:3], 0 offen', None), ('s_cbranch_scc1 65534', 0), ('v_mov_b32_e32 v5, v9', None), ('s_endpgm', None)]
  assert isa_inflight.check_cfg(tail, overwrites=True) and not isa_inflight.check_cfg(tail)
  assert not isa_inflight.check_cfg(tail[:2] + [('s_waitcnt vmcnt(0)', None)] + tail[2:], overwrites=True)


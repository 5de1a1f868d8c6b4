#!/usr/bin/env python
"""Builds profiles/<tag>_traffic.json (HBM bytes per launch of the cfg2 hot kernels) from the condensed PMC table
written by tools/prof_summary.py.

  python tools/make_traffic.py profiles/r01m_cfg2_pmc.csv profiles/r01m_cfg2_traffic.json

Region names are bench.py's (ops.Profiler regions); a region's kernel is the matching row with the largest traffic
(the Dense GEMM template serves the FC layer and the small heads: the FC launch is the big one).  FETCH_SIZE is doubled
as /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950 (the counter reports half of a wide coalesced read
stream); WRITE_SIZE is used as reported.
"""
import csv
import json
import sys

REGIONS = [
    ('conv_fwd[4x4/2 16->32 @20x20]', 'wfx::wfx_kernel'),                      # r4: bf16x6 forward / data gradient
    ('conv_dgrad[4x4/2 16->32 @20x20]', 'wdx::wdx_kernel'),
    ('conv_wgrad[4x4/2 16->32 @20x20]', 'wgx::wgx_kernel'),                    # r5: bf16x6 weight gradient (transposing LDS reads)
    ('conv_fwd[1x1/1 2592->256 @1x1]', 'xg8::xg8_kernel<0'),                   # r4: bf16x6 Dense kernels
    ('conv_wgrad[1x1/1 2592->256 @1x1]', 'xg8::xg8_kernel<1'),
    ('conv_dgrad[1x1/1 2592->256 @1x1]', 'xg8::xg8_kernel<2'),
    ('conv_fwd[1x1/1 2592->256 @1x1]', 'xg::xgemm_kernel<true, false'),
    ('conv_dgrad[1x1/1 2592->256 @1x1]', 'xg::xgemm_kernel<true, true'),
    ('conv_wgrad[1x1/1 2592->256 @1x1]', 'xg::xgemm_kernel<false, false'),
    ('stack_conv_fwd', 'stackconv::stackconv_fwd'),
    ('stack_conv_wgrad', 'stackconv::stackconv_wgrad'),
    ('conv_fwd[4x4/2 16->32 @20x20]', 'wsgemm::ws_tab_kernel<2, 8, 0'),
    ('conv_dgrad[4x4/2 16->32 @20x20]', 'wsgemm::ws_tab_kernel<4, 4, 1'),
    ('conv_fwd[4x4/2 16->32 @20x20]', 'wsgemm::ws_fast_kernel<2, 8, 0>'),     # SEEDHIP_WS_TAB=0
    ('conv_dgrad[4x4/2 16->32 @20x20]', 'wsgemm::ws_fast_kernel<4, 4, 1>'),
    ('conv_fwd[4x4/2 16->32 @20x20]', 'wsgemm::ws_kernel<1, 2,'),            # SEEDHIP_WS_FAST=0
    ('conv_dgrad[4x4/2 16->32 @20x20]', 'wsgemm::ws_kernel<1, 4,'),
    ('conv_wgrad[4x4/2 16->32 @20x20]', 'wsw::wsw_kernel'),
    ('conv_wgrad[4x4/2 16->32 @20x20]', 'false, false, true, false, false, 1>(seedhip::gemm::Params)'),   # SEEDHIP_WSW=0
    ('conv_fwd[1x1/1 2592->256 @1x1]', 'true, false, false, false, false, 2>(seedhip::gemm::Params)'),
    ('conv_dgrad[1x1/1 2592->256 @1x1]', 'true, true, false, false, false, 2>(seedhip::gemm::Params)'),
    ('conv_wgrad[1x1/1 2592->256 @1x1]', 'false, false, false, false, false, 2>(seedhip::gemm::Params)'),
]


REGIONS_CFG3 = [
    ('conv_fwd[3x3/1 16->16 @36x48]', 'wsy::wsy_kernel<seedhip::wsy::Geo<36, 48>, false>'),
    ('conv_dgrad[3x3/1 16->16 @36x48]', 'wsy::wsy_kernel<seedhip::wsy::Geo<36, 48>, true>'),
    ('conv_fwd[3x3/1 32->32 @18x24]', 'wsx::wsx_kernel<seedhip::wsx::Geo<18, 24>, false>'),
    ('conv_dgrad[3x3/1 32->32 @18x24]', 'wsx::wsx_kernel<seedhip::wsx::Geo<18, 24>, true>'),
    ('conv_fwd[3x3/1 32->32 @9x12]', 'wsx::wsx_kernel<seedhip::wsx::Geo<9, 12>, false>'),
    ('conv_dgrad[3x3/1 32->32 @9x12]', 'wsx::wsx_kernel<seedhip::wsx::Geo<9, 12>, true>'),
    ('conv_wgrad[3x3/1 16->16 @36x48]', 'wgx::wgx_kernel<seedhip::wgx::Geo<3, 3, 1, 1, 16, 16'),
    ('conv_wgrad[3x3/1 16->32 @36x48]', 'wgx::wgx_kernel<seedhip::wgx::Geo<3, 3, 1, 1, 16, 32'),
    ('conv_wgrad[3x3/1 32->32 @18x24]', 'wgx::wgx_kernel<seedhip::wgx::Geo<3, 3, 1, 1, 32, 32, 18'),
    ('conv_wgrad[3x3/1 32->32 @9x12]', 'wgx::wgx_kernel<seedhip::wgx::Geo<3, 3, 1, 1, 32, 32, 9'),
    ('conv_fwd[3x3/1 16->32 @36x48]', 'fgx::fgx_kernel<seedhip::fgx::Geo<16, 32'),
    ('conv_dgrad_pool[3x3/1 16->32 @36x48]', 'fgx::fgx_kernel<seedhip::fgx::Geo<32, 16, 36, 48, 4, 1, true>, true>'),
    ('conv_dgrad[3x3/1 16->32 @36x48]', 'fgx::fgx_kernel<seedhip::fgx::Geo<32, 16, 36, 48, 4, 1, true>, false>'),
    ('conv_fwd[3x3/1 16->32 @36x48]', 'halo::halo_fwd_kernel<3, 2, false'),
    ('conv_dgrad[3x3/1 16->32 @36x48]', 'halo::halo_fwd_kernel<3, 1, true'),
    ('convpool_fwd[72x96x3->16]', 'convpool_fwd_mfma_kernel'),
    ('convpool_bwd[72x96x3->16]', 'convpool_bwd_mfma_kernel'),
    ('maxpool_fwd[36x48x32]', 'maxpool_fwd_kernel'),
    ('maxpool_bwd[18x24x32]', 'maxpool_bwd_pair_kernel'),
    ('maxpool_bwd[36x48x32]', 'maxpool_bwd_kernel'),
]


def main():
  src, out = sys.argv[1], sys.argv[2]
  global REGIONS
  cfg = sys.argv[3] if len(sys.argv) > 3 else 'cfg2'
  if cfg == 'cfg3':
    REGIONS = REGIONS_CFG3
  rows = list(csv.reader(open(src)))[1:]
  traffic, fetch, write, kernel = {}, {}, {}, {}
  for region, pat in REGIONS:
    if region in traffic:
      continue                                               # an earlier (newer-kernel) pattern already matched
    best = None
    for r in rows:
      if pat not in r[0] or not r[3] or not r[4]:
        continue
      f, w = float(r[3]), float(r[4])
      if best is None or 2 * f + w > 2 * best[1] + best[2]:
        best = (r[0], f, w)
    if best:
      kernel[region], fetch[region], write[region] = best[0][:100], best[1], best[2]
      traffic[region] = int((2 * best[1] + best[2]) * 1024)
  # MFMA-pipe utilisation of the same kernels from the SQ counters (tools/pmc_mfma.sh / pmc_mfma.py), when condensed
  mfma = {}
  mpath = src.replace('_pmc.csv', '_mfma.csv')
  import os
  if os.path.exists(mpath):
    mrows = list(csv.DictReader(open(mpath)))
    for region, pat in REGIONS:
      if region in mfma:
        continue
      for r in mrows:
        if pat in r['Kernel']:
          mfma[region] = dict(mfma_utilisation=float(r['mfma_utilisation']), effective_clock_GHz=float(r['effective_clock_GHz']),
                              duration_us=float(r['avg_duration_us']))
          break
  # what the profile describes: the kernel sources on the GPU box (tools/profile_round.sh writes their digest next to
  # the rocprofv3 output) and the commit this tree was at when the summary was made -- bench.py prints `traffic` only
  # while HEAD's kernel sources still have that digest
  import subprocess
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  dig_path = src.replace('_pmc.csv', '_csrc.sha256')
  if os.path.exists(dig_path):
    digest = open(dig_path).read().split()[0]
  else:
    sys.path.insert(0, root)
    from seed_rl_amd import build as _b
    digest = _b.csrc_digest()
  try:
    sha = subprocess.check_output(['git', '-C', root, 'rev-parse', 'HEAD'], text=True).strip()
    dirty = bool(subprocess.check_output(['git', '-C', root, 'status', '--porcelain', '--', 'seed_rl_amd/csrc'], text=True).strip())
  except Exception:                                          # pylint: disable=broad-except
    sha, dirty = None, None
  json.dump({
      'git_sha': sha, 'git_csrc_dirty': dirty, 'csrc_sha256': digest,
      'note': 'HBM bytes per launch from rocprofv3 PMC passes (%s): (2 x FETCH_SIZE + WRITE_SIZE) x 1024; FETCH_SIZE '
              'doubled per MI355X_MICROARCH.md (gfx950 reports half of a coalesced read stream)' % src,
      'config': 'cfg2 Atari shallow T=20 B=512 A=18' if cfg == 'cfg2' else 'cfg3 DMLab ImpalaDeep + LSTM T=20 B=256 A=9',
      'traffic_bytes': traffic, 'fetch_kb_raw': fetch, 'write_kb': write, 'kernel': kernel,
      'mfma_pmc': mfma,
      'mfma_note': 'mfma_utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs), separate --pmc pass; '
                   'executed MFMA work incl. structural zeros (conv1 data gradient: border taps of the super-pixel GEMM, +23 %)',
  }, open(out, 'w'), indent=1)
  for k, v in traffic.items():
    print('%-40s %8.1f MB' % (k, v / 1e6))


if __name__ == '__main__':
  main()

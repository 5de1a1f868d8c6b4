"""Host-side enumeration of the two facts seed_rl_amd/csrc/wfx.h's LDS ring rests on (the bf16x6 forward of the second
Atari conv, atari/networks.py:236): (1) rounds r and r + 1 of a run of images never span more than kRU rows of one
parity, so the rows written for round r + 1 cannot land on a row round r still reads; (2) with rows of one parity 41
sixteen-byte slots apart, the 16 lanes of a ds_read_b128 phase hit 16 different slots unless their pixels cross an
image boundary or the ring's wrap.  The constants are read from the header, not restated."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _consts():
  src = open(os.path.join(ROOT, 'seed_rl_amd', 'csrc', 'wfx.h')).read()
  get = lambda n: int(re.search(r'constexpr int %s = (\d+);' % n, src).group(1))
  m = re.search(r'constexpr int kIH = (\d+), kIW = (\d+), kOW = (\d+), kP = (\d+);', src)
  return dict(IH=int(m.group(1)), IW=int(m.group(2)), OW=int(m.group(3)), P=int(m.group(4)), RU=get('kRU'),
              PITCH=get('kPitch'), ROUND=get('kRound'), ITEMS=get('kItems'),
              THREADS=int(re.search(r'__launch_bounds__\((\d+)', src).group(1)))


def _end_row(r, total, rows, c):
  pl = min(c['ROUND'] * r + c['ROUND'] - 1, total - 1)
  li, pix = divmod(pl, c['P'])
  return min(c['IH'] * li + 2 * (pix // c['OW']) + 4, rows)


def test_ring_holds_two_rounds_and_a_set_holds_a_round():
  c = _consts()
  for nimg in (1, 2, 3, 5, 8, 9, 33, 34, 81, 128):
    total, rows = nimg * c['P'], nimg * c['IH']
    rounds = -(-total // c['ROUND'])
    for r in range(rounds):
      li, pix = divmod(c['ROUND'] * r, c['P'])
      first = c['IH'] * li + 2 * (pix // c['OW'])                 # first row round r reads
      last = _end_row(r + 1, total, rows, c) - 1                  # last row written while it reads
      assert (last >> 1) - (first >> 1) + 1 <= c['RU'], (nimg, r)
      new = _end_row(r, total, rows, c) - (_end_row(r - 1, total, rows, c) if r else 0)
      assert 0 <= new * 40 <= c['THREADS'] * c['ITEMS'], (nimg, r, new)    # 32-byte items of a round fit one register set
    assert _end_row(rounds - 1, total, rows, c) == rows


def test_pixel_operand_reads_spread_over_the_banks():
  c = _consts()
  assert c['PITCH'] % 16 == 0 and (c['PITCH'] // 16) % 16 == c['OW'] % 16
  worst, total, n = 0, 0, 0
  for start in range(0, c['P'] * 40, 16):                         # every 16-lane phase of 40 images' tiles
    for ky in range(4):
      for kx in range(4):
        slots = {}
        for l in range(16):
          li, pix = divmod(start + l, c['P'])
          oy, ox = divmod(pix, c['OW'])
          u = (c['IH'] // 2) * li + oy + (ky >> 1)
          a = (u % c['RU']) * c['PITCH'] + (kx & 1) * 160 + (ox + (kx >> 1)) * 16
          s = (a // 16) % 16
          slots[s] = slots.get(s, 0) + 1
        worst = max(worst, max(slots.values())); total += max(slots.values()); n += 1
  assert worst <= 3 and total / n < 1.25, (worst, total / n)


def test_wdx_ring_holds_two_rounds_and_a_set_holds_a_round():
  """seed_rl_amd/csrc/wdx.h (data gradient of the same conv as a super-pixel GEMM): padded dY rows, 11 per image; rounds
  r and r + 1 together never span more than the ring's rows, a round never brings more new rows than one register set
  of 32-byte items holds, and rows one super-pixel row apart continue the 16-byte slot sequence."""
  src = open(os.path.join(ROOT, 'seed_rl_amd', 'csrc', 'wdx.h')).read()
  get = lambda n: int(re.search(r'constexpr int %s = (\d+)' % n, src).group(1))
  R, ROUND, ITEMS, SP, PR = get('kR'), get('kRound'), get('kItems'), get('kSP'), get('kPR')
  rs = re.search(r'constexpr int kRS = (\d+) \* 16;', src)
  assert int(rs.group(1)) % 16 == 10 % 16                    # ten super-pixels per row
  threads = int(re.search(r'__launch_bounds__\((\d+)', src).group(1))

  def end_row(r, total, rows):
    s = min(ROUND * r + ROUND - 1, total - 1)
    li, sp = divmod(s, SP)
    return min(PR * li + sp // 10 + 2, rows)

  for nimg in (1, 2, 3, 8, 9, 10, 33, 34, 64):
    total, rows = nimg * SP, nimg * PR
    rounds = -(-total // ROUND)
    for r in range(rounds):
      li, sp = divmod(ROUND * r, SP)
      first = PR * li + sp // 10                              # dy = 1 of the round's first super-pixel
      assert end_row(r + 1, total, rows) - first <= R, (nimg, r)
      new = end_row(r, total, rows) - (end_row(r - 1, total, rows) if r else 0)
      assert 0 <= new * 44 <= threads * ITEMS, (nimg, r, new)
    assert end_row(rounds - 1, total, rows) == rows


def test_wsx_rings_and_row_pitch():
  """seed_rl_amd/csrc/wsx.h (ImpalaDeep's 32 -> 32 3x3 'same' layers): for both served maps, rounds r and r + 1 never
  span more padded rows than the ring holds, a round's new rows fit one register set of 32-byte items, and the row pitch
  continues the 16-byte slot sequence from one image row to the next (= map width mod 16)."""
  src = open(os.path.join(ROOT, 'seed_rl_amd', 'csrc', 'wsx.h')).read()
  ROUND = int(re.search(r'constexpr int kRound = (\d+);', src).group(1))
  ITEMS = int(re.search(r'constexpr int kItems = (\d+);', src).group(1))
  threads = int(re.search(r'__launch_bounds__\((\d+)', src).group(1))
  assert 'kR = W >= 24 ? 16 : 32' in src and 'kRSslots = 4 * kWP + ((W - 4 * kWP) % 16 + 16) % 16' in src
  for H, W in ((18, 24), (9, 12)):
    HP, WP, PX = H + 2, W + 2, H * W
    R = 16 if W >= 24 else 32
    rs = 4 * WP + ((W - 4 * WP) % 16 + 16) % 16
    assert rs % 16 == W % 16 and rs >= 4 * WP
    for nimg in (1, 2, 3, 7, 21, 22):
      total, rows = nimg * PX, nimg * HP

      def end_row(r):
        pl = min(ROUND * r + ROUND - 1, total - 1)
        li, rem = divmod(pl, PX)
        return min(HP * li + rem // W + 3, rows)
      rounds = -(-total // ROUND)
      for r in range(rounds):
        li, rem = divmod(ROUND * r, PX)
        first = HP * li + rem // W
        assert end_row(r + 1) - first <= R, (H, W, nimg, r)
        new = end_row(r) - (end_row(r - 1) if r else 0)
        assert 0 <= new * 4 * WP <= threads * ITEMS, (H, W, nimg, r, new)
      assert end_row(rounds - 1) == rows


def test_wsy_ring_and_tiles():
  """seed_rl_amd/csrc/wsy.h (ImpalaDeep's 16 -> 16 3x3 layers on 36 x 48 maps, rounds of 256 pixels): two rounds span at
  most the ring's 16 padded rows, a round's new rows fit one register set, and a 16-pixel tile never crosses an image
  row (48 and 36 x 48 are multiples of 16), which is what makes its operand reads conflict free."""
  src = open(os.path.join(ROOT, 'seed_rl_amd', 'csrc', 'wsy.h')).read()
  ROUND = int(re.search(r'constexpr int kRound = (\d+);', src).group(1))
  ITEMS = int(re.search(r'constexpr int kItems = (\d+);', src).group(1))
  R = int(re.search(r'static constexpr int kR = (\d+);', src).group(1))
  threads = int(re.search(r'__launch_bounds__\((\d+)', src).group(1))
  H, W = 36, 48
  HP, WP, PX = H + 2, W + 2, H * W
  assert W % 16 == 0 and PX % 16 == 0 and ROUND % 16 == 0
  for nimg in (1, 2, 3, 20, 21):
    total, rows = nimg * PX, nimg * HP

    def end_row(r):
      pl = min(ROUND * r + ROUND - 1, total - 1)
      li, rem = divmod(pl, PX)
      return min(HP * li + rem // W + 3, rows)
    rounds = -(-total // ROUND)
    for r in range(rounds):
      li, rem = divmod(ROUND * r, PX)
      assert end_row(r + 1) - (HP * li + rem // W) <= R, (nimg, r)
      new = end_row(r) - (end_row(r - 1) if r else 0)
      assert 0 <= new * 2 * WP <= threads * ITEMS, (nimg, r, new)
    assert end_row(rounds - 1) == rows

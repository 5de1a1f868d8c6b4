import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
import torch.nn.functional as F
from seed_rl_amd import ops
dev = torch.device('cuda')
for n in (1024, 1026):
  rng = np.random.default_rng(n)
  x = rng.normal(size=(n, 9, 9, 64)).astype(np.float32)
  wt = (rng.normal(size=(3, 3, 64, 64)) / 24).astype(np.float32)
  g = ops.conv_geom(n, 9, 9, 64, 3, 3, 1, 'valid', 64)
  out = torch.full((n, 7, 7, 64), 7.0, device=dev)
  ops.conv2d_fwd(g, torch.tensor(x, device=dev), torch.tensor(wt, device=dev), None, out)
  ref = F.conv2d(torch.tensor(x).permute(0, 3, 1, 2).double(), torch.tensor(wt).permute(3, 2, 0, 1).double()).permute(0, 2, 3, 1).numpy()
  got = out.cpu().numpy().astype(np.float64)
  err = np.abs(got - ref)
  print(n, 'max err', err.max(), 'bad frac', (err > 1e-4).mean())
  bad = np.argwhere(err > 1e-4)
  if len(bad):
    print('first bad', bad[:5], 'last bad', bad[-3:])
    imgs = np.unique(bad[:, 0]); print('bad images', imgs[:20], len(imgs))
    chans = np.unique(bad[:, 3]); print('bad channels', chans[:70], len(chans))
    pix = np.unique(bad[:, 1] * 7 + bad[:, 2]); print('bad pixels', pix)
    i0 = tuple(bad[0]); print('got', got[i0], 'ref', ref[i0])

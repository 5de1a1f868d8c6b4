import sys, os
sys.path.insert(0, os.getcwd())
import torch
from seed_rl_amd import ops
n=8448
g = ops.conv_geom(n, 20, 20, 16, 4, 4, 2, 'valid', 32)
x = torch.randn((n,20,20,16), device='cuda'); w = torch.randn((4,4,16,32), device='cuda')/16; b = torch.randn(32, device='cuda')
out = torch.empty((n,9,9,32), device='cuda')
ops.conv2d_fwd(g, x, w, b, out, out_relu=True)
torch.cuda.synchronize()

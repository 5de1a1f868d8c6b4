import sys, os
sys.path.insert(0, os.getcwd())
import torch
from seed_rl_amd import ops
dev = torch.device('cuda')
for B, H in [(256, 512), (256, 256)]:
    hin = torch.randn(B, H, device=dev); cin = torch.randn(B, H, device=dev)
    U = torch.randn(H, 4 * H, device=dev) / H ** 0.5; zx = torch.randn(B, 4 * H, device=dev)
    up = torch.empty_like(U); ops.lstm_permute_u(U, H, up)
    z = torch.empty(B, 4 * H, device=dev); h = torch.empty(B, H, device=dev); hn = torch.empty_like(h); cn = torch.empty_like(h)
    done = torch.zeros(B, dtype=torch.uint8, device=dev)
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3): ops.lstm_step_fwd(hin, up, zx, cin, done, B, H, z, h, H, hn, cn)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, capture_error_mode='relaxed'):
            for _ in range(100): ops.lstm_step_fwd(hin, up, zx, cin, done, B, H, z, h, H, hn, cn)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g.replay(); torch.cuda.synchronize()
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    print('B=%d H=%d fused step: %.2f us per step (graph of 100)' % (B, H, e0.elapsed_time(e1) * 10))
    gu = ops.dense_geom(B, H, 4 * H)
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        for _ in range(3):
            ops.conv2d_fwd(gu, hin, U, None, z, residual=zx); ops.lstm_gates_fwd(z, cin, done, B, H, h, H, hn, cn)
        torch.cuda.synchronize()
        with torch.cuda.graph(g2, capture_error_mode='relaxed'):
            for _ in range(100):
                ops.conv2d_fwd(gu, hin, U, None, z, residual=zx); ops.lstm_gates_fwd(z, cin, done, B, H, h, H, hn, cn)
    torch.cuda.synchronize()
    g2.replay(); torch.cuda.synchronize()
    e0.record(); g2.replay(); e1.record(); torch.cuda.synchronize()
    print('B=%d H=%d gemm + gates: %.2f us per step (graph of 100)' % (B, H, e0.elapsed_time(e1) * 10))

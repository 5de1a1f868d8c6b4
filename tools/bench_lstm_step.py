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
    T1 = 100
    if ops.lstm_seq_supported(T1, B, H):
        zx3 = torch.randn(T1, B, 4 * H, device=dev); z3 = torch.empty_like(zx3); h3 = torch.empty(T1 * B, H, device=dev)
        hin3 = torch.randn(T1 + 1, B, H, device=dev); cin3 = torch.randn(T1 + 1, B, H, device=dev)
        done3 = torch.zeros(T1, B, dtype=torch.uint8, device=dev); sync = torch.zeros(2, dtype=torch.int32, device=dev)
        for mode in os.environ.get('SEQ_MODES', '1').split(','):
            os.environ['SEEDHIP_LSTM_SEQ_MODE'] = mode
            for _ in range(3): ops.lstm_seq_fwd(up, zx3, done3, T1, B, H, z3, h3, H, hin3, cin3, sync)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(10): ops.lstm_seq_fwd(up, zx3, done3, T1, B, H, z3, h3, H, hin3, cin3, sync)
            e1.record(); torch.cuda.synchronize()
            print('B=%d H=%d whole-unroll kernel mode %s: %.2f us per step (T1 = 100, abort flag %d)' % (B, H, mode, e0.elapsed_time(e1), int(sync[1])))
        ring = torch.empty(ops.lstm_seq_bwd_workspace_bytes(B, H) // 4, device=dev)
        dz3 = torch.empty_like(zx3); dh3 = torch.randn(T1 * B, H, device=dev) * 0.01
        for dbg in os.environ.get('SEQ_DBG', '0').split(','):
            os.environ['SEEDHIP_LSTM_SEQ_FAULT'] = dbg
            for _ in range(3): ops.lstm_seq_bwd(up, zx3, cin3, dh3, H, done3, T1, B, H, dz3, ring, sync)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(10): ops.lstm_seq_bwd(up, zx3, cin3, dh3, H, done3, T1, B, H, dz3, ring, sync)
            e1.record(); torch.cuda.synchronize()
            print('B=%d H=%d whole-recurrence backward dbg %s: %.2f us per step (T1 = 100, abort flag %d)' % (B, H, dbg, e0.elapsed_time(e1), int(sync[1])))
        os.environ.pop('SEEDHIP_LSTM_SEQ_FAULT', None)

"""Winograd F(3x3,2x2) weight gradient (asr_tap_wgrad) against the direct kernels (asr_tap_wgrad_direct) on the DFCNN layer shapes:
time, TFLOP/s in direct-conv flops, max difference."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from asr_dfcnn_transformer_amd import ops
from asr_dfcnn_transformer_amd.ops import Plane

B = int(os.environ.get('B', 32))
SHAPES = [('c1_1 800x100 32->32 (m2)', 800, 100, 32, 32), ('c2 800x100 32->64', 800, 100, 32, 64), ('c2_1 400x50 64->64', 400, 50, 64, 64), ('c3 400x50 64->128', 400, 50, 64, 128),
          ('c4 200x25 128->128', 200, 25, 128, 128), ('c6 200x25 128->256', 200, 25, 128, 256), ('c5 200x25 32->256', 200, 25, 32, 256)]
if os.environ.get('ONLY'):
    SHAPES = [s for s in SHAPES if s[0].startswith(os.environ['ONLY'])]
if os.environ.get('SHAPE'):                      # SHAPE=H,W,cin,cout[;H,W,cin,cout...]  (batch from B)
    SHAPES = [('%sx%s %s->%s' % tuple(t.split(',')),) + tuple(int(v) for v in t.split(',')) for t in os.environ['SHAPE'].split(';')]


def timeit(fn, iters=10):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for name, H, W, cin, cout in SHAPES:
    g = torch.Generator(device='cuda').manual_seed(0)
    x = Plane(B, H, W, cin); x.set_interior(torch.randn(B, H, W, cin, device='cuda', generator=g))
    dz = Plane(B, H, W, cout); dz.set_interior(torch.randn(B, H, W, cout, device='cuda', generator=g))
    d = ops.gemm_desc(x.NP, cin, cout, cin, cout, ntaps=9, B=B, H=H, W=W)
    ws = torch.zeros(ops.tap_wgrad_workspace(d) // 4 + 64, device='cuda')
    dw0, dw1 = torch.zeros(9 * cin * cout, device='cuda'), torch.zeros(9 * cin * cout, device='cuda')
    t0 = timeit(lambda: ops.tap_wgrad(d, x, dz, cout, dw0, ws, direct=True))
    t1 = timeit(lambda: ops.tap_wgrad(d, x, dz, cout, dw1, ws))
    fl = 2.0 * B * H * W * 9 * cin * cout
    err = (dw0 - dw1).abs().max().item() / dw0.abs().max().item()
    print('%-22s wgrad direct %7.1f us %6.1f TF | winograd %7.1f us %6.1f TF-equivalent (x%.2f) | max|diff| / max %.2e'
          % (name, 1e3 * t0, fl / t0 / 1e9, 1e3 * t1, fl / t1 / 1e9, t0 / t1, err), flush=True)

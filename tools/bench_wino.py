"""EXPERIMENTAL Winograd F(2x2,3x3) conv (asr_tap_gemm_wino) against the tap-GEMM on fragment-order weights (asr_tap_gemm_pw)
on the DFCNN layer shapes: time, TFLOP/s in direct-conv flops, max difference."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from asr_dfcnn_transformer_amd import ops
from asr_dfcnn_transformer_amd.ops import Plane

B = int(os.environ.get('B', 32))
SHAPES = [('c2 800x100 32->64', 800, 100, 32, 64), ('c2_1 400x50 64->64', 400, 50, 64, 64), ('c3 400x50 64->128', 400, 50, 64, 128),
          ('c4 200x25 128->128', 200, 25, 128, 128), ('c6 200x25 128->256', 200, 25, 128, 256), ('c5 200x25 32->256', 200, 25, 32, 256),
          ('small 6x5 8->64', 6, 5, 8, 64), ('small 4x2 16->64', 4, 2, 16, 64)]
if os.environ.get('ONLY'):
    SHAPES = [s for s in SHAPES if s[0].startswith(os.environ['ONLY'])]


def timeit(fn, iters=10):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for name, H, W, cin, cout in SHAPES:
    Bq = B if H > 10 else 3
    g = torch.Generator(device='cuda').manual_seed(0)
    x = Plane(Bq, H, W, cin); x.set_interior(torch.randn(Bq, H, W, cin, device='cuda', generator=g))
    w = torch.randn(3, 3, cin, cout, device='cuda', generator=g) * (2.0 / (9 * cin)) ** 0.5
    bias = torch.randn(cout, device='cuda', generator=g) * 0.1
    sc = 1 + 0.2 * torch.randn(cout, device='cuda', generator=g); sh = 0.1 * torch.randn(cout, device='cuda', generator=g)
    a0, y0, a1, y1 = Plane(Bq, H, W, cout), Plane(Bq, H, W, cout), Plane(Bq, H, W, cout), Plane(Bq, H, W, cout)
    d = ops.gemm_desc(x.NP, cin, cout, cin, cout, cout, cout, ntaps=9, B=Bq, H=H, W=W, relu=1)
    wf = ops.arrange_weights(w, 9, cin, cout, cout, 0)
    assert ops.winograd_supported(d), name
    wt = ops.winograd_weights(w, cin, cout, cout, 0)
    t0 = timeit(lambda: ops.tap_gemm_pw(d, x, wf, bias, sc, sh, a0, y0))
    t1 = timeit(lambda: ops.tap_gemm_wino(d, x, wt, bias, sc, sh, a1, y1))
    fl = 2.0 * Bq * H * W * 9 * cin * cout
    ea = (a0.interior() - a1.interior()).abs().max().item(); ey = (y0.interior() - y1.interior()).abs().max().item()
    border = (a1.buf.abs().sum() - a1.interior().abs().sum()).item()
    print('%-22s fwd   pw %7.1f us %6.1f TF | wino %7.1f us %6.1f TF-equivalent (x%.2f) | max|diff| a %.2e y %.2e  border %.1e'
          % (name, 1e3 * t0, fl / t0 / 1e9, 1e3 * t1, fl / t1 / 1e9, t0 / t1, ea, ey, border), flush=True)
    if cin % 64 == 0:
        dz = Plane(Bq, H, W, cout); dz.set_interior(torch.randn(Bq, H, W, cout, device='cuda', generator=g))
        dx0, dx1 = Plane(Bq, H, W, cin), Plane(Bq, H, W, cin)
        bd = ops.gemm_desc(x.NP, cout, cin, cout, cout, 0, cin, ntaps=9, B=Bq, H=H, W=W, wmode=1)
        wfd = ops.arrange_weights(w, 9, cout, cin, cout, 1)
        wtd = ops.winograd_weights(w, cout, cin, cout, 1)
        t0 = timeit(lambda: ops.tap_gemm_pw(bd, dz, wfd, None, None, None, None, dx0))
        t1 = timeit(lambda: ops.tap_gemm_wino(bd, dz, wtd, None, None, None, None, dx1))
        err = (dx0.interior() - dx1.interior()).abs().max().item()
        print('%-22s dgrad pw %7.1f us %6.1f TF | wino %7.1f us %6.1f TF-equivalent (x%.2f) | max|diff| %.2e'
              % ('', 1e3 * t0, fl / t0 / 1e9, 1e3 * t1, fl / t1 / 1e9, t0 / t1, err), flush=True)

"""A/B of the experimental split-bf16 tap-GEMM (asr_tap_gemm_bx6) against the fp32-MFMA one on the DFCNN layer shapes:
forward conv and data-gradient (through the pre-split weight view), time and error against the fp32 kernel / float64."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from asr_dfcnn_transformer_amd import ops
from asr_dfcnn_transformer_amd.ops import Plane

B = int(os.environ.get('B', 32))
SHAPES = [('c1_1 800x100 32->32', 800, 100, 32, 32), ('c2 800x100 32->64', 800, 100, 32, 64),
          ('c2_1 400x50 64->64', 400, 50, 64, 64), ('c3 400x50 64->128', 400, 50, 64, 128),
          ('c4 200x25 128->128', 200, 25, 128, 128), ('c6 200x25 128->256', 200, 25, 128, 256)]


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
    w = torch.randn(3, 3, cin, cout, device='cuda', generator=g) * (2.0 / (9 * cin)) ** 0.5
    bias = torch.randn(cout, device='cuda', generator=g) * 0.1
    y32, y6 = Plane(B, H, W, cout), Plane(B, H, W, cout)
    d = ops.gemm_desc(x.NP, cin, cout, cin, cout, cout, 0, ntaps=9, B=B, H=H, W=W, relu=1)
    ws = ops.split_weights(w, 9, cin, cout, cout, 0)
    t32 = timeit(lambda: ops.tap_gemm(d, x, w, bias, None, None, y32, None))
    t6 = timeit(lambda: ops.tap_gemm_bx6(d, x, ws, bias, None, None, y6, None))
    fl = 2.0 * B * H * W * 9 * cin * cout
    err = (y32.interior() - y6.interior()).abs().max().item()
    # float64 reference on one image row block
    xr = x.interior()[:1].double().permute(0, 3, 1, 2)
    ref = torch.relu(torch.nn.functional.conv2d(xr, w.double().permute(3, 2, 0, 1), bias.double(), padding=1)).permute(0, 2, 3, 1)
    e32 = (y32.interior()[:1].double() - ref).abs().max().item(); e6 = (y6.interior()[:1].double() - ref).abs().max().item()
    print('%-22s fwd  fp32 %7.1f us %6.1f TF | bx6 %7.1f us %6.1f TF (x%.2f) | max|fp32-bx6| %.2e  err vs f64: fp32 %.2e bx6 %.2e'
          % (name, 1e3 * t32, fl / t32 / 1e9, 1e3 * t6, fl / t6 / 1e9, t32 / t6, err, e32, e6))
    # data-gradient: dX = dZ (*) W^T mirrored == forward conv of dZ with the wmode-1 view
    dz = Plane(B, H, W, cout); dz.set_interior(torch.randn(B, H, W, cout, device='cuda', generator=g))
    dx32, dx6 = Plane(B, H, W, cin), Plane(B, H, W, cin)
    bd = ops.gemm_desc(x.NP, cout, cin, cout, cout, 0, cin, ntaps=9, B=B, H=H, W=W, wmode=1)
    wsd = ops.split_weights(w, 9, cout, cin, cout, 1)
    t32 = timeit(lambda: ops.tap_gemm(bd, dz, w, None, None, None, None, dx32))
    t6 = timeit(lambda: ops.tap_gemm_bx6(bd, dz, wsd, None, None, None, None, dx6))
    err = (dx32.interior() - dx6.interior()).abs().max().item()
    print('%-22s dgrad fp32 %7.1f us %6.1f TF | bx6 %7.1f us %6.1f TF (x%.2f) | max|fp32-bx6| %.2e'
          % ('', 1e3 * t32, fl / t32 / 1e9, 1e3 * t6, fl / t6 / 1e9, t32 / t6, err))


DENSE = [('dense 6400x6400x1536', 6400, 6400, 1536), ('proj 32768x512x512', 32768, 512, 512), ('ffn1 32768x512x2048', 32768, 512, 2048),
         ('ffn2 32768x2048x512', 32768, 2048, 512)]
for name, M, K, N in DENSE:
    g = torch.Generator(device='cuda').manual_seed(1)
    a = torch.randn(M, K, device='cuda', generator=g)
    w = torch.randn(K, N, device='cuda', generator=g) * (1.0 / K) ** 0.5
    y32, y6 = torch.zeros(M, N, device='cuda'), torch.zeros(M, N, device='cuda')
    d = ops.gemm_desc(M, K, N, K, N, 0, N, ntaps=1)
    ws = ops.split_weights(w, 1, K, N, N, 0)
    t32 = timeit(lambda: ops.tap_gemm(d, a, w, None, None, None, None, y32))
    t6 = timeit(lambda: ops.tap_gemm_bx6(d, a, ws, None, None, None, None, y6))
    fl = 2.0 * M * K * N
    ref = a[:256].double() @ w.double()
    print('%-22s fwd  fp32 %7.1f us %6.1f TF | bx6 %7.1f us %6.1f TF (x%.2f) | err vs f64: fp32 %.2e bx6 %.2e'
          % (name, 1e3 * t32, fl / t32 / 1e9, 1e3 * t6, fl / t6 / 1e9, t32 / t6,
             (y32[:256].double() - ref).abs().max().item(), (y6[:256].double() - ref).abs().max().item()))


print('--- weight gradient')
for name, H, W, cin, cout in SHAPES:
    if cout <= 64:
        continue
    g = torch.Generator(device='cuda').manual_seed(2)
    x = Plane(B, H, W, cin); x.set_interior(torch.randn(B, H, W, cin, device='cuda', generator=g))
    dz = Plane(B, H, W, cout); dz.set_interior(torch.randn(B, H, W, cout, device='cuda', generator=g))
    wd = ops.gemm_desc(x.NP, cin, cout, cin, cout, ntaps=9, B=B, H=H, W=W)
    ws = torch.zeros(ops.tap_wgrad_workspace(wd) // 4 + 16, device='cuda')
    g32, g6 = torch.zeros(3, 3, cin, cout, device='cuda'), torch.zeros(3, 3, cin, cout, device='cuda')
    t32 = timeit(lambda: ops.tap_wgrad(wd, x, dz, cout, g32, ws))
    t6 = timeit(lambda: ops.tap_wgrad_bx6(wd, x, dz, cout, g6, ws))
    fl = 2.0 * B * H * W * 9 * cin * cout
    # float64 reference of one tap (centre) on the whole batch
    ref = torch.einsum('bhwk,bhwn->kn', x.interior().double(), dz.interior().double())
    e32 = (g32[1, 1].double() - ref).abs().max().item() / ref.abs().max().item()
    e6 = (g6[1, 1].double() - ref).abs().max().item() / ref.abs().max().item()
    print('%-22s wgrad fp32 %7.1f us %6.1f TF | bx6 %7.1f us %6.1f TF (x%.2f) | max|fp32-bx6| %.2e (scale %.1f)  rel err vs f64: fp32 %.2e bx6 %.2e'
          % (name, 1e3 * t32, fl / t32 / 1e9, 1e3 * t6, fl / t6 / 1e9, t32 / t6, (g32 - g6).abs().max().item(), g32.abs().max().item(), e32, e6))

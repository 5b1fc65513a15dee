"""A/B of the fp32 contraction on pre-arranged weights (asr_arrange_weights + asr_tap_gemm_pw, tap_gemm_kernel_v5) against
asr_tap_gemm on the DFCNN layer shapes (forward conv and data-gradient) and the dense / Transformer GEMM shapes.
(The tile sweeps behind the launcher's rules used an ASR_PW_CFG switch that has since been removed.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
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


for name, H, W, cin, cout in ([] if os.environ.get('DENSE_ONLY') else SHAPES):
    g = torch.Generator(device='cuda').manual_seed(0)
    x = Plane(B, H, W, cin); x.set_interior(torch.randn(B, H, W, cin, device='cuda', generator=g))
    w = torch.randn(3, 3, cin, cout, device='cuda', generator=g) * (2.0 / (9 * cin)) ** 0.5
    bias = torch.randn(cout, device='cuda', generator=g) * 0.1
    y0, y1 = Plane(B, H, W, cout), Plane(B, H, W, cout)
    d = ops.gemm_desc(x.NP, cin, cout, cin, cout, cout, 0, ntaps=9, B=B, H=H, W=W, relu=1)
    wf = ops.arrange_weights(w, 9, cin, cout, cout, 0)
    t0 = timeit(lambda: ops.tap_gemm(d, x, w, bias, None, None, y0, None))
    t1 = timeit(lambda: ops.tap_gemm_pw(d, x, wf, bias, None, None, y1, None))
    fl = 2.0 * B * H * W * 9 * cin * cout
    err = (y0.interior() - y1.interior()).abs().max().item()
    print('%-22s fwd   v1 %7.1f us %6.1f TF | pw %7.1f us %6.1f TF (x%.2f) | max|diff| %.2e'
          % (name, 1e3 * t0, fl / t0 / 1e9, 1e3 * t1, fl / t1 / 1e9, t0 / t1, err), flush=True)
    dz = Plane(B, H, W, cout); dz.set_interior(torch.randn(B, H, W, cout, device='cuda', generator=g))
    dx0, dx1 = Plane(B, H, W, cin), Plane(B, H, W, cin)
    bd = ops.gemm_desc(x.NP, cout, cin, cout, cout, 0, cin, ntaps=9, B=B, H=H, W=W, wmode=1)
    wfd = ops.arrange_weights(w, 9, cout, cin, cout, 1)
    t0 = timeit(lambda: ops.tap_gemm(bd, dz, w, None, None, None, None, dx0))
    t1 = timeit(lambda: ops.tap_gemm_pw(bd, dz, wfd, None, None, None, None, dx1))
    err = (dx0.interior() - dx1.interior()).abs().max().item()
    print('%-22s dgrad v1 %7.1f us %6.1f TF | pw %7.1f us %6.1f TF (x%.2f) | max|diff| %.2e'
          % ('', 1e3 * t0, fl / t0 / 1e9, 1e3 * t1, fl / t1 / 1e9, t0 / t1, err), flush=True)

DENSE = [('dense 6400x6400x1536', 6400, 6400, 1536), ('hidden 6400x6400x128', 6400, 6400, 128), ('proj 32768x512x512', 32768, 512, 512),
         ('ffn1 32768x512x2048', 32768, 512, 2048), ('ffn2 32768x2048x512', 32768, 2048, 512), ('ragged 1000x72x100', 1000, 72, 100)]
for name, M, K, N in ([] if os.environ.get('CONV_ONLY') else DENSE):
    g = torch.Generator(device='cuda').manual_seed(1)
    a = torch.randn(M, K, device='cuda', generator=g)
    w = torch.randn(K, N, device='cuda', generator=g) * (1.0 / K) ** 0.5
    bias = torch.randn(N, device='cuda', generator=g) * 0.1
    y0, y1 = torch.zeros(M, N, device='cuda'), torch.zeros(M, N, device='cuda')
    d = ops.gemm_desc(M, K, N, K, N, N, 0, ntaps=1, relu=1)
    wf = ops.arrange_weights(w, 1, K, N, N, 0)
    t0 = timeit(lambda: ops.tap_gemm(d, a, w, bias, None, None, y0, None))
    t1 = timeit(lambda: ops.tap_gemm_pw(d, a, wf, bias, None, None, y1, None))
    fl = 2.0 * M * K * N
    ref = torch.relu(a[:256].double() @ w.double() + bias.double())
    print('%-22s fwd   v1 %7.1f us %6.1f TF | pw %7.1f us %6.1f TF (x%.2f) | err vs f64: v1 %.2e pw %.2e'
          % (name, 1e3 * t0, fl / t0 / 1e9, 1e3 * t1, fl / t1 / 1e9, t0 / t1, (y0[:256].double() - ref).abs().max().item(),
             (y1[:256].double() - ref).abs().max().item()), flush=True)
    # data-gradient view: dA = dY . W^T
    dy = torch.randn(M, N, device='cuda', generator=g)
    da0, da1 = torch.zeros(M, K, device='cuda'), torch.zeros(M, K, device='cuda')
    bd = ops.gemm_desc(M, N, K, N, N, 0, K, ntaps=1, wmode=1)
    wfd = ops.arrange_weights(w, 1, N, K, N, 1)
    t0 = timeit(lambda: ops.tap_gemm(bd, dy, w, None, None, None, None, da0))
    t1 = timeit(lambda: ops.tap_gemm_pw(bd, dy, wfd, None, None, None, None, da1))
    print('%-22s dgrad v1 %7.1f us %6.1f TF | pw %7.1f us %6.1f TF (x%.2f) | max|diff| %.2e'
          % ('', 1e3 * t0, fl / t0 / 1e9, 1e3 * t1, fl / t1 / 1e9, t0 / t1, (da0 - da1).abs().max().item()), flush=True)

#!/usr/bin/env python3
"""Times the contraction kernels on the exact layer shapes of the DFCNN graphs (B=32, T_pad=1600),
one shape at a time, many launches each (HIP events).  Used to A/B kernel variants:
    ASR_TAPGEMM_VARIANT=1 python tools/bench_layers.py      # previous generation
    python tools/bench_layers.py                            # current
Prints TFLOP/s (algorithmic flops: interior pixels only)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from asr_dfcnn_transformer_amd import ops
from asr_dfcnn_transformer_amd.ops import Plane

B = int(os.environ.get('B', 32))
SHAPES = [  # (name, H, W, cin, cout)
    ('c1_1 800x100 32->32', 800, 100, 32, 32), ('c2 800x100 32->64', 800, 100, 32, 64),
    ('c2_1 400x50 64->64', 400, 50, 64, 64), ('c3 400x50 64->128', 400, 50, 64, 128),
    ('c4 200x25 128->128', 200, 25, 128, 128), ('c6 200x25 128->256', 200, 25, 128, 256),
    ('m1 200x25 32->256', 200, 25, 32, 256)]
DENSE = [('dense 6400x6400x1536', 6400, 6400, 1536), ('dense 6400x6400x128', 6400, 6400, 128),
         ('proj 32768x512x512', 32768, 512, 512), ('ffn1 32768x512x2048', 32768, 512, 2048)]


def timeit(fn, iters=10):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    which = sys.argv[1:] or ['fwd', 'dgrad', 'wgrad', 'dense']
    for name, H, W, cin, cout in SHAPES:
        x, a, y = Plane(B, H, W, cin), Plane(B, H, W, cout), Plane(B, H, W, cout)
        x.interior().normal_(); a.interior().normal_()
        w = torch.randn(9 * cin * cout, device='cuda') * 0.05
        bias = torch.zeros(cout, device='cuda'); sc = torch.ones(cout, device='cuda')
        fl = 2.0 * B * H * W * 9 * cin * cout
        line = '%-22s' % name
        if 'fwd' in which:
            d = ops.gemm_desc(x.NP, cin, cout, cin, cout, cout, cout, ntaps=9, B=B, H=H, W=W, relu=1)
            ms = timeit(lambda: ops.tap_gemm(d, x, w, bias, sc, bias, a, y))
            line += ' fwd %7.1f us %6.1f TF' % (ms * 1e3, fl / ms / 1e9)
        if 'dgrad' in which:
            d = ops.gemm_desc(x.NP, cout, cin, cout, cout, 0, cin, ntaps=9, B=B, H=H, W=W, wmode=1)
            ms = timeit(lambda: ops.tap_gemm(d, a, w, None, None, None, None, x))
            line += ' | dgrad %7.1f us %6.1f TF' % (ms * 1e3, fl / ms / 1e9)
        if 'wgrad' in which:
            d = ops.gemm_desc(x.NP, cin, cout, cin, cout, ntaps=9, B=B, H=H, W=W)
            ws = torch.zeros(max(4, ops.tap_wgrad_workspace(d) // 4), device='cuda')
            dw = torch.zeros(9 * cin * cout, device='cuda')
            ms = timeit(lambda: ops.tap_wgrad(d, x, a, cout, dw, ws))
            line += ' | wgrad %7.1f us %6.1f TF' % (ms * 1e3, fl / ms / 1e9)
        print(line, flush=True)
        del x, a, y
    if 'dense' in which:
        for name, M, K, N in DENSE:
            xx = torch.randn(M, K, device='cuda'); ww = torch.randn(K, N, device='cuda') * 0.02
            out = torch.zeros(M, N, device='cuda'); dx = torch.zeros(M, K, device='cuda')
            fl = 2.0 * M * K * N
            d = ops.gemm_desc(M, K, N, K, N, 0, N, ntaps=1)
            ms1 = timeit(lambda: ops.tap_gemm(d, xx, ww, None, None, None, None, out))
            d2 = ops.gemm_desc(M, N, K, N, N, 0, K, ntaps=1, wmode=1)
            ms2 = timeit(lambda: ops.tap_gemm(d2, out, ww, None, None, None, None, dx))
            d3 = ops.gemm_desc(M, K, N, K, N, ntaps=1)
            ws = torch.zeros(max(4, ops.tap_wgrad_workspace(d3) // 4), device='cuda')
            dw = torch.zeros(K * N, device='cuda')
            ms3 = timeit(lambda: ops.tap_wgrad(d3, xx, out, N, dw, ws))
            print('%-22s fwd %7.1f us %6.1f TF | dgrad %7.1f us %6.1f TF | wgrad %7.1f us %6.1f TF' %
                  (name, ms1 * 1e3, fl / ms1 / 1e9, ms2 * 1e3, fl / ms2 / 1e9, ms3 * 1e3, fl / ms3 / 1e9), flush=True)


if __name__ == '__main__':
    main()

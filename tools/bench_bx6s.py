"""A/B of the experimental pre-split split-bf16 GEMM (asr_split_rows + asr_gemm_bx6s, csrc/gemm_bx6.hip) against the fp32
MFMA GEMM (asr_tap_gemm, one tap) on the Transformer shapes: forward, data-gradient view, time and error against float64.
ASR_BX6S_CFG=n selects a tile configuration of the kernel under test."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from asr_dfcnn_transformer_amd import ops


def timeit(fn, iters=20):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


DENSE = [('proj 32768x512x512', 32768, 512, 512), ('ffn1 32768x512x2048', 32768, 512, 2048),
         ('ffn2 32768x2048x512', 32768, 2048, 512), ('vocab 32768x512x6348', 32768, 512, 6348),
         ('dense 6400x6400x1536', 6400, 6400, 1536), ('ragged 1000x72x100', 1000, 72, 100)]
for name, M, K, N in DENSE:
    g = torch.Generator(device='cuda').manual_seed(1)
    a = torch.randn(M, K, device='cuda', generator=g)
    w = torch.randn(K, N, device='cuda', generator=g) * (1.0 / K) ** 0.5
    bias = torch.randn(N, device='cuda', generator=g) * 0.1
    y32, y6 = torch.zeros(M, N, device='cuda'), torch.zeros(M, N, device='cuda')
    d = ops.gemm_desc(M, K, N, K, N, N, 0, ntaps=1, relu=1)
    ws = ops.split_weights(w, 1, K, N, N, 0)                     # [3][N][Kp]
    as_ = ops.split_rows(a, M, K, K)
    t32 = timeit(lambda: ops.tap_gemm(d, a, w, bias, None, None, y32, None))
    tsp = timeit(lambda: ops.split_rows(a, M, K, K, as_))
    t6 = timeit(lambda: ops.gemm_bx6s(as_, ws, M, K, N, bias, 1, 0, y6, N))
    fl = 2.0 * M * K * N
    rows = slice(0, min(M, 512))
    ref = torch.relu(a[rows].double() @ w.double() + bias.double())
    e32 = (y32[rows].double() - ref).abs().max().item(); e6 = (y6[rows].double() - ref).abs().max().item()
    print('%-24s fp32 %7.1f us %6.1f TF | bx6s %7.1f us %6.1f TF (x%.2f; split pass %6.1f us, %.2f TB/s) | err vs f64: fp32 %.2e bx6s %.2e'
          % (name, 1e3 * t32, fl / t32 / 1e9, 1e3 * t6, fl / t6 / 1e9, t32 / t6, 1e3 * tsp, 10.0 * M * K / tsp / 1e9, e32, e6), flush=True)
    # split output feeding the next GEMM: Ysplit must equal split_rows(Y)
    if N % 32 == 0:
        ysp = torch.zeros(ops._lib.load().asr_split_rows_bytes(M, N), dtype=torch.uint8, device='cuda')
        ops.gemm_bx6s(as_, ws, M, K, N, bias, 1, 0, y6, N, ysp)
        want = ops.split_rows(y6, M, N, N)
        print('%-24s split epilogue identical to split_rows(Y): %s' % ('', bool(torch.equal(ysp, want))), flush=True)

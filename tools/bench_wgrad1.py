"""Dense (1-tap) weight-gradient on the Transformer / head shapes: time and TFLOP/s, error vs float64."""
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


for name, M, K, N in [('proj 32768x512x512', 32768, 512, 512), ('ffn1 32768x512x2048', 32768, 512, 2048),
                      ('ffn2 32768x2048x512', 32768, 2048, 512), ('vocab 32768x512x6348', 32768, 512, 6348),
                      ('dense 6400x6400x1536', 6400, 6400, 1536), ('lm 6400x128x128', 6400, 128, 128)]:
    g = torch.Generator(device='cuda').manual_seed(1)
    x = torch.randn(M, K, device='cuda', generator=g)
    dy = torch.randn(M, N, device='cuda', generator=g)
    dw = torch.zeros(K, N, device='cuda')
    d = ops.gemm_desc(M, K, N, K, N, ntaps=1)
    ws = torch.zeros(ops.tap_wgrad_workspace(d) // 4 + 64, device='cuda')
    t = timeit(lambda: ops.tap_wgrad(d, x, dy, N, dw, ws))
    ref = x[:, :64].double().t() @ dy[:, :64].double()
    err = (dw[:64, :64].double() - ref).abs().max().item() / ref.abs().max().item()
    print('%-24s %8.1f us %6.1f TF  rel err %.1e' % (name, 1e3 * t, 2.0 * M * K * N / t / 1e9, err), flush=True)

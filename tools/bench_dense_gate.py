"""asr_tap_gemm_gated_dense against the two passes it replaces (asr_tap_gemm data-gradient + asr_cell_bwd_pre, layout 2) at the head shapes of
the models, B = 32: plain DFCNN 6400 -> 128 (acoustic_model.py:49), SE-DFCNN 6400 -> 1536 (acoustic_model2.py:66).  LIB=<other build>."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from asr_dfcnn_transformer_amd import _lib
if os.environ.get('LIB'):
    _lib.LIB_PATH = os.path.abspath(os.environ['LIB'])
import torch
from asr_dfcnn_transformer_amd import ops
from asr_dfcnn_transformer_amd.ops import Plane


def timeit(fn, iters=20):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


g = torch.Generator(device='cuda').manual_seed(0)
rnd = lambda *s: torch.randn(*s, device='cuda', generator=g)
B, H, W, Cc = 32, 200, 25, 256
for K in (128, 1536):
    rows, cin = B * H, W * Cc
    dzd = rnd(rows, K); w = rnd(cin, K) * 0.05
    a = Plane(B, H, W, Cc); a.set_interior(torch.relu(rnd(B, H, W, Cc)))
    sc = 1.0 + 0.2 * rnd(Cc); sh = 0.1 * rnd(Cc)
    d = ops.gemm_desc(rows, K, cin, K, K, 0, cin, ntaps=1, wmode=1)
    dflat = torch.zeros(rows, cin, device='cuda')
    dz0, dz1 = Plane(B, H, W, Cc), Plane(B, H, W, Cc)
    r = [torch.zeros(Cc, device='cuda') for _ in range(3)]
    ws = torch.zeros(max(ops.cell_bwd_pre_workspace(B, H, W, Cc), ops.tap_gemm_gated_dense_workspace(d, W, Cc)) // 4 + 64, device='cuda')
    t0 = timeit(lambda: ops.tap_gemm(d, dzd, w, None, None, None, None, dflat)); k0 = ops.last_kernel()
    t1 = timeit(lambda: ops.cell_bwd_pre(dflat, 2, a, sc, sh, 0, dz0, r[0], r[1], r[2], ws))
    t2 = timeit(lambda: ops.tap_gemm_gated_dense(d, dzd, w, a, sc, sh, dz1, r[0], r[1], r[2], ws)); k2 = ops.last_kernel()
    print('6400 -> %4d: data-gradient %s %7.1f us + asr_cell_bwd_pre %6.1f us = %7.1f us | fused %s (+ fold) %7.1f us | same dZ: %s'
          % (K, k0, t0, t1, t0 + t1, k2, t2, torch.equal(dz0.buf, dz1.buf)), flush=True)

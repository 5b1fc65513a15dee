"""The two max-pooled cells of the plain DFCNN step (32 -> 64 at 800 x 100, 64 -> 128 at 400 x 50; B = 32) in the form with the pre-pool
activation plane (asr_tap_gemm_wino_pool / asr_tap_gemm_gated pool 2) and in the compact form (asr_tap_gemm_wino_poolmax /
asr_tap_gemm_gated_poolmax), forward and gated data-gradient, one process.  usage: python tools/bench_poolmax.py   (LIB=<other build>)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from asr_dfcnn_transformer_amd import _lib
if os.environ.get('LIB'):
    _lib.LIB_PATH = os.path.abspath(os.environ['LIB'])
import torch
from asr_dfcnn_transformer_amd import ops
from asr_dfcnn_transformer_amd.ops import Plane

B = int(os.environ.get('B', 32))


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
for name, H, W, cin, N, K2 in (('c2 32->64 / h3 128->64', 800, 100, 32, 64, 128), ('c3 64->128 / h4 128->128', 400, 50, 64, 128, 128)):
    x = Plane(B, H, W, cin); x.set_interior(rnd(B, H, W, cin))
    w = rnd(3, 3, cin, N) * (2.0 / (9 * cin)) ** 0.5
    bias = rnd(N) * 0.1; sc = 1 + 0.2 * rnd(N); sh = 0.1 * rnd(N)
    H2, W2 = H // 2, W // 2
    fd = ops.gemm_desc(x.NP, cin, N, cin, N, N, 0, ntaps=9, B=B, H=H, W=W, relu=1)
    wt = ops.winograd_weights(w, cin, N, N, 0)
    a, y1, y2, amax = Plane(B, H, W, N), Plane(B, H2, W2, N), Plane(B, H2, W2, N), Plane(B, H2, W2, N)
    idx = ops.poolmax_index(B, H2, W2, N)
    t1 = timeit(lambda: ops.tap_gemm_wino_pool(fd, x, wt, bias, sc, sh, a, 2, y1)); k1 = ops.last_kernel()
    t2 = timeit(lambda: ops.tap_gemm_wino_poolmax(fd, x, wt, bias, sc, sh, y2, amax, idx)); k2 = ops.last_kernel()
    print('%-26s forward  %s %7.1f us | %s %7.1f us | same y: %s' % (name, k1, t1, k2, t2, torch.equal(y1.buf, y2.buf)), flush=True)
    w2 = rnd(3, 3, N, K2) * 0.05
    dzk = Plane(B, H2, W2, K2); dzk.set_interior(rnd(B, H2, W2, K2))
    bd = ops.gemm_desc(dzk.NP, K2, N, K2, K2, 0, N, ntaps=9, B=B, H=H2, W=W2, wmode=1)
    wtb = ops.winograd_weights(w2, K2, N, K2, 1)
    ws = torch.zeros(ops.tap_gemm_gated_workspace(bd) // 4 + 64, device='cuda')
    dz1, dz2 = Plane(B, H, W, N), Plane(B, H, W, N)
    s = [torch.zeros(N, device='cuda') for _ in range(3)]
    t1 = timeit(lambda: ops.tap_gemm_gated(bd, dzk, wtb, 2, 2, a, sc, sh, None, dz1, s[0], s[1], s[2], ws)); k1 = ops.last_kernel()
    t2 = timeit(lambda: ops.tap_gemm_gated_poolmax(bd, dzk, wtb, H, W, amax, idx, sc, sh, None, dz2, s[0], s[1], s[2], ws)); k2 = ops.last_kernel()
    print('%-26s backward %s %7.1f us | %s %7.1f us | same dZ: %s' % (name, k1, t1, k2, t2, torch.equal(dz1.buf, dz2.buf)), flush=True)
    del x, a, y1, y2, amax, dz1, dz2, dzk

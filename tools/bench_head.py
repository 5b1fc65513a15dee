"""The dense-head GEMMs of the DFCNN graphs (6400 rows = 32 utterances x 200 frames) in the three directions, as the library routes
them (asr_tap_gemm_nt / asr_tap_gemm wmode 1 / asr_tap_wgrad): time and TFLOP/s.  The narrow ones are what the engine runs split-K."""
import sys, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from asr_dfcnn_transformer_amd import ops
def timeit(fn, iters=20):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
g = torch.Generator(device='cuda').manual_seed(0)
for (M, cin, cout) in [(6400, 6400, 128), (6400, 128, 1536), (6400, 6400, 1536)]:
    x = torch.randn(M, cin, device='cuda', generator=g); w = torch.randn(cin, cout, device='cuda', generator=g) * 0.02
    dy = torch.randn(M, cout, device='cuda', generator=g); dx = torch.zeros(M, cin, device='cuda'); y = torch.zeros(M, cout, device='cuda')
    b = torch.zeros(cout, device='cuda')
    fd = ops.gemm_desc(M, cin, cout, cin, cout, cout, 0, ntaps=1, relu=1)
    bd = ops.gemm_desc(M, cout, cin, cout, cout, 0, cin, ntaps=1, wmode=1)
    wd = ops.gemm_desc(M, cin, cout, cin, cout, ntaps=1)
    wT = w.t().contiguous()
    ws = torch.zeros(ops.tap_wgrad_workspace(wd) // 4 + 64, device='cuda'); dw = torch.zeros(cin * cout, device='cuda')
    fl = 2.0 * M * cin * cout
    t0 = timeit(lambda: ops.tap_gemm_nt(fd, x, w, wT, cin, b, None, None, y, None))
    t1 = timeit(lambda: ops.tap_gemm(bd, dy, w, None, None, None, None, dx))
    t2 = timeit(lambda: ops.tap_wgrad(wd, x, dy, cout, dw, ws))
    print('M %d  %d -> %d: fwd %.1f us %.1f TF | dgrad %.1f us %.1f TF | wgrad %.1f us %.1f TF' % (M, cin, cout, 1e3*t0, fl/t0/1e9, 1e3*t1, fl/t1/1e9, 1e3*t2, fl/t2/1e9))
# the split-K forms of the narrow layers: 64 x 64 register-staged tiles (asr_tap_gemm_splitk) against the LDS-DMA kernel (asr_tap_gemm_nt_splitk)
for (M, K, N, s_old, s_new) in [(6400, 6400, 128, 8, 10), (6400, 1536, 128, 8, 8), (6400, 6400, 128, 8, 8), (6400, 6400, 128, 8, 5)]:
    x = torch.randn(M, K, device='cuda', generator=g); w = torch.randn(K, N, device='cuda', generator=g) * 0.02
    wT = w.t().contiguous(); b = torch.zeros(N, device='cuda'); y = torch.zeros(M, N, device='cuda')
    d = ops.gemm_desc(M, K, N, K, N, N, N, ntaps=1, relu=1)
    ws = torch.zeros(ops.tap_gemm_nt_splitk_workspace(d, 16) // 4 + 4, device='cuda')
    fl = 2.0 * M * K * N
    t0 = timeit(lambda: ops.tap_gemm_splitk(d, x, w, b, None, None, None, y, s_old, ws))
    t1 = timeit(lambda: ops.tap_gemm_nt_splitk(d, x, wT, K, b, None, None, None, y, s_new, ws)); k1 = ops.last_kernel()
    print('M %d  %d -> %d: split-K x%d on 64 x 64 tiles %.1f us %.1f TF | x%d on %s %.1f us %.1f TF' % (M, K, N, s_old, 1e3*t0, fl/t0/1e9, s_new, k1, 1e3*t1, fl/t1/1e9))

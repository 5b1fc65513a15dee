"""The dense GEMMs of Language_Model (B 64 x T 100 = 6400 rows) in the three directions as the library routes them: time, TFLOP/s."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from asr_dfcnn_transformer_amd import ops
def timeit(fn, iters=30):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
g = torch.Generator(device='cuda').manual_seed(0)
M = int(os.environ.get('M', 6400))
for (cin, cout) in [(512, 1536), (512, 512), (512, 2048), (2048, 512), (512, 6348)]:
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
    print('M %d  %d -> %d: fwd %.1f us %.1f TF | dgrad %.1f us %.1f TF | wgrad %.1f us %.1f TF' % (M, cin, cout, 1e3*t0, fl/t0/1e9, 1e3*t1, fl/t1/1e9, 1e3*t2, fl/t2/1e9), flush=True)

"""The five Winograd forward layers of the plain DFCNN (acoustic_model.py) at B = 32, 10 s audio, timed alone."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from asr_dfcnn_transformer_amd import ops
from asr_dfcnn_transformer_amd.ops import Plane
def timeit(fn, iters=10):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
B = 32
for (H, W, cin, cout, both) in ((800, 100, 32, 64, 0), (400, 50, 64, 128, 0), (200, 25, 128, 128, 1), (200, 25, 128, 256, 1), (200, 25, 32, 256, 1)):
    x = Plane(B, H, W, cin); x.interior().normal_()
    w = torch.randn(3, 3, cin, cout, device='cuda') * 0.05
    a1 = Plane(B, H, W, cout); y1 = Plane(B, H, W, cout) if both else None
    bias = torch.zeros(cout, device='cuda'); sc = torch.ones(cout, device='cuda'); sh = torch.zeros(cout, device='cuda')
    d = ops.gemm_desc(x.NP, cin, cout, cin, cout, cout, cout if both else 0, ntaps=9, B=B, H=H, W=W, relu=1)
    wt = ops.winograd_weights(w, cin, cout, cout, 0)
    t1 = timeit(lambda: ops.tap_gemm_wino(d, x, wt, bias, sc, sh, a1, y1))
    print('%dx%d %d->%d: wino %.1f us' % (H, W, cin, cout, 1e3 * t1), flush=True)

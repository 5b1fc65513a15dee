"""How much of the forward conv kernel's gap is the partial last round of workgroups?  Sweeps the batch size on the
128->128 200x25 layer: tiles = ceil(B*201*26/128) * 2 on 768 resident slots (3 workgroups per CU)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from asr_dfcnn_transformer_amd import ops
from asr_dfcnn_transformer_amd.ops import Plane
from bench_layers import timeit
H, W, cin, cout = 200, 25, 128, 128
for B in (24, 28, 29, 30, 32, 34, 36, 37, 38, 40, 44, 45, 48, 64, 128):
    x, a, y = Plane(B, H, W, cin), Plane(B, H, W, cout), Plane(B, H, W, cout)
    x.interior().normal_()
    w = torch.randn(9 * cin * cout, device='cuda') * 0.05
    wf = ops.arrange_weights(w, 9, cin, cout, cout, 0)
    bias = torch.zeros(cout, device='cuda'); sc = torch.ones(cout, device='cuda')
    d = ops.gemm_desc(x.NP, cin, cout, cin, cout, cout, cout, ntaps=9, B=B, H=H, W=W, relu=1)
    ms = timeit(lambda: ops.tap_gemm_pw(d, x, wf, bias, sc, bias, a, y), iters=20)
    tiles = -(-x.NP // 128) * (cout // 64)
    fl = 2.0 * B * H * W * 9 * cin * cout
    print('B %3d tiles %5d rounds %.2f  %7.1f us  %6.1f TF' % (B, tiles, tiles / 768.0, ms * 1e3, fl / ms / 1e9), flush=True)
    del x, a, y

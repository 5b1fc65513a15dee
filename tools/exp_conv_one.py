"""One layer, forward conv on pre-arranged weights, many launches: the target of a rocprofv3 --pmc pass."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from asr_dfcnn_transformer_amd import ops
from asr_dfcnn_transformer_amd.ops import Plane
B, H, W, cin, cout = 32, 200, 25, 128, 128
x, a, y = Plane(B, H, W, cin), Plane(B, H, W, cout), Plane(B, H, W, cout)
x.interior().normal_()
w = torch.randn(9 * cin * cout, device='cuda') * 0.05
wf = ops.arrange_weights(w, 9, cin, cout, cout, 0)
bias = torch.zeros(cout, device='cuda'); sc = torch.ones(cout, device='cuda')
d = ops.gemm_desc(x.NP, cin, cout, cin, cout, cout, cout, ntaps=9, B=B, H=H, W=W, relu=1)
for _ in range(12):
    ops.tap_gemm_pw(d, x, wf, bias, sc, bias, a, y)
torch.cuda.synchronize()

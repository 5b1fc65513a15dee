"""The pre-net's stride-2 data-gradient alone at the bench size (B 64, 512 x 80 plane, C 64): asr_conv_s2_dgrad (one launch per phase) against the
one 4-tap asr_tap_gemm.  LIB=<other build> for an A/B of tile configurations."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from asr_dfcnn_transformer_amd import _lib
if os.environ.get('LIB'):
    _lib.LIB_PATH = os.path.abspath(os.environ['LIB'])
import torch
from asr_dfcnn_transformer_amd import ops

B, H2, W2, C = 64, 512, 80, 64
g = torch.Generator(device='cuda').manual_seed(0)
dz = ops.Plane(B, H2, W2, C); dz.interior().normal_(generator=g)
dx1, dx2 = ops.Plane(B, H2, W2, 4 * C), ops.Plane(B, H2, W2, 4 * C)
w = torch.randn(3, 3, C, C, device='cuda', generator=g) * 0.05
W4 = torch.zeros(4, 4 * C, C, device='cuda')
ops.conv_s2_expand(w, C, C, W4)
Wf9 = torch.zeros(ops.conv_s2_arrange_bytes(C) // 4, device='cuda')
ops.conv_s2_arrange(W4, C, Wf9)
d = ops.gemm_desc(dz.NP, C, 4 * C, C, C, 0, 4 * C, ntaps=4, B=B, H=H2, W=W2, wmode=1)


def timeit(fn, n=10):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


t_old = timeit(lambda: ops.tap_gemm(d, dz, W4, None, None, None, None, dx1))
t_new = timeit(lambda: ops.conv_s2_dgrad(d, dz, Wf9, dx2))
err = (dx1.interior() - dx2.interior()).abs().max().item() / dx1.interior().abs().max().item()
print('stride-2 data-gradient: one 4-tap GEMM %.2f ms, per phase %.2f ms (196 GFLOP useful: %.0f / %.0f TFLOP/s), max difference %.1e of scale' %
      (t_old, t_new, 0.196 / t_old * 1e3, 0.196 / t_new * 1e3, err))

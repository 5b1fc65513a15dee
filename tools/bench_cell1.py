"""First cell (Cin = 1) forward / backward alone at the full size (B 32, 1600 x 200, 32 channels): LIB=<other build> for an A/B."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from asr_dfcnn_transformer_amd import _lib
if os.environ.get('LIB'):
    _lib.LIB_PATH = os.path.abspath(os.environ['LIB'])
from asr_dfcnn_transformer_amd import ops

B, T, F, C = int(os.environ.get('B', 32)), 1600, 200, 32


def timeit(fn, iters=10):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


g = torch.Generator(device='cuda').manual_seed(0)
x = torch.randn(B, T, F, device='cuda', generator=g)
w = torch.randn(9 * C, device='cuda', generator=g) * 0.3
bias = torch.randn(C, device='cuda', generator=g) * 0.1
sc = 1 + 0.2 * torch.randn(C, device='cuda', generator=g)
sh = 0.1 * torch.randn(C, device='cuda', generator=g)
for pm, name in ((1, 'avg'), (2, 'max')):
    y = ops.Plane(B, T // 2, F // 2, C)
    dy = ops.Plane(B, T // 2, F // 2, C); dy.interior().normal_()
    dw, db, dsc, dsh = (torch.zeros(9 * C, device='cuda'), torch.zeros(C, device='cuda'), torch.zeros(C, device='cuda'), torch.zeros(C, device='cuda'))
    ws = torch.zeros(ops.cell1_bwd_workspace(B, T, F, C) // 4 + 4, device='cuda')
    tf = timeit(lambda: ops.cell1_fwd(x, w, bias, sc, sh, pm, y))
    tb = timeit(lambda: ops.cell1_bwd(x, w, bias, sc, sh, pm, dy, dw, db, dsc, dsh, ws))
    print('%s: fwd %.1f us  bwd %.1f us  (checksums %.6e %.6e)' % (name, tf, tb, y.interior().double().sum().item(), dw.double().sum().item()), flush=True)

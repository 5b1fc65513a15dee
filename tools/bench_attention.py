"""Times asr_attention_fwd/bwd (causal and not) on random ReLU'd projections; run on the GPU box."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from asr_dfcnn_transformer_amd import _lib
if os.environ.get('LIB'):
    _lib.LIB_PATH = os.path.abspath(os.environ['LIB'])
import torch
from asr_dfcnn_transformer_amd import ops

N, T, C, H = 64, 512, 512, 8
g = torch.Generator(device='cuda').manual_seed(0)
Q, K, V, dO = [torch.randn(N, T, C, device='cuda', generator=g).relu_() for _ in range(4)]
O = torch.zeros_like(Q); lse = torch.zeros(2, N, H, T, device='cuda')
gq, gk, gv = torch.zeros_like(Q), torch.zeros_like(Q), torch.zeros_like(Q)
ws = torch.zeros(N * H * T + 16, device='cuda')
flops_fwd = 4.0 * N * H * T * T * 64
for causal in (False, True):
    for name, fn, fl in (('fwd', lambda: ops.attention_fwd(Q, K, V, N, T, T, C, H, causal, O, lse), flops_fwd),
                         ('bwd', lambda: ops.attention_bwd(Q, K, V, O, dO, lse, N, T, T, C, H, causal, gq, gk, gv, ws), 2.5 * flops_fwd)):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print('causal=%d %s %.3f ms  %.1f TF (dense-equivalent)' % (causal, name, ms, fl / ms / 1e9))
    print('lse max row0..3', lse[0, 0, 0, :4].tolist())

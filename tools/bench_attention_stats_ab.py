"""asr_attention_fwd_s / _bwd_s with precomputed statistics at the Transformer's size; LIB=<other build> for an A/B on one box."""
import os, sys
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
stats = torch.zeros(ops.attention_stats_floats(N, T, T, H), device='cuda')
ops.attention_stats(Q, K, N, T, T, C, H, stats)
def timeit(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 10
for causal in (False, True):
    tf = timeit(lambda: ops.attention_fwd(Q, K, V, N, T, T, C, H, causal, O, lse, dropout_rate=0.2, seed=3, stats=stats))
    tb = timeit(lambda: ops.attention_bwd(Q, K, V, O, dO, lse, N, T, T, C, H, causal, gq, gk, gv, ws, relu_grad=True, dropout_rate=0.2, seed=3, stats=stats))
    print('causal=%d  fwd %.3f ms  bwd %.3f ms   (checksums %.6e %.6e)' % (causal, tf, tb, float(O.double().sum()), float(gk.double().abs().sum())))

"""Times the attention entry points at a small sequence length (Language_Model: T = 100; joint AM+LM: T = 200) with dropout, as the
engines call them (row statistics taken once per call).  usage (GPU box): [LIB=tools/libasrhip_x.so] python3 tools/bench_attention_small.py [T] [N] [causal]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from asr_dfcnn_transformer_amd import _lib
if os.environ.get('LIB'):
    _lib.LIB_PATH = os.path.abspath(os.environ['LIB'])
import torch
from asr_dfcnn_transformer_amd import ops

T = int(sys.argv[1]) if len(sys.argv) > 1 else 100
N = int(sys.argv[2]) if len(sys.argv) > 2 else 64
causal = bool(int(sys.argv[3])) if len(sys.argv) > 3 else True
C, H = 512, 8
g = torch.Generator(device='cuda').manual_seed(0)
Q, K, V, dO = [torch.randn(N, T, C, device='cuda', generator=g).relu_() for _ in range(4)]
O = torch.zeros_like(Q); lse = torch.zeros(2, N, H, T, device='cuda')
gq, gk, gv = torch.zeros_like(Q), torch.zeros_like(Q), torch.zeros_like(Q)
ws = torch.zeros(N * H * T + 16, device='cuda')
stats = torch.zeros(ops.attention_stats_floats(N, T, T, H), device='cuda')


def timed(fn, reps=200):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps


ops.attention_stats(Q, K, N, T, T, C, H, stats)
f = timed(lambda: ops.attention_fwd(Q, K, V, N, T, T, C, H, causal, O, lse, dropout_rate=0.2, seed=7, stats=stats))
b = timed(lambda: ops.attention_bwd(Q, K, V, O, dO, lse, N, T, T, C, H, causal, gq, gk, gv, ws, relu_grad=1, dropout_rate=0.2, seed=7, stats=stats))
st = timed(lambda: ops.attention_stats(Q, K, N, T, T, C, H, stats))
f0 = timed(lambda: ops.attention_fwd(Q, K, V, N, T, T, C, H, causal, O, lse, dropout_rate=0.2, seed=7))
b0 = timed(lambda: ops.attention_bwd(Q, K, V, O, dO, lse, N, T, T, C, H, causal, gq, gk, gv, ws, relu_grad=1, dropout_rate=0.2, seed=7))
print('   statistics launch %.1f us; without precomputed statistics: fwd %.1f us  bwd %.1f us   (sum with %.1f, without %.1f)' % (st, f0, b0, st + f + b, f0 + b0))
print('T %d N %d causal %d: fwd %.1f us  bwd (delta + dK/dV + dQ) %.1f us   checksums %.6e %.6e %.6e' %
      (T, N, causal, f, b, float(O.double().sum()), float(gq.double().sum()), float(gk.double().sum())))

"""asr_attention_stats at the Transformer's size (B 64 x T 512 x 8 heads), dense rows and the fused [rows][3C] projection buffer."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from asr_dfcnn_transformer_amd import ops
N, T, C, H = 64, 512, 512, 8
g = torch.Generator(device='cuda').manual_seed(0)
buf = torch.randn(N * T, 3 * C, device='cuda', generator=g).relu_()
Q = torch.randn(N, T, C, device='cuda', generator=g).relu_(); K = torch.randn(N, T, C, device='cuda', generator=g).relu_()
stats = torch.zeros(ops.attention_stats_floats(N, T, T, H), device='cuda')
def timeit(fn, iters=30):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
print('dense Q, K:             %.1f us' % timeit(lambda: ops.attention_stats(Q, K, N, T, T, C, H, stats)))
print('column blocks of [3C]:  %.1f us' % timeit(lambda: ops.attention_stats(buf[:, :C], buf[:, C:], N, T, T, C, H, stats, ldq=3 * C, ldk=3 * C)))

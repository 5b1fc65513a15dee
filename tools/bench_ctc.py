"""The loss head's row kernels at the headline size (B = 32, T8 = 200 frames of which 125 are valid, V = 1536, 32 labels): asr_softmax_log_fwd,
asr_ctc_loss (check + rows + lattice), asr_softmax_log_bwd, the bias column sum, greedy decode.  LIB=<other build> for an A/B on one box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from asr_dfcnn_transformer_amd import _lib
if os.environ.get('LIB'):
    _lib.LIB_PATH = os.path.abspath(os.environ['LIB'])
import numpy as np, torch
from asr_dfcnn_transformer_amd import ops


def timeit(fn, iters=50):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


B, T8, V, ML = 32, 200, 1536, 64
g = torch.Generator(device='cuda').manual_seed(0)
d = torch.randn(B * T8, V, device='cuda', generator=g)
logits = torch.zeros(T8, B, V, device='cuda'); grad = torch.zeros(T8, B, V, device='cuda'); dd = torch.zeros(B * T8, V, device='cuda')
lab = np.zeros((B, ML), dtype=np.int32); lab[:, :32] = np.random.default_rng(0).integers(1, V - 1, (B, 32))
labels = torch.from_numpy(lab).cuda(); ll = torch.full((B,), 32, dtype=torch.int32, device='cuda'); sl = torch.full((B,), 125, dtype=torch.int32, device='cuda')
loss = torch.zeros(B, device='cuda'); status = torch.zeros(B, dtype=torch.int32, device='cuda')
ws = torch.zeros(ops.ctc_workspace(T8, B, ML) // 8 + 8, dtype=torch.float64, device='cuda')
bias = torch.zeros(V, device='cuda'); cws = torch.zeros(ops.colsum_workspace(B * T8, V) // 4 + 64, device='cuda')
print('softmax_log_fwd  %7.1f us' % timeit(lambda: ops.softmax_log_fwd(d, B, T8, V, 1e-7, logits)))
print('ctc_loss         %7.1f us' % timeit(lambda: ops.ctc_loss(logits, T8, B, V, labels, ML, ll, sl, V - 1, loss, grad, status, ws)))
print('softmax_log_bwd  %7.1f us' % timeit(lambda: ops.softmax_log_bwd(logits, grad, B, T8, V, 1e-7, 1.0 / B, dd)))
print('colsum 6400x1536 %7.1f us' % timeit(lambda: ops.colsum(dd, B * T8, V, V, bias, cws)))
print('loss[0:4]', loss[:4].tolist(), 'grad checksum %.9e' % float(grad.double().abs().sum()), 'dd checksum %.9e' % float(dd.double().abs().sum()))

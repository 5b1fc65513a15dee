"""Experiment: the latency-bound language half of the joint graph (d_model 128: ~110 forward / ~250 backward launches of
10-80 us) eagerly enqueued against replayed from a captured hipGraph (torch.cuda.CUDAGraph around the ctypes launches).
Dropout must be off: the dropout seeds are kernel scalars and would be frozen into the graph."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from asr_dfcnn_transformer_amd.joint_engine import AMLMEngine

B = 32
eng = AMLMEngine(B=B, blocks=12, pos_max=200, dropout_rate=0.0)
lm = eng.lm
h7 = torch.randn(B * 200, 128, device='cuda').abs_()
rng = np.random.default_rng(0)
tp = np.zeros((B, 64), dtype=np.int32); tp[:, :32] = rng.integers(1, 1535, (B, 32))
eng.am.set_targets(np.full(B, 125), tp, np.full(B, 32))


def fwd_bwd():
    lm.forward(h7)
    lm.loss_and_decode(eng.am.labels, eng.am.label_len, eng.am.seq_len)
    lm.backward()


def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3


print('eager  language half fwd+loss+bwd: %.3f ms' % timeit(fwd_bwd), flush=True)
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    fwd_bwd(); fwd_bwd()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g, stream=s):
        fwd_bwd()
    torch.cuda.synchronize()
    print('graph  language half fwd+loss+bwd: %.3f ms' % timeit(g.replay), flush=True)
except Exception as e:
    print('capture failed:', repr(e)[:300])

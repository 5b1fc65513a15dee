"""Experiment: one DFCNN engine at B=32 against two engines at B=16 whose steps run concurrently on two streams (the
HBM-bound kernels of one half overlap the MFMA-bound ones of the other).  Timing only (the two halves have their own
weights here)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from asr_dfcnn_transformer_amd.engine import DFCNNEngine

variant = os.environ.get('VARIANT', 'm1')
T, F, V = 1600, 200, 1536


def mk(B):
    e = DFCNNEngine(model=variant, vocab=V, B=B, T=T, F=F, seed=0, device='cuda')
    x = torch.randn(B, T, F, device='cuda')
    rng = np.random.default_rng(1)
    target = np.zeros((B, 64), dtype=np.int32); target[:, :32] = rng.integers(1, V - 1, (B, 32))
    seq = np.full(B, 125, dtype=np.int32)
    return e, x, seq, target


def step(e, x, seq, target):
    e.forward(x); e.set_targets(seq, target); e.loss_and_decode(defer_decode_join=True); e.backward(); e.apply_adam(1.0)


def bench(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3


one = mk(32)
print('one engine  B=32: %.3f ms/step' % bench(lambda: step(*one)), flush=True)
del one
a, b = mk(16), mk(16)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()


def both():
    with torch.cuda.stream(sa): step(*a)
    with torch.cuda.stream(sb): step(*b)


print('two engines B=16 on two streams: %.3f ms per 32 utterances' % bench(both), flush=True)
print('two engines B=16 back to back  : %.3f ms per 32 utterances' % bench(lambda: (step(*a), step(*b))), flush=True)

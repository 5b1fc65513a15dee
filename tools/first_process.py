"""Why is the FIRST heavy GPU process on a freshly acquired box ~15 % slower?  Times blocks of 30 plain-DFCNN steps inside ONE process
with idle pauses / re-allocations in between; run it as the first GPU process of a gpurun call, then once more as the second."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from asr_dfcnn_transformer_amd.engine import DFCNNEngine
B, T, F, V = 32, 1600, 200, 1536


def make():
    eng = DFCNNEngine(model='m1', vocab=V, B=B, T=T, F=F)
    x = torch.randn(B, T, F, device='cuda')
    target = np.zeros((B, 64), dtype=np.int32); target[:, :32] = np.random.default_rng(0).integers(1, V - 1, (B, 32))
    seq = np.full(B, 125, dtype=np.int32)
    def step():
        eng.forward(x); eng.set_targets(seq, target); eng.loss_and_decode(defer_decode_join=True); eng.backward(); eng.apply_adam()
    return eng, step


def block(step, n=30):
    for _ in range(3): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): step()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


eng, step = make()
print('block 1 (fresh process)            %.3f ms/step' % block(step), flush=True)
print('block 2 (right after)              %.3f' % block(step), flush=True)
time.sleep(5)
print('block 3 (after 5 s idle)           %.3f' % block(step), flush=True)
del eng, step
torch.cuda.empty_cache()
eng, step = make()
print('block 4 (engine rebuilt, memory re-allocated)  %.3f' % block(step), flush=True)
big = torch.empty(60 << 30, dtype=torch.uint8, device='cuda'); big.zero_(); torch.cuda.synchronize(); del big; torch.cuda.empty_cache()
del eng, step
torch.cuda.empty_cache()
eng, step = make()
print('block 5 (after touching and releasing 60 GB, rebuilt)  %.3f' % block(step), flush=True)

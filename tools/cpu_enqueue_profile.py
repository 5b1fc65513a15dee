"""cProfile of the host side of DFCNN training steps (enqueue only, the GPU drains behind): where the ~6 ms of Python per step go."""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from asr_dfcnn_transformer_amd.engine import DFCNNEngine
B, T, F, V = 32, 1600, 200, 1536
eng = DFCNNEngine(model=os.environ.get('MODEL', 'm1'), vocab=V, B=B, T=T, F=F)
x = torch.randn(B, T, F, device='cuda')
target = np.zeros((B, 64), dtype=np.int32); target[:, :32] = np.random.default_rng(0).integers(1, V - 1, (B, 32))
seq = np.full(B, 125, dtype=np.int32)
def step():
    eng.forward(x); eng.set_targets(seq, target); eng.loss_and_decode(defer_decode_join=True); eng.backward(); eng.apply_adam()
for _ in range(3): step()
torch.cuda.synchronize()
# (a) small T: the GPU is never the limit -> pure host cost per step
n = 20
t0 = time.perf_counter()
for _ in range(n): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print('host enqueue %.2f ms/step, GPU complete %.2f ms/step' % (1e3 * (t1 - t0) / n, 1e3 * (t2 - t0) / n))
pr = cProfile.Profile()
pr.enable()
for _ in range(n): step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats('cumulative').print_stats(45)
st.sort_stats('tottime').print_stats(30)

import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np, torch
from oracle import transformer as otr
import test_fullsize_transformer_gpu as t
from torch_transformer_ref import e2e_ids_grads
from asr_dfcnn_transformer_amd.transformer_engine import E2EEngine
P = t._retie(t.f32(otr.init_e2e(None, t.VOUT, t.C, t.H, t.BLK, 600, seed=4, perturb=True, tie=True, vin=t.VIN)), t.BLK)
x, y_in, y = t._e2e_batch(2, seed=9)
x, y_in, y = x[1:2], y_in[1:2], y[1:2]
torch.set_num_threads(16)
g64, l64, _ = e2e_ids_grads(P, x, y_in, y, t.H, t.BLK, torch.float64)
g32, l32, _ = e2e_ids_grads(P, x, y_in, y, t.H, t.BLK, torch.float32)
eng = E2EEngine(vin=t.VIN, vout=t.VOUT, N=1, T=t.T3, L=t.T3, C=t.C, heads=t.H, blocks=t.BLK, pos_max=600, tie=True)
eng.load_params(eng.flat_from_oracle(P))
eng.forward(x, y_in, y); eng.backward(); torch.cuda.synchronize()
G = eng.grads_dict()
w = [0, 0, 0]
for k in sorted(g64):
    a = t.rel(G[k], g64[k]); b = t.rel(G[k], g32[k]); c = t.rel(g32[k], g64[k])
    w = [max(w[0], a), max(w[1], b), max(w[2], c)]
    if a > 1e-4 or b > 1e-4:
        print('%-14s gpu-f64 %.2e  gpu-torchf32 %.2e  torchf32-f64 %.2e' % (k, a, b, c))
print('worst gpu-f64 %.2e gpu-torchf32 %.2e torchf32-f64 %.2e' % tuple(w))

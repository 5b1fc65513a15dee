"""Same-process A/B of the Transformer step with / without the ReLU backward fused into the feed-forward data-gradient's epilogue
(asr_tap_gemm_relu_bwd against asr_tap_gemm + asr_relu_bwd): alternating blocks of steps on one engine."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from asr_dfcnn_transformer_amd import ops
from asr_dfcnn_transformer_amd.transformer_engine import E2EEngine
N, T, C, H, blocks, Vin, Vout = 64, 512, 512, 8, 6, 1536, 6347
eng = E2EEngine(vin=Vin, vout=Vout, N=N, T=T, L=T, C=C, heads=H, blocks=blocks, pos_max=600, tie=True, dropout_rate=0.2, drop_seed=0)
rng = np.random.default_rng(7)
x = rng.integers(1, Vin, (N, T)); y = rng.integers(3, Vout, (N, T))
y_in = np.concatenate([np.ones((N, 1), dtype=np.int64), y[:, :-1]], axis=1)
fused = ops.tap_gemm_relu_bwd
def two_calls(desc, dY, W, Hh, dX):
    ops.tap_gemm(desc, dY, W, None, None, None, None, dX)
    ops.relu_bwd(dX, Hh, dX)
def step():
    eng.forward(x, y_in, y); eng.backward(); eng.apply_adam(1.0)
for _ in range(3): step()
for rnd in range(3):
    for name, fn in (('two calls', two_calls), ('fused', fused)):
        ops.tap_gemm_relu_bwd = fn
        step(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(8): step()
        torch.cuda.synchronize()
        print('%-10s %.3f ms/step' % (name, 1e3 * (time.perf_counter() - t0) / 8), flush=True)

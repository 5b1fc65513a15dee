"""Timing bound for a second backward stream in the Transformer engine: every weight / bias gradient (tap_wgrad, its scatter copies, colsum) is
enqueued on a side stream behind an event of the main stream -- WITHOUT protecting the shared scratch buffers it reads (so the gradients of
this run are not to be trusted); what it shows is how much of the HBM-bound backward work (LayerNorm backward, ReLU backward, reductions) would
hide under the weight-gradient GEMMs.  usage (GPU box): python3 tools/exp_transformer_side.py [--no-side]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from asr_dfcnn_transformer_amd import ops
from asr_dfcnn_transformer_amd.transformer_engine import E2EEngine, _Base

use_side = '--no-side' not in sys.argv
N, T, C, H, blocks, Vin, Vout = 64, 512, 512, 8, 6, 1536, 6347
eng = E2EEngine(vin=Vin, vout=Vout, N=N, T=T, L=T, C=C, heads=H, blocks=blocks, pos_max=600, tie=True, dropout_rate=0.2, drop_seed=0)
rng = np.random.default_rng(7)
x = rng.integers(1, Vin, (N, T)); y = rng.integers(3, Vout, (N, T))
y_in = np.concatenate([np.ones((N, 1), dtype=np.int64), y[:, :-1]], axis=1)
side = torch.cuda.Stream()
ws_side = torch.zeros_like(eng.ws)
wtmp_side = torch.zeros_like(eng._wtmp)
btmp_side = torch.zeros_like(eng._btmp)

if use_side:
    def on_side(fn):
        def wrapped(self, *a, **k):
            ev = torch.cuda.Event(); ev.record()
            side.wait_event(ev)
            ws, wt, bt = self.ws, self._wtmp, self._btmp
            self.ws, self._wtmp, self._btmp = ws_side, wtmp_side, btmp_side
            try:
                with torch.cuda.stream(side):
                    return fn(self, *a, **k)
            finally:
                self.ws, self._wtmp, self._btmp = ws, wt, bt
        return wrapped
    for name in ('_wgrad', '_wgrad_packed', '_bgrad'):
        setattr(_Base, name, on_side(getattr(_Base, name)))

def step():
    eng.forward(x, y_in, y)
    eng.backward()
    if use_side:
        ev = torch.cuda.Event()
        with torch.cuda.stream(side):
            ev.record()
        torch.cuda.current_stream().wait_event(ev)
    eng.apply_adam(1.0)

for _ in range(3): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 10
for _ in range(n): step()
torch.cuda.synchronize()
print('%s: %.3f ms/step' % ('weight / bias gradients on a second stream (unsafe timing probe)' if use_side else 'one stream', 1e3 * (time.perf_counter() - t0) / n))

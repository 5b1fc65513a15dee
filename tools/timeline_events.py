#!/usr/bin/env python3
"""Timeline of ONE real (two-stream, prefetching) training step from HIP events: every library call is bracketed by a pair of
events on the stream it is launched on; timestamps are relative to the step's first call.  rocprofv3 serialises dispatches, so a
kernel trace cannot show how the main stream, the second backward stream and the feature-prefetch stream overlap -- this does
(at the price of two events per launch: the instrumented step is a few per cent slower than the timed one).
usage (on the GPU box): python3 tools/timeline_events.py [--workload dfcnn|se_dfcnn] [engine A/B flags as for bench.py]"""
import argparse, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from asr_dfcnn_transformer_amd import _lib, ops
from asr_dfcnn_transformer_amd.engine import DFCNNEngine
from asr_dfcnn_transformer_amd.wav_util import FbankExtractor


class Recorder:
    def __init__(self, lib):
        self.lib, self.on, self.rec = lib, False, []

    def __getattr__(self, name):
        fn = getattr(self.lib, name)
        if not name.startswith('asr_') or name in ('asr_last_kernel', 'asr_last_error', 'asr_version') or name.endswith(('_bytes', '_workspace', '_supported', '_floats', '_rows')):
            return fn

        def wrapped(*a):
            if not self.on:
                return fn(*a)
            st = torch.cuda.current_stream()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            rc = fn(*a)
            e1.record(st)
            self.rec.append((name, st.cuda_stream, e0, e1))
            return rc
        return wrapped


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--workload', default='dfcnn')
    ap.add_argument('--single-stream', action='store_true')
    ap.add_argument('--prefetch-early', action='store_true')
    args = ap.parse_args()
    dev = 'cuda'
    B, T, F, V = 32, 1600, 200, 1536
    eng = DFCNNEngine(model='m1' if args.workload == 'dfcnn' else 'm2', vocab=V, B=B, T=T, F=F, seed=0, device=dev, dual_stream=not args.single_stream)
    rec = Recorder(_lib.load())
    _lib._lib = rec
    fb = FbankExtractor(nfilt=F, device=dev)
    ns = 160000
    host = np.stack([(0.1 * np.random.default_rng(1234 + b).standard_normal(ns)).astype(np.float32) for b in range(B)])
    signal = torch.from_numpy(host).to(dev)
    nsamp = torch.full((B,), ns, dtype=torch.int32, device=dev)
    target = np.zeros((B, 64), dtype=np.int32)
    target[:, :32] = np.random.default_rng(99).integers(1, V - 1, (B, 32))
    seq = np.full(B, 125, dtype=np.int32)
    feats = [torch.empty(B, T, F, dtype=torch.float32, device=dev) for _ in range(2)]
    pf = torch.cuda.Stream(device=dev)
    ready, consumed, state = [None, None], [None, None], {'i': 0}

    def produce(slot, after=None):
        with torch.cuda.stream(pf):
            if consumed[slot] is not None:
                pf.wait_event(consumed[slot])
            if after is not None:
                pf.wait_event(after)
            fb.batch(signal, nsamp, T, out=feats[slot])
            ev = torch.cuda.Event(); ev.record()
            ready[slot] = ev

    def step():
        cur = state['i'] & 1
        if ready[cur] is None:
            produce(cur)
        torch.cuda.current_stream().wait_event(ready[cur])
        eng.forward(feats[cur])
        fwd_done = None
        if not args.prefetch_early:
            fwd_done = torch.cuda.Event(); fwd_done.record()
        produce(cur ^ 1, fwd_done)
        eng.set_targets(seq, target)
        eng.loss_and_decode(defer_decode_join=True)
        eng.backward()
        eng.apply_adam(1.0)
        ev = torch.cuda.Event(); ev.record()
        consumed[cur] = ev
        ready[cur] = None
        state['i'] += 1

    for _ in range(6):
        step()
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for _ in range(20):
        step()
    torch.cuda.synchronize()
    print('un-instrumented: %.3f ms/step' % (1e3 * (time.perf_counter() - t0) / 20))
    rec.on = True
    base = torch.cuda.Event(enable_timing=True)
    step(); rec.rec.clear()                      # instrumented warm-up
    torch.cuda.synchronize()
    base.record()
    step()
    end = torch.cuda.Event(enable_timing=True); end.record()
    torch.cuda.synchronize()
    rec.on = False
    print('instrumented step: %.3f ms' % base.elapsed_time(end))
    main_s = torch.cuda.current_stream().cuda_stream
    names = {main_s: 'main', pf.cuda_stream: 'feat'}
    if eng.side is not None:
        names[eng.side.cuda_stream] = 'side'
    rows = []
    for name, st, e0, e1 in rec.rec:
        rows.append((base.elapsed_time(e0) * 1e3, base.elapsed_time(e1) * 1e3, names.get(st, hex(st)), name))
    rows.sort()
    print('%10s %10s %9s  %-5s %s' % ('reached', 'done', 'span', 'strm', 'call'))
    for a, b, s, n in rows:
        print('%10.1f %10.1f %9.1f  %-5s %s' % (a, b, b - a, s, n))
    for s in sorted(set(r[2] for r in rows)):
        print('stream %s: sum of spans %.1f us, last done %.1f us' % (s, sum(r[1] - r[0] for r in rows if r[2] == s), max(r[1] for r in rows if r[2] == s)))


if __name__ == '__main__':
    main()

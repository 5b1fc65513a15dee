"""Experiment: one-utterance inference (fbank is left out: features resident) -- DFCNN forward + greedy decode enqueued eagerly
(about sixty ctypes launches) against the same launches replayed from a captured hipGraph (torch.cuda.CUDAGraph around the
ctypes calls on the capture stream; single-stream engine so that nothing inside forks to the side stream)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from asr_dfcnn_transformer_amd import ops
from asr_dfcnn_transformer_amd.engine import DFCNNEngine

V, F = 1536, 200
for variant in ('m1', 'm2'):
    for B, T in ((1, 1600), (1, 1000), (4, 1600), (32, 1600)):
        eng = DFCNNEngine(model=variant, vocab=V, B=B, T=T, F=F, seed=0, dual_stream=False)
        x = torch.randn(B, T, F, device='cuda')
        eng.set_targets(np.full(B, min(200, T // 8)), np.ones((B, 4), dtype=np.int32))

        def run():
            eng.forward(x)
            ops.ctc_greedy(eng.logits, eng.T8, B, V, eng.seq_len, V - 1, eng.dec_ids, eng.dec_len, eng.neg_sum, eng.dec_ws)

        def timeit(fn, n=200):
            for _ in range(5): fn()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(n): fn()
            torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

        t_eager = timeit(run)
        ids_eager = eng.dec_ids.clone(); logits_eager = eng.logits.clone()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            run(); run()
        torch.cuda.current_stream().wait_stream(s)
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=s):
            run()
        eng.logits.zero_(); eng.dec_ids.zero_()
        t_graph = timeit(gr.replay)
        same = bool(torch.equal(eng.logits, logits_eager) and torch.equal(eng.dec_ids, ids_eager))
        print('%s B %2d T_pad %4d: eager %.3f ms, graph replay %.3f ms per batch (same bits: %s)' % (variant, B, T, t_eager, t_graph, same), flush=True)
        del eng, gr
        torch.cuda.empty_cache()

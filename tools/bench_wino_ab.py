"""Timing of the ten Winograd forward / data-gradient launches of the plain DFCNN step (acoustic_model.py, B = 32, T_pad 1600)
exactly as the engine issues them -- forward with the fused 2x2 pool where the cell is pooled, data-gradients with the fused
backward prologue (asr_tap_gemm_gated) where the engine fuses one.  (Round 4 used it with a development switch to time
wino11_kernel against round 3's wino9 / wino10 in one process: 4575 -> 3778 us for the ten launches, profiles/r04_wino11_ab.txt.)
Round 5: the max-pooled cells in the COMPACT form the engine uses (asr_tap_gemm_wino_poolmax / asr_tap_gemm_gated_poolmax), a checksum
per launch so that two builds can be compared bit for bit, LIB=<other build of libasrhip.so> for A/B runs on one box.
usage: python tools/bench_wino_ab.py     env B, TPAD, LIB, ONLY=<substring of a layer name>"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from asr_dfcnn_transformer_amd import _lib
if os.environ.get('LIB'):
    _lib.LIB_PATH = os.path.abspath(os.environ['LIB'])
import torch
from asr_dfcnn_transformer_amd import ops
from asr_dfcnn_transformer_amd.ops import Plane
ONLY = os.environ.get('ONLY', '')


def checksum(*ts):
    return '%016x' % (sum(int(t.view(torch.int32).to(torch.int64).sum().item()) * (i + 1) for i, t in enumerate(ts)) & 0xFFFFFFFFFFFFFFFF)

B = int(os.environ.get('B', 32))
TP = int(os.environ.get('TPAD', 1600))
lib = _lib.load()
setgen = lambda gen: None
gens = [11]


def timeit(fn, iters=20):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


g = torch.Generator(device='cuda').manual_seed(0)
rnd = lambda *s: torch.randn(*s, device='cuda', generator=g)
H8, W8 = TP // 8, 25
# (name, H, W, cin, cout, pool of the cell: 0 none / 2 max)
FWD = [('c2 32->64', TP // 2, 100, 32, 64, 2), ('c3 64->128', TP // 4, 50, 64, 128, 2), ('c4 128->128', H8, W8, 128, 128, 0),
       ('c5a 128->256', H8, W8, 128, 256, 0), ('c5 32->256', H8, W8, 32, 256, 0)]
# data-gradients: (name, H, W, K = cout of the cell, N = cin, gate: None or (pool of the cell in front, its channel count = N))
BWD = [('h5 256->32', H8, W8, 256, 32, (0,)), ('h5a 256->128', H8, W8, 256, 128, (0,)), ('h4 128->128', H8, W8, 128, 128, (2,)),
       ('h3 128->64', TP // 4, 50, 128, 64, (2,)), ('h2 64->32', TP // 2, 100, 64, 32, None)]
tot = {gen: 0.0 for gen in gens}
print('B = %d, T_pad = %d; us per launch (TFLOP/s in direct-conv flops)' % (B, TP))
for name, H, W, cin, cout, pool in FWD:
    if ONLY not in name:
        continue
    x = Plane(B, H, W, cin); x.set_interior(rnd(B, H, W, cin))
    w = rnd(3, 3, cin, cout) * (2.0 / (9 * cin)) ** 0.5
    bias = rnd(cout) * 0.1; sc = 1 + 0.2 * rnd(cout); sh = 0.1 * rnd(cout)
    a = Plane(B, H, W, cout)
    y = Plane(B, H // 2, W // 2, cout) if pool else Plane(B, H, W, cout)
    d = ops.gemm_desc(x.NP, cin, cout, cin, cout, cout, 0 if pool else cout, ntaps=9, B=B, H=H, W=W, relu=1)
    wt = ops.winograd_weights(w, cin, cout, cout, 0)
    fl = 2.0 * B * H * W * 9 * cin * cout
    row, ref = [], None
    for gen in gens:
        setgen(gen)
        if pool:
            amax, idx = Plane(B, H // 2, W // 2, cout), ops.poolmax_index(B, H // 2, W // 2, cout)
            fn = lambda: ops.tap_gemm_wino_poolmax(d, x, wt, bias, sc, sh, y, amax, idx)
        else:
            fn = lambda: ops.tap_gemm_wino(d, x, wt, bias, sc, sh, a, y)
        t = timeit(fn)
        tot[gen] += t
        cur = (amax.buf.clone(), y.buf.clone()) if pool else (a.buf.clone(), y.buf.clone())
        diff = 0.0 if ref is None else max((cur[0] - ref[0]).abs().max().item(), (cur[1] - ref[1]).abs().max().item())
        ref = ref or cur
        row.append('%-20s %7.1f us (%5.1f)  sum %s' % (ops.last_kernel()[:20], t, fl / t / 1e6, checksum(*cur)))
    print('fwd   %-14s %dx%d  ' % (name, H, W) + ' | '.join(row), flush=True)
    del x, a, y
for name, H, W, K, N, gate in BWD:
    if ONLY not in name:
        continue
    dz = Plane(B, H, W, K); dz.set_interior(rnd(B, H, W, K))
    w = rnd(3, 3, N, K) * 0.05
    bd = ops.gemm_desc(dz.NP, K, N, K, K, 0, N, ntaps=9, B=B, H=H, W=W, wmode=1)
    wt = ops.winograd_weights(w, K, N, K, 1)
    fl = 2.0 * B * H * W * 9 * K * N
    row, ref = [], None
    if gate is None:
        dx = Plane(B, H, W, N)
        for gen in gens:
            setgen(gen)
            t = timeit(lambda: ops.tap_gemm_wino(bd, dz, wt, None, None, None, None, dx))
            tot[gen] += t
            cur = dx.buf.clone()
            diff = 0.0 if ref is None else (cur - ref).abs().max().item()
            ref = cur if ref is None else ref
            row.append('%-20s %7.1f us (%5.1f)  sum %s' % (ops.last_kernel()[:20], t, fl / t / 1e6, checksum(cur)))
    else:
        pool = gate[0]
        gh, gw = (H, W) if pool == 0 else (2 * H, 2 * W)
        act = Plane(B, gh, gw, N); act.set_interior(torch.relu(rnd(B, gh, gw, N)))
        sc = 1 + 0.2 * rnd(N); sh = 0.1 * rnd(N)
        dzo = Plane(B, gh, gw, N)
        sums = [torch.zeros(N, device='cuda') for _ in range(3)]
        ws = torch.zeros(ops.tap_gemm_gated_workspace(bd) // 4 + 64, device='cuda')
        if pool == 2:
            amax = Plane(B, H, W, N); amax.set_interior(torch.relu(rnd(B, H, W, N)))
            idx = ops.poolmax_index(B, H, W, N)
            idx.copy_(torch.randint(0, 2 ** 31 - 1, idx.shape, device='cuda', generator=g, dtype=torch.int64).to(idx.dtype))
        for gen in gens:
            setgen(gen)
            if pool == 2:
                t = timeit(lambda: ops.tap_gemm_gated_poolmax(bd, dz, wt, gh, gw, amax, idx, sc, sh, None, dzo, sums[0], sums[1], sums[2], ws))
            else:
                t = timeit(lambda: ops.tap_gemm_gated(bd, dz, wt, 2, pool, act, sc, sh, None, dzo, sums[0], sums[1], sums[2], ws))
            tot[gen] += t
            cur = (dzo.buf.clone(), torch.stack(sums).clone())
            diff = 0.0 if ref is None else max((cur[0] - ref[0]).abs().max().item(), ((cur[1] - ref[1]).abs().max() / ref[1].abs().max()).item())
            ref = ref or cur
            row.append('%-20s %7.1f us (%5.1f)  sum %s' % (ops.last_kernel()[:20], t, fl / t / 1e6, checksum(*cur)))
    print('dgrad %-14s %dx%d  ' % (name, H, W) + ' | '.join(row), flush=True)
print('sum of the launches: %.1f us' % tot[gens[0]])

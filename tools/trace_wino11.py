"""In-kernel phase stamps of wino11_kernel (development; wino.hip built with -DW11_TRACE): one workgroup writes s_memtime (shader
clock) at the phase boundaries of its items 1..8.  Build + run on the GPU box:
  bash tools/trace_wino11.sh
Stamps: 0 item start, 9 / 10 after the first / second chunk (+ barrier), 1 chunks done, 2 phase-0 column sums written, 3 exchange
barrier passed, 4 both phases done, 5 next item's DMA landed, 6 tables / patch offsets of the next tile block written, 7 pool done,
8 epilogue issued, 11 end-of-item barrier passed."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from asr_dfcnn_transformer_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), os.environ.get('TRACE_LIB', 'libasrhip_trace.so'))
import torch
from asr_dfcnn_transformer_amd import ops
from asr_dfcnn_transformer_amd.ops import Plane
lib = _lib.load()
B = 32
ORDER = [0, 9, 10, 1, 2, 3, 4, 5, 6, 7, 8, 11]
g = torch.Generator(device='cuda').manual_seed(0)
rnd = lambda *s: torch.randn(*s, device='cuda', generator=g)


def dump(title):
    torch.cuda.synchronize()
    buf = np.zeros(8 * 8 * 16, dtype=np.int64)
    assert lib.asr_w11_trace_dump(buf.ctypes.data_as(C.c_void_p)) == 0
    t = buf.reshape(8, 8, 16)
    print(title + ': cycles (shader clock) since the item start; stamps ' + ' '.join('%5d' % k for k in ORDER[1:]))
    r0, r1 = t[1, 0], t[6, 0]
    if r1[12] > r0[12]:
        print('  in-kernel clock (items 1..6, wave 0): %.0f MHz' % ((r1[0] - r0[0]) / (r1[12] - r0[12]) * 100.0))
    for item in (2, 3, 4):
        for wave in (0, 3, 7):
            r = t[item, wave]
            print('  item %d wave %d: ' % (item, wave) + ' '.join('%6d' % (r[k] - r[0]) for k in ORDER[1:]) + '   | next item starts at %6d' % (t[item + 1, wave, 0] - r[0]))


for name, H, W, cin, cout, pool in (('fwd c2 32->64 pooled', 800, 100, 32, 64, 2), ('fwd c3 64->128 pooled', 400, 50, 64, 128, 2), ('fwd c5a 128->256', 200, 25, 128, 256, 0)):
    x = Plane(B, H, W, cin); x.set_interior(rnd(B, H, W, cin))
    w = rnd(3, 3, cin, cout) * 0.05
    bias = rnd(cout) * 0.1; sc = 1 + 0.2 * rnd(cout); sh = 0.1 * rnd(cout)
    a = Plane(B, H, W, cout)
    y = Plane(B, H // 2, W // 2, cout) if pool else Plane(B, H, W, cout)
    d = ops.gemm_desc(x.NP, cin, cout, cin, cout, cout, 0 if pool else cout, ntaps=9, B=B, H=H, W=W, relu=1)
    wt = ops.winograd_weights(w, cin, cout, cout, 0)
    if pool:
        amax, idx = Plane(B, H // 2, W // 2, cout), ops.poolmax_index(B, H // 2, W // 2, cout)
    for _ in range(3):
        if pool: ops.tap_gemm_wino_poolmax(d, x, wt, bias, sc, sh, y, amax, idx)       # the compact form the engine uses
        else: ops.tap_gemm_wino(d, x, wt, bias, sc, sh, a, y)
    dump(name)
    del x, a, y
for name, H, W, K, N, pool in (('dgrad h3 128->64 gated compact max-pool', 400, 50, 128, 64, 2), ('dgrad h4 128->128 gated compact max-pool', 200, 25, 128, 128, 2), ('dgrad h2 64->32 plain', 800, 100, 64, 32, None)):
    dz = Plane(B, H, W, K); dz.set_interior(rnd(B, H, W, K))
    w = rnd(3, 3, N, K) * 0.05
    bd = ops.gemm_desc(dz.NP, K, N, K, K, 0, N, ntaps=9, B=B, H=H, W=W, wmode=1)
    wt = ops.winograd_weights(w, K, N, K, 1)
    if pool is None:
        dx = Plane(B, H, W, N)
        for _ in range(3): ops.tap_gemm_wino(bd, dz, wt, None, None, None, None, dx)
    else:
        gh, gw = 2 * H, 2 * W
        act = Plane(B, gh, gw, N); act.set_interior(torch.relu(rnd(B, gh, gw, N)))
        sc = 1 + 0.2 * rnd(N); sh = 0.1 * rnd(N)
        dzo = Plane(B, gh, gw, N)
        sums = [torch.zeros(N, device='cuda') for _ in range(3)]
        ws = torch.zeros(ops.tap_gemm_gated_workspace(bd) // 4 + 64, device='cuda')
        amax = Plane(B, H, W, N); amax.set_interior(torch.relu(rnd(B, H, W, N)))
        idx = ops.poolmax_index(B, H, W, N)
        idx.copy_(torch.randint(0, 2 ** 31 - 1, idx.shape, device='cuda', generator=g, dtype=torch.int64).to(idx.dtype))
        for _ in range(3): ops.tap_gemm_gated_poolmax(bd, dz, wt, gh, gw, amax, idx, sc, sh, None, dzo, sums[0], sums[1], sums[2], ws)
    dump(name)

"""The dense GEMMs of the secondary workloads on gemm1_kernel with the tile the library picks for each grid (128 x 128 or 128 x 64:
gemm1_blocks(), csrc/gemm1.hip).  Round 4 timed both tiles on every shape through a development switch: profiles/r04_gemm1_tiles.txt.
usage: python tools/bench_gemm1_tiles.py"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from asr_dfcnn_transformer_amd import ops, _lib
lib = _lib.load()
def timeit(fn, iters=30):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
g = torch.Generator(device='cuda').manual_seed(0)
SHAPES = [(6400, 512, 1536), (6400, 512, 512), (6400, 512, 2048), (6400, 2048, 512), (6400, 512, 6348), (6400, 1536, 512), (6400, 6348, 512),
          (6400, 256, 1424), (6400, 1424, 256), (6400, 3200, 256), (6400, 256, 3200), (12800, 512, 512), (3200, 512, 1536), (3200, 512, 512),
          (32768, 512, 512), (32768, 512, 1536), (32768, 512, 2048), (32768, 2048, 512)]
for (M, cin, cout) in SHAPES:
    x = torch.randn(M, cin, device='cuda', generator=g); w = torch.randn(cin, cout, device='cuda', generator=g) * 0.02
    y = torch.zeros(M, cout, device='cuda'); b = torch.zeros(cout, device='cuda')
    fd = ops.gemm_desc(M, cin, cout, cin, cout, cout, 0, ntaps=1, relu=1)
    wT = w.t().contiguous()
    fl = 2.0 * M * cin * cout
    t = timeit(lambda: ops.tap_gemm_nt(fd, x, w, wT, cin, b, None, None, y, None))
    print('M %5d K %4d N %4d | %-20s %7.1f us %6.1f TF' % (M, cin, cout, ops.last_kernel(), 1e3 * t, fl / t / 1e9), flush=True)

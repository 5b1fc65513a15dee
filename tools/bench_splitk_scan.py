"""asr_tap_gemm_nt_splitk over the split counts a layer admits (the library picks the tile from the grid's fill): the two narrow head layers."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from asr_dfcnn_transformer_amd import ops
def timeit(fn, iters=30):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
g = torch.Generator(device='cuda').manual_seed(0)
for (M, K, N) in [(6400, 6400, 128), (6400, 1536, 128), (4000, 6400, 128)]:
    x = torch.randn(M, K, device='cuda', generator=g); w = torch.randn(K, N, device='cuda', generator=g) * 0.02
    wT = w.t().contiguous(); b = torch.zeros(N, device='cuda'); y = torch.zeros(M, N, device='cuda')
    d = ops.gemm_desc(M, K, N, K, N, N, N, ntaps=1, relu=1)
    ws = torch.zeros(ops.tap_gemm_nt_splitk_workspace(d, 16) // 4 + 4, device='cuda')
    for s in range(2, 17):
        if K % (32 * s): continue
        t = timeit(lambda: ops.tap_gemm_nt_splitk(d, x, wT, K, b, None, None, None, y, s, ws))
        print('M %d %d -> %d  x%-2d %s  %.1f us' % (M, K, N, s, ops.last_kernel(), 1e3 * t))

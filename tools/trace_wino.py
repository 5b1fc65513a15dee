"""In-kernel phase stamps of the eight-wave Winograd kernel (wino.hip, WINO_TRACE): workgroup 0 writes s_memrealtime
(100 MHz) at the phase boundaries of its first items.  Needs a trace build of the library next to this file:
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DWINO_TRACE -c asr_dfcnn_transformer_amd/csrc/wino.hip -o /tmp/wino_trace.o
  hipcc --offload-arch=gfx950 -shared -fPIC -o tools/libasrhip_trace.so $(ls asr_dfcnn_transformer_amd/build/*.o | grep -v wino.hip.o) /tmp/wino_trace.o
Stamps: 0 item start, 1 row tables written, 2 chunk 0 landed, 3 chunks done, 4 =3, 5 column sums written, 6 exchange barrier,
7 partner read, 8 barrier, 9 epilogue stores issued, 10 barrier, 11.. end of chunks 0..4."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from asr_dfcnn_transformer_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libasrhip_trace.so')
import torch
from asr_dfcnn_transformer_amd import ops
from asr_dfcnn_transformer_amd.ops import Plane
lib = _lib.load()
B, H, W = 32, 400, 50
for cin, cout, relu in ((64, 128, 1), (32, 64, 1)):
    x = Plane(B, H, W, cin); x.interior().normal_()
    w = torch.randn(3, 3, cin, cout, device='cuda') * 0.05
    a1 = Plane(B, H, W, cout); y1 = Plane(B, H, W, cout)
    bias = torch.zeros(cout, device='cuda'); sc = torch.ones(cout, device='cuda'); sh = torch.zeros(cout, device='cuda')
    d = ops.gemm_desc(x.NP, cin, cout, cin, cout, cout, cout if relu else 0, ntaps=9, B=B, H=H, W=W, relu=relu)
    wt = ops.winograd_weights(w, cin, cout, cout, 0)
    for _ in range(3):
        if relu: ops.tap_gemm_wino(d, x, wt, bias, sc, sh, a1, y1)
        else: ops.tap_gemm_wino(d, x, wt, None, None, None, a1, None)
    torch.cuda.synchronize()
    buf = np.zeros(4 * 8 * 16, dtype=np.int64)
    assert lib.asr_wino_trace_dump(buf.ctypes.data_as(C.c_void_p)) == 0
    t = buf.reshape(4, 8, 16)
    print('cin %d cout %d relu %d: stamps in us relative to item start (100 MHz counter), items 1-2, waves 0 and 4' % (cin, cout, relu))
    for item in (1, 2):
        for wave in (0, 4, 7):
            r = t[item, wave, :16]
            print('  item %d wave %d: ' % (item, wave) + ' '.join('%6.2f' % ((v - r[0]) / 100.0) for v in r))

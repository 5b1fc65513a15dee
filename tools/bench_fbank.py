"""Times asr_fbank (log-mel + standardisation) on the bench batch: 32 x 10 s of 16 kHz audio, 200 mel bins, T_pad 1600."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from asr_dfcnn_transformer_amd import _lib
if os.environ.get('LIB'):
    _lib.LIB_PATH = os.path.abspath(os.environ['LIB'])
import numpy as np, torch
from asr_dfcnn_transformer_amd.wav_util import FbankExtractor
B, ns, T = 32, 160000, 1600
sig = torch.from_numpy(np.stack([(0.1 * np.random.default_rng(1234 + b).standard_normal(ns)).astype(np.float32) for b in range(B)])).cuda()
n = torch.full((B,), ns, dtype=torch.int32, device='cuda')
for nfilt in (200, 80):
    fb = FbankExtractor(nfilt=nfilt)
    out = torch.empty(B, T, nfilt, device='cuda')
    for _ in range(3): fb.batch(sig, n, T, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fb.batch(sig, n, T, out=out)
    e1.record(); torch.cuda.synchronize()
    print('nfilt %3d: %.1f us per batch (log-mel + scale), checksum %.6f' % (nfilt, 1e3 * e0.elapsed_time(e1) / 20, float(out.double().abs().sum())))

"""CPU checks of the drop-in boundary: the C-ABI library loads and exports every symbol
that include/asr_hip.h declares, and the host-side tables match the oracle.  No compute
call is made here (there is no GPU in this tier)."""
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, 'include', 'asr_hip.h')).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    return sorted(set(re.findall(r'\b(asr_[a-z0-9_]+)\s*\(', txt)))


def test_library_exports_every_declared_symbol():
    from asr_dfcnn_transformer_amd import _lib
    lib = _lib.load()
    names = _declared()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), 'libasrhip.so does not export %s' % n
        assert n in _lib.SIGNATURES, 'ctypes signature missing for %s' % n
    assert lib.asr_version() >= 100


def test_banded_filterbank_matches_oracle():
    from asr_dfcnn_transformer_amd import wav_util
    from oracle import fbank as ofb
    for nfilt in (200, 80, 26):
        st, cnt, w, width = wav_util.mel_filterbank_banded(nfilt, 512, 16000)
        fb = ofb.get_filterbanks(nfilt, 512, 16000)
        dense = np.zeros_like(fb)
        for j in range(nfilt):
            dense[j, st[j]:st[j] + cnt[j]] = w[j, :cnt[j]]
        assert np.array_equal(dense, fb)
    assert wav_util.num_frames(160000) == ofb.num_frames(160000) == 999

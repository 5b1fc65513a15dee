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


def test_host_utils_match_oracle():
    from asr_dfcnn_transformer_amd import utils
    from oracle import fbank as ofb, ctc as octc
    x = np.arange(33, dtype=np.float64).reshape(11, 3)
    assert np.array_equal(utils.build_LFR_features(x, 4, 3), ofb.build_LFR_features(x, 4, 3))
    assert utils.GetEditDistance('abcd', 'abxyd') == octc.get_edit_distance_difflib('abcd', 'abxyd') == 2
    ind, val, shp = utils.sparse_tuple_from([[1, 2], [], [3]])
    assert ind.tolist() == [[0, 0], [0, 1], [2, 0]] and val.tolist() == [1, 2, 3] and shp.tolist() == [3, 2]


def test_hparams_surface():
    from asr_dfcnn_transformer_amd.hparams import AmLmHparams, AmDataHparams
    a = AmLmHparams().args
    assert (a.am_lr, a.dacay_step, a.min_learning_rate, a.am_batch_size, a.feature_dim, a.feature_max_length) == \
        (0.0007, 5000, 1e-6, 16, 200, 1600)
    assert AmDataHparams.args.pinyin_dict == 'mixdict.txt' and AmDataHparams.args.lfr_m == 4

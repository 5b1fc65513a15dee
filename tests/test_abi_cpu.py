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


def test_size_helpers_and_argument_checks_without_a_gpu():
    """Host-only parts of the boundary: buffer-size helpers follow the layouts include/asr_hip.h documents, and malformed
    calls are rejected with ASR_ERR_BAD_ARG before anything is launched (so this runs on the CPU tier)."""
    import ctypes as C
    from asr_dfcnn_transformer_amd import _lib, ops
    lib = _lib.load()
    # fragment-order fp32 weights: [taps][ceil(K/8)][ceil(N/32)][64 lanes][4] floats
    assert lib.asr_arrange_weights_bytes(9, 64, 128) == 9 * 8 * 4 * 256 * 4
    assert lib.asr_arrange_weights_bytes(1, 72, 100) == 1 * 9 * 4 * 256 * 4
    # split-bf16 weights: 3 pieces, K rounded up to 32, N rounded up to 32, 2 bytes
    assert lib.asr_split_weights_bytes(9, 64, 128) == 9 * 3 * 128 * 64 * 2
    assert lib.asr_split_weights_bytes(1, 72, 100) == 3 * 128 * 96 * 2
    assert lib.asr_split_rows_bytes(1000, 72) == 3 * 1000 * 96 * 2
    d = ops.gemm_desc(32 * 201 * 26, 128, 128, 128, 128, ntaps=9, B=32, H=200, W=25)
    ws = lib.asr_tap_wgrad_workspace(C.byref(d))
    assert ws % (9 * 128 * 128 * 4) == 0 and ws >= 2 * 9 * 128 * 128 * 4          # whole partial slabs
    null = C.c_void_p(0)
    bad = lib.asr_tap_gemm_pw(C.byref(d), null, null, null, null, null, null, null, null)
    assert bad < 0
    assert lib.asr_arrange_weights(null, 9, 64, 128, 128, 0, null, null) < 0
    assert lib.asr_split_rows(null, 10, 8, 8, null, null) < 0
    assert lib.asr_gemm_bx6s(null, null, 10, 8, 8, null, 0, 0, null, 8, null, null) < 0
    assert lib.asr_tap_gemm(C.byref(d), null, null, null, null, null, null, null, null) < 0
    assert lib.asr_ctc_loss(null, 8, 2, 9, null, 64, null, null, 8, null, null, null, null, null) < 0

"""Against outputs of the reference's OWN util/utils.py functions build_LFR_features (:7-31), GetEditDistance (:43-54) and
sparse_tuple_from (:69-88), executed unmodified in the build container by tests/golden/make_reference_utils_golden.py (which explains how:
the three FunctionDef nodes of the file, compiled as they stand, with the real numpy / difflib; nothing copied, nothing stubbed) and
committed as tests/golden/reference_utils.npz.  Held to them, bit for bit: the oracle's restatement (oracle/fbank.py, oracle/ctc.py), the
package's host utilities (asr_dfcnn_transformer_amd/utils.py) and -- on the GPU -- the asr_lfr kernel.  Reads the
committed file only."""
import json
import math
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = np.load(os.path.join(ROOT, 'tests', 'golden', 'reference_utils.npz'))


def _lfr_cases():
    for i, (T, D, m, n) in enumerate(GOLD['lfr_cases'].tolist()):
        yield i, T, D, m, n, GOLD['lfr_in_%d' % i], GOLD['lfr_out_%d' % i]


def test_lfr_oracle_and_host_utility_equal_the_reference():
    from oracle import fbank as ofb
    from asr_dfcnn_transformer_amd import utils
    seen = 0
    for i, T, D, m, n, x, want in _lfr_cases():
        assert want.shape == (math.ceil(T / n), m * D) and want.dtype == np.float32
        for fn in (ofb.build_LFR_features, utils.build_LFR_features):
            got = fn(x, m, n)
            assert got.shape == want.shape and np.array_equal(np.asarray(got, dtype=np.float32), want), (fn.__module__, T, D, m, n)
        seen += 1
    assert seen >= 18


def test_edit_distance_helper_equals_the_reference():
    from oracle import ctc as octc
    from asr_dfcnn_transformer_amd import utils
    pairs = json.loads(str(GOLD['edit_pairs_json']))
    want = GOLD['edit_dist'].tolist()
    assert len(pairs) == len(want) >= 50
    for (a, b), w in zip(pairs, want):
        assert utils.GetEditDistance(a, b) == w, (a, b)
        assert octc.get_edit_distance_difflib(a, b) == w, (a, b)


def test_sparse_tuple_from_equals_the_reference():
    from asr_dfcnn_transformer_amd import utils
    for i, s in enumerate(json.loads(str(GOLD['sparse_seqs_json']))):
        ind, val, shp = utils.sparse_tuple_from(s)
        for got, key in ((ind, 'sparse_ind_%d'), (val, 'sparse_val_%d'), (shp, 'sparse_shape_%d')):
            want = GOLD[key % i]
            assert got.dtype == want.dtype and np.array_equal(got, want), (i, key)


@pytest.mark.gpu
def test_lfr_kernel_equals_the_reference():
    """asr_lfr (csrc/fbank.hip) on a padded batch that holds every golden case of one (D, m, n) family, and each case alone."""
    import torch
    from asr_dfcnn_transformer_amd import ops
    ran = 0
    for i, T, D, m, n, x, want in _lfr_cases():
        if D % 4:                                                       # the kernel's documented contract (include/asr_hip.h: D % 4 == 0)
            with pytest.raises(Exception):
                ops.lfr(torch.zeros(1, T, D, device='cuda'), torch.tensor([T], dtype=torch.int32, device='cuda'), m, n, math.ceil(T / n))
            continue
        ran += 1
        t_pad = T + 3
        feat = np.zeros((2, t_pad, D), dtype=np.float32)
        feat[0, :T] = x
        feat[1, :max(T - 1, 1)] = x[:max(T - 1, 1)]                       # a shorter neighbour in the same launch
        frames = torch.tensor([T, max(T - 1, 1)], dtype=torch.int32, device='cuda')
        t_out = math.ceil(T / n)
        got = ops.lfr(torch.tensor(feat, device='cuda'), frames, m, n, t_out).cpu().numpy()
        assert np.array_equal(got[0], want), (T, D, m, n)
        if T > 1:                                                       # the neighbour = the reference on the shorter input? only its prefix rule is checked here
            assert not got[1, math.ceil((T - 1) / n):].any()
    assert ran >= 10

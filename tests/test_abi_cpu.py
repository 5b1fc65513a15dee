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
    return sorted(set(re.findall(r'\b(asr_[A-Za-z0-9_]+)\s*\(', txt)))


def test_library_exports_every_declared_symbol():
    from asr_dfcnn_transformer_amd import _lib
    lib = _lib.load()
    names = _declared()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), 'libasrhip.so does not export %s' % n
        assert n in _lib.SIGNATURES, 'ctypes signature missing for %s' % n
    assert lib.asr_version() >= 100


def test_library_exports_nothing_the_header_does_not_declare():
    """The dynamic symbol table of libasrhip.so == the declarations of include/asr_hip.h, compared over ALL defined dynamic symbols
    (C++-mangled ones included; only the `__hip_*` registration objects hipcc emits per translation unit are set aside): the library
    is built with -fvisibility=hidden and the header's declarations sit inside a `visibility push(default)` region, so cross-file
    helpers, inline reductions and device stubs stay out."""
    import subprocess
    from asr_dfcnn_transformer_amd import _build
    out = subprocess.run(['nm', '-D', '--defined-only', _build.LIB], check=True, stdout=subprocess.PIPE).stdout.decode()
    exported = sorted(l.split()[-1] for l in out.splitlines() if l.split() and not l.split()[-1].startswith('__hip_'))
    assert exported == _declared(), (sorted(set(exported) - set(_declared())), sorted(set(_declared()) - set(exported)))


def test_integration_md_lists_every_declared_symbol():
    """INTEGRATION.md ends with a table generated from the header (tools/gen_entry_point_table.py): every entry point, the reference
    lines its header comment cites, the ops.py wrapper that binds it.  A symbol added to the header without regenerating it fails here."""
    txt = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    missing = [n for n in _declared() if '`%s`' % n not in txt]
    assert not missing, 'run python3 tools/gen_entry_point_table.py: %s' % missing


def test_winograd_weight_buffer_contract():
    """asr_winograd_weights2 fills TWO layouts (2 x 16 K N floats, asr_winograd_weights_bytes) and refuses a smaller buffer before
    anything is launched; the round-4 name, whose contract had changed silently, is gone."""
    from asr_dfcnn_transformer_amd import _lib
    lib = _lib.load()
    for K, N in ((32, 64), (128, 256), (8, 32), (24, 40)):
        assert lib.asr_winograd_weights_bytes(K, N) == 2 * 16 * K * N * 4
    assert not hasattr(lib, 'asr_winograd_weights')
    dummy = 4096                                            # never dereferenced: the size check comes first
    assert lib.asr_winograd_weights2(dummy, 32, 64, 64, 0, dummy, 16 * 32 * 64 * 4, None) == -1      # ASR_ERR_BAD_ARG


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
    assert np.array_equal(utils.build_LFR_features(x, 4, 3), ofb.build_LFR_features(x, 4, 3))   # gather vs explicit loops
    # hand-written fixture ([10, 3], (m, n) = (4, 3); util/utils.py:9-31): T_lfr = ceil(10 / 3) = 4; frame i starts at
    # input row 3 i; frame 3 starts at row 9, the last one, and is completed with three more copies of it (:25-29)
    r = lambda t: [3 * t, 3 * t + 1, 3 * t + 2]
    x10 = np.arange(30, dtype=np.float32).reshape(10, 3)
    want = np.array([r(0) + r(1) + r(2) + r(3),
                     r(3) + r(4) + r(5) + r(6),
                     r(6) + r(7) + r(8) + r(9),
                     r(9) + r(9) + r(9) + r(9)], dtype=np.float32)
    got = utils.build_LFR_features(x10, 4, 3)
    assert got.shape == (4, 12) and got.dtype == np.float32 and np.array_equal(got, want)
    # T = 8: frame 2 starts at row 6 and has rows 6, 7 left -> two copies of row 7 follow
    x8 = x10[:8]
    assert np.array_equal(utils.build_LFR_features(x8, 4, 3)[2], np.array(r(6) + r(7) + r(7) + r(7), dtype=np.float32))
    assert np.array_equal(utils.build_LFR_features(x10, 1, 1), x10)                  # m = n = 1: identity (:10)
    assert np.array_equal(utils.build_LFR_features(x10, 1, 3), x10[::3])             # m = 1: skipping (:11)
    assert np.array_equal(utils.build_LFR_features(x10, 2, 1)[:9], np.hstack([x10[:-1], x10[1:]]))   # n = 1: right-stacking (:12)
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
    d = ops.gemm_desc(32 * 201 * 26, 128, 128, 128, 128, ntaps=9, B=32, H=200, W=25)
    ws = lib.asr_tap_wgrad_workspace(C.byref(d))
    assert ws % (9 * 128 * 128 * 4) == 0 and ws >= 2 * 9 * 128 * 128 * 4          # whole partial slabs
    null = C.c_void_p(0)
    bad = lib.asr_tap_gemm_pw(C.byref(d), null, null, null, null, null, null, null, null)
    assert bad < 0
    assert lib.asr_arrange_weights(null, 9, 64, 128, 128, 0, null, null) < 0
    assert lib.asr_tap_gemm(C.byref(d), null, null, null, null, null, null, null, null) < 0
    assert lib.asr_ctc_loss(null, 8, 2, 9, null, 64, null, null, 8, null, null, null, null, null) < 0


def test_winograd_predicate_odd_heights_and_the_2_gib_bound():
    """asr_winograd_supported (csrc/wino.hip) is a pure function of the descriptor: odd plane heights are Winograd shapes since
    round 4 (T_pad 1000 -> 125 x 25 planes), and an input plane of 2 GiB or more is refused -- wino11_kernel reads it through
    buffer-form DMA with num_records 0x7FFFFFF0 and 32-bit offsets, where an out-of-range read returns zeros without an error;
    such layers stay on the direct kernels (64-bit addressing).  The 1600 x 200 x 32 plane of the full-size model crosses the
    bound between B = 52 and B = 53."""
    from asr_dfcnn_transformer_amd import ops

    def desc(B, H, W, K, N, wmode=0):
        return ops.gemm_desc(B * (H + 1) * (W + 1), K, N, K, N if wmode == 0 else K, N, N, ntaps=9, B=B, H=H, W=W, wmode=wmode)

    assert ops.winograd_supported(desc(32, 200, 25, 128, 128))
    assert ops.winograd_supported(desc(32, 125, 25, 128, 128))                     # odd height
    assert ops.winograd_supported(desc(32, 125, 25, 256, 32, wmode=1))             # ... on the 32-wide channel blocks too
    assert ops.winograd_supported(desc(2, 7, 5, 8, 64)) and ops.winograd_supported(desc(2, 1, 25, 32, 64))
    assert not ops.winograd_supported(desc(2, 7, 5, 12, 64))                       # K % 8
    assert not ops.winograd_supported(desc(2, 7, 5, 8, 48))                        # N % 32
    assert 52 * 1601 * 201 * 32 * 4 < 0x7FFFFFF0 <= 53 * 1601 * 201 * 32 * 4
    assert ops.winograd_supported(desc(52, 1600, 200, 32, 64))
    assert not ops.winograd_supported(desc(53, 1600, 200, 32, 64))                 # 2 GiB input plane
    assert 103 * 801 * 101 * 64 * 4 < 0x7FFFFFF0 <= 104 * 801 * 101 * 64 * 4
    assert not ops.winograd_supported(desc(104, 800, 100, 64, 128))                # [B][801][101][64] at B = 104
    assert ops.winograd_supported(desc(103, 800, 100, 64, 128))
    # the weight-gradient route has the same bound (ww_plan): the Winograd workspace is only counted below it
    import ctypes as C
    from asr_dfcnn_transformer_amd import _lib
    lib = _lib.load()
    small, big = desc(52, 1600, 200, 32, 64), desc(53, 1600, 200, 32, 64)
    assert lib.asr_tap_wgrad_workspace(C.byref(small)) > 0 and lib.asr_tap_wgrad_workspace(C.byref(big)) > 0
    # a gated Winograd launch writes 4 partial rows per tile block; planes of few tile rows have mostly empty blocks, and the
    # workspace helper must cover them (it used to assume the direct kernels' M / 32 rows)
    for B, H, W in ((8, 2, 25), (8, 1, 25), (4, 125, 25), (32, 200, 25)):
        d = desc(B, H, W, 64, 64, wmode=1)
        TH, TW = (H + 1) // 2, (W + 1) // 2
        blocks = B * -(-TH * TW // 64)
        assert lib.asr_tap_gemm_gated_workspace(C.byref(d)) >= 4 * blocks * 3 * 64 * 4


def test_layernorm_backward_workspace_covers_every_smaller_row_count():
    """The engines size ONE workspace for their largest row count and reuse it for smaller problems; the rows-per-block rule of
    asr_layernorm_bwd switches at 32768 rows (16 rows per block below: more blocks), so the size must never shrink as rows grow."""
    from asr_dfcnn_transformer_amd import _lib
    lib = _lib.load()
    sizes = [lib.asr_layernorm_bwd_workspace(r, 512) for r in (1, 100, 6400, 20000, 32767, 32768, 40000, 65536, 200000)]
    assert all(a <= b for a, b in zip(sizes, sizes[1:])), sizes
    assert sizes[4] >= 2048 * 2 * 512 * 4                     # 32767 rows = 2048 blocks of [2][512] partials

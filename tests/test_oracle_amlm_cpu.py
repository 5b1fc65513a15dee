"""CPU checks of the joint AM+LM oracle (oracle/amlm.py, restating lm_and_am/model/am_lm_model.py): analytic gradients
against central finite differences of mean_loss = am_mean_loss + lm_mean_loss, through both halves (the language half's
gradient must reach the acoustic trunk through h7), and the kept-as-written label rule K1 (zeros kept)."""
import numpy as np

from oracle import amlm, ctc, dfcnn


def _setup():
    P, ops = amlm.init_params(24, 40, feat=16, widths=(4, 8, 16, 4, 128), heads=2, blocks=2, pos_max=16, seed=0, perturb=True)
    rng = np.random.default_rng(0)
    x = rng.standard_normal((2, 64, 16, 1))
    tp = np.zeros((2, 8), dtype=np.int32); tp[0, :3] = [3, 0, 7]; tp[1, :2] = [5, 5]
    return P, ops, x, tp, [3, 2], [8, 6]


def test_graph_shape_and_dense_labels():
    ops = dfcnn.graph('amlm', 1536)
    assert [o[2] for o in ops if o[0] == 'dense'] == ['h7', 'd'] and ops[-2][3:5] == (25 * 128, 128)
    assert sum(1 for o in ops if o[0] == 'cell' and o[5] == 1) == 3          # three NiN 1x1 convs (am_lm_model.py:59-61)
    assert amlm.dense_labels(np.array([[3, 0, 7, 0], [5, 5, 0, 0]]), [3, 2]) == [[3, 0, 7], [5, 5]]   # K1: zeros kept


def test_joint_gradients_match_finite_differences():
    P, ops, x, tp, tl, wl = _setup()
    out = amlm.train_step(P, ops, x, wl, tp, tl, 2, 2)
    assert np.isfinite(out['mean_loss']) and abs(out['mean_loss'] - out['am_mean_loss'] - out['lm_mean_loss']) < 1e-12
    assert np.abs(out['dh7_from_lm']).max() > 0

    def loss_of():
        return amlm.train_step(P, ops, x, wl, tp, tl, 2, 2, want_grads=False)['mean_loss']
    rng = np.random.default_rng(1)
    for path in [('am', 'h2', 'w'), ('am', 'h7', 'w'), ('am', 'h4n', 'w'), ('am', 'h1', 'gamma'), ('am', 'd', 'b'),
                 ('lm', 'pos'), ('lm', 'mha1', 'wq'), ('lm', 'ffn', 'w1'), ('lm', 'out_w')]:
        ref, g = P, out['grads']
        for k in path[:-1]:
            ref = ref[k]
        for k in path:
            g = g[k]
        a = ref[path[-1]]
        idx = tuple(int(rng.integers(0, s)) for s in a.shape)
        old, h = a[idx], 1e-5
        a[idx] = old + h; lp = loss_of()
        a[idx] = old - h; lm = loss_of()
        a[idx] = old
        fd = (lp - lm) / (2 * h)
        assert abs(fd - g[idx]) <= 1e-5 * max(1e-3, abs(fd)), (path, idx, fd, g[idx])


def test_language_half_uses_the_pinyin_labels_and_blank():
    """K2: the lm CTC is the CTC of its own logits against the acoustic labels with blank = V_pinyin - 1."""
    P, ops, x, tp, tl, wl = _setup()
    out = amlm.train_step(P, ops, x, wl, tp, tl, 2, 2, want_grads=False)
    loss, _ = ctc.ctc_loss_and_grad(out['lm_logits'], amlm.dense_labels(tp, tl), wl, blank=23)
    assert np.allclose(loss, out['lm_loss'])
    assert out['lm_logits'].shape == (8, 2, 40) and out['am_logits'].shape == (8, 2, 24)

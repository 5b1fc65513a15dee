"""CPU tests pinning the oracle against independent implementations available
offline (torch-CPU autograd, scipy, sklearn) and closed-form known answers.
The reference holds no fixtures for this path, so these are the
strongest pins available: 'parity unpinned' still applies w.r.t. TensorFlow."""
import math

import numpy as np
import pytest
import torch

from oracle import fbank as ofb, nn as onn, ctc as octc, dfcnn, optim as oopt, torch_ref


# ----------------------------------------------------------------- fbank
def test_fbank_shapes_and_empty_filters():
    rng = np.random.default_rng(1234)
    sig = 0.1 * rng.standard_normal(160000)
    f = ofb.compute_fbank_from_api(sig, 16000, nfilt=200)
    assert f.shape == (999, 200)
    fb = ofb.get_filterbanks(200, 512, 16000)
    empty = np.where(fb.sum(axis=1) == 0)[0]
    assert len(empty) == 43 and empty.max() < 89
    assert int((fb != 0).sum()) == 353
    assert np.all(f[:, empty] == 0.0)
    nz = np.setdiff1d(np.arange(200), empty)
    assert np.allclose(f[:, nz].mean(axis=0), 0, atol=1e-12)
    assert np.allclose(f[:, nz].std(axis=0), 1, atol=1e-12)


def test_fbank_vs_scipy_sklearn():
    from scipy import fft as sfft
    from sklearn import preprocessing
    rng = np.random.default_rng(7)
    sig = rng.standard_normal(16000) * 0.05
    lf = ofb.logfbank(sig, 16000, nfilt=200)
    # independent recomputation of one frame
    pre = np.append(sig[0], sig[1:] - 0.97 * sig[:-1])
    fr = pre[160 * 5:160 * 5 + 400]
    ps = np.abs(sfft.rfft(fr, 512)) ** 2 / 512
    fb = ofb.get_filterbanks(200, 512, 16000)
    e = ps @ fb.T
    e[e == 0] = np.finfo(float).eps
    assert np.allclose(lf[5], np.log(e), rtol=1e-12, atol=1e-12)
    assert np.allclose(ofb.scale(lf), preprocessing.scale(lf), atol=1e-10)


def test_fbank_zero_signal_and_short():
    f = ofb.compute_fbank_from_api(np.zeros(16000), 16000)
    assert f.shape == (99, 200) and np.all(f == 0)
    assert ofb.num_frames(160000) == 999 and ofb.num_frames(300) == 1 and ofb.num_frames(16000) == 99


def test_lfr_docstring_contract():
    x = np.arange(30, dtype=np.float64).reshape(10, 3)
    y = ofb.build_LFR_features(x, 4, 3)
    assert y.shape == (4, 12)
    assert np.array_equal(y[0], x[0:4].ravel())
    assert np.array_equal(y[3], np.concatenate([x[9], x[9], x[9], x[9]]))
    assert np.array_equal(ofb.build_LFR_features(x, 1, 1), x)


# ----------------------------------------------------------------- layers vs torch autograd
def _t(a):
    return torch.tensor(a, dtype=torch.float64, requires_grad=True)


@pytest.mark.parametrize("pool", [None, "avg", "max"])
@pytest.mark.parametrize("k", [3, 1])
def test_cell_fwd_bwd_vs_torch(pool, k):
    rng = np.random.default_rng(0)
    B, H, W, cin, cout = 2, 9, 7, 3, 5
    x = rng.standard_normal((B, H, W, cin))
    p = {'w': rng.standard_normal((k, k, cin, cout)) * 0.3, 'b': rng.standard_normal(cout) * 0.1,
         'gamma': 1 + 0.2 * rng.standard_normal(cout), 'beta': 0.1 * rng.standard_normal(cout)}
    out, cache = onn.cell_fwd(x, p, pool)
    dout = rng.standard_normal(out.shape)
    dx, g = onn.cell_bwd(cache, p, dout, pool)
    tx = _t(x)
    tp = {kk: _t(v) for kk, v in p.items()}
    ty = torch_ref.cell(tx.permute(0, 3, 1, 2), tp, pool)
    assert np.allclose(out, ty.permute(0, 2, 3, 1).detach().numpy(), atol=1e-12)
    ty.backward(torch.tensor(dout).permute(0, 3, 1, 2))
    assert np.allclose(dx, tx.grad.numpy(), atol=1e-11)
    for kk in p:
        assert np.allclose(g[kk], tp[kk].grad.numpy(), atol=1e-10), kk


def test_se_fwd_bwd_vs_torch():
    rng = np.random.default_rng(1)
    B, H, W, C, hid = 2, 5, 4, 6, 3
    x = rng.standard_normal((B, H, W, C))
    p = {'gamma': 1 + 0.2 * rng.standard_normal(C), 'beta': 0.1 * rng.standard_normal(C),
         'w1': rng.standard_normal((C, hid)), 'b1': rng.standard_normal(hid),
         'w2': rng.standard_normal((hid, C)), 'b2': rng.standard_normal(C)}
    out, cache = onn.se_fwd(x, p)
    dout = rng.standard_normal(out.shape)
    dx, g = onn.se_bwd(cache, p, dout)
    tx = _t(x)
    tp = {kk: _t(v) for kk, v in p.items()}
    ty = torch_ref.se(tx.permute(0, 3, 1, 2), tp, True)
    assert np.allclose(out, ty.permute(0, 2, 3, 1).detach().numpy(), atol=1e-12)
    ty.backward(torch.tensor(dout).permute(0, 3, 1, 2))
    assert np.allclose(dx, tx.grad.numpy(), atol=1e-11)
    for kk in p:
        assert np.allclose(g[kk], tp[kk].grad.numpy(), atol=1e-10), kk


def test_bn_batch_vs_torch():
    rng = np.random.default_rng(2)
    a = rng.standard_normal((3, 4, 5, 6))
    gm, bt = 1 + 0.1 * rng.standard_normal(6), rng.standard_normal(6)
    y, cache = onn.bn_batch(a, gm, bt)
    dy = rng.standard_normal(y.shape)
    da, dg, db = onn.bn_batch_bwd(cache, gm, dy)
    ta, tg, tb = _t(a), _t(gm), _t(bt)
    ty = torch.nn.functional.batch_norm(ta.permute(0, 3, 1, 2), None, None, tg, tb, True, 0.0, 1e-3)
    assert np.allclose(y, ty.permute(0, 2, 3, 1).detach().numpy(), atol=1e-12)
    ty.backward(torch.tensor(dy).permute(0, 3, 1, 2))
    assert np.allclose(da, ta.grad.numpy(), atol=1e-10)
    assert np.allclose(dg, tg.grad.numpy(), atol=1e-10) and np.allclose(db, tb.grad.numpy(), atol=1e-10)


def test_maxpool_first_max_tie_rule():
    y = np.zeros((1, 2, 2, 1))
    dy = onn.maxpool2_bwd(y, np.ones((1, 1, 1, 1)))
    assert dy[0, 0, 0, 0] == 1 and dy.sum() == 1
    y = np.array([1., 3., 3., 2.]).reshape(1, 2, 2, 1)
    dy = onn.maxpool2_bwd(y, np.ones((1, 1, 1, 1)))
    assert dy.ravel().tolist() == [0, 1, 0, 0]


# ----------------------------------------------------------------- CTC
def test_ctc_closed_form():
    V = 4
    blank = V - 1
    # uniform logits: every symbol has prob 1/V after TF's softmax
    x = np.zeros((1, 1, V))
    loss, g = octc.ctc_loss_and_grad(x, [[1]], [1], blank)
    assert np.isclose(loss[0], math.log(V))
    x = np.zeros((2, 1, V))
    loss, _ = octc.ctc_loss_and_grad(x, [[1]], [2], blank)      # paths: 1-, -1, 11
    assert np.isclose(loss[0], -math.log(3.0 / V ** 2))
    x = np.zeros((3, 1, V))
    loss, _ = octc.ctc_loss_and_grad(x, [[2, 2]], [3], blank)   # only 2-2
    assert np.isclose(loss[0], -math.log(1.0 / V ** 3))
    with pytest.raises(ValueError):
        octc.ctc_loss_and_grad(np.zeros((2, 1, V)), [[2, 2]], [2], blank)
    # empty label: only the all-blank path
    loss, _ = octc.ctc_loss_and_grad(np.zeros((3, 1, V)), [[]], [3], blank)
    assert np.isclose(loss[0], 3 * math.log(V))


def test_ctc_vs_torch_random():
    rng = np.random.default_rng(3)
    T, B, V = 12, 3, 7
    x = rng.standard_normal((T, B, V))
    labels = [[1, 1, 2], [3], [4, 5, 4, 4]]
    seq = [12, 7, 10]
    loss, g = octc.ctc_loss_and_grad(x, labels, seq, V - 1)
    tx = _t(x)
    lp = torch.log_softmax(tx, dim=-1)
    tl = torch.nn.functional.ctc_loss(lp, torch.tensor([v for l in labels for v in l]),
                                      torch.tensor(seq), torch.tensor([len(l) for l in labels]),
                                      blank=V - 1, reduction='none')
    assert np.allclose(loss, tl.detach().numpy(), atol=1e-10)
    tl.sum().backward()
    assert np.allclose(g, tx.grad.numpy(), atol=1e-9)
    assert np.all(g[7:, 1] == 0)


def test_dense_to_sparse_drops_zeros():
    assert octc.dense_to_sparse([[5, 0, 7, 0], [0, 0, 0, 0]]) == [[5, 7], []]


def test_greedy_decode_rules():
    V = 4      # blank = 3
    def onehot(seq):
        x = np.full((len(seq), 1, V), -5.0)
        for t, k in enumerate(seq):
            x[t, 0, k] = 0.0
        return x
    dec, neg = octc.ctc_greedy_decode(onehot([1, 1, 3, 1, 2, 2, 3]), [7])
    assert dec == [[1, 1, 2]] and neg[0] == 0.0
    dec, _ = octc.ctc_greedy_decode(onehot([1, 1, 3, 1, 2, 2, 3]), [2])
    assert dec == [[1]]
    x = np.zeros((1, 1, V))                     # tie -> lowest index 0
    dec, neg = octc.ctc_greedy_decode(x, [1])
    assert dec == [[0]]
    idx, val, shp = octc.decoded_to_sparse([[1, 2], [], [3]])
    assert idx.tolist() == [[0, 0], [0, 1], [2, 0]] and val.tolist() == [1, 2, 3] and shp.tolist() == [3, 2]


def test_edit_distance():
    assert octc.levenshtein([1, 2, 3], [1, 3]) == 1
    assert octc.edit_distance_normalized([1, 2, 3], [1, 3]) == 0.5
    assert octc.edit_distance_normalized([], []) == 0.0
    assert octc.edit_distance_normalized([1], []) == float('inf')
    assert octc.get_edit_distance_difflib('abcd', 'abxyd') == 2


# ----------------------------------------------------------------- whole model vs torch autograd
@pytest.mark.parametrize("model,widths", [("m2", (4, 6, 8, 10)), ("m1", (4, 6, 8, 10, 3, 7)), ("m3", (4, 6, 8, 10))])
def test_model_step_vs_torch(model, widths):
    V, B, T, Fq = 11, 2, 32, 16
    ops = dfcnn.graph(model, V, widths, feat=Fq)
    P = dfcnn.init_params(ops, seed=1, perturb=True)
    rng = np.random.default_rng(5)
    x = rng.standard_normal((B, T, Fq, 1))
    x[1, 24:] = 0
    target = np.array([[3, 0, 4, 4], [9, 1, 0, 0]])
    seq = [4, 3]
    out = dfcnn.train_step_oracle(ops, P, x, seq, target)
    tP = torch_ref.to_torch_params(P)
    labels = octc.dense_to_sparse(target)
    logits_tm, loss, mean_loss, _ = torch_ref.train_step(ops, tP, torch.tensor(x), seq, labels)
    assert np.allclose(out['logits'], logits_tm.numpy(), atol=1e-10)
    assert np.allclose(out['loss'][:, 0], loss.numpy(), atol=1e-9)
    for name in P:
        for kk in P[name]:
            assert np.allclose(out['grads'][name][kk], tP[name][kk].grad.numpy(), atol=1e-9), (name, kk)


# ----------------------------------------------------------------- optimiser
def test_polynomial_decay_cycle():
    lr0, end = 7e-4, 1e-6
    assert oopt.polynomial_decay(lr0, 0) == lr0
    assert np.isclose(oopt.polynomial_decay(lr0, 1), (lr0 - end) * math.sqrt(1 - 1 / 5000) + end)
    assert np.isclose(oopt.polynomial_decay(lr0, 5000), end)
    assert np.isclose(oopt.polynomial_decay(lr0, 5001), (lr0 - end) * math.sqrt(1 - 5001 / 10000) + end)
    assert np.isclose(oopt.polynomial_decay(lr0, 4999), (lr0 - end) * math.sqrt(1 - 4999 / 5000) + end)


def test_adam_tf_vs_torch_adam_first_steps():
    # torch.optim.Adam uses eps outside the bias-corrected sqrt -> differs from TF's
    # "epsilon hat" form; check TF's form against its own closed-form first step.
    g = np.array([0.5, -2.0])
    th, m, v = oopt.adam_tf_step(np.zeros(2), g, np.zeros(2), np.zeros(2), 1e-3, 1)
    lr_t = 1e-3 * math.sqrt(1 - 0.999) / (1 - 0.9)
    assert np.allclose(th, -lr_t * (0.1 * g) / (np.sqrt(0.001 * g * g) + 1e-8))

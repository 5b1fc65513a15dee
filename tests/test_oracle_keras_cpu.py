"""CPU checks of the Keras-variant oracle (oracle/keras_dfcnn.py): graph shape per cnn_ctc.py:27-49, the
K.ctc_batch_cost label convention (first label_length ids, zeros kept), gradients against finite differences."""
import numpy as np
import torch

from oracle import keras_dfcnn as ok

CELLS = [(4, True), (8, True), (8, True), (8, False)]


def test_graph_shapes_and_softmax_rows():
    P = ok.init_params(11, feat=16, cells=CELLS, hidden=12, seed=0)
    assert P['c1a/w'].shape == (3, 3, 1, 4) and P['c2a/w'].shape == (3, 3, 4, 8) and P['d1/w'].shape == (2 * 8, 12)
    y = ok.forward({k: torch.tensor(v) for k, v in P.items()}, torch.tensor(np.random.default_rng(0).standard_normal((2, 32, 16))), CELLS).detach().numpy()
    assert y.shape == (2, 4, 11) and np.allclose(y.sum(-1), 1.0)


def test_labels_keep_zero_ids_up_to_label_length():
    rng = np.random.default_rng(1)
    P = ok.init_params(9, feat=16, cells=CELLS, hidden=8, seed=1)
    x = rng.standard_normal((2, 32, 16))
    a = ok.train_step(P, x, np.array([[3, 0, 5, 0], [2, 2, 0, 0]]), [3, 2], [4, 4], CELLS)      # label 0 is a real class here
    b = ok.train_step(P, x, np.array([[3, 0, 5, 7], [2, 2, 6, 6]]), [3, 2], [4, 4], CELLS)      # ids past label_length ignored
    assert np.allclose(a['loss'], b['loss']) and a['loss'].shape == (2,)
    c = ok.train_step(P, x, np.array([[3, 5, 0, 0], [2, 2, 0, 0]]), [2, 2], [4, 4], CELLS)
    assert not np.allclose(a['loss'][0], c['loss'][0])                                             # ... and 0 was not dropped


def test_gradients_match_finite_differences():
    rng = np.random.default_rng(2)
    P = ok.init_params(7, feat=8, cells=[(4, True), (4, True), (4, True)], hidden=8, seed=2)
    cells = [(4, True), (4, True), (4, True)]
    x = rng.standard_normal((2, 16, 8))
    labels, ll, il = np.array([[1, 2], [3, 0]]), [2, 1], [2, 2]
    r = ok.train_step(P, x, labels, ll, il, cells)
    f = lambda PP: ok.train_step(PP, x, labels, ll, il, cells)['mean_loss']
    for name, idx in (('c1a/w', (1, 1, 0, 2)), ('c2b/g', (1,)), ('c3a/b', (0,)), ('d1/w', (3, 5)), ('d2/b', (4,)), ('c1b/be', (2,))):
        Pp = {k: v.copy() for k, v in P.items()}; Pm = {k: v.copy() for k, v in P.items()}
        h = 1e-6
        Pp[name][idx] += h; Pm[name][idx] -= h
        fd = (f(Pp) - f(Pm)) / (2 * h)
        assert abs(fd - r['grads'][name][idx]) < 1e-5 * max(1.0, abs(fd)), (name, fd, r['grads'][name][idx])

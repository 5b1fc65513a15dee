"""CPU checks of the pre-net oracle (oracle/prenet.py): TF 'same' stride-2 geometry against explicit loops,
batch-statistics BatchNorm against its definition, the dead first loop iteration / mask=False reading of
end2end/model.py:225-267 expressed as properties, and the invariances the parity tests rely on."""
import numpy as np
import torch

from oracle import prenet as opn


def test_same_padding_rule_matches_tensorflow_formula():
    # out = ceil(n / s); pad_total = max((out-1)*s + k - n, 0); before = total // 2
    assert opn.same_pad(320, 3, 2) == (0, 1) and opn.same_pad(7, 3, 2) == (1, 1)
    assert opn.same_pad(80, 3, 1) == (1, 1) and opn.same_pad(1, 3, 2) == (1, 1)


def test_stride2_conv_against_explicit_loops():
    rng = np.random.default_rng(0)
    for H, W in ((6, 8), (5, 7)):
        x = rng.standard_normal((1, H, W, 2)); w = rng.standard_normal((3, 3, 2, 3)); b = rng.standard_normal(3)
        y = opn.conv(torch.tensor(x), torch.tensor(w), torch.tensor(b), 2).numpy()[0]
        pt, _ = opn.same_pad(H, 3, 2); pl, _ = opn.same_pad(W, 3, 2)
        Ho, Wo = -(-H // 2), -(-W // 2)
        ref = np.zeros((Ho, Wo, 3))
        for i in range(Ho):
            for j in range(Wo):
                acc = b.copy()
                for dh in range(3):
                    for dw in range(3):
                        r, c = 2 * i + dh - pt, 2 * j + dw - pl
                        if 0 <= r < H and 0 <= c < W:
                            acc += x[0, r, c] @ w[dh, dw]
                ref[i, j] = acc
        assert np.abs(y - ref).max() < 1e-12


def test_batch_norm_is_batch_moments_with_biased_variance():
    rng = np.random.default_rng(1)
    x = rng.standard_normal((3, 4, 5, 6)) * 2 + 1
    g, b = rng.standard_normal(6), rng.standard_normal(6)
    y = opn.batch_norm(torch.tensor(x), torch.tensor(g), torch.tensor(b)).numpy()
    mu, var = x.mean(axis=(0, 1, 2)), x.var(axis=(0, 1, 2))
    assert np.abs(y - (g * (x - mu) / np.sqrt(var + 1e-3) + b)).max() < 1e-12


def test_forward_shapes_and_invariances():
    rng = np.random.default_rng(2)
    P = opn.init_params(seed=1)
    x = rng.standard_normal((2, 8, 16))
    out = opn.forward(opn.to_torch(P, False), torch.tensor(x)).numpy()
    assert out.shape == (2, 2, 4, 64) and out.min() >= 0.0           # closes with a ReLU
    # a conv bias directly in front of a batch-stat BN cannot change anything; neither can the BN shift of k
    # (a constant added to every key shifts each score row by a constant, which the softmax ignores)
    Q = {k: v.copy() for k, v in P.items()}
    for name in ('q/b', 'k/b', 'v/b', 'f2/b', 'bnk/b'):
        Q[name] = Q[name] + rng.standard_normal(Q[name].shape)
    out2 = opn.forward(opn.to_torch(Q, False), torch.tensor(x)).numpy()
    assert np.abs(out - out2).max() < 1e-9


def test_autograd_gradients_match_finite_differences():
    rng = np.random.default_rng(3)
    P = opn.init_params(seed=2)
    x = rng.standard_normal((1, 8, 8))
    dout = rng.standard_normal((1, 2, 2, 64))
    _, grads, _ = opn.forward_backward(P, x, dout)
    f = lambda PP: float((opn.forward(opn.to_torch(PP, False), torch.tensor(x)).numpy() * dout).sum())
    for name, idx in (('conv1/w', (1, 2, 0, 5)), ('conv2/w', (0, 1, 3, 7)), ('bn2/g', (4,)), ('merge/w', (2, 2, 100, 9)),
                      ('ln/b', (11,)), ('f1/b', (3,)), ('v/w', (1, 1, 8, 8))):
        Pp = {k: v.copy() for k, v in P.items()}; Pm = {k: v.copy() for k, v in P.items()}
        h = 1e-5
        Pp[name][idx] += h; Pm[name][idx] -= h
        fd = (f(Pp) - f(Pm)) / (2 * h)
        assert abs(fd - grads[name][idx]) < 1e-5 * max(1.0, abs(fd)), (name, fd, grads[name][idx])

"""Pins oracle/transformer.py (numpy, hand-derived backward) against an independent
torch-CPU autograd restatement of end2end/transformer.py's semantics."""
import numpy as np
import pytest
import torch

from oracle import transformer as otr

FILL = float(-2 ** 32 + 1)


def t64(a, grad=True):
    return torch.tensor(np.asarray(a, dtype=np.float64), requires_grad=grad)


def t_layer_norm(x, g, b):
    mu = x.mean(-1, keepdim=True)
    var = x.var(-1, unbiased=False, keepdim=True)
    return g * (x - mu) / (var + 1e-8) ** 0.5 + b


def t_mha(q_in, k_in, p, h, causal):
    Q, K, V = torch.relu(q_in @ p['wq']), torch.relu(k_in @ p['wk']), torch.relu(k_in @ p['wv'])
    N, Tq, C = Q.shape
    Tk = K.shape[1]
    sp = lambda x, T: torch.cat(torch.split(x, C // h, dim=2), dim=0)          # (h*N, T, C/h) like tf.concat(tf.split)
    Q_, K_, V_ = sp(Q, Tq), sp(K, Tk), sp(V, Tk)
    S = Q_ @ K_.transpose(1, 2) / (C // h) ** 0.5
    km = torch.sign(torch.abs(K_.sum(-1)))[:, None, :].expand(-1, Tq, -1)
    S = torch.where(km == 0, torch.full_like(S, FILL), S)
    if causal:
        tril = torch.tril(torch.ones(Tq, Tk, dtype=S.dtype))[None].expand(S.shape[0], -1, -1)
        S = torch.where(tril == 0, torch.full_like(S, FILL), S)
    Pm = torch.softmax(S, dim=-1)
    qm = torch.sign(torch.abs(Q_).sum(-1))[:, :, None]
    O = (Pm * qm) @ V_
    O = torch.cat(torch.split(O, N, dim=0), dim=2)
    Z = torch.relu(O @ p['wo'])
    return t_layer_norm(Z + q_in, p['ln_g'], p['ln_b'])


def t_ffn(x, p):
    y = torch.relu(x @ p['w1'] + p['b1']) @ p['w2'] + p['b2']
    return t_layer_norm(y + x, p['ln_g'], p['ln_b'])


def t_ce(logits, target):
    V = logits.shape[-1]
    oh = torch.zeros_like(logits)
    valid = (target >= 0) & (target < V)
    idx = torch.nonzero(valid, as_tuple=True)
    oh[idx[0], idx[1], target[valid]] = 1.0
    ys = 0.9 * oh + 0.1 / V
    loss = -(ys * torch.log_softmax(logits, -1)).sum(-1)
    ist = (target != 0).double()
    return (loss * ist).sum() / ist.sum()


def to_t(P):
    return {k: (to_t(v) if isinstance(v, dict) else t64(v)) for k, v in P.items()}


def check_grads(G, tP, path=''):
    for k, v in G.items():
        if isinstance(v, dict):
            check_grads(v, tP[k], path + k + '/')
        else:
            assert tP[k].grad is not None, path + k
            assert np.allclose(v, tP[k].grad.numpy(), atol=1e-9), path + k


def make_x(rng, N, T, V, pad_tail=True):
    x = rng.integers(1, V, (N, T))
    if pad_tail:
        x[0, T - 2:] = 0                      # padded keys -> embedding row 0 = zeros -> key masked
    return x


def test_lm_step_vs_torch():
    rng = np.random.default_rng(0)
    N, T, C, h, blocks, Vin, Vout = 2, 7, 16, 4, 3, 11, 13
    P = otr.init_lm(Vin, Vout, C, h, blocks, pos_max=10, seed=1, perturb=True)
    x = make_x(rng, N, T, Vin)
    y = rng.integers(1, Vout, (N, T))
    y[0, T - 2:] = 0
    out = otr.lm_step(P, x, y, h, blocks)
    tP = to_t(P)
    emb = torch.cat([torch.zeros(1, C, dtype=torch.float64), tP['emb'][1:]], 0)[torch.tensor(x)] * C ** 0.5
    enc = emb + tP['pos'][torch.arange(T)][None]
    for i in range(blocks):
        enc = t_mha(enc, enc, tP['mha%d' % i], h, True)
    logits = t_ffn(enc, tP['ffn']) @ tP['out_w'] + tP['out_b']
    loss = t_ce(logits, torch.tensor(y))
    assert np.allclose(out['logits'], logits.detach().numpy(), atol=1e-10)
    assert np.isclose(out['mean_loss'], loss.item(), atol=1e-12)
    loss.backward()
    check_grads(out['grads'], tP)
    assert np.all(out['grads']['emb'][0] == 0)            # zero_pad row never receives gradient


@pytest.mark.parametrize("tie", [True, False])
def test_e2e_step_vs_torch(tie):
    rng = np.random.default_rng(1)
    N, T, L, Din, C, h, blocks, Vout = 2, 9, 6, 12, 16, 4, 2, 15
    P = otr.init_e2e(Din, Vout, C, h, blocks, pos_max=12, seed=2, perturb=True, tie=tie)
    xf = rng.standard_normal((N, T, Din))
    y_in = rng.integers(1, Vout, (N, L))
    y_tgt = rng.integers(1, Vout, (N, L))
    y_tgt[1, L - 2:] = -1                               # IGNORE padding still counted (Q9)
    y_tgt[0, L - 1] = 0
    out = otr.e2e_step(P, xf, y_in, y_tgt, h, blocks, tie=tie)
    # build torch params with the same sharing structure
    tP, shared = {}, {}
    for k, v in P.items():
        if isinstance(v, dict):
            tP[k] = {}
            for kk, vv in v.items():
                key = id(vv)
                if key not in shared:
                    shared[key] = t64(vv)
                tP[k][kk] = shared[key]
        else:
            tP[k] = t64(v)
    u = torch.relu(t64(xf, False) @ tP['in_w'] + tP['in_b'])
    enc = t_layer_norm(u, tP['in_ln_g'], tP['in_ln_b']) + tP['enc_pe'][torch.arange(T)][None]
    dec = tP['dec_input'][torch.tensor(y_in)] + tP['dec_pe'][torch.arange(L)][None]
    for i in range(blocks):
        enc = t_mha(enc, enc, tP['enc%d' % i], h, False)
    mem = t_ffn(enc, tP['enc_ffn'])
    for i in range(blocks):
        dec = t_mha(dec, mem, tP['dec%d' % i], h, True)
    logits = t_ffn(dec, tP['dec_ffn']) @ tP['out_w'] + tP['out_b']
    loss = t_ce(logits, torch.tensor(y_tgt))
    assert np.allclose(out['logits'], logits.detach().numpy(), atol=1e-10)
    assert np.isclose(out['mean_loss'], loss.item(), atol=1e-12)
    loss.backward()
    check_grads(out['grads'], tP)


def test_e2e_step_with_id_input_vs_torch():
    """init_e2e(vin=...) / e2e_step on ids (BASELINE configs[3], pinyin -> hanzi): encoder input = embedding with zero_pad
    and sqrt(C) scale (language_model.py:28) + positions; checked against torch autograd, incl. the never-updated row 0."""
    rng = np.random.default_rng(4)
    N, T, L, C, h, blocks, Vin, Vout = 2, 7, 5, 16, 4, 2, 11, 13
    P = otr.init_e2e(None, Vout, C, h, blocks, pos_max=9, seed=3, perturb=True, tie=True, vin=Vin)
    assert 'enc_emb' in P and 'in_w' not in P
    x = rng.integers(1, Vin, (N, T)); x[0, T - 2:] = 0              # padded positions: zero rows -> key-masked
    y_in = rng.integers(1, Vout, (N, L)); y_tgt = rng.integers(1, Vout, (N, L)); y_tgt[1, L - 1] = -1
    out = otr.e2e_step(P, x, y_in, y_tgt, h, blocks, tie=True)
    tP, shared = {}, {}
    for k, v in P.items():
        if isinstance(v, dict):
            tP[k] = {}
            for kk, vv in v.items():
                if id(vv) not in shared:
                    shared[id(vv)] = t64(vv)
                tP[k][kk] = shared[id(vv)]
        else:
            tP[k] = t64(v)
    xt = torch.tensor(x)
    emb = tP['enc_emb'][xt] * (xt != 0).unsqueeze(-1) * (C ** 0.5)
    enc = emb + tP['enc_pe'][torch.arange(T)][None]
    dec = tP['dec_input'][torch.tensor(y_in)] + tP['dec_pe'][torch.arange(L)][None]
    for i in range(blocks):
        enc = t_mha(enc, enc, tP['enc%d' % i], h, False)
    mem = t_ffn(enc, tP['enc_ffn'])
    for i in range(blocks):
        dec = t_mha(dec, mem, tP['dec%d' % i], h, True)
    logits = t_ffn(dec, tP['dec_ffn']) @ tP['out_w'] + tP['out_b']
    loss = t_ce(logits, torch.tensor(y_tgt))
    assert np.allclose(out['logits'], logits.detach().numpy(), atol=1e-10)
    assert np.isclose(out['mean_loss'], loss.item(), atol=1e-12)
    loss.backward()
    check_grads(out['grads'], tP)
    assert np.all(out['grads']['enc_emb'][0] == 0)                    # zero_pad: row 0 never receives gradient


def test_all_keys_masked_row_is_uniform_and_has_no_score_gradient():
    # one head whose keys are all zero rows -> every score is the fill value -> uniform softmax
    rng = np.random.default_rng(2)
    Q = np.abs(rng.standard_normal((1, 3, 4)))
    K = np.zeros((1, 5, 4))
    V = np.abs(rng.standard_normal((1, 5, 4)))
    O, cache = otr.attention_core(Q, K, V, 1, causal=False)
    assert np.allclose(O[0], V[0].mean(axis=0)[None].repeat(3, 0))
    dQ, dK, dV = otr.attention_core_bwd(cache, np.ones_like(O))
    assert np.all(dQ == 0) and np.all(dK == 0) and np.allclose(dV, 3 / 5)


def test_smoothed_ce_ignore_target():
    logits = np.zeros((1, 2, 4))
    ml, acc, preds, loss, dl = otr.smoothed_ce(logits, np.array([[2, -1]]))
    # valid target: ys sums to 1 -> loss = log 4 ; IGNORE target: ys = 0.1/4 each -> loss = 0.1*log 4
    assert np.allclose(loss, [[np.log(4), 0.1 * np.log(4)]]) and np.isclose(ml, 1.1 * np.log(4) / 2)


def test_dropout_mask_generator_properties():
    """The counter-based mask of asr_dropout as the oracle restates it: deterministic, seed-sensitive, keep rate 1 - rate,
    inverted scaling, and rate 0 keeps everything."""
    m = otr.drop_scale_mask((1000, 50), 0.2, 77)
    assert np.array_equal(m, otr.drop_scale_mask((1000, 50), 0.2, 77))
    assert not np.array_equal(m != 0, otr.drop_scale_mask((1000, 50), 0.2, 78) != 0)
    assert abs((m != 0).mean() - 0.8) < 5e-3 and np.allclose(m[m != 0], 1.25)
    assert np.all(otr.drop_scale_mask((64,), 0.0, 5) == 1.0)
    # site seeds are distinct per site and step
    seeds = {otr.drop_site_seed(1, st, s) for st in (0, 1) for s in ('emb', 'ffn', ('mha', 0, 'att'), ('mha', 0, 'out'), ('dec', 5, 'out'))}
    assert len(seeds) == 10


def test_dropout_backward_is_the_masked_gradient():
    rng = np.random.default_rng(9)
    N, T, C, H = 2, 5, 128, 2
    p = otr.init_mha(rng, C, perturb=True)
    x = rng.standard_normal((N, T, C))
    drop = otr.Drop(0.3, seed=11, step=2)
    out, cache = otr.mha_fwd(x, x, p, H, causal=True, drop=drop, site=('mha', 1))
    dout = rng.standard_normal(out.shape)
    dx, _, g = otr.mha_bwd(cache, p, dout, self_attn=True)
    f = lambda xx, pp: float((otr.mha_fwd(xx, xx, pp, H, causal=True, drop=drop, site=('mha', 1))[0] * dout).sum())
    h = 1e-6
    xp, xm = x.copy(), x.copy(); xp[1, 2, 7] += h; xm[1, 2, 7] -= h
    assert abs((f(xp, p) - f(xm, p)) / (2 * h) - dx[1, 2, 7]) < 1e-5
    pp = {k: v.copy() for k, v in p.items()}; pm = {k: v.copy() for k, v in p.items()}
    pp['wo'][3, 4] += h; pm['wo'][3, 4] -= h
    assert abs((f(x, pp) - f(x, pm)) / (2 * h) - g['wo'][3, 4]) < 1e-5

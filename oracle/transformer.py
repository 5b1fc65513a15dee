"""Oracle for the Transformer path (float64 numpy, hand-derived backward).  Test
infrastructure only; parity unpinned (see oracle/__init__.py).

Restates end2end/transformer.py:
  layer_norm :4-27, embedding :30-55, mask :58-86, scaled_dot_product_attention :89-115,
  multihead_attention :118-158, feedforward :204-231, label_smoothing :332-340
and the live graphs of lm_and_am/model/language_model.py:22-78 and
end2end/model.py:267-370 as SURVEY.md Q7-Q9 describe them: in every block loop the FFN
output is stored in another attribute than the one the next block reads, so N stacked MHA
sub-layers + ONE live FFN (of the last block) remain; projections are Dense(relu, no bias);
key mask = rows of the per-head K whose sum is 0, fill value -2**32+1; query mask multiplies
the post-softmax matrix; LayerNorm eps 1e-8 with biased variance.  Dropout is the identity
here (parity runs use rate 0 / is_training False: SURVEY Q5).
"""
import numpy as np

MASK_FILL = float(-2 ** 32 + 1)
LN_EPS = 1e-8


# ----------------------------------------------------------------- pieces
def layer_norm(x, gamma, beta, eps=LN_EPS):
    mu = x.mean(axis=-1, keepdims=True)
    var = x.var(axis=-1, keepdims=True)
    xh = (x - mu) / np.sqrt(var + eps)
    return gamma * xh + beta, (xh, var)


def layer_norm_bwd(cache, gamma, dy, eps=LN_EPS):
    xh, var = cache
    n = xh.shape[-1]
    dg = (dy * xh).reshape(-1, n).sum(axis=0)
    db = dy.reshape(-1, n).sum(axis=0)
    dxh = dy * gamma
    dx = (dxh - dxh.mean(axis=-1, keepdims=True) - xh * (dxh * xh).mean(axis=-1, keepdims=True)) / np.sqrt(var + eps)
    return dx, dg, db


def embedding(table, ids, zero_pad, scale):
    """transformer.py:30-55: row 0 reads as zeros when zero_pad; output * sqrt(d) when scale."""
    t = table.copy()
    if zero_pad:
        t[0] = 0.0
    out = t[ids]
    if scale:
        out = out * (table.shape[1] ** 0.5)
    return out


def embedding_bwd(table_shape, ids, dout, zero_pad, scale):
    g = np.zeros(table_shape)
    d = dout * (table_shape[1] ** 0.5) if scale else dout
    np.add.at(g, ids.reshape(-1), d.reshape(-1, table_shape[1]))
    if zero_pad:
        g[0] = 0.0
    return g


# ----------------------------------------------------------------- dropout (tf.layers.dropout with training=True)
# TensorFlow's random stream cannot be reproduced; the build defines its own counter-based mask (include/asr_hip.h,
# asr_dropout) and the oracle restates that generator: element i of the tensor drawn for `seed` is kept when the top
# 24 bits of murmur3_fmix32(i * 0x9E3779B1 + seed) reach rate * 2^24; kept values are scaled by 1 / (1 - rate).
DROP_SITES = {'emb_enc': 0, 'emb_dec': 1, 'enc_ffn': 70, 'dec_ffn': 71, 'ffn': 70, 'emb': 0}


def drop_site_seed(base_seed, step, site):
    """site: a DROP_SITES name, or ('enc'|'dec'|'mha', block, 'att'|'out')."""
    if isinstance(site, tuple):
        kind, i, which = site
        sid = {'enc': 10, 'mha': 10, 'dec': 40}[kind] + 2 * i + (0 if which == 'att' else 1)
    else:
        sid = DROP_SITES[site]
    return (int(base_seed) + 1009 * int(step) + 7919 * sid) & 0xFFFFFFFF


def drop_scale_mask(shape, rate, seed):
    """float64 array of `shape`: 1/(1-rate) where kept, 0 where dropped (flat C-order element index)."""
    n = int(np.prod(shape))
    h = (np.arange(n, dtype=np.uint64) * np.uint64(0x9E3779B1) + np.uint64(seed)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(16); h = (h * np.uint64(0x85EBCA6B)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(13); h = (h * np.uint64(0xC2B2AE35)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(16)
    thr = np.uint64(int(min(max(float(rate) * 16777216.0, 0.0), 16777215.0) + 0.5))
    keep = (h >> np.uint64(8)) >= thr
    return (keep.astype(np.float64) / (1.0 - float(rate))).reshape(shape)


class Drop:
    """Dropout context of one training step: rate, base seed, step (all sites derive their seed from these)."""

    def __init__(self, rate, seed=0, step=0):
        self.rate, self.seed, self.step = float(rate), int(seed), int(step)

    def mask(self, shape, site):
        return drop_scale_mask(shape, self.rate, drop_site_seed(self.seed, self.step, site))


def _split(x, h):       # [N,T,C] -> [N,h,T,C/h]
    N, T, C = x.shape
    return x.reshape(N, T, h, C // h).transpose(0, 2, 1, 3)


def _merge(x):          # [N,h,T,d] -> [N,T,h*d]
    N, h, T, d = x.shape
    return x.transpose(0, 2, 1, 3).reshape(N, T, h * d)


def attention_core(Q, K, V, h, causal, drop_mask=None):
    """scaled_dot_product_attention on head-split relu'd projections Q,K,V [N,T,C].
    drop_mask [N,h,Tq,Tk] (0 or 1/(1-rate)): dropout of the attention weights, transformer.py:111."""
    Qh, Kh, Vh = _split(Q, h), _split(K, h), _split(V, h)
    d = Qh.shape[-1]
    S = np.einsum('nhqd,nhkd->nhqk', Qh, Kh) / (d ** 0.5)
    kmask = np.sign(np.abs(Kh.sum(axis=-1)))                       # [N,h,Tk]
    keep = np.broadcast_to(kmask[:, :, None, :] != 0, S.shape).copy()
    if causal:
        Tq, Tk = S.shape[-2:]
        keep &= np.tril(np.ones((Tq, Tk), dtype=bool))
    Sm = np.where(keep, S, MASK_FILL)
    m = Sm.max(axis=-1, keepdims=True)
    e = np.exp(Sm - m)
    P = e / e.sum(axis=-1, keepdims=True)
    qmask = np.sign(np.abs(Qh).sum(axis=-1))                       # [N,h,Tq]
    Pq = P * qmask[..., None]
    if drop_mask is not None:
        Pq = Pq * drop_mask
    O = np.einsum('nhqk,nhkd->nhqd', Pq, Vh)
    return _merge(O), (Qh, Kh, Vh, P, keep, qmask, drop_mask)


def attention_core_bwd(cache, dO):
    Qh, Kh, Vh, P, keep, qmask, drop_mask = cache
    h = Qh.shape[1]
    d = Qh.shape[-1]
    dOh = _split(dO, h)
    Pq = P * qmask[..., None]
    if drop_mask is not None:
        Pq = Pq * drop_mask
    dV = np.einsum('nhqk,nhqd->nhkd', Pq, dOh)
    dPq = np.einsum('nhqd,nhkd->nhqk', dOh, Vh)
    if drop_mask is not None:
        dPq = dPq * drop_mask
    dP = dPq * qmask[..., None]
    dSm = P * (dP - (dP * P).sum(axis=-1, keepdims=True))
    dS = np.where(keep, dSm, 0.0) / (d ** 0.5)                     # tf.where: no gradient into masked scores
    dQ = np.einsum('nhqk,nhkd->nhqd', dS, Kh)
    dK = np.einsum('nhqk,nhqd->nhkd', dS, Qh)
    return _merge(dQ), _merge(dK), _merge(dV)


def mha_fwd(q_in, k_in, p, h, causal, drop=None, site=None):
    """multihead_attention (transformer.py:118-158).  p: wq, wk, wv, wo [C,C], ln_g, ln_b.
    drop (Drop) + site ('enc'|'dec'|'mha', block): dropout of the attention weights (:111) and of the output (:154)."""
    Q = np.maximum(q_in @ p['wq'], 0)
    K = np.maximum(k_in @ p['wk'], 0)
    V = np.maximum(k_in @ p['wv'], 0)
    N, Tq, C = Q.shape
    m_att = drop.mask((N, h, Tq, K.shape[1]), site + ('att',)) if drop is not None else None
    A, c_att = attention_core(Q, K, V, h, causal, m_att)
    Z = np.maximum(A @ p['wo'], 0)
    m_out = drop.mask(Z.shape, site + ('out',)) if drop is not None else None
    Zd = Z * m_out if m_out is not None else Z
    out, c_ln = layer_norm(Zd + q_in, p['ln_g'], p['ln_b'])
    return out, (q_in, k_in, Q, K, V, A, Z, c_att, c_ln, m_out)


def mha_bwd(cache, p, dout, self_attn):
    q_in, k_in, Q, K, V, A, Z, c_att, c_ln, m_out = cache
    C = q_in.shape[-1]
    dr, dg, db = layer_norm_bwd(c_ln, p['ln_g'], dout)
    dq_in = dr.copy()
    dZ = dr * (Z > 0)
    if m_out is not None:
        dZ = dZ * m_out
    g = {'ln_g': dg, 'ln_b': db, 'wo': A.reshape(-1, C).T @ dZ.reshape(-1, C)}
    dA = dZ @ p['wo'].T
    dQ, dK, dV = attention_core_bwd(c_att, dA)
    dQ = dQ * (Q > 0); dK = dK * (K > 0); dV = dV * (V > 0)
    g['wq'] = q_in.reshape(-1, C).T @ dQ.reshape(-1, C)
    g['wk'] = k_in.reshape(-1, C).T @ dK.reshape(-1, C)
    g['wv'] = k_in.reshape(-1, C).T @ dV.reshape(-1, C)
    dq_in += dQ @ p['wq'].T
    dk_in = dK @ p['wk'].T + dV @ p['wv'].T
    if self_attn:
        return dq_in + dk_in, None, g
    return dq_in, dk_in, g


def ffn_fwd(x, p, drop=None, site=None):
    """feedforward (transformer.py:204-231): conv1d(k=1) = dense with bias; dropout of the output (:226)."""
    H = np.maximum(x @ p['w1'] + p['b1'], 0)
    Y = H @ p['w2'] + p['b2']
    m = drop.mask(Y.shape, site) if drop is not None else None
    Yd = Y * m if m is not None else Y
    out, c_ln = layer_norm(Yd + x, p['ln_g'], p['ln_b'])
    return out, (x, H, c_ln, m)


def ffn_bwd(cache, p, dout):
    x, H, c_ln, m = cache
    dr, dg, db = layer_norm_bwd(c_ln, p['ln_g'], dout)
    C, Fh = p['w1'].shape
    dY = dr * m if m is not None else dr
    g = {'ln_g': dg, 'ln_b': db, 'w2': H.reshape(-1, Fh).T @ dY.reshape(-1, C), 'b2': dY.reshape(-1, C).sum(axis=0)}
    dH = (dY @ p['w2'].T) * (H > 0)
    g['w1'] = x.reshape(-1, C).T @ dH.reshape(-1, Fh)
    g['b1'] = dH.reshape(-1, Fh).sum(axis=0)
    return dr + dH @ p['w1'].T, g


def smoothed_ce(logits, target, pad_id=0, eps=0.1):
    """label_smoothing(one_hot(target)) + softmax_cross_entropy_with_logits_v2 + masked mean
    (model.py:342-356, language_model.py:55-67).  target may hold -1 (IGNORE): one_hot is
    all-zero there, yet the position counts because the mask is target != PAD."""
    N, T, V = logits.shape
    oh = np.zeros((N, T, V))
    valid = (target >= 0) & (target < V)
    n_i, t_i = np.nonzero(valid)
    oh[n_i, t_i, target[valid]] = 1.0
    ys = (1 - eps) * oh + eps / V
    m = logits.max(axis=-1, keepdims=True)
    lse = m + np.log(np.exp(logits - m).sum(axis=-1, keepdims=True))
    logp = logits - lse
    loss = -(ys * logp).sum(axis=-1)
    ist = (target != pad_id).astype(np.float64)
    mean_loss = (loss * ist).sum() / ist.sum()
    preds = logits.argmax(axis=-1)
    acc = ((preds == target) * ist).sum() / ist.sum()
    dlogits = (np.exp(logp) * ys.sum(axis=-1, keepdims=True) - ys) * (ist / ist.sum())[..., None]
    return mean_loss, acc, preds, loss, dlogits


# ----------------------------------------------------------------- parameter init
def _glorot(rng, shape):
    lim = np.sqrt(6.0 / (shape[0] + shape[1]))
    return rng.uniform(-lim, lim, size=shape)


def init_mha(rng, C, perturb=False):
    p = {k: _glorot(rng, (C, C)) for k in ('wq', 'wk', 'wv', 'wo')}
    p['ln_g'], p['ln_b'] = np.ones(C), np.zeros(C)
    if perturb:
        p['ln_g'] = 1 + 0.1 * rng.standard_normal(C); p['ln_b'] = 0.1 * rng.standard_normal(C)
    return p


def init_ffn(rng, C, Fh, perturb=False):
    p = {'w1': _glorot(rng, (C, Fh)), 'b1': np.zeros(Fh), 'w2': _glorot(rng, (Fh, C)), 'b2': np.zeros(C),
         'ln_g': np.ones(C), 'ln_b': np.zeros(C)}
    if perturb:
        p['b1'] = 0.1 * rng.standard_normal(Fh); p['b2'] = 0.1 * rng.standard_normal(C)
        p['ln_g'] = 1 + 0.1 * rng.standard_normal(C); p['ln_b'] = 0.1 * rng.standard_normal(C)
    return p


def init_lm(vin, vout, C, heads, blocks, pos_max, seed=0, perturb=False):
    rng = np.random.default_rng(seed)
    P = {'emb': _glorot(rng, (vin, C)), 'pos': _glorot(rng, (pos_max, C)),
         'out_w': _glorot(rng, (C, vout)), 'out_b': np.zeros(vout)}
    if perturb:
        P['out_b'] = 0.1 * rng.standard_normal(vout)
    for i in range(blocks):
        P['mha%d' % i] = init_mha(rng, C, perturb)
    P['ffn'] = init_ffn(rng, C, 4 * C, perturb)        # only the last block's FFN is live (Q7)
    return P


# ----------------------------------------------------------------- Language_Model (language_model.py:22-78)
def lm_step(P, x, y, heads, blocks, want_grads=True, drop=None):
    C = P['emb'].shape[1]
    N, T = x.shape
    e = embedding(P['emb'], x, zero_pad=True, scale=True)
    pos = np.broadcast_to(np.arange(T)[None, :], (N, T))
    enc = e + embedding(P['pos'], pos, zero_pad=False, scale=False)
    m_emb = drop.mask(enc.shape, 'emb') if drop is not None else None          # language_model.py:34
    if m_emb is not None:
        enc = enc * m_emb
    caches = []
    for i in range(blocks):
        enc, c = mha_fwd(enc, enc, P['mha%d' % i], heads, causal=True, drop=drop, site=('mha', i))
        caches.append(c)
    outputs, c_ffn = ffn_fwd(enc, P['ffn'], drop, 'ffn')
    logits = outputs @ P['out_w'] + P['out_b']
    mean_loss, acc, preds, loss, dlogits = smoothed_ce(logits, y)
    out = {'logits': logits, 'mean_loss': mean_loss, 'acc': acc, 'preds': preds, 'enc': enc}
    if not want_grads:
        return out
    G = {'out_w': outputs.reshape(-1, C).T @ dlogits.reshape(N * T, -1), 'out_b': dlogits.reshape(N * T, -1).sum(axis=0)}
    d = dlogits @ P['out_w'].T
    d, G['ffn'] = ffn_bwd(c_ffn, P['ffn'], d)
    for i in reversed(range(blocks)):
        d, _, G['mha%d' % i] = mha_bwd(caches[i], P['mha%d' % i], d, self_attn=True)
    if m_emb is not None:
        d = d * m_emb
    G['emb'] = embedding_bwd(P['emb'].shape, x, d, zero_pad=True, scale=True)
    G['pos'] = embedding_bwd(P['pos'].shape, pos, d, zero_pad=False, scale=False)
    out['grads'] = G
    return out


# ----------------------------------------------------------------- end2end enc-dec (model.py:267-370)
def init_e2e(din, vout, C, heads, blocks, pos_max, seed=0, perturb=False, tie=True, vin=None):
    """``vin``: the encoder input is a sequence of ids (BASELINE.json configs[3], pinyin -> hanzi) looked up in an
    embedding table with the language model's input convention (language_model.py:28: zero_pad, scale by sqrt(C),
    transformer.py:47-53) instead of the dense(relu) + LayerNorm of embedding_input over pre-net features."""
    rng = np.random.default_rng(seed)
    P = {'enc_pe': _glorot(rng, (pos_max, C)), 'dec_pe': _glorot(rng, (pos_max, C)), 'dec_input': _glorot(rng, (vout, C)),
         'out_w': _glorot(rng, (C, vout)), 'out_b': np.zeros(vout)}
    if vin is None:
        P.update({'in_w': _glorot(rng, (din, C)), 'in_b': np.zeros(C), 'in_ln_g': np.ones(C), 'in_ln_b': np.zeros(C)})
    else:
        P['enc_emb'] = _glorot(rng, (vin, C))
    if perturb:
        P['out_b'] = 0.1 * rng.standard_normal(vout)
        if vin is None:
            P['in_b'] = 0.1 * rng.standard_normal(C)
            P['in_ln_g'] = 1 + 0.1 * rng.standard_normal(C); P['in_ln_b'] = 0.1 * rng.standard_normal(C)
    for i in range(blocks):
        P['enc%d' % i] = init_mha(rng, C, perturb)
        d = init_mha(rng, C, perturb)
        if tie:      # decoder block i re-uses encoder block i's dense kernels; LN params stay distinct
            for k in ('wq', 'wk', 'wv', 'wo'):
                d[k] = P['enc%d' % i][k]
        P['dec%d' % i] = d
    P['enc_ffn'] = init_ffn(rng, C, 4 * C, perturb)
    P['dec_ffn'] = init_ffn(rng, C, 4 * C, perturb)
    if tie:
        for k in ('w1', 'b1', 'w2', 'b2'):
            P['dec_ffn'][k] = P['enc_ffn'][k]
    return P


def e2e_step(P, x_feat, y_in, y_tgt, heads, blocks, tie=True, want_grads=True, drop=None):
    """x_feat [N,T,Din] (the flattened pre_net output fed to embedding_input), y_in/y_tgt [N,L]."""
    ids_in = 'enc_emb' in P                       # x_feat is then [N,T] ids (init_e2e(vin=...))
    if ids_in:
        x_ids = np.asarray(x_feat)
        N, T = x_ids.shape
        C = P['enc_emb'].shape[1]
    else:
        N, T, Din = x_feat.shape
        C = P['in_w'].shape[1]
    L = y_in.shape[1]
    posx = np.broadcast_to(np.arange(T)[None, :], (N, T))
    if ids_in:
        enc = embedding(P['enc_emb'], x_ids, True, True) + embedding(P['enc_pe'], posx, False, False)
    else:
        u = np.maximum(x_feat @ P['in_w'] + P['in_b'], 0)
        iv, c_inln = layer_norm(u, P['in_ln_g'], P['in_ln_b'])
        enc = iv + embedding(P['enc_pe'], posx, False, False)
    posy = np.broadcast_to(np.arange(L)[None, :], (N, L))
    dec = embedding(P['dec_input'], y_in, False, False) + embedding(P['dec_pe'], posy, False, False)
    # model.py:290 drops the encoder input only (the decoder input goes into its blocks as it is, :312-329)
    m_enc = drop.mask(enc.shape, 'emb_enc') if drop is not None else None
    if m_enc is not None:
        enc = enc * m_enc
    ce = []
    for i in range(blocks):
        enc, c = mha_fwd(enc, enc, P['enc%d' % i], heads, causal=False, drop=drop, site=('enc', i))
        ce.append(c)
    memory, c_effn = ffn_fwd(enc, P['enc_ffn'], drop, 'enc_ffn')
    cd = []
    for i in range(blocks):
        dec, c = mha_fwd(dec, memory, P['dec%d' % i], heads, causal=True, drop=drop, site=('dec', i))
        cd.append(c)
    outputs, c_dffn = ffn_fwd(dec, P['dec_ffn'], drop, 'dec_ffn')
    logits = outputs @ P['out_w'] + P['out_b']
    mean_loss, acc, preds, loss, dlogits = smoothed_ce(logits, y_tgt)
    out = {'logits': logits, 'mean_loss': mean_loss, 'acc': acc, 'preds': preds, 'memory': memory}
    if not want_grads:
        return out
    G = {'out_w': outputs.reshape(-1, C).T @ dlogits.reshape(N * L, -1), 'out_b': dlogits.reshape(N * L, -1).sum(axis=0)}
    d = dlogits @ P['out_w'].T
    d, G['dec_ffn'] = ffn_bwd(c_dffn, P['dec_ffn'], d)
    dmem = np.zeros_like(memory)
    for i in reversed(range(blocks)):
        d, dk, G['dec%d' % i] = mha_bwd(cd[i], P['dec%d' % i], d, self_attn=False)
        dmem += dk
    G['dec_input'] = embedding_bwd(P['dec_input'].shape, y_in, d, False, False)
    G['dec_pe'] = embedding_bwd(P['dec_pe'].shape, posy, d, False, False)
    de, G['enc_ffn'] = ffn_bwd(c_effn, P['enc_ffn'], dmem)
    for i in reversed(range(blocks)):
        de, _, G['enc%d' % i] = mha_bwd(ce[i], P['enc%d' % i], de, self_attn=True)
    if m_enc is not None:
        de = de * m_enc
    G['enc_pe'] = embedding_bwd(P['enc_pe'].shape, posx, de, False, False)
    if ids_in:
        G['enc_emb'] = embedding_bwd(P['enc_emb'].shape, x_ids, de, True, True)
    else:
        du, G['in_ln_g'], G['in_ln_b'] = layer_norm_bwd(c_inln, P['in_ln_g'], de)
        du = du * (u > 0)
        G['in_w'] = x_feat.reshape(-1, Din).T @ du.reshape(-1, C)
        G['in_b'] = du.reshape(-1, C).sum(axis=0)
        out['dx_feat'] = du @ P['in_w'].T            # dL/d(x_feat): what the pre-net (oracle/prenet.py) backpropagates
    if tie:          # shared tensors receive the sum of both uses
        for i in range(blocks):
            for k in ('wq', 'wk', 'wv', 'wo'):
                s = G['enc%d' % i][k] + G['dec%d' % i][k]
                G['enc%d' % i][k] = s; G['dec%d' % i][k] = s
        for k in ('w1', 'b1', 'w2', 'b2'):
            s = G['enc_ffn'][k] + G['dec_ffn'][k]
            G['enc_ffn'][k] = s; G['dec_ffn'][k] = s
    out['grads'] = G
    return out

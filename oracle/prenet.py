"""CPU restatement of the end2end pre-net (end2end/model.py:214-264, dot_product_attention :134-172)
in torch float64 with autograd as the gradient oracle.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
PARITY UNPINNED: the reference holds no fixtures for this path and TensorFlow is not installed.

Graph as the reference wires it (is_training=True):
  x [B, T, F] -> expand_dims(-1)                                                 model.py:217
  x1 = BN(tanh(conv3x3 s2 SAME (1 -> 64)));  x2 = BN(tanh(conv3x3 s2 SAME (64 -> 64)))   :219-222
  the `for i in range(2)` loop (:225) reads `input_x2` in BOTH iterations and only the second one's
  result reaches `self.pre_out`: iteration 0 is dead code, one block is live:
    q, k, v = BN(conv3x3(x2)) each (no activation)                               :227-232
    time attention: per (batch, channel) softmax(Q K^T / sqrt(F')) V over [T', F'] matrices    :234-255
    freq attention: the same on the transposed [F', T'] matrices (scale 1/sqrt(T'))            :242-256
    (`mask=False` is "not None", so `False * -1e9 = 0` is added: no masking)    :163-164
    out = layer_norm(conv3x3(concat[time, freq]) + x2)       (eps 1e-8, over the 64 channels) :258-261
    f1 = BN(relu(conv3x3(out)));  f2 = BN(conv3x3(f1));  pre_out = relu(f2 + out)              :263-267
  pre_out [B, T', F', 64] is flattened to [B, T', F'*64] by embedding_input (:268-272).
tf.layers.batch_normalization(training=True): batch moments (biased variance), eps 1e-3, gamma/beta trainable.
tf.layers.conv2d 'same' with stride 2 pads only at the bottom/right for even sizes (TF SAME rule).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-3
LN_EPS = 1e-8
CH = 64

CONVS = [('conv1', 1, CH), ('conv2', CH, CH), ('q', CH, CH), ('k', CH, CH), ('v', CH, CH), ('merge', 2 * CH, CH),
         ('f1', CH, CH), ('f2', CH, CH)]
BNS = ['bn1', 'bn2', 'bnq', 'bnk', 'bnv', 'bnf1', 'bnf2']


def init_params(seed=0, scale=1.0):
    """glorot_normal kernels (model.py: kernel_initializer='glorot_normal'), zero biases, BN/LN gamma 1 beta 0,
    then a small perturbation of every gamma/beta/bias so no gradient path is trivially symmetric."""
    rng = np.random.default_rng(seed)
    P = {}
    for name, cin, cout in CONVS:
        std = math.sqrt(2.0 / (9 * cin + 9 * cout))
        P[name + '/w'] = rng.standard_normal((3, 3, cin, cout)) * std * scale          # HWIO
        P[name + '/b'] = 0.05 * rng.standard_normal(cout)
    for name in BNS + ['ln']:
        P[name + '/g'] = 1.0 + 0.1 * rng.standard_normal(CH)
        P[name + '/b'] = 0.1 * rng.standard_normal(CH)
    return P


def same_pad(n, k, s):
    out = -(-n // s)
    total = max((out - 1) * s + k - n, 0)
    return total // 2, total - total // 2


def conv(x, w, b, stride=1):
    """x NHWC, w HWIO; TF 'same' padding."""
    xn = x.permute(0, 3, 1, 2)
    pt, pb = same_pad(x.shape[1], 3, stride)
    pl, pr = same_pad(x.shape[2], 3, stride)
    xn = F.pad(xn, (pl, pr, pt, pb))
    y = F.conv2d(xn, w.permute(3, 2, 0, 1), b, stride=stride)
    return y.permute(0, 2, 3, 1)


def batch_norm(x, g, b):
    mu = x.mean(dim=(0, 1, 2), keepdim=True)
    var = ((x - mu) ** 2).mean(dim=(0, 1, 2), keepdim=True)
    return g * (x - mu) / torch.sqrt(var + BN_EPS) + b


def layer_norm(x, g, b):
    mu = x.mean(dim=-1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=-1, keepdim=True)
    return g * (x - mu) / torch.sqrt(var + LN_EPS) + b


def attention(q, k, v):
    """dot_product_attention on [..., Lq, d] without a mask."""
    s = q @ k.transpose(-1, -2) / math.sqrt(k.shape[-1])
    return torch.softmax(s, dim=-1) @ v


def forward(P, x, return_all=False):
    """P: dict of torch tensors; x [B, T, F] -> pre_out [B, T/4, F/4, 64]."""
    t = {}
    h = x.unsqueeze(-1)
    t['a1'] = torch.tanh(conv(h, P['conv1/w'], P['conv1/b'], 2))
    t['x1'] = batch_norm(t['a1'], P['bn1/g'], P['bn1/b'])
    t['a2'] = torch.tanh(conv(t['x1'], P['conv2/w'], P['conv2/b'], 2))
    t['x2'] = batch_norm(t['a2'], P['bn2/g'], P['bn2/b'])
    x2 = t['x2']
    q = batch_norm(conv(x2, P['q/w'], P['q/b']), P['bnq/g'], P['bnq/b'])
    k = batch_norm(conv(x2, P['k/w'], P['k/b']), P['bnk/g'], P['bnk/b'])
    v = batch_norm(conv(x2, P['v/w'], P['v/b']), P['bnv/g'], P['bnv/b'])
    t['q'], t['k'], t['v'] = q, k, v
    qt, kt, vt = [z.permute(0, 3, 1, 2) for z in (q, k, v)]          # [B, c, T', F']
    qf, kf, vf = [z.permute(0, 3, 2, 1) for z in (q, k, v)]          # [B, c, F', T']
    at = attention(qt, kt, vt).permute(0, 2, 3, 1)                  # -> [B, T', F', c]
    af = attention(qf, kf, vf).permute(0, 3, 2, 1)
    t['att_time'], t['att_freq'] = at, af
    cat = torch.cat([at, af], dim=-1)
    t['out'] = layer_norm(conv(cat, P['merge/w'], P['merge/b']) + x2, P['ln/g'], P['ln/b'])
    t['f1'] = batch_norm(torch.relu(conv(t['out'], P['f1/w'], P['f1/b'])), P['bnf1/g'], P['bnf1/b'])
    t['f2'] = batch_norm(conv(t['f1'], P['f2/w'], P['f2/b']), P['bnf2/g'], P['bnf2/b'])
    t['pre_out'] = torch.relu(t['f2'] + t['out'])
    return (t['pre_out'], t) if return_all else t['pre_out']


def to_torch(P, requires_grad=True):
    return {k: torch.tensor(np.asarray(v), dtype=torch.float64, requires_grad=requires_grad) for k, v in P.items()}


def forward_backward(P_np, x_np, dout_np):
    """Returns (pre_out, grads dict, intermediates) as numpy float64; dout = dL/d(pre_out)."""
    P = to_torch(P_np)
    x = torch.tensor(x_np, dtype=torch.float64)
    out, t = forward(P, x, return_all=True)
    out.backward(torch.tensor(dout_np, dtype=torch.float64))
    grads = {k: v.grad.numpy() for k, v in P.items()}
    return out.detach().numpy(), grads, {k: v.detach().numpy() for k, v in t.items()}

"""CPU restatement of the Keras DFCNN of lm_and_am/model/cnn_ctc.py (torch float64, autograd for the gradients;
the CTC loss and its gradient come from oracle/ctc.py).  TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (Keras 2.3.1 /
TensorFlow are not installed; the reference has no fixtures).

Graph (cnn_ctc.py:27-49, 96-131):
  cnn_cell(size, x, pool) = norm(conv3x3_relu(size)(x)) -> norm(conv3x3_relu(size)(.)) -> [MaxPooling2D(2,2)]
  h1..h5 = cells of 32, 64, 128 (pooled) and 128, 128 (not pooled);  Reshape(-1, (F/8)*128)
  Dropout(0.3) -> dense(128, relu) -> Dropout(0.3) -> dense(vocab, softmax)
  loss = K.ctc_batch_cost(labels, y_pred, input_length, label_length)  (:51-59,149-152), i.e. tf.nn.ctc_loss on
  log(transpose(y_pred) + 1e-7) with blank = vocab - 1 and the first label_length ids of each row as the label
  (zeros are NOT dropped here, unlike tf.contrib.layers.dense_to_sparse in acoustic_model*.py), mean over the batch
  by Keras' loss reduction.
norm = BatchNormalization(axis=-1): batch moments while fitting (biased variance, epsilon 1e-3), gamma/beta trainable.
Dropout(0.3) in front of both dense layers (:38,40) is stochastic under fit(); Keras' random stream cannot be reproduced,
the build's counter-based mask (asr_dropout, restated in oracle/transformer.py) stands in: sites 0 (h6) and 1 (h7),
seed = base + 1009 * step + 7919 * site.  rate 0 = identity.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import ctc as octc
from .transformer import drop_scale_mask

BN_EPS = 1e-3
K_EPSILON = 1e-7
CELLS = [(32, True), (64, True), (128, True), (128, False), (128, False)]


def layer_names(cells=CELLS):
    return ['c%d%s' % (i + 1, j) for i in range(len(cells)) for j in 'ab']


def init_params(vocab, feat=200, cells=CELLS, hidden=128, seed=0, perturb=True):
    """he_normal kernels, zero biases, BN gamma 1 / beta 0 (+ a perturbation of gammas / betas / biases so that no
    gradient path is trivially symmetric)."""
    rng = np.random.default_rng(seed)
    P, cin = {}, 1
    for i, (size, pool) in enumerate(cells):
        for j in 'ab':
            n = 'c%d%s' % (i + 1, j)
            P[n + '/w'] = rng.standard_normal((3, 3, cin, size)) * math.sqrt(2.0 / (9 * cin))
            P[n + '/b'] = 0.05 * rng.standard_normal(size) if perturb else np.zeros(size)
            P[n + '/g'] = 1 + (0.1 * rng.standard_normal(size) if perturb else 0)
            P[n + '/be'] = 0.1 * rng.standard_normal(size) if perturb else np.zeros(size)
            cin = size
    npool = sum(1 for _, p in cells if p)
    din = (feat >> npool) * cin
    P['d1/w'] = rng.standard_normal((din, hidden)) * math.sqrt(2.0 / din)
    P['d1/b'] = 0.05 * rng.standard_normal(hidden) if perturb else np.zeros(hidden)
    P['d2/w'] = rng.standard_normal((hidden, vocab)) * math.sqrt(2.0 / hidden)
    P['d2/b'] = 0.05 * rng.standard_normal(vocab) if perturb else np.zeros(vocab)
    return P


def batch_norm(x, g, b):
    mu = x.mean(dim=(0, 1, 2), keepdim=True)
    var = ((x - mu) ** 2).mean(dim=(0, 1, 2), keepdim=True)
    return g * (x - mu) / torch.sqrt(var + BN_EPS) + b


def batch_norm_inference(x, g, b, mov_mean, mov_var):
    """Keras learning phase 0 (Model.predict, cnn_ctc.py:82): K.batch_normalization on the moving statistics."""
    return g * (x - mov_mean) / torch.sqrt(mov_var + BN_EPS) + b


def moving_update(mov_mean, mov_var, x, momentum=0.99):
    """One fit() step of a BatchNormalization's moving statistics (Keras 2.3.1 layers/normalization.py, call()): batch mean and
    the batch variance made unbiased by sample_size / (sample_size - (1 + epsilon)), then
    K.moving_average_update: variable -= (variable - value) * (1 - momentum)."""
    mu = x.mean(dim=(0, 1, 2))
    var = ((x - mu) ** 2).mean(dim=(0, 1, 2))
    n = float(x.shape[0] * x.shape[1] * x.shape[2])
    var = var * (n / (n - (1.0 + BN_EPS)))
    return mov_mean - (mov_mean - mu) * (1 - momentum), mov_var - (mov_var - var) * (1 - momentum)


def conv_relu(x, w, b):
    y = F.conv2d(x.permute(0, 3, 1, 2), w.permute(3, 2, 0, 1), b, padding=1)
    return torch.relu(y).permute(0, 2, 3, 1)


def drop_seed(base, step, site):
    return (int(base) + 1009 * int(step) + 7919 * int(site)) & 0xFFFFFFFF


def forward(P, x, cells=CELLS, drop=None, moving=None, collect=None):
    """x [B, T, F] -> softmax outputs y_pred [B, T/8, vocab] (torch, differentiable).  drop = (rate, base_seed, step).
    ``moving`` = {layer: (moving_mean, moving_var)}: inference mode (learning phase 0) -- BatchNormalization on those
    statistics, no dropout.  ``collect`` (a dict): filled with the moving statistics after ONE training step from the values in
    ``collect`` itself ({layer: (mean, var)}, default 0 / 1)."""
    h = x.unsqueeze(-1)
    for i, (size, pool) in enumerate(cells):
        for j in 'ab':
            n = 'c%d%s' % (i + 1, j)
            a = conv_relu(h, P[n + '/w'], P[n + '/b'])
            if moving is not None:
                h = batch_norm_inference(a, P[n + '/g'], P[n + '/be'], moving[n][0], moving[n][1])
                continue
            if collect is not None:
                m0, v0 = collect.get(n, (torch.zeros(size, dtype=a.dtype), torch.ones(size, dtype=a.dtype)))
                collect[n] = tuple(t.detach() for t in moving_update(m0, v0, a.detach()))
            h = batch_norm(a, P[n + '/g'], P[n + '/be'])
        if pool:
            h = F.max_pool2d(h.permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1)
    B, H, W, C = h.shape
    h = h.reshape(B, H, W * C)
    if drop is not None:
        h = h * torch.tensor(drop_scale_mask(tuple(h.shape), drop[0], drop_seed(drop[1], drop[2], 0)))
    h = torch.relu(h @ P['d1/w'] + P['d1/b'])
    if drop is not None:
        h = h * torch.tensor(drop_scale_mask(tuple(h.shape), drop[0], drop_seed(drop[1], drop[2], 1)))
    return torch.softmax(h @ P['d2/w'] + P['d2/b'], dim=-1)


def train_step(P_np, x_np, labels, label_len, input_len, cells=CELLS, drop=None):
    """Returns dict(y_pred, logits (time-major log(y_pred+eps)), loss [B], mean_loss, grads)."""
    P = {k: torch.tensor(np.asarray(v), dtype=torch.float64, requires_grad=True) for k, v in P_np.items()}
    y = forward(P, torch.tensor(x_np, dtype=torch.float64), cells, drop)
    logits = torch.log(y.permute(1, 0, 2) + K_EPSILON)
    B, V = y.shape[0], y.shape[2]
    labs = [list(np.asarray(labels[b])[:label_len[b]]) for b in range(B)]
    loss, grad = octc.ctc_loss_and_grad(logits.detach().numpy(), labs, list(input_len), V - 1)
    logits.backward(torch.tensor(grad / B))                      # Keras averages the per-sample losses
    dec, _ = octc.ctc_greedy_decode(logits.detach().numpy(), list(input_len))
    return {'y_pred': y.detach().numpy(), 'logits': logits.detach().numpy(), 'loss': np.asarray(loss).reshape(-1),
            'mean_loss': float(np.mean(loss)), 'decoded': dec, 'grads': {k: v.grad.numpy() for k, v in P.items()}}

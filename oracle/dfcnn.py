"""Oracle for the whole DFCNN(+SE)+CTC training step (float64 numpy).
Test infrastructure only; parity unpinned (see oracle/__init__.py).

Graphs restated from
  'm2'  lm_and_am/model/acoustic_model2.py:37-74   (SE-DFCNN; avg-pool; the model train.py:8 imports)
  'm1'  lm_and_am/model/acoustic_model.py:37-62    (plain DFCNN; max-pool; NiN cell; 6400->128->V head)
  'm3'  lm_and_am/model/acoustic_model3.py:37-67   (SE on the pooled cell itself, no BN inside SE)
  'amlm' lm_and_am/model/am_lm_model.py:56-66      (acoustic half of the joint AM+LM graph; see oracle/amlm.py)
Loss: acoustic_model2.py:76-83; optimiser: acoustic_model2.py:85-91.
"""
import numpy as np

from . import nn, ctc


def graph(model, vocab, widths=None, feat=200):
    """Returns the op list for a model.  ``widths`` = (c1, c2, c3, c4[, c5]) channel
    counts (default: the reference's).  Ops:
      ('cell', src, dst, cin, cout, ksize, pool)
      ('se',   main, branch, dst, C, hidden, use_bn)      dst = main + SE(branch)
      ('dense', src, dst, cin, cout, act)                 on [B,T,W*C]
    """
    if model == 'm2':
        c1, c2, c3, c6 = widths or (32, 64, 128, 256)
        ops = [('cell', 'x', 'h1', 1, c1, 3, 'avg'), ('cell', 'h1', 'h1_1', c1, c1, 3, None),
               ('se', 'h1', 'h1_1', 'h1s', c1, int(c1 / 1), True),
               ('cell', 'h1s', 'h2', c1, c2, 3, 'avg'), ('cell', 'h2', 'h2_1', c2, c2, 3, None),
               ('se', 'h2', 'h2_1', 'h2s', c2, int(c2 / 2), True),
               ('cell', 'h2s', 'h3', c2, c3, 3, 'avg'), ('cell', 'h3', 'h3_1', c3, c3, 3, None),
               ('se', 'h3', 'h3_1', 'h3s', c3, int(c3 / 2), True),
               ('cell', 'h3s', 'h4', c3, c3, 3, None), ('cell', 'h4', 'h4_1', c3, c3, 3, None),
               ('se', 'h4', 'h4_1', 'h4s', c3, int(c3 / 2), True),
               ('cell', 'h4s', 'h5', c3, c3, 3, None), ('cell', 'h5', 'h5_1', c3, c3, 3, None),
               ('se', 'h5', 'h5_1', 'h5s', c3, int(c3 / 2), True),
               ('cell', 'h5s', 'h6', c3, c6, 3, None),
               ('dense', 'h6', 'd', (feat // 8) * c6, vocab, 'softmax')]
    elif model == 'm1':
        c1, c2, c3, c5, nin, hid = widths or (32, 64, 128, 256, 32, 128)
        ops = [('cell', 'x', 'h1', 1, c1, 3, 'max'), ('cell', 'h1', 'h2', c1, c2, 3, 'max'),
               ('cell', 'h2', 'h3', c2, c3, 3, 'max'), ('cell', 'h3', 'h4', c3, c3, 3, None),
               ('cell', 'h4', 'h5a', c3, c5, 3, None), ('cell', 'h5a', 'h5n', c5, nin, 1, None),
               ('cell', 'h5n', 'h5', nin, c5, 3, None),
               ('dense', 'h5', 'h7', (feat // 8) * c5, hid, 'relu'),
               ('dense', 'h7', 'd', hid, vocab, 'softmax')]
    elif model == 'm3':
        c1, c2, c3, c6 = widths or (32, 64, 128, 256)
        ops = [('cell', 'x', 'h1', 1, c1, 3, 'avg'), ('se', 'h1', 'h1', 'h1s', c1, int(c1 / 1), False),
               ('cell', 'h1s', 'h1b', c1, c1, 3, None),
               ('cell', 'h1b', 'h2', c1, c2, 3, 'avg'), ('se', 'h2', 'h2', 'h2s', c2, int(c2 / 2), False),
               ('cell', 'h2s', 'h2b', c2, c2, 3, None),
               ('cell', 'h2b', 'h3', c2, c3, 3, 'avg'), ('se', 'h3', 'h3', 'h3s', c3, int(c3 / 2), False),
               ('cell', 'h3s', 'h3b', c3, c3, 3, None),
               ('cell', 'h3b', 'h6a', c3, c3, 3, None), ('cell', 'h6a', 'h6', c3, c6, 3, None),
               ('dense', 'h6', 'd', (feat // 8) * c6, vocab, 'softmax')]
    elif model == 'amlm':
        # am_lm_model.py:56-66,163-175: cnn_cell(32), cnn_cell(64), then three NiN cells of 128 (3x3 -> 1x1 to 32 -> 3x3),
        # pooled after the first one only; Reshape; dense(128, relu) = h7; dense(V_pinyin, softmax).  BN frozen
        # (tf.layers.batch_normalization without training=True), dropout(…, 0.3) without training=True = identity.
        c1, c2, c3, nin, hid = widths or (32, 64, 128, 32, 128)
        ops = [('cell', 'x', 'h1', 1, c1, 3, 'max'), ('cell', 'h1', 'h2', c1, c2, 3, 'max'),
               ('cell', 'h2', 'h3a', c2, c3, 3, None), ('cell', 'h3a', 'h3n', c3, nin, 1, None), ('cell', 'h3n', 'h3', nin, c3, 3, 'max'),
               ('cell', 'h3', 'h4a', c3, c3, 3, None), ('cell', 'h4a', 'h4n', c3, nin, 1, None), ('cell', 'h4n', 'h4', nin, c3, 3, None),
               ('cell', 'h4', 'h5a', c3, c3, 3, None), ('cell', 'h5a', 'h5n', c3, nin, 1, None), ('cell', 'h5n', 'h5', nin, c3, 3, None),
               ('dense', 'h5', 'h7', (feat // 8) * c3, hid, 'relu'),
               ('dense', 'h7', 'd', hid, vocab, 'softmax')]
    elif model == 'small':
        # BASELINE.json configs[0] "DFCNN-small (32ch, 4 conv blocks) + CTC" (channels (32, 32, 32, 32), 4 conv
        # cells, B = 4): four cnn_cell()s of acoustic_model2.py:126-133 (conv3x3 + bias -> ReLU -> frozen BN -> avg "maxpool"),
        # the first three pooled so that the CTC axis is T/8 as the data loader assumes (data_loader.py:132), then the
        # reshape + dense(V, softmax) head of acoustic_model2.py:62-68
        c1, c2, c3, c4 = widths or (32, 32, 32, 32)
        ops = [('cell', 'x', 'h1', 1, c1, 3, 'avg'), ('cell', 'h1', 'h2', c1, c2, 3, 'avg'), ('cell', 'h2', 'h3', c2, c3, 3, 'avg'),
               ('cell', 'h3', 'h4', c3, c4, 3, None), ('dense', 'h4', 'd', (feat // 8) * c4, vocab, 'softmax')]
    else:
        raise ValueError(model)
    return ops


def glorot(rng, shape, fan_in, fan_out):
    lim = np.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-lim, lim, size=shape)


def init_params(ops, seed=0, perturb=False):
    """tf.layers defaults: Glorot-uniform kernels, zero bias, BN gamma=1 beta=0
    (acoustic_model2.py:35 initializer=None; Appendix A3/A4).  ``perturb``
    randomises biases/gamma/beta too so parity tests exercise every term."""
    rng = np.random.default_rng(seed)
    P = {}
    for op in ops:
        if op[0] == 'cell':
            _, src, dst, cin, cout, k, pool = op
            P[dst] = {'w': glorot(rng, (k, k, cin, cout), k * k * cin, k * k * cout),
                      'b': np.zeros(cout), 'gamma': np.ones(cout), 'beta': np.zeros(cout)}
            if perturb:
                P[dst]['b'] = rng.normal(0, 0.1, cout)
                P[dst]['gamma'] = 1.0 + rng.normal(0, 0.2, cout)
                P[dst]['beta'] = rng.normal(0, 0.1, cout)
        elif op[0] == 'se':
            _, main, br, dst, C, hid, use_bn = op
            P[dst] = {'w1': glorot(rng, (C, hid), C, hid), 'b1': np.zeros(hid),
                      'w2': glorot(rng, (hid, C), hid, C), 'b2': np.zeros(C)}
            if use_bn:
                P[dst]['gamma'] = np.ones(C)
                P[dst]['beta'] = np.zeros(C)
            if perturb:
                P[dst]['b1'] = rng.normal(0, 0.1, hid)
                P[dst]['b2'] = rng.normal(0, 0.1, C)
                if use_bn:
                    P[dst]['gamma'] = 1.0 + rng.normal(0, 0.2, C)
                    P[dst]['beta'] = rng.normal(0, 0.1, C)
        elif op[0] == 'dense':
            _, src, dst, cin, cout, act = op
            P[dst] = {'w': glorot(rng, (cin, cout), cin, cout), 'b': np.zeros(cout)}
            if perturb:
                P[dst]['b'] = rng.normal(0, 0.1, cout)
    return P


def forward(ops, P, x):
    """x [B,T,F,1] -> (dense pre-activation d [B,T/8,V], caches)."""
    acts = {'x': np.asarray(x, dtype=np.float64)}
    caches = {}
    for op in ops:
        if op[0] == 'cell':
            _, src, dst, cin, cout, k, pool = op
            acts[dst], caches[dst] = nn.cell_fwd(acts[src], P[dst], pool)
        elif op[0] == 'se':
            _, main, br, dst, C, hid, use_bn = op
            p = dict(P[dst])
            if not use_bn:
                p['gamma'] = np.full(C, np.sqrt(1.0 + nn.BN_EPS))   # identity affine
                p['beta'] = np.zeros(C)
            s, caches[dst] = nn.se_fwd(acts[br], p)
            acts[dst] = acts[main] + s
        elif op[0] == 'dense':
            _, src, dst, cin, cout, act = op
            h = acts[src]
            if h.ndim == 4:
                h = h.reshape(h.shape[0], h.shape[1], -1)
            z = nn.dense_fwd(h, P[dst]['w'], P[dst]['b'])
            caches[dst] = (h, z)
            acts[dst] = np.maximum(z, 0.0) if act == 'relu' else z   # softmax applied by the loss head
    return acts[ops[-1][2]], (acts, caches)


def backward(ops, P, state, dd, extra=None):
    """dd = dL/d(dense pre-activation of the last layer); extra = {activation name: additional dL/d(activation)} (the
    joint AM+LM graph feeds h7 to a second branch).  Returns grads dict."""
    acts, caches = state
    G = {}
    dacts = {ops[-1][2]: dd}
    for k, v in (extra or {}).items():
        dacts[k] = dacts.get(k, 0) + np.asarray(v, dtype=np.float64).reshape(acts[k].shape)
    for op in reversed(ops):
        if op[0] == 'dense':
            _, src, dst, cin, cout, act = op
            h, z = caches[dst]
            dz = dacts.pop(dst)
            if act == 'relu':
                dz = dz * (z > 0)
            dh, dw, db = nn.dense_bwd(h, P[dst]['w'], dz)
            G[dst] = {'w': dw, 'b': db}
            dh = dh.reshape(acts[src].shape)
            dacts[src] = dacts.get(src, 0) + dh
        elif op[0] == 'se':
            _, main, br, dst, C, hid, use_bn = op
            dout = dacts.pop(dst)
            p = dict(P[dst])
            if not use_bn:
                p['gamma'] = np.full(C, np.sqrt(1.0 + nn.BN_EPS))
                p['beta'] = np.zeros(C)
            dbr, g = nn.se_bwd(caches[dst], p, dout)
            if not use_bn:
                g.pop('gamma'), g.pop('beta')
            G[dst] = g
            dacts[main] = dacts.get(main, 0) + dout
            dacts[br] = dacts.get(br, 0) + dbr
        elif op[0] == 'cell':
            _, src, dst, cin, cout, k, pool = op
            dout = dacts.pop(dst)
            dx, g = nn.cell_bwd(caches[dst], P[dst], dout, pool)
            G[dst] = g
            if src != 'x':
                dacts[src] = dacts.get(src, 0) + dx
    return G


def train_step_oracle(ops, P, x, logits_length, target_py, want_grads=True):
    """One forward(+backward) of the reference graph.  Returns dict with
    logits_tm [T,B,V] (= self.logits), loss [B,1], mean_loss, decoded, label_err, grads."""
    d, state = forward(ops, P, x)
    logits_tm = nn.log_softmax_eps_tm(d)
    labels = ctc.dense_to_sparse(target_py)
    V = d.shape[-1]
    loss, g_tm = ctc.ctc_loss_and_grad(logits_tm, labels, logits_length, blank=V - 1)
    decoded, neg = ctc.ctc_greedy_decode(logits_tm, logits_length)
    out = {'d': d, 'logits': logits_tm, 'loss': loss[:, None], 'mean_loss': float(loss.mean()),
           'decoded': decoded, 'neg_sum_logits': neg,
           'label_err': ctc.label_error_rate(decoded, labels)}
    if want_grads:
        B = d.shape[0]
        dd = nn.log_softmax_eps_tm_bwd(d, g_tm / B)       # mean over the batch
        out['dd'] = dd
        out['grads'] = backward(ops, P, state, dd)
        out['acts'] = state[0]
    return out

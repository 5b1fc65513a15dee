"""The original Keras DFCNN (lm_and_am/model/cnn_ctc.py) on the libasrhip kernels: a configuration permutation of
the acoustic path -- two 3x3 ReLU convs per cell, each followed by a BatchNormalization that runs on
BATCH statistics while fitting, 2x2 max-pooling after the first three cells, Reshape -> dense(128, relu) ->
dense(vocab, softmax), K.ctc_batch_cost, Keras Adam(beta_2 0.999, epsilon 1e-7).

    cnn_cell   cnn_ctc.py:124-131     _model_init :27-49     ctc_lambda :149-152     opt_init :62-65

Kernels: 9-tap tap_gemm / tap_wgrad for every conv (the first one reads a 4-channel plane whose channels 1..3 are
zero: K must be a multiple of 4), bn_stats / bn_apply / bn_bwd of the pre-net (batch moments, ReLU derivative fused
into the backward), pool_fwd + maxpool_bwd, the M1 head kernels for dense / log-softmax / CTC / greedy / edit
distance.  Dropout(0.3) in front of both dense layers (cnn_ctc.py:38,40): `dropout_rate` > 0 applies the counter-based
mask of asr_dropout (Keras' own random stream cannot be reproduced); 0 = identity, which the parity runs use.
"""
import math

import numpy as np
import torch

from . import ops
from .ops import Plane

BN_EPS = 1e-3            # keras.layers.BatchNormalization default epsilon
BN_MOMENTUM = 0.99       # ... and momentum
K_EPSILON = 1e-7         # keras.backend.epsilon()
MAX_LABEL = 64
CELLS = [(32, True), (64, True), (128, True), (128, False), (128, False)]


class KerasDFCNNEngine:
    def __init__(self, vocab=1424, B=4, T=1600, F=200, cells=CELLS, hidden=128, lr=8e-4, seed=0, device='cuda',
                 dropout_rate=0.0, drop_seed=0):
        npool = sum(1 for _, p in cells if p)
        assert T % (1 << npool) == 0 and F % (1 << npool) == 0
        self.V, self.B, self.T, self.F, self.cells, self.hidden, self.device = vocab, B, T, F, list(cells), hidden, device
        self.lr, self.beta1, self.beta2, self.adam_eps, self.global_step = lr, 0.9, 0.999, 1e-7, 0
        self.dropout_rate, self.drop_seed = float(dropout_rate), int(drop_seed)
        # ---- parameters: name -> (offset, physical shape); conv 1 is stored with 4 input channels (3 dead)
        self.entries, self.logical, off = {}, {}, 0

        def add(name, shape, phys=None):
            nonlocal off
            phys = tuple(phys or shape)
            self.entries[name], self.logical[name] = (off, phys), tuple(shape)
            off += (int(np.prod(phys)) + 3) // 4 * 4

        self.convs = []          # (name, cin_phys, cout, H, W, pool_after)
        cin, H, W = 1, T, F
        for i, (size, pool) in enumerate(cells):
            for j in 'ab':
                n = 'c%d%s' % (i + 1, j)
                cp = max(cin, 4)
                add(n + '/w', (3, 3, cin, size), (3, 3, cp, size)); add(n + '/b', (size,))
                add(n + '/g', (size,)); add(n + '/be', (size,))
                # moving_mean / moving_variance of the BatchNormalization (non-trainable: their gradient stays 0, so Adam
                # never moves them; living in theta they are part of every checkpoint like the Keras weights file)
                add(n + '/mm', (size,)); add(n + '/mv', (size,))
                self.convs.append((n, cp, size, H, W, pool and j == 'b'))
                cin = size
            if pool:
                H, W = H // 2, W // 2
        self.T8, self.W8, self.Clast = H, W, cin
        self.din = W * cin
        add('d1/w', (self.din, hidden)); add('d1/b', (hidden,))
        add('d2/w', (hidden, vocab)); add('d2/b', (vocab,))
        z = lambda n, dt=torch.float32: torch.zeros(n, dtype=dt, device=device)
        self.theta, self.grad, self.adam_m, self.adam_v = z(off), z(off), z(off), z(off)
        self.init_params(seed)

        # ---- buffers
        self.x4 = Plane(B, T, F, 4, device)
        self.a, self.y, self.yp, self.stats = {}, {}, {}, {}
        self.dz, self.dplane = {}, {}
        self.fdesc, self.bdesc, self.wdesc = {}, {}, {}
        self.pw = True     # 3x3 convs on weights in MFMA fragment order (asr_tap_gemm_pw)
        self.wf_f, self.wf_b, self._dims = {}, {}, {}
        ws = 1 << 20
        for idx, (n, cp, cout, H, W, pool_after) in enumerate(self.convs):
            last = idx == len(self.convs) - 1
            self.a[n] = Plane(B, H, W, cout, device)
            self.stats[n] = (z(cout), z(cout))
            if last:
                assert not pool_after
                self.y[n] = z(B * H * W * cout).view(B, H, W, cout)        # feeds the Reshape: plain NHWC
            else:
                self.y[n] = Plane(B, H, W, cout, device)
                if pool_after:
                    self.yp[n] = Plane(B, H // 2, W // 2, cout, device)
            NP = self.a[n].NP
            self.fdesc[n] = ops.gemm_desc(NP, cp, cout, cp, cout, cout, 0, ntaps=9, B=B, H=H, W=W, relu=1)
            self.bdesc[n] = ops.gemm_desc(NP, cout, cp, cout, cout, 0, cp, ntaps=9, B=B, H=H, W=W, wmode=1)
            self.wdesc[n] = ops.gemm_desc(NP, cp, cout, cp, cout, ntaps=9, B=B, H=H, W=W)
            geo = (H, W, cout)
            if geo not in self.dz:
                self.dz[geo] = Plane(B, H, W, cout, device)
                self.dplane[geo] = Plane(B, H, W, cout, device)           # d(BN output) at this geometry
            ws = max(ws, ops.tap_wgrad_workspace(self.wdesc[n]), ops.bn_workspace(self.a[n]), ops.colsum_workspace(NP, cout))
            if self.pw:        # forward and data-gradient views of the weights in MFMA fragment order (asr_tap_gemm_pw)
                nb = ops._lib.load().asr_arrange_weights_bytes
                self.wf_f[n] = z(nb(9, cp, cout) // 4)
                self.wf_b[n] = z(nb(9, cout, cp) // 4)
                self._dims[n] = (cp, cout)
        rows = B * self.T8
        self.h7, self.d = z(rows * hidden).view(rows, hidden), z(rows * vocab).view(rows, vocab)
        self.dh6, self.dh7, self.dd = z(rows * self.din).view(rows, self.din), z(rows * hidden).view(rows, hidden), z(rows * vocab).view(rows, vocab)
        self.f1 = ops.gemm_desc(rows, self.din, hidden, self.din, hidden, 0, hidden, ntaps=1, relu=1)
        self.f2 = ops.gemm_desc(rows, hidden, vocab, hidden, vocab, 0, vocab, ntaps=1)
        self.b1 = ops.gemm_desc(rows, hidden, self.din, hidden, hidden, 0, self.din, ntaps=1, wmode=1)
        self.b2 = ops.gemm_desc(rows, vocab, hidden, vocab, vocab, 0, hidden, ntaps=1, wmode=1)
        self.w1 = ops.gemm_desc(rows, self.din, hidden, self.din, hidden, ntaps=1)
        self.w2 = ops.gemm_desc(rows, hidden, vocab, hidden, vocab, ntaps=1)
        ws = max(ws, ops.tap_wgrad_workspace(self.w1), ops.tap_wgrad_workspace(self.w2), ops.colsum_workspace(rows, vocab))
        self.ws = z(ws // 4 + 64)
        T8, V = self.T8, vocab
        self.logits, self.ctc_grad = z(T8 * B * V).view(T8, B, V), z(T8 * B * V).view(T8, B, V)
        self.loss, self.ctc_status = z(B), z(B, torch.int32)
        self.ctc_ws = z(ops.ctc_workspace(T8, B, MAX_LABEL) // 8 + 8, torch.float64)
        self.dec_ids, self.dec_len = z(B * T8, torch.int32).view(B, T8), z(B, torch.int32)
        self.dec_ws = z(ops.ctc_greedy_workspace(T8, B) // 4 + 4, torch.int32)
        self.neg_sum, self.dist, self.scalars = z(B), z(B), z(8)
        self.labels, self.label_len, self.seq_len = z(B * MAX_LABEL, torch.int32).view(B, MAX_LABEL), z(B, torch.int32), z(B, torch.int32)
        cmax = max(c for c, _ in cells)
        self.ones, self.zeros = torch.ones(cmax, device=device), z(cmax)

    # ---- parameters
    def p(self, name, buf=None):
        off, shape = self.entries[name]
        return (self.theta if buf is None else buf)[off:off + int(np.prod(shape))]

    def g(self, name):
        return self.p(name, self.grad)

    def init_params(self, seed=0):
        """kernel_initializer='he_normal' (cnn_ctc.py:96-117), zero biases, BN gamma 1 / beta 0."""
        rng = np.random.default_rng(seed)
        flat = {}
        for name, shp in self.logical.items():
            if name.endswith('/w'):
                fan_in = int(np.prod(shp[:-1]))
                flat[name] = rng.standard_normal(shp) * math.sqrt(2.0 / fan_in)
            elif name.endswith('/g') or name.endswith('/mv'):
                flat[name] = np.ones(shp)
            else:
                flat[name] = np.zeros(shp)
        self.load_params(flat)

    def load_params(self, flat):
        """flat: {name: ndarray} in logical shapes; moving statistics that are not given start at Keras' initial values
        (moving_mean 0, moving_variance 1)."""
        host = self.theta.cpu().numpy()
        for name, (off, phys) in self.entries.items():
            if name not in flat and name.endswith(('/mm', '/mv')):
                flat = dict(flat); flat[name] = np.zeros(self.logical[name]) if name.endswith('/mm') else np.ones(self.logical[name])
            v = np.asarray(flat[name], dtype=np.float32)
            assert tuple(v.shape) == self.logical[name], (name, v.shape, self.logical[name])
            buf = np.zeros(phys, dtype=np.float32)
            buf[tuple(slice(0, s) for s in v.shape)] = v
            host[off:off + buf.size] = buf.ravel()
        self.theta.copy_(torch.from_numpy(host))

    def grads_dict(self, buf=None):
        host = (self.grad if buf is None else buf).cpu().numpy()
        out = {}
        for name, (off, phys) in self.entries.items():
            a = host[off:off + int(np.prod(phys))].reshape(phys)
            out[name] = a[tuple(slice(0, s) for s in self.logical[name])].copy()
        return out

    def params_dict(self):
        return self.grads_dict(self.theta)

    # ---- step
    def _seed(self, site):
        return (self.drop_seed + 1009 * self.global_step + 7919 * site) & 0xFFFFFFFF

    def forward(self, x, train=True):
        """x: [B, T, F] float32 on the device -> time-major log(softmax + 1e-7) logits [T/8, B, vocab].
        train=True is Keras' learning phase 1 (ctc_model.fit / train_on_batch): BatchNormalization on BATCH moments, the
        moving statistics take a momentum-0.99 step, Dropout live.  train=False is phase 0 (Model.predict, cnn_ctc.py:82;
        evaluate): BatchNormalization on the MOVING statistics, Dropout off."""
        assert tuple(x.shape) == (self.B, self.T, self.F)
        self._rate = self.dropout_rate if train else 0.0
        self._train = bool(train)
        self.x4.interior()[..., 0].copy_(x)
        src = self.x4
        for n, cp, cout, H, W, pool_after in self.convs:
            if self.pw:
                cp_, cout_ = self._dims[n]
                ops.arrange_weights(self.p(n + '/w'), 9, cp_, cout_, cout_, 0, self.wf_f[n])
                ops.arrange_weights(self.p(n + '/w'), 9, cout_, cp_, cout_, 1, self.wf_b[n])
                ops.tap_gemm_pw(self.fdesc[n], src, self.wf_f[n], self.p(n + '/b'), None, None, self.a[n], None)
            else:
                ops.tap_gemm(self.fdesc[n], src, self.p(n + '/w'), self.p(n + '/b'), None, None, self.a[n], None)
            mean, rstd = self.stats[n]
            if train:
                ops.bn_stats(self.a[n], BN_EPS, mean, rstd, self.ws)
                ops.bn_moving(mean, rstd, cout, BN_EPS, self.B * H * W, BN_MOMENTUM, self.p(n + '/mm'), self.p(n + '/mv'))
                ops.bn_apply(self.a[n], mean, rstd, self.p(n + '/g'), self.p(n + '/be'), self.y[n])
            else:
                ops.bn_moving(None, None, cout, BN_EPS, 0, BN_MOMENTUM, self.p(n + '/mm'), self.p(n + '/mv'), rstd)
                ops.bn_apply(self.a[n], self.p(n + '/mm'), rstd, self.p(n + '/g'), self.p(n + '/be'), self.y[n])
            src = self.y[n]
            if pool_after:
                ops.pool_fwd(self.y[n], self.ones[:cout], self.zeros[:cout], 2, self.yp[n])
                src = self.yp[n]
        self.h6 = src.view(self.B * self.T8, self.din)
        self._s6, self._s7 = self._seed(0), self._seed(1)
        if self._rate > 0:
            ops.dropout(self.h6, self._rate, self._s6)          # in place: the last BN output is only read again as dense input
        ops.tap_gemm(self.f1, self.h6, self.p('d1/w'), self.p('d1/b'), None, None, None, self.h7)
        if self._rate > 0:
            ops.dropout(self.h7, self._rate, self._s7)
        ops.tap_gemm(self.f2, self.h7, self.p('d2/w'), self.p('d2/b'), None, None, None, self.d)
        ops.softmax_log_fwd(self.d, self.B, self.T8, self.V, K_EPSILON, self.logits)
        return self.logits

    def set_targets(self, input_length, labels, label_length):
        """K.ctc_batch_cost arguments: the first label_length ids of each row are the label (zeros are kept)."""
        lab = np.zeros((self.B, MAX_LABEL), dtype=np.int32)
        ll = np.asarray(label_length, dtype=np.int32).reshape(self.B)
        sl = np.asarray(input_length, dtype=np.int32).reshape(self.B)
        for b in range(self.B):
            ids = np.asarray(labels[b])[:ll[b]]
            rep = int(np.sum(ids[1:] == ids[:-1]))
            if sl[b] < len(ids) + rep or sl[b] <= 0 or sl[b] > self.T8:
                raise ValueError('Not enough time for target transition sequence (required: %d, available: %d) in batch %d'
                                 % (len(ids) + rep, sl[b], b))
            lab[b, :len(ids)] = ids
        self.labels.copy_(torch.from_numpy(lab)); self.label_len.copy_(torch.from_numpy(ll)); self.seq_len.copy_(torch.from_numpy(sl))

    def loss_and_decode(self):
        B, T8, V = self.B, self.T8, self.V
        ops.ctc_loss(self.logits, T8, B, V, self.labels, MAX_LABEL, self.label_len, self.seq_len, V - 1,
                     self.loss, self.ctc_grad, self.ctc_status, self.ctc_ws)
        ops.ctc_greedy(self.logits, T8, B, V, self.seq_len, V - 1, self.dec_ids, self.dec_len, self.neg_sum, self.dec_ws)
        ops.edit_distance(self.dec_ids, T8, self.dec_len, self.labels, MAX_LABEL, self.label_len, B, self.dist)
        ops.colsum(self.loss, B, 1, 1, self.scalars[0:1], self.ws)

    def backward(self):
        if not getattr(self, '_train', True):
            # an inference forward left the MOVING rstd beside the batch mean of an earlier step in self.stats: the BatchNormalization
            # backward would mix the two without an error
            raise RuntimeError('backward() needs a training forward (forward(x, train=True)); the last forward was an inference pass')
        B, T8, V, rows = self.B, self.T8, self.V, self.B * self.T8
        ops.softmax_log_bwd(self.logits, self.ctc_grad, B, T8, V, K_EPSILON, 1.0 / B, self.dd)
        ops.tap_wgrad(self.w2, self.h7, self.dd, V, self.g('d2/w'), self.ws)
        ops.colsum(self.dd, rows, V, V, self.g('d2/b'), self.ws)
        ops.tap_gemm(self.b2, self.dd, self.p('d2/w'), None, None, None, None, self.dh7)
        ops.relu_bwd(self.dh7, self.h7, self.dh7)
        if self._rate > 0:
            ops.dropout(self.dh7, self._rate, self._s7)
        ops.tap_wgrad(self.w1, self.h6, self.dh7, self.hidden, self.g('d1/w'), self.ws)
        ops.colsum(self.dh7, rows, self.hidden, self.hidden, self.g('d1/b'), self.ws)
        ops.tap_gemm(self.b1, self.dh7, self.p('d1/w'), None, None, None, None, self.dh6)
        if self._rate > 0:
            ops.dropout(self.dh6, self._rate, self._s6)
        dy = self.dh6.view(B, T8, self.W8, self.Clast)                    # d(last BN output), plain NHWC
        for idx in reversed(range(len(self.convs))):
            n, cp, cout, H, W, pool_after = self.convs[idx]
            geo = (H, W, cout)
            if pool_after:                                                 # dy is d(pooled): route it to the arg-max
                ops.maxpool_bwd(dy, self.y[n], self.dplane[geo])
                dy = self.dplane[geo]
            mean, rstd = self.stats[n]
            dz = self.dz[geo]
            ops.bn_bwd(dy, self.a[n], mean, rstd, self.p(n + '/g'), 1, dz, self.g(n + '/g'), self.g(n + '/be'), self.ws)
            src = self.x4 if idx == 0 else (self.yp[self.convs[idx - 1][0]] if self.convs[idx - 1][5] else self.y[self.convs[idx - 1][0]])
            ops.tap_wgrad(self.wdesc[n], src, dz, cout, self.g(n + '/w'), self.ws)
            ops.colsum(dz.body, dz.NP, cout, cout, self.g(n + '/b'), self.ws)
            if idx > 0:
                pgeo = (src.H, src.W, src.C)
                dst = self._dsrc(pgeo, pooled=self.convs[idx - 1][5])
                if self.pw:
                    ops.tap_gemm_pw(self.bdesc[n], dz, self.wf_b[n], None, None, None, None, dst)
                else:
                    ops.tap_gemm(self.bdesc[n], dz, self.p(n + '/w'), None, None, None, None, dst)
                dy = dst

    def _dsrc(self, geo, pooled):
        """gradient plane of a conv input: the shared d-plane of its geometry, or (for a pooled input, whose geometry has no
        conv output of its own before the next cell) a lazily made one."""
        key = ('in',) + geo
        if key not in self.dplane:
            self.dplane[key] = Plane(self.B, geo[0], geo[1], geo[2], self.device)
        return self.dplane[key]

    def apply_adam(self, gscale=1.0):
        t = self.global_step + 1
        lr_t = self.lr * math.sqrt(1.0 - self.beta2 ** t) / (1.0 - self.beta1 ** t)
        ops.adam_tf(self.theta, self.grad, self.adam_m, self.adam_v, lr_t, self.beta1, self.beta2, self.adam_eps, gscale)
        self.global_step += 1

    def fetch_loss(self):
        return float(self.scalars.cpu().numpy()[0]) / self.B

    def decoded_lists(self):
        ids, n = self.dec_ids.cpu().numpy(), self.dec_len.cpu().numpy()
        return [ids[b, :n[b]].tolist() for b in range(self.B)]

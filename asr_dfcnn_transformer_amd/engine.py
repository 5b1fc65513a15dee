"""Host-side engine of the DFCNN(+SE)+CTC acoustic model: owns the flat parameter /
gradient / Adam buffers and the activation planes, and enqueues the libasrhip kernels of
one forward / backward / update on the current HIP stream.  No arithmetic happens here.

Graphs follow the reference model files:
  'm2'  lm_and_am/model/acoustic_model2.py:37-74   SE-DFCNN, "maxpool" = average pool
  'm1'  lm_and_am/model/acoustic_model.py:37-62    plain DFCNN, max pool, NiN cell, 6400->128->V head
  'm3'  lm_and_am/model/acoustic_model3.py:37-67   SE applied to the pooled cell itself, no BN in SE
  'amlm' lm_and_am/model/am_lm_model.py:56-66      acoustic half of the joint AM+LM graph (joint_engine.py)
Every cell is conv -> +bias -> ReLU -> frozen-affine BN -> [pool].
"""
import math

import numpy as np
import torch

from . import ops
from .ops import Plane

BN_EPS = 1e-3                      # tf.layers.batch_normalization default
RS = 1.0 / math.sqrt(1.0 + BN_EPS)  # moving_var = 1, never updated in the reference (Q1)
K_EPSILON = 1e-7                   # keras.backend.epsilon()
MAX_LABEL = 64                     # lm_and_am/data_loader.py:109


def graph(model, vocab, widths=None, feat=200):
    """Op list of a reference model.
      ('cell', src, dst, cin, cout, ksize, pool)    pool in (None, 'avg', 'max')
      ('se',   main, branch, dst, C, hidden, use_bn)   dst = main + SE(branch)
      ('dense', src, dst, cin, cout, act)
    """
    if model == 'm2':
        c1, c2, c3, c6 = widths or (32, 64, 128, 256)
        g = [('cell', 'x', 'h1', 1, c1, 3, 'avg'), ('cell', 'h1', 'h1_1', c1, c1, 3, None),
             ('se', 'h1', 'h1_1', 'h1s', c1, int(c1 / 1), True),
             ('cell', 'h1s', 'h2', c1, c2, 3, 'avg'), ('cell', 'h2', 'h2_1', c2, c2, 3, None),
             ('se', 'h2', 'h2_1', 'h2s', c2, int(c2 / 2), True),
             ('cell', 'h2s', 'h3', c2, c3, 3, 'avg'), ('cell', 'h3', 'h3_1', c3, c3, 3, None),
             ('se', 'h3', 'h3_1', 'h3s', c3, int(c3 / 2), True)]
        prev = 'h3s'
        for i in (4, 5):
            g += [('cell', prev, 'h%d' % i, c3, c3, 3, None), ('cell', 'h%d' % i, 'h%d_1' % i, c3, c3, 3, None),
                  ('se', 'h%d' % i, 'h%d_1' % i, 'h%ds' % i, c3, int(c3 / 2), True)]
            prev = 'h%ds' % i
        g += [('cell', prev, 'h6', c3, c6, 3, None), ('dense', 'h6', 'd', (feat // 8) * c6, vocab, 'softmax')]
    elif model == 'm1':
        c1, c2, c3, c5, nin, hid = widths or (32, 64, 128, 256, 32, 128)
        g = [('cell', 'x', 'h1', 1, c1, 3, 'max'), ('cell', 'h1', 'h2', c1, c2, 3, 'max'),
             ('cell', 'h2', 'h3', c2, c3, 3, 'max'), ('cell', 'h3', 'h4', c3, c3, 3, None),
             ('cell', 'h4', 'h5a', c3, c5, 3, None), ('cell', 'h5a', 'h5n', c5, nin, 1, None),
             ('cell', 'h5n', 'h5', nin, c5, 3, None),
             ('dense', 'h5', 'h7', (feat // 8) * c5, hid, 'relu'), ('dense', 'h7', 'd', hid, vocab, 'softmax')]
    elif model == 'm3':
        c1, c2, c3, c6 = widths or (32, 64, 128, 256)
        g = [('cell', 'x', 'h1', 1, c1, 3, 'avg'), ('se', 'h1', 'h1', 'h1s', c1, int(c1 / 1), False),
             ('cell', 'h1s', 'h1b', c1, c1, 3, None),
             ('cell', 'h1b', 'h2', c1, c2, 3, 'avg'), ('se', 'h2', 'h2', 'h2s', c2, int(c2 / 2), False),
             ('cell', 'h2s', 'h2b', c2, c2, 3, None),
             ('cell', 'h2b', 'h3', c2, c3, 3, 'avg'), ('se', 'h3', 'h3', 'h3s', c3, int(c3 / 2), False),
             ('cell', 'h3s', 'h3b', c3, c3, 3, None),
             ('cell', 'h3b', 'h6a', c3, c3, 3, None), ('cell', 'h6a', 'h6', c3, c6, 3, None),
             ('dense', 'h6', 'd', (feat // 8) * c6, vocab, 'softmax')]
    elif model == 'amlm':
        # acoustic half of the joint graph, am_lm_model.py:56-66,163-175: cnn_cell(32), cnn_cell(64), three NiN cells of
        # 128 (3x3 -> 1x1 to 32 -> 3x3; the first one pooled), dense(128, relu) = h7 (fed to the language half), dense(V)
        c1, c2, c3, nin, hid = widths or (32, 64, 128, 32, 128)
        g = [('cell', 'x', 'h1', 1, c1, 3, 'max'), ('cell', 'h1', 'h2', c1, c2, 3, 'max'),
             ('cell', 'h2', 'h3a', c2, c3, 3, None), ('cell', 'h3a', 'h3n', c3, nin, 1, None), ('cell', 'h3n', 'h3', nin, c3, 3, 'max'),
             ('cell', 'h3', 'h4a', c3, c3, 3, None), ('cell', 'h4a', 'h4n', c3, nin, 1, None), ('cell', 'h4n', 'h4', nin, c3, 3, None),
             ('cell', 'h4', 'h5a', c3, c3, 3, None), ('cell', 'h5a', 'h5n', c3, nin, 1, None), ('cell', 'h5n', 'h5', nin, c3, 3, None),
             ('dense', 'h5', 'h7', (feat // 8) * c3, hid, 'relu'), ('dense', 'h7', 'd', hid, vocab, 'softmax')]
    elif model == 'small':
        # BASELINE.json configs[0] "DFCNN-small (32ch, 4 conv blocks) + CTC" (channels (32, 32, 32, 32), 4 conv
        # cells, B = 4): four cnn_cell()s of acoustic_model2.py:126-133 (conv3x3 + bias -> ReLU -> frozen BN -> avg "maxpool"),
        # the first three pooled so that the CTC axis is T/8 as the data loader assumes (data_loader.py:132), then the
        # reshape + dense(V, softmax) head of acoustic_model2.py:62-68
        c1, c2, c3, c4 = widths or (32, 32, 32, 32)
        g = [('cell', 'x', 'h1', 1, c1, 3, 'avg'), ('cell', 'h1', 'h2', c1, c2, 3, 'avg'), ('cell', 'h2', 'h3', c2, c3, 3, 'avg'),
               ('cell', 'h3', 'h4', c3, c4, 3, None), ('dense', 'h4', 'd', (feat // 8) * c4, vocab, 'softmax')]
    else:
        raise ValueError('unknown model %r' % (model,))
    return g


def fwd_flops_per_utt(g, T, F):
    """2*MACs of every conv/dense layer for one utterance (SURVEY.md 8d convention)."""
    res = {'x': (T, F)}
    total, first = 0.0, 0.0
    for op in g:
        if op[0] == 'cell':
            _, src, dst, cin, cout, k, pool = op
            H, W = res[src]
            f = 2.0 * H * W * k * k * cin * cout
            total += f
            if src == 'x':
                first = f
            res[dst] = (H // 2, W // 2) if pool else (H, W)
        elif op[0] == 'se':
            res[op[3]] = res[op[1]]
        elif op[0] == 'dense':
            _, src, dst, cin, cout, act = op
            H = res[src][0]
            total += 2.0 * H * cin * cout
            res[dst] = (H, 1)
    return total, first


def step_flops_per_utt(g, T, F):
    """F_step = 3*F_fwd - F_c1 (no data gradient for the first conv)."""
    total, first = fwd_flops_per_utt(g, T, F)
    return 3.0 * total - first


def param_layout(g):
    """Flat fp32 parameter layout of a graph: [all BN gammas][dense layers][everything else], every tensor 16-byte
    aligned.  Returns (entries {(layer, key): (offset, shape)}, n_gamma, dense_end, total floats).  The gamma segment
    lets one launch derive every bn_scale; the dense segment is the first gradient bucket that is final in backward
    (its all-reduce overlaps the conv-stack backward); the three data-parallel buckets are
    [n_gamma, dense_end), [0, n_gamma), [dense_end, total)."""
    ent = {}
    off = 0

    def add(layer, key, shape):
        nonlocal off
        n = int(np.prod(shape))
        ent[(layer, key)] = (off, tuple(shape))
        off += (n + 3) // 4 * 4

    for op in g:
        if op[0] == 'cell':
            add(op[2], 'gamma', (op[4],))
        elif op[0] == 'se' and op[6]:
            add(op[3], 'gamma', (op[4],))
    n_gamma = off
    for op in g:
        if op[0] == 'dense':
            add(op[2], 'w', (op[3], op[4]))
            add(op[2], 'b', (op[4],))
    dense_end = off
    for op in g:
        if op[0] == 'cell':
            _, src, dst, cin, cout, k, pool = op
            add(dst, 'w', (k, k, cin, cout)); add(dst, 'b', (cout,)); add(dst, 'beta', (cout,))
        elif op[0] == 'se':
            _, main, br, dst, Cc, hid, use_bn = op
            if use_bn:
                add(dst, 'beta', (Cc,))
            add(dst, 'w1', (Cc, hid)); add(dst, 'b1', (hid,)); add(dst, 'w2', (hid, Cc)); add(dst, 'b2', (Cc,))
    return ent, n_gamma, dense_end, off


class DFCNNEngine:
    def __init__(self, model='m2', vocab=1536, B=32, T=1600, F=200, widths=None, seed=0, device='cuda',
                 lr=7e-4, decay_steps=5000, min_lr=1e-6, beta1=0.9, beta2=0.999, adam_eps=1e-8,
                 dual_stream=True, wino=True, fuse_prologues=True, fuse_se=True, compact_pool=True, side_priority=0,
                 dense_wgrad_side=True, fuse_dense=True, se_sums=True, nt_splitk=True):
        """Options (constructor arguments only -- nothing here reads the environment; `options()` reports them):
        ``dual_stream``: weight gradients / decode on a second stream (``side_priority``: its HIP stream priority);
        ``wino``: Winograd F(2x2,3x3) / F(3x3,2x2) for the 3x3 layers the kernels support instead of the direct tap-GEMM;
        ``fuse_prologues``: cell backward prologues inside the data-gradient epilogues; ``fuse_se``: an SE block's backward also
        runs its branch cell's BN / ReLU backward (asr_se_bwd_cell); ``compact_pool``: pooled cells keep, instead of the pre-pool
        plane, the activation at each window's maximum + its position (max pool: same bits) or the window's activation sum + the ReLU signs
        (average pool: dZ the same bits, the BN scale gradient rounded once per window instead of four times); ``dense_wgrad_side``: the dense layers' weight / bias gradients on
        the second stream beside their data-gradients (False: in front of them on the main stream, as until round 4); ``fuse_dense``: the
        data-gradient of a dense layer fed by a cell also runs that cell's BN / ReLU backward (asr_tap_gemm_gated_dense); ``se_sums``: the
        forward launch of an SE block's branch cell also makes the block's squeeze sums (asr_tap_gemm_wino_sums; the squeeze is then a
        sum in another order: to rounding); ``nt_splitk``: the split-K dense layers on the LDS-DMA kernel (asr_tap_gemm_nt_splitk; False: the
        64 x 64 register-staged tiles of asr_tap_gemm_splitk, eight splits, as until round 4: to rounding).  Each of them leaves the results bitwise (streams, fusions,
        compact form) or to rounding (Winograd) unchanged; they exist for A/B measurements and tests."""
        assert T % 8 == 0 and F >= 8
        self.opt_dual, self.opt_wino, self.opt_fuse = bool(dual_stream), bool(wino), bool(fuse_prologues)
        self.opt_fuse_se, self.opt_compact, self.side_priority = bool(fuse_se), bool(compact_pool), int(side_priority)
        self.opt_dense_side, self.opt_fuse_dense, self.opt_se_sums = bool(dense_wgrad_side), bool(fuse_dense), bool(se_sums)
        self.opt_nt_splitk = bool(nt_splitk)
        self.model, self.V, self.B, self.T, self.F = model, vocab, B, T, F
        self.device = device
        self.g = graph(model, vocab, widths, F)
        self.lr0, self.decay_steps, self.min_lr = lr, decay_steps, min_lr
        self.beta1, self.beta2, self.adam_eps = beta1, beta2, adam_eps
        self.global_step = 0
        self.n_valid, self.loss_denom = B, float(B)
        self._layout_params()
        self.init_params(seed)
        self._alloc_buffers()

    # ------------------------------------------------------------------ parameters
    def _layout_params(self):
        ent, self.n_gamma, self.dense_end, off = param_layout(self.g)
        self.entries = ent
        self.n_params_padded = off
        self.n_params = sum(int(np.prod(s)) for _, s in ent.values())
        z = lambda: torch.zeros(off, dtype=torch.float32, device=self.device)
        self.theta, self.grad, self.adam_m, self.adam_v = z(), z(), z(), z()
        self.bn_scale = torch.zeros(max(4, self.n_gamma), dtype=torch.float32, device=self.device)
        self.dscale = torch.zeros(max(4, self.n_gamma), dtype=torch.float32, device=self.device)
        self.ones = None

    def _view(self, buf, layer, key):
        off, shape = self.entries[(layer, key)]
        return buf[off:off + int(np.prod(shape))]

    def p(self, layer, key):
        return self._view(self.theta, layer, key)

    def gview(self, layer, key):
        return self._view(self.grad, layer, key)

    def scale_of(self, layer):
        off, shape = self.entries[(layer, 'gamma')]
        return self.bn_scale[off:off + shape[0]]

    def dscale_of(self, layer):
        off, shape = self.entries[(layer, 'gamma')]
        return self.dscale[off:off + shape[0]]

    def init_params(self, seed=0):
        """tf.layers defaults (acoustic_model2.py:35 initializer=None): Glorot-uniform kernels,
        zero biases, BN gamma = 1, beta = 0 (moving mean 0 / variance 1 stay frozen, Q1)."""
        rng = np.random.default_rng(seed)
        host = np.zeros(self.n_params_padded, dtype=np.float32)

        def put(layer, key, val):
            off, shape = self.entries[(layer, key)]
            host[off:off + val.size] = val.astype(np.float32).ravel()

        for op in self.g:
            if op[0] == 'cell':
                _, src, dst, cin, cout, k, pool = op
                lim = math.sqrt(6.0 / (k * k * cin + k * k * cout))
                put(dst, 'w', rng.uniform(-lim, lim, (k, k, cin, cout)))
                put(dst, 'gamma', np.ones(cout))
            elif op[0] == 'se':
                _, main, br, dst, Cc, hid, use_bn = op
                put(dst, 'w1', rng.uniform(-1, 1, (Cc, hid)) * math.sqrt(6.0 / (Cc + hid)))
                put(dst, 'w2', rng.uniform(-1, 1, (hid, Cc)) * math.sqrt(6.0 / (Cc + hid)))
                if use_bn:
                    put(dst, 'gamma', np.ones(Cc))
            elif op[0] == 'dense':
                _, src, dst, cin, cout, act = op
                put(dst, 'w', rng.uniform(-1, 1, (cin, cout)) * math.sqrt(6.0 / (cin + cout)))
        self.theta.copy_(torch.from_numpy(host))
        self.adam_m.zero_(); self.adam_v.zero_()
        self.global_step = 0

    def load_params(self, P):
        """P: {layer: {key: ndarray}} in the reference's tensor layouts (HWIO kernels)."""
        host = self.theta.cpu().numpy()
        for (layer, key), (off, shape) in self.entries.items():
            v = np.asarray(P[layer][key], dtype=np.float32)
            assert tuple(v.shape) == shape, (layer, key, v.shape, shape)
            host[off:off + v.size] = v.ravel()
        self.theta.copy_(torch.from_numpy(host))

    def params_dict(self, buf=None):
        host = (self.theta if buf is None else buf).cpu().numpy()
        out = {}
        for (layer, key), (off, shape) in self.entries.items():
            out.setdefault(layer, {})[key] = host[off:off + int(np.prod(shape))].reshape(shape).copy()
        return out

    def grads_dict(self):
        return self.params_dict(self.grad)

    # ------------------------------------------------------------------ buffers
    def _alloc_buffers(self):
        B, dev = self.B, self.device
        self.res = {'x': (self.T, self.F, 1)}
        self.y = {}          # name -> Plane (cell/SE outputs in padded layout)
        self.a = {}          # cell name -> Plane of post-ReLU pre-pool activations
        self.flat = {}       # name -> [rows, C] tensor (inputs/outputs of dense layers)
        self.se_state = {}
        self.dz_pool = {}    # geometry -> shared dZ plane
        self.dy = {}         # name -> Plane gradient buffers (allocated lazily per geometry)
        self.fdesc, self.bdesc, self.wdesc = {}, {}, {}
        ws_bytes = 1 << 20
        self.splitk = {}
        self.dsplitk, self.dsplitk_n = {}, {}
        consumers = {}
        for op in self.g:
            srcs = [op[1]] if op[0] != 'se' else [op[1], op[2]]
            for s in srcs:
                consumers.setdefault(s, []).append(op)
        self.consumers = consumers
        for op in self.g:
            if op[0] == 'cell':
                _, src, dst, cin, cout, k, pool = op
                H, W, _ = self.res[src]
                Ho, Wo = (H // 2, W // 2) if pool else (H, W)
                self.res[dst] = (Ho, Wo, cout)
                feeds_dense = any(c[0] == 'dense' for c in consumers.get(dst, []))
                if src == 'x':
                    self.y[dst] = Plane(B, Ho, Wo, cout, dev)
                    ws_bytes = max(ws_bytes, ops.cell1_bwd_workspace(B, H, W, cout))
                    continue
                self.a[dst] = Plane(B, H, W, cout, dev)
                NP = B * (H + 1) * (W + 1)
                if feeds_dense:
                    assert not pool
                    self.flat[dst] = torch.zeros(B * H * W, cout, dtype=torch.float32, device=dev)
                else:
                    self.y[dst] = Plane(B, Ho, Wo, cout, dev)
                nt = 9 if k == 3 else 1
                self.fdesc[dst] = ops.gemm_desc(NP, cin, cout, cin, cout, cout, cout, ntaps=nt, B=B, H=H, W=W, relu=1,
                                                y_unpadded=1 if feeds_dense else 0)
                self.bdesc[dst] = ops.gemm_desc(NP, cout, cin, cout, cout, 0, cin, ntaps=nt, B=B, H=H, W=W, wmode=1)
                self.wdesc[dst] = ops.gemm_desc(NP, cin, cout, cin, cout, ntaps=nt, B=B, H=H, W=W)
                ws_bytes = max(ws_bytes, ops.tap_wgrad_workspace(self.wdesc[dst]),
                               ops.cell_bwd_pre_workspace(B, H, W, cout))
                geo = (H, W, cout)
                if geo not in self.dz_pool:
                    self.dz_pool[geo] = Plane(B, H, W, cout, dev)
            elif op[0] == 'se':
                _, main, br, dst, Cc, hid, use_bn = op
                H, W, _ = self.res[main]
                self.res[dst] = (H, W, Cc)
                self.y[dst] = Plane(B, H, W, Cc, dev)
                self.se_state[dst] = torch.zeros(ops.se_state_floats(B, Cc, hid), dtype=torch.float32, device=dev)
                ws_bytes = max(ws_bytes, ops.se_fwd_workspace(B, H, W, Cc), ops.se_bwd_workspace(B, H, W, Cc, hid),
                               ops.se_bwd_cell_workspace(B, H, W, Cc, hid) + 4 * 4096)
            elif op[0] == 'dense':
                _, src, dst, cin, cout, act = op
                H = self.res[src][0]
                rows = B * H
                self.res[dst] = (H, 1, cout)
                if src not in self.flat:
                    raise ValueError('dense input %s must come from a cell or dense' % src)
                self.flat[dst] = torch.zeros(rows, cout, dtype=torch.float32, device=dev)
                self.fdesc[dst] = ops.gemm_desc(rows, cin, cout, cin, cout, 0, cout, ntaps=1, relu=1 if act == 'relu' else 0)
                self.bdesc[dst] = ops.gemm_desc(rows, cout, cin, cout, cout, 0, cin, ntaps=1, wmode=1)
                self.wdesc[dst] = ops.gemm_desc(rows, cin, cout, cin, cout, ntaps=1)
                ws_bytes = max(ws_bytes, ops.tap_wgrad_workspace(self.wdesc[dst]), ops.colsum_workspace(rows, cout))
                # a deep contraction with few output tiles (6400 -> 128: 200 workgroups of 200 chunk steps) is split eight ways
                # over the grid (asr_tap_gemm_splitk: 185 -> 124 us with the second pass)
                # (decided by the layer's widths only, never by the batch: an utterance alone must give bitwise the logits it
                # gives inside a batch, tests/test_fullsize_gpu.py.  The invariant behind that test, rule by rule: (1) Winograd vs
                # direct 3x3 kernels and the Winograd kernel itself: widths and plane geometry only (asr_winograd_supported,
                # wino11_takes); (2) gemm1_kernel vs tap_gemm_kernel_v1 and gemm1's 128 x 64 / 128 x 128 tile DO look at the row count,
                # but all three sum an output element's K products in the same order -- bitwise equal on the same rows,
                # tests/test_gemm1_gpu.py::test_the_tile_choice_changes_no_bit; (3) split-K, here: widths only; (4) the chunk plans
                # of the weight-gradient kernels depend on batch and CU count, but weight gradients are sums over the batch
                # anyway and never enter this property)
                # Round 5: on the LDS-DMA kernel (asr_tap_gemm_nt_splitk, on the transposed kernel the other dense layers use anyway), the
                # split count the largest whose 128 x 64 tiles of a batch of 32 stay within 512 workgroups (tools/bench_splitk_scan.py:
                # five splits of forty chunks for 6400 -> 128 at T_pad 1600, 114 -> 85-94 us; eight at T_pad 1000).  A function of the
                # widths and T only (rule 3 above).
                def nt_splits(depth, width):
                    nominal = -(-32 * H // 128) * -(-width // 64)
                    return next((k for k in range(16, 1, -1) if depth % (32 * k) == 0 and nominal * k <= 512), 2)
                if cout <= 128 and cin >= 2048 and cin % 256 == 0:
                    self.splitk[dst] = nt_splits(cin, cout) if self.opt_nt_splitk else 8
                    ws_bytes = max(ws_bytes, ops.tap_gemm_nt_splitk_workspace(self.fdesc[dst], self.splitk[dst]))
                # the mirror case in the backward pass: the data-gradient of a wide layer over a narrow input (128 -> 1536:
                # dX [rows][128] = dZ [rows][1536] . W^T is 50 tiles of 128 x 128 with a 1536-deep contraction -- 103 us on 50 CUs)
                # runs as the same split-K GEMM on the transposed kernel the forward pass keeps anyway (self.wT)
                if cin <= 128 and cout >= 1024 and cout % 256 == 0 and src != 'x':
                    self.dsplitk[dst] = ops.gemm_desc(rows, cout, cin, cout, cin, 0, cin, ntaps=1)
                    self.dsplitk_n[dst] = nt_splits(cout, cin) if self.opt_nt_splitk else 8
                    ws_bytes = max(ws_bytes, ops.tap_gemm_nt_splitk_workspace(self.dsplitk[dst], self.dsplitk_n[dst]))
        self.T8 = self.res[self.g[-1][2]][0]
        T8, V = self.T8, self.V
        self.logits = torch.zeros(T8, B, V, dtype=torch.float32, device=dev)      # self.logits of the reference
        self.ctc_grad = torch.zeros(T8, B, V, dtype=torch.float32, device=dev)
        self.dflat = {}       # gradients of flat tensors
        for name, t in self.flat.items():
            self.dflat[name] = torch.zeros_like(t)
        self.loss = torch.zeros(B, dtype=torch.float32, device=dev)
        self.ctc_status = torch.zeros(B, dtype=torch.int32, device=dev)
        self.ctc_ws = torch.zeros(ops.ctc_workspace(T8, B, MAX_LABEL) // 8 + 8, dtype=torch.float64, device=dev)
        self.dec_ids = torch.zeros(B, T8, dtype=torch.int32, device=dev)
        self.dec_len = torch.zeros(B, dtype=torch.int32, device=dev)
        self.dec_ws = torch.zeros(ops.ctc_greedy_workspace(T8, B) // 4 + 4, dtype=torch.int32, device=dev)
        self.neg_sum = torch.zeros(B, dtype=torch.float32, device=dev)
        self.dist = torch.zeros(B, dtype=torch.float32, device=dev)
        self.scalars = torch.zeros(8, dtype=torch.float32, device=dev)     # [0] sum loss, [1] sum dist
        self.ws = torch.zeros(ws_bytes // 4 + 64, dtype=torch.float32, device=dev)
        # Second stream for the backward pass (dual_stream=False turns it off): the weight-gradient of a cell
        # (MFMA-bound) runs beside its data-gradient and the NEXT cell's HBM-bound backward prologue -- they only share the
        # read-only dZ plane -- so one kernel's last partial round of workgroups is filled by the others and the
        # HBM-bound prologues hide under MFMA work (+6 % M1, +8 % M2 at B = 32).  Per-kernel durations of the
        # overlapped kernels then overlap in any profile.  Results are bitwise the same as with one stream.
        # (The data-gradient is enqueued before the weight gradient -- it is on the critical path; starting the weight gradient
        # first or only after the data-gradient has finished were measured in round 2 and lose on both graphs.)
        self.side = torch.cuda.Stream(device=dev, priority=self.side_priority) if self.opt_dual else None
        self.ws_side = torch.zeros_like(self.ws) if self.side is not None else None
        # with the side stream every geometry gets a second dZ plane, used alternately, so the next cell's (HBM-bound)
        # backward prologue can run while the previous weight-gradient (MFMA-bound) still reads the other one
        self.dz_alt = {geo: Plane(p.B, p.H, p.W, p.C, dev) for geo, p in self.dz_pool.items()} if self.side is not None else {}
        self._decode_done = None
        self._cell_dims = {op[2]: (op[3], op[4]) for op in self.g if op[0] == 'cell'}
        # 3x3 convs (forward and data-gradient) on weights pre-arranged in MFMA fragment order (asr_arrange_weights /
        # asr_tap_gemm_pw: same fp32 arithmetic, no weight tile in LDS, +6..20 % per layer over asr_tap_gemm).
        # Both views are regenerated from the fp32 parameters at the start of every forward pass.
        self.wf_f, self.wf_b = {}, {}
        for op in self.g:
            if op[0] == 'cell' and op[1] != 'x' and op[5] == 3:
                _, src, dst, cin, cout, k, pool = op
                self.wf_f[dst] = torch.zeros(ops._lib.load().asr_arrange_weights_bytes(9, cin, cout) // 4, dtype=torch.float32, device=dev)
                self.wf_b[dst] = torch.zeros(ops._lib.load().asr_arrange_weights_bytes(9, cout, cin) // 4, dtype=torch.float32, device=dev)
        # Winograd F(2x2,3x3) kernels (wino.hip) for the 3x3 convs they support, forward and data-gradient (plain and gated):
        # 16 instead of 36 multiplies per 2x2 output tile, still fp32; 1.3-1.6x the tap-GEMM per layer.  For a pooled cell
        # the 2x2 pool is computed inside the forward launch.  wino=False keeps every layer on the tap-GEMM.
        # dense layers: the kernel transposed ([cout][cin], one batched launch per step), so that the large forward GEMMs read
        # both operands K-contiguous through LDS-DMA (asr_tap_gemm_nt, gemm1.hip: 6400 x 6400 x 1536 of acoustic_model2.py
        # 109 -> 125 TFLOP/s); small ones fall through to asr_tap_gemm inside the library
        self.wT, items = {}, []
        for op in self.g:
            if op[0] == 'dense':
                _, src, dst, cin, cout, act = op
                self.wT[dst] = torch.zeros(cout * cin, dtype=torch.float32, device=dev)
                items.append((self.wT[dst], cin, self.p(dst, 'w'), cout, cin, cout))
        self._wT_batch = ops.Copy2dBatch(items) if items else None
        self.wt_f, self.wt_b = {}, {}
        if self.opt_wino:
            for op in self.g:
                if op[0] == 'cell' and op[1] != 'x' and op[5] == 3:
                    _, src, dst, cin, cout, k, pool = op
                    if ops.winograd_supported(self.fdesc[dst]):
                        self.wt_f[dst] = torch.zeros(ops.winograd_weights_floats(cin, cout), dtype=torch.float32, device=dev)
                    if ops.winograd_supported(self.bdesc[dst]):
                        self.wt_b[dst] = torch.zeros(ops.winograd_weights_floats(cin, cout), dtype=torch.float32, device=dev)
        self.labels = torch.zeros(B, MAX_LABEL, dtype=torch.int32, device=dev)
        self.label_len = torch.zeros(B, dtype=torch.int32, device=dev)
        self.seq_len = torch.zeros(B, dtype=torch.int32, device=dev)
        # gradient planes: one per named tensor that feeds a cell (except x) or an SE branch.
        # dL/d(main) of an SE block IS dL/d(its output) (identity branch), so the two share one
        # plane: later contributions to main accumulate into the block's dout buffer.
        self.alias = {}
        for op in self.g:
            if op[0] == 'se' and op[1] != op[2]:
                self.alias[op[1]] = op[3]
        for op in self.g:
            if op[0] == 'cell' and op[1] != 'x':
                self._dplane(op[1])
            if op[0] == 'se':
                self._dplane(op[2])
                self._dplane(op[3])
        self._plan_fused_prologues()
        # Max-pooled cells whose forward conv and whose gated data-gradient both run on the Winograd kernel keep no pre-pool
        # activation plane: the forward writes the activation at each window's maximum + its position, the backward reads those
        # (asr_tap_gemm_wino_poolmax / asr_tap_gemm_gated_poolmax: a quarter of the traffic, the same bits).  Widths and geometry only.
        # SE blocks whose branch is a non-pooled conv cell that nothing else reads (acoustic_model2: cell -> squeeze-excitation): the SE
        # backward's last pass also runs that cell's BN / ReLU backward (asr_se_bwd_cell) -- no dL/dy plane, no asr_cell_bwd_pre pass
        self.se_cell = {}
        if self.opt_fuse and self.opt_fuse_se:
            for op in self.g:
                if op[0] != 'se':
                    continue
                br = op[2]
                cell = next((c for c in self.g if c[0] == 'cell' and c[2] == br), None)
                if cell is not None and cell[1] != 'x' and cell[6] is None and br not in self.dflat and br not in self.fuse.values() \
                        and self.consumers.get(br, []) == [op]:
                    self.se_cell[op[3]] = cell
        # Dense layers whose input is the flattened output of an un-pooled cell that nothing else reads (the reshape -> dense heads):
        # the dense data-gradient's epilogue runs that cell's BN / ReLU backward (asr_tap_gemm_gated_dense) -- no dL/d(flat)
        # tensor, no asr_cell_bwd_pre pass.  {dense layer -> cell op}.  Widths and geometry only.
        self.dense_gate = {}
        if self.opt_fuse and self.opt_fuse_dense:
            for op in self.g:
                if op[0] != 'dense':
                    continue
                cell = next((c for c in self.g if c[0] == 'cell' and c[2] == op[1]), None)
                if cell is None or cell[1] == 'x' or cell[6] is not None or self.consumers.get(op[1], []) != [op] or op[2] in self.dsplitk:
                    continue
                Hc, Wc, _ = self.res[cell[1]]
                if ops.tap_gemm_gated_dense_supported(self.bdesc[op[2]], Hc, Wc, cell[4]):
                    self.dense_gate[op[2]] = cell
                    self.ws_gate = max(self.ws_gate, ops.tap_gemm_gated_dense_workspace(self.bdesc[op[2]], Wc, cell[4]))
                    self.dflat[op[1]] = None          # dL/d(flat) of that cell is never materialised (the key stays: "feeds a dense layer")
        # SE blocks whose branch is a non-pooled Winograd cell that only the block reads: the cell's forward launch also writes the
        # per-image channel sums of its output (asr_tap_gemm_wino_sums), the block's squeeze takes them (asr_se_fwd_sums) instead of a
        # pass of its own over the plane.  {branch cell -> (SE block, partial rows buffer, rows per image)}
        self.se_sums = {}
        if self.opt_se_sums:
            for op in self.g:
                if op[0] != 'se' or op[1] == op[2]:
                    continue
                br = op[2]
                cell = next((c for c in self.g if c[0] == 'cell' and c[2] == br), None)
                if cell is None or cell[1] == 'x' or cell[6] is not None or br not in self.wt_f or br in self.flat \
                        or self.consumers.get(br, []) != [op]:
                    continue
                rows = ops.winograd_sum_rows(self.fdesc[br])
                if rows > 0 and rows % B == 0:
                    self.se_sums[br] = (op[3], torch.zeros(rows * cell[4], dtype=torch.float32, device=dev), rows // B)
        # ... and the mirror in the backward pass: the (plain, first-writer) Winograd data-gradient of the ONE cell that reads an SE block's
        # output also leaves the block's first backward reduction, sum dout * (sc * x + sh) (asr_tap_gemm_wino_sesum -> asr_se_bwd_cell_sums).
        # {consumer cell -> (SE block op, partial rows buffer, rows per image)}
        self.se_xsum = {}
        if self.opt_se_sums:
            for op in self.g:
                if op[0] != 'se' or op[3] not in self.se_cell:
                    continue
                cons = self.consumers.get(op[3], [])
                if len(cons) != 1 or cons[0][0] != 'cell' or cons[0][2] not in self.wt_b or cons[0][2] in self.fuse:
                    continue
                rows = ops.winograd_sum_rows(self.bdesc[cons[0][2]])
                if rows > 0 and rows % B == 0:
                    self.se_xsum[cons[0][2]] = (op, torch.zeros(rows * op[4], dtype=torch.float32, device=dev), rows // B)
        self._se_xsum_ready = {}
        self.compact, self.compact_avg = {}, {}
        for writer, tgt in (self.fuse.items() if self.opt_compact else ()):
            top = next(o for o in self.g if o[0] == 'cell' and o[2] == tgt)
            if top[6] in ('max', 'avg') and tgt in self.wt_f and writer in self.wt_b and ops.poolmax_supported(self.fdesc[tgt], self.bdesc[writer]):
                Ho, Wo, cout = self.res[tgt]
                index = ops.poolmax_index(B, Ho, Wo, cout, dev) if top[6] == 'max' else ops.poolavg_index(B, Ho, Wo, cout, dev)
                self.compact[tgt] = (Plane(B, Ho, Wo, cout, dev), index)      # activation at the maximum / sum of the window's activations
                self.compact_avg[tgt] = top[6] == 'avg'
                del self.a[tgt]
        if self.fuse and not self.dz_alt:
            # a fused data-gradient reads dZ of its own cell while its epilogue writes dZ of the cell in front: two planes
            # per geometry are needed on one stream as well (the two cells may share a geometry)
            self.dz_alt = {geo: Plane(p.B, p.H, p.W, p.C, dev) for geo, p in self.dz_pool.items()}
        if self.ws_gate > self.ws.numel() * 4:
            self.ws = torch.zeros(self.ws_gate // 4 + 64, dtype=torch.float32, device=dev)
            self.ws_side = torch.zeros_like(self.ws) if self.side is not None else None

    def options(self):
        """The switches this engine was built with and what they resolved to on this graph (bench.py puts it in its JSON line)."""
        return {'dual_stream': self.opt_dual, 'wino': self.opt_wino, 'fuse_prologues': self.opt_fuse, 'fuse_se': self.opt_fuse_se,
                'compact_pool': self.opt_compact, 'side_priority': self.side_priority, 'dense_wgrad_side': self.opt_dense_side, 'fuse_dense': self.opt_fuse_dense, 'se_sums': self.opt_se_sums, 'nt_splitk': self.opt_nt_splitk,
                'se_squeeze_in_the_branch_conv': sorted(self.se_sums), 'se_backward_reduction_in_the_consumer_dgrad': sorted(self.se_xsum),
                'winograd_layers_fwd': sorted(self.wt_f), 'winograd_layers_dgrad': sorted(self.wt_b),
                'fused_prologues': len(self.fuse), 'dense_gradients_with_cell_backward': sorted(self.dense_gate), 'se_blocks_fused_with_cell_backward': len(self.se_cell),
                'compact_pool_cells': sorted(self.compact)}

    def _plan_fused_prologues(self):
        """Cells whose backward prologue (pool -> BN -> ReLU backward + the three channel sums, asr_cell_bwd_pre) moves into
        the epilogue of the data-gradient GEMM that COMPLETES their output gradient (asr_tap_gemm_gated): {writer cell ->
        gated cell}.  The gradient plane of a cell's output may receive several contributions (an SE block's identity
        branch is the plane itself, other consumers accumulate); the prologue can only run on the sum, i.e. in the LAST
        writer of the backward pass = the FIRST writer in graph order, and only if that writer is a conv cell.  Not fused:
        cells read by a dense layer (their gradient arrives in the dense layout), the first cell (own kernel) and pooled
        cells with odd plane sizes."""
        self.fuse, self.ws_gate = {}, 0
        if not self.opt_fuse:
            return
        for c in self.g:
            if c[0] != 'cell' or c[1] == 'x' or c[2] in self.dflat:
                continue
            H, W, _ = self.res[c[1]]
            if c[6] and (H % 2 or W % 2):
                continue
            P = self._root(c[2])
            writers = [o for o in self.g if (o[0] == 'cell' and o[1] != 'x' and self._root(o[1]) == P)
                       or (o[0] == 'se' and self._root(o[2]) == P)]
            if writers and writers[0][0] == 'cell' and writers[0][2] not in self.fuse:
                self.fuse[writers[0][2]] = c[2]
                self.ws_gate = max(self.ws_gate, ops.tap_gemm_gated_workspace(self.bdesc[writers[0][2]]))

    def _root(self, name):
        while name in self.alias:
            name = self.alias[name]
        return name

    def _dplane(self, name):
        name = self._root(name)
        if name not in self.dy:
            H, W, Cc = self.res[name]
            self.dy[name] = Plane(self.B, H, W, Cc, self.device)
        return self.dy[name]

    # ------------------------------------------------------------------ forward
    def refresh_bn(self):
        if self.n_gamma:
            ops.axpy(self.bn_scale[:self.n_gamma], self.theta[:self.n_gamma], RS, False)

    def pad_batch(self, x):
        """The reference feeds the B' <= B rows that survived the data loader's checks (placeholders with a None batch
        dimension, acoustic_model2.py:29-32; data_loader.py:149-156).  The engine's planes are sized for B rows, so a
        short batch is copied into a persistent [B, T, F] buffer whose remaining rows are zero; set_targets(...,
        n_valid=B') then marks those rows as padding (CTC loss 0, gradient 0, not counted in any mean)."""
        n = x.shape[0]
        if n == self.B:
            return x
        if n > self.B:
            raise ValueError('batch of %d rows for an engine built for %d' % (n, self.B))
        if getattr(self, '_xpad', None) is None:
            self._xpad = torch.zeros(self.B, self.T, self.F, dtype=torch.float32, device=self.device)
        self._xpad[:n].copy_(x)
        self._xpad[n:].zero_()
        return self._xpad

    def _forward_weights(self):
        """Per-step weight views the FORWARD convs read: Winograd-transformed where the layer runs on that kernel, else
        MFMA fragment order (a layer never needs both)."""
        for dst, buf in self.wf_f.items():
            cin, cout = self._cell_dims[dst]
            if dst in self.wt_f:
                ops.winograd_weights(self.p(dst, 'w'), cin, cout, cout, 0, self.wt_f[dst])
            else:
                ops.arrange_weights(self.p(dst, 'w'), 9, cin, cout, cout, 0, buf)
        if self._wT_batch is not None:
            self._wT_batch.run_transposed()

    def _backward_weights(self):
        """The same for the data-gradient views (mirrored taps); not needed before the backward pass."""
        for dst, buf in self.wf_b.items():
            cin, cout = self._cell_dims[dst]
            if dst in self.wt_b:
                ops.winograd_weights(self.p(dst, 'w'), cout, cin, cout, 1, self.wt_b[dst])
            else:
                ops.arrange_weights(self.p(dst, 'w'), 9, cout, cin, cout, 1, buf)

    def forward(self, x):
        """x: [B, T, F] float32 on the device (the wav_input placeholder without its last axis); fewer rows are padded
        (pad_batch)."""
        if x.shape[0] != self.B:
            x = self.pad_batch(x)
        assert x.is_contiguous() and tuple(x.shape) == (self.B, self.T, self.F)
        self.x = x
        self.refresh_bn()
        # the fragment-order weight copies (two small launches per 3x3 layer, ~20 in all) are not needed before the second
        # cell: they run on the side stream beside the VALU-bound first cell instead of in front of it
        # The forward views first: the second cell waits for them only (they used to sit behind the backward views: 36 us of
        # an idle main stream per step); the backward pass waits for the rest (self._wb_ready).
        wf_ready = None
        self._wb_ready = None
        if self.wf_f and self.side is not None:
            params_final = torch.cuda.Event()
            params_final.record()                         # Adam of the previous step / load_params are on the main stream
            self.side.wait_event(params_final)
            with torch.cuda.stream(self.side):
                self._forward_weights()
                wf_ready = torch.cuda.Event()
                wf_ready.record()
                self._backward_weights()
                self._wb_ready = torch.cuda.Event()
                self._wb_ready.record()
        else:
            self._forward_weights()
            self._backward_weights()
        for op in self.g:
            if op[0] == 'cell':
                _, src, dst, cin, cout, k, pool = op
                sc, sh = self.scale_of(dst), self.p(dst, 'beta')
                pm = {None: 0, 'avg': 1, 'max': 2}[pool]
                if src == 'x':
                    ops.cell1_fwd(x, self.p(dst, 'w'), self.p(dst, 'b'), sc, sh, pm, self.y[dst])
                    continue
                if wf_ready is not None:
                    torch.cuda.current_stream().wait_event(wf_ready)
                    wf_ready = None
                out_y = None if pool else (self.flat[dst] if dst in self.flat else self.y[dst])
                if dst in self.compact:
                    (ops.tap_gemm_wino_poolavg if self.compact_avg[dst] else ops.tap_gemm_wino_poolmax)(
                        self.fdesc[dst], self.y[src], self.wt_f[dst], self.p(dst, 'b'), sc, sh, self.y[dst], *self.compact[dst])
                    continue
                if dst in self.wt_f and pool:
                    # conv + bias + ReLU -> BN -> 2x2 pool in one launch: a Winograd tile is a pooling window
                    ops.tap_gemm_wino_pool(self.fdesc[dst], self.y[src], self.wt_f[dst], self.p(dst, 'b'), sc, sh, self.a[dst], pm, self.y[dst])
                    continue
                elif dst in self.se_sums:
                    ops.tap_gemm_wino_sums(self.fdesc[dst], self.y[src], self.wt_f[dst], self.p(dst, 'b'), sc, sh, self.a[dst], out_y, self.se_sums[dst][1])
                elif dst in self.wt_f:
                    ops.tap_gemm_wino(self.fdesc[dst], self.y[src], self.wt_f[dst], self.p(dst, 'b'), sc, sh, self.a[dst], out_y)
                elif dst in self.wf_f:
                    ops.tap_gemm_pw(self.fdesc[dst], self.y[src], self.wf_f[dst], self.p(dst, 'b'), sc, sh, self.a[dst], out_y)
                else:
                    ops.tap_gemm(self.fdesc[dst], self.y[src], self.p(dst, 'w'), self.p(dst, 'b'), sc, sh,
                                 self.a[dst], out_y)
                if pool:
                    ops.pool_fwd(self.a[dst], sc, sh, pm, self.y[dst])
            elif op[0] == 'se':
                _, main, br, dst, Cc, hid, use_bn = op
                sc, sh = self._se_affine(dst, Cc, use_bn)
                if br in self.se_sums:
                    _, sums, per = self.se_sums[br]
                    ops.se_fwd_sums(self.y[main], self.y[br], hid, sc, sh, self.p(dst, 'w1'), self.p(dst, 'b1'),
                                    self.p(dst, 'w2'), self.p(dst, 'b2'), self.se_state[dst], sums, per, self.y[dst])
                    continue
                ops.se_fwd(self.y[main], self.y[br], hid, sc, sh, self.p(dst, 'w1'), self.p(dst, 'b1'),
                           self.p(dst, 'w2'), self.p(dst, 'b2'), self.se_state[dst], self.ws, self.y[dst])
            elif op[0] == 'dense':
                _, src, dst, cin, cout, act = op
                if dst in self.splitk:
                    if self.opt_nt_splitk:
                        ops.tap_gemm_nt_splitk(self.fdesc[dst], self.flat[src], self.wT[dst], cin, self.p(dst, 'b'), None, None,
                                               None, self.flat[dst], self.splitk[dst], self.ws)
                    else:
                        ops.tap_gemm_splitk(self.fdesc[dst], self.flat[src], self.p(dst, 'w'), self.p(dst, 'b'), None, None,
                                            None, self.flat[dst], self.splitk[dst], self.ws)
                else:
                    ops.tap_gemm_nt(self.fdesc[dst], self.flat[src], self.p(dst, 'w'), self.wT[dst], cin, self.p(dst, 'b'), None, None,
                                    None, self.flat[dst])
        ops.softmax_log_fwd(self.flat[self.g[-1][2]], self.B, self.T8, self.V, K_EPSILON, self.logits)
        return self.logits

    def _se_affine(self, dst, Cc, use_bn):
        if use_bn:
            return self.scale_of(dst), self.p(dst, 'beta')
        if self.ones is None or self.ones.numel() < 2 * Cc:
            self.ones = torch.cat([torch.ones(1024, device=self.device), torch.zeros(1024, device=self.device)])
        return self.ones[:Cc], self.ones[1024:1024 + Cc]

    # ------------------------------------------------------------------ loss / decode
    def set_targets(self, logits_length, target_py, target_length=None, n_valid=None, loss_denom=None):
        """logits_length [B'] ints, target_py [B', <=64] zero-padded ids, B' = n_valid <= B (default B): rows
        n_valid..B-1 are padding (sequence length 0, no labels: asr_ctc_loss status 2).  ``loss_denom`` is the row count the
        gradient of the mean loss is divided by (reduce_mean, acoustic_model2.py:83): B' by default; under data
        parallelism the GLOBAL number of surviving rows, so that the summed all-reduce is the global-batch mean.  Mirrors
        tf.contrib.layers.dense_to_sparse (acoustic_model2.py:71): every 0 is dropped (Q6).
        With ``target_length`` [B] the labels are the DENSE form of tf.nn.ctc_loss_v2(labels=target_py,
        label_length=target_length) (am_lm_model.py:72): the first target_length ids of each row, zeros kept.
        Raises ValueError where TF raises InvalidArgumentError (no valid CTC alignment)."""
        self.commit_targets(self.prepare_targets(logits_length, target_py, target_length, n_valid, loss_denom))

    def prepare_targets(self, logits_length, target_py, target_length=None, n_valid=None, loss_denom=None):
        """The host half of set_targets: every check (and its ValueError) and the label arrays, nothing touches the device or
        the engine.  A data-parallel caller runs it BEFORE its first collective of the step, so that a rank whose batch cannot
        be aligned fails together with the others instead of leaving them inside an all-reduce (acoustic_model.run)."""
        tp = np.asarray(target_py)
        nv = self.B if n_valid is None else int(n_valid)
        if not 0 <= nv <= self.B:
            raise ValueError('n_valid %d outside [0, %d]' % (nv, self.B))
        lab = np.zeros((self.B, MAX_LABEL), dtype=np.int32)
        ll = np.zeros(self.B, dtype=np.int32)
        sl = np.zeros(self.B, dtype=np.int32)
        sl[:nv] = np.asarray(logits_length, dtype=np.int32).reshape(-1)[:nv]
        for b in range(nv):
            ids = tp[b][tp[b] != 0] if target_length is None else np.asarray(tp[b])[:int(target_length[b])]
            if len(ids) > MAX_LABEL:
                raise ValueError('label longer than %d' % MAX_LABEL)
            rep = int(np.sum(ids[1:] == ids[:-1]))
            if sl[b] < len(ids) + rep or sl[b] <= 0 or sl[b] > self.T8:
                raise ValueError('Not enough time for target transition sequence (required: %d, available: %d) '
                                 'in batch %d' % (len(ids) + rep, sl[b], b))
            lab[b, :len(ids)] = ids
            ll[b] = len(ids)
        return nv, loss_denom, lab, ll, sl

    def commit_targets(self, prepared, loss_denom=None):
        nv, denom, lab, ll, sl = prepared
        if loss_denom is not None:
            denom = loss_denom
        self.n_valid = nv
        self.loss_denom = float(denom if denom is not None else max(nv, 1))
        self.labels.copy_(torch.from_numpy(lab), non_blocking=True)
        self.label_len.copy_(torch.from_numpy(ll), non_blocking=True)
        self.seq_len.copy_(torch.from_numpy(sl), non_blocking=True)
        self._host_labels = [lab[b, :ll[b]].tolist() for b in range(nv)]

    def loss_and_decode(self, defer_decode_join=False):
        """CTC loss + gradient, greedy decode, edit distance.  With the side stream the decode runs beside the CTC
        lattice; ``defer_decode_join=True`` (training loops) leaves it running until the end of backward(), otherwise
        the main stream waits for it before this returns (so dec_ids / dist / neg_sum can be read right away)."""
        B, T8, V = self.B, self.T8, self.V
        side = self.side
        if side is not None:
            # greedy decode + edit distance only need the logits: they run beside the CTC lattice (both are latency-bound,
            # a few dozen workgroups) and the head's backward; the main stream joins them again in backward()
            logits_ready = torch.cuda.Event()
            logits_ready.record()
            if self._decode_done is not None:           # (the previous step's decode is long finished; formal ordering only)
                side.wait_event(self._decode_done)
            side.wait_event(logits_ready)
            with torch.cuda.stream(side):
                ops.ctc_greedy(self.logits, T8, B, V, self.seq_len, V - 1, self.dec_ids, self.dec_len, self.neg_sum, self.dec_ws)
                ops.edit_distance(self.dec_ids, T8, self.dec_len, self.labels, MAX_LABEL, self.label_len, B, self.dist)
                ops.colsum(self.dist, B, 1, 1, self.scalars[1:2], self.ws_side)
        ops.ctc_loss(self.logits, T8, B, V, self.labels, MAX_LABEL, self.label_len, self.seq_len, V - 1,
                     self.loss, self.ctc_grad, self.ctc_status, self.ctc_ws)
        if side is None:
            ops.ctc_greedy(self.logits, T8, B, V, self.seq_len, V - 1, self.dec_ids, self.dec_len, self.neg_sum, self.dec_ws)
            ops.edit_distance(self.dec_ids, T8, self.dec_len, self.labels, MAX_LABEL, self.label_len, B, self.dist)
            ops.colsum(self.dist, B, 1, 1, self.scalars[1:2], self.ws)
            ops.colsum(self.loss, B, 1, 1, self.scalars[0:1], self.ws)
        else:
            # the sum of the per-utterance losses is a fetch, not an input of the backward pass: on the second stream behind the decode
            # (one launch and its gap less between the lattice and the head's backward); _decode_done covers both
            loss_ready = torch.cuda.Event()
            loss_ready.record()
            side.wait_event(loss_ready)
            with torch.cuda.stream(side):
                ops.colsum(self.loss, B, 1, 1, self.scalars[0:1], self.ws_side)
                self._decode_done = torch.cuda.Event()
                self._decode_done.record()
        if side is not None and not defer_decode_join:
            torch.cuda.current_stream().wait_event(self._decode_done)

    # ------------------------------------------------------------------ backward
    def backward(self, on_dense_grads_ready=None, extra=None):
        """``extra`` = {dense activation name: device tensor}: an additional dL/d(activation) from a second consumer of
        that activation (the language half of the joint graph reads h7), added before the layer's own backward."""
        B, T8, V = self.B, self.T8, self.V
        last = self.g[-1][2]
        if getattr(self, '_wb_ready', None) is not None:          # the data-gradient weight views (side stream, forward())
            torch.cuda.current_stream().wait_event(self._wb_ready)
            self._wb_ready = None
        ops.softmax_log_bwd(self.logits, self.ctc_grad, B, T8, V, K_EPSILON, 1.0 / self.loss_denom, self.dflat[last])
        ready = set()          # gradient planes that already hold a value this step

        def grad_target(name):
            name = self._root(name)
            acc = name in ready
            ready.add(name)
            return self.dy[name], acc

        dense_pending = sum(1 for op in self.g if op[0] == 'dense')
        side_busy, dz_reader, flip = None, {}, {}
        fused_dz = {}          # cell name -> dZ plane already written by the data-gradient that completed its gradient

        def acquire_dz(geo):
            """The dZ plane of this geometry to write next (two per geometry, used alternately, when the weight-gradients
            run on the side stream), after waiting for the weight-gradient that may still read it."""
            dzp = self.dz_pool[geo]
            if geo in self.dz_alt:
                flip[geo] = not flip.get(geo, False)
                if flip[geo]:
                    dzp = self.dz_alt[geo]
            if id(dzp) in dz_reader:
                torch.cuda.current_stream().wait_event(dz_reader.pop(id(dzp)))
            return dzp
        for op in reversed(self.g):
            if op[0] == 'dense':
                _, src, dst, cin, cout, act = op
                dz = self.dflat[dst]
                if extra is not None and dst in extra:
                    ops.axpy(dz, extra[dst], 1.0, True)
                if act == 'relu':
                    ops.relu_bwd(dz, self.flat[dst], dz)
                rows = dz.shape[0]
                if self.side is not None and self.opt_dense_side:
                    # weight and bias gradient of a dense layer beside its data-gradient (round 5: they sat on the main stream in front
                    # of it -- ~0.1 ms of the critical path of the plain model while the second stream had nothing to do yet)
                    dz_final = torch.cuda.Event()
                    dz_final.record()
                    self.side.wait_event(dz_final)
                    with torch.cuda.stream(self.side):
                        ops.tap_wgrad(self.wdesc[dst], self.flat[src], dz, cout, self.gview(dst, 'w'), self.ws_side)
                        ops.colsum(dz, rows, cout, cout, self.gview(dst, 'b'), self.ws_side)
                        side_busy = torch.cuda.Event()
                        side_busy.record()
                else:
                    ops.tap_wgrad(self.wdesc[dst], self.flat[src], dz, cout, self.gview(dst, 'w'), self.ws)
                    ops.colsum(dz, rows, cout, cout, self.gview(dst, 'b'), self.ws)
                if dst in self.dense_gate:
                    cell = self.dense_gate[dst]
                    Hc, Wc, _ = self.res[cell[1]]
                    dzt = acquire_dz((Hc, Wc, cell[4]))
                    ops.tap_gemm_gated_dense(self.bdesc[dst], dz, self.p(dst, 'w'), self.a[src], self.scale_of(src), self.p(src, 'beta'), dzt,
                                             self.dscale_of(src), self.gview(src, 'beta'), self.gview(src, 'b'), self.ws)
                    fused_dz[src] = dzt
                elif dst in self.dsplitk:
                    # dX = dZ . W^T: the kernel W [cin][cout] is the K-contiguous second operand as it lies
                    if self.opt_nt_splitk:
                        ops.tap_gemm_nt_splitk(self.dsplitk[dst], dz, self.p(dst, 'w'), cout, None, None, None, None, self.dflat[src],
                                               self.dsplitk_n[dst], self.ws)
                    else:
                        ops.tap_gemm_splitk(self.dsplitk[dst], dz, self.wT[dst], None, None, None, None, self.dflat[src], 8, self.ws)
                else:
                    ops.tap_gemm(self.bdesc[dst], dz, self.p(dst, 'w'), None, None, None, None, self.dflat[src])
                dense_pending -= 1
                if dense_pending == 0 and on_dense_grads_ready is not None:
                    if side_busy is not None:          # the dense gradients are final once the second stream has written them
                        torch.cuda.current_stream().wait_event(side_busy)
                    on_dense_grads_ready()
            elif op[0] == 'se':
                _, main, br, dst, Cc, hid, use_bn = op
                dout = self._dplane(dst)
                sc, sh = self._se_affine(dst, Cc, use_bn)
                if use_bn:
                    dsc, dsh = self.dscale_of(dst), self.gview(dst, 'beta')
                else:
                    dsc, dsh = self.ws[-2048:-1024], self.ws[-1024:]
                same = (main == br)
                dx, acc = grad_target(br)
                assert not acc
                if dst in self.se_cell:
                    # the branch cell's backward prologue in the same pass: its dZ plane and channel sums, no dL/dy plane
                    cell = self.se_cell[dst]
                    Hc, Wc, _ = self.res[cell[1]]
                    dzt = acquire_dz((Hc, Wc, cell[4]))
                    xs = self._se_xsum_ready.pop(dst, None)
                    if xs is not None:
                        ops.se_bwd_cell_sums(dout, self.y[br], hid, sc, sh, self.p(dst, 'w1'), self.p(dst, 'w2'), self.se_state[dst],
                                             1 if same else 0, dsc, dsh, self.gview(dst, 'w1'), self.gview(dst, 'b1'),
                                             self.gview(dst, 'w2'), self.gview(dst, 'b2'), self.a[br], self.scale_of(br), dzt,
                                             self.dscale_of(br), self.gview(br, 'beta'), self.gview(br, 'b'), xs[0], xs[1], self.ws[:-2048])
                        fused_dz[br] = dzt
                        continue
                    ops.se_bwd_cell(dout, self.y[br], hid, sc, sh, self.p(dst, 'w1'), self.p(dst, 'w2'), self.se_state[dst],
                                    1 if same else 0, dsc, dsh, self.gview(dst, 'w1'), self.gview(dst, 'b1'),
                                    self.gview(dst, 'w2'), self.gview(dst, 'b2'), self.a[br], self.scale_of(br), dzt,
                                    self.dscale_of(br), self.gview(br, 'beta'), self.gview(br, 'b'), self.ws[:-2048])
                    fused_dz[br] = dzt
                    continue
                ops.se_bwd(dout, self.y[br], hid, sc, sh, self.p(dst, 'w1'), self.p(dst, 'w2'), self.se_state[dst],
                           1 if same else 0, dx, dsc, dsh, self.gview(dst, 'w1'), self.gview(dst, 'b1'),
                           self.gview(dst, 'w2'), self.gview(dst, 'b2'), self.ws[:-2048])
                # when main != branch, dL/d(main) = dout lives in the same plane (self.alias)
            elif op[0] == 'cell':
                _, src, dst, cin, cout, k, pool = op
                sc, sh = self.scale_of(dst), self.p(dst, 'beta')
                pm = {None: 0, 'avg': 1, 'max': 2}[pool]
                if src == 'x':
                    H, W, _ = self.res['x']
                    ops.cell1_bwd(self.x, self.p(dst, 'w'), self.p(dst, 'b'), sc, sh, pm, self._dplane(dst),
                                  self.gview(dst, 'w'), self.gview(dst, 'b'), self.dscale_of(dst),
                                  self.gview(dst, 'beta'), self.ws)
                    continue
                H, W, _ = self.res[src]
                if dst in fused_dz:
                    dz = fused_dz.pop(dst)           # written (with the channel sums) by the GEMM that completed dL/dy
                else:
                    dz = acquire_dz((H, W, cout))
                    if dst in self.dflat:
                        dyv, layout = self.dflat[dst], 2
                    else:
                        dyv, layout = self._dplane(dst), (1 if pool else 0)
                    ops.cell_bwd_pre(dyv, layout, self.a[dst], sc, sh, pm, dz, self.dscale_of(dst),
                                     self.gview(dst, 'beta'), self.gview(dst, 'b'), self.ws)
                wgrad = ops.tap_wgrad
                def run_wgrad():
                    nonlocal side_busy
                    if self.side is None:
                        wgrad(self.wdesc[dst], self.y[src], dz, cout, self.gview(dst, 'w'), self.ws)
                        return
                    self.side.wait_event(dz_ready)
                    with torch.cuda.stream(self.side):
                        wgrad(self.wdesc[dst], self.y[src], dz, cout, self.gview(dst, 'w'), self.ws_side)
                        side_busy = torch.cuda.Event()
                        side_busy.record()
                        dz_reader[id(dz)] = side_busy
                if self.side is not None:
                    dz_ready = torch.cuda.Event()
                    dz_ready.record()
                # The data-gradient is on the critical path (the next cell's prologue waits for it); the weight-gradient
                # is needed only at the end of backward: the data-gradient is enqueued first, so that its workgroups are not
                # locked out by the weight-gradient's one-round grid.
                dx, acc = grad_target(src)
                d = self.bdesc[dst]
                d.accumulate = 1 if acc else 0
                tgt = self.fuse.get(dst)
                if tgt is not None:
                    # this GEMM completes dL/dy of cell `tgt`: its epilogue applies tgt's pool / BN / ReLU backward and
                    # writes dZ(tgt) and the channel sums -- no asr_cell_bwd_pre pass for tgt
                    top = next(o for o in self.g if o[0] == 'cell' and o[2] == tgt)
                    Hf, Wf, _ = self.res[top[1]]
                    dzt = acquire_dz((Hf, Wf, top[4]))
                    if tgt in self.compact:
                        (ops.tap_gemm_gated_poolavg if self.compact_avg[tgt] else ops.tap_gemm_gated_poolmax)(
                                                   d, dz, self.wt_b[dst], Hf, Wf, *self.compact[tgt], self.scale_of(tgt), self.p(tgt, 'beta'),
                                                   dx if acc else None, dzt, self.dscale_of(tgt), self.gview(tgt, 'beta'),
                                                   self.gview(tgt, 'b'), self.ws)
                    else:
                        ops.tap_gemm_gated(d, dz, self.wt_b[dst] if dst in self.wt_b else self.wf_b[dst] if dst in self.wf_b else self.p(dst, 'w'),
                                           2 if dst in self.wt_b else 1 if dst in self.wf_b else 0,
                                           {None: 0, 'avg': 1, 'max': 2}[top[6]], self.a[tgt], self.scale_of(tgt), self.p(tgt, 'beta'),
                                           dx if acc else None, dzt, self.dscale_of(tgt), self.gview(tgt, 'beta'),
                                           self.gview(tgt, 'b'), self.ws)
                    fused_dz[tgt] = dzt
                elif dst in self.se_xsum and not acc:
                    se_op, xs, per = self.se_xsum[dst]
                    ssc, ssh = self._se_affine(se_op[3], se_op[4], se_op[6])
                    ops.tap_gemm_wino_sesum(d, dz, self.wt_b[dst], self.y[se_op[2]], ssc, ssh, dx, xs)
                    self._se_xsum_ready[se_op[3]] = (xs, per)
                elif dst in self.wt_b:
                    ops.tap_gemm_wino(d, dz, self.wt_b[dst], None, None, None, None, dx)
                elif dst in self.wf_b:
                    ops.tap_gemm_pw(d, dz, self.wf_b[dst], None, None, None, None, dx)
                else:
                    ops.tap_gemm(d, dz, self.p(dst, 'w'), None, None, None, None, dx)
                run_wgrad()
        if side_busy is not None:
            torch.cuda.current_stream().wait_event(side_busy)
        if self._decode_done is not None:
            torch.cuda.current_stream().wait_event(self._decode_done)
        if self.n_gamma:
            ops.axpy(self.grad[:self.n_gamma], self.dscale[:self.n_gamma], RS, False)

    # ------------------------------------------------------------------ optimiser
    def current_learning_rate(self, step=None):
        """tf.train.polynomial_decay(lr, global_step, decay_steps, end, cycle=True, power=0.5)
        (acoustic_model2.py:86-88, Appendix A10)."""
        step = float(self.global_step if step is None else step)
        ds = float(self.decay_steps)
        mult = 1.0 if step == 0 else math.ceil(step / ds)
        p = step / (ds * mult)
        return (self.lr0 - self.min_lr) * math.sqrt(max(0.0, 1.0 - p)) + self.min_lr

    def apply_adam(self, gscale=1.0):
        lr = self.current_learning_rate()
        t = self.global_step + 1
        lr_t = lr * math.sqrt(1.0 - self.beta2 ** t) / (1.0 - self.beta1 ** t)
        ops.adam_tf(self.theta, self.grad, self.adam_m, self.adam_v, lr_t, self.beta1, self.beta2, self.adam_eps, gscale)
        self.global_step += 1
        return lr

    # ------------------------------------------------------------------ fetches (host sync)
    def fetch_scalars(self):
        """(mean_loss, label_err): means over the n_valid rows of the batch (padding rows contribute 0 to both sums)."""
        s = self.scalars.cpu().numpy()
        n = max(self.n_valid, 1)
        return float(s[0]) / n, float(s[1]) / n

    def decoded_lists(self):
        ids = self.dec_ids.cpu().numpy()
        n = self.dec_len.cpu().numpy()
        return [ids[b, :n[b]].tolist() for b in range(self.n_valid)]

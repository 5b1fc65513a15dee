"""Transformer_Model.pre_net (end2end/model.py:214-264) on the HIP kernels of libasrhip.so: forward + backward + Adam.

    x [B, T, F=320] -> tanh conv s2 + BN -> tanh conv s2 + BN -> { q,k,v = BN(conv) ; time-axis and frequency-axis
    attention per channel ; conv(concat) + residual -> LayerNorm ; relu conv + BN ; conv + BN ; relu(. + LN out) }
    -> pre_out [B, T/4, (F/4)*64]      (what embedding_input flattens, model.py:268-272)

As the reference wires it: both iterations of the `for i in range(2)` loop read `input_x2` and only the last one
reaches `self.pre_out`, so ONE attention block is live; `mask=False` adds 0 (no masking); BatchNorm uses batch
moments (training=True), per replica under data parallelism.
Buffers follow DESIGN.md 2: padded planes [B][H+1][W+1][C] for everything a 3x3 conv reads or writes, the
"phase split" plane for the input of the 64->64 stride-2 conv, [B][T'][64][80] for the attention operands.
"""
import math

import numpy as np
import torch

from . import ops
from .ops import Plane

BN_EPS = 1e-3        # tf.layers.batch_normalization default
LN_EPS = 1e-8        # end2end/transformer.py layer_norm default
CH = 64

CONVS = [('conv1', 1, CH), ('conv2', CH, CH), ('q', CH, CH), ('k', CH, CH), ('v', CH, CH), ('merge', 2 * CH, CH),
         ('f1', CH, CH), ('f2', CH, CH)]
BNS = ['bn1', 'bn2', 'bnq', 'bnk', 'bnv', 'bnf1', 'bnf2']


def fwd_flops_per_seq(T, F=320):
    """Algorithmic forward flops of one sequence (2 x MACs)."""
    H1, W1, H2, W2 = T // 2, F // 2, T // 4, F // 4
    conv = 2 * 9 * (H1 * W1 * 1 * CH + H2 * W2 * CH * CH * 6 + H2 * W2 * 2 * CH * CH)
    attn = CH * (2 * 2 * H2 * H2 * W2 + 2 * 2 * W2 * W2 * H2)
    return conv + attn


class PreNetEngine:
    def __init__(self, B, T, F=320, lr=5e-4, beta2=0.98, seed=0, device='cuda', dual_stream=True, wino=True, s2_per_phase=True):
        assert F == 320, 'the attention kernels are built for 80 frequency bins after the two stride-2 convs (4 x 80 / 4)'
        assert T % 4 == 0 and T >= 4
        self.B, self.T, self.F, self.device = B, T, F, device
        # A/B switch: the stride-2 layer's data-gradient as one launch per phase (asr_conv_s2_dgrad, round 6) or as one 4-tap GEMM (False)
        self.opt_s2_per_phase = bool(s2_per_phase)
        self.H1, self.W1, self.H2, self.W2 = T // 2, F // 2, T // 4, F // 4
        self.lr0, self.beta1, self.beta2, self.adam_eps = lr, 0.9, beta2, 1e-8
        self.global_step = 0
        self.entries, off = {}, 0
        for name, cin, cout in CONVS:
            for k, shp in (('w', (3, 3, cin, cout)), ('b', (cout,))):
                self.entries['%s/%s' % (name, k)] = (off, shp); off += int(np.prod(shp))
        for name in BNS + ['ln']:
            for k in ('g', 'b'):
                self.entries['%s/%s' % (name, k)] = (off, (CH,)); off += CH
        z = lambda n, dt=torch.float32: torch.zeros(n, dtype=dt, device=device)
        self.theta, self.grad, self.adam_m, self.adam_v = z(off), z(off), z(off), z(off)
        self.init_params(seed)

        B_, H2, W2 = B, self.H2, self.W2
        pl = lambda c=CH: Plane(B_, H2, W2, c, device=device)
        self.a1 = z(B * self.H1 * self.W1 * CH).view(B, self.H1, self.W1, CH)
        self.x1s = pl(4 * CH)                 # BN(a1), phase split
        self.a2, self.x2 = pl(), pl()
        self.z = {k: pl() for k in 'qkv'}     # conv outputs (BN inputs)
        self.n = {k: pl() for k in 'qkv'}     # normalised q, k, v
        tl = lambda: z(B * H2 * CH * W2)
        self.QT, self.KT, self.VT, self.OT, self.OF = tl(), tl(), tl(), tl(), tl()
        self.lse = z(B * CH * H2)
        self.P = z(B * CH * W2 * W2)
        self.cat = pl(2 * CH)
        self.m, self.out, self.xhat = pl(), pl(), pl()
        self.ln_rstd = z(B * H2 * W2)
        self.af1, self.f1n, self.zf2 = pl(), pl(), pl()
        self.pre_out = z(B * H2 * W2 * CH).view(B, H2, W2, CH)
        self.stats = {name: (z(CH), z(CH)) for name in BNS}          # batch mean, 1/sqrt(var + eps)
        self.W4 = z(4 * 4 * CH * CH)           # 2x2-tap weights of the stride-2 conv
        # backward
        self.dsum, self.dz, self.dA, self.ds = pl(), pl(), pl(), pl()
        self.dcat = pl(2 * CH)
        self.dx1s = pl(4 * CH)
        self.dOT, self.dOF = tl(), tl()
        self.dT = [tl() for _ in range(6)]
        self.dS = z(B * CH * W2 * W2)
        self.dz1 = z(B * self.H1 * self.W1 * CH).view(B, self.H1, self.W1, CH)
        self.dW4 = z(4 * 4 * CH * CH)
        self.Wf9 = z(ops.conv_s2_arrange_bytes(CH) // 4)       # the stride-2 weights per (phase, tap) block, data-gradient fragment order
        NP = self.x2.NP
        self.d_c2 = ops.gemm_desc(NP, 4 * CH, CH, 4 * CH, CH, CH, 0, ntaps=4, B=B, H=H2, W=W2, relu=2)
        self.d_c2_dx = ops.gemm_desc(NP, CH, 4 * CH, CH, CH, 0, 4 * CH, ntaps=4, B=B, H=H2, W=W2, wmode=1)
        self.d_c2_dw = ops.gemm_desc(NP, 4 * CH, CH, 4 * CH, CH, ntaps=4, B=B, H=H2, W=W2)
        conv = lambda cin, act: ops.gemm_desc(NP, cin, CH, cin, CH, CH, 0, ntaps=9, B=B, H=H2, W=W2, relu=act)
        self.d_conv = {'q': conv(CH, 0), 'k': conv(CH, 0), 'v': conv(CH, 0), 'merge': conv(2 * CH, 0), 'f1': conv(CH, 1),
                       'f2': conv(CH, 0)}
        self.d_dx = {k: ops.gemm_desc(NP, CH, cin, CH, CH, 0, cin, ntaps=9, B=B, H=H2, W=W2, wmode=1)
                     for k, cin in (('q', CH), ('k', CH), ('v', CH), ('merge', 2 * CH), ('f1', CH), ('f2', CH))}
        self.d_dw = {k: ops.gemm_desc(NP, cin, CH, cin, CH, ntaps=9, B=B, H=H2, W=W2)
                     for k, cin in (('q', CH), ('k', CH), ('v', CH), ('merge', 2 * CH), ('f1', CH), ('f2', CH))}
        ws = max([ops.tap_wgrad_workspace(d) for d in self.d_dw.values()] + [ops.tap_wgrad_workspace(self.d_c2_dw),
                 ops.bn_workspace(self.a1), ops.bn_workspace(self.x2), ops.pix_ln_bwd_workspace(self.x2),
                 ops.prenet_conv1_bwd_workspace(B, T, F), ops.colsum_workspace(NP, CH), 4 * (B * CH * H2 + 64), 1 << 20])
        self.ws = z(ws // 4 + 64)
        # Second stream for the backward pass (dual_stream=False turns it off; see engine.py): the weight- and
        # bias-gradient of a conv run on it while the main stream goes on with the data-gradient and the next (HBM-bound)
        # BatchNorm / transpose / LayerNorm backward.  The pre-activation gradient planes alternate between two buffers.
        dual = bool(dual_stream)
        self.side = torch.cuda.Stream(device=device) if dual else None
        self.ws_side = z(ws // 4 + 64) if self.side is not None else None
        self.dz_alt = pl() if self.side is not None else None
        self._busy = {}            # id(plane) -> event of the last side-stream reader
        self._flip = False
        self._cin = {'q': CH, 'k': CH, 'v': CH, 'merge': 2 * CH, 'f1': CH, 'f2': CH}
        # the six 3x3 stride-1 convs on weights pre-arranged in MFMA fragment order (asr_tap_gemm_pw) ...
        self.wf_f, self.wf_b = {}, {}
        nb = ops._lib.load().asr_arrange_weights_bytes
        for k, cin in self._cin.items():
            self.wf_f[k] = torch.zeros(nb(9, cin, CH) // 4, dtype=torch.float32, device=device)
            self.wf_b[k] = torch.zeros(nb(9, CH, cin) // 4, dtype=torch.float32, device=device)
        # ... or, where supported, on the Winograd F(2x2,3x3) kernel (wino.hip; see engine.py), forward and data-gradients;
        # wino=False turns it off
        self.wt_f, self.wt_b = {}, {}
        if wino:
            for k, cin in self._cin.items():
                if ops.winograd_supported(self.d_conv[k]):
                    self.wt_f[k] = torch.zeros(ops.winograd_weights_floats(cin, CH), dtype=torch.float32, device=device)
                if ops.winograd_supported(self.d_dx[k]):
                    self.wt_b[k] = torch.zeros(ops.winograd_weights_floats(cin, CH), dtype=torch.float32, device=device)

    # ---- parameters
    def p(self, name, buf=None):
        off, shape = self.entries[name]
        return (self.theta if buf is None else buf)[off:off + int(np.prod(shape))]

    def g(self, name):
        return self.p(name, self.grad)

    def init_params(self, seed=0):
        """kernel_initializer='glorot_normal' (model.py:219-265), zero biases, BN / LN gamma 1, beta 0."""
        rng = np.random.default_rng(seed)
        flat = {}
        for name, cin, cout in CONVS:
            flat[name + '/w'] = rng.standard_normal((3, 3, cin, cout)) * math.sqrt(2.0 / (9 * cin + 9 * cout))
            flat[name + '/b'] = np.zeros(cout)
        for name in BNS + ['ln']:
            flat[name + '/g'], flat[name + '/b'] = np.ones(CH), np.zeros(CH)
        self.load_params(flat)

    def load_params(self, flat):
        host = self.theta.cpu().numpy()
        for name, (off, shp) in self.entries.items():
            v = np.asarray(flat[name], dtype=np.float32)
            assert tuple(v.shape) == tuple(shp), (name, v.shape, shp)
            host[off:off + v.size] = v.ravel()
        self.theta.copy_(torch.from_numpy(host))

    def grads_dict(self, buf=None):
        host = (self.grad if buf is None else buf).cpu().numpy()
        return {name: host[off:off + int(np.prod(shp))].reshape(shp).copy() for name, (off, shp) in self.entries.items()}

    def params_dict(self):
        return self.grads_dict(self.theta)

    # ---- helpers
    def _bn(self, name, src, dst, **kw):
        mean, rstd = self.stats[name]
        ops.bn_stats(src, BN_EPS, mean, rstd, self.ws)
        ops.bn_apply(src, mean, rstd, self.p(name + '/g'), self.p(name + '/b'), dst, **kw)

    def _next_dz(self):
        """The pre-activation gradient plane for the next conv: alternates when the side stream is on."""
        if self.side is None:
            return self.dz
        self._flip = not self._flip
        return self.dz_alt if self._flip else self.dz

    def _wait_readers(self, plane):
        ev = self._busy.pop(id(plane), None)
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)

    def _bn_bwd(self, name, dy, a, act, dz, **kw):
        mean, rstd = self.stats[name]
        if isinstance(dz, Plane):
            self._wait_readers(dz)                  # an earlier weight-gradient may still read this plane
        ops.bn_bwd(dy, a, mean, rstd, self.p(name + '/g'), act, dz, self.g(name + '/g'), self.g(name + '/b'), self.ws, **kw)

    def _on_side(self, reads, fn):
        """Run fn(workspace) on the side stream once the main stream's work so far is done; `reads` = planes it reads that
        the main stream will overwrite later."""
        if self.side is None:
            fn(self.ws)
            return
        ready = torch.cuda.Event()
        ready.record()
        self.side.wait_event(ready)
        with torch.cuda.stream(self.side):
            fn(self.ws_side)
            done = torch.cuda.Event()
            done.record()
        for p in reads:
            self._busy[id(p)] = done
        self._last_side = done

    def _conv(self, name, src, dst):
        if name in self.wt_f:
            ops.tap_gemm_wino(self.d_conv[name], src, self.wt_f[name], self.p(name + '/b'), None, None, dst, None)
        elif name in self.wf_f:
            ops.tap_gemm_pw(self.d_conv[name], src, self.wf_f[name], self.p(name + '/b'), None, None, dst, None)
        else:
            ops.tap_gemm(self.d_conv[name], src, self.p(name + '/w'), self.p(name + '/b'), None, None, dst, None)

    def _conv_bwd(self, name, src, dz, dx, accumulate):
        """parameter gradients of conv `name` (input plane src, pre-activation gradient dz) and dx (+)= its data gradient"""
        def grads(ws):
            ops.tap_wgrad(self.d_dw[name], src, dz, CH, self.g(name + '/w'), ws)
            ops.colsum(dz.body, dz.NP, CH, CH, self.g(name + '/b'), ws)
        self._on_side([dz], grads)
        if dx is not None:
            self._wait_readers(dx)                  # dx may be a plane an earlier weight-gradient still reads (ds)
            d = self.d_dx[name]
            d.accumulate = 1 if accumulate else 0
            if name in self.wt_b:
                ops.tap_gemm_wino(d, dz, self.wt_b[name], None, None, None, None, dx)
            elif name in self.wf_b:
                ops.tap_gemm_pw(d, dz, self.wf_b[name], None, None, None, None, dx)
            else:
                ops.tap_gemm(d, dz, self.p(name + '/w'), None, None, None, None, dx)

    # ---- forward / backward
    def forward(self, x):
        """x: float32 [B, T, F] on the device -> pre_out [B, T/4, 80*64] (a view of an engine buffer)."""
        B, H2, W2 = self.B, self.H2, self.W2
        assert tuple(x.shape) == (B, self.T, self.F) and x.is_contiguous() and x.dtype == torch.float32
        self.x = x
        for k, buf in self.wf_f.items():
            cin = self._cin[k]
            ops.arrange_weights(self.p(k + '/w'), 9, cin, CH, CH, 0, buf)
            ops.arrange_weights(self.p(k + '/w'), 9, CH, cin, CH, 1, self.wf_b[k])
        for k, buf in self.wt_f.items():
            ops.winograd_weights(self.p(k + '/w'), self._cin[k], CH, CH, 0, buf)
        for k, buf in self.wt_b.items():
            ops.winograd_weights(self.p(k + '/w'), CH, self._cin[k], CH, 1, buf)
        ops.prenet_conv1_fwd(x, self.p('conv1/w'), self.p('conv1/b'), self.a1)
        self._bn('bn1', self.a1, self.x1s, dst_phase_split=True)
        ops.conv_s2_expand(self.p('conv2/w'), CH, CH, self.W4)
        if self.opt_s2_per_phase:
            ops.conv_s2_arrange(self.W4, CH, self.Wf9)
        ops.tap_gemm(self.d_c2, self.x1s, self.W4, self.p('conv2/b'), None, None, self.a2, None)
        self._bn('bn2', self.a2, self.x2)
        for k, dstT in (('q', self.QT), ('k', self.KT), ('v', self.VT)):
            self._conv(k, self.x2, self.z[k])
            self._bn('bn' + k, self.z[k], self.n[k])
            ops.plane_to_T(self.n[k], 0, dstT)
        ops.attention_nomask_fwd(self.QT, self.KT, self.VT, B, H2, H2, CH * W2, CH, self.OT, self.lse)
        ops.freq_attention_fwd(self.QT, self.KT, self.VT, B, H2, self.P, self.OF)
        ops.T_to_plane(self.OT, None, self.cat, 0)
        ops.T_to_plane(self.OF, None, self.cat, CH)
        self._conv('merge', self.cat, self.m)
        ops.pix_add_ln_fwd(self.m, self.x2, self.p('ln/g'), self.p('ln/b'), LN_EPS, self.out, self.xhat, self.ln_rstd)
        self._conv('f1', self.out, self.af1)
        self._bn('bnf1', self.af1, self.f1n)
        self._conv('f2', self.f1n, self.zf2)
        self._bn('bnf2', self.zf2, self.pre_out, res=self.out, relu=True)
        return self.pre_out.view(B, H2, W2 * CH)

    def backward(self, d_pre):
        """d_pre: dL/d(pre_out) [B, T/4, 80*64]; fills self.grad (=)."""
        B, H2, W2 = self.B, self.H2, self.W2
        d_pre = d_pre.view(B, H2, W2, CH)
        ops.relu_mask(d_pre, self.pre_out, self.dsum)                       # d(f2n) = d(out) so far
        dz = self._next_dz()
        self._bn_bwd('bnf2', self.dsum, self.zf2, 0, dz)
        self._conv_bwd('f2', self.f1n, dz, self.dA, False)                  # dA = d(f1n)
        dz = self._next_dz()
        self._bn_bwd('bnf1', self.dA, self.af1, 1, dz)
        self._conv_bwd('f1', self.out, dz, self.dsum, True)                 # dsum = d(out)
        self._wait_readers(self.ds)
        ops.pix_ln_bwd(self.dsum, self.xhat, self.ln_rstd, self.p('ln/g'), self.ds, self.g('ln/g'), self.g('ln/b'), self.ws)
        # ds = d(merge conv output) = the residual's share of d(x2); later contributions accumulate into it
        self._conv_bwd('merge', self.cat, self.ds, self.dcat, False)
        ops.plane_to_T(self.dcat, 0, self.dOT)
        ops.plane_to_T(self.dcat, CH, self.dOF)
        dq1, dk1, dv1, dq2, dk2, dv2 = self.dT
        ops.attention_nomask_bwd(self.QT, self.KT, self.VT, self.OT, self.dOT, self.lse, B, H2, H2, CH * W2, CH,
                                 dq1, dk1, dv1, self.ws)
        ops.freq_attention_bwd(self.QT, self.KT, self.VT, self.P, self.dOF, B, H2, dq2, dk2, dv2, self.dS)
        for k, g1, g2 in (('q', dq1, dq2), ('k', dk1, dk2), ('v', dv1, dv2)):
            ops.T_to_plane(g1, g2, self.dA, 0)                              # d(normalised q/k/v)
            dz = self._next_dz()
            self._bn_bwd('bn' + k, self.dA, self.z[k], 0, dz)
            self._conv_bwd(k, self.x2, dz, self.ds, True)                   # ds = d(x2)
        dz = self._next_dz()
        self._bn_bwd('bn2', self.ds, self.a2, 2, dz)

        def grads2(ws):
            ops.tap_wgrad(self.d_c2_dw, self.x1s, dz, CH, self.dW4, ws)
            ops.conv_s2_gather(self.dW4, CH, CH, self.g('conv2/w'))
            ops.colsum(dz.body, dz.NP, CH, CH, self.g('conv2/b'), ws)
        self._on_side([dz], grads2)
        if self.opt_s2_per_phase:
            ops.conv_s2_dgrad(self.d_c2_dx, dz, self.Wf9, self.dx1s)    # one launch per phase of the input gradient (round 6)
        else:
            ops.tap_gemm(self.d_c2_dx, dz, self.W4, None, None, None, None, self.dx1s)
        self._bn_bwd('bn1', self.dx1s, self.a1, 2, self.dz1, dy_phase_split=True)
        ops.prenet_conv1_bwd(self.x, self.dz1, self.g('conv1/w'), self.g('conv1/b'), self.ws)
        if self.side is not None and getattr(self, '_last_side', None) is not None:
            torch.cuda.current_stream().wait_event(self._last_side)        # all gradients are in self.grad for Adam / all-reduce
            self._busy.clear()

    def apply_adam(self, lr, gscale=1.0):
        t = self.global_step + 1
        lr_t = lr * math.sqrt(1.0 - self.beta2 ** t) / (1.0 - self.beta1 ** t)
        ops.adam_tf(self.theta, self.grad, self.adam_m, self.adam_v, lr_t, self.beta1, self.beta2, self.adam_eps, gscale)
        self.global_step += 1

"""Joint acoustic + language model training graph of lm_and_am/model/am_lm_model.py (BASELINE configs[4])
on the existing kernels.  The reference file does not run as written; the graph built here is its evident one, with the
forced deviations D1-D5 and the kept-as-written oddities K1-K3 listed in DESIGN.md section 10 (and in oracle/amlm.py,
the CPU restatement the parity tests compare against):

  acoustic half  (am_lm_model.py:56-80)   DFCNNEngine graph 'amlm' -> h7 = dense(128, relu) -> dense(V_pinyin, softmax)
                                          -> log(transpose + 1e-7) -> CTC(dense pinyin labels, blank V_pinyin - 1)
  language half  (:82-131)                lm_in = am_out := h7 (D1) + position embedding -> dropout -> num_blocks x
                                          non-causal self-attention MHA -> one live FFN (K3) -> dense(V_hanzi, softmax)
                                          -> log(transpose + 1e-7) -> CTC(the SAME pinyin labels and blank, K2)
  loss (:145-154)                         mean_loss = am_mean_loss + lm_mean_loss
  optimiser (:133-139)                    Adam(am_lr, 0.9, 0.999, 1e-8), constant learning rate, over both halves

The gradient of the language half reaches the acoustic trunk through h7 (DFCNNEngine.backward(extra=...)).
No arithmetic happens here: every tensor op is a libasrhip launch.
"""
import numpy as np
import torch

from . import ops
from .engine import DFCNNEngine, K_EPSILON, MAX_LABEL
from .transformer_engine import _Base, _r4

NEG_PAD = -1.0e30       # bias of the padding columns of the hanzi projection: exp() = 0, they drop out of the softmax


class _LanguageHalf(_Base):
    def __init__(self, v_hanzi, blank, N, T, C, heads, blocks, pos_max, lr, seed, device, dropout_rate, drop_seed):
        # constant learning rate: polynomial decay from lr to lr
        super().__init__(C, heads, device, lr, 0.999, 5000, lr, dropout_rate, drop_seed)
        assert T <= pos_max, 'positions >= position_max_length index past the table (am_lm_model.py:85-87)'
        self.V, self.Vp, self.N, self.T, self.blocks, self.blank = v_hanzi, _r4(v_hanzi), N, T, blocks, blank
        self._add('pos', (pos_max, C))
        for i in range(blocks):
            self._add_mha('mha%d' % i)
        self._add_ffn('ffn')
        self._add('out_w', (C, v_hanzi), (C, self.Vp)); self._add('out_b', (v_hanzi,), (self.Vp,))
        self._finish_params()
        self.init_params(seed)
        rows = N * T
        self.tidx = torch.arange(T, dtype=torch.int32, device=device).repeat(N).contiguous()   # unused by the kernel (table = NULL)
        self.x0 = self._t(rows, C)
        self.mha = [self._mha_alloc(N, T, T) for _ in range(blocks)]
        self.ffn = self._ffn_alloc(rows)
        self.d = self._t(rows, self.Vp)                 # dense pre-activations [B][T][Vp]
        self.dd = self._t(rows, self.Vp)
        self.logits = self._t(T, N, self.Vp)            # log(softmax + 1e-7), time-major
        self.ctc_grad = self._t(T, N, self.Vp)
        self.loss = self._t(N)
        self.loss_sum = self._t(4)
        self.ctc_status = self._t(N, dtype=torch.int32)
        self.ctc_ws = torch.zeros(ops.ctc_workspace(T, N, MAX_LABEL) // 8 + 8, dtype=torch.float64, device=device)
        self.dec_ids = self._t(N, T, dtype=torch.int32)
        self.dec_len = self._t(N, dtype=torch.int32)
        self.dec_ws = self._t(ops.ctc_greedy_workspace(T, N) // 4 + 4, dtype=torch.int32)
        self.neg_sum = self._t(N)
        self.dstream = [self._t(rows * C), self._t(rows * C)]
        self._alloc_scratch(rows, max(C * self.Vp, 4 * C * C),
                            [(rows, C, C), (rows, C, 4 * C), (rows, 4 * C, C), (rows, C, self.Vp)])

    def load_params(self, flat):
        super().load_params(flat)
        self._pad_bias()

    def _pad_bias(self):
        if self.Vp > self.V:
            self.p('out_b')[self.V:].fill_(NEG_PAD)     # set once; its gradient is exactly 0 (softmax weight 0), so Adam leaves it

    def flat_from_oracle(self, L):
        flat = {'pos': L['pos'], 'out_w': L['out_w'], 'out_b': L['out_b']}
        for i in range(self.blocks):
            for k, v in L['mha%d' % i].items():
                flat['mha%d/%s' % (i, k)] = v
        for k, v in L['ffn'].items():
            flat['ffn/%s' % k] = v
        return flat

    def forward(self, h7, train=True):
        self._begin_forward()
        N, T, C = self.N, self.T, self.C
        rows = N * T
        ops.embed_fwd(None, self.tidx, self.p('pos'), N, T, C, False, 1.0, self.x0)       # x0 = pos[t]
        ops.axpy(self.x0, h7, 1.0, True)                                                  # + am_out (D1: h7)
        self._rate = self.dropout_rate if train else 0.0
        self._seed_emb = self._drop_seed('emb')
        if self._rate > 0:
            ops.dropout(self.x0, self._rate, self._seed_emb)                              # am_lm_model.py:91-93
        enc = self.x0
        for i in range(self.blocks):
            enc = self._mha_fwd('mha%d' % i, self.mha[i], enc, enc, False)
        out = self._ffn_fwd('ffn', self.ffn, enc)
        self._dense(out, rows, C, self.Vp, self.p('out_w'), self.p('out_b'), self.d, False)
        ops.softmax_log_fwd(self.d, N, T, self.Vp, K_EPSILON, self.logits)
        return self.logits

    def loss_and_decode(self, labels, label_len, seq_len):
        N, T = self.N, self.T
        ops.ctc_loss(self.logits, T, N, self.Vp, labels, MAX_LABEL, label_len, seq_len, self.blank,
                     self.loss, self.ctc_grad, self.ctc_status, self.ctc_ws)
        ops.colsum(self.loss, N, 1, 1, self.loss_sum[0:1], self.ws)
        ops.ctc_greedy(self.logits, T, N, self.Vp, seq_len, self.blank, self.dec_ids, self.dec_len, self.neg_sum, self.dec_ws)

    def backward(self):
        """-> d(mean lm loss)/d(h7) as a [rows, C] device tensor (a view of an engine buffer)."""
        N, T, C = self.N, self.T, self.C
        rows = N * T
        self.grad.zero_()
        self._written = set()
        ops.softmax_log_bwd(self.logits, self.ctc_grad, N, T, self.Vp, K_EPSILON, 1.0 / N, self.dd)
        d0, d1 = self.dstream
        out = self.ffn['out']
        self._wgrad(out, self.dd, rows, C, self.Vp, 'out_w')
        self._bgrad(self.dd, rows, self.Vp, 'out_b')
        self._dense_dgrad(self.dd, rows, C, self.Vp, self.p('out_w'), d0, False)
        self._ffn_bwd('ffn', self.ffn, d0, d1, False)
        cur, nxt = d1, d0
        for i in reversed(range(self.blocks)):
            self._mha_bwd('mha%d' % i, self.mha[i], cur, nxt, False, nxt, True)
            cur, nxt = nxt, cur
        if self._rate > 0:
            ops.dropout(cur, self._rate, self._seed_emb)
        ops.colsum(cur, N, T * C, T * C, self.g('pos')[:T * C], self.ws)
        self._ln_reduce()
        return cur.view(rows, C)


class AMLMEngine:
    """CNNCTCModel of am_lm_model.py: one training step = forward(x) -> set_targets -> loss_and_decode -> backward ->
    apply_adam.  ``hidden`` (= hidden_units = the width of h7) must be heads * 64."""

    def __init__(self, v_pinyin=1536, v_hanzi=6345, B=8, T=1600, F=200, widths=None, heads=2, blocks=6, pos_max=200,
                 lr=7e-4, seed=0, device='cuda', dropout_rate=0.0, drop_seed=0, am_options=None):
        """``am_options``: A/B switches handed to the acoustic half's DFCNNEngine (dual_stream, wino, ...)."""
        self.B, self.T8 = B, T // 8
        self.am = DFCNNEngine(model='amlm', vocab=v_pinyin, B=B, T=T, F=F, widths=widths, seed=seed, device=device,
                              lr=lr, decay_steps=5000, min_lr=lr, **(am_options or {}))   # constant lr (:134)
        hid = self.am.flat['h7'].shape[1]
        heads = heads or hid // 64
        assert hid == 64 * heads, 'h7 (%d wide) is the language half\'s hidden_units: heads must be %d' % (hid, hid // 64)
        self.lm = _LanguageHalf(v_hanzi, v_pinyin - 1, B, self.T8, hid, heads, blocks, pos_max, lr, seed + 1, device,
                                dropout_rate, drop_seed)
        self.lm._pad_bias()
        self.hanzi = torch.zeros(B, MAX_LABEL, dtype=torch.int32, device=device)
        self.hanzi_len = torch.zeros(B, dtype=torch.int32, device=device)
        self.han_dist = torch.zeros(B, dtype=torch.float32, device=device)
        self._has_hanzi = False

    def forward(self, x, train=True):
        """x [B, T, F] float32 on the device -> (am_logits [T/8, B, V_pinyin], lm_logits [T/8, B, Vp_hanzi]), time-major."""
        self.am.forward(x)
        self.lm.forward(self.am.flat['h7'], train)
        return self.am.logits, self.lm.logits

    def set_targets(self, wav_length, target_py, target_py_length, target_hanzi=None):
        self.am.set_targets(wav_length, target_py, target_py_length)          # K1: dense labels, zeros kept
        # han_wer (:121-123) compares the language half's decode with dense_to_sparse(target_hanzi): zeros dropped
        self._has_hanzi = target_hanzi is not None
        if self._has_hanzi:
            th = np.asarray(target_hanzi)
            lab = np.zeros((self.B, MAX_LABEL), dtype=np.int32)
            ll = np.zeros(self.B, dtype=np.int32)
            for b in range(self.B):
                ids = th[b][th[b] != 0][:MAX_LABEL]
                lab[b, :len(ids)] = ids
                ll[b] = len(ids)
            self.hanzi.copy_(torch.from_numpy(lab), non_blocking=True)
            self.hanzi_len.copy_(torch.from_numpy(ll), non_blocking=True)

    def loss_and_decode(self):
        self.am.loss_and_decode()
        self.lm.loss_and_decode(self.am.labels, self.am.label_len, self.am.seq_len)       # K2: same labels and blank
        if self._has_hanzi:
            ops.edit_distance(self.lm.dec_ids, self.T8, self.lm.dec_len, self.hanzi, MAX_LABEL, self.hanzi_len, self.B, self.han_dist)
            ops.colsum(self.han_dist, self.B, 1, 1, self.lm.loss_sum[1:2], self.lm.ws)

    def han_wer(self):
        return float(self.lm.loss_sum.cpu().numpy()[1]) / self.B

    def backward(self, reducers=None):
        """``reducers`` = (BucketedAllReduce over lm.grad, BucketedAllReduce over am.grad in engine.param_layout's three buckets):
        under data parallelism the language half's gradients are summed while the acoustic backward runs, the acoustic dense
        head as soon as it is final, the rest at the end (bench.py --workload am_lm)."""
        dh7 = self.lm.backward()
        if reducers is None:
            self.am.backward(extra={'h7': dh7})
            return
        red_lm, red_am = reducers
        red_lm.launch(0)
        self.am.backward(on_dense_grads_ready=lambda: red_am.launch(0), extra={'h7': dh7})
        red_am.launch(1); red_am.launch(2)
        red_lm.wait(); red_am.wait()

    def make_reducers(self):
        from .parallel import BucketedAllReduce
        am, lm = self.am, self.lm
        return (BucketedAllReduce(lm.grad, [(0, lm.grad.numel())]),
                BucketedAllReduce(am.grad, [(am.n_gamma, am.dense_end), (0, am.n_gamma), (am.dense_end, am.grad.numel())]))

    def apply_adam(self, gscale=1.0):
        self.lm.apply_adam(gscale)
        return self.am.apply_adam(gscale)

    def fetch(self):
        """-> (am_mean_loss, lm_mean_loss, mean_loss, label_err of the acoustic half)"""
        am_mean, err = self.am.fetch_scalars()
        lm_mean = float(self.lm.loss_sum.cpu().numpy()[0]) / self.B
        return am_mean, lm_mean, am_mean + lm_mean, err

    def decoded_lists(self):
        ids, n = self.lm.dec_ids.cpu().numpy(), self.lm.dec_len.cpu().numpy()
        return self.am.decoded_lists(), [ids[b, :n[b]].tolist() for b in range(self.B)]

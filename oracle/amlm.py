"""Oracle for the joint acoustic + language model graph of lm_and_am/model/am_lm_model.py (float64 numpy).
TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (no TensorFlow offline; the reference holds no fixtures) -- and more than
that: the reference file does NOT RUN as written, so this is a restatement of its evident graph with every deviation
listed here and in DESIGN.md section 10.

As written (am_lm_model.py):
  am_model  :56-80   DFCNN with NiN cells (oracle/dfcnn.py graph 'amlm') -> h7 = dense(128, relu) -> dense(V_pinyin, softmax)
                     -> log(transpose + 1e-7) -> tf.nn.ctc_loss_v2(self.target_py (DENSE), am_logits, target_py_length,
                     wav_length, blank = V_pinyin - 1) -> am_mean_loss
  language_model :82-131  lm_in = self.am_out; enc = lm_in + position embedding; dropout; num_blocks x
                     [enc = multihead_attention(enc, enc, causality=False); self.outputs = feedforward(enc)];
                     dense(V_hanzi, softmax) -> log(transpose + 1e-7) -> ctc_loss_v2(self.target_py, lm_logits,
                     target_py_length, wav_length, blank = V_pinyin - 1) -> lm_mean_loss
  calc_loss :145-154 mean_loss = am_mean_loss + lm_mean_loss;  opt_init :133-139 Adam(am_lr, 0.9, 0.999, 1e-8), constant lr

Deviations forced by the source not running (same numbering as DESIGN.md section 10):
  D1  `self.am_out` is never assigned (:84).  The only [B, 200, ...] tensor the acoustic half leaves behind that can be
      added to a position embedding is h7 (:64, width 128), so am_out := h7 and hidden_units := 128; with the 64-wide
      heads of the attention kernels that makes num_heads = 2 (hparams.py's 512 / 8 cannot be added to a 128-wide tensor).
  D2  position_max_length (100) is smaller than the 200 positions it is indexed with (:85-87): the table gets T/8 rows.
  D3  opt_init() re-creates global_step (:135) -- one counter here.
  D4  ctc_greedy_decoder is given target_py_length as sequence length (:75,120); the metrics here decode over wav_length
      instead (summaries only, no effect on losses or gradients).
  D5  feedforward(self.enc, num_units=[...]) (:108) omits the required dropout_rate / is_training arguments
      (transformer.py:204): the model's own values are used, as language_model.py does.
Kept AS WRITTEN although they look unintended (they define the arithmetic):
  K1  both CTC losses take the DENSE pinyin labels and target_py_length: the first target_py_length ids of each row, zeros
      included (ctc_loss_v2 with dense labels; same rule as K.ctc_batch_cost in cnn_ctc.py), NOT dense_to_sparse.
  K2  the language-model CTC is computed against the PINYIN labels with blank = V_pinyin - 1 inside the V_hanzi-wide softmax
      (:117-118); target_hanzi only feeds the han_wer summary.
  K3  each block's feedforward overwrites self.outputs and does not feed the next block (:108): one live FFN, as in
      language_model.py.
"""
import numpy as np

from . import ctc, dfcnn, nn
from . import transformer as tr


def init_params(v_pinyin, v_hanzi, feat=200, widths=None, heads=2, blocks=2, pos_max=200, seed=0, perturb=False):
    ops = dfcnn.graph('amlm', v_pinyin, widths, feat)
    C = ops[-2][4]                                   # width of h7 = hidden_units (D1)
    assert C == 64 * heads
    rng = np.random.default_rng(seed + 1)
    lm = {'pos': tr._glorot(rng, (pos_max, C)), 'out_w': tr._glorot(rng, (C, v_hanzi)), 'out_b': np.zeros(v_hanzi)}
    if perturb:
        lm['out_b'] = 0.1 * rng.standard_normal(v_hanzi)
    for i in range(blocks):
        lm['mha%d' % i] = tr.init_mha(rng, C, perturb)
    lm['ffn'] = tr.init_ffn(rng, C, 4 * C, perturb)
    return {'am': dfcnn.init_params(ops, seed, perturb), 'lm': lm}, ops


def dense_labels(target_py, target_len):
    """K1: the first target_len ids of each row, zeros kept."""
    return [list(np.asarray(target_py[b])[:int(target_len[b])]) for b in range(len(target_len))]


def train_step(P, ops, x, wav_length, target_py, target_len, heads, blocks, want_grads=True, drop=None):
    """x [B, T, F, 1]; wav_length [B] (<= T/8); target_py [B, <=64] zero padded, target_len [B].
    Returns am/lm logits (time-major), the two mean losses, mean_loss and all gradients ({'am': ..., 'lm': ...})."""
    d_am, state = dfcnn.forward(ops, P['am'], x)
    acts = state[0]
    B = d_am.shape[0]
    Vp = d_am.shape[-1]
    labels = dense_labels(target_py, target_len)
    am_logits = nn.log_softmax_eps_tm(d_am)
    am_loss, g_am = ctc.ctc_loss_and_grad(am_logits, labels, list(wav_length), blank=Vp - 1)
    # ---- language half on h7
    L = P['lm']
    h7 = acts['h7']                                          # [B, T8, C] (post-ReLU)
    T8, C = h7.shape[1], h7.shape[2]
    pos = np.broadcast_to(np.arange(T8)[None, :], (B, T8))
    enc = h7 + tr.embedding(L['pos'], pos, zero_pad=False, scale=False)
    m_emb = drop.mask(enc.shape, 'emb') if drop is not None else None
    if m_emb is not None:
        enc = enc * m_emb
    caches = []
    for i in range(blocks):
        enc, c = tr.mha_fwd(enc, enc, L['mha%d' % i], heads, causal=False, drop=drop, site=('mha', i))
        caches.append(c)
    outputs, c_ffn = tr.ffn_fwd(enc, L['ffn'], drop, 'ffn')
    d_lm = outputs @ L['out_w'] + L['out_b']
    lm_logits = nn.log_softmax_eps_tm(d_lm)
    lm_loss, g_lm = ctc.ctc_loss_and_grad(lm_logits, labels, list(wav_length), blank=Vp - 1)      # K2
    dec_am, _ = ctc.ctc_greedy_decode(am_logits, list(wav_length))
    dec_lm, _ = ctc.ctc_greedy_decode(lm_logits, list(wav_length))
    out = {'am_logits': am_logits, 'lm_logits': lm_logits, 'am_loss': am_loss, 'lm_loss': lm_loss,
           'am_mean_loss': float(am_loss.mean()), 'lm_mean_loss': float(lm_loss.mean()),
           'mean_loss': float(am_loss.mean() + lm_loss.mean()), 'h7': h7, 'decoded_am': dec_am, 'decoded_lm': dec_lm}
    if not want_grads:
        return out
    # ---- backward: language half first (its gradient reaches the acoustic trunk through h7)
    dd_lm = nn.log_softmax_eps_tm_bwd(d_lm, g_lm / B)
    G = {'out_w': outputs.reshape(-1, C).T @ dd_lm.reshape(B * T8, -1), 'out_b': dd_lm.reshape(B * T8, -1).sum(axis=0)}
    d = dd_lm @ L['out_w'].T
    d, G['ffn'] = tr.ffn_bwd(c_ffn, L['ffn'], d)
    for i in reversed(range(blocks)):
        d, _, G['mha%d' % i] = tr.mha_bwd(caches[i], L['mha%d' % i], d, self_attn=True)
    if m_emb is not None:
        d = d * m_emb
    G['pos'] = tr.embedding_bwd(L['pos'].shape, pos, d, zero_pad=False, scale=False)
    dd_am = nn.log_softmax_eps_tm_bwd(d_am, g_am / B)
    out['grads'] = {'lm': G, 'am': dfcnn.backward(ops, P['am'], state, dd_am, extra={'h7': d})}
    out['dh7_from_lm'] = d
    return out

"""Oracle for CTC loss / gradient, greedy decoding and edit distance
(float64 numpy + plain Python loops).  Test infrastructure only; parity
unpinned (see oracle/__init__.py).

Restates:
  tf.nn.ctc_loss_v2(sparse_labels, logits, ..., blank_index=V-1)   acoustic_model2.py:79-80
  tf.contrib.layers.dense_to_sparse (drops every 0)                acoustic_model2.py:71
  tf.nn.ctc_greedy_decoder(logits, seq_len)                        acoustic_model2.py:69
  tf.edit_distance(hyp, truth, normalize=True) + reduce_mean       acoustic_model2.py:72-73
"""
import numpy as np

NEG_INF = -np.inf


def dense_to_sparse(target):
    """Rows of a zero-padded int matrix -> list of label lists with every 0 removed."""
    return [[int(v) for v in row if v != 0] for row in np.asarray(target)]


def _lse(*xs):
    m = max(xs)
    if m == NEG_INF:
        return NEG_INF
    return m + np.log(sum(np.exp(x - m) for x in xs))


def ctc_loss_and_grad(logits_tm, labels, seq_lens, blank):
    """logits_tm [T,B,V] are *unnormalised* inputs (TF applies its own softmax).
    labels: list of B label lists.  Returns (loss [B], grad [T,B,V]) where grad
    is d loss_b / d logits_tm[:, b, :] (zero for t >= seq_len[b]).
    Raises ValueError when no alignment exists (TF: InvalidArgumentError
    "Not enough time for target transition sequence")."""
    x = np.asarray(logits_tm, dtype=np.float64)
    T, B, V = x.shape
    loss = np.zeros(B)
    grad = np.zeros_like(x)
    for b in range(B):
        Tb = int(seq_lens[b])
        lab = list(labels[b])
        L = len(lab)
        rep = sum(1 for i in range(1, L) if lab[i] == lab[i - 1])
        if Tb < L + rep or Tb <= 0:
            raise ValueError('Not enough time for target transition sequence '
                             '(required: %d, available: %d) in batch %d' % (L + rep, Tb, b))
        xb = x[:Tb, b, :]
        m = xb.max(axis=1, keepdims=True)
        lsm = xb - m - np.log(np.exp(xb - m).sum(axis=1, keepdims=True))   # log-softmax
        ext = [blank]
        for l in lab:
            ext += [l, blank]
        S = len(ext)
        alpha = np.full((Tb, S), NEG_INF)
        beta = np.full((Tb, S), NEG_INF)
        alpha[0, 0] = lsm[0, ext[0]]
        if S > 1:
            alpha[0, 1] = lsm[0, ext[1]]
        for t in range(1, Tb):
            for s in range(S):
                c = [alpha[t - 1, s]]
                if s >= 1:
                    c.append(alpha[t - 1, s - 1])
                if s >= 2 and ext[s] != blank and ext[s] != ext[s - 2]:
                    c.append(alpha[t - 1, s - 2])
                alpha[t, s] = _lse(*c) + lsm[t, ext[s]]
        beta[Tb - 1, S - 1] = lsm[Tb - 1, ext[S - 1]]
        if S > 1:
            beta[Tb - 1, S - 2] = lsm[Tb - 1, ext[S - 2]]
        for t in range(Tb - 2, -1, -1):
            for s in range(S):
                c = [beta[t + 1, s]]
                if s + 1 < S:
                    c.append(beta[t + 1, s + 1])
                if s + 2 < S and ext[s] != blank and ext[s] != ext[s + 2]:
                    c.append(beta[t + 1, s + 2])
                beta[t, s] = _lse(*c) + lsm[t, ext[s]]
        ll = _lse(alpha[Tb - 1, S - 1], alpha[Tb - 1, S - 2]) if S > 1 else alpha[Tb - 1, 0]
        loss[b] = -ll
        # grad_k = softmax_k - sum_{s: ext[s]=k} exp(alpha+beta - lsm_k - ll)
        g = np.exp(lsm)
        for t in range(Tb):
            for s in range(S):
                ab = alpha[t, s] + beta[t, s]
                if ab > NEG_INF:
                    g[t, ext[s]] -= np.exp(ab - lsm[t, ext[s]] - ll)
        grad[:Tb, b, :] = g
    return loss, grad


def ctc_greedy_decode(logits_tm, seq_lens, blank=None):
    """tf.nn.ctc_greedy_decoder(merge_repeated=True): returns (list of id lists,
    neg_sum_logits [B]); blank = V-1; argmax ties -> lowest index."""
    x = np.asarray(logits_tm)
    T, B, V = x.shape
    if blank is None:
        blank = V - 1
    out, neg = [], np.zeros(B, dtype=np.float64)
    for b in range(B):
        ids, prev = [], -1
        for t in range(int(seq_lens[b])):
            k = int(np.argmax(x[t, b]))
            neg[b] += -float(x[t, b, k])
            if k != blank and k != prev:
                ids.append(k)
            prev = k
        out.append(ids)
    return out, neg


def decoded_to_sparse(decoded):
    """list of id lists -> (indices [n,2] int64, values [n] int64, dense_shape [2])
    as tf.nn.ctc_greedy_decoder's SparseTensor."""
    idx, val = [], []
    for b, ids in enumerate(decoded):
        for j, v in enumerate(ids):
            idx.append((b, j))
            val.append(v)
    maxlen = max([len(d) for d in decoded] + [0])
    return (np.asarray(idx, dtype=np.int64).reshape(-1, 2), np.asarray(val, dtype=np.int64),
            np.asarray([len(decoded), maxlen], dtype=np.int64))


def levenshtein(a, b):
    a, b = list(a), list(b)
    prev = list(range(len(b) + 1))
    for i in range(1, len(a) + 1):
        cur = [i] + [0] * len(b)
        for j in range(1, len(b) + 1):
            cur[j] = min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (a[i - 1] != b[j - 1]))
        prev = cur
    return prev[len(b)]


def edit_distance_normalized(hyp, truth):
    """tf.edit_distance(normalize=True) for one pair."""
    d = levenshtein(hyp, truth)
    if len(truth) == 0:
        return float('inf') if len(hyp) > 0 else 0.0
    return d / len(truth)


def label_error_rate(decoded, truths):
    return float(np.mean([edit_distance_normalized(h, t) for h, t in zip(decoded, truths)]))


def get_edit_distance_difflib(s1, s2):
    """util/utils.py:43-53 -- the difflib-opcode cost the eval scripts use
    (NOT Levenshtein: a 'replace' block costs max(len_a, len_b))."""
    import difflib
    cost = 0
    for tag, i1, i2, j1, j2 in difflib.SequenceMatcher(None, s1, s2).get_opcodes():
        cost += {'replace': max(i2 - i1, j2 - j1), 'insert': j2 - j1,
                 'delete': i2 - i1, 'equal': 0}[tag]
    return cost

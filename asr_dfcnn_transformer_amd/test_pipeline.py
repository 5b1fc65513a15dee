"""Inference / evaluation pipeline: the counterpart of ``speech_test`` in lm_and_am/test.py:25-101.
Acoustic model greedy decode (pinyin ids) -> ``sparse_tensor_to_dense(default_value=0)`` -> language
model argmax (hanzi ids) -> accuracy bookkeeping with the difflib-based ``GetEditDistance``
(util/utils.py:43-53), capped at the sentence length exactly as test.py:76-90 does.
Batched: the reference decodes one utterance per session call; here a whole batch goes through
both engines (rows are independent)."""
import numpy as np

from .utils import GetEditDistance


def dense_from_sparse(sp, default_value=0):
    """tf.sparse_tensor_to_dense(decoded, default_value=0) (test.py:51)."""
    out = np.full(tuple(sp.dense_shape), default_value, dtype=np.int64)
    if len(sp.values):
        out[sp.indices[:, 0], sp.indices[:, 1]] = sp.values
    return out


class SpeechRecognizer:
    def __init__(self, am_model, lm_model, index2pinyin=None, index2word=None):
        self.am, self.lm = am_model, lm_model
        self.index2pinyin, self.index2word = index2pinyin, index2word

    def recognize(self, wav_input, logits_length):
        """wav_input [B, T_pad, F, 1] features, logits_length [B] -> (pinyin id lists, hanzi id lists)."""
        am, lm = self.am, self.lm
        decoded = am.run(am.decoded[0], {am.wav_input: wav_input, am.logits_length: logits_length})
        py_dense = dense_from_sparse(decoded)
        B = py_dense.shape[0]
        lens = [int((decoded.indices[:, 0] == b).sum()) for b in range(B)]
        if py_dense.shape[1] == 0:
            return [[] for _ in range(B)], [[] for _ in range(B)]
        han = lm.run(lm.preds, {lm.x: py_dense[:, :lm.position_max_length]})
        pinyin_ids = [py_dense[b, :lens[b]].tolist() for b in range(B)]
        han_ids = [han[b, :min(lens[b], han.shape[1])].tolist() for b in range(B)]
        return pinyin_ids, han_ids

    def to_text(self, pinyin_ids, han_ids):
        py = [' '.join(str(self.index2pinyin[k]) for k in ids) for ids in pinyin_ids] if self.index2pinyin else None
        hz = [''.join(str(self.index2word.get(k)) for k in ids) for ids in han_ids] if self.index2word else None
        return py, hz


class AccuracyMeter:
    """word accuracy ratio = 1 - errors / words with errors = min(GetEditDistance, len(truth)) (test.py:74-101)."""

    def __init__(self):
        self.words, self.errors = 0, 0

    def update(self, truth, pred):
        n = len(truth)
        d = GetEditDistance(list(truth), list(pred))
        self.words += n
        self.errors += d if d <= n else n

    @property
    def accuracy(self):
        return 1.0 - self.errors / self.words if self.words else 0.0

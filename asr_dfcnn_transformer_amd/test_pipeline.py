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


def speech_test(am_model, lm_model, dataloader, num, pred_dir=None, start=None, rng=None, verbose=True):
    """``speech_test`` of lm_and_am/test.py:25-101: ``num`` consecutive utterances of the test index, starting at a random one
    (:31), each decoded by the acoustic model (greedy CTC, densified with 0, :48-51) and the language model (argmax, :60-61);
    pinyin and hanzi word errors by the difflib ``GetEditDistance`` capped at the sentence length (:74-90); the transcript
    log and the two accuracy lines written to ``<pred_dir>/pred_log`` (:92-97).  An utterance that raises ValueError in the
    loader is skipped (:62-63).  Returns (pinyin word accuracy, hanzi word accuracy) as fractions."""
    import os
    import random
    from .const import Const
    num_data = len(dataloader.pny_lst)
    length = getattr(dataloader.data, 'data_length', None) or num_data
    ran_num = start if start is not None else (rng or random).randint(0, length - 1)
    rec = SpeechRecognizer(am_model, lm_model, dataloader.index2pinyin, dataloader.index2word)
    py_meter, han_meter = AccuracyMeter(), AccuracyMeter()
    data = ''
    for i in range(num):
        index = (ran_num + i) % num_data
        try:
            hanzi = str(dataloader.han_lst[index])
            hanzi_vec = [dataloader.word2index.get(word, Const.PAD) for word in hanzi]
            inputs, input_length, label, _ = dataloader.get_fbank_and_pinyin_data(index)
            pinyin_ids, han_ids = rec.recognize(inputs, input_length)
        except ValueError:
            continue
        py_text, han_text = rec.to_text(pinyin_ids, han_ids)
        y = str(dataloader.pny_lst[index])
        lines = ['原文汉字结果:' + hanzi, '原文拼音结果:' + y, '预测拼音结果:' + py_text[0], '预测汉字结果:' + han_text[0]]
        if verbose:
            print('\nthe ', i + 1, 'th example.')
            print('\n'.join(lines))
        data += '\n'.join(lines) + '\n'
        py_meter.update(list(label), pinyin_ids[0])
        han_meter.update(hanzi_vec, han_ids[0])
    tail = ['*[Test Result] Speech Recognition test set 拼音 word accuracy ratio: ' + str(py_meter.accuracy * 100) + '%',
            '*[Test Result] Speech Recognition test set 汉字 word accuracy ratio: ' + str(han_meter.accuracy * 100) + '%']
    data += ''.join(tail)
    if pred_dir is not None:
        os.makedirs(pred_dir, exist_ok=True)
        with open(os.path.join(pred_dir, 'pred_log'), 'w', encoding='utf-8') as f:
            f.writelines(data)
    if verbose:
        print('\n'.join(tail))
    return py_meter.accuracy, han_meter.accuracy

"""Batch assembly for the acoustic model: the reference's ``DataLoader`` surface
(lm_and_am/data_loader.py:19-41, 85-162, 246-280) on top of device fbank extraction.

Vocabulary loading reproduces the reference's pandas calls, so sizes match SURVEY Q10
(mixdict.txt -> 1536 incl. '_' blank, hanzi.txt -> 6345 incl. <pad>).  Audio decoding from
disk (soundfile) is out of scope; utterances arrive as float arrays in [-1, 1) -- from a
user-supplied ``read_audio`` callable or from the seeded synthetic source."""
import math
import os

import numpy as np

from .const import Const


def load_acoustic_vocab(pinyin_dict):
    """data_loader.py:85-92: first TSV column + '_' (the CTC blank, last id)."""
    import pandas as pd
    text = pd.read_table(pinyin_dict, header=None)
    symbols = text.iloc[:, 0].tolist()
    symbols.append('_')
    return len(symbols), {s: i for i, s in enumerate(symbols)}, {i: s for i, s in enumerate(symbols)}


def load_language_vocab(hanzi_dict):
    """data_loader.py:95-103: '<pad>' + the characters of hanzi.txt."""
    import pandas as pd
    data = pd.read_csv(hanzi_dict, header=None)
    words = [Const.PAD_FLAG] + data.T.values.tolist()[0]
    return len(words), {w: i for i, w in enumerate(words)}, {i: w for i, w in enumerate(words)}


def ctc_input_length(num_frames):
    """data_loader.py:132: min(200, T//8 + 1)."""
    return min(200, int(math.ceil(num_frames // 8 + 1)))


class SyntheticSource:
    """Seeded stand-in for DataUtil (util/data_util.py): 10 s / 16 kHz Gaussian 'audio' and
    random pinyin ids, as BASELINE.md section 3 prescribes."""

    FAULTS = ('long_audio', 'long_label', 'label_ge_input', 'unknown_token', 'long_hanzi')

    def __init__(self, n_utts, seconds=10.0, sample_rate=16000, label_len=32, vocab=1536, seed=1234, shuffle=False,
                 faults=None):
        """``faults`` = {utterance index: one of FAULTS} plants a sample that one of the reference's checks rejects
        (data_loader.py:139-147, pny2id / han2id KeyError -> ValueError): audio longer than feature_max_length frames,
        more than 64 labels, at least as many labels as CTC input frames, a token outside the dictionary, more than 64
        hanzi."""
        self.n, self.ns, self.sr, self.L, self.V, self.seed = n_utts, int(seconds * sample_rate), sample_rate, label_len, vocab, seed
        self.shuffle = shuffle
        self.faults = dict(faults or {})
        self.path_lst = ['synthetic_%06d' % i for i in range(n_utts)]
        rng = np.random.default_rng(seed)
        self.labels = rng.integers(1, vocab - 1, size=(n_utts, label_len))       # never 0 (Q6), never blank
        self.pny_lst = [' '.join(str(v) for v in row) for row in self.labels]
        self.han_lst = ['' for _ in range(n_utts)]
        for i, kind in self.faults.items():
            if kind not in self.FAULTS:
                raise ValueError(kind)
            if kind == 'long_label':
                self.pny_lst[i] = ' '.join(str(v) for v in rng.integers(1, vocab - 1, 65))
            elif kind == 'unknown_token':
                self.pny_lst[i] = self.pny_lst[i] + ' zzz9'
            elif kind == 'long_hanzi':
                self.han_lst[i] = '\u4e00' * 65            # 65 x the first hanzi of hanzi.txt

    def read_audio(self, path):
        i = int(path.rsplit('_', 1)[1])
        rng = np.random.default_rng(self.seed + i)
        ns = self.ns
        kind = self.faults.get(i)
        if kind == 'long_audio':
            ns = int(17.0 * self.sr)                 # 1699 frames > feature_max_length 1600
        elif kind == 'label_ge_input':
            ns = 160 * 8 * (self.L - 2)              # T//8 + 1 <= label_len
        return (0.1 * rng.standard_normal(ns)).astype(np.float32), self.sr


class DataLoader:
    def __init__(self, data_util, data_args, train_args, read_audio=None, device='cuda'):
        self.am_batch_size = train_args.am_batch_size
        self.lm_batch_size = train_args.lm_batch_size
        self.feature_dim = train_args.feature_dim
        self.feature_max_length = train_args.feature_max_length
        root = Const.DictFolder
        pd_path = data_args.pinyin_dict if os.path.isabs(data_args.pinyin_dict) else os.path.join(root, data_args.pinyin_dict)
        hz_path = data_args.hanzi_dict if os.path.isabs(data_args.hanzi_dict) else os.path.join(root, data_args.hanzi_dict)
        self.acoustic_vocab_size, self.pinyin2index, self.index2pinyin = load_acoustic_vocab(pd_path)
        self.language_vocab_size, self.word2index, self.index2word = load_language_vocab(hz_path)
        self.data = data_util
        self.path_lst, self.pny_lst, self.han_lst = data_util.path_lst, data_util.pny_lst, data_util.han_lst
        self.shuffle = data_util.shuffle
        self.indexes = list(range(len(self.path_lst)))
        self.read_audio = read_audio or data_util.read_audio
        self.device = device
        self._fbank = None

    def pny2id(self, line):
        # any lookup failure becomes the ValueError the callers catch (the reference's `except ValueError: raise
        # ValueError`, data_loader.py:52-53, lets the KeyError of an unknown token escape and end the epoch)
        try:
            if isinstance(self.data, SyntheticSource):
                return [int(t) for t in line.strip().split(' ')]
            return [self.pinyin2index[p] for p in line.strip().split(' ')]
        except Exception:
            raise ValueError

    def han2id(self, line):
        try:
            flags = {Const.PAD_FLAG: Const.PAD, Const.SOS_FLAG: Const.SOS, Const.EOS_FLAG: Const.EOS}
            return [flags[h] if h in flags else self.word2index[h] for h in line.strip()]
        except Exception:
            raise ValueError

    def __len__(self):
        return len(self.path_lst) // self.am_batch_size

    def data_generation(self, batch_datas, py_label_datas, han_label_datas):
        """data_loader.py:105-162.  Returns (wav [B',1600,200,1] float32 CUDA tensor, input_length,
        py labels [B',64], label_length, han labels [B',64], word_length); rows failing the
        reference's checks (T > 1600, L > 64, L >= input_length) are dropped."""
        import torch
        from .wav_util import FbankExtractor, num_frames
        if self._fbank is None:
            self._fbank = FbankExtractor(nfilt=self.feature_dim, device=self.device)
        keep, sigs, in_len, py, han = [], [], [], [], []
        for i, path in enumerate(batch_datas):
            try:
                signal, sr = self.read_audio(path)
                nf = num_frames(len(signal), self._fbank.frame_len, self._fbank.frame_step)
                data_length = ctc_input_length(nf)
                ids = self.pny2id(py_label_datas[i])
                hz = self.han2id(han_label_datas[i])
                if nf > self.feature_max_length or len(ids) > 64 or len(ids) >= data_length:
                    raise ValueError
                if len(hz) > 64:
                    # data_loader.py:147 `batch_han_data[i, 0:len(seq_ids)] = seq_ids` cannot broadcast -> numpy ValueError
                    # -> the row is deleted.  (The reference has by then appended the row's lengths, :143-145, so ITS
                    # length arrays run one entry ahead of the rows from there on; here rows and lengths stay aligned.)
                    raise ValueError
                keep.append(i); sigs.append(np.asarray(signal, dtype=np.float32))
                in_len.append(data_length); py.append(ids); han.append(hz)
            except ValueError:
                continue
        n = len(keep)
        batch_label = np.zeros((n, 64), dtype=np.int32)
        batch_han = np.zeros((n, 64), dtype=np.int32)
        for r in range(n):
            batch_label[r, :len(py[r])] = py[r]
            batch_han[r, :len(han[r])] = han[r]
        self.last_kept = keep
        if n == 0:                    # every row deleted: the reference returns arrays with a 0-row batch axis
            empty = torch.zeros(0, self.feature_max_length, self.feature_dim, 1, dtype=torch.float32, device=self.device)
            z = np.zeros(0, dtype=np.int64)
            return empty, z, batch_label, z, batch_han, z
        mx = max(len(s) for s in sigs)
        host = np.zeros((n, mx), dtype=np.float32)
        for r, s in enumerate(sigs):
            host[r, :len(s)] = s
        sig = torch.from_numpy(host).to(self.device)
        ns = torch.tensor([len(s) for s in sigs], dtype=torch.int32, device=self.device)
        feat, _ = self._fbank.batch(sig, ns, self.feature_max_length)
        return (feat.unsqueeze(-1), np.array(in_len), batch_label, np.array([len(p) for p in py]),
                batch_han, np.array([len(p) for p in py]))

    def get_lm_batch(self, rng=None, select=None):
        """Batches for the language model, pinyin ids -> hanzi ids (lm_and_am/data_loader.py:164-193): every
        lm_batch_size consecutive (or shuffled) transcripts, each row zero-padded to the longest pinyin sequence of the
        batch; yields (input_data [B', L] , input_length [B'], label_data [B', L]).  As in the reference, input_length is the
        CHARACTER length of the pinyin string (:190 ``len(self.pny_lst[i])``; nothing downstream reads it) and rows whose
        tokens are not in the dictionaries are skipped (:192).  A transcript whose hanzi count differs from its pinyin count
        would give a ragged row there (np.array of unequal lists); it is skipped here."""
        import random
        order = list(range(len(self.pny_lst)))
        if self.shuffle:
            (rng or random).shuffle(order)
        for k in range(len(self.pny_lst) // self.lm_batch_size):
            if select is not None and k not in select:           # another rank's batch (train.rank_batches)
                continue
            index_list = order[k * self.lm_batch_size:(k + 1) * self.lm_batch_size]
            max_len = max(len(str(self.pny_lst[i]).strip().split(' ')) for i in index_list)
            input_data, label_data, input_length = [], [], []
            for i in index_list:
                try:
                    py = self.pny2id(str(self.pny_lst[i]))
                    han = self.han2id(str(self.han_lst[i]))
                    if len(han) != len(py):
                        raise ValueError
                    input_data.append(py + [0] * (max_len - len(py)))
                    label_data.append(han + [0] * (max_len - len(han)))
                    input_length.append(len(str(self.pny_lst[i])))
                except ValueError:
                    continue
            yield (np.array(input_data, dtype=np.int32).reshape(len(input_data), max_len), np.array(input_length),
                   np.array(label_data, dtype=np.int32).reshape(len(label_data), max_len))

    def get_fbank_and_pinyin_data(self, index):
        """One utterance for the evaluation loop (lm_and_am/data_loader.py:217-245; test.py:47): (wav [1, 1600, F, 1] device
        tensor, data_length [1] = T // 8 + 1 -- NOT capped at 200 here, unlike data_generation :132 --, label ids, len_label).
        Raises ValueError for more than 64 labels or more labels than CTC frames (:238-239)."""
        import torch
        from .wav_util import FbankExtractor, num_frames
        if self._fbank is None:
            self._fbank = FbankExtractor(nfilt=self.feature_dim, device=self.device)
        signal, sr = self.read_audio(self.path_lst[index])
        signal = np.asarray(signal, dtype=np.float32)
        nf = num_frames(len(signal), self._fbank.frame_len, self._fbank.frame_step)
        data_length = nf // 8 + 1
        label = np.array(self.pny2id(str(self.pny_lst[index])))
        if nf > self.feature_max_length:          # wav_data[0, 0:len(input_data)] = input_data cannot broadcast (:231)
            raise ValueError
        if len(label) > 64 or len(label) > data_length:
            raise ValueError
        sig = torch.from_numpy(signal).reshape(1, -1).to(self.device)
        ns = torch.tensor([len(signal)], dtype=torch.int32, device=self.device)
        feat, _ = self._fbank.batch(sig, ns, self.feature_max_length)
        return feat.unsqueeze(-1), np.array([data_length]), label, len(label)

    def __getitem__(self, index):
        idx = self.indexes[index * self.am_batch_size:(index + 1) * self.am_batch_size]
        return self.data_generation([self.path_lst[k] for k in idx], [self.pny_lst[k] for k in idx],
                                    [self.han_lst[k] for k in idx])

    def am_generator(self):
        for i in range(len(self)):
            yield self[i]

    def end2end_generator(self):
        """lm_and_am/data_loader.py:257-266: the same six-tuple per batch, as the joint loop's tf.data generator
        (am_lm_train.py:47-49)."""
        for i in range(len(self)):
            yield self[i]

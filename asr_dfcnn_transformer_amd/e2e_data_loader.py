"""``dataloader`` of the end-to-end model (end2end/data_loader.py:13-302): hanzi targets with SOS / EOS, fbank (feature_dim
80) -> low-frame-rate stacking (lfr_m 4, lfr_n 3) -> zero padding to the batch's longest utterance, decoder inputs padded
with EOS, decoder targets padded with IGNORE (-1) (:263-302).  Feature extraction and stacking run on the GPU for the whole
batch (asr_fbank + asr_lfr); the index comes from a DataUtil (or any object with path_lst / pny_lst / han_lst / read_audio,
e.g. data_loader.SyntheticSource).

Deviation: a sample that raises (unknown hanzi) is dropped from all three arrays.  The reference records the BATCH number
instead of the sample's position (``error_count.append(i)``, :290) and np.delete()s that row of arrays that never contained
the failed sample -- a live bug there."""
import math
import os

import numpy as np

from .const import Const
from .data_loader import load_acoustic_vocab, load_language_vocab


class dataloader:
    def __init__(self, train_args, data_args, data_util=None, read_audio=None, device='cuda'):
        self.batch_size = train_args.batch_size
        self.feature_dim = train_args.feature_dim
        self.feature_max_length = getattr(train_args, 'feature_max_length', 2000)
        self.mode = getattr(train_args, 'mode', 'train')
        self.data_length = getattr(train_args, 'data_length', None)
        self.shuffle = getattr(train_args, 'shuffle', False)
        self.lfr_m, self.lfr_n = data_args.lfr_m, data_args.lfr_n
        root = Const.DictFolder
        pd_path = data_args.pinyin_dict if os.path.isabs(data_args.pinyin_dict) else os.path.join(root, data_args.pinyin_dict)
        hz_path = data_args.hanzi_dict if os.path.isabs(data_args.hanzi_dict) else os.path.join(root, data_args.hanzi_dict)
        self.acoustic_vocab_size, self.pinyin2index, self.inde2pinyin = load_acoustic_vocab(pd_path)
        self.language_vocab_size, self.word2index, self.index2word = self.get_language_vocab_list(hz_path)
        if data_util is None:
            from .data_util import DataUtil
            data_util = DataUtil(data_args, self.batch_size, self.mode, self.data_length, self.shuffle)
        self.data = data_util
        self.path_lst, self.pny_lst, self.han_lst = data_util.path_lst, data_util.pny_lst, data_util.han_lst
        self.read_audio = read_audio or data_util.read_audio
        self.device = device
        self._fbank = None

    @staticmethod
    def get_language_vocab_list(hanzi_dict):
        """end2end/data_loader.py: '<pad>', '<sos>', '</sos>' + the characters of hanzi.txt -> 6347 entries."""
        n, w2i, i2w = load_language_vocab(hanzi_dict)
        words = [i2w[i] for i in range(n)]
        words = [Const.PAD_FLAG, Const.SOS_FLAG, Const.EOS_FLAG] + words[1:]
        return len(words), {w: i for i, w in enumerate(words)}, {i: w for i, w in enumerate(words)}

    def __len__(self):
        return len(self.path_lst) // self.batch_size

    def han2id(self, line):
        try:
            flags = {Const.PAD_FLAG: Const.PAD, Const.SOS_FLAG: Const.SOS, Const.EOS_FLAG: Const.EOS}
            return [flags[h] if h in flags else self.word2index[h] for h in str(line).strip()]
        except Exception:
            raise ValueError

    def wav_padding(self, wav_data_lst):
        """end2end/data_loader.py:84-99 (host form, for feature lists computed elsewhere)."""
        lens = np.array([len(d) for d in wav_data_lst])
        out = np.zeros((len(wav_data_lst), int(lens.max()), wav_data_lst[0].shape[1]), dtype=np.float32)
        for i, d in enumerate(wav_data_lst):
            out[i, :d.shape[0], :] = d
        return out, lens

    def label_padding(self, label_data_lst, pad_idx):
        """end2end/data_loader.py:101-114."""
        lens = np.array([len(l) for l in label_data_lst])
        out = np.zeros((len(label_data_lst), int(lens.max())), dtype=np.int32) + pad_idx
        for i, l in enumerate(label_data_lst):
            out[i][:len(l)] = l
        return out, lens

    def _features(self, signals):
        """fbank(feature_dim) -> LFR(m, n) -> zero padding, for a list of 1-D float arrays: device tensor
        [len(signals), max ceil(frames / n), m * feature_dim] and the stacked lengths."""
        import torch
        from . import ops
        from .wav_util import FbankExtractor, num_frames
        if self._fbank is None:
            self._fbank = FbankExtractor(nfilt=self.feature_dim, device=self.device)
        fb = self._fbank
        n = len(signals)
        mx = max(len(s) for s in signals)
        host = np.zeros((n, mx), dtype=np.float32)
        for r, s in enumerate(signals):
            host[r, :len(s)] = s
        frames = [num_frames(len(s), fb.frame_len, fb.frame_step) for s in signals]
        t_pad = max(frames)
        feat, fr = fb.batch(torch.from_numpy(host).to(self.device),
                            torch.tensor([len(s) for s in signals], dtype=torch.int32, device=self.device), t_pad)
        lens = np.array([int(math.ceil(f / self.lfr_n)) for f in frames])
        return ops.lfr(feat, fr, self.lfr_m, self.lfr_n, int(lens.max())), lens

    def get_transformer_batch(self, rng=None, select=None):
        """Yields (pad_wav_data [B', T_lfr, lfr_m * feature_dim] float32 device tensor, pad_label_data [B', L + 1] int32 =
        SOS + hanzi ids padded with EOS, pad_target_data [B', L + 1] int32 = hanzi ids + EOS padded with IGNORE);
        ``select``: only these batch numbers (data-parallel ranks take disjoint ones)."""
        import random
        order = list(range(len(self.path_lst)))
        if self.shuffle:
            (rng or random).shuffle(order)
        for i in range(len(self.path_lst) // self.batch_size):
            if select is not None and i not in select:          # another rank's batch (train.rank_batches)
                continue
            signals, target_label_lst, input_label_lst = [], [], []
            for index in order[i * self.batch_size:(i + 1) * self.batch_size]:
                try:
                    label = self.han2id(self.han_lst[index])
                    signal, _ = self.read_audio(self.path_lst[index])
                    signals.append(np.asarray(signal, dtype=np.float32))
                    input_label_lst.append([Const.SOS] + label)
                    target_label_lst.append(label + [Const.EOS])
                except ValueError:
                    continue
            if not signals:         # every sample failed: an empty batch, so that all ranks yield the same number of times
                import torch
                z = np.zeros((0, 1), dtype=np.int32)
                yield torch.zeros(0, 1, self.lfr_m * self.feature_dim, device=self.device), z, z
                continue
            pad_wav_data, _ = self._features(signals)
            pad_label_data, _ = self.label_padding(input_label_lst, Const.EOS)
            pad_target_data, _ = self.label_padding(target_label_lst, Const.IGNORE)
            yield pad_wav_data, pad_label_data, pad_target_data

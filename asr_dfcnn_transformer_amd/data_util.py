"""``DataUtil``: the corpus index reader of the reference (util/data_util.py:12-106).

One index file per corpus and split -- tab-separated ``path<TAB>pinyin<TAB>hanzi``, no header
(util/data_util.py:83-86) -- named exactly as the reference names them (``thchs_train.txt``,
``aishell_dev.txt``, ``prime.txt``, ``noise_data.txt`` ...), selected by the boolean corpus flags of
``AmDataHparams`` / ``LmDataHparams`` / ``TransDataHparams`` and the mode (:40-78), optionally
shuffled (:91-96), then truncated to a whole number of batches (:98-105).  The reference reads them
from ``<cwd>/data``; here the directory is an argument (``data_dir``).

Audio: ``read_audio(path) -> (float samples in [-1, 1), sample_rate)`` resolves ``path`` under
``audio_root`` first and ``noise_root`` second, like data_generation (lm_and_am/data_loader.py:119-128),
and decodes 16-bit PCM WAV with the standard library -- ``soundfile`` (the reference's decoder) is not a
dependency of this package; any other decoder can be passed to ``DataLoader(read_audio=...)``."""
import os
import random
import wave

import numpy as np

_TRAIN = (('thchs30', 'thchs_train.txt'), ('aishell', 'aishell_train.txt'), ('stcmd', 'stcmd_train.txt'),
          ('aidatatang', 'aidatatang_train.txt'), ('aidatatang_1505', 'aidatatang_1505_train.txt'),
          ('prime', 'prime.txt'), ('noise', 'noise_data.txt'))
_DEV = (('thchs30', 'thchs_dev.txt'), ('aishell', 'aishell_dev.txt'), ('stcmd', 'stcmd_dev.txt'),
        ('aidatatang', 'aidatatang_dev.txt'), ('aidatatang_1505', 'aidatatang_1505_dev.txt'))
_TEST = (('thchs30', 'thchs_test.txt'), ('aishell', 'aishell_test.txt'), ('stcmd', 'stcmd_test.txt'),
         ('aidatatang', 'aidatatang_test.txt'), ('aidatatang_1505', 'aidatatang_1505_test.txt'))


def index_files(data_args, mode):
    """The index files a DataUtil of this mode reads, in the reference's order (util/data_util.py:40-78)."""
    table = {'train': _TRAIN, 'dev': _DEV, 'test': _TEST}.get(mode)
    if table is None:
        return []                      # the reference reads nothing for an unknown mode
    return [name for flag, name in table if getattr(data_args, flag) == True]      # noqa: E712  (`== True`, :41-78)


def read_index(path):
    """``pd.read_table(file, header=None)`` and columns 0, 1, 2 (:83-86)."""
    import pandas as pd
    data = pd.read_table(path, header=None)
    return data.iloc[:, 0].tolist(), data.iloc[:, 1].tolist(), data.iloc[:, 2].tolist()


def read_wav_pcm16(path):
    """16-bit PCM WAV -> (mono float64 samples in [-1, 1), sample rate): what soundfile.read returns for such a file
    (first channel of a multi-channel file)."""
    with wave.open(path, 'rb') as w:
        if w.getsampwidth() != 2:
            raise ValueError('%s: only 16-bit PCM is decoded here' % path)
        n, ch, sr = w.getnframes(), w.getnchannels(), w.getframerate()
        raw = w.readframes(n)
    data = np.frombuffer(raw, dtype='<i2').reshape(-1, ch)[:, 0]
    return data.astype(np.float64) / 32768.0, sr


class DataUtil:
    def __init__(self, data_args, batch_size, mode='train', data_length=None, shuffle=False, data_dir='data',
                 audio_root='', noise_root=None, seed=None, verbose=False):
        self.batch_size = batch_size
        self.mode = mode
        self.data_length = data_length
        self.shuffle = shuffle
        self.thchs30, self.aishell, self.stcmd = data_args.thchs30, data_args.aishell, data_args.stcmd
        self.aidatatang, self.aidatatang_1505 = data_args.aidatatang, data_args.aidatatang_1505
        self.prime, self.noise = data_args.prime, data_args.noise
        self.data_dir, self.audio_root, self.noise_root = data_dir, audio_root, noise_root
        self._rng = random.Random(seed)
        self._verbose = verbose
        self.path_lst, self.pny_lst, self.han_lst = [], [], []
        self.source_init(data_args)

    def source_init(self, data_args):
        for name in index_files(data_args, self.mode):
            if self._verbose:
                print('load ', name, ' data...')
            paths, pny, hanzi = read_index(os.path.join(self.data_dir, name))
            self.path_lst.extend(paths); self.pny_lst.extend(pny); self.han_lst.extend(hanzi)
        if self.shuffle:
            order = list(range(len(self.path_lst)))
            self._rng.shuffle(order)
            self.path_lst = [self.path_lst[i] for i in order]
            self.pny_lst = [self.pny_lst[i] for i in order]
            self.han_lst = [self.han_lst[i] for i in order]
        # whole batches only (:98-101): of data_length utterances when given, else of everything
        total = self.data_length if self.data_length else len(self.path_lst)
        stay = total // self.batch_size * self.batch_size
        self.path_lst = np.array(self.path_lst[:stay])
        self.pny_lst = np.array(self.pny_lst[:stay])
        self.han_lst = np.array(self.han_lst[:stay])

    def read_audio(self, path):
        f1 = os.path.join(self.audio_root, path)
        if os.path.isfile(f1):
            return read_wav_pcm16(f1)
        if self.noise_root is not None:
            f2 = os.path.join(self.noise_root, path)
            if os.path.isfile(f2):
                return read_wav_pcm16(f2)
        raise FileNotFoundError(path)       # the reference prints "file path Error" and returns 0 (data_loader.py:126-128)

    def generate_dict(self):
        """Hanzi of the loaded transcripts by descending frequency (util/data_util.py:107-117, without writing a file)."""
        from collections import Counter
        c = Counter(ch for han in self.han_lst for ch in han)
        return [w for w, n in sorted(c.items(), key=lambda kv: kv[1], reverse=True) if n > 0]

"""Host-side data seams (no GPU): the TSV index reader DataUtil (util/data_util.py:12-106) on the hand-written fixtures
of tests/golden/index, corpus-flag defaults of the three data hparams classes (util/hparams.py:37-91), the language-model
batch generator (lm_and_am/data_loader.py:164-193) and the label conventions of the end-to-end loader
(end2end/data_loader.py:101-114,155-156,294-296)."""
import os
import random
import wave

import numpy as np
import pytest

from asr_dfcnn_transformer_amd.const import Const
from asr_dfcnn_transformer_amd.data_util import DataUtil, index_files, read_wav_pcm16
from asr_dfcnn_transformer_amd.hparams import AmDataHparams, AmLmHparams, LmDataHparams, TransDataHparams

INDEX = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'index')


def test_corpus_flag_defaults_follow_the_reference():
    a, l, t = AmDataHparams().args, LmDataHparams().args, TransDataHparams().args
    for hp in (a, l):                                                         # util/hparams.py:39-45, 58-64
        assert (hp.thchs30, hp.aishell, hp.prime, hp.stcmd, hp.aidatatang, hp.aidatatang_1505, hp.noise) == \
               (True, True, True, True, False, False, False)
    assert (t.thchs30, t.aishell, t.prime, t.stcmd, t.aidatatang, t.aidatatang_1505, t.noise) == \
           (True, True, False, False, False, False, False)                   # util/hparams.py:77-83
    assert (t.lfr_m, t.lfr_n, t.pinyin_dict, t.hanzi_dict) == (4, 3, 'mixdict.txt', 'hanzi.txt')


def test_index_files_per_mode_and_flags():
    t = TransDataHparams().args
    assert index_files(t, 'train') == ['thchs_train.txt', 'aishell_train.txt']
    assert index_files(t, 'dev') == ['thchs_dev.txt', 'aishell_dev.txt'] and index_files(t, 'test') == ['thchs_test.txt', 'aishell_test.txt']
    a = AmDataHparams().args
    assert index_files(a, 'train') == ['thchs_train.txt', 'aishell_train.txt', 'stcmd_train.txt', 'prime.txt']
    a.noise = True
    assert index_files(a, 'train')[-1] == 'noise_data.txt' and 'noise_data.txt' not in index_files(a, 'dev')
    assert index_files(a, 'other') == []


def test_datautil_reads_tsv_and_keeps_whole_batches():
    hp = TransDataHparams().args                                # thchs30 + aishell
    d = DataUtil(hp, batch_size=3, mode='train', data_dir=INDEX)
    assert len(d.path_lst) == 6                                 # 5 + 3 = 8 utterances -> 2 whole batches of 3
    assert d.path_lst[0] == 'data_thchs30/train/A2_0.wav' and d.pny_lst[0] == 'ni3 hao3 zhong1 guo2' and d.han_lst[0] == '你好中国'
    assert d.path_lst[5] == 'data_aishell/wav/train/S0002/B1.wav'            # files in the reference's order, rows in file order
    assert isinstance(d.path_lst, np.ndarray)
    d2 = DataUtil(hp, batch_size=3, mode='train', data_length=5, data_dir=INDEX)
    assert len(d2.path_lst) == 3                                # data_length // batch * batch (:98-99)
    hp.aishell = False
    assert len(DataUtil(hp, batch_size=2, mode='test', data_dir=INDEX).path_lst) == 2      # 3 test utterances -> 1 batch of 2
    # shuffle: a permutation of the same rows, triples kept together, reproducible from the seed
    hp.aishell = True
    s1 = DataUtil(hp, 1, 'train', shuffle=True, data_dir=INDEX, seed=5)
    s2 = DataUtil(hp, 1, 'train', shuffle=True, data_dir=INDEX, seed=5)
    plain = DataUtil(hp, 1, 'train', data_dir=INDEX)
    assert list(s1.path_lst) == list(s2.path_lst) and sorted(s1.path_lst) == sorted(plain.path_lst)
    assert list(s1.path_lst) != list(plain.path_lst)
    lookup = dict(zip(plain.path_lst, zip(plain.pny_lst, plain.han_lst)))
    assert all(lookup[p] == (q, h) for p, q, h in zip(s1.path_lst, s1.pny_lst, s1.han_lst))
    assert d.generate_dict()[0] in '中国你好学'                  # most frequent hanzi first


def test_wav_reader_and_audio_lookup(tmp_path):
    sig = (0.25 * np.sin(np.arange(1600) * 0.05) * 32768).astype('<i2')
    os.makedirs(tmp_path / 'noise' / 'prime')
    with wave.open(str(tmp_path / 'noise' / 'prime' / 'P1.wav'), 'wb') as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000); w.writeframes(sig.tobytes())
    x, sr = read_wav_pcm16(str(tmp_path / 'noise' / 'prime' / 'P1.wav'))
    assert sr == 16000 and x.dtype == np.float64 and np.array_equal(x, sig.astype(np.float64) / 32768.0) and np.abs(x).max() < 1
    hp = AmDataHparams().args
    hp.thchs30 = hp.aishell = hp.stcmd = False                 # prime.txt only
    d = DataUtil(hp, 1, 'train', data_dir=INDEX, audio_root=str(tmp_path / 'speech'), noise_root=str(tmp_path / 'noise'))
    y, _ = d.read_audio(d.path_lst[0])                          # not under audio_root -> found under noise_root (:119-124)
    assert np.array_equal(x, y)
    with pytest.raises(FileNotFoundError):
        d.read_audio('nowhere.wav')


def _lm_loader(batch=2, shuffle=False):
    from asr_dfcnn_transformer_amd.data_loader import DataLoader
    hp = TransDataHparams().args
    am = AmLmHparams().args
    am.lm_batch_size = batch
    src = DataUtil(hp, batch, 'train', shuffle=shuffle, data_dir=INDEX, seed=1)
    return DataLoader(src, hp, am, device='cpu'), src


def test_get_lm_batch_layouts_and_quirks():
    dl, src = _lm_loader(batch=2)
    assert dl.acoustic_vocab_size == 1536 and dl.language_vocab_size == 6345
    batches = list(dl.get_lm_batch())
    assert len(batches) == 4                                    # 8 utterances / 2
    x, n, y = batches[0]
    assert x.shape == (2, 5) and y.shape == (2, 5) and x.dtype == np.int32      # padded to the longer pinyin sequence (5)
    assert x[0].tolist() == [dl.pinyin2index[p] for p in 'ni3 hao3 zhong1 guo2'.split()] + [0]
    assert y[0].tolist() == [dl.word2index[c] for c in '你好中国'] + [0]
    assert n.tolist() == [len('ni3 hao3 zhong1 guo2'), len('jin1 tian1 tian1 qi4 hao3')]       # string lengths (:190)
    # rank sharding: `select` yields exactly those batch numbers
    sel = list(dl.get_lm_batch(select={1, 3}))
    assert len(sel) == 2 and np.array_equal(sel[0][0], batches[1][0]) and np.array_equal(sel[1][2], batches[3][2])
    # an unknown token drops its row (:192); a hanzi / pinyin count mismatch would be ragged in the reference -> dropped
    src.pny_lst[2] = 'wo3 men2 zzz9 zhong1 guo2 ren2'
    src.han_lst[3] = '北京大'
    b1 = list(dl.get_lm_batch())[1]
    assert b1[0].shape[0] == 0 and b1[2].shape == (0, 6)
    # shuffled order is a function of the rng passed in (all ranks pass the same seed)
    dls, _ = _lm_loader(batch=2, shuffle=True)
    a = [b[0].tolist() for b in dls.get_lm_batch(rng=random.Random(3))]
    b = [b[0].tolist() for b in dls.get_lm_batch(rng=random.Random(3))]
    c = [b[0].tolist() for b in dls.get_lm_batch(rng=random.Random(4))]
    assert a == b and a != c


def test_e2e_label_conventions():
    from asr_dfcnn_transformer_amd.e2e_data_loader import dataloader
    from asr_dfcnn_transformer_amd.e2e_model import E2EHparams
    hp = TransDataHparams().args
    args = E2EHparams()
    args.batch_size = 2
    dl = dataloader(args, hp, data_util=DataUtil(hp, 2, 'train', data_dir=INDEX), device='cpu')
    assert dl.language_vocab_size == 6347 and dl.acoustic_vocab_size == 1536
    assert (dl.word2index['<pad>'], dl.word2index['<sos>'], dl.word2index['</sos>']) == (Const.PAD, Const.SOS, Const.EOS) == (0, 1, 2)
    assert dl.word2index['一'] == 3 and len(dl) == 4
    ids = dl.han2id('你好')
    assert ids == [dl.word2index['你'], dl.word2index['好']] and min(ids) >= 3
    with pytest.raises(ValueError):
        dl.han2id('你A好')
    inp, _ = dl.label_padding([[Const.SOS] + ids, [Const.SOS] + ids + ids], Const.EOS)
    tgt, lens = dl.label_padding([ids + [Const.EOS], ids + ids + [Const.EOS]], Const.IGNORE)
    assert inp.tolist() == [[1] + ids + [2, 2], [1] + ids + ids] and lens.tolist() == [3, 5]
    assert tgt.tolist() == [ids + [2, -1, -1], ids + ids + [2]] and tgt.dtype == np.int32
    w, wl = dl.wav_padding([np.ones((3, 4), np.float32), np.ones((5, 4), np.float32)])
    assert w.shape == (2, 5, 4) and wl.tolist() == [3, 5] and not w[0, 3:].any()

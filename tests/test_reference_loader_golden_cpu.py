"""Against outputs of the reference's own data-loader methods (lm_and_am/data_loader.py:43-103, end2end/data_loader.py:59-111,314-333) run
unmodified in the build container on the reference's dictionary files (tests/golden/make_reference_loader_golden.py explains how; the
fixture tests/golden/reference_loader.json holds inputs and outputs only): the vocabularies the package builds from its packaged copies of
those dictionaries (1536 / 1424 acoustic symbols incl. the CTC blank, 6345 / 6347 characters), the pinyin / hanzi id lookups of both loaders
and the end-to-end loader's padding layouts must be the same, entry for entry.  Where the reference lets an exception escape, the package
raises ValueError (the exception its callers drop a row for) -- for an unknown character in end2end/data_loader.py the reference's KeyError
would end the epoch instead (asr_dfcnn_transformer_amd/data_loader.py:99-101 says so); that difference is asserted here, not hidden."""
import argparse
import json
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'reference_loader.json'), encoding='utf-8'))
INDEX = os.path.join(ROOT, 'tests', 'golden', 'index')


def _norm(seq):
    """pandas reads the toneless syllable `nan` of mixdict.txt / dict.txt as a float NaN -- in the reference and here alike (so that token
    can never be looked up); NaN != NaN, so both sides are compared through a placeholder"""
    return ['<NaN>' if isinstance(x, float) and x != x else x for x in seq]


def test_vocabularies_equal_the_reference():
    from asr_dfcnn_transformer_amd import data_loader, e2e_data_loader
    from asr_dfcnn_transformer_amd.const import Const
    d = Const.DictFolder
    for dic in ('mixdict.txt', 'dict.txt'):
        g = GOLD['acoustic_' + dic]
        n, p2i, i2p = data_loader.load_acoustic_vocab(os.path.join(d, dic))
        assert n == g['size'] and _norm(i2p[i] for i in range(n)) == _norm(g['symbols']) and len(p2i) == g['n_distinct']
        assert i2p[n - 1] == '_'                                            # the CTC blank is the last id
    assert GOLD['acoustic_mixdict.txt']['size'] == 1536                     # BASELINE.json's V
    g = GOLD['language_lm_and_am']
    n, w2i, i2w = data_loader.load_language_vocab(os.path.join(d, 'hanzi.txt'))
    assert n == g['size'] == 6345 and [i2w[i] for i in range(n)] == g['words'] and len(w2i) == g['n_distinct']
    g = GOLD['language_end2end']
    n, w2i, i2w = e2e_data_loader.dataloader.get_language_vocab_list(os.path.join(d, 'hanzi.txt'))
    assert n == g['size'] == 6347 and [i2w[i] for i in range(n)] == g['words'] and len(w2i) == g['n_distinct']


def _loaders():
    from asr_dfcnn_transformer_amd import data_loader, e2e_data_loader
    from asr_dfcnn_transformer_amd.data_util import DataUtil
    from asr_dfcnn_transformer_amd.hparams import AmLmHparams, TransDataHparams
    hp = TransDataHparams().args
    du = DataUtil(hp, batch_size=1, mode='train', data_dir=INDEX)
    am = data_loader.DataLoader(du, hp, AmLmHparams().args, device='cpu')
    tr = argparse.Namespace(batch_size=1, feature_dim=80)
    e2e = e2e_data_loader.dataloader(tr, hp, data_util=du, device='cpu')
    return am, e2e


def test_id_lookups_equal_the_reference():
    am, e2e = _loaders()
    checked = escaped = 0
    for key, fn in (('pny2id', am.pny2id), ('han2id_lm_and_am', am.han2id), ('han2id_end2end', e2e.han2id)):
        for case in GOLD[key]:
            if 'ok' in case:
                assert fn(case['line']) == case['ok'], (key, case['line'])
                checked += 1
            else:
                with pytest.raises(ValueError):                             # reference: ValueError, or (end2end, unknown character) KeyError
                    fn(case['line'])
                escaped += case['raises'] != 'ValueError'
    assert checked >= 50 and escaped == 1


def test_padding_layouts_equal_the_reference():
    _, e2e = _loaders()
    for case in GOLD['padding']:
        feats = [np.asarray(f, dtype=np.float32) for f in case['feats']]
        w, wl = e2e.wav_padding(feats)
        assert str(w.dtype) == case['wav_dtype'] and np.array_equal(w, np.asarray(case['wav'], dtype=np.float32)) and wl.tolist() == case['wav_lens']
        for pad, key in ((0, 'lab_pad0'), (2, 'lab_pad2')):
            lab, ll = e2e.label_padding(case['labels'], pad)
            assert str(lab.dtype) == case['lab_dtype'] and lab.tolist() == case[key] and ll.tolist() == case['lab_lens']


def test_language_model_batches_equal_the_reference():
    """DataLoader.get_lm_batch (lm_and_am/data_loader.py:164-193) on the index fixtures, lists from DataUtil, no shuffle: the same batches,
    rows, padding and (character-count) input_length.  The package yields int32 arrays (device ids), the reference int64: values compared."""
    from asr_dfcnn_transformer_amd import data_loader
    from asr_dfcnn_transformer_amd.data_util import DataUtil
    from asr_dfcnn_transformer_amd.hparams import AmLmHparams, TransDataHparams
    hp = TransDataHparams().args                                             # thchs30 + aishell, as the fixture was generated
    n = 0
    for case in GOLD['lm_batches']:
        tr = AmLmHparams().args
        tr.lm_batch_size = case['batch_size']
        du = DataUtil(hp, batch_size=case['batch_size'], mode=case['mode'], data_dir=INDEX)
        got = list(data_loader.DataLoader(du, hp, tr, device='cpu').get_lm_batch())
        assert len(got) == len(case['batches']), (case['mode'], case['batch_size'])
        for (a, b, c), want in zip(got, case['batches']):
            assert a.tolist() == want['input_data'] and b.tolist() == want['input_length'] and c.tolist() == want['label_data']
            n += 1
    assert n >= 15

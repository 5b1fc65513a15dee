"""Against outputs of the REFERENCE ITSELF (tests/golden/reference_host.json, written by tests/golden/make_reference_host_golden.py,
which imports /root/reference's util/hparams.py and util/data_util.py -- the two modules of it that import here without TensorFlow):
the host-side mirrors asr_dfcnn_transformer_amd.hparams and .data_util.DataUtil give the same namespaces, the same data lists in the
same order for every combination of corpora / mode / batch size / data_length, and the same frequency-ordered hanzi vocabulary.
This pins the configuration and data-list seam to the reference; the arithmetic of the hot path stays
unpinned (oracle/__init__.py).  Reads the committed JSON only: nothing here touches /root/reference."""
import argparse
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'reference_host.json'), encoding='utf-8'))
INDEX = os.path.join(ROOT, 'tests', 'golden', 'index')


def test_hparams_namespaces_equal_the_reference():
    from asr_dfcnn_transformer_amd import hparams
    assert sorted(GOLD['hparams']) == ['AmDataHparams', 'AmLmHparams', 'LmDataHparams', 'TransDataHparams']
    for cls, want in GOLD['hparams'].items():
        got = vars(getattr(hparams, cls)().args)
        assert sorted(got) == sorted(want), (cls, sorted(set(got) ^ set(want)))
        for k, v in want.items():
            assert got[k] == v and type(got[k]) is type(v), (cls, k, got[k], v)
        assert vars(getattr(hparams, cls).args) == got          # the class attribute the reference's callers read (X.args)


def test_token_constants_equal_the_reference():
    from asr_dfcnn_transformer_amd.const import Const
    assert len(GOLD['const']) == 7
    for k, v in GOLD['const'].items():
        assert getattr(Const, k) == v and type(getattr(Const, k)) is type(v), k


def test_datautil_lists_equal_the_reference():
    from asr_dfcnn_transformer_amd.data_util import DataUtil
    assert len(GOLD['data_util']) >= 100
    seen_cut = seen_empty = 0
    for case in GOLD['data_util']:
        d = DataUtil(argparse.Namespace(**case['flags']), batch_size=case['batch_size'], mode=case['mode'],
                     data_length=case['data_length'], shuffle=False, data_dir=INDEX)
        key = (case['flags'], case['mode'], case['batch_size'], case['data_length'])
        assert [str(x) for x in d.path_lst] == case['path_lst'], key
        assert [str(x) for x in d.pny_lst] == case['pny_lst'], key
        assert [str(x) for x in d.han_lst] == case['han_lst'], key
        seen_cut += case['data_length'] is not None and len(case['path_lst']) > 0
        seen_empty += len(case['path_lst']) == 0
    assert seen_cut and seen_empty          # the fixture covers the data_length cut and batches that do not fill


def test_generate_dict_equals_the_file_the_reference_writes():
    from asr_dfcnn_transformer_amd.data_util import DataUtil
    g = GOLD['generate_dict']
    d = DataUtil(argparse.Namespace(**g['flags']), batch_size=g['batch_size'], mode=g['mode'], data_dir=INDEX)
    assert '\n'.join(d.generate_dict()) == g['new_hanzi_txt']

#!/usr/bin/env python3
"""Golden vectors from the PURE methods of the reference's two data loaders, executed here unmodified on the reference's own dictionary
files: lm_and_am/data_loader.py `DataLoader.get_acoustic_vocab_list` (:85-92), `get_language_vocab_list` (:95-103), `pny2id` (:43-58),
`han2id` (:61-82), `get_lm_batch` (:164-193), and end2end/data_loader.py `dataloader.get_acoustic_vocab_list` (:314-321), `get_language_vocab_list` (:324-333),
`han2id` (:59-80), `wav_padding` (:82-96), `label_padding` (:98-111).

As tests/golden/make_reference_utils_golden.py: both modules stop at `import keras` when imported (ordinary ModuleNotFoundError; nothing is
installed or stubbed); the methods above need numpy, pandas, os / pathlib and util.const.Const only -- the last one imports fine and IS
imported from the reference (bytecode writing off).  Their FunctionDef nodes are taken from the class bodies with `ast`, compiled as they
stand and called with a plain namespace object as `self` that carries the attributes the real `__init__` would have set from the same
methods (pinyin_dict / hanzi_dict paths, pinyin2index, word2index).  Nothing of the reference is copied into this repository; the fixture
(tests/golden/reference_loader.json) holds inputs and outputs only.  Methods that read audio or call TensorFlow are not run.
usage: python3 tests/golden/make_reference_loader_golden.py   (in the build container: needs /root/reference)"""
import ast
import json
import os
import sys
import types
from pathlib import Path

import numpy as np
import pandas as pd

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference'


def methods(path, cls, names, extra):
    tree = ast.parse(open(path, encoding='utf-8').read(), filename=path)
    c = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == cls)
    nodes = [n for n in c.body if isinstance(n, ast.FunctionDef) and n.name in names]
    assert sorted(n.name for n in nodes) == sorted(names), (path, [n.name for n in nodes])
    ns = dict(extra)
    exec(compile(ast.Module(body=nodes, type_ignores=[]), path, 'exec'), ns)
    return {k: ns[k] for k in names}


def outcome(fn, *a):
    try:
        return {'ok': fn(*a)}
    except Exception as e:                             # which exception escapes is part of the behaviour
        return {'raises': type(e).__name__}


def main():
    if not os.path.isdir(REF):
        raise SystemExit('%s is not here: the fixture is generated in the build container only' % REF)
    sys.argv = ['make_reference_loader_golden']
    sys.path.insert(0, REF)
    from util.const import Const                       # imports without TensorFlow
    common = {'np': np, 'pd': pd, 'os': os, 'Path': Path, 'Const': Const, 'home_dir': REF}      # home_dir = os.getcwd() of a run from the checkout root
    am = methods(os.path.join(REF, 'lm_and_am', 'data_loader.py'), 'DataLoader',
                 ('get_acoustic_vocab_list', 'get_language_vocab_list', 'pny2id', 'han2id', 'get_lm_batch'), common)
    e2e = methods(os.path.join(REF, 'end2end', 'data_loader.py'), 'dataloader',
                  ('get_acoustic_vocab_list', 'get_language_vocab_list', 'han2id', 'wav_padding', 'label_padding'), common)
    out = {}
    for dic in ('mixdict.txt', 'dict.txt'):
        me = types.SimpleNamespace(pinyin_dict=os.path.join(REF, dic))
        n, p2i, i2p = am['get_acoustic_vocab_list'](me)
        n2, p2i2, _ = e2e['get_acoustic_vocab_list'](types.SimpleNamespace(pinyin_dict=dic))
        assert n == n2 and p2i == p2i2
        out['acoustic_' + dic] = {'size': n, 'symbols': [i2p[i] for i in range(n)], 'n_distinct': len(p2i)}
    me = types.SimpleNamespace(hanzi_dict='hanzi.txt')
    n, w2i, i2w = am['get_language_vocab_list'](me)
    out['language_lm_and_am'] = {'size': n, 'words': [i2w[i] for i in range(n)], 'n_distinct': len(w2i)}
    n3, w2i3, i2w3 = e2e['get_language_vocab_list'](me)
    out['language_end2end'] = {'size': n3, 'words': [i2w3[i] for i in range(n3)], 'n_distinct': len(w2i3)}
    # id lookups: the transcripts of the committed index fixtures + lines that must fail
    lines_py, lines_han = [], []
    for f in sorted(os.listdir(os.path.join(HERE, 'index'))):
        if f.endswith('.txt'):
            for row in open(os.path.join(HERE, 'index', f), encoding='utf-8').read().splitlines():
                cols = row.split('\t')
                if len(cols) == 3:
                    lines_py.append(cols[1]); lines_han.append(cols[2])
    lines_py += [' ni3 hao3 ', 'ni3  hao3', 'ni3 hao9', '']
    lines_han += [' 你好 ', '你好\n', '你☃好', '']
    _, p2i, _ = am['get_acoustic_vocab_list'](types.SimpleNamespace(pinyin_dict=os.path.join(REF, 'mixdict.txt')))
    me_am = types.SimpleNamespace(pinyin2index=p2i, word2index=w2i)
    me_e2e = types.SimpleNamespace(pinyin2index=p2i, word2index=w2i3)
    out['pny2id'] = [{'line': l, **outcome(am['pny2id'], me_am, l)} for l in lines_py]
    out['han2id_lm_and_am'] = [{'line': l, **outcome(am['han2id'], me_am, l)} for l in lines_han]
    out['han2id_end2end'] = [{'line': l, **outcome(e2e['han2id'], me_e2e, l)} for l in lines_han]
    # language-model batches (lm_and_am/data_loader.py:164-193; no audio involved): the lists come from the reference's own DataUtil
    # (util/data_util.py imports here) reading copies of the committed index fixtures from a scratch directory
    import contextlib, io, shutil, tempfile
    work = tempfile.mkdtemp(prefix='refloader_')
    os.makedirs(os.path.join(work, 'data'))
    for f in sorted(os.listdir(os.path.join(HERE, 'index'))):
        if f.endswith('.txt'):
            shutil.copy(os.path.join(HERE, 'index', f), os.path.join(work, 'data', f))
    os.chdir(work)
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            from util.data_util import DataUtil
        out['lm_batches'] = []
        for mode, bs in (('train', 1), ('train', 2), ('train', 3), ('test', 2), ('dev', 4)):
            with contextlib.redirect_stdout(io.StringIO()):
                du = DataUtil(types.SimpleNamespace(thchs30=True, aishell=True, stcmd=False, aidatatang=False, aidatatang_1505=False, prime=False,
                                                    noise=False), batch_size=bs, mode=mode, data_length=None, shuffle=False)
            me = types.SimpleNamespace(pny_lst=du.pny_lst, han_lst=du.han_lst, shuffle=False, lm_batch_size=bs, pinyin2index=p2i, word2index=w2i)
            me.pny2id = types.MethodType(am['pny2id'], me)
            me.han2id = types.MethodType(am['han2id'], me)
            batches = [{'input_data': np.asarray(a).tolist(), 'input_length': np.asarray(b).tolist(), 'label_data': np.asarray(c).tolist()}
                       for a, b, c in am['get_lm_batch'](me)]
            out['lm_batches'].append({'mode': mode, 'batch_size': bs, 'batches': batches})
    finally:
        os.chdir(HERE)
        shutil.rmtree(work, ignore_errors=True)
    rng = np.random.default_rng(7)
    pads = []
    for lens, dim in (((5, 3, 7), 4), ((1,), 2), ((2, 2), 3), ((9, 1, 4, 6), 8)):
        feats = [rng.standard_normal((t, dim)).astype(np.float32) for t in lens]
        w, wl = e2e['wav_padding'](None, feats)
        labels = [rng.integers(3, 50, int(t)).tolist() for t in lens]
        lab, ll = e2e['label_padding'](None, labels, 0)
        lab2, _ = e2e['label_padding'](None, labels, 2)
        pads.append({'feats': [f.tolist() for f in feats], 'wav': w.tolist(), 'wav_dtype': str(w.dtype), 'wav_lens': wl.tolist(),
                     'labels': labels, 'lab_pad0': lab.tolist(), 'lab_pad2': lab2.tolist(), 'lab_dtype': str(lab.dtype), 'lab_lens': ll.tolist()})
    out['padding'] = pads
    with open(os.path.join(HERE, 'reference_loader.json'), 'w', encoding='utf-8') as f:
        json.dump(out, f, ensure_ascii=False, sort_keys=True)
    print('vocabularies %d / %d / %d / %d entries, %d + %d lookups, %d padding cases -> tests/golden/reference_loader.json' % (
        out['acoustic_mixdict.txt']['size'], out['acoustic_dict.txt']['size'], n, n3, len(lines_py), len(lines_han), len(pads)))


if __name__ == '__main__':
    main()

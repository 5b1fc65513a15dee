#!/usr/bin/env python3
"""Golden vectors from the REFERENCE ITSELF, for the host-side modules of it that import in this container without TensorFlow / Keras:
`util/hparams.py` (the four argparse namespaces, evaluated at import: util/hparams.py:5-91), `util/const.py` (token ids) and `util/data_util.py` (the TSV index
reader DataUtil: util/data_util.py:12-106).  Everything else of the reference stops at an ordinary ModuleNotFoundError (keras,
tensorflow, soundfile, librosa, python_speech_features: SURVEY 8c), so this pins the data-list / configuration seam only -- the
arithmetic of the hot path stays "parity unpinned" (oracle/__init__.py).

Runs only where /root/reference exists (this container, never the GPU box); writes tests/golden/reference_host.json, which
tests/test_reference_host_golden_cpu.py compares with asr_dfcnn_transformer_amd.hparams / .data_util.  The reference is imported with
bytecode writing switched off and from a scratch working directory: the reference tree is not written to, and DataUtil's
`os.getcwd()/data/<file>` convention finds COPIES of the committed index fixtures (tests/golden/index/*.txt) there.
usage: python3 tests/golden/make_reference_host_golden.py"""
import contextlib
import io
import json
import os
import shutil
import sys
import tempfile

sys.dont_write_bytecode = True                      # no __pycache__ in the reference tree
HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference'


def main():
    if not os.path.isdir(REF):
        raise SystemExit('%s is not here: the fixture is generated in the build container only' % REF)
    work = tempfile.mkdtemp(prefix='refhost_')
    try:
        os.makedirs(os.path.join(work, 'data'))
        for f in sorted(os.listdir(os.path.join(HERE, 'index'))):
            if f.endswith('.txt'):
                shutil.copy(os.path.join(HERE, 'index', f), os.path.join(work, 'data', f))
        os.chdir(work)                               # util/data_util.py:2 home_dir = os.getcwd() at import
        sys.argv = ['make_reference_host_golden']    # util/hparams.py parses sys.argv at class definition
        sys.path.insert(0, REF)
        with contextlib.redirect_stdout(io.StringIO()):
            from util import hparams as rh
            from util.data_util import DataUtil
            from util.const import Const
        out = {'hparams': {}, 'data_util': []}
        # token ids and flags (util/const.py:35-41); the machine-specific path table below them is out of scope
        out['const'] = {k: getattr(Const, k) for k in ('IGNORE', 'PAD', 'SOS', 'EOS', 'PAD_FLAG', 'SOS_FLAG', 'EOS_FLAG')}
        for cls in ('AmLmHparams', 'AmDataHparams', 'LmDataHparams', 'TransDataHparams'):
            out['hparams'][cls] = {k: v for k, v in sorted(vars(getattr(rh, cls).args).items())}
        import argparse
        have = set(os.listdir(os.path.join(work, 'data')))

        def flags(**kw):
            base = dict(thchs30=False, aishell=False, stcmd=False, aidatatang=False, aidatatang_1505=False, prime=False, noise=False)
            base.update(kw)
            return base

        cases = []
        for fl in (flags(thchs30=True), flags(aishell=True), flags(thchs30=True, aishell=True), flags(thchs30=True, aishell=True, prime=True)):
            for mode in ('train', 'dev', 'test'):
                for batch in (1, 2, 3):
                    for length in (None, 5, 2):
                        cases.append((fl, mode, batch, length))
        for fl, mode, batch, length in cases:
            need = {'train': {'thchs30': 'thchs_train.txt', 'aishell': 'aishell_train.txt', 'prime': 'prime.txt'},
                    'dev': {'thchs30': 'thchs_dev.txt', 'aishell': 'aishell_dev.txt'},
                    'test': {'thchs30': 'thchs_test.txt', 'aishell': 'aishell_test.txt'}}[mode]
            if any(fl[k] and f not in have for k, f in need.items()):
                continue
            with contextlib.redirect_stdout(io.StringIO()):
                d = DataUtil(argparse.Namespace(**fl), batch_size=batch, mode=mode, data_length=length, shuffle=False)
            out['data_util'].append({'flags': fl, 'mode': mode, 'batch_size': batch, 'data_length': length,
                                     'path_lst': [str(x) for x in d.path_lst], 'pny_lst': [str(x) for x in d.pny_lst],
                                     'han_lst': [str(x) for x in d.han_lst]})
        # generate_dict (util/data_util.py:108-117) writes new_hanzi.txt into os.getcwd(): the scratch directory here
        with contextlib.redirect_stdout(io.StringIO()):
            d = DataUtil(argparse.Namespace(**flags(thchs30=True, aishell=True)), batch_size=1, mode='train', data_length=None, shuffle=False)
            d.generate_dict()
        out['generate_dict'] = {'flags': flags(thchs30=True, aishell=True), 'mode': 'train', 'batch_size': 1,
                                'new_hanzi_txt': open(os.path.join(work, 'new_hanzi.txt'), encoding='utf-8').read()}
        os.chdir(HERE)
        with open(os.path.join(HERE, 'reference_host.json'), 'w', encoding='utf-8') as f:
            json.dump(out, f, ensure_ascii=False, indent=1, sort_keys=True)
        print('%d DataUtil cases, %d hparams classes -> tests/golden/reference_host.json' % (len(out['data_util']), len(out['hparams'])))
    finally:
        os.chdir(HERE)
        shutil.rmtree(work, ignore_errors=True)


if __name__ == '__main__':
    main()

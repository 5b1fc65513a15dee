#!/usr/bin/env python3
"""Golden vectors from three functions of the reference's util/utils.py, EXECUTED HERE UNMODIFIED: build_LFR_features (:7-31; the
low-frame-rate stacking the end-to-end data loader applies, end2end/data_loader.py:284 -- SURVEY 8 row a14), GetEditDistance (:43-54, the
word-error-rate helper of the evaluation loops) and sparse_tuple_from (:69-88, the label layout fed to CTC).

How, and what this is not: `import util.utils` stops at `from keras import backend as K` (an ordinary ModuleNotFoundError; keras and
tensorflow are not installed and nothing is installed or stubbed here).  These three functions do not touch keras / tensorflow: they
need `numpy` and `difflib` only.  This script reads util/utils.py, takes the three FunctionDef nodes out of its syntax tree (`ast`),
compiles exactly those nodes and runs them with the real numpy / difflib -- no line of the reference is copied into this repository, no
missing library is replaced by a stand-in, the module's keras-dependent functions (decode_ctc) are not run.  The outputs are the
reference's own for the inputs generated here (seeded); only inputs and outputs are written to tests/golden/reference_utils.npz.

Runs only where /root/reference exists (this container).  The reference tree is opened read-only; nothing is imported from it, so no
bytecode is written there.  usage: python3 tests/golden/make_reference_utils_golden.py"""
import ast
import difflib
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = '/root/reference/util/utils.py'
WANTED = ('build_LFR_features', 'GetEditDistance', 'sparse_tuple_from')


def reference_functions():
    tree = ast.parse(open(SRC, encoding='utf-8').read(), filename=SRC)
    nodes = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in WANTED]
    assert sorted(n.name for n in nodes) == sorted(WANTED)
    ns = {'np': np, 'difflib': difflib}                    # what util/utils.py:1-2 binds; :3-4 (keras, tensorflow) are not needed by these three
    exec(compile(ast.Module(body=nodes, type_ignores=[]), SRC, 'exec'), ns)
    return [ns[k] for k in WANTED]


def main():
    if not os.path.exists(SRC):
        raise SystemExit('%s is not here: the fixture is generated in the build container only' % SRC)
    lfr, edit, sparse = reference_functions()
    rng = np.random.default_rng(20261005)
    out = {}
    cases = []
    # (T, D, m, n): the reference's own setting (4, 3) on lengths around the padding rule, identity, skipping, right-stacking, T < m
    for i, (T, D, m, n) in enumerate([(10, 3, 4, 3), (8, 3, 4, 3), (41, 80, 4, 3), (40, 80, 4, 3), (39, 80, 4, 3), (1, 5, 4, 3), (2, 5, 4, 3),
                                      (3, 5, 4, 3), (1, 8, 4, 3), (2, 8, 4, 3), (3, 8, 4, 3), (5, 12, 6, 4), (7, 4, 1, 1), (7, 4, 1, 3), (7, 4, 2, 1), (9, 2, 3, 2), (100, 16, 7, 5), (5, 6, 6, 4)]):
        x = rng.standard_normal((T, D)).astype(np.float32)
        y = lfr(x, m, n)
        out['lfr_in_%d' % i], out['lfr_out_%d' % i] = x, np.asarray(y)
        cases.append([T, D, m, n])
    out['lfr_cases'] = np.asarray(cases, dtype=np.int64)
    pairs = [('abcd', 'abxyd'), ('', 'abc'), ('abc', ''), ('', ''), ('kitten', 'sitting'), ('aaaa', 'aa'), ('abcabc', 'cabcab'),
             ('ni3 hao3 zhong1 guo2'.split(), 'ni3 hao3 bei3 jing1 ren2'.split()), ([1, 2, 3, 4, 5], [1, 3, 5]), ([7], [7]),
             ('今天天气好', '今天气很好')]
    for _ in range(40):
        a = rng.integers(0, 6, int(rng.integers(0, 12))).tolist()
        b = rng.integers(0, 6, int(rng.integers(0, 12))).tolist()
        pairs.append((a, b))
    out['edit_pairs_json'] = np.asarray(json.dumps([[list(a) if not isinstance(a, str) else a, list(b) if not isinstance(b, str) else b] for a, b in pairs],
                                                   ensure_ascii=False))
    out['edit_dist'] = np.asarray([edit(a, b) for a, b in pairs], dtype=np.int64)
    seqs = [[[1, 2], [], [3]], [[5, 6, 7, 8], [1], [2, 3]], [[9]], [[1, 2, 0, 0, 0, 0]] * 4]
    out['sparse_seqs_json'] = np.asarray(json.dumps(seqs))
    for i, s in enumerate(seqs):
        ind, val, shp = sparse(s)
        out['sparse_ind_%d' % i], out['sparse_val_%d' % i], out['sparse_shape_%d' % i] = ind, val, shp
    np.savez_compressed(os.path.join(HERE, 'reference_utils.npz'), **out)
    print('%d LFR cases, %d edit-distance pairs, %d label batches -> tests/golden/reference_utils.npz' % (len(cases), len(pairs), len(seqs)))


if __name__ == '__main__':
    main()

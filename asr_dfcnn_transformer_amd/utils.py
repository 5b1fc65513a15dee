"""Host helpers of the reference's util/utils.py that sit on the path."""
import difflib

import numpy as np


def build_LFR_features(inputs, m, n):
    """util/utils.py:7-31: stack ``m`` frames, hop ``n``; frames past the end repeat the last
    row.  [T, D] -> [ceil(T/n), m*D]."""
    inputs = np.asarray(inputs)
    T, D = inputs.shape
    T_lfr = -(-T // n)
    rows = np.minimum(np.arange(T_lfr)[:, None] * n + np.arange(m)[None, :], T - 1)
    return inputs[rows].reshape(T_lfr, m * D)


def GetEditDistance(str1, str2):
    """util/utils.py:43-53: difflib-opcode cost used by the eval scripts (a 'replace' block
    costs max(len_a, len_b); this is not Levenshtein)."""
    cost = 0
    for tag, i1, i2, j1, j2 in difflib.SequenceMatcher(None, str1, str2).get_opcodes():
        if tag == 'replace':
            cost += max(i2 - i1, j2 - j1)
        elif tag == 'insert':
            cost += j2 - j1
        elif tag == 'delete':
            cost += i2 - i1
    return cost


def sparse_tuple_from(sequences, dtype=np.int32):
    """util/utils.py:69-87: list of sequences -> (indices, values, dense_shape)."""
    idx = [(b, j) for b, s in enumerate(sequences) for j in range(len(s))]
    indices = np.asarray(idx, dtype=np.int64).reshape(-1, 2)
    values = np.asarray([v for s in sequences for v in s], dtype=dtype)
    shape = np.asarray([len(sequences), (indices[:, 1].max() + 1) if len(idx) else 0], dtype=np.int64)
    return indices, values, shape

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


def decode_ctc(num_result, input_length, log_time_major=False):
    """util/utils.py:57-66: ``K.ctc_decode(num_result, [input_length], greedy=True)`` of ONE utterance -- the softmax output
    ``num_result`` [1, T, V] of the acoustic model goes through log(transpose(.) + 1e-7) and tf.nn.ctc_greedy_decoder
    (merge_repeated, blank = V - 1) over the first ``input_length`` frames; returns the ids of that utterance (r1[0]).
    Runs asr_ctc_greedy on the device.  log(p + eps) is monotone in p, so the arg-max path is taken on ``num_result`` itself
    (for one utterance [1, T, V] and time-major [T, 1, V] are the same memory); ``log_time_major=True`` says the tensor
    already is the time-major log form [T, 1, V] an engine holds."""
    import torch
    from . import ops
    x = num_result if torch.is_tensor(num_result) else torch.as_tensor(np.asarray(num_result, dtype=np.float32))
    x = x.to('cuda', dtype=torch.float32).contiguous()
    if x.dim() != 3 or (x.shape[1] if log_time_major else x.shape[0]) != 1:
        raise ValueError('decode_ctc takes the output of one utterance ([1, T, V]); K.ctc_decode is called with one length (utils.py:59-61)')
    T, V = (x.shape[0], x.shape[2]) if log_time_major else (x.shape[1], x.shape[2])
    n = int(np.asarray(input_length).reshape(-1)[0])
    if not 0 <= n <= T:
        raise ValueError('input_length %d outside [0, %d]' % (n, T))
    seq = torch.tensor([n], dtype=torch.int32, device='cuda')
    ids = torch.zeros(1, T, dtype=torch.int32, device='cuda')
    cnt = torch.zeros(1, dtype=torch.int32, device='cuda')
    neg = torch.zeros(1, dtype=torch.float32, device='cuda')
    ws = torch.zeros(ops.ctc_greedy_workspace(T, 1) // 4 + 4, dtype=torch.int32, device='cuda')
    ops.ctc_greedy(x, T, 1, V, seq, V - 1, ids, cnt, neg, ws)
    return ids[0, :int(cnt.item())].cpu().numpy().astype(np.int64)

"""Test infrastructure: an independent torch-CPU autograd restatement of the encoder-decoder graph with id input
(oracle.transformer.init_e2e(vin=...)), built from the blocks that pin the numpy oracle in
tests/test_oracle_transformer_cpu.py, runnable in float64 (== the oracle) or float32 (what a plain float32 framework
computes: the yardstick for how far ANY float32 implementation sits from the float64 oracle at full width)."""
import numpy as np
import torch

from test_oracle_transformer_cpu import t_mha, t_ffn


def e2e_ids_grads(P, x, y_in, y_tgt, heads, blocks, dtype=torch.float64):
    """-> (flat {name: gradient as float64 ndarray}, mean_loss, logits ndarray).  P: nested oracle parameters whose shared
    (tied) tensors are the same objects."""
    tP, shared = {}, {}
    for k, v in P.items():
        if isinstance(v, dict):
            tP[k] = {}
            for kk, vv in v.items():
                if id(vv) not in shared:
                    shared[id(vv)] = torch.tensor(np.asarray(vv), dtype=dtype, requires_grad=True)
                tP[k][kk] = shared[id(vv)]
        else:
            tP[k] = torch.tensor(np.asarray(v), dtype=dtype, requires_grad=True)
    C = tP['enc_emb'].shape[1]
    N, T = x.shape
    L = y_in.shape[1]
    xt = torch.tensor(x)
    enc = tP['enc_emb'][xt] * (xt != 0).unsqueeze(-1) * (C ** 0.5) + tP['enc_pe'][torch.arange(T)][None]
    dec = tP['dec_input'][torch.tensor(y_in)] + tP['dec_pe'][torch.arange(L)][None]
    for i in range(blocks):
        enc = t_mha(enc, enc, tP['enc%d' % i], heads, False)
    mem = t_ffn(enc, tP['enc_ffn'])
    for i in range(blocks):
        dec = t_mha(dec, mem, tP['dec%d' % i], heads, True)
    logits = t_ffn(dec, tP['dec_ffn']) @ tP['out_w'] + tP['out_b']
    V = logits.shape[-1]
    target = torch.tensor(y_tgt)
    oh = torch.zeros_like(logits)
    valid = (target >= 0) & (target < V)
    idx = torch.nonzero(valid, as_tuple=True)
    oh[idx[0], idx[1], target[valid]] = 1.0
    ys = 0.9 * oh + 0.1 / V
    loss = -(ys * torch.log_softmax(logits, -1)).sum(-1)
    ist = (target != 0).to(dtype)
    mean_loss = (loss * ist).sum() / ist.sum()
    mean_loss.backward()
    out = {}
    for k, v in tP.items():
        if isinstance(v, dict):
            for kk, vv in v.items():
                out['%s/%s' % (k, kk)] = vv.grad.double().numpy()
        else:
            out[k] = v.grad.double().numpy()
    return out, float(mean_loss), logits.detach().double().numpy()

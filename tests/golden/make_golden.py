#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the float64 oracle (oracle/).  The reference ships no
fixtures for this path and cannot be executed offline, so these vectors pin
the oracle's restatement (regression pins + closed-form known answers), not TensorFlow.
Run from the repo root:  python tests/golden/make_golden.py"""
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import fbank as ofb, ctc as octc, dfcnn, optim as oopt, prenet as opn, amlm as oam  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def fbank_cases():
    rng = np.random.default_rng(1234)
    n = 8000
    sigs = {
        'zero': np.zeros(n, dtype=np.float32),
        'impulse': np.eye(1, n, 1000, dtype=np.float32)[0],
        # a chirp, not a stationary sine: with a 10 ms hop a 1 kHz sine gives identical frames, every column is
        # constant up to rounding and the standardised output is noise/noise (ill-conditioned, not a parity vector)
        'chirp': (0.5 * np.sin(2 * np.pi * (300 * np.arange(n) / 16000 + 2000 * (np.arange(n) / 16000) ** 2))).astype(np.float32),
        'gauss': (0.1 * rng.standard_normal(n)).astype(np.float32),
        'short': (0.1 * rng.standard_normal(300)).astype(np.float32),
    }
    out = {}
    for k, s in sigs.items():
        out['sig_' + k] = s
        out['feat_' + k] = ofb.compute_fbank_from_api(s.astype(np.float64), 16000, nfilt=200).astype(np.float32)
    np.savez_compressed(os.path.join(OUT, 'fbank.npz'), **out)


def ctc_cases():
    rng = np.random.default_rng(7)
    T, B, V = 12, 4, 9
    x = (rng.standard_normal((T, B, V)) * 2).astype(np.float32)
    labels = [[1, 1, 2], [3], [], [4, 5, 4, 4]]
    seq = [12, 7, 5, 10]
    loss, grad = octc.ctc_loss_and_grad(x, labels, seq, V - 1)
    dec, neg = octc.ctc_greedy_decode(x, seq)
    lab = np.zeros((B, 64), dtype=np.int32)
    for b, l in enumerate(labels):
        lab[b, :len(l)] = l
    dec_pad = np.full((B, T), -1, dtype=np.int32)
    for b, d in enumerate(dec):
        dec_pad[b, :len(d)] = d
    # closed forms on uniform inputs (tests/test_oracle_cpu.py::test_ctc_closed_form)
    closed = np.array([math.log(4), -math.log(3.0 / 16), -math.log(1.0 / 64)])
    np.savez_compressed(os.path.join(OUT, 'ctc.npz'), logits=x, labels=lab, label_len=np.array([len(l) for l in labels]),
                        seq_len=np.array(seq), loss=loss, grad=grad.astype(np.float32), decoded=dec_pad,
                        decoded_len=np.array([len(d) for d in dec]), neg_sum=neg, closed_form_losses=closed)


def model_case(model, widths, tag):
    B, T, F, V = 2, 32, 16, 12
    g = dfcnn.graph(model, V, widths, feat=F)
    P = dfcnn.init_params(g, seed=3, perturb=True)
    P = {l: {k: v.astype(np.float32).astype(np.float64) for k, v in d.items()} for l, d in P.items()}
    rng = np.random.default_rng(11)
    x = rng.standard_normal((B, T, F, 1)).astype(np.float32)
    target = np.zeros((B, 64), dtype=np.int32)
    target[:, :2] = rng.integers(1, V - 1, (B, 2))
    seq = [4, 3]
    r = dfcnn.train_step_oracle(g, P, x.astype(np.float64), seq, target)
    out = {'x': x, 'target': target, 'seq': np.array(seq), 'logits': r['logits'].astype(np.float32),
           'loss': r['loss'], 'label_err': np.array(r['label_err'])}
    dec = np.full((B, T // 8), -1, dtype=np.int32)
    for b, d in enumerate(r['decoded']):
        dec[b, :len(d)] = d
    out['decoded'] = dec
    for l in P:
        for k in P[l]:
            out['p/%s/%s' % (l, k)] = P[l][k].astype(np.float32)
            out['g/%s/%s' % (l, k)] = r['grads'][l][k].astype(np.float32)
    np.savez_compressed(os.path.join(OUT, 'model_%s.npz' % tag), **out)


def misc():
    x = np.arange(30, dtype=np.float64).reshape(10, 3)
    lr = np.array([oopt.polynomial_decay(7e-4, s) for s in (0, 1, 4999, 5000, 5001)])
    np.savez_compressed(os.path.join(OUT, 'misc.npz'), lfr_in=x, lfr_out=ofb.build_LFR_features(x, 4, 3),
                        lr_steps=np.array([0, 1, 4999, 5000, 5001]), lr=lr, vocab_sizes=np.array([1536, 1424, 6345]))


def prenet_case():
    """end2end pre-net (oracle/prenet.py): B=2, T=16 stacked frames of 320 -> pre_out [2, 4, 80, 64] and all gradients."""
    rng = np.random.default_rng(21)
    B, T, F = 2, 16, 320
    P = {k: np.asarray(v, np.float32).astype(np.float64) for k, v in opn.init_params(seed=9).items()}
    x = rng.standard_normal((B, T, F)).astype(np.float32)
    dout = rng.standard_normal((B, T // 4, F // 4, 64)).astype(np.float32)
    out, grads, inter = opn.forward_backward(P, x.astype(np.float64), dout.astype(np.float64))
    z = {'x': x, 'dout': dout, 'pre_out': out.astype(np.float32), 'x2': inter['x2'].astype(np.float32)}
    for k in P:
        z['p/' + k] = P[k].astype(np.float32)
        z['g/' + k] = grads[k].astype(np.float32)
    np.savez_compressed(os.path.join(OUT, 'prenet.npz'), **z)


AMLM = dict(vp=24, vh=41, feat=16, widths=(4, 8, 16, 4, 128), heads=2, blocks=2, pos_max=16, B=3, T=64)


def amlm_case():
    """joint acoustic + language model step (oracle/amlm.py, am_lm_model.py): both logit tensors, the three losses and a
    selection of gradients from both halves."""
    c = AMLM
    P, ops = oam.init_params(c['vp'], c['vh'], feat=c['feat'], widths=c['widths'], heads=c['heads'], blocks=c['blocks'],
                             pos_max=c['pos_max'], seed=0, perturb=True)
    P = {'am': {l: {k: np.asarray(v, np.float32).astype(np.float64) for k, v in d.items()} for l, d in P['am'].items()},
         'lm': {k: ({kk: np.asarray(vv, np.float32).astype(np.float64) for kk, vv in v.items()} if isinstance(v, dict)
                    else np.asarray(v, np.float32).astype(np.float64)) for k, v in P['lm'].items()}}
    rng = np.random.default_rng(3)
    x = rng.standard_normal((c['B'], c['T'], c['feat'])).astype(np.float32)
    tp = np.zeros((c['B'], 8), dtype=np.int32)
    tp[0, :3] = [3, 0, 7]; tp[1, :2] = [5, 5]; tp[2, :4] = [1, 2, 2, 9]
    tl, wl = np.array([3, 2, 4]), np.array([8, 6, 7])
    out = oam.train_step(P, ops, x.astype(np.float64)[..., None], list(wl), tp, list(tl), c['heads'], c['blocks'])
    z = {'x': x, 'target_py': tp, 'target_len': tl, 'wav_len': wl, 'am_logits': out['am_logits'].astype(np.float32),
         'lm_logits': out['lm_logits'].astype(np.float32),
         'losses': np.array([out['am_mean_loss'], out['lm_mean_loss'], out['mean_loss']])}
    for l, d in P['am'].items():
        for k, v in d.items():
            z['pa/%s/%s' % (l, k)] = v.astype(np.float32)
            z['ga/%s/%s' % (l, k)] = out['grads']['am'][l][k].astype(np.float32)
    for k, v in P['lm'].items():
        for kk, vv in (v.items() if isinstance(v, dict) else [(None, v)]):
            name = k if kk is None else '%s/%s' % (k, kk)
            g = out['grads']['lm'][k] if kk is None else out['grads']['lm'][k][kk]
            z['pl/' + name] = np.asarray(vv).astype(np.float32)
            z['gl/' + name] = np.asarray(g).astype(np.float32)
    np.savez_compressed(os.path.join(OUT, 'amlm.npz'), **z)


if __name__ == '__main__':
    fbank_cases(); ctc_cases(); misc(); prenet_case(); amlm_case()
    model_case('m2', (8, 16, 32, 64), 'm2')
    model_case('m1', (8, 16, 32, 64, 8, 32), 'm1')
    print('golden vectors written to', OUT)

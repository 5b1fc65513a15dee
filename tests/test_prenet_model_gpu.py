"""Whole-graph parity of the end2end pre-net engine (end2end/model.py:214-264) against oracle/prenet.py (torch
float64 + autograd), alone and chained in front of the encoder-decoder (speech features -> hanzi logits,
gradients flowing back through embedding_input into the pre-net).  Bars: outputs 1e-3 abs, gradients 1e-3 of
their scale (north_star); biases in front of a batch-statistics BatchNorm have an exactly-zero gradient, those
are compared absolutely."""
import numpy as np
import pytest
import torch

from oracle import prenet as opn
from oracle import transformer as otr

pytestmark = pytest.mark.gpu

# exactly-zero gradients: a conv bias followed directly by a batch-stat BN; and the BN shift of k -- adding a constant
# to every key of a channel adds a per-query constant to its scores, which both softmaxes ignore
ZERO_GRAD = ('q/b', 'k/b', 'v/b', 'f2/b', 'bnk/b')


def f32(P):
    return {k: np.asarray(v, np.float32).astype(np.float64) for k, v in P.items()}


def check_grads(G, R, tol=1e-3):
    worst = 0.0
    for k in R:
        want = np.asarray(R[k], np.float64)
        err = np.abs(np.asarray(G[k], np.float64) - want).max()
        if k in ZERO_GRAD:
            assert np.abs(want).max() < 1e-9, k          # the oracle agrees that it is zero
            assert err < 5e-4, (k, err)
            continue
        r = err / max(1e-6, np.abs(want).max())
        worst = max(worst, r)
        assert r < tol, (k, r)
    return worst


@pytest.mark.parametrize("B,T", [(2, 16), (1, 72)])
def test_prenet_matches_oracle(B, T):
    from asr_dfcnn_transformer_amd.prenet_engine import PreNetEngine
    rng = np.random.default_rng(0)
    F = 320
    P = f32(opn.init_params(seed=3))
    x = rng.standard_normal((B, T, F)).astype(np.float32)
    dout = rng.standard_normal((B, T // 4, F // 4, 64)).astype(np.float32)
    ref, R, inter = opn.forward_backward(P, x.astype(np.float64), dout.astype(np.float64))
    eng = PreNetEngine(B, T, F)
    eng.load_params(P)
    out = eng.forward(torch.tensor(x, device='cuda'))
    eng.backward(torch.tensor(dout, device='cuda').view(B, T // 4, -1))
    torch.cuda.synchronize()
    got = out.cpu().numpy().reshape(ref.shape)
    print('pre_out err %.3e (scale %.2f)' % (np.abs(got - ref).max(), np.abs(ref).max()))
    assert np.abs(got - ref).max() < 1e-3
    for name, buf in (('x2', eng.x2), ('out', eng.out), ('f1', eng.f1n)):
        e = np.abs(buf.interior().cpu().numpy() - inter[name]).max()
        print('  %-4s err %.3e' % (name, e))
        assert e < 1e-3, name
    for p in (eng.x2, eng.out, eng.f1n, eng.cat, eng.x1s):                 # borders of conv inputs stay zero
        assert float(p.view()[:, 0].abs().max()) == 0 and float(p.view()[:, :, 0].abs().max()) == 0
    worst = check_grads(eng.grads_dict(), R)
    print('prenet worst grad rel err %.3e' % worst)
    # same step twice: bitwise identical gradients (fixed-order reductions, no atomics)
    g1 = eng.grad.clone()
    eng.forward(torch.tensor(x, device='cuda'))
    eng.backward(torch.tensor(dout, device='cuda').view(B, T // 4, -1))
    assert torch.equal(g1, eng.grad)


def test_prenet_plus_encoder_decoder_matches_oracle():
    """x [N, T, 320] -> pre_net -> embedding_input dense/LN -> encoder -> decoder -> loss, one backward pass."""
    from asr_dfcnn_transformer_amd.prenet_engine import PreNetEngine
    from asr_dfcnn_transformer_amd.transformer_engine import E2EEngine
    rng = np.random.default_rng(1)
    N, T, F, L, C, H, blocks, Vout, pos_max = 2, 24, 320, 5, 128, 2, 2, 17, 16
    Tp, Din = T // 4, (F // 4) * 64
    Pp = f32(opn.init_params(seed=4))
    Pe = otr.init_e2e(Din, Vout, C, H, blocks, pos_max, seed=5, perturb=True, tie=True)
    Pe = {k: ({kk: np.asarray(vv, np.float32).astype(np.float64) for kk, vv in v.items()} if isinstance(v, dict)
              else np.asarray(v, np.float32).astype(np.float64)) for k, v in Pe.items()}
    for i in range(blocks):
        for k in ('wq', 'wk', 'wv', 'wo'):
            Pe['dec%d' % i][k] = Pe['enc%d' % i][k]
    for k in ('w1', 'b1', 'w2', 'b2'):
        Pe['dec_ffn'][k] = Pe['enc_ffn'][k]
    x = rng.standard_normal((N, T, F)).astype(np.float32)
    y_in = rng.integers(1, Vout, (N, L))
    y_tgt = rng.integers(1, Vout, (N, L)); y_tgt[1, L - 1] = 0
    # oracle: torch pre-net (autograd) feeding the numpy encoder-decoder
    Pt = opn.to_torch(Pp)
    feats = opn.forward(Pt, torch.tensor(x, dtype=torch.float64))
    ref = otr.e2e_step(Pe, feats.detach().numpy().reshape(N, Tp, Din), y_in, y_tgt, H, blocks, tie=True)
    feats.backward(torch.tensor(ref['dx_feat'].reshape(feats.shape)))
    Rp = {k: v.grad.numpy() for k, v in Pt.items()}
    # device
    pre = PreNetEngine(N, T, F)
    pre.load_params(Pp)
    eng = E2EEngine(din=Din, vout=Vout, N=N, T=Tp, L=L, C=C, heads=H, blocks=blocks, pos_max=pos_max, tie=True, need_dx=True)
    eng.load_params(eng.flat_from_oracle(Pe))
    logits = eng.forward(pre.forward(torch.tensor(x, device='cuda')), y_in, y_tgt)
    eng.backward()
    pre.backward(eng.dx_feat)
    torch.cuda.synchronize()
    got = logits.cpu().numpy().reshape(N, L, -1)[:, :, :Vout]
    print('logits err %.3e' % np.abs(got - ref['logits']).max())
    assert np.abs(got - ref['logits']).max() < 1e-3
    ml, _ = eng.fetch()
    assert abs(ml - ref['mean_loss']) < 1e-3
    e = np.abs(eng.dx_feat.cpu().numpy().reshape(N, Tp, Din) - ref['dx_feat']).max() / np.abs(ref['dx_feat']).max()
    print('dL/d(features) rel err %.3e' % e)
    assert e < 1e-3
    print('prenet worst grad rel err (through the whole model) %.3e' % check_grads(pre.grads_dict(), Rp))


def test_transformer_model_shim_from_stacked_frames():
    """Transformer_Model.run with the reference's x_input [batch, T, 4*dimension] (model.py:203): pre-net + encoder +
    decoder + one Adam over all variables; the loss of a repeated batch goes down."""
    from asr_dfcnn_transformer_amd.e2e_model import Transformer_Model, E2EHparams
    hp = E2EHparams()
    hp.batch_size, hp.num_blocks, hp.hidden_units, hp.num_heads, hp.position_max_length = 2, 2, 128, 2, 32
    m = Transformer_Model(hp, label_vocab_size=23).build_transformer()
    rng = np.random.default_rng(6)
    feed = {m.x_input: rng.standard_normal((2, 32, 320)).astype(np.float32), m.y_input: rng.integers(1, 23, (2, 6)),
            m.y_target: rng.integers(1, 23, (2, 6)), m.learning_rate: 1e-3}
    losses = [m.run([m.mean_loss, m.train_op], feed)[0] for _ in range(8)]
    print('losses', ['%.3f' % v for v in losses])
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]
    assert m.prenet.global_step == 8 and m.engine.global_step == 8 and m.engine.T == 8
    with pytest.raises(ValueError):
        m.run(m.mean_loss, {m.x_input: np.zeros((2, 8, 77), np.float32), m.y_input: feed[m.y_input]})

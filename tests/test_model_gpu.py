"""Whole-model parity on the GPU: the engine's forward / CTC / greedy decode / backward /
Adam step against the float64 oracle (oracle/dfcnn.py) on seeded tiny configurations of
the three reference graphs.  Bars: logits and loss within 1e-3 (north_star), decoded ids
bit-exact, gradients within 1e-3 of their scale."""
import math

import numpy as np
import pytest
import torch

from oracle import dfcnn, optim as oopt

pytestmark = pytest.mark.gpu

CASES = [
    ('m2', (8, 16, 32, 64), 2, 32, 16, 12),
    ('m2', (16, 32, 32, 64), 3, 64, 24, 20),
    ('m1', (8, 16, 32, 64, 8, 32), 2, 32, 16, 12),
    ('m3', (8, 16, 32, 64), 2, 48, 16, 12),
    ('small', (32, 32, 32, 32), 4, 64, 24, 20),      # BASELINE configs[0]: DFCNN-small, 4 cells x 32 channels, batch 4
]


def rel(got, want):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    return np.abs(got - want).max() / max(1e-6, np.abs(want).max())


@pytest.mark.parametrize("model,widths,B,T,F,V", CASES)
def test_train_step_matches_oracle(model, widths, B, T, F, V):
    from asr_dfcnn_transformer_amd.engine import DFCNNEngine, graph
    ops_ref = dfcnn.graph(model, V, widths, feat=F)
    assert [tuple(o) for o in graph(model, V, widths, F)] == [tuple(o) for o in ops_ref]
    P = dfcnn.init_params(ops_ref, seed=3, perturb=True)
    P = {l: {k: v.astype(np.float32).astype(np.float64) for k, v in d.items()} for l, d in P.items()}
    rng = np.random.default_rng(11)
    x = rng.standard_normal((B, T, F, 1)).astype(np.float32)
    x[-1, T - T // 4:] = 0                      # zero-padded tail like a short utterance
    T8 = T // 8
    target = np.zeros((B, 64), dtype=np.int32)
    seq = []
    for b in range(B):
        L = min(2, T8 - 1)
        target[b, :L] = rng.integers(1, V - 1, L)
        if b == 0 and L >= 2:
            target[b, 1] = 0                     # a real 0 id vanishes
        seq.append(T8 if b % 2 == 0 else max(2, T8 - 1))
    ref = dfcnn.train_step_oracle(ops_ref, P, x.astype(np.float64), seq, target)

    eng = DFCNNEngine(model=model, vocab=V, B=B, T=T, F=F, widths=widths, seed=0)
    eng.load_params(P)
    logits = eng.forward(torch.tensor(x.reshape(B, T, F), device='cuda'))
    eng.set_targets(seq, target)
    eng.loss_and_decode()
    eng.backward()
    torch.cuda.synchronize()
    got_logits = logits.cpu().numpy()
    print('logits max abs err', np.abs(got_logits - ref['logits']).max())
    assert np.abs(got_logits - ref['logits']).max() < 1e-3
    loss = eng.loss.cpu().numpy()
    print('loss', loss, ref['loss'][:, 0])
    assert np.abs(loss - ref['loss'][:, 0]).max() < 1e-3
    mean_loss, label_err = eng.fetch_scalars()
    assert abs(mean_loss - ref['mean_loss']) < 1e-3
    assert eng.decoded_lists() == ref['decoded']                     # bit-exact ids
    assert abs(label_err - ref['label_err']) < 1e-5 or (math.isinf(label_err) and math.isinf(ref['label_err']))
    G = eng.grads_dict()
    worst = 0.0
    for layer in P:
        for key in P[layer]:
            r = rel(G[layer][key], ref['grads'][layer][key])
            worst = max(worst, r)
            assert r < 1e-3, (layer, key, r)
    print('worst relative gradient error', worst)

    # one TF-Adam step on the polynomial-decay schedule
    lr = eng.apply_adam()
    assert abs(lr - oopt.polynomial_decay(7e-4, 0)) < 1e-12
    torch.cuda.synchronize()
    newP = eng.params_dict()
    for layer in P:
        for key in P[layer]:
            th, _, _ = oopt.adam_tf_step(P[layer][key], ref['grads'][layer][key], 0.0, 0.0, lr, 1)
            # first Adam step moves every weight by ~lr*sign(g); compare the update, not theta
            upd_ref = th - P[layer][key]
            upd = newP[layer][key].astype(np.float64) - P[layer][key]
            big = np.abs(ref['grads'][layer][key]) > 1e-6       # sign(g) ill-defined at ~0 gradients
            if big.any():
                assert np.abs(upd - upd_ref)[big].max() < 2e-6, (layer, key)
    assert eng.global_step == 1


def test_learning_rate_schedule_points():
    from asr_dfcnn_transformer_amd.engine import DFCNNEngine
    eng = DFCNNEngine(model='m2', vocab=12, B=1, T=16, F=16, widths=(8, 8, 8, 8))
    for step in (0, 1, 4999, 5000, 5001, 12345):
        assert abs(eng.current_learning_rate(step) - oopt.polynomial_decay(7e-4, step)) < 1e-15


def test_infeasible_label_raises_like_tf():
    from asr_dfcnn_transformer_amd.engine import DFCNNEngine
    eng = DFCNNEngine(model='m2', vocab=12, B=1, T=16, F=16, widths=(8, 8, 8, 8))
    with pytest.raises(ValueError):
        eng.set_targets([2], np.array([[3, 3, 0, 0]]))      # needs 3 frames, has 2


def test_step_is_bitwise_reproducible():
    from asr_dfcnn_transformer_amd.engine import DFCNNEngine
    rng = np.random.default_rng(5)
    B, T, F, V = 2, 64, 24, 20
    x = torch.tensor(rng.standard_normal((B, T, F)).astype(np.float32), device='cuda')
    tgt = np.zeros((B, 64), dtype=np.int32)
    tgt[:, :3] = rng.integers(1, V - 1, (B, 3))
    outs = []
    for _ in range(2):
        eng = DFCNNEngine(model='m2', vocab=V, B=B, T=T, F=F, widths=(16, 32, 32, 64), seed=1)
        eng.forward(x)
        eng.set_targets([8, 8], tgt)
        eng.loss_and_decode()
        eng.backward()
        torch.cuda.synchronize()
        outs.append(eng.grad.clone())
    assert torch.equal(outs[0], outs[1])


def test_session_style_model_trains_and_checkpoints(tmp_path):
    """CNNCTCModel.run with the reference's feed/fetch lists (lm_and_am/train.py:59-69): loss goes
    down on a fixed synthetic batch, decode-only fetch works, checkpoint round-trips."""
    from asr_dfcnn_transformer_amd.acoustic_model import CNNCTCModel
    from asr_dfcnn_transformer_amd.hparams import AmLmHparams
    from asr_dfcnn_transformer_amd import train as tr
    hp = AmLmHparams().args
    hp.feature_max_length, hp.feature_dim, hp.am_batch_size, hp.am_lr = 64, 16, 2, 1e-3
    m = CNNCTCModel(hp, 12, 6345, widths=(8, 16, 32, 64))
    rng = np.random.default_rng(0)
    x = rng.standard_normal((2, 64, 16, 1)).astype(np.float32)
    y = np.zeros((2, 64), dtype=np.int32)
    y[:, :3] = rng.integers(1, 10, (2, 3))
    feed = {m.wav_input: x, m.logits_length: np.array([8, 7]), m.target_py: y, m.target_length: np.array([3, 3]),
            m.drop_rate: 0.5}
    losses = []
    for _ in range(25):
        loss, mean_loss, lr, summary, label_err, _ = m.run(
            [m.loss, m.mean_loss, m.current_learning, m.summary, m.label_err, m.train_op], feed_dict=feed)
        losses.append(mean_loss)
    assert loss.shape == (2, 1) and losses[-1] < 0.7 * losses[0], losses
    assert m.global_step == 25 and 0 < lr <= 1e-3
    dec = m.run(m.decoded[0], {m.wav_input: x, m.logits_length: np.array([8, 7])})
    assert dec.dense_shape[0] == 2 and dec.indices.shape[1] == 2
    p = str(tmp_path / 'ck.pt')
    tr.save_checkpoint(m, p)
    m2 = CNNCTCModel(hp, 12, 6345, widths=(8, 16, 32, 64), seed=5)
    tr.load_checkpoint(m2, p)
    assert torch.equal(m2.engine.theta, m.engine.theta) and m2.global_step == 25


@pytest.mark.parametrize("model,widths", [('m2', (16, 32, 32, 64)), ('m1', (8, 16, 32, 64, 8, 32)), ('m3', (8, 16, 32, 64))])
def test_stream_and_fusion_modes_give_the_same_bits(model, widths):
    """The same step with the backward on one stream or two, and with the backward prologues fused into the data-gradient
    epilogues or run as passes of their own: dZ is the same arithmetic either way, so the weight / bias gradients are
    bitwise equal; the three BN / bias channel sums of fused cells are folded in another (fixed) order -> 1e-6."""
    from asr_dfcnn_transformer_amd.engine import DFCNNEngine
    rng = np.random.default_rng(5)
    B, T, F, V = 2, 64, 24, 20
    x = torch.tensor(rng.standard_normal((B, T, F)).astype(np.float32), device='cuda')
    tgt = np.zeros((B, 64), dtype=np.int32)
    tgt[:, :3] = rng.integers(1, V - 1, (B, 3))
    grads = {}
    for dual in ('1', '0'):
        for fuse in ('1', '0'):
            eng = DFCNNEngine(model=model, vocab=V, B=B, T=T, F=F, widths=widths, seed=1, dual_stream=dual == '1',
                              fuse_prologues=fuse == '1')
            assert bool(eng.fuse) == (fuse == '1') and (eng.side is not None) == (dual == '1')
            for _ in range(2):
                eng.forward(x); eng.set_targets([8, 8], tgt); eng.loss_and_decode(); eng.backward()
            torch.cuda.synchronize()
            grads[(dual, fuse)] = (eng.grad.clone(), eng)
    assert torch.equal(grads[('1', '1')][0], grads[('0', '1')][0])          # streams: same bits
    assert torch.equal(grads[('1', '0')][0], grads[('0', '0')][0])
    a, e = grads[('1', '1')]
    b, _ = grads[('1', '0')]
    for (layer, key), (off, shape) in e.entries.items():
        n = int(np.prod(shape))
        ga, gb = a[off:off + n], b[off:off + n]
        if key in ('w', 'w1', 'w2', 'b1', 'b2'):
            assert torch.equal(ga, gb), (layer, key)
        else:
            assert (ga - gb).abs().max().item() <= 1e-6 * max(1e-6, gb.abs().max().item()), (layer, key)


def test_se_squeeze_made_by_the_branch_conv_gives_the_same_step():
    """se_sums=True (default): the forward launch of an SE block's branch cell also writes the block's squeeze sums
    (asr_tap_gemm_wino_sums -> asr_se_fwd_sums) and the data-gradient of the cell that reads the block's output writes the block's first
    backward reduction (asr_tap_gemm_wino_sesum -> asr_se_bwd_cell_sums); se_sums=False: both are passes of their own over the plane.
    The same sums in another order: logits, loss and every gradient agree to rounding; the planes in front of the first SE block are bitwise equal."""
    from asr_dfcnn_transformer_amd.engine import DFCNNEngine
    rng = np.random.default_rng(9)
    B, T, F, V = 2, 64, 200, 20
    x = torch.tensor(rng.standard_normal((B, T, F)).astype(np.float32), device='cuda')
    tgt = np.zeros((B, 64), dtype=np.int32); tgt[:, :3] = rng.integers(1, V - 1, (B, 3))
    out = []
    for flag in (True, False):
        eng = DFCNNEngine(model='m2', vocab=V, B=B, T=T, F=F, widths=(32, 64, 64, 64), seed=2, se_sums=flag)
        assert bool(eng.se_sums) == flag and (not flag or len(eng.se_sums) == 5) and (not flag or len(eng.se_xsum) == 5)
        logits = eng.forward(x).clone()
        eng.set_targets([8, 8], tgt); eng.loss_and_decode(); eng.backward()
        torch.cuda.synchronize()
        out.append((logits, eng.loss.clone(), eng.grad.clone(), eng.y['h1_1'].buf.clone()))
    (l0, s0, g0, y0), (l1, s1, g1, y1) = out
    assert torch.equal(y0, y1)
    assert (l0 - l1).abs().max().item() < 1e-4
    assert (s0 - s1).abs().max().item() < 1e-4 * max(1.0, s0.abs().max().item())
    assert (g0 - g1).abs().max().item() < 1e-4 * max(1e-6, g0.abs().max().item())


@pytest.mark.parametrize("model,widths", [('m2', (32, 64, 64, 64)), ('m1', (32, 64, 64, 64, 32, 32))])
def test_compact_pool_forms_give_the_same_step(model, widths):
    """compact_pool=True keeps no pre-pool activation plane for pooled Winograd cells (max pool: activation at the maximum + position;
    average pool, round 5: activation sum + ReLU signs).  Logits, loss and every weight / bias gradient are the bits of the plane form;
    the BN scale gradient of an AVERAGE-pooled cell is the same sum with one rounding per window instead of four (1e-6)."""
    from asr_dfcnn_transformer_amd.engine import DFCNNEngine
    rng = np.random.default_rng(13)
    B, T, F, V = 2, 64, 200, 20
    x = torch.tensor(rng.standard_normal((B, T, F)).astype(np.float32), device='cuda')
    tgt = np.zeros((B, 64), dtype=np.int32); tgt[:, :3] = rng.integers(1, V - 1, (B, 3))
    out = []
    for flag in (True, False):
        eng = DFCNNEngine(model=model, vocab=V, B=B, T=T, F=F, widths=widths, seed=2, compact_pool=flag)
        assert len(eng.compact) == (2 if flag else 0)
        logits = eng.forward(x).clone()
        eng.set_targets([8, 8], tgt); eng.loss_and_decode(); eng.backward()
        torch.cuda.synchronize()
        out.append((logits, eng.grad.clone(), eng))
    (l0, g0, e), (l1, g1, _) = out
    assert torch.equal(l0, l1)
    for (layer, key), (off, shape) in e.entries.items():
        n = int(np.prod(shape))
        a, b = g0[off:off + n], g1[off:off + n]
        if key == 'gamma' and model == 'm2':
            assert (a - b).abs().max().item() <= 1e-6 * max(1e-6, b.abs().max().item()), (layer, key)
        else:
            assert torch.equal(a, b), (layer, key)

"""Whole-graph parity of the Transformer engines (LM and encoder-decoder) against the float64
oracle: logits and loss within 1e-3 (north_star), argmax predictions exact, gradients within
1e-3 of their scale, plus one TF-Adam step."""
import numpy as np
import pytest
import torch

from oracle import transformer as otr

pytestmark = pytest.mark.gpu


def rel(got, want):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    return np.abs(got - want).max() / max(1e-6, np.abs(want).max())


def f32(P):
    return {k: (f32(v) if isinstance(v, dict) else np.asarray(v, np.float32).astype(np.float64)) for k, v in P.items()}


def flat_grads(G):
    out = {}
    for k, v in G.items():
        if isinstance(v, dict):
            for kk, vv in v.items():
                out['%s/%s' % (k, kk)] = vv
        else:
            out[k] = v
    return out


def test_lm_step_matches_oracle():
    from asr_dfcnn_transformer_amd.transformer_engine import LMEngine
    rng = np.random.default_rng(0)
    N, T, C, H, blocks, Vin, Vout, pos_max = 3, 9, 128, 2, 3, 21, 13, 12
    P = f32(otr.init_lm(Vin, Vout, C, H, blocks, pos_max, seed=1, perturb=True))
    x = rng.integers(1, Vin, (N, T)); x[0, T - 3:] = 0
    y = rng.integers(1, Vout, (N, T)); y[0, T - 3:] = 0
    ref = otr.lm_step(P, x, y, H, blocks)
    eng = LMEngine(vin=Vin, vout=Vout, N=N, T=T, C=C, heads=H, blocks=blocks, pos_max=pos_max)
    eng.load_params(eng.flat_from_oracle(P))
    logits = eng.forward(x, y)
    eng.backward()
    torch.cuda.synchronize()
    got = logits.cpu().numpy().reshape(N, T, -1)[:, :, :Vout]
    print('lm logits err', np.abs(got - ref['logits']).max())
    assert np.abs(got - ref['logits']).max() < 1e-3
    ml, acc = eng.fetch()
    assert abs(ml - ref['mean_loss']) < 1e-3 and abs(acc - ref['acc']) < 1e-6
    assert np.array_equal(eng.preds.cpu().numpy().reshape(N, T), ref['preds'])
    G, R = eng.grads_dict(), flat_grads(ref['grads'])
    worst = 0
    for k in R:
        r = rel(G[k], R[k]); worst = max(worst, r)
        assert r < 1e-3, (k, r)
    print('lm worst grad rel err', worst)
    lr = eng.apply_adam()
    assert abs(lr - 5e-5) < 1e-12 and eng.global_step == 1


@pytest.mark.parametrize("tie", [True, False])
def test_e2e_step_matches_oracle(tie):
    from asr_dfcnn_transformer_amd.transformer_engine import E2EEngine
    rng = np.random.default_rng(1)
    N, T, L, Din, C, H, blocks, Vout, pos_max = 2, 11, 6, 24, 128, 2, 2, 15, 16
    P = f32(otr.init_e2e(Din, Vout, C, H, blocks, pos_max, seed=2, perturb=True, tie=tie))
    if tie:      # f32() broke the object sharing; restore it
        for i in range(blocks):
            for k in ('wq', 'wk', 'wv', 'wo'):
                P['dec%d' % i][k] = P['enc%d' % i][k]
        for k in ('w1', 'b1', 'w2', 'b2'):
            P['dec_ffn'][k] = P['enc_ffn'][k]
    xf = rng.standard_normal((N, T, Din)).astype(np.float32)
    y_in = rng.integers(1, Vout, (N, L))
    y_tgt = rng.integers(1, Vout, (N, L)); y_tgt[1, L - 2:] = -1; y_tgt[0, L - 1] = 0
    ref = otr.e2e_step(P, xf.astype(np.float64), y_in, y_tgt, H, blocks, tie=tie)
    eng = E2EEngine(din=Din, vout=Vout, N=N, T=T, L=L, C=C, heads=H, blocks=blocks, pos_max=pos_max, tie=tie)
    eng.load_params(eng.flat_from_oracle(P))
    logits = eng.forward(torch.tensor(xf, device='cuda'), y_in, y_tgt)
    eng.backward()
    torch.cuda.synchronize()
    got = logits.cpu().numpy().reshape(N, L, -1)[:, :, :Vout]
    print('e2e logits err', np.abs(got - ref['logits']).max())
    assert np.abs(got - ref['logits']).max() < 1e-3
    ml, acc = eng.fetch()
    assert abs(ml - ref['mean_loss']) < 1e-3 and abs(acc - ref['acc']) < 1e-6
    G, R = eng.grads_dict(), flat_grads(ref['grads'])
    worst = 0
    for k in R:
        r = rel(G[k], R[k]); worst = max(worst, r)
        assert r < 1e-3, (k, r)
    print('e2e worst grad rel err', worst)


def test_e2e_with_pinyin_ids_runs_and_learns():
    """configs[3] wiring (pinyin ids -> hanzi): loss decreases on a fixed batch."""
    from asr_dfcnn_transformer_amd.transformer_engine import E2EEngine
    rng = np.random.default_rng(2)
    N, T, L = 4, 20, 20
    eng = E2EEngine(vin=50, vout=40, N=N, T=T, L=L, C=128, heads=2, blocks=2, pos_max=32, lr=2e-3)
    x = rng.integers(1, 50, (N, T)); y = rng.integers(3, 40, (N, L))
    y_in = np.concatenate([np.ones((N, 1), dtype=np.int64), y[:, :-1]], axis=1)
    losses = []
    for _ in range(30):
        eng.forward(x, y_in, y)
        eng.backward()
        eng.apply_adam()
        losses.append(eng.fetch()[0])
    assert losses[-1] < 0.8 * losses[0], losses


def test_language_model_shim_session_api():
    """Language_Model with the reference's feed/fetch names (lm_and_am/train.py:138-141); batches
    of different max length share one engine because the causal mask hides trailing pads."""
    from asr_dfcnn_transformer_amd.language_model import Language_Model
    from asr_dfcnn_transformer_amd.hparams import AmLmHparams
    hp = AmLmHparams().args
    hp.hidden_units, hp.num_heads, hp.num_blocks, hp.position_max_length, hp.lm_batch_size, hp.lm_lr = 128, 2, 2, 16, 4, 2e-3
    m = Language_Model(hp, 30, 25)
    rng = np.random.default_rng(3)
    x = rng.integers(1, 30, (4, 9)); y = rng.integers(1, 25, (4, 9))
    x[0, 6:] = 0; y[0, 6:] = 0
    losses = [m.run([m.mean_loss, m.current_learning, m.train_op], {m.x: x, m.y: y})[0] for _ in range(25)]
    assert losses[-1] < 0.8 * losses[0]
    p_short = m.run(m.preds, {m.x: x[:, :5]})
    p_long = m.run(m.preds, {m.x: x})
    assert p_short.shape == (4, 5) and np.array_equal(p_short, p_long[:, :5])     # causal: prefix-invariant
    acc = m.run(m.acc, {m.x: x, m.y: y})
    assert 0.0 <= acc <= 1.0


def test_transformer_model_shim_rebuilds_per_shape():
    from asr_dfcnn_transformer_amd.e2e_model import Transformer_Model, E2EHparams
    hp = E2EHparams()
    hp.batch_size, hp.num_blocks, hp.hidden_units, hp.num_heads, hp.position_max_length = 2, 2, 128, 2, 32
    m = Transformer_Model(hp, label_vocab_size=19, input_dim=20).build_transformer()
    rng = np.random.default_rng(4)
    for T, L in ((10, 6), (12, 7)):
        feed = {m.x_input: rng.standard_normal((2, T, 20)).astype(np.float32), m.y_input: rng.integers(1, 19, (2, L)),
                m.y_target: rng.integers(1, 19, (2, L)), m.learning_rate: 5e-4}
        ml, merged, lr, _ = m.run([m.mean_loss, m.merged, m.current_learning, m.train_op], feed)
        assert np.isfinite(ml) and abs(merged['mean_loss'] - ml) < 1e-6
    assert m.engine.global_step == 2 and (m.engine.T, m.engine.L) == (12, 7)


def test_am_to_lm_inference_pipeline():
    """lm_and_am/test.py:44-61 wiring: AM greedy ids -> dense (0-filled) -> LM argmax; batched result equals
    running the two models by hand, and the accuracy meter follows test.py:74-90."""
    from asr_dfcnn_transformer_amd.acoustic_model import CNNCTCModel
    from asr_dfcnn_transformer_amd.language_model import Language_Model
    from asr_dfcnn_transformer_amd.hparams import AmLmHparams
    from asr_dfcnn_transformer_amd.test_pipeline import SpeechRecognizer, AccuracyMeter, dense_from_sparse
    hp = AmLmHparams().args
    hp.feature_max_length, hp.feature_dim, hp.am_batch_size = 64, 16, 2
    hp.hidden_units, hp.num_heads, hp.num_blocks, hp.position_max_length, hp.lm_batch_size = 128, 2, 2, 16, 2
    am = CNNCTCModel(hp, 12, 25, widths=(8, 16, 32, 64))
    lm = Language_Model(hp, 12, 25)
    rng = np.random.default_rng(0)
    x = rng.standard_normal((2, 64, 16, 1)).astype(np.float32)
    rec = SpeechRecognizer(am, lm)
    py, han = rec.recognize(x, np.array([8, 6]))
    dec = am.run(am.decoded[0], {am.wav_input: x, am.logits_length: np.array([8, 6])})
    dense = dense_from_sparse(dec)
    assert [dense[b, :len(py[b])].tolist() for b in range(2)] == py
    if dense.shape[1]:
        manual = lm.run(lm.preds, {lm.x: dense})
        assert [manual[b, :len(py[b])].tolist() for b in range(2)] == han
    m = AccuracyMeter()
    m.update([1, 2, 3, 4], [1, 2, 4])
    m.update([1, 2], [5, 6, 7, 8, 9])          # more errors than words -> capped at the sentence length
    assert m.words == 6 and m.errors == 1 + 2 and abs(m.accuracy - 0.5) < 1e-12


@pytest.mark.parametrize("which", ["lm", "e2e"])
def test_steps_with_dropout_match_oracle(which):
    """dropout_rate 0.2 at every site of the reference (embedding input, attention weights, MHA output, FFN output) with
    the build's counter-based mask: engines vs the oracle's restatement of the same generator, step 0 and step 1."""
    from asr_dfcnn_transformer_amd.transformer_engine import LMEngine, E2EEngine
    rng = np.random.default_rng(7)
    rate, seed = 0.2, 4242
    if which == "lm":
        N, T, C, H, blocks, Vin, Vout, pos_max = 3, 9, 128, 2, 3, 21, 13, 12
        P = f32(otr.init_lm(Vin, Vout, C, H, blocks, pos_max, seed=1, perturb=True))
        x = rng.integers(1, Vin, (N, T)); x[0, T - 3:] = 0
        y = rng.integers(1, Vout, (N, T)); y[0, T - 3:] = 0
        eng = LMEngine(vin=Vin, vout=Vout, N=N, T=T, C=C, heads=H, blocks=blocks, pos_max=pos_max, dropout_rate=rate, drop_seed=seed)
        eng.load_params(eng.flat_from_oracle(P))
        run_ref = lambda step: otr.lm_step(P, x, y, H, blocks, drop=otr.Drop(rate, seed, step))
        run_dev = lambda: eng.forward(x, y)
        shape = (N, T)
    else:
        N, T, L, Din, C, H, blocks, Vout, pos_max = 2, 11, 6, 24, 128, 2, 2, 15, 16
        P = f32(otr.init_e2e(Din, Vout, C, H, blocks, pos_max, seed=2, perturb=True, tie=True))
        for i in range(blocks):
            for k in ('wq', 'wk', 'wv', 'wo'):
                P['dec%d' % i][k] = P['enc%d' % i][k]
        for k in ('w1', 'b1', 'w2', 'b2'):
            P['dec_ffn'][k] = P['enc_ffn'][k]
        xf = rng.standard_normal((N, T, Din)).astype(np.float32)
        y_in = rng.integers(1, Vout, (N, L)); y_tgt = rng.integers(1, Vout, (N, L))
        eng = E2EEngine(din=Din, vout=Vout, N=N, T=T, L=L, C=C, heads=H, blocks=blocks, pos_max=pos_max, tie=True,
                        dropout_rate=rate, drop_seed=seed)
        eng.load_params(eng.flat_from_oracle(P))
        xd = torch.tensor(xf, device='cuda')
        run_ref = lambda step: otr.e2e_step(P, xf.astype(np.float64), y_in, y_tgt, H, blocks, tie=True, drop=otr.Drop(rate, seed, step))
        run_dev = lambda: eng.forward(xd, y_in, y_tgt)
        shape = (N, L)
    ref0 = run_ref(0)
    nodrop = (otr.lm_step(P, x, y, H, blocks) if which == "lm" else otr.e2e_step(P, xf.astype(np.float64), y_in, y_tgt, H, blocks, tie=True))
    assert np.abs(ref0['logits'] - nodrop['logits']).max() > 1e-2          # the masks do something
    logits = run_dev()
    eng.backward()
    torch.cuda.synchronize()
    got = logits.cpu().numpy().reshape(shape + (-1,))[:, :, :ref0['logits'].shape[-1]]
    print(which, 'logits err (dropout)', np.abs(got - ref0['logits']).max())
    assert np.abs(got - ref0['logits']).max() < 1e-3
    G, R = eng.grads_dict(), flat_grads(ref0['grads'])
    worst = max(rel(G[k], R[k]) for k in R)
    print(which, 'worst grad rel err (dropout)', worst)
    assert worst < 1e-3
    # the masks change with the step counter; evaluation (train=False) has no dropout
    eng.global_step = 1
    got1 = run_dev().cpu().numpy().reshape(shape + (-1,))[:, :, :ref0['logits'].shape[-1]]
    assert np.abs(got1 - run_ref(1)['logits']).max() < 1e-3 and np.abs(got1 - got).max() > 1e-2
    if which == "lm":
        ev = eng.forward(x, y, train=False).cpu().numpy().reshape(shape + (-1,))[:, :, :ref0['logits'].shape[-1]]
        assert np.abs(ev - nodrop['logits']).max() < 1e-3


def test_checkpoint_roundtrip_for_any_engine(tmp_path):
    """save_checkpoint / load_checkpoint: variables + Adam slots + global_step; a resumed engine continues bit-identically;
    a checkpoint of another model is refused."""
    from asr_dfcnn_transformer_amd.train import save_checkpoint, load_checkpoint
    from asr_dfcnn_transformer_amd.transformer_engine import LMEngine
    rng = np.random.default_rng(8)
    mk = lambda blocks=2: LMEngine(vin=17, vout=13, N=2, T=8, C=128, heads=2, blocks=blocks, pos_max=8, dropout_rate=0.1, drop_seed=3)
    x, y = rng.integers(1, 17, (2, 8)), rng.integers(1, 13, (2, 8))
    a = mk()
    for _ in range(2):
        a.forward(x, y); a.backward(); a.apply_adam()
    save_checkpoint(a, str(tmp_path / 'lm.pt'))
    b = mk()
    load_checkpoint(b, str(tmp_path / 'lm.pt'))
    assert b.global_step == 2 and torch.equal(a.theta, b.theta) and torch.equal(a.adam_v, b.adam_v)
    for e in (a, b):
        e.forward(x, y); e.backward(); e.apply_adam()
    assert torch.equal(a.theta, b.theta)
    with pytest.raises(ValueError):
        load_checkpoint(mk(blocks=3), str(tmp_path / 'lm.pt'))

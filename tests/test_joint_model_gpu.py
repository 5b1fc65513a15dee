"""GPU parity of the joint acoustic + language model step (asr_dfcnn_transformer_amd/joint_engine.py; am_lm_model.py, SURVEY.md
8f.2) against oracle/amlm.py on a small configuration: both time-major logit tensors and both losses (1e-3 bar of
north_star; observed ~1e-5), every gradient of both halves (the acoustic trunk's include the language half's contribution
through h7), bitwise reproducibility, and one Adam step."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

VP, VH, F, T, B = 24, 41, 16, 64, 3           # VH = 41: the hanzi projection is padded to 44 columns
WIDTHS = (4, 8, 16, 4, 128)


def _build(dropout=0.0):
    from asr_dfcnn_transformer_amd.joint_engine import AMLMEngine
    from oracle import amlm
    P, ops = amlm.init_params(VP, VH, feat=F, widths=WIDTHS, heads=2, blocks=2, pos_max=16, seed=0, perturb=True)
    eng = AMLMEngine(v_pinyin=VP, v_hanzi=VH, B=B, T=T, F=F, widths=WIDTHS, heads=2, blocks=2, pos_max=16, lr=1e-3,
                     dropout_rate=dropout, drop_seed=5)
    eng.am.load_params(P['am'])
    eng.lm.load_params(eng.lm.flat_from_oracle(P['lm']))
    rng = np.random.default_rng(3)
    x = rng.standard_normal((B, T, F)).astype(np.float32)
    tp = np.zeros((B, 8), dtype=np.int32)
    tp[0, :3] = [3, 0, 7]; tp[1, :2] = [5, 5]; tp[2, :4] = [1, 2, 2, 9]
    return eng, P, ops, x, tp, [3, 2, 4], [8, 6, 7]


def _rel(a, b):
    return float(np.abs(np.asarray(a, dtype=np.float64) - b).max() / max(1e-12, np.abs(b).max()))


def test_joint_step_matches_oracle():
    from oracle import amlm
    eng, P, ops, x, tp, tl, wl = _build()
    ref = amlm.train_step(P, ops, x.astype(np.float64)[..., None], wl, tp, tl, 2, 2)
    xd = torch.from_numpy(x).cuda()
    eng.forward(xd); eng.set_targets(wl, tp, tl); eng.loss_and_decode(); eng.backward()
    torch.cuda.synchronize()
    am_logits = eng.am.logits.cpu().numpy()
    lm_logits = eng.lm.logits.cpu().numpy()[:, :, :VH]
    assert np.abs(am_logits - ref['am_logits']).max() < 1e-3 and np.abs(lm_logits - ref['lm_logits']).max() < 1e-3
    print('logits err', np.abs(am_logits - ref['am_logits']).max(), np.abs(lm_logits - ref['lm_logits']).max())
    am_mean, lm_mean, mean, _ = eng.fetch()
    assert abs(am_mean - ref['am_mean_loss']) < 1e-3 * abs(ref['am_mean_loss'])
    assert abs(lm_mean - ref['lm_mean_loss']) < 1e-3 * abs(ref['lm_mean_loss'])
    assert abs(mean - ref['mean_loss']) < 1e-3 * abs(ref['mean_loss'])
    dec_am, dec_lm = eng.decoded_lists()
    assert dec_am == [list(d) for d in ref['decoded_am']] and dec_lm == [list(d) for d in ref['decoded_lm']]
    # gradients of the language half
    g = eng.lm.grads_dict()
    flat = eng.lm.flat_from_oracle(ref['grads']['lm'])
    worst = 0.0
    for name, want in flat.items():
        r = _rel(g[name], np.asarray(want)); worst = max(worst, r)
        assert r < 2e-3, (name, r)
    # gradients of the acoustic half (trunk gradients include the language half's contribution through h7)
    ga = eng.am.grads_dict()
    for layer, d in ref['grads']['am'].items():
        for key, want in d.items():
            r = _rel(ga[layer][key], want); worst = max(worst, r)
            assert r < 2e-3, (layer, key, r)
    print('worst relative gradient error %.2e' % worst)
    # the language half really feeds the trunk: the acoustic-only gradient of an early conv differs
    from oracle import ctc, dfcnn, nn
    x64 = x.astype(np.float64)[..., None]
    d_am, state = dfcnn.forward(ops, P['am'], x64)
    _, g_am = ctc.ctc_loss_and_grad(ref['am_logits'], amlm.dense_labels(tp, tl), wl, blank=VP - 1)
    only_am = dfcnn.backward(ops, P['am'], state, nn.log_softmax_eps_tm_bwd(d_am, g_am / B))
    assert _rel(only_am['h2']['w'], ref['grads']['am']['h2']['w']) > 1e-2


def test_joint_step_is_reproducible_and_trains():
    eng, P, ops, x, tp, tl, wl = _build()
    xd = torch.from_numpy(x).cuda()

    def run():
        eng.forward(xd); eng.set_targets(wl, tp, tl); eng.loss_and_decode(); eng.backward()
        torch.cuda.synchronize()
        return eng.am.grad.clone(), eng.lm.grad.clone(), eng.fetch()[2]
    a0, l0, m0 = run()
    a1, l1, m1 = run()
    assert torch.equal(a0, a1) and torch.equal(l0, l1) and m0 == m1
    losses = [m0]
    for _ in range(5):
        eng.apply_adam()
        losses.append(run()[2])
    assert losses[-1] < losses[0], losses
    pad_bias = eng.lm.p('out_b')[VH:].cpu().numpy()
    assert np.all(pad_bias < -1e29)                  # padding columns stay out of the softmax through the updates


def test_joint_step_with_dropout_matches_oracle():
    from oracle import amlm
    from oracle.transformer import Drop
    eng, P, ops, x, tp, tl, wl = _build(dropout=0.2)
    ref = amlm.train_step(P, ops, x.astype(np.float64)[..., None], wl, tp, tl, 2, 2, drop=Drop(0.2, 5, 0))
    eng.forward(torch.from_numpy(x).cuda()); eng.set_targets(wl, tp, tl); eng.loss_and_decode(); eng.backward()
    torch.cuda.synchronize()
    assert np.abs(eng.lm.logits.cpu().numpy()[:, :, :VH] - ref['lm_logits']).max() < 1e-3
    ga = eng.am.grads_dict()
    assert _rel(ga['h7']['w'], ref['grads']['am']['h7']['w']) < 2e-3


def test_reference_named_shim_runs_a_training_step():
    """am_lm_train.py:64-75: feeds and fetches by the reference's names; han_wer against hand-computed edit distances."""
    from asr_dfcnn_transformer_amd.am_lm_model import CNNCTCModel
    from asr_dfcnn_transformer_amd.hparams import AmLmHparams
    from oracle import ctc
    hp = AmLmHparams().args
    hp.feature_dim, hp.feature_max_length, hp.num_blocks, hp.dropout_rate = F, T, 2, 0.0
    m = CNNCTCModel(hp, VP, VH, batch_size=B, widths=WIDTHS)
    assert (m.hidden_units, m.num_heads, m.position_max_length) == (128, 2, max(hp.position_max_length, T // 8))
    rng = np.random.default_rng(3)
    x = rng.standard_normal((B, T, F, 1)).astype(np.float32)
    tp = np.zeros((B, 8), dtype=np.int32); tp[0, :3] = [3, 4, 7]; tp[1, :2] = [5, 5]; tp[2, :4] = [1, 2, 2, 9]
    hz = np.zeros((B, 8), dtype=np.int32); hz[0, :2] = [30, 31]; hz[1, :1] = [12]; hz[2, :3] = [7, 8, 9]
    feed = {m.wav_input: x, m.wav_length: [8, 6, 7], m.target_py: tp, m.target_py_length: [3, 2, 4],
            m.target_hanzi: hz, m.target_hanzi_length: [2, 1, 3]}
    first = None
    for _ in range(4):
        lm_mean, err, wer, summary, _ = m.run([m.lm_mean_loss, m.label_err, m.han_wer, m.summary, m.train_op], feed)
        first = first if first is not None else lm_mean
    assert np.isfinite(lm_mean) and lm_mean < first and m.global_step == 4
    dec = m.run(m.decoded, feed)
    hyp = [[] for _ in range(B)]
    for (b, j), v in zip(dec.indices, dec.values):
        hyp[b].append(int(v))
    truth = [[30, 31], [12], [7, 8, 9]]
    want = np.mean([ctc.edit_distance_normalized(h, t) for h, t in zip(hyp, truth)])
    assert abs(m.run(m.han_wer, feed) - want) < 1e-6


def test_dense_label_mode_keeps_zeros_and_rejects_infeasible_alignments():
    """am_lm_model.py:72 hands ctc_loss_v2 DENSE labels + lengths: zeros inside the first target_py_length ids are labels
    (K1); where TF raises InvalidArgumentError (not enough frames for the label + the blanks between repeats) so do we."""
    eng, P, ops, x, tp, tl, wl = _build()
    eng.set_targets(wl, tp, tl)
    assert eng.am._host_labels == [[3, 0, 7], [5, 5], [1, 2, 2, 9]]
    eng.am.set_targets(wl, tp)                                   # the sparse path of acoustic_model*.py drops the zero
    assert eng.am._host_labels[0] == [3, 7]
    with pytest.raises(ValueError):
        eng.set_targets([8, 2, 7], tp, tl)                       # [5, 5] needs 3 frames (a blank between the repeats)
    with pytest.raises(ValueError):
        eng.set_targets([8, 6, 9], tp, tl)                       # T/8 = 8 frames exist
    tp_long = np.zeros((B, 8), dtype=np.int32); tp_long[:, :8] = 1 + np.arange(8)
    with pytest.raises(ValueError):
        eng.set_targets([8, 6, 7], tp_long, [8, 8, 8])           # 8 labels in 6 / 7 frames


def test_joint_training_loop_feeds_fetches_evaluates_and_checkpoints(tmp_path):
    """am_lm_train.train_model (lm_and_am/am_lm_train.py:27-116) on a small synthetic source: every batch of
    DataLoader.end2end_generator is fed as the six placeholders and [lm_mean_loss, label_err, han_wer, summary, train_op]
    fetched; a batch that lost a row is skipped (static batch); the dev pass runs without dropout and without updates; the
    per-epoch and final checkpoints are written; resume=True continues from final_model."""
    from asr_dfcnn_transformer_amd import am_lm_train
    from asr_dfcnn_transformer_amd.data_loader import DataLoader, SyntheticSource
    from asr_dfcnn_transformer_amd.hparams import AmLmHparams, AmDataHparams
    hp, dhp = AmLmHparams().args, AmDataHparams().args
    hp.am_batch_size, hp.epochs, hp.feature_dim, hp.feature_max_length = 2, 3, 16, 64
    hp.num_blocks, hp.dropout_rate, hp.am_lr, hp.position_max_length = 1, 0.1, 2e-3, 8
    V = 24
    train = SyntheticSource(8, seconds=0.5, label_len=3, vocab=V, seed=5, faults={3: 'long_label'})     # batch 1 loses a row
    dev = SyntheticSource(4, seconds=0.5, label_len=3, vocab=V, seed=9)
    loader = DataLoader(train, dhp, hp)
    assert [b[0].shape[0] for b in loader.end2end_generator()] == [2, 1, 2, 2]
    assert all(len(b) == 6 for b in loader.end2end_generator())

    class SmallVocabLoader(DataLoader):                    # the synthetic ids live in a 24-entry vocabulary
        def __init__(self, *a, **k):
            super().__init__(*a, **k)
            self.acoustic_vocab_size, self.language_vocab_size = V, 41
    kw = dict(widths=(4, 8, 16, 4, 128), seed=2)
    model, hist = am_lm_train.train_model(dhp, hp, train, dev, ckpt_dir=str(tmp_path), loader_cls=SmallVocabLoader, model_kw=kw,
                                          log_every=100)
    assert len(hist) == 3 * 3 and model.global_step == 9                   # 4 batches per epoch, one skipped
    assert all(np.isfinite(h[0]) for h in hist)
    assert np.mean([h[0] for h in hist[-3:]]) < np.mean([h[0] for h in hist[:3]])
    files = sorted(os.listdir(tmp_path))
    assert sum(f.startswith('model_') for f in files) == 3
    if 'final_model.pt' not in files:
        # final_model is only written when the dev label error rate drops below its best so far, starting at 1
        # (am_lm_train.py:56,110-112) -- nine steps of a toy model do not get there: take the last epoch's checkpoint
        import shutil
        shutil.copy(os.path.join(tmp_path, [f for f in files if f.startswith('model_2-')][0]), os.path.join(tmp_path, 'final_model.pt'))
    hp.epochs = 1
    m2, h2 = am_lm_train.train_model(dhp, hp, train, dev, ckpt_dir=str(tmp_path), loader_cls=SmallVocabLoader, model_kw=kw,
                                     resume=True, log_every=100)
    assert m2.global_step > 3 and len(h2) == 3                             # continued from a saved final_model, not from scratch
    m3, _ = am_lm_train.train_model(dhp, hp, train, None, ckpt_dir=str(tmp_path), loader_cls=SmallVocabLoader, model_kw=kw,
                                    log_every=100)
    assert m3.global_step == 3                                             # the reference's `latest = None`: no resume by default

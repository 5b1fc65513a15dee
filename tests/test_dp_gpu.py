"""Short batches (B' <= B rows, lm_and_am/data_loader.py:149-156 + the None batch dimension of
acoustic_model2.py:29-32) and data parallelism (equal shards + mean of gradients ==
the single-process global-batch step, reduce_mean acoustic_model2.py:83) on the GPU."""
import math
import os
import socket

import numpy as np
import pytest
import torch

from oracle import dfcnn

pytestmark = pytest.mark.gpu


def rel(got, want):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    return np.abs(got - want).max() / max(1e-6, np.abs(want).max())


def _case(model, widths, B, T, F, V, seed=11, L=2):
    g = dfcnn.graph(model, V, widths, feat=F)
    P = dfcnn.init_params(g, seed=3, perturb=True)
    P = {l: {k: v.astype(np.float32).astype(np.float64) for k, v in d.items()} for l, d in P.items()}
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((B, T, F, 1)).astype(np.float32)
    T8 = T // 8
    target = np.zeros((B, 64), dtype=np.int32)
    target[:, :L] = rng.integers(1, V - 1, (B, L))
    seq = [T8 if b % 2 == 0 else max(L + 1, T8 - 1) for b in range(B)]
    return g, P, x, seq, target


@pytest.mark.parametrize("model,widths", [('m2', (8, 16, 32, 64)), ('m1', (8, 16, 32, 64, 8, 32)), ('small', (8, 8, 8, 8))])
def test_short_batch_step_equals_oracle_step_on_the_surviving_rows(model, widths):
    """B' = 3 rows fed to an engine built for B = 4: the step is the oracle's B = 3 step (loss rows, mean over 3,
    decoded ids, every gradient), the padding row contributes nothing."""
    from asr_dfcnn_transformer_amd.engine import DFCNNEngine
    B, n, T, F, V = 4, 3, 32, 16, 12
    g, P, x, seq, target = _case(model, widths, n, T, F, V)
    ref = dfcnn.train_step_oracle(g, P, x.astype(np.float64), seq, target)
    eng = DFCNNEngine(model=model, vocab=V, B=B, T=T, F=F, widths=widths, seed=0)
    eng.load_params(P)
    logits = eng.forward(torch.tensor(x.reshape(n, T, F), device='cuda'))
    eng.set_targets(seq, target, n_valid=n)
    eng.loss_and_decode()
    eng.backward()
    torch.cuda.synchronize()
    assert np.abs(logits.cpu().numpy()[:, :n] - ref['logits']).max() < 1e-3
    loss = eng.loss.cpu().numpy()
    assert np.abs(loss[:n] - ref['loss'][:, 0]).max() < 1e-3 and loss[n] == 0.0
    assert eng.ctc_status.cpu().numpy().tolist() == [0, 0, 0, 2]
    assert not eng.ctc_grad[:, n].any().item()                    # padding row: zero CTC gradient
    mean_loss, label_err = eng.fetch_scalars()
    assert abs(mean_loss - ref['mean_loss']) < 1e-3
    assert abs(label_err - ref['label_err']) < 1e-5 or (math.isinf(label_err) and math.isinf(ref['label_err']))
    assert eng.decoded_lists() == ref['decoded']
    G = eng.grads_dict()
    for layer in P:
        for key in P[layer]:
            assert rel(G[layer][key], ref['grads'][layer][key]) < 1e-3, (layer, key)


@pytest.mark.parametrize("model,widths", [('m1', (8, 16, 32, 64, 8, 32)), ('m2', (8, 16, 32, 64))])
def test_two_half_batch_engines_equal_one_full_batch_engine(model, widths):
    """DP equivalence (acoustic_model2.py:83 reduce_mean): two B/2 engines on the two halves of a batch, gradients
    summed and scaled by 1/world, equal the B engine's gradient to 1e-6 relative -- in both weightings the code
    uses: per-rank mean x 1/world (bench.py) and global denominator + plain sum (CNNCTCModel.run)."""
    from asr_dfcnn_transformer_amd.engine import DFCNNEngine
    B, T, F, V = 4, 64, 24, 20
    g, P, x, seq, target = _case(model, widths, B, T, F, V, L=3)
    xd = torch.tensor(x.reshape(B, T, F), device='cuda')

    def grads(rows, denom=None):
        e = DFCNNEngine(model=model, vocab=V, B=len(rows), T=T, F=F, widths=widths, seed=0)
        e.load_params(P)
        e.forward(xd[rows].contiguous())
        e.set_targets([seq[r] for r in rows], target[rows], loss_denom=denom)
        e.loss_and_decode()
        e.backward()
        torch.cuda.synchronize()
        return e.grad.double().cpu().numpy(), e

    full, ef = grads([0, 1, 2, 3])
    scale = np.abs(full).max()
    a, _ = grads([0, 1]); b, _ = grads([2, 3])
    assert np.abs((a + b) * 0.5 - full).max() <= 1e-6 * scale
    a, _ = grads([0, 1], denom=4); b, _ = grads([2, 3], denom=4)
    assert np.abs((a + b) - full).max() <= 1e-6 * scale
    # unequal shards: 3 + 1 rows weighted by the global row count
    a, _ = grads([0, 1, 2], denom=4); b, _ = grads([3], denom=4)
    assert np.abs((a + b) - full).max() <= 1e-6 * scale
    # per-layer: no tensor hides behind the largest one
    ent = ef.entries
    a, _ = grads([0, 1]); b, _ = grads([2, 3])
    for (layer, key), (off, shape) in ent.items():
        n = int(np.prod(shape))
        f = full[off:off + n]
        assert np.abs((a[off:off + n] + b[off:off + n]) * 0.5 - f).max() <= 2e-6 * max(np.abs(f).max(), 1e-6), (layer, key)


def test_session_model_trains_on_surviving_rows():
    """CNNCTCModel.run fed 3 rows on a batch-4 model: fetch shapes follow B', and the update equals the one a
    batch-3 model makes on the same rows."""
    from asr_dfcnn_transformer_amd.acoustic_model import CNNCTCModel
    from asr_dfcnn_transformer_amd.hparams import AmLmHparams
    hp = AmLmHparams().args
    hp.feature_max_length, hp.feature_dim, hp.am_batch_size, hp.am_lr = 64, 16, 4, 1e-3
    rng = np.random.default_rng(0)
    x = rng.standard_normal((3, 64, 16, 1)).astype(np.float32)
    y = np.zeros((3, 64), dtype=np.int32)
    y[:, :3] = rng.integers(1, 10, (3, 3))
    thetas = []
    for bs in (4, 3):
        m = CNNCTCModel(hp, 12, 6345, widths=(8, 16, 32, 64), batch_size=bs)
        feed = {m.wav_input: x, m.logits_length: np.array([8, 7, 8]), m.target_py: y, m.target_length: np.array([3, 3, 3]),
                m.drop_rate: 0.5}
        loss, mean_loss, lr, summary, label_err, _ = m.run(
            [m.loss, m.mean_loss, m.current_learning, m.summary, m.label_err, m.train_op], feed_dict=feed)
        assert loss.shape == (3, 1) and abs(mean_loss - float(loss.mean())) < 1e-5
        dec = m.run(m.decoded[0], {m.wav_input: x, m.logits_length: np.array([8, 7, 8])})
        assert dec.dense_shape[0] == 3
        thetas.append((m.engine.params_dict(), m.engine.grads_dict()))
    for layer in thetas[0][1]:
        for key in thetas[0][1][layer]:
            assert rel(thetas[0][1][layer][key], thetas[1][1][layer][key]) < 1e-5, (layer, key)


def test_data_generation_drop_rules_and_layouts():
    """DataLoader.data_generation / __getitem__ / am_generator (lm_and_am/data_loader.py:105-162,246-280): one sample per
    rejection rule -> those rows are deleted; input_length = min(200, T//8 + 1); wav [B', 1600, 200, 1]; labels [B', 64]
    zero-padded; features equal compute_fbank_from_api of the surviving utterances."""
    from asr_dfcnn_transformer_amd.data_loader import DataLoader, SyntheticSource, ctc_input_length
    from asr_dfcnn_transformer_amd.hparams import AmLmHparams, AmDataHparams
    from asr_dfcnn_transformer_amd.wav_util import compute_fbank_from_api
    hp = AmLmHparams().args
    hp.am_batch_size = 8
    faults = {1: 'long_audio', 2: 'long_label', 4: 'label_ge_input', 5: 'unknown_token', 6: 'long_hanzi'}
    src = SyntheticSource(16, seconds=2.0, label_len=20, faults=faults)
    dl = DataLoader(src, AmDataHparams().args, hp)
    assert len(dl) == 2
    wav, in_len, py, py_len, han, han_len = dl[0]
    keep = [0, 3, 7]
    assert dl.last_kept == keep
    assert tuple(wav.shape) == (3, 1600, 200, 1) and wav.dtype == torch.float32
    nf = 1 + math.ceil((32000 - 400) / 160)
    assert in_len.tolist() == [min(200, nf // 8 + 1)] * 3 == [ctc_input_length(nf)] * 3
    assert py.shape == (3, 64) and py.dtype == np.int32 and han.shape == (3, 64)
    assert py_len.tolist() == [20, 20, 20] and han_len.tolist() == [20, 20, 20]       # word_length := len_label (:145)
    for r, k in enumerate(keep):
        assert py[r, :20].tolist() == src.labels[k].tolist() and not py[r, 20:].any()
        sig, sr = src.read_audio(src.path_lst[k])
        f = compute_fbank_from_api(sig, sr)
        got = wav[r, :, :, 0].cpu().numpy()
        assert np.abs(got[:f.shape[0]] - f).max() < 1e-4 and not got[f.shape[0]:].any()
    # second batch: nothing planted -> all 8 rows; the generator yields both batches
    items = list(dl.am_generator())
    assert [it[0].shape[0] for it in items] == [3, 8]
    # every row rejected -> arrays with a 0-row batch axis, like np.delete of all rows
    src0 = SyntheticSource(2, seconds=1.0, label_len=20, faults={0: 'long_label', 1: 'unknown_token'})
    hp.am_batch_size = 2
    w0, l0, p0, _, _, _ = DataLoader(src0, AmDataHparams().args, hp)[0]
    assert w0.shape[0] == 0 and len(l0) == 0 and p0.shape == (0, 64)


def test_composite_checkpoints_round_trip(tmp_path):
    """save_checkpoint / load_checkpoint of the joint AM+LM engine (two parts) and of the speech Transformer shim
    (pre-net + encoder-decoder): every part's variables, Adam slots and global_step come back; a checkpoint of a
    different composite is refused."""
    from asr_dfcnn_transformer_amd import train as tr
    from asr_dfcnn_transformer_amd.joint_engine import AMLMEngine
    kw = dict(v_pinyin=12, v_hanzi=15, B=2, T=64, F=16, widths=(4, 8, 16, 8, 128), heads=2, blocks=1, pos_max=8)
    a = AMLMEngine(seed=1, **kw)
    rng = np.random.default_rng(0)
    x = torch.tensor(rng.standard_normal((2, 64, 16)).astype(np.float32), device='cuda')
    tp = np.zeros((2, 64), dtype=np.int32); tp[:, :2] = rng.integers(1, 10, (2, 2))
    for _ in range(2):
        a.forward(x); a.set_targets([8, 8], tp, [2, 2]); a.loss_and_decode(); a.backward(); a.apply_adam()
    p = str(tmp_path / 'amlm.pt')
    tr.save_checkpoint(a, p)
    b = AMLMEngine(seed=7, **kw)
    tr.load_checkpoint(b, p)
    for part in ('am', 'lm'):
        ea, eb = getattr(a, part), getattr(b, part)
        assert torch.equal(ea.theta, eb.theta) and torch.equal(ea.adam_m, eb.adam_m) and torch.equal(ea.adam_v, eb.adam_v)
        assert eb.global_step == 2
    from asr_dfcnn_transformer_amd.engine import DFCNNEngine
    with pytest.raises(ValueError):
        tr.load_checkpoint(DFCNNEngine(model='m2', vocab=12, B=1, T=16, F=16, widths=(8, 8, 8, 8)), p)


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _train_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), ASR_DIST_BACKEND='gloo')
    import torch as th
    from asr_dfcnn_transformer_amd import train as tr
    from asr_dfcnn_transformer_amd.acoustic_model import CNNCTCModel
    from asr_dfcnn_transformer_amd.data_loader import SyntheticSource
    from asr_dfcnn_transformer_amd.hparams import AmLmHparams, AmDataHparams

    class Small(CNNCTCModel):
        def __init__(self, args, av, lv):
            super().__init__(args, av, lv, widths=(8, 16, 16, 32))

    hp = AmLmHparams().args
    hp.am_batch_size, hp.epochs = 2, 1
    # 7 batches of 2 on 2 ranks -> 3 steps each, batch 6 left over; utterance 2 (rank 1's first batch) loses a row,
    # utterances 8 and 9 (rank 0's third batch) are both rejected: that rank steps with ZERO rows
    src = SyntheticSource(15, seconds=1.0, label_len=4, faults={2: 'long_label', 8: 'unknown_token', 9: 'long_label'})
    model, hist = tr.train_acoustic_model(AmDataHparams().args, hp, src, log_every=1000, model_cls=Small)
    th.cuda.synchronize()
    q.put((rank, len(hist), model.global_step, model.engine.theta.cpu().numpy()))
    th.distributed.barrier()
    th.distributed.destroy_process_group()


def test_two_rank_training_with_dropped_rows_keeps_ranks_in_step():
    """train_acoustic_model on 2 ranks (gloo, both on this GPU) with an odd batch count, a batch that lost one row and a
    batch that lost all rows: no rank skips a collective (the run finishes), both take batch_nums // world steps and end
    with bitwise identical parameters."""
    import torch.multiprocessing as mp
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_train_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted([q.get(timeout=600) for _ in range(world)], key=lambda t: t[0])
    for p in ps:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert [r[1] for r in res] == [3, 3] and [r[2] for r in res] == [3, 3]
    assert np.array_equal(res[0][3], res[1][3])
    assert np.isfinite(res[0][3]).all()

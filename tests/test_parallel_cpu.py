"""world_size-2 gloo test of the data-parallel gradient exchange (the N>1 path of bench.py
and CNNCTCModel.run): bucketed async sum-all-reduce of the flat gradient + 1/world scale."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from asr_dfcnn_transformer_amd.parallel import init_from_env, BucketedAllReduce
    r, w, _ = init_from_env(backend='gloo')
    assert (r, w) == (rank, world)
    n = 1000
    flat = torch.arange(n, dtype=torch.float32) * (rank + 1)
    red = BucketedAllReduce(flat, [(100, 400), (0, 100), (400, n)])
    red.launch(0)                      # "dense head" bucket first, while backward would continue
    flat_mid = flat.clone()
    red.launch(1)
    red.launch(2)
    red.wait()
    want = torch.arange(n, dtype=torch.float32) * sum(range(1, world + 1))
    ok = torch.equal(flat, want) and abs(red.grad_scale - 1.0 / world) < 1e-12
    # empty bucket and world bookkeeping
    red2 = BucketedAllReduce(flat, [(5, 5)])
    red2.launch(0)
    red2.wait()
    q.put((rank, bool(ok), float(flat_mid[0])))
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_allreduce_gloo_world2():
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res)


def test_single_process_is_noop():
    from asr_dfcnn_transformer_amd.parallel import BucketedAllReduce
    flat = torch.ones(10)
    red = BucketedAllReduce(flat, [(0, 10)])
    red.launch(0)
    red.wait()
    assert red.world == 1 and red.grad_scale == 1.0 and torch.equal(flat, torch.ones(10))


# ---------------------------------------------------------------------------------------------------------------------
# The engine's real flat-gradient layout through the three buckets, and the training loop's collective discipline.
def _oracle_flat_grad(rows, denom):
    """Oracle gradient of the mean CTC loss over `rows` of a fixed tiny batch, packed in the engine's flat layout
    (engine.param_layout: [gammas][dense head][conv/SE]) and scaled so that the plain SUM over shards is the
    global-batch mean gradient."""
    import numpy as np
    from oracle import dfcnn
    from asr_dfcnn_transformer_amd.engine import graph, param_layout
    model, widths, B, T, F, V = 'm2', (4, 4, 8, 8), 4, 16, 8, 9
    g = dfcnn.graph(model, V, widths, feat=F)
    assert [tuple(o) for o in g] == [tuple(o) for o in graph(model, V, widths, F)]
    P = dfcnn.init_params(g, seed=3, perturb=True)
    rng = np.random.default_rng(5)
    x = rng.standard_normal((B, T, F, 1))
    target = np.zeros((B, 64), dtype=np.int32)
    target[:, 0] = rng.integers(1, V - 1, B)
    ref = dfcnn.train_step_oracle(g, P, x[rows], [T // 8] * len(rows), target[rows])
    ent, n_gamma, dense_end, total = param_layout(g)
    flat = np.zeros(total, dtype=np.float64)
    for (layer, key), (off, shape) in ent.items():
        v = ref['grads'][layer][key]
        flat[off:off + v.size] = v.ravel() * (len(rows) / float(denom))      # oracle grads are means over its own rows
    return flat, (n_gamma, dense_end, total)


def _layout_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import numpy as np
    from asr_dfcnn_transformer_amd.parallel import init_from_env, BucketedAllReduce
    init_from_env(backend='gloo')
    rows = [[0, 1, 2], [3]][rank]                       # unequal shards: 3 + 1 surviving rows
    red0 = BucketedAllReduce(torch.zeros(4), [(0, 4)])
    denom = red0.sum_count(len(rows))
    flat_np, (n_gamma, dense_end, total) = _oracle_flat_grad(rows, denom)
    flat = torch.from_numpy(flat_np.copy())
    red = BucketedAllReduce(flat, [(n_gamma, dense_end), (0, n_gamma), (dense_end, total)])
    red.launch(0); red.launch(1); red.launch(2); red.wait()
    full, _ = _oracle_flat_grad([0, 1, 2, 3], 4)
    err = float(np.abs(flat.numpy() - full).max() / np.abs(full).max())
    q.put((rank, denom, err, int((full != 0).sum())))
    dist.barrier()
    dist.destroy_process_group()


def test_engine_gradient_layout_through_the_three_buckets_gloo_world2():
    """Two ranks hold the oracle gradients of a 3-row and a 1-row shard in the engine's flat layout; the bucketed sum
    over gloo, with each rank's contribution weighted by rows / global rows (BucketedAllReduce.sum_count), is the
    full-batch mean gradient in every tensor (DP semantics of SURVEY 7 / acoustic_model2.py:83)."""
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_layout_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, denom, err, nz in res:
        assert denom == 4 and err < 1e-12 and nz > 100, (rank, denom, err, nz)


class _FakeLoader:
    """Stands in for DataLoader (device fbank) on the CPU: 7 batches of up to 2 rows; batch 1 lost a row, batch 4 lost
    both (data_loader.py:149-156)."""
    acoustic_vocab_size, language_vocab_size = 12, 20

    def __init__(self, source, data_args, hp):
        self.rows = [2, 1, 2, 2, 0, 2, 2]

    def __len__(self):
        return len(self.rows)

    def __getitem__(self, i):
        import numpy as np
        n = self.rows[i]
        x = np.full((n, 4), float(i + 1), dtype=np.float32)
        return x, np.full(n, 3), np.zeros((n, 64), dtype=np.int32), np.full(n, 1), np.zeros((n, 64), dtype=np.int32), np.full(n, 1)


class _FakeModel:
    """Stands in for CNNCTCModel (HIP kernels): same feed / fetch handles, same collective sequence per step as
    acoustic_model.CNNCTCModel.run -- sum_count, then the three bucket all-reduces -- and a plain SGD update."""
    wav_input, logits_length, target_py, target_length, drop_rate = 'wav_input', 'logits_length', 'target_py', 'target_length', 'dr'
    loss, mean_loss, current_learning, summary, label_err, train_op = 'loss', 'mean_loss', 'lr', 'summary', 'err', 'train_op'

    def __init__(self, hp, av, lv):
        from asr_dfcnn_transformer_amd.parallel import BucketedAllReduce
        self.theta = torch.zeros(12, dtype=torch.float64)
        self.grad = torch.zeros(12, dtype=torch.float64)
        self.reducer = BucketedAllReduce(self.grad, [(2, 8), (0, 2), (8, 12)])
        self.global_step = 0
        self.calls = 0

    def run(self, fetches, feed_dict):
        x = torch.as_tensor(feed_dict[self.wav_input], dtype=torch.float64)
        n = x.shape[0]
        denom = self.reducer.sum_count(n)
        self.grad.zero_()
        if n:
            self.grad += x.sum() / max(denom, 1)
        self.reducer.launch(0); self.reducer.launch(1); self.reducer.launch(2); self.reducer.wait()
        if denom > 0:
            self.theta -= 0.1 * self.grad
            self.global_step += 1
        self.calls += 1
        return [None, float(n), 1e-3, None, 0.0, None]


def _loop_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), ASR_DIST_BACKEND='gloo')
    from asr_dfcnn_transformer_amd import train as tr
    from asr_dfcnn_transformer_amd.hparams import AmLmHparams, AmDataHparams
    hp = AmLmHparams().args
    hp.epochs = 2
    model, hist = tr.train_acoustic_model(AmDataHparams().args, hp, None, log_every=1000, model_cls=_FakeModel,
                                          loader_cls=_FakeLoader)
    q.put((rank, model.calls, model.global_step, model.theta.tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_training_loop_keeps_ranks_in_step_with_odd_batch_count_and_dropped_rows():
    """train.train_acoustic_model on 2 gloo ranks over 7 batches (odd), one of which lost a row and one all rows: the old
    loop gave rank 0 four steps and rank 1 three, and skipped short batches rank-locally -- mismatched collectives, a
    hang.  Now both ranks take 7 // 2 = 3 steps per epoch, enter every collective, and end with identical parameters
    equal to the global-row-weighted mean updates."""
    from asr_dfcnn_transformer_amd.train import rank_batches
    assert rank_batches(7, 2, 0) == [0, 2, 4] and rank_batches(7, 2, 1) == [1, 3, 5]
    assert rank_batches(4, 8, 3) == [] and rank_batches(4, 8, 7) == []         # the advisor's example: nobody steps
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_loop_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda t: t[0])
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[1] for r in res] == [6, 6] and [r[2] for r in res] == [6, 6]
    assert res[0][3] == res[1][3]
    # expected: per step the gradient is (sum over both ranks' rows of 4 * batch_value) / global rows
    rows = [2, 1, 2, 2, 0, 2, 2]
    theta = 0.0
    for _ in range(2):
        for s in range(3):
            b0, b1 = 2 * s, 2 * s + 1
            tot = rows[b0] + rows[b1]
            theta -= 0.1 * (4 * (b0 + 1) * rows[b0] + 4 * (b1 + 1) * rows[b1]) / tot
    assert abs(res[0][3][0] - theta) < 1e-12


def _bad_step_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from asr_dfcnn_transformer_amd.parallel import init_from_env, BucketedAllReduce
    init_from_env(backend='gloo')
    red = BucketedAllReduce(torch.zeros(8), [(0, 8)])
    first = red.sum_count(3 + rank)                       # a healthy step: the global row count
    try:
        red.sum_count(4, ok=(rank != 1))                  # rank 1's batch failed its host-side target check
        raised = False
    except RuntimeError:
        raised = True
    after = red.sum_count(1)                              # the group is still usable: nobody was left inside a collective
    q.put((rank, first, raised, after))
    dist.barrier()
    dist.destroy_process_group()


def test_a_rank_that_cannot_take_the_step_stops_every_rank_before_the_gradient_allreduce():
    """CNNCTCModel.run validates the targets on the host BEFORE the step's first collective and sends the verdict along with the
    row count: a bad batch on one rank raises on every rank at that point (no hang until the RCCL timeout)."""
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_bad_step_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [(0, 7, True, 2), (1, 7, True, 2)]

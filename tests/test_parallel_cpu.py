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

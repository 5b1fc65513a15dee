"""First contact with RCCL (north_star: "data-parallel with RCCL all-reduce of gradients over xGMI") on ONE GPU:
torch.distributed backend "nccl" (= RCCL on ROCm) in a group of one rank, the plain-DFCNN step (lm_and_am/model/acoustic_model.py:37-62,
driver lm_and_am/train.py:54-74) with the engine's real three gradient buckets, the dense head launched from inside the backward pass.
A one-rank sum all-reduce must return its input bit for bit, so the gradients and the updated parameters must equal the step
without any collective -- which checks the parts a one-GPU box can check: librccl loads, the communicator initialises, and
c10d's collective stream is ordered correctly against the engine's main and side streams (an all-reduce that ran before its
bucket was final, or an optimiser that ran before the all-reduce, would show up as different bits)."""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    import torch as th
    import torch.distributed as dist
    from asr_dfcnn_transformer_amd.engine import DFCNNEngine
    from asr_dfcnn_transformer_amd.parallel import BucketedAllReduce
    th.cuda.set_device(0)
    dist.init_process_group(backend='nccl', rank=0, world_size=1)
    assert dist.get_backend() == 'nccl'
    B, T, F, V = 4, 320, 200, 1536                      # the real widths (32..256 channels, 6400 -> 128 -> 1536 head), a short batch
    rng = np.random.default_rng(5)
    x = th.tensor(rng.standard_normal((B, T, F)).astype(np.float32), device='cuda')
    target = np.zeros((B, 64), dtype=np.int32)
    target[:, :6] = rng.integers(1, V - 1, (B, 6))
    seq = [T // 8] * B

    def run(collective, steps=2):
        eng = DFCNNEngine(model='m1', vocab=V, B=B, T=T, F=F, seed=0)
        red = BucketedAllReduce(eng.grad, [(eng.n_gamma, eng.dense_end), (0, eng.n_gamma), (eng.dense_end, eng.grad.numel())],
                                always_collective=collective)
        launched = []
        grads = []
        for _ in range(steps):
            eng.forward(x)
            eng.set_targets(seq, target)
            eng.loss_and_decode(defer_decode_join=True)

            def early():
                launched.append(len(red.pending))
                red.launch(0)
            eng.backward(on_dense_grads_ready=early)
            red.launch(1); red.launch(2)
            n_pending = len(red.pending)
            red.wait()
            grads.append(eng.grad.clone())
            eng.apply_adam(red.grad_scale)
        th.cuda.synchronize()
        return grads, eng.theta.clone(), n_pending

    g_plain, th_plain, n0 = run(False)
    g_rccl, th_rccl, n1 = run(True)
    g_again, th_again, _ = run(True)
    ok = {'pending_plain': n0, 'pending_rccl': n1,
          'grads_equal': all(th.equal(a, b) for a, b in zip(g_plain, g_rccl)),
          'theta_equal': bool(th.equal(th_plain, th_rccl)),
          'twice_equal': all(th.equal(a, b) for a, b in zip(g_rccl, g_again)) and bool(th.equal(th_rccl, th_again)),
          'finite': bool(th.isfinite(th_rccl).all().item()), 'moved': bool((g_rccl[0] != 0).any().item()),
          'nccl_version': '.'.join(str(v) for v in th.cuda.nccl.version())}
    # a one-rank RCCL sum over a buffer with a known value: the collective really touches device memory on its own stream
    t = th.arange(1 << 20, dtype=th.float32, device='cuda')
    w = dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=True)
    w.wait()
    ok['arange_kept'] = bool(th.equal(t, th.arange(1 << 20, dtype=th.float32, device='cuda')))
    dist.barrier()
    dist.destroy_process_group()
    q.put(ok)


def test_the_dfcnn_step_through_a_one_rank_rccl_group_equals_the_step_without_a_collective():
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    p = ctx.Process(target=_worker, args=(_free_port(), q))
    p.start()
    res = q.get(timeout=600)
    p.join(timeout=120)
    assert p.exitcode == 0
    assert res['pending_plain'] == 0 and res['pending_rccl'] == 3, res        # three real collectives were in flight per step
    assert res['grads_equal'] and res['theta_equal'] and res['twice_equal'], res
    assert res['finite'] and res['moved'] and res['arange_kept'], res


def test_init_from_env_refuses_more_nccl_ranks_than_gpus(monkeypatch):
    """torchrun with more ranks than GPUs: one clear line (SystemExit) instead of `invalid device ordinal`."""
    from asr_dfcnn_transformer_amd import parallel
    n = torch.cuda.device_count()
    monkeypatch.setenv('WORLD_SIZE', str(n + 1)); monkeypatch.setenv('RANK', str(n)); monkeypatch.setenv('LOCAL_RANK', str(n))
    with pytest.raises(SystemExit) as e:
        parallel.init_from_env(backend='nccl')
    assert 'has no GPU of its own' in str(e.value)

"""Data parallelism for the acoustic model: one process per GPU, the flat fp32 gradient
buffer is summed over ranks with RCCL (torch.distributed backend "nccl" on ROCm) in two
buckets -- the dense head first (its gradients are ready before the conv-stack backward
starts, so that all-reduce overlaps the rest of backward), the remainder at the end --
and Adam divides by the world size.  The reference has no distributed code (SURVEY.md
section 5); utterances are independent (frozen BN, per-sample SE and CTC), so equal
shards + mean of gradients reproduce the single-GPU global-batch step (section 8e).

Works on any torch.distributed backend, which is how the CPU tests cover it (gloo)."""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialises the default process group from RANK / WORLD_SIZE / MASTER_* when the
    launcher set them.  Returns (rank, world, local_rank)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend is None:
            backend = os.environ.get('ASR_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
        if backend == 'nccl':
            # RCCL needs one GPU per rank: say so in one line instead of `invalid device ordinal` under a launcher traceback
            ndev = torch.cuda.device_count()
            if local >= ndev:
                raise SystemExit('asr_dfcnn_transformer_amd.parallel: rank %d (LOCAL_RANK %d) of %d has no GPU of its own: %d GPU(s) '
                                 'visible (backend nccl = RCCL needs one per rank; ASR_DIST_BACKEND=gloo rehearses more ranks than '
                                 'GPUs)' % (rank, local, world, ndev))
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    if torch.cuda.is_available() and torch.cuda.device_count() > 0:
        local = local % torch.cuda.device_count()     # rehearsals with more ranks than GPUs (gloo) share devices
    return rank, world, local


def all_agree(ok):
    """True only when ``ok`` holds on every rank (one blocking MIN all-reduce of a flag; a no-op in a single process).
    Collective: every rank must call it at the same point."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return bool(ok)
    dev = 'cuda' if dist.get_backend() == 'nccl' else 'cpu'
    t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(t.item())


class BucketedAllReduce:
    """Sum-all-reduce of a flat gradient tensor in contiguous buckets ``ranges`` =
    [(lo, hi), ...], each launched asynchronously as soon as the caller says its range is
    final (``launch(i)``); ``wait()`` joins them before the optimiser reads the buffer."""

    def __init__(self, flat_grad, ranges, group=None, always_collective=False):
        """``always_collective``: enter the collective even in a group of ONE rank (the default skips it there).  With backend
        nccl that is a real RCCL all-reduce on c10d's collective stream -- library load, communicator, stream ordering
        against the engine's main / side streams -- whose result must be the input bit for bit (tests/test_rccl_gpu.py)."""
        self.always = bool(always_collective) and dist.is_initialized()
        self.flat = flat_grad
        self.ranges = [(int(lo), int(hi)) for lo, hi in ranges]
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.pending = []

    @property
    def num_buckets(self):
        return len(self.ranges)

    def launch(self, i):
        lo, hi = self.ranges[i]
        if (self.world == 1 and not self.always) or hi <= lo:
            return
        self.pending.append(dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def wait(self):
        for w in self.pending:
            w.wait()
        self.pending = []

    def sum_count(self, n, ok=True):
        """Sum of a per-rank integer over the ranks (rows that survived the loader: the denominator of the global
        mean).  One tiny blocking all-reduce; every rank must call it the same number of times.  ``ok=False`` reports that
        this rank cannot take the step (its batch failed a host-side check): the same collective carries the flag, and EVERY
        rank then gets RuntimeError here -- before any gradient all-reduce is entered -- instead of the healthy ranks
        waiting for the failed one until the RCCL timeout."""
        if self.world == 1:
            if not ok:
                raise RuntimeError('this step cannot run (host-side check failed)')
            return int(n)
        dev = self.flat.device if dist.get_backend(self.group) == 'nccl' else 'cpu'
        t = torch.tensor([int(n), 0 if ok else 1], dtype=torch.int64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        tot, bad = (int(v) for v in t.tolist())
        if bad:
            raise RuntimeError('%d rank(s) cannot take this step (host-side target check failed); no rank entered the gradient '
                               'all-reduce' % bad)
        return tot

    @property
    def grad_scale(self):
        return 1.0 / self.world

"""bench.py as an N-rank job (the driver's `python -m torch.distributed.run ... bench.py --gpus N`), rehearsed on ONE GPU
with gloo ranks: the job must END -- every rank enters the same collectives whatever its own timing table says.  Round 5 found a run
that never did: the dominant kernel symbol was a per-rank decision, and ranks whose symbol runs alone in the two-stream step skipped
the roofline pass's extra steps (which carry gradient collectives) that the other ranks took.  The test hook makes the ranks disagree
on purpose; rank 0's choice is broadcast before anything depends on it."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize('disagree', [False, True])
def test_two_gloo_ranks_finish_and_report_one_line(disagree):
    env = dict(os.environ, ASR_DIST_BACKEND='gloo', ASR_BENCH_TRACEBACK_AFTER='300')
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    if disagree:
        env['ASR_BENCH_TEST_RANKS_DISAGREE'] = '1'
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1', '--batch', '2', '--tpad', '1000',
           '--no-cpu-baseline']
    p = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=360)
    err = p.stderr.decode(errors='replace')
    assert p.returncode == 0, err[-3000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith('{')]
    assert len(lines) == 1, (lines, err[-2000:])
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['config']['ranks'] == 2 and out['config']['dist_backend'] == 'gloo'
    assert out['config']['global_batch'] == 4 and out['config']['gradient_collectives_per_step'] == 3
    assert out['value'] > 0 and out['roofline']['kernel'] and out['roofline']['achieved'] > 0

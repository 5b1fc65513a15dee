"""Host logic of bench.py that runs before any GPU call: the priming child (argument filtering, environment, failure
tolerance), the profiler check that forbids it, and the device count from sysfs."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import bench


def test_under_profiler_reads_the_environment(monkeypatch):
    for k in list(os.environ):
        if k.startswith(('ROCPROF', 'ROCP_')):
            monkeypatch.delenv(k)
    monkeypatch.setenv('LD_PRELOAD', '')
    assert not bench.under_profiler()
    monkeypatch.setenv('LD_PRELOAD', '/opt/rocm/lib/librocprofiler-sdk-tool.so')
    assert bench.under_profiler()
    monkeypatch.setenv('LD_PRELOAD', '')
    monkeypatch.setenv('ROCPROF_OUTPUT_PATH', '/tmp/x')
    assert bench.under_profiler()


class _Done:
    def __init__(self, rc, out):
        self.returncode, self.stdout = rc, out


def test_prime_child_gets_the_workload_but_not_the_timing_flags(monkeypatch):
    seen = {}

    def fake_run(cmd, env=None, stdout=None, stderr=None, timeout=None):
        seen['cmd'], seen['env'] = cmd, env
        return _Done(0, (json.dumps({'ms_per_step': 6.7}) + '\n').encode())

    monkeypatch.setattr(subprocess, 'run', fake_run)
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '8', '--steps', '20', '--warmup=5', '--workload', 'se_dfcnn', '--tpad', '1000',
                                      '--kernel-table', '--prime-steps', '40'])
    monkeypatch.setenv('RANK', '3'); monkeypatch.setenv('LOCAL_RANK', '3'); monkeypatch.setenv('WORLD_SIZE', '8')
    monkeypatch.setenv('MASTER_ADDR', '127.0.0.1'); monkeypatch.setenv('MASTER_PORT', '29500')
    assert bench.prime_gpu(40) == 6.7
    cmd, env = seen['cmd'], seen['env']
    tail = cmd[2:]
    assert tail[:4] == ['--workload', 'se_dfcnn', '--tpad', '1000']                 # the workload travels
    assert tail[4:] == ['--gpus', '1', '--steps', '40', '--warmup', '3', '--prime-steps', '0', '--no-cpu-baseline']
    assert '--kernel-table' not in tail and '--warmup=5' not in tail
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):    # the child is a single process on the rank's GPU
        assert k not in env
    assert env['ASR_BENCH_DEVICE'] == '3'


def test_a_failing_prime_child_is_ignored(monkeypatch):
    monkeypatch.setattr(sys, 'argv', ['bench.py'])
    monkeypatch.setattr(subprocess, 'run', lambda *a, **k: _Done(1, b''))
    assert bench.prime_gpu(40) is None

    def boom(*a, **k):
        raise subprocess.TimeoutExpired('bench.py', 180)

    monkeypatch.setattr(subprocess, 'run', boom)
    assert bench.prime_gpu(40) is None


def test_visible_gpu_count_respects_the_visible_devices_lists(monkeypatch):
    n = bench.visible_gpu_count()
    if n is None or n == 0:
        pytest.skip('no KFD topology in this container')
    monkeypatch.setenv('HIP_VISIBLE_DEVICES', '0')
    assert bench.visible_gpu_count() == 1

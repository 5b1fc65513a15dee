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
        return _DoneErr(0, (json.dumps({'ms_per_step': 6.7}) + '\n').encode())

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


class _DoneErr(_Done):
    def __init__(self, rc, out, err=b''):
        super().__init__(rc, out)
        self.stderr = err


def test_a_cleanly_failing_prime_child_is_ignored_but_recorded(monkeypatch):
    monkeypatch.setattr(sys, 'argv', ['bench.py'])
    monkeypatch.setattr(subprocess, 'run', lambda *a, **k: _DoneErr(1, b'', b'Traceback ...\nValueError: bad flag\n'))
    assert bench.prime_gpu(40) is None
    assert bench.PRIME['rc'] == 1 and 'ValueError: bad flag' in bench.PRIME['error']
    bench.PRIME['steps'] = 40
    note = bench.prime_note()
    assert note['prime_child_rc'] == 1 and 'bad flag' in note['prime_child_error']
    bench.PRIME['steps'] = 0
    assert bench.prime_note() == {'primed_by_child_process_steps': 0}


def test_a_prime_child_that_hung_or_was_killed_stops_the_measurement(monkeypatch):
    monkeypatch.setattr(sys, 'argv', ['bench.py'])

    def boom(*a, **k):
        raise subprocess.TimeoutExpired('bench.py', 180, stderr=b'HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION')

    monkeypatch.setattr(subprocess, 'run', boom)
    with pytest.raises(bench.PrimeChildDied) as e:
        bench.prime_gpu(40)
    assert 'APERTURE' in str(e.value) and bench.PRIME['rc'] == 'timeout'
    monkeypatch.setattr(subprocess, 'run', lambda *a, **k: _DoneErr(-6, b'', b'Memory access fault by GPU node-1'))
    with pytest.raises(bench.PrimeChildDied) as e:
        bench.prime_gpu(40)
    assert 'signal 6' in str(e.value) and 'Memory access fault' in str(e.value)
    # main(): exit code 3, nothing measured
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--prime-steps', '5'])
    monkeypatch.setattr(bench, 'under_profiler', lambda: False)
    monkeypatch.delenv('WORLD_SIZE', raising=False)
    assert bench.main() == 3


def test_no_prime_by_default_and_never_with_several_ranks(monkeypatch):
    calls = []
    monkeypatch.setattr(bench, 'prime_gpu', lambda n: calls.append(n) or 1.0)
    monkeypatch.setattr(bench, 'under_profiler', lambda: False)

    class Stop(Exception):
        pass

    def stop(args):
        raise Stop()

    monkeypatch.setattr(bench, 'run_lm', stop)
    monkeypatch.delenv('WORLD_SIZE', raising=False)
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--workload', 'lm'])
    with pytest.raises(Stop):
        bench.main()
    assert calls == []
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--workload', 'lm', '--prime-steps', '7', '--gpus', '2'])
    monkeypatch.setenv('WORLD_SIZE', '2')
    with pytest.raises(Stop):
        bench.main()
    assert calls == []
    monkeypatch.delenv('WORLD_SIZE')
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--workload', 'lm', '--prime-steps', '7'])
    with pytest.raises(Stop):
        bench.main()
    assert calls == [7]


def test_visible_gpu_count_respects_the_visible_devices_lists(monkeypatch):
    n = bench.visible_gpu_count()
    if n is None or n == 0:
        pytest.skip('no KFD topology in this container')
    monkeypatch.setenv('HIP_VISIBLE_DEVICES', '0')
    assert bench.visible_gpu_count() == 1


_HANG = r'''
import sys, time
sys.path.insert(0, %r)
import torch.distributed as dist
import bench
dist.init_process_group(backend='gloo', init_method='tcp://127.0.0.1:%%d' %% int(sys.argv[1]), rank=0, world_size=1)
if sys.argv[2] == 'hang':
    real = dist.destroy_process_group
    dist.destroy_process_group = lambda: time.sleep(3600)
bench.shutdown_process_group(seconds=int(sys.argv[3]))
print('left through the normal exit')
'''


def _free_port():
    import socket
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close()
    return p


def test_a_teardown_that_hangs_is_a_failed_run_not_a_green_one():
    """ADVICE r5 (medium) / VERDICT r5 weak 8: a rank stuck in destroy_process_group() leaves with a NON-zero code and its stacks."""
    r = subprocess.run([sys.executable, '-c', _HANG % ROOT, str(_free_port()), 'hang', '2'], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       timeout=120)
    assert r.returncode == 4, (r.returncode, r.stderr.decode()[-800:])
    err = r.stderr.decode()
    assert 'did not return within 2 s' in err and 'shutdown_process_group' in err      # the message and the dumped stack
    assert b'left through the normal exit' not in r.stdout


def test_a_teardown_that_returns_leaves_through_the_normal_exit():
    r = subprocess.run([sys.executable, '-c', _HANG % ROOT, str(_free_port()), 'ok', '30'], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       timeout=120)
    assert r.returncode == 0, r.stderr.decode()[-800:]
    assert b'left through the normal exit' in r.stdout


def test_stream_calibration_keeps_two_streams_unless_one_wins_clearly():
    """bench.py --streams auto: the records of round 6 (profiles/r06z_bench_*: 6.12 vs 6.44, 4.18 vs 4.33 ms) keep two streams, a sample that
    flatters one stream by 1-3 % (4.39 vs 4.34: the mis-pick of the first version) still keeps two, the bimodal box of round 5 (7.2 vs 6.7) does not."""
    assert bench.keep_two_streams(6.115, 6.44) and bench.keep_two_streams(4.181, 4.325)
    assert bench.keep_two_streams(4.387, 4.335) and bench.keep_two_streams(6.911, 6.829)
    assert not bench.keep_two_streams(7.2, 6.7) and not bench.keep_two_streams(7.3, 6.75)

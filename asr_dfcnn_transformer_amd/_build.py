"""Builds libasrhip.so (gfx950) in-tree with hipcc.  No torch involved."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libasrhip.so')
SOURCES = ['runtime.hip', 'tap_gemm.hip', 'tap_wgrad.hip', 'cells.hip', 'head.hip', 'fbank.hip',
           'transformer.hip', 'attention.hip', 'attention_small.hip', 'prenet.hip', 'attention_nm.hip', 'wino.hip', 'gemm1.hip', 'wino_wgrad.hip']


def _hipcc():
    for c in (os.environ.get('HIPCC'), shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if c and os.path.exists(c):
            return c
    raise RuntimeError('hipcc not found')


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    hipcc = _hipcc()
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    hdrs = [os.path.join(CSRC, h) for h in os.listdir(CSRC) if h.endswith('.h')]
    hdrs.append(os.path.join(os.path.dirname(HERE), 'include', 'asr_hip.h'))
    objdir = os.path.join(HERE, 'build')
    os.makedirs(objdir, exist_ok=True)
    objs = []
    procs = []
    for s in srcs:
        o = os.path.join(objdir, os.path.basename(s) + '.o')
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            cmd = [hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-fvisibility=hidden', '-c', s, '-o', o]
            if verbose:
                print(' '.join(cmd), flush=True)
            procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode())
            raise RuntimeError('hipcc failed on %s' % s)
    if force or _stale(LIB, objs):
        cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)

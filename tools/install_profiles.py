#!/usr/bin/env python3
"""Copies the summaries tools/collect_profiles.sh left under gpurun_out/<tag>/ into profiles/ under the names bench.py and
the docs use:  python tools/install_profiles.py r02a"""
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAME = {'dfcnn': 'dfcnn_m1', 'se_dfcnn': 'se_dfcnn_m2'}


def main():
    tag = sys.argv[1]
    src = os.path.join(ROOT, 'gpurun_out', tag)
    dst = os.path.join(ROOT, 'profiles')
    n = 0
    for wl in ('dfcnn', 'se_dfcnn', 'transformer', 'e2e_prenet', 'am_lm', 'lm'):
        nm = NAME.get(wl, wl)
        for suffix in ('kernel_stats.csv', 'single_stream_kernel_stats.csv', 'pmc_summary.csv',
                       'traffic.json'):
            f = os.path.join(src, '%s_%s' % (wl, suffix))
            if os.path.exists(f):
                shutil.copy(f, os.path.join(dst, '%s_%s_%s' % (tag, nm, suffix))); n += 1
        for log, label in (('bench_%s.log', 'bench_%s_under_rocprof.json'), ('bench1_%s.log', 'bench_%s_single_stream_under_rocprof.json')):
            f = os.path.join(src, log % wl)
            if not os.path.exists(f):
                continue
            lines = [l for l in open(f, errors='replace') if l.startswith('{"metric"')]
            if lines:
                json.loads(lines[-1])
                open(os.path.join(dst, '%s_%s' % (tag, label % wl)), 'w').write(lines[-1]); n += 1
    print('installed', n, 'files for', tag)


if __name__ == '__main__':
    main()

#!/usr/bin/env python3
"""Reads a rocprofv3 kernel-trace CSV of `bench.py` and reports, for the last timed step, how the wall time splits into
GPU-busy time (union of all kernel intervals), idle gaps, and per-stream sums.  usage: timeline.py <kernel_trace.csv> <steps>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
iv = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Stream_Id', '0')) for r in rows)
# steps are delimited by the Adam kernel (ADAMS=<n> in the environment: n Adam launches per step, e.g. 2 for the joint graph)
import os
adam = [i for i, v in enumerate(iv) if 'adam_tf' in v[2]]
na = int(os.environ.get('ADAMS', '1'))
a, b = adam[-1 - na] + 1, adam[-1] + 1
step = iv[a:b]
t0, t1 = step[0][0], max(v[1] for v in step)
busy, cur_s, cur_e = 0, None, None
for s, e, _, _ in step:
    if cur_e is None or s > cur_e:
        if cur_e is not None: busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
per = {}
for s, e, n, st in step:
    per[st] = per.get(st, 0) + e - s
print('last step: wall %.3f ms, GPU busy (union) %.3f ms, idle %.3f ms, kernels %d' % ((t1 - t0) / 1e6, busy / 1e6, (t1 - t0 - busy) / 1e6, len(step)))
for st, v in sorted(per.items(), key=lambda kv: -kv[1]):
    print('  stream %s: sum of kernel durations %.3f ms' % (st, v / 1e6))
gaps = []
prev_e = step[0][1]
for s, e, n, st in step[1:]:
    if s > prev_e: gaps.append((s - prev_e, n))
    prev_e = max(prev_e, e)
gaps.sort(reverse=True)
print('largest gaps (us, before kernel):')
for g, n in gaps[:12]:
    print('  %7.1f  %s' % (g / 1e3, n.replace('void (anonymous namespace)::', '')[:70]))
if len(sys.argv) > 3 and sys.argv[3] == 'list':
    print('kernels of the last step (start us, duration us, stream, name):')
    for s, e, n, st in step:
        print('  %9.1f %8.1f  s%s  %s' % ((s - t0) / 1e3, (e - s) / 1e3, st, n.replace('void (anonymous namespace)::', '').replace('(anonymous namespace)::', '')[:90]))

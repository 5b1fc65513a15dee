#!/usr/bin/env python3
"""Aggregates rocprofv3 --pmc counter_collection CSVs (one directory per pass) into a per-kernel
table: mean counter value per dispatch, joined with the kernel-trace durations.
usage: python tools/pmc_summary.py out.csv dir1 [dir2 ...] [--traffic out.json]

--traffic writes, per kernel symbol, what bench.py's roofline.traffic reads:
  hbm_bytes_per_launch = (2*FETCH_SIZE + WRITE_SIZE) * 1024   (rocprofv3 reports KB; gfx950 tallies a 128-B read
                                                               request as 64 B -- MI355X_MICROARCH.md, HBM section)
  clock_mhz            = GRBM_GUI_ACTIVE / 8 XCDs / duration
  mfma_busy            = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * GRBM_GUI_ACTIVE / 8)"""
import json
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    name = name.replace('(anonymous namespace)::', '').replace('void ', '')
    return name.split('(')[0]


def main():
    traffic = None
    if '--traffic' in sys.argv:
        i = sys.argv.index('--traffic')
        traffic = sys.argv[i + 1]
        del sys.argv[i:i + 2]
    out = sys.argv[1]
    agg = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    dur = defaultdict(lambda: [0.0, 0])
    for d in sys.argv[2:]:
        for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
            for r in csv.DictReader(open(f)):
                k = short(r['Kernel_Name'])
                c = agg[k][r['Counter_Name']]
                c[0] += float(r['Counter_Value']); c[1] += 1
        for f in glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True):
            for r in csv.DictReader(open(f)):
                k = short(r['Kernel_Name'])
                dur[k][0] += float(r['End_Timestamp']) - float(r['Start_Timestamp']); dur[k][1] += 1
    counters = sorted({c for k in agg for c in agg[k]})
    rows = []
    for k in agg:
        n = max(v[1] for v in agg[k].values())
        row = {'kernel': k, 'dispatches': n, 'avg_us': round(dur[k][0] / max(dur[k][1], 1) / 1e3, 2)}
        for c in counters:
            v = agg[k].get(c)
            row[c] = round(v[0] / v[1], 1) if v else ''
        rows.append(row)
    rows.sort(key=lambda r: -r['avg_us'] * r['dispatches'])
    with open(out, 'w', newline='') as fh:
        w = csv.DictWriter(fh, fieldnames=['kernel', 'dispatches', 'avg_us'] + counters)
        w.writeheader()
        w.writerows(rows)
    print('wrote', out, len(rows), 'kernels')
    if traffic:
        tj = {}
        for r in rows:
            if r.get('FETCH_SIZE', '') == '' or r.get('WRITE_SIZE', '') == '':
                continue
            e = {'avg_us': r['avg_us'], 'hbm_bytes_per_launch': int((2 * r['FETCH_SIZE'] + r['WRITE_SIZE']) * 1024)}
            g = r.get('GRBM_GUI_ACTIVE', '')
            if g != '' and r['avg_us'] > 0:
                e['clock_mhz'] = int(g / 8 / r['avg_us'])
                m = r.get('SQ_VALU_MFMA_BUSY_CYCLES', '')
                if m != '':
                    e['mfma_busy'] = round(m / (1024 * g / 8), 3)
            tj[r['kernel']] = e
        with open(traffic, 'w') as fh:
            json.dump(tj, fh, indent=1)
        print('wrote', traffic)


if __name__ == '__main__':
    main()

#!/bin/bash
# One-GPU table behind the strong-scaling prediction (global batch 32 -> 32 / N per GPU): the plain and SE-DFCNN step at
# B = 32 / 16 / 8 / 4 per GPU, with the three gradient all-reduces through a one-rank RCCL group (--rccl-world1: what a rank of a 1 / 2 / 4 / 8-way
# strong run executes, minus the bytes on xGMI) and without.  bash tools/strong_table.sh <tag>   (on the GPU box; writes gpurun_out/<tag>/)
TAG=${1:-strong}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
for WL in dfcnn se_dfcnn; do
  for B in 32 16 8 4; do
    for F in "" "--rccl-world1"; do
      python3 bench.py --workload $WL --batch $B --steps 40 --warmup 5 --no-cpu-baseline $F 2>$OUT/err_${WL}_${B}.txt | tail -1 > $OUT/line.json || { echo "failed: $WL $B $F"; tail -5 $OUT/err_${WL}_${B}.txt; }
      python3 - "$WL" "$B" "$F" $OUT/line.json <<'PY' | tee -a $OUT/strong_table.txt
import json, sys
wl, b, f, path = sys.argv[1:5]
d = json.loads(open(path).read())
print('%-9s B %2d %-14s %7.3f ms/step  %8.1f utt/s  %6.1f us/utt' % (wl, int(b), f or '(no collective)', d['ms_per_step'], d['value'], 1e3 * d['ms_per_step'] / int(b)))
PY
    done
  done
done
# where the small batches lose: the single-stream kernel table at B = 4 and at B = 32
for B in 4 32; do
  python3 bench.py --workload dfcnn --batch $B --steps 5 --warmup 3 --no-cpu-baseline --kernel-table 2> $OUT/kernel_table_dfcnn_B$B.txt > /dev/null
done

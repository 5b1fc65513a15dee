#!/bin/bash
# cell1 rows-per-workgroup A/B on one box: the kernels alone (tools/bench_cell1.py) and the whole step, per library build
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $ROOT
for L in "$@"; do
  if [ "$L" = "new" ]; then unset LIB; else export LIB=tools/libasrhip_$L.so; fi
  echo "== $L"; python3 tools/bench_cell1.py 2>/dev/null | grep max
done
bash tools/ab_step.sh dfcnn 3 "$@"

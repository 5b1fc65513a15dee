#!/bin/bash
# Causal attention A/B on one box: T = 100 (Language_Model) and T = 512 (configs[3]) per library build, then the two workloads' steps
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $ROOT
for L in "$@"; do
  if [ "$L" = "new" ]; then unset LIB; else export LIB=tools/libasrhip_$L.so; fi
  echo "== $L"
  python3 tools/bench_attention_small.py 100 64 1 2>/dev/null | tail -1
  python3 tools/bench_attention_small.py 512 64 1 2>/dev/null | tail -1
  python3 tools/bench_attention_small.py 512 64 0 2>/dev/null | tail -1
done
bash tools/ab_step.sh lm 2 "$@"
bash tools/ab_step.sh transformer 2 "$@"

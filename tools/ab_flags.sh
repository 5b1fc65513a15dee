#!/bin/bash
# Same-box A/B of bench.py FLAG sets (one library): bash tools/ab_flags.sh <workload> <rounds> "<flags A>" "<flags B>" ...
WL=$1; R=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
python3 bench.py --workload $WL --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
for i in $(seq $R); do
  for F in "$@"; do
    python3 bench.py --workload $WL --steps 30 --warmup 3 --no-cpu-baseline $F 2>/dev/null | \
      python3 -c "import json,sys; print('[$F]', json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"
  done
done

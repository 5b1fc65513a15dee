#!/bin/bash
# Same-box A/B of WHOLE bench steps against other builds of the library (the only comparison that decides a change: a kernel that
# is faster alone can make the two-stream step slower).
# usage (on the GPU box): bash tools/ab_step.sh <workload> <rounds> <lib> [<lib> ...]
#   <lib> = "new" (the in-tree libasrhip.so) or a name N for tools/libasrhip_N.so (tools/build_variant.sh: another revision or other flags of a
#   source file, linked with the other objects of asr_dfcnn_transformer_amd/build/; tools/libasrhip_*.so is git-ignored but
#   travels to the box).  Libraries alternate within every round; one untimed run in front (first-process effect).
WL=$1; R=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
python3 bench.py --workload $WL --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
for i in $(seq $R); do
  for L in "$@"; do
    if [ "$L" = "new" ]; then unset LIB; else export LIB=tools/libasrhip_$L.so; fi
    python3 tools/bench_with_lib.py --workload $WL --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | \
      python3 -c "import json,sys; print('$L', json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"
  done
done

#!/bin/bash
# Kernel trace (start / end timestamps, both streams) of a few bench steps + the timeline of the last one.
# usage (on the GPU box): bash tools/timeline_run.sh <tag> <workload> [bench args]
set -e -o pipefail
TAG=$1; WL=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_$WL -o $WL -- python3 $ROOT/bench.py --workload $WL --steps 4 --warmup 2 --no-cpu-baseline "$@" > $OUT/trace_bench_$WL.log 2>&1
F=$(find $OUT/trace_$WL -name "*kernel_trace.csv" | head -1)
python3 $ROOT/tools/timeline.py $F 4 list > $OUT/timeline_$WL.txt
rm -rf $OUT/trace_$WL

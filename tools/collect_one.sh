#!/bin/bash
# Kernel-trace stats (two-stream and single-stream) of ONE bench workload on the GPU box: bash tools/collect_one.sh <tag> <workload> [bench args]
set -e -o pipefail
TAG=$1; WL=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$WL -o $WL -- python3 $ROOT/bench.py --workload $WL --steps 4 --warmup 2 --no-cpu-baseline "$@" > $OUT/bench_$WL.log 2>&1
find $OUT/stats_$WL -name "*kernel_stats.csv" -exec cp {} $OUT/${WL}_kernel_stats.csv \;
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats1_$WL -o $WL -- python3 $ROOT/bench.py --workload $WL --steps 4 --warmup 2 --no-cpu-baseline --single-stream --no-prefetch "$@" > $OUT/bench1_$WL.log 2>&1
find $OUT/stats1_$WL -name "*kernel_stats.csv" -exec cp {} $OUT/${WL}_single_stream_kernel_stats.csv \;
rm -rf $OUT/stats_$WL $OUT/stats1_$WL
ls $OUT

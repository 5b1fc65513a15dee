#!/bin/bash
# Un-profiled bench lines of every workload (+ T_pad 1000, + host-buffer input) on the GPU box: bash tools/final_benches.sh <tag>
set -e -o pipefail
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
for wl in dfcnn se_dfcnn transformer e2e_prenet am_lm lm; do
  python3 bench.py --workload $wl --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_$wl.json 2> $OUT/bench_$wl.err
  echo "$wl done"
done
python3 bench.py --workload dfcnn --tpad 1000 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_dfcnn_t1000.json 2> $OUT/bench_dfcnn_t1000.err
python3 bench.py --workload se_dfcnn --tpad 1000 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_se_dfcnn_t1000.json 2> $OUT/bench_se_dfcnn_t1000.err
python3 bench.py --workload dfcnn --host-input --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_dfcnn_host_input.json 2> $OUT/bench_dfcnn_host_input.err
python3 bench.py --workload dfcnn --host-input --no-prefetch --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_dfcnn_host_input_inline.json 2> $OUT/bench_dfcnn_host_input_inline.err
python3 bench.py --workload dfcnn --single-stream --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_dfcnn_single_stream.json 2> $OUT/bench_dfcnn_single_stream.err
echo "extras done"

#!/bin/bash
# Round-6 evidence set in one gpurun call: box probe, un-profiled bench lines (driver's command line first), kernel stats, PMC passes.
TAG=${1:-r06w}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
mkdir -p gpurun_out/$TAG
bash tools/bimodal_probe.sh ${TAG}_probe 4 > gpurun_out/$TAG/probe.log 2>&1
echo "probe done"; tail -6 gpurun_out/${TAG}_probe/runs.txt
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/$TAG/bench_dfcnn_driver_command.json 2> gpurun_out/$TAG/bench_dfcnn_driver_command.err
echo "driver command done"; cut -c1-200 gpurun_out/$TAG/bench_dfcnn_driver_command.json
bash tools/final_benches.sh ${TAG}_bench > gpurun_out/$TAG/final_benches.log 2>&1
echo "final benches done"
PHASE=stats bash tools/collect_profiles.sh $TAG > gpurun_out/$TAG/collect_stats.log 2>&1
echo "stats done"
PHASE=pmc bash tools/collect_profiles.sh $TAG > gpurun_out/$TAG/collect_pmc.log 2>&1
echo "pmc done"

#!/bin/bash
# Un-profiled lines of round 6's last state: box probe, the driver's command line (incl. the CPU leg), every workload + extras
TAG=${1:-r06x}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
mkdir -p gpurun_out/$TAG
bash tools/bimodal_probe.sh ${TAG}_probe 4 > gpurun_out/$TAG/probe.log 2>&1
tail -6 gpurun_out/${TAG}_probe/runs.txt
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/$TAG/bench_dfcnn_driver_command.json 2> gpurun_out/$TAG/bench_dfcnn_driver_command.err
bash tools/final_benches.sh $TAG > gpurun_out/$TAG/final_benches.log 2>&1
python3 bench.py --workload se_dfcnn --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/$TAG/bench_se_dfcnn_20.json 2>/dev/null
python3 bench.py --workload dfcnn --streams two --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/$TAG/bench_dfcnn_streams_two.json 2>/dev/null
python3 bench.py --workload dfcnn --rccl-world1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/$TAG/bench_dfcnn_rccl_world1.json 2>/dev/null
python3 bench.py --workload dfcnn --inference --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/$TAG/bench_inf_dfcnn.json 2>/dev/null
python3 bench.py --workload dfcnn --inference --batch 1 --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/$TAG/bench_inf_dfcnn_b1.json 2>/dev/null
for f in gpurun_out/$TAG/bench_*.json; do python3 -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1]); c=d['config']
print('%-44s %9.1f %-14s %8.3f ms  %s' % ('$(basename $f)', d['value'], d['unit'], d['ms_per_step'], c.get('stream_calibration') or ''))"; done | tee gpurun_out/$TAG/summary.txt

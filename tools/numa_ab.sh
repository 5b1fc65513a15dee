#!/bin/bash
# Does the NUMA node the host process runs on move the step?  bench.py pinned to the CPUs of each node, alternating.  bash tools/numa_ab.sh <tag> [rounds] [bench flags]
TAG=${1:-numa}; R=${2:-2}; shift 2
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $ROOT
python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
for f in /sys/class/kfd/kfd/topology/nodes/*/properties; do
  if grep -q "simd_count [1-9]" $f 2>/dev/null; then
    loc=$(grep location_id $f | awk '{print $2}'); dom=$(grep "^domain" $f | awk '{print $2}')
    bdf=$(printf "%04x:%02x:%02x.%x" $dom $((loc >> 8)) $(((loc >> 3) & 31)) $((loc & 7)))
    echo "GPU $bdf numa_node $(cat /sys/bus/pci/devices/$bdf/numa_node)" | tee $OUT/numa_ab.txt
  fi
done
for i in $(seq $R); do
  for nd in /sys/devices/system/node/node*; do
    CPUS=$(cat $nd/cpulist)
    taskset -c $CPUS python3 bench.py --steps 30 --warmup 3 --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print('%-34s %7.3f ms/step   host enqueue %6.3f ms/step' % ('$(basename $nd) cpus $CPUS', d['ms_per_step'], c['host_enqueue_ms_per_step']))" | tee -a $OUT/numa_ab.txt
  done
done

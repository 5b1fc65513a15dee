#!/bin/bash
# Box-state probe: consecutive bench.py processes on ONE box; their two-stream ms/step, host enqueue time, and where the
# process ran (allowed CPUs, NUMA node of the GPU).  When the processes disagree by > 4 %, the pinned variants (CPUs of the GPU's NUMA node
# / of the other node) and an event timeline of a slow process follow.   bash tools/bimodal_probe.sh <tag> [processes]
TAG=${1:-bimodal}; N=${2:-5}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
{
  echo "== host"; nproc; lscpu | grep -i "numa\|model name\|socket" ; echo "allowed: $(taskset -cp $$ 2>/dev/null)"
  for f in /sys/class/kfd/kfd/topology/nodes/*/properties; do
    if grep -q "simd_count [1-9]" $f; then echo "kfd node $f: $(grep -E 'location_id|domain|unique_id|drm_render_minor' $f | tr '\n' ' ')"; fi
  done
  for d in /sys/bus/pci/devices/*; do
    if [ -f $d/class ] && grep -q "^0x1200\|^0x0380\|^0x0302" $d/class 2>/dev/null; then echo "pci $(basename $d) numa_node $(cat $d/numa_node 2>/dev/null) local_cpulist $(cat $d/local_cpulist 2>/dev/null)"; fi
  done
} > $OUT/host.txt 2>&1
run() {   # label, command prefix...
  local label=$1; shift
  "$@" python3 bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print('%-28s %7.3f ms/step   host enqueue %6.3f ms/step' % ('$label', d['ms_per_step'], c['host_enqueue_ms_per_step']))" | tee -a $OUT/runs.txt
}
: > $OUT/runs.txt
for i in $(seq $N); do run "process $i" env ASR_NOP=1; done
python3 - $OUT/runs.txt <<'PY'
import sys
v = [float(l.split()[2]) for l in open(sys.argv[1]) if l.startswith('process')]
print('spread: min %.3f max %.3f  (%.1f %%)' % (min(v), max(v), 100 * (max(v) / min(v) - 1)))
sys.exit(0 if max(v) / min(v) > 1.04 else 7)
PY
if [ $? -eq 0 ]; then
  echo "== processes disagree: pinned variants" | tee -a $OUT/runs.txt
  NODES=$(ls -d /sys/devices/system/node/node* 2>/dev/null | wc -l)
  for nd in $(seq 0 $((NODES-1))); do
    CPUS=$(cat /sys/devices/system/node/node$nd/cpulist)
    run "taskset node$nd ($CPUS) a" taskset -c $CPUS
    run "taskset node$nd ($CPUS) b" taskset -c $CPUS
  done
  run "unpinned again" env ASR_NOP=1
  python3 bench.py --steps 30 --warmup 3 --no-cpu-baseline --single-stream 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('--single-stream               %7.3f ms/step' % d['ms_per_step'])" | tee -a $OUT/runs.txt
  python3 bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-prefetch 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('--no-prefetch                 %7.3f ms/step' % d['ms_per_step'])" | tee -a $OUT/runs.txt
  python3 tools/timeline_events.py > $OUT/timeline.txt 2>&1
fi
cat $OUT/host.txt

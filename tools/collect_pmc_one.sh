#!/bin/bash
# PMC passes (counters only with --kernel-trace; FETCH_SIZE and WRITE_SIZE in passes of their own) of ONE bench workload on the GPU
# box, single-stream: bash tools/collect_pmc_one.sh <tag> <workload> [bench args]   -> gpurun_out/<tag>/<workload>_pmc_summary.csv, _traffic.json
set -e -o pipefail
TAG=$1; WL=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for ctr in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA"; do
  i=$((i+1))
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/pmc_${WL}_$i -o p -- python3 $ROOT/bench.py --workload $WL --steps 2 --warmup 1 --no-cpu-baseline --single-stream --no-prefetch "$@" > $OUT/log_pmc_${WL}_$i.txt 2>&1 || echo "pass $i failed"
  echo "pmc $WL pass $i done"
done
python3 $ROOT/tools/pmc_summary.py $OUT/${WL}_pmc_summary.csv $OUT/pmc_${WL}_1 $OUT/pmc_${WL}_2 $OUT/pmc_${WL}_3 $OUT/pmc_${WL}_4 $OUT/pmc_${WL}_5 --traffic $OUT/${WL}_traffic.json
rm -rf $OUT/pmc_${WL}_[0-9]

#!/bin/bash
# PMC passes over the Transformer bench: matrix-pipe busy / waits / LDS conflicts per kernel
set -e -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_tr
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for ctr in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/p$i -o p -- python3 $ROOT/bench.py --workload transformer --steps 2 --warmup 1 --no-cpu-baseline > $OUT/log$i.txt 2>&1 || echo "pass $i failed"
  echo "pass $i done"
done
python3 $ROOT/tools/pmc_summary.py $OUT/summary.csv $OUT/p1 $OUT/p2 $OUT/p3 $OUT/p4
rm -rf $OUT/p[0-9]

#!/bin/bash
# PMC passes over the DFCNN bench in split-bf16 mode (single stream): where do the bx6 conv kernels' cycles go
set -e -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_bx6_conv
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export ASR_BX6=1 ASR_DUAL_STREAM=0
i=0
for ctr in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/p$i -o p -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-experimental --no-prefetch > $OUT/log$i.txt 2>&1 || echo "pass $i failed"
  echo "pass $i done"
done
python3 $ROOT/tools/pmc_summary.py $OUT/summary.csv $OUT/p1 $OUT/p2 $OUT/p3 $OUT/p4 $OUT/p5 $OUT/p6
rm -rf $OUT/p[0-9]

#!/bin/bash
# PMC passes over tools/bench_attention_drop.py (configs[3]'s attention call with dropout 0.2 and without, causal and not):
# vector-instruction and matrix-pipe counters per kernel instantiation -> gpurun_out/<tag>/attention_pmc_summary.csv
set -e -o pipefail
TAG=${1:-pmc_attn}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for ctr in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_INSTS_SALU" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/p$i -o p -- python3 $ROOT/tools/bench_attention_drop.py > $OUT/log$i.txt 2>&1 || echo "pass $i failed"
  echo "pass $i done"
done
python3 $ROOT/tools/pmc_summary.py $OUT/attention_pmc_summary.csv $OUT/p1 $OUT/p2 $OUT/p3 $OUT/p4
rm -rf $OUT/p[0-9]

#!/bin/bash
# FETCH_SIZE / WRITE_SIZE / MFMA-busy passes over tools/bench_wino_ab.py for one library: bash tools/pmc_wino_ab.sh <tag> <name> [lib]
set -e -o pipefail
TAG=$1; NAME=$2; LIBP=$3
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
if [ -n "$LIBP" ]; then export LIB=$ROOT/$LIBP; fi
cd /tmp && export TMPDIR=/tmp
i=0
for ctr in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/pmcw_${NAME}_$i -o p -- python3 $ROOT/tools/bench_wino_ab.py > $OUT/log_pmcw_${NAME}_$i.txt 2>&1 || echo "pass $i failed"
done
python3 $ROOT/tools/pmc_summary.py $OUT/wino_ab_${NAME}_pmc_summary.csv $OUT/pmcw_${NAME}_1 $OUT/pmcw_${NAME}_2 $OUT/pmcw_${NAME}_3 --traffic $OUT/wino_ab_${NAME}_traffic.json
rm -rf $OUT/pmcw_${NAME}_[0-9]

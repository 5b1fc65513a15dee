#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats for the six bench workloads and the separate PMC passes for
# the headline one.  usage: bash tools/collect_profiles.sh <tag>   (outputs under gpurun_out/<tag>/, summaries are
# then copied into profiles/ by hand).  rocprofv3 gets the python program directly after `--` (no wrapper hops).
set -e -o pipefail
TAG=${1:-r01x}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PHASE=${PHASE:-all}          # stats | pmc | all  (two gpurun calls of <= 20 min each)
if [ "$PHASE" != "pmc" ]; then
for wl in dfcnn se_dfcnn transformer e2e_prenet am_lm lm; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$wl -o $wl -- \
      python3 $ROOT/bench.py --workload $wl --steps 4 --warmup 2 --no-cpu-baseline > $OUT/bench_$wl.log 2>&1
  echo "stats $wl done"
done
# the same steps with the backward on ONE stream: per-kernel durations that do not overlap
for wl in dfcnn se_dfcnn; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats1_$wl -o $wl -- \
      python3 $ROOT/bench.py --workload $wl --steps 4 --warmup 2 --no-cpu-baseline --single-stream --no-prefetch > $OUT/bench1_$wl.log 2>&1
  echo "single-stream stats $wl done"
done
for wl in dfcnn se_dfcnn transformer e2e_prenet am_lm lm; do
  find $OUT/stats_$wl -name "*kernel_stats.csv" -exec cp {} $OUT/${wl}_kernel_stats.csv \;
done
for wl in dfcnn se_dfcnn; do
  find $OUT/stats1_$wl -name "*kernel_stats.csv" -exec cp {} $OUT/${wl}_single_stream_kernel_stats.csv \;
done
rm -rf $OUT/stats_* $OUT/stats1_*
fi
if [ "$PHASE" != "stats" ]; then
# PMC passes (counters only with --kernel-trace; FETCH_SIZE and WRITE_SIZE in passes of their own): every workload, so
# that each bench line can carry roofline.traffic for its dominant kernel
for wl in dfcnn se_dfcnn transformer e2e_prenet am_lm lm; do
  i=0
  for ctr in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES"; do
    i=$((i+1))
    rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/pmc_${wl}_$i -o p -- \
        python3 $ROOT/bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline --single-stream --no-prefetch > $OUT/log_pmc_${wl}_$i.txt 2>&1
    echo "pmc $wl pass $i done"
  done
  python3 $ROOT/tools/pmc_summary.py $OUT/${wl}_pmc_summary.csv $OUT/pmc_${wl}_1 $OUT/pmc_${wl}_2 $OUT/pmc_${wl}_3 $OUT/pmc_${wl}_4 \
      --traffic $OUT/${wl}_traffic.json
done
rm -rf $OUT/pmc_*_[0-9]
fi
ls -la $OUT

#!/bin/bash
# Timing-only ablation of wino_wgrad_kernel on the GPU box: rebuilds wino_wgrad.hip with -DWW_ABL=<mask> (1 no DMA, 2 no LDS reads,
# 4 no stage barrier -- results are wrong by construction), relinks the library in place and times the DFCNN layer shapes.
# Usage: tools/ablate_wino_wgrad.sh <outfile> <mask> [<mask> ...]
set -e
cd "$(dirname "$0")/.."
out=$1; shift
P=asr_dfcnn_transformer_amd
cp $P/libasrhip.so /tmp/libasrhip_good.so
trap 'cp /tmp/libasrhip_good.so $P/libasrhip.so' EXIT      # the good library comes back even when a compile or a bench fails partway
objs=$(ls $P/build/*.hip.o | grep -v wino_wgrad.hip.o)
for m in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DASR_DEV_HOOKS -DWW_ABL=$m -I include -c $P/csrc/wino_wgrad.hip -o /tmp/ww_abl.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $P/libasrhip.so $objs /tmp/ww_abl.o
  echo "== WW_ABL=$m" >> $out
  python tools/bench_wino_wgrad.py 2>&1 | grep -v amdgpu.ids | cut -c1-100 >> $out
done
cp /tmp/libasrhip_good.so $P/libasrhip.so

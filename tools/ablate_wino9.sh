#!/bin/bash
# Timing-only ablation of wino9_kernel on the GPU box: rebuilds wino.hip with -DWINO_ABL=<mask> (1 no patch reads, 2 no weight
# reads, 4 no raw DMA, 8 no weight DMA, 16 no chunk barrier, 32 no item tail -- results are wrong by construction), relinks the
# library in place and times the DFCNN layer shapes.  Usage: tools/ablate_wino9.sh <outfile> <mask> [<mask> ...]
set -e
cd "$(dirname "$0")/.."
out=$1; shift
P=asr_dfcnn_transformer_amd
cp $P/libasrhip.so /tmp/libasrhip_good.so
trap 'cp /tmp/libasrhip_good.so $P/libasrhip.so' EXIT      # the good library comes back even when a compile or a bench fails partway
objs=$(ls $P/build/*.hip.o | grep -v wino.hip.o)
for m in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DWINO_ABL=$m ${WINO_EXTRA:-} -I include -c $P/csrc/wino.hip -o /tmp/wino_abl.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $P/libasrhip.so $objs /tmp/wino_abl.o
  echo "== WINO_ABL=$m" >> $out
  ONLY=${ONLY:-c} python tools/bench_wino.py 2>&1 | grep -v amdgpu.ids | cut -c1-110 >> $out
done
cp /tmp/libasrhip_good.so $P/libasrhip.so

#!/bin/bash
# Timing-only ablation of wino11_kernel on the GPU box (development): rebuilds wino.hip with -DW11_ABL=<mask> (1 no patch reads,
# 2 no weight reads, 4 no raw DMA, 8 no weight DMA, 16 no chunk barrier, 32 no item tail -- results are wrong by construction),
# relinks the library in place and times the plain DFCNN's launches.  Usage: tools/ablate_wino11.sh <outfile> <mask> [<mask> ...]
set -e
cd "$(dirname "$0")/.."
out=$1; shift
P=asr_dfcnn_transformer_amd
cp $P/libasrhip.so /tmp/libasrhip_good.so
trap 'cp /tmp/libasrhip_good.so $P/libasrhip.so' EXIT      # the good library comes back even when a compile or a bench fails partway
objs=$(ls $P/build/*.hip.o | grep -v "/wino.hip.o")
for m in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DW11_ABL=$m ${WINO_EXTRA:-} -I include -c $P/csrc/wino.hip -o /tmp/wino_abl.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $P/libasrhip.so $objs /tmp/wino_abl.o
  echo "== W11_ABL=$m" >> $out
  python tools/bench_wino_ab.py 11 2>&1 | grep -v amdgpu.ids | cut -c1-120 >> $out
done

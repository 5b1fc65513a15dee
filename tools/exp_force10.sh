#!/bin/bash
# Experiment behind the kernel-selection rule in wino_impl (csrc/wino.hip): wino10_kernel forced for EVERY Winograd launch
# (-DWINO_FORCE10) against the rule, per DFCNN layer and for the whole step.  Runs on the GPU box; restores the library.
set -e
cd "$(dirname "$0")/.."
P=asr_dfcnn_transformer_amd
cp $P/libasrhip.so /tmp/libasrhip_good.so
trap 'cp /tmp/libasrhip_good.so $P/libasrhip.so' EXIT      # the good library comes back even when a compile or a bench fails partway
objs=$(ls $P/build/*.hip.o | grep -v "/wino.hip.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DWINO_FORCE10 -I include -c $P/csrc/wino.hip -o /tmp/wino_f10.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $P/libasrhip.so $objs /tmp/wino_f10.o
echo "== wino10_kernel forced"
python tools/bench_wino.py 2>&1 | grep -v amdgpu.ids | cut -c1-130
python bench.py --steps 20 --warmup 5 --kernel-table --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | cut -c1-200 | head -8
cp /tmp/libasrhip_good.so $P/libasrhip.so
echo "== the library's own rule"
python tools/bench_wino.py 2>&1 | grep -v amdgpu.ids | cut -c1-130

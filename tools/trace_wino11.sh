#!/bin/bash
# builds the trace library next to the tools and runs tools/trace_wino11.py (GPU box)
set -e
cd "$(dirname "$0")/.."
P=asr_dfcnn_transformer_amd
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DASR_DEV_HOOKS -DW11_TRACE ${W11_EXTRA:-} -c $P/csrc/wino.hip -o /tmp/wino_trace.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/libasrhip_trace.so $(ls $P/build/*.o | grep -v "/wino.hip.o") /tmp/wino_trace.o
python tools/trace_wino11.py

#!/bin/bash
# builds the trace library next to the tools (here, CPU container or GPU box) and, with "run", runs tools/trace_wino11.py (GPU box)
# usage: tools/trace_wino11.sh [build|run] [name] [extra hipcc flags]      -> tools/libasrhip_trace[_name].so
set -e
cd "$(dirname "$0")/.."
mode=${1:-run}; name=${2:-}; shift 2 || true
P=asr_dfcnn_transformer_amd
lib=tools/libasrhip_trace${name:+_$name}.so
if [ "$mode" = "build" ] || [ ! -f $lib ]; then
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DASR_DEV_HOOKS -DW11_TRACE ${W11_EXTRA:-} "$@" -c $P/csrc/wino.hip -o /tmp/wino_trace.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $lib $(ls $P/build/*.o | grep -v "/wino.hip.o") /tmp/wino_trace.o
fi
if [ "$mode" = "run" ]; then TRACE_LIB=$(basename $lib) python tools/trace_wino11.py; fi

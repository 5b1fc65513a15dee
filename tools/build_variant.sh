#!/bin/bash
# Development A/B: builds tools/libasrhip_<name>.so = the in-tree objects with ONE source recompiled with extra flags.
# usage: tools/build_variant.sh <name> <source.hip> <flags...>      e.g. tools/build_variant.sh nopair wino.hip -DW11_PAIR=0
# (git-ignored; travels to the GPU box with the snapshot; run benches against it with LIB=tools/libasrhip_<name>.so)
set -e
cd "$(dirname "$0")/.."
name=$1; src=$2; shift 2
P=asr_dfcnn_transformer_amd
python3 -c "from asr_dfcnn_transformer_amd import _build; _build.build(verbose=False)"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden "$@" -c $P/csrc/$src -o /tmp/variant_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/libasrhip_$name.so $(ls $P/build/*.hip.o | grep -v "/$src.o") /tmp/variant_$name.o
echo tools/libasrhip_$name.so

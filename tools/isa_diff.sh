#!/bin/bash
# Did a source change alter the code of kernels it was not meant to touch?  Compiles csrc/<file.hip> of a git commit and of the working tree to
# gfx950 assembly and diffs every kernel whose mangled name matches <pattern>, comments stripped (round 5: a template generalisation left the
# instruction COUNTS of wino_wgrad4_kernel<64> / <32> equal but re-allocated their registers -- same standalone times, two-stream step +0.09 ms).
# usage: tools/isa_diff.sh <commit> <file.hip> [pattern]      e.g. tools/isa_diff.sh e44d233 wino.hip wino11_kernel
set -e
cd "$(dirname "$0")/.."
commit=$1; f=$2; pat=${3:-.}
tmp=$(mktemp -d)
mkdir -p $tmp/old && git archive $commit asr_dfcnn_transformer_amd/csrc include | tar -x -C $tmp/old
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -S $tmp/old/asr_dfcnn_transformer_amd/csrc/$f -o $tmp/old.s 2>/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -S asr_dfcnn_transformer_amd/csrc/$f -o $tmp/new.s 2>/dev/null
for k in $(grep -oE "^_Z[A-Za-z0-9_]+" $tmp/old.s $tmp/new.s | sed 's/.*://' | sort -u | grep -E "$pat"); do
  for v in old new; do awk -v K="$k" 'index($0,K":")==1{f=1} f{print} /s_endpgm/{if(f)exit}' $tmp/$v.s | grep -v "^\s*;" | sed 's/;.*//; s/\.LBB[0-9]*_/.LBB_/g; s/\.Ltmp[0-9]*/.Ltmp/g' > $tmp/k_$v.s; done
  lo=$(wc -l < $tmp/k_old.s); ln=$(wc -l < $tmp/k_new.s)
  if [ "$lo" = "0" ]; then echo "NEW      $k ($ln lines)"; elif [ "$ln" = "0" ]; then echo "GONE     $k"; else
    d=$(diff $tmp/k_old.s $tmp/k_new.s | grep -c "^[<>]" || true); [ "$d" = "0" ] && echo "same     $k ($lo lines)" || echo "DIFFERS  $k ($lo -> $ln lines, $d changed)"; fi
done
rm -rf $tmp

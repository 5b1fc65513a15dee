#!/bin/bash
# Register / spill / scratch metadata of the kernels of one source: tools/kernel_regs.sh <source.hip> [pattern] [extra hipcc flags]
cd "$(dirname "$0")/.."
src=$1; pat=${2:-.}; shift 2
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only "$@" -S asr_dfcnn_transformer_amd/csrc/$src -o /tmp/kregs.s 2>/dev/null
grep -E "^\s+\.name:|\.vgpr_count|\.private_segment_fixed_size|\.sgpr_spill_count|\.vgpr_spill_count|\.sgpr_count" /tmp/kregs.s | paste - - - - - - | grep -E "$pat" | sed 's/  */ /g; s/\.private_segment_fixed_size/scratch/; s/_count//g' | cut -c1-200

// Development hooks of the Winograd kernels.  The PRODUCT build (asr_dfcnn_transformer_amd/_build.py) never defines
// ASR_DEV_HOOKS and therefore never includes this file; tools/trace_wino11.sh and tools/ablate_wino_wgrad.sh do.
//   wino.hip        -DASR_DEV_HOOKS -DW11_TRACE [-DW11_TRACE_WG=<workgroup>]   in-kernel phase stamps of wino11_kernel
//   wino.hip        -DASR_DEV_HOOKS -DW11_ONE_PER_CU   one persistent workgroup per CU instead of two (launcher only)
//   wino_wgrad.hip  -DASR_DEV_HOOKS -DWW_ABL=<mask> (1: no DMA -- results wrong by construction, timing only)
//                   -DWW_SLOT(i)=<expr>  where a DMA piece goes between the eight MFMAs of a tile pair
#pragma once

#ifdef W11_TRACE
__device__ long long w11_trace_buf[8 * 8 * 16];           // [item 8][wave 8][stamp 16], workgroup W11_TRACE_WG only
#ifndef W11_TRACE_WG
#define W11_TRACE_WG 0
#endif
#define W11T_ON (blockIdx.x == W11_TRACE_WG && titem >= 0 && titem < 8 && lane == 0)
#define W11T(k) do { if (W11T_ON) w11_trace_buf[(titem * 8 + wave) * 16 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#define W11T_REALTIME(k) do { if (W11T_ON) w11_trace_buf[(titem * 8 + wave) * 16 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define W11_TRACE_DUMP                                                                                                   \
    extern "C" int asr_w11_trace_dump(long long* host) {                                                                 \
        return hipMemcpyFromSymbol(host, HIP_SYMBOL(w11_trace_buf), sizeof(long long) * 8 * 8 * 16) == hipSuccess ? 0 : 1; \
    }
#endif

#if defined(WW_ABL) && (WW_ABL & 1)
#define WW_NO_DMA 1
#endif

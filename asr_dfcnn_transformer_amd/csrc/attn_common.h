// Helpers shared by the Transformer kernels (transformer.hip) and the fused attention kernels (attention.hip).
#pragma once
#include "asr_common.h"
#include <math.h>

namespace {

constexpr float MASK_FILL = -4294967296.0f;     // float32(-2**32 + 1)
constexpr int DH = 64;                          // head width (hidden_units / num_heads = 512 / 8)
constexpr int KP = DH + 4;                      // LDS pitch of tiles read with ds_read_b128
// The fp32 MFMA shares the SIMD's vector issue with every other vector instruction (tools/mfma_valu.hip: an MFMA-only
// wave and a VALU-only wave on one SIMD take the SUM of their times), so each vector instruction of the softmax costs
// its full issue time.  The scores are therefore kept in base-2 units (q pre-scaled by log2(e) / sqrt(d)): one v_sub +
// one v_exp_f32 per element instead of expf's 13 instructions, and the key mask is one compare + select against a per-key
// bias staged with the tile (a wave-uniform branch around the mask code cost 300 spilled registers instead).
constexpr float LOG2E = 1.44269504088896340736f;
constexpr float QSCALE2 = 0.125f * LOG2E;        // 1 / sqrt(64), in base-2 units
constexpr float FILL2 = MASK_FILL * LOG2E;       // the fill value in the same units
__device__ __forceinline__ float ex2(float x) { return __builtin_amdgcn_exp2f(x); }
// min / max as the bare instructions: fminf / fmaxf on an MFMA result come with a canonicalising v_max_f32 x, x each (IEEE mode),
// one more vector instruction per score in kernels where every vector instruction adds to the MFMA time.  No operand is a NaN
// here (scores are finite, the key bias is +-inf or the fill value).
__device__ __forceinline__ float vmin(float a, float b) { float d; asm("v_min_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ float vmax(float a, float b) { float d; asm("v_max_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ float vmax3(float a, float b, float c) { float d; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }

__device__ __forceinline__ int rowidx(int r, int lh) { return (r & 3) + 8 * (r >> 2) + 4 * lh; }

// Counter-based dropout mask (tf.layers.dropout, transformer.py:111,154,226; model.py:290): element `idx` of the tensor
// drawn for `seed` is kept when the top 24 bits of a murmur3-finalised hash reach the threshold rate * 2^24; kept values
// are scaled by 1 / (1 - rate).  The same function regenerates the mask in the backward pass (nothing is stored) and
// in oracle/transformer.py.  TensorFlow's own random stream cannot be reproduced; parity is against this generator.
__device__ __forceinline__ bool drop_keep(uint32_t idx, uint32_t seed, uint32_t thr) {
    uint32_t h = idx * 0x9E3779B1u + seed;
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return (h >> 8) >= thr;
}

}  // namespace

static inline uint32_t drop_threshold(float rate) {
    double t = (double)rate * 16777216.0;
    if (t < 0) t = 0;
    if (t > 16777215.0) t = 16777215.0;
    return (uint32_t)(t + 0.5);
}

// Internal helpers shared by the gfx950 kernels of libasrhip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/asr_hip.h"

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

#define ASR_WAVE 64
// cross-file helpers that are NOT part of the C ABI (include/asr_hip.h): kept out of the dynamic symbol table
#define ASR_INTERNAL __attribute__((visibility("hidden")))

void asr_set_error(const char* what, hipError_t e);
// name of the contraction kernel instantiation a launcher has just enqueued (asr_last_kernel(), include/asr_hip.h)
void asr_set_last_kernel(const char* name);

#include <stdio.h>
// one formatted name per launcher instantiation (built on first use), then a pointer store per launch
#define ASR_NOTE_KERNEL(...)                                            \
    do {                                                                \
        static char nm__[160];                                          \
        static bool done__ = false;                                     \
        if (!done__) { snprintf(nm__, sizeof(nm__), __VA_ARGS__); done__ = true; } \
        asr_set_last_kernel(nm__);                                      \
    } while (0)

#define ASR_CHECK_LAUNCH(name)                                   \
    do {                                                         \
        hipError_t e__ = hipGetLastError();                      \
        if (e__ != hipSuccess) {                                 \
            asr_set_error(name, e__);                            \
            return ASR_ERR_LAUNCH;                               \
        }                                                        \
    } while (0)

static inline int asr_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// Partial rows a gated launch is about to write, against what the caller's partials buffer holds: on entry *rows is that capacity in
// rows (0: not stated), on success it becomes the number the launch writes.  Called by every launcher BEFORE its hipLaunchKernelGGL, so
// that a disagreement between a launcher's tiling and the entry point's workspace formula is an ASR_ERR_UNSUPPORTED, never an overrun
// that is noticed afterwards.
static inline bool asr_gate_rows_fit(int* rows, int needed) {
    if (!rows) return true;
    if (*rows > 0 && needed > *rows) return false;
    *rows = needed;
    return true;
}

// Bijective XCD-aware remap: consecutive logical tiles land on the same XCD (blocks are
// dealt round-robin over the 8 XCDs), so neighbouring tiles share halo rows / weight
// panels in that XCD's L2.  Speed only; any placement is correct.
__device__ __forceinline__ int asr_xcd_swizzle(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, local = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
}

__device__ __forceinline__ float asr_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double asr_wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float asr_wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

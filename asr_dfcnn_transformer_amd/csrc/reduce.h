// Deterministic column sums: out[c] = sum_r x[r*ldx + c], fixed summation order.
#pragma once
#include "asr_common.h"

namespace asr_reduce {

constexpr int kSplits = 64;

static __global__ void colsum_pass_kernel(const float* __restrict__ x, int rows, int cols, long ldx,
                                   int rows_per_split, float* __restrict__ out, long ldo) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= cols) return;
    const int r0 = blockIdx.y * rows_per_split;
    int r1 = r0 + rows_per_split;
    if (r1 > rows) r1 = rows;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int r = r0;
    for (; r + 4 <= r1; r += 4) {       // four independent chains, folded in a fixed order
        s0 += x[(long)r * ldx + c];
        s1 += x[(long)(r + 1) * ldx + c];
        s2 += x[(long)(r + 2) * ldx + c];
        s3 += x[(long)(r + 3) * ldx + c];
    }
    for (; r < r1; ++r) s0 += x[(long)r * ldx + c];
    out[(long)blockIdx.y * ldo + c] = (s0 + s1) + (s2 + s3);
}

inline size_t colsum_tmp_floats(int rows, int cols) {
    return rows > kSplits ? (size_t)kSplits * cols : 4;
}

// tmp: colsum_tmp_floats(rows, cols) floats (may be null when rows <= kSplits)
inline int colsum(const float* x, int rows, int cols, long ldx, float* out, float* tmp, hipStream_t st) {
    const int threads = 256;
    const int gx = asr_cdiv(cols, threads);
    if (rows <= kSplits) {
        hipLaunchKernelGGL(colsum_pass_kernel, dim3(gx, 1), dim3(threads), 0, st, x, rows, cols, ldx, rows, out, (long)cols);
    } else {
        const int rps = asr_cdiv(rows, kSplits);
        const int ns = asr_cdiv(rows, rps);
        hipLaunchKernelGGL(colsum_pass_kernel, dim3(gx, ns), dim3(threads), 0, st, x, rows, cols, ldx, rps, tmp, (long)cols);
        hipLaunchKernelGGL(colsum_pass_kernel, dim3(gx, 1), dim3(threads), 0, st, tmp, ns, cols, (long)cols, ns, out, (long)cols);
    }
    ASR_CHECK_LAUNCH("colsum");
    return ASR_OK;
}

}  // namespace asr_reduce

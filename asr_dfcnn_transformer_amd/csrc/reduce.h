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

// Column sums of a [rows][ld] partial matrix whose columns form up to four consecutive segments that go
// to different destinations (e.g. {dscale, dshift, dbias} of a cell): one launch instead of one per segment.
struct Multi {
    int nseg;
    int width[4];       // columns per segment
    float* out[4];
};

static __global__ void colsum_multi_kernel(const float* __restrict__ x, int rows, int cols, long ldx,
                                           int rows_per_split, float* __restrict__ tmp, Multi m) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= cols) return;
    const int r0 = blockIdx.y * rows_per_split;
    int r1 = r0 + rows_per_split;
    if (r1 > rows) r1 = rows;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int r = r0;
    for (; r + 4 <= r1; r += 4) {
        s0 += x[(long)r * ldx + c];
        s1 += x[(long)(r + 1) * ldx + c];
        s2 += x[(long)(r + 2) * ldx + c];
        s3 += x[(long)(r + 3) * ldx + c];
    }
    for (; r < r1; ++r) s0 += x[(long)r * ldx + c];
    const float v = (s0 + s1) + (s2 + s3);
    if (tmp) { tmp[(long)blockIdx.y * cols + c] = v; return; }
    int cc = c;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (i < m.nseg) {
            if (cc < m.width[i]) { m.out[i][cc] = v; return; }
            cc -= m.width[i];
        }
    }
}

// tmp: colsum_tmp_floats(rows, total columns) floats
inline int colsum_multi(const float* x, int rows, long ldx, const Multi& m, float* tmp, hipStream_t st) {
    int cols = 0;
    for (int i = 0; i < m.nseg; ++i) cols += m.width[i];
    const int threads = 256;
    const int gx = asr_cdiv(cols, threads);
    if (rows <= kSplits) {
        hipLaunchKernelGGL(colsum_multi_kernel, dim3(gx, 1), dim3(threads), 0, st, x, rows, cols, ldx, rows, (float*)nullptr, m);
    } else {
        const int rps = asr_cdiv(rows, kSplits);
        const int ns = asr_cdiv(rows, rps);
        hipLaunchKernelGGL(colsum_multi_kernel, dim3(gx, ns), dim3(threads), 0, st, x, rows, cols, ldx, rps, tmp, m);
        hipLaunchKernelGGL(colsum_multi_kernel, dim3(gx, 1), dim3(threads), 0, st, (const float*)tmp, ns, cols, (long)cols, ns, (float*)nullptr, m);
    }
    ASR_CHECK_LAUNCH("colsum_multi");
    return ASR_OK;
}

}  // namespace asr_reduce

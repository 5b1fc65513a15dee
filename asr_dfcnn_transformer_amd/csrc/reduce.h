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

// Tall, narrow matrices (bias gradients over millions of pixels x 64 channels): the column-per-thread pass above
// would run on cols/256 x 64 workgroups.  Here cols/4 threads own one row (float4 each), 256/(cols/4) rows per
// workgroup iteration, up to kTallBlocks workgroups; block partials are folded by the two-level pass.
constexpr int kTallBlocks = 1024;
constexpr int kTallMinRows = 16384;

inline bool colsum_is_tall(int rows, int cols) {
    const int cpp = cols / 4;
    return rows >= kTallMinRows && (cols & 3) == 0 && cpp >= 1 && cpp <= 256 && (256 % cpp) == 0;
}
inline int colsum_tall_blocks(int rows, int cols) {
    const int rpb = 256 / (cols / 4);
    const int nb = asr_cdiv(rows, (long)rpb * 8);
    return nb > kTallBlocks ? kTallBlocks : nb;
}

static __global__ __launch_bounds__(256) void colsum_tall_kernel(const float* __restrict__ x, int rows, int cols, long ldx,
                                                                 float* __restrict__ partial) {
    __shared__ float red[1024];                       // [rows per block][cols] = 256 * 4 floats
    const int cpp = cols / 4, rpb = 256 / cpp;
    const int c4 = threadIdx.x % cpp, rl = threadIdx.x / cpp;
    float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
    long r = (long)blockIdx.x * rpb + rl;
    const long stride = (long)gridDim.x * rpb;
    for (; r + stride < rows; r += 2 * stride) {      // two independent chains, folded in a fixed order
        const float4 a = *(const float4*)(x + r * ldx + c4 * 4);
        const float4 b = *(const float4*)(x + (r + stride) * ldx + c4 * 4);
        s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w;
        s1.x += b.x; s1.y += b.y; s1.z += b.z; s1.w += b.w;
    }
    if (r < rows) {
        const float4 a = *(const float4*)(x + r * ldx + c4 * 4);
        s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w;
    }
    *(float4*)(red + rl * cols + c4 * 4) = make_float4(s0.x + s1.x, s0.y + s1.y, s0.z + s1.z, s0.w + s1.w);
    __syncthreads();
    for (int c = threadIdx.x; c < cols; c += 256) {
        float t = 0.f;
        for (int k = 0; k < rpb; ++k) t += red[k * cols + c];
        partial[(long)blockIdx.x * cols + c] = t;
    }
}

// Wide matrices with a few thousand rows (the bias gradient of a [6400][1536] head, of [32768][2048] feed-forward layers): too
// wide for the tall form, and the column-per-thread pass runs them on a handful of waves per CU with one 4-byte load per row in
// flight (35 us for 39 MB).  Here a workgroup owns a strip of 64 columns x one row split: sixteen threads cover the strip's 256
// bytes of a row, sixteen row lanes stride over the split's rows with four independent float4 chains; the row lanes are folded
// through LDS in a fixed order.  Split partials go to the two-level pass's second level.
constexpr int kStripMinRows = 1024;

inline bool colsum_is_strip(int rows, int cols) { return rows >= kStripMinRows && (cols & 3) == 0 && cols >= 64 && !colsum_is_tall(rows, cols); }
inline int colsum_strip_splits(int rows, int cols) {
    int ns = 2048 / ((cols + 63) / 64);
    if (ns > kSplits) ns = kSplits;
    if (ns > rows / 64) ns = rows / 64;
    return ns < 1 ? 1 : ns;
}

static __global__ __launch_bounds__(256) void colsum_strip_kernel(const float* __restrict__ x, int rows, int cols, long ldx,
                                                                  int rows_per_split, float* __restrict__ partial) {
    __shared__ float red[16 * 64];
    const int ct = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const float* xc = x + blockIdx.x * 64 + ct * 4;
    const int r0 = blockIdx.y * rows_per_split;
    int r1 = r0 + rows_per_split;
    if (r1 > rows) r1 = rows;
    if (blockIdx.x * 64 + ct * 4 >= cols) r1 = r0;     // the last strip of a width that is no multiple of 64 (cols % 4 == 0): these lanes own no column
    float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0, s2 = s0, s3 = s0;
    int r = r0 + rl;
    for (; r + 48 < r1; r += 64) {
        const float4 a = *(const float4*)(xc + (long)r * ldx);
        const float4 b = *(const float4*)(xc + (long)(r + 16) * ldx);
        const float4 c = *(const float4*)(xc + (long)(r + 32) * ldx);
        const float4 d = *(const float4*)(xc + (long)(r + 48) * ldx);
        s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w;
        s1.x += b.x; s1.y += b.y; s1.z += b.z; s1.w += b.w;
        s2.x += c.x; s2.y += c.y; s2.z += c.z; s2.w += c.w;
        s3.x += d.x; s3.y += d.y; s3.z += d.z; s3.w += d.w;
    }
    for (; r < r1; r += 16) {
        const float4 a = *(const float4*)(xc + (long)r * ldx);
        s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w;
    }
    *(float4*)(red + rl * 64 + ct * 4) = make_float4((s0.x + s1.x) + (s2.x + s3.x), (s0.y + s1.y) + (s2.y + s3.y),
                                                     (s0.z + s1.z) + (s2.z + s3.z), (s0.w + s1.w) + (s2.w + s3.w));
    __syncthreads();
    if (threadIdx.x < 64) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k * 64 + threadIdx.x];
        if (blockIdx.x * 64 + threadIdx.x < cols) partial[(long)blockIdx.y * cols + blockIdx.x * 64 + threadIdx.x] = t;
    }
}

inline size_t colsum_tmp_floats(int rows, int cols) {
    if (colsum_is_tall(rows, cols)) return (size_t)(colsum_tall_blocks(rows, cols) + kSplits) * cols;
    return rows > kSplits ? (size_t)kSplits * cols : 4;
}

// tmp: colsum_tmp_floats(rows, cols) floats (may be null when rows <= kSplits)
inline int colsum(const float* x, int rows, int cols, long ldx, float* out, float* tmp, hipStream_t st) {
    const int threads = 256;
    const int gx = asr_cdiv(cols, threads);
    const bool strip = colsum_is_strip(rows, cols) && (ldx & 3) == 0 && (((uintptr_t)x) & 15) == 0 && tmp;
    if (colsum_is_tall(rows, cols) && (ldx & 3) == 0 && (((uintptr_t)x) & 15) == 0 && tmp) {
        const int nb = colsum_tall_blocks(rows, cols);
        float* part = tmp + (size_t)kSplits * cols;
        hipLaunchKernelGGL(colsum_tall_kernel, dim3(nb), dim3(256), 0, st, x, rows, cols, ldx, part);
        x = part; rows = nb; ldx = cols;
    }
    if (strip) {
        const int rps = asr_cdiv(rows, colsum_strip_splits(rows, cols));
        const int ns = asr_cdiv(rows, rps);
        hipLaunchKernelGGL(colsum_strip_kernel, dim3(asr_cdiv(cols, 64), ns), dim3(256), 0, st, x, rows, cols, ldx, rps, tmp);
        hipLaunchKernelGGL(colsum_pass_kernel, dim3(gx, 1), dim3(threads), 0, st, (const float*)tmp, ns, cols, (long)cols, ns, out, (long)cols);
    } else if (rows <= kSplits) {
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


// ---- several of those reductions in TWO launches (asr_colsum_multi_batch): the producers leave their block partials where they
// are, the caller lists them in a device table and reduces all of them at the end of the backward pass (gradients of affine
// parameters are needed by the optimiser only).  Per item exactly the arithmetic of colsum_multi: level 1 folds rows_per_split
// rows per split (four chains in fixed order; one row per split when rows <= kSplits, i.e. a copy), level 2 the splits.
struct BatchItem {
    const float* x; float* tmp;          // partials [rows][ld]; tmp: kSplits * cols floats
    int rows, ld, nseg;
    int width[4];
    float* out[4];
};

static __global__ void colsum_batch_kernel(const BatchItem* __restrict__ items, int level) {
    const BatchItem it = items[blockIdx.z];
    int cols = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) cols += i < it.nseg ? it.width[i] : 0;
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= cols) return;
    const int rps = it.rows > kSplits ? (it.rows + kSplits - 1) / kSplits : 1;
    const int ns = (it.rows + rps - 1) / rps;
    const float* x; long ldx; int r0, r1;
    if (level == 1) {
        if ((int)blockIdx.y >= ns) return;
        x = it.x; ldx = it.ld; r0 = blockIdx.y * rps; r1 = r0 + rps < it.rows ? r0 + rps : it.rows;
    } else {
        x = it.tmp; ldx = cols; r0 = 0; r1 = ns;
    }
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
    if (level == 1) { it.tmp[(long)blockIdx.y * cols + c] = v; return; }
    int cc = c;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (i < it.nseg) {
            if (cc < it.width[i]) { it.out[i][cc] = v; return; }
            cc -= it.width[i];
        }
    }
}

inline int colsum_multi_batch(const BatchItem* items_dev, int n_items, int max_cols, hipStream_t st) {
    const int gx = asr_cdiv(max_cols, 256);
    hipLaunchKernelGGL(colsum_batch_kernel, dim3(gx, kSplits, n_items), dim3(256), 0, st, items_dev, 1);
    hipLaunchKernelGGL(colsum_batch_kernel, dim3(gx, 1, n_items), dim3(256), 0, st, items_dev, 2);
    ASR_CHECK_LAUNCH("colsum_multi_batch");
    return ASR_OK;
}

}  // namespace asr_reduce

// Transformer pieces of end2end/transformer.py on gfx950:
//   * (the fused multi-head attention kernels live in attention.hip);
//   * (residual add +) LayerNorm forward / backward (eps 1e-8, biased variance);
//   * embedding gather (+ sqrt(d) scale, zero_pad row, learned position table) and its
//     deterministic scatter (sorted index lists from the host: no float atomics);
//   * label-smoothed softmax cross-entropy with the reference's masking (-1 targets count).
// The projection / FFN / vocabulary GEMMs are the 1-tap tap_gemm / tap_wgrad kernels.
#include "asr_common.h"
#include "reduce.h"
#include "attn_common.h"
#include <math.h>

namespace {
// ------------------------------------------------------------------ (add +) LayerNorm
// one wave per row; y = gamma * (x - mean) / sqrt(var + eps) + beta with x = a (+ b)
__global__ __launch_bounds__(256) void add_ln_fwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         int rows, int C, float eps, float* __restrict__ y,
                                                         float* __restrict__ xhat, float* __restrict__ rstd) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const float* pa = a + (long)row * C;
    const float* pb = b ? b + (long)row * C : nullptr;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += pa[c] + (pb ? pb[c] : 0.f);
    const float mean = asr_wave_sum(s) / (float)C;
    float v = 0.f;
    for (int c = lane; c < C; c += 64) { const float d = pa[c] + (pb ? pb[c] : 0.f) - mean; v += d * d; }
    const float var = asr_wave_sum(v) / (float)C;
    const float rs = 1.f / sqrtf(var + eps);
    if (lane == 0) rstd[row] = rs;
    for (int c = lane; c < C; c += 64) {
        const float xh = (pa[c] + (pb ? pb[c] : 0.f) - mean) * rs;
        xhat[(long)row * C + c] = xh;
        y[(long)row * C + c] = gamma[c] * xh + beta[c];
    }
}

// float4 version: the row (a + b) stays in registers, one read of the inputs.  A lane owns float4 columns lane + 64*i.
template <int NV>
__global__ __launch_bounds__(256) void add_ln_fwd_vec_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             int rows, int C, float eps, float* __restrict__ y,
                                                             float* __restrict__ xhat, float* __restrict__ rstd,
                                                             uint32_t dthr, uint32_t dseed, float dscale) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63, n4 = C >> 2;
    const float* pa = a + (long)row * C;
    const float* pb = b ? b + (long)row * C : nullptr;
    float4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int q = lane + i * 64;
        v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (q < n4) {
            v[i] = *(const float4*)(pa + q * 4);
            if (dthr) {        // asr_add_layernorm_fwd_dropout: a is dropped with asr_dropout's mask (element row * C + col) first
                const uint32_t e = (uint32_t)row * (uint32_t)C + (uint32_t)(q * 4);
                v[i].x = drop_keep(e, dseed, dthr) ? v[i].x * dscale : 0.f;
                v[i].y = drop_keep(e + 1, dseed, dthr) ? v[i].y * dscale : 0.f;
                v[i].z = drop_keep(e + 2, dseed, dthr) ? v[i].z * dscale : 0.f;
                v[i].w = drop_keep(e + 3, dseed, dthr) ? v[i].w * dscale : 0.f;
            }
            if (pb) { const float4 w = *(const float4*)(pb + q * 4); v[i].x += w.x; v[i].y += w.y; v[i].z += w.z; v[i].w += w.w; }
        }
        s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    const float mean = asr_wave_sum(s) / (float)C;
    float var = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        if (lane + i * 64 < n4) {
            v[i].x -= mean; v[i].y -= mean; v[i].z -= mean; v[i].w -= mean;
            var += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
        }
    }
    var = asr_wave_sum(var) / (float)C;
    const float rs = 1.f / sqrtf(var + eps);
    if (lane == 0) rstd[row] = rs;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int q = lane + i * 64;
        if (q >= n4) continue;
        const float4 xh = make_float4(v[i].x * rs, v[i].y * rs, v[i].z * rs, v[i].w * rs);
        const float4 g = *(const float4*)(gamma + q * 4), be = *(const float4*)(beta + q * 4);
        *(float4*)(xhat + (long)row * C + q * 4) = xh;
        *(float4*)(y + (long)row * C + q * 4) = make_float4(g.x * xh.x + be.x, g.y * xh.y + be.y, g.z * xh.z + be.z, g.w * xh.w + be.w);
    }
}

// rows per block in the backward (block partials for dgamma / dbeta): 64, or 16 when that leaves fewer than 512 blocks (the
// language model's 6400 rows were 100 blocks of sixteen rows per wave on 256 CUs: 27 us for 39 MB of traffic)
static inline int ln_rows_per_block(int rows) { return rows >= 64 * 512 ? 64 : 16; }

// dx = rstd * (g - mean(g) - xhat * mean(g*xhat)),  g = dy*gamma;  partials[blk][2][C] = sum dy*xhat, sum dy
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ xhat,
                                                     const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                     int rows, int C, float* __restrict__ dx, int accumulate,
                                                     float* __restrict__ partials, int rpb) {
    extern __shared__ float sm[];        // [4 waves][2][C]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float* mine = sm + wave * 2 * C;
    for (int c = lane; c < 2 * C; c += 64) mine[c] = 0.f;
    const int r0 = blockIdx.x * rpb;
    for (int rr = wave; rr < rpb; rr += 4) {
        const int row = r0 + rr;
        if (row >= rows) break;
        const float* pdy = dy + (long)row * C;
        const float* pxh = xhat + (long)row * C;
        float s1 = 0.f, s2 = 0.f;
        for (int c = lane; c < C; c += 64) { const float g = pdy[c] * gamma[c]; s1 += g; s2 += g * pxh[c]; }
        s1 = asr_wave_sum(s1) / (float)C;
        s2 = asr_wave_sum(s2) / (float)C;
        const float rs = rstd[row];
        for (int c = lane; c < C; c += 64) {
            const float d = pdy[c], xh = pxh[c];
            float v = rs * (d * gamma[c] - s1 - xh * s2);
            float* o = dx + (long)row * C + c;
            if (accumulate) v += *o;
            *o = v;
            mine[c] += d * xh;
            mine[C + c] += d;
        }
    }
    __syncthreads();
    float* out = partials + (long)blockIdx.x * 2 * C;
    for (int c = threadIdx.x; c < 2 * C; c += 256) out[c] = (sm[c] + sm[2 * C + c]) + (sm[4 * C + c] + sm[6 * C + c]);
}

// The same pass with float4 loads, the row kept in registers (one read of dy / xhat) and the dgamma / dbeta partial
// sums of a wave in registers instead of read-modify-write LDS traffic (120 -> ~45 us at 32768 x 512).
// A lane owns float4 columns lane + 64*i, i < NV (C <= 1024*NV/4... i.e. C <= 256*NV).
// Fused consumers of dx (asr_layernorm_bwd_fused): dx2 (+)= dx is the residual fan-in that used to be an axpy of its own, dz =
// (z > 0) ? dx * zscale : 0 the backward of the Dense(relu) [+ in-place dropout] that produced the normalised sum's other
// operand; dx itself is optional then.  One read of dy / xhat instead of writing dx and reading it back twice.
struct LnBwdExtra { float* dx2; int acc2; const float* z; float zscale; float* dz; uint32_t dthr, dseed; };

template <int NV>
__global__ __launch_bounds__(256) void ln_bwd_vec_kernel(const float* __restrict__ dy, const float* __restrict__ xhat,
                                                         const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                         int rows, int C, float* __restrict__ dx, int accumulate,
                                                         float* __restrict__ partials, int rpb, LnBwdExtra ex) {
    extern __shared__ float sm[];        // [4 waves][2][C]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n4 = C >> 2;
    float4 g4[NV], ag[NV], ab[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int q = lane + i * 64;
        g4[i] = (q < n4) ? *(const float4*)(gamma + q * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        ag[i] = make_float4(0.f, 0.f, 0.f, 0.f); ab[i] = ag[i];
    }
    const int r0 = blockIdx.x * rpb;
    for (int rr = wave; rr < rpb; rr += 4) {
        const int row = r0 + rr;
        if (row >= rows) break;
        const float* pdy = dy + (long)row * C;
        const float* pxh = xhat + (long)row * C;
        float4 d[NV], xh[NV];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int q = lane + i * 64;
            d[i] = (q < n4) ? *(const float4*)(pdy + q * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            xh[i] = (q < n4) ? *(const float4*)(pxh + q * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 gd = make_float4(d[i].x * g4[i].x, d[i].y * g4[i].y, d[i].z * g4[i].z, d[i].w * g4[i].w);
            s1 += (gd.x + gd.y) + (gd.z + gd.w);
            s2 += (gd.x * xh[i].x + gd.y * xh[i].y) + (gd.z * xh[i].z + gd.w * xh[i].w);
        }
        s1 = asr_wave_sum(s1) / (float)C;
        s2 = asr_wave_sum(s2) / (float)C;
        const float rs = rstd[row];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int q = lane + i * 64;
            if (q >= n4) continue;
            const float4 v0 = make_float4(rs * (d[i].x * g4[i].x - s1 - xh[i].x * s2), rs * (d[i].y * g4[i].y - s1 - xh[i].y * s2),
                                          rs * (d[i].z * g4[i].z - s1 - xh[i].z * s2), rs * (d[i].w * g4[i].w - s1 - xh[i].w * s2));
            const long off = (long)row * C + q * 4;
            if (dx) {
                float4 v = v0;
                if (accumulate) { const float4 p = *(const float4*)(dx + off); v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w; }
                *(float4*)(dx + off) = v;
            }
            if (ex.dx2) {
                float4 v = v0;
                if (ex.acc2) { const float4 p = *(const float4*)(ex.dx2 + off); v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w; }
                *(float4*)(ex.dx2 + off) = v;
            }
            if (ex.dz) {
                float4 a = make_float4(1.f, 1.f, 1.f, 1.f);
                if (ex.z) a = *(const float4*)(ex.z + off);
                if (ex.dthr) {
                    const uint32_t e = (uint32_t)off;
                    a.x = drop_keep(e, ex.dseed, ex.dthr) ? a.x : 0.f; a.y = drop_keep(e + 1, ex.dseed, ex.dthr) ? a.y : 0.f;
                    a.z = drop_keep(e + 2, ex.dseed, ex.dthr) ? a.z : 0.f; a.w = drop_keep(e + 3, ex.dseed, ex.dthr) ? a.w : 0.f;
                }
                *(float4*)(ex.dz + off) = make_float4(a.x > 0.f ? v0.x * ex.zscale : 0.f, a.y > 0.f ? v0.y * ex.zscale : 0.f,
                                                      a.z > 0.f ? v0.z * ex.zscale : 0.f, a.w > 0.f ? v0.w * ex.zscale : 0.f);
            }
            ag[i].x += d[i].x * xh[i].x; ag[i].y += d[i].y * xh[i].y; ag[i].z += d[i].z * xh[i].z; ag[i].w += d[i].w * xh[i].w;
            ab[i].x += d[i].x; ab[i].y += d[i].y; ab[i].z += d[i].z; ab[i].w += d[i].w;
        }
    }
    float* mine = sm + wave * 2 * C;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int q = lane + i * 64;
        if (q < n4) { *(float4*)(mine + q * 4) = ag[i]; *(float4*)(mine + C + q * 4) = ab[i]; }
    }
    __syncthreads();
    float* out = partials + (long)blockIdx.x * 2 * C;
    for (int c = threadIdx.x; c < 2 * C; c += 256) out[c] = (sm[c] + sm[2 * C + c]) + (sm[4 * C + c] + sm[6 * C + c]);
}

// ------------------------------------------------------------------ embedding
__global__ void embed_fwd_kernel(const float* __restrict__ table, const int32_t* __restrict__ ids,
                                 const float* __restrict__ pos, int N, int T, int C, int zero_pad, float scale,
                                 float* __restrict__ out) {
    const int C4 = C >> 2;
    const long total = (long)N * T * C4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        const long row = i / C4;
        const int t = (int)(row % T);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (table) {
            const int id = ids[row];
            if (!(zero_pad && id == 0)) v = *(const float4*)(table + (long)id * C + c4 * 4);
            v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
        }
        if (pos) {
            const float4 p = *(const float4*)(pos + (long)t * C + c4 * 4);
            v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
        }
        *(float4*)(out + row * C + c4 * 4) = v;
    }
}

// dtable[uniq[u]] = scale * sum over the (sorted, fixed order) positions of that id
__global__ __launch_bounds__(256) void embed_bwd_kernel(const float* __restrict__ dout, const int32_t* __restrict__ perm,
                                                        const int32_t* __restrict__ uniq, const int32_t* __restrict__ seg,
                                                        int C, int zero_pad, float scale, float* __restrict__ dtable) {
    const int u = blockIdx.x;
    const int id = uniq[u];
    const int beg = seg[u], end = seg[u + 1];
    for (int c = threadIdx.x; c < C; c += 256) {
        float s = 0.f;
        for (int j = beg; j < end; ++j) s += dout[(long)perm[j] * C + c];
        dtable[(long)id * C + c] = (zero_pad && id == 0) ? 0.f : s * scale;
    }
}

// The same sums with no host-side sort: one wave per table row v scans the id array 64 positions at a time (a ballot of
// ids[p] == v) and adds the matching rows of dout in ascending position -- the order of a stable sort by id, so the bits are
// those of embed_bwd_kernel.  V * rows / 64 wave iterations over an L2-resident array: microseconds for the models' tables
// (1536 x 6400, 6348 x 32768), against an argsort + unique on the host and three uploads per step.
__global__ __launch_bounds__(256) void embed_bwd_ids_kernel(const float* __restrict__ dout, const int32_t* __restrict__ ids, int rows,
                                                            int V, int C, int zero_pad, float scale, float* __restrict__ dtable) {
    const int v = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (v >= V) return;
    const int lane = threadIdx.x & 63;
    const int n4 = C >> 2;                       // float4 columns: lane owns lane, lane + 64, ... (C <= 1024)
    float4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    bool any = false;
    for (int p0 = 0; p0 < rows; p0 += 256) {              // four independent id loads in flight: the scan is load-latency bound
        int idv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int p = p0 + u * 64 + lane; idv[u] = p < rows ? ids[p] : -1; }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            unsigned long long m = __ballot(idv[u] == v);
            any |= m != 0;
            while (m) {
                const int j = __builtin_ctzll(m);
                m &= m - 1;
                const float* src = dout + (long)(p0 + u * 64 + j) * C;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int q = lane + i * 64;
                    if (q < n4) {
                        const float4 t = *(const float4*)(src + q * 4);
                        acc[i].x += t.x; acc[i].y += t.y; acc[i].z += t.z; acc[i].w += t.w;
                    }
                }
            }
        }
    }
    if (!any) return;                            // an id that does not occur: its row is not written
    const bool zero = zero_pad && v == 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = lane + i * 64;
        if (q < n4) *(float4*)(dtable + (long)v * C + q * 4) = zero ? make_float4(0.f, 0.f, 0.f, 0.f)
                                                                     : make_float4(acc[i].x * scale, acc[i].y * scale, acc[i].z * scale, acc[i].w * scale);
    }
}

// ------------------------------------------------------------------ label-smoothed CE
// one wave per row.  loss = (sum ys)*lse - sum ys*logit, ys = (1-eps)*onehot + eps/V (onehot = 0 for ids outside
// [0,V)); dlogits = ((sum ys)*softmax - ys) * w with w = (target != pad) * inv_count.  stats row: [loss*ist, correct*ist]
__global__ __launch_bounds__(256) void smoothed_ce_kernel(const float* __restrict__ logits, int ld, const int32_t* __restrict__ target,
                                                          int rows, int V, float eps, int pad_id, float inv_count,
                                                          float* __restrict__ loss_rows, int32_t* __restrict__ preds,
                                                          float* __restrict__ stats, float* __restrict__ dlogits) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const float* x = logits + (long)row * ld;
    const int tg = target[row];
    float m = -INFINITY; int am = 0x7fffffff;
    float sumx = 0.f;
    for (int k = lane; k < V; k += 64) {
        const float v = x[k];
        sumx += v;
        if (v > m || am == 0x7fffffff) { m = v; am = k; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(m, o, 64);
        const int ok = __shfl_xor(am, o, 64);
        if (ov > m || (ov == m && ok < am)) { m = ov; am = ok; }
    }
    sumx = asr_wave_sum(sumx);
    float se = 0.f;
    for (int k = lane; k < V; k += 64) se += expf(x[k] - m);
    se = asr_wave_sum(se);
    const float lse = m + logf(se);
    const bool valid = tg >= 0 && tg < V;
    const float ysum = (valid ? (1.f - eps) : 0.f) + eps;
    const float xt = valid ? x[tg] : 0.f;
    const float loss = ysum * lse - (valid ? (1.f - eps) * xt : 0.f) - (eps / (float)V) * sumx;
    const float ist = (tg != pad_id) ? 1.f : 0.f;
    if (lane == 0) {
        loss_rows[row] = loss;
        preds[row] = am;
        stats[(long)row * 2] = loss * ist;
        stats[(long)row * 2 + 1] = (am == tg) ? ist : 0.f;
    }
    if (dlogits) {
        const float w = ist * inv_count;
        float* d = dlogits + (long)row * ld;
        for (int k = lane; k < ld; k += 64) {
            float g = 0.f;
            if (k < V) {
                const float ys = eps / (float)V + ((valid && k == tg) ? (1.f - eps) : 0.f);
                g = (ysum * expf(x[k] - lse) - ys) * w;
            }
            d[k] = g;
        }
    }
}

// Same arithmetic with the whole row held in registers (TWO waves per row, NV float4 per lane): one HBM read of the
// logits and one write of the gradient instead of three passes of 4-byte loads.  One wave per row needs 26 float4 per
// lane at V = 6348, which leaves one wave per SIMD (256 VGPRs) and is slower than the loop version.
// Rows up to 128 * 4 * NV columns; the loop version above serves wider vocabularies.
template <int NV>
__global__ __launch_bounds__(256) void smoothed_ce_reg_kernel(const float* __restrict__ logits, int ld, const int32_t* __restrict__ target,
                                                              int rows, int V, float eps, int pad_id, float inv_count,
                                                              float* __restrict__ loss_rows, int32_t* __restrict__ preds,
                                                              float* __restrict__ stats, float* __restrict__ dlogits) {
    __shared__ float sh_m[4], sh_sx[4], sh_se[4];
    __shared__ int sh_am[4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, half = wave & 1;
    const int row = blockIdx.x * 2 + (wave >> 1);
    const bool active = row < rows;
    const float* x = logits + (long)(active ? row : 0) * ld;
    const int tg = active ? target[row] : 0;
    const int n4 = ld >> 2;
    float4 v[NV];
    float m = -INFINITY; int am = 0x7fffffff;
    float sumx = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int q = half * 64 + lane + i * 128;
        v[i] = (active && q < n4) ? *(const float4*)(x + q * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        const float e[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = q * 4 + j;
            if (k < V) {
                sumx += e[j];
                if (e[j] > m || am == 0x7fffffff) { m = e[j]; am = k; }      // k ascends within a lane: first maximum kept
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(m, o, 64);
        const int ok = __shfl_xor(am, o, 64);
        if (ov > m || (ov == m && ok < am)) { m = ov; am = ok; }
    }
    sumx = asr_wave_sum(sumx);
    if (lane == 0) { sh_m[wave] = m; sh_am[wave] = am; sh_sx[wave] = sumx; }
    __syncthreads();
    {
        const float om = sh_m[wave ^ 1]; const int oa = sh_am[wave ^ 1];
        if (om > m || (om == m && oa < am)) { m = om; am = oa; }
        sumx = sh_sx[wave & ~1] + sh_sx[wave | 1];
    }
    float se = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int k0 = (half * 64 + lane + i * 128) * 4;
        // keep exp(x - max) in place of x: the gradient needs it again
        v[i].x = (k0 + 0 < V) ? expf(v[i].x - m) : 0.f; v[i].y = (k0 + 1 < V) ? expf(v[i].y - m) : 0.f;
        v[i].z = (k0 + 2 < V) ? expf(v[i].z - m) : 0.f; v[i].w = (k0 + 3 < V) ? expf(v[i].w - m) : 0.f;
        se += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    se = asr_wave_sum(se);
    if (lane == 0) sh_se[wave] = se;
    __syncthreads();
    se = sh_se[wave & ~1] + sh_se[wave | 1];
    if (!active) return;
    const float lse = m + logf(se);
    const bool valid = tg >= 0 && tg < V;
    const float ysum = (valid ? (1.f - eps) : 0.f) + eps;
    const float xt = valid ? x[tg] : 0.f;
    const float loss = ysum * lse - (valid ? (1.f - eps) * xt : 0.f) - (eps / (float)V) * sumx;
    const float ist = (tg != pad_id) ? 1.f : 0.f;
    if (lane == 0 && half == 0) {
        loss_rows[row] = loss;
        preds[row] = am;
        stats[(long)row * 2] = loss * ist;
        stats[(long)row * 2 + 1] = (am == tg) ? ist : 0.f;
    }
    if (dlogits) {
        const float w = ist * inv_count;
        const float pscale = ysum / se;                 // ysum * exp(x - lse) = ysum * exp(x - m) / se
        const float ys0 = eps / (float)V;
        float* d = dlogits + (long)row * ld;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int q = half * 64 + lane + i * 128;
            if (q >= n4) continue;
            const int k0 = q * 4;
            float4 g;
            g.x = (k0 + 0 < V) ? (pscale * v[i].x - ys0 - ((valid && k0 + 0 == tg) ? (1.f - eps) : 0.f)) * w : 0.f;
            g.y = (k0 + 1 < V) ? (pscale * v[i].y - ys0 - ((valid && k0 + 1 == tg) ? (1.f - eps) : 0.f)) * w : 0.f;
            g.z = (k0 + 2 < V) ? (pscale * v[i].z - ys0 - ((valid && k0 + 2 == tg) ? (1.f - eps) : 0.f)) * w : 0.f;
            g.w = (k0 + 3 < V) ? (pscale * v[i].w - ys0 - ((valid && k0 + 3 == tg) ? (1.f - eps) : 0.f)) * w : 0.f;
            *(float4*)(d + k0) = g;
        }
    }
}

// y[i] = keep(i) ? x[i] / (1 - rate) : 0   (in place allowed; applied to a gradient with the same seed it is the backward)
__global__ void dropout_kernel(const float* __restrict__ x, size_t n, uint32_t thr, uint32_t seed, float scale, float* __restrict__ y) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        y[i] = drop_keep((uint32_t)i, seed, thr) ? x[i] * scale : 0.f;
}

inline int grid_for(long total, int threads) {
    long b = (total + threads - 1) / threads;
    return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

}  // namespace

// ===================================================================== C ABI
namespace {
__global__ void copy2d_kernel(float* __restrict__ dst, int ldd, const float* __restrict__ src, int lds, int rows, int c4n,
                              int accumulate) {
    const long total = (long)rows * c4n;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / c4n;
        const int c = (int)(i - r * c4n) * 4;
        float4 v = *(const float4*)(src + r * lds + c);
        float* d = dst + r * ldd + c;
        if (accumulate) { const float4 o = *(const float4*)d; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
        *(float4*)d = v;
    }
}
}  // namespace

extern "C" int asr_copy2d(float* dst, int ldd, const float* src, int lds, int rows, int cols, int accumulate, void* stream) {
    if (!dst || !src || rows < 1 || cols < 4 || (cols & 3) || (ldd & 3) || (lds & 3) || ldd < cols || lds < cols) return ASR_ERR_BAD_ARG;
    if (((uintptr_t)dst | (uintptr_t)src) & 15) return ASR_ERR_BAD_ARG;
    hipLaunchKernelGGL(copy2d_kernel, dim3(grid_for((long)rows * (cols / 4), 256)), dim3(256), 0, (hipStream_t)stream, dst, ldd, src,
                       lds, rows, cols / 4, accumulate);
    ASR_CHECK_LAUNCH("copy2d");
    return ASR_OK;
}

// several strided copies in ONE launch: the table (asr_copy2d_item, device memory) is built once by the caller
namespace {
__global__ void copy2d_batch_kernel(const asr_copy2d_item* __restrict__ items, int accumulate) {
    const asr_copy2d_item it = items[blockIdx.y];
    const int c4n = it.cols >> 2;
    const long total = (long)it.rows * c4n;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / c4n;
        const int c = (int)(i - r * c4n) * 4;
        float4 v = *(const float4*)(it.src + r * it.lds + c);
        float* d = it.dst + r * it.ldd + c;
        if (accumulate) { const float4 o = *(const float4*)d; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
        *(float4*)d = v;
    }
}
}  // namespace

extern "C" int asr_copy2d_batch(const asr_copy2d_item* items_dev, int n_items, int max_elems, int accumulate, void* stream) {
    if (!items_dev || n_items < 1 || n_items > 65535 || max_elems < 4) return ASR_ERR_BAD_ARG;
    hipLaunchKernelGGL(copy2d_batch_kernel, dim3(grid_for((long)max_elems / 4, 256), n_items), dim3(256), 0, (hipStream_t)stream,
                       items_dev, accumulate);
    ASR_CHECK_LAUNCH("copy2d_batch");
    return ASR_OK;
}

extern "C" int asr_dropout(const float* x, size_t n, float rate, unsigned int seed, float* y, void* stream) {
    if (!x || !y || n == 0 || rate < 0.f || rate >= 1.f || n >= 4294967296ull) return ASR_ERR_BAD_ARG;
    long b = (long)((n + 255) / 256);
    if (b > 16384) b = 16384;
    hipLaunchKernelGGL(dropout_kernel, dim3((int)b), dim3(256), 0, (hipStream_t)stream, x, n, drop_threshold(rate), (uint32_t)seed,
                       1.0f / (1.0f - rate), y);
    ASR_CHECK_LAUNCH("dropout");
    return ASR_OK;
}

static int add_ln_launch(const float* a, const float* b, const float* gamma, const float* beta, int rows, int C, float eps, float* y,
                         float* xhat, float* rstd, uint32_t dthr, uint32_t dseed, float dscale, hipStream_t st) {
    const bool vec = (C & 3) == 0 &&
        ((((uintptr_t)a | (uintptr_t)b | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)y | (uintptr_t)xhat)) & 15) == 0;
    const dim3 grid(asr_cdiv(rows, 4));
    if (vec && C <= 256) hipLaunchKernelGGL(add_ln_fwd_vec_kernel<1>, grid, dim3(256), 0, st, a, b, gamma, beta, rows, C, eps, y, xhat, rstd, dthr, dseed, dscale);
    else if (vec && C <= 512) hipLaunchKernelGGL(add_ln_fwd_vec_kernel<2>, grid, dim3(256), 0, st, a, b, gamma, beta, rows, C, eps, y, xhat, rstd, dthr, dseed, dscale);
    else if (vec && C <= 2048) hipLaunchKernelGGL(add_ln_fwd_vec_kernel<8>, grid, dim3(256), 0, st, a, b, gamma, beta, rows, C, eps, y, xhat, rstd, dthr, dseed, dscale);
    else if (dthr) return ASR_ERR_UNSUPPORTED;
    else hipLaunchKernelGGL(add_ln_fwd_kernel, grid, dim3(256), 0, st, a, b, gamma, beta, rows, C, eps, y, xhat, rstd);
    ASR_CHECK_LAUNCH("add_layernorm_fwd");
    return ASR_OK;
}

extern "C" int asr_add_layernorm_fwd(const float* a, const float* b, const float* gamma, const float* beta, int rows, int C,
                                     float eps, float* y, float* xhat, float* rstd, void* stream) {
    if (!a || !gamma || !beta || !y || !xhat || !rstd || rows < 1 || C < 1) return ASR_ERR_BAD_ARG;
    return add_ln_launch(a, b, gamma, beta, rows, C, eps, y, xhat, rstd, 0u, 0u, 1.f, (hipStream_t)stream);
}

extern "C" int asr_add_layernorm_fwd_dropout(const float* a, const float* b, const float* gamma, const float* beta, int rows, int C,
                                             float eps, float rate, unsigned seed, float* y, float* xhat, float* rstd, void* stream) {
    if (!a || !gamma || !beta || !y || !xhat || !rstd || rows < 1 || C < 1 || rate < 0.f || rate >= 1.f) return ASR_ERR_BAD_ARG;
    if ((unsigned long long)rows * (unsigned long long)C >= 4294967296ull) return ASR_ERR_BAD_ARG;
    const uint32_t thr = drop_threshold(rate);
    return add_ln_launch(a, b, gamma, beta, rows, C, eps, y, xhat, rstd, thr, (uint32_t)seed, thr ? 1.0f / (1.0f - rate) : 1.f, (hipStream_t)stream);
}

// Callers size ONE workspace for their largest row count and use it for smaller problems too, so the size must not shrink
// where the rows-per-block rule switches (a 32767-row problem has four times the blocks of a 32768-row one): the result covers
// every row count <= rows.
extern "C" size_t asr_layernorm_bwd_workspace(int rows, int C) {
    size_t nblk = (size_t)asr_cdiv(rows, ln_rows_per_block(rows));
    if (rows >= 64 * 512) { const size_t below = (size_t)asr_cdiv(64 * 512 - 1, ln_rows_per_block(64 * 512 - 1)); if (below > nblk) nblk = below; }
    return (nblk * 2 * C + asr_reduce::colsum_tmp_floats((int)nblk, 2 * C) + 16) * sizeof(float);
}

static int ln_bwd_launch(const float* dy, const float* xhat, const float* rstd, const float* gamma, int rows, int C, float* dx,
                         int accumulate, float* dgamma, float* dbeta, float* partials, const LnBwdExtra& ex, hipStream_t st) {
    if ((size_t)8 * C * sizeof(float) > 64 * 1024) return ASR_ERR_UNSUPPORTED;
    const int rpb = ln_rows_per_block(rows);
    const int nblk = asr_cdiv(rows, rpb);
    const size_t lds = (size_t)8 * C * sizeof(float);
    uintptr_t al = (uintptr_t)dy | (uintptr_t)xhat | (uintptr_t)dx | (uintptr_t)gamma | (uintptr_t)ex.dx2 | (uintptr_t)ex.z | (uintptr_t)ex.dz;
    const bool vec = (C & 3) == 0 && (al & 15) == 0;
    const bool fused = ex.dx2 || ex.dz;
    if (vec && C <= 256) hipLaunchKernelGGL(ln_bwd_vec_kernel<1>, dim3(nblk), dim3(256), lds, st, dy, xhat, rstd, gamma, rows, C, dx, accumulate, partials, rpb, ex);
    else if (vec && C <= 512) hipLaunchKernelGGL(ln_bwd_vec_kernel<2>, dim3(nblk), dim3(256), lds, st, dy, xhat, rstd, gamma, rows, C, dx, accumulate, partials, rpb, ex);
    else if (vec && C <= 2048) hipLaunchKernelGGL(ln_bwd_vec_kernel<8>, dim3(nblk), dim3(256), lds, st, dy, xhat, rstd, gamma, rows, C, dx, accumulate, partials, rpb, ex);
    else if (fused) return ASR_ERR_UNSUPPORTED;
    else hipLaunchKernelGGL(ln_bwd_kernel, dim3(nblk), dim3(256), lds, st, dy, xhat, rstd, gamma, rows, C, dx, accumulate, partials, rpb);
    ASR_CHECK_LAUNCH("layernorm_bwd");
    if (!dgamma) return ASR_OK;          // deferred: the block partials [nblk][2 C] stay in `partials` for asr_colsum_multi_batch
    float* tmp = partials + (size_t)nblk * 2 * C;
    asr_reduce::Multi m;
    m.nseg = 2; m.width[0] = C; m.width[1] = C; m.width[2] = 0; m.width[3] = 0;
    m.out[0] = dgamma; m.out[1] = dbeta; m.out[2] = nullptr; m.out[3] = nullptr;
    return asr_reduce::colsum_multi(partials, nblk, 2 * C, m, tmp, st);
}

extern "C" int asr_layernorm_bwd(const float* dy, const float* xhat, const float* rstd, const float* gamma, int rows, int C,
                                 float* dx, int accumulate, float* dgamma, float* dbeta, float* partials, void* stream) {
    if (!dy || !xhat || !rstd || !gamma || !dx || !dgamma || !dbeta || !partials || rows < 1 || C < 1) return ASR_ERR_BAD_ARG;
    LnBwdExtra ex; ex.dx2 = nullptr; ex.acc2 = 0; ex.z = nullptr; ex.zscale = 0.f; ex.dz = nullptr; ex.dthr = 0; ex.dseed = 0;
    return ln_bwd_launch(dy, xhat, rstd, gamma, rows, C, dx, accumulate, dgamma, dbeta, partials, ex, (hipStream_t)stream);
}

extern "C" int asr_layernorm_bwd_fused(const float* dy, const float* xhat, const float* rstd, const float* gamma, int rows, int C,
                                       float* dx, float* dx2, int accumulate2, const float* z, float drop_rate, unsigned drop_seed,
                                       float zscale, float* dz, float* dgamma, float* dbeta, float* partials, void* stream) {
    if (!dy || !xhat || !rstd || !gamma || (!dgamma != !dbeta) || !partials || rows < 1 || C < 1) return ASR_ERR_BAD_ARG;
    if ((!dx && !dx2 && !dz) || drop_rate < 0.f || drop_rate >= 1.f) return ASR_ERR_BAD_ARG;
    if ((unsigned long long)rows * (unsigned long long)C >= 4294967296ull) return ASR_ERR_BAD_ARG;
    LnBwdExtra ex; ex.dx2 = dx2; ex.acc2 = accumulate2; ex.z = z; ex.zscale = zscale; ex.dz = dz;
    ex.dthr = drop_threshold(drop_rate); ex.dseed = (uint32_t)drop_seed;
    return ln_bwd_launch(dy, xhat, rstd, gamma, rows, C, dx, 0, dgamma, dbeta, partials, ex, (hipStream_t)stream);
}

extern "C" int asr_layernorm_bwd_blocks(int rows) { return rows < 1 ? 0 : asr_cdiv(rows, ln_rows_per_block(rows)); }

static_assert(sizeof(asr_reduce_item) == sizeof(asr_reduce::BatchItem), "asr_reduce_item is the kernel's table entry");

extern "C" int asr_colsum_multi_batch(const asr_reduce_item* items_dev, int n_items, int max_cols, void* stream) {
    if (!items_dev || n_items < 1 || n_items > 65535 || max_cols < 1) return ASR_ERR_BAD_ARG;
    return asr_reduce::colsum_multi_batch((const asr_reduce::BatchItem*)items_dev, n_items, max_cols, (hipStream_t)stream);
}

extern "C" int asr_embed_fwd(const float* table, const int32_t* ids, const float* pos, int N, int T, int C,
                             int zero_pad, float scale, float* out, void* stream) {
    if ((!table && !pos) || (table && !ids) || !out || (C & 3) || N < 1 || T < 1) return ASR_ERR_BAD_ARG;
    hipLaunchKernelGGL(embed_fwd_kernel, dim3(grid_for((long)N * T * (C / 4), 256)), dim3(256), 0, (hipStream_t)stream, table, ids, pos, N, T, C, zero_pad, scale, out);
    ASR_CHECK_LAUNCH("embed_fwd");
    return ASR_OK;
}

extern "C" int asr_embed_bwd(const float* dout, const int32_t* perm, const int32_t* uniq, const int32_t* seg, int n_uniq,
                             int C, int zero_pad, float scale, float* dtable, void* stream) {
    if (!dout || !perm || !uniq || !seg || !dtable || n_uniq < 0) return ASR_ERR_BAD_ARG;
    if (n_uniq == 0) return ASR_OK;
    hipLaunchKernelGGL(embed_bwd_kernel, dim3(n_uniq), dim3(256), 0, (hipStream_t)stream, dout, perm, uniq, seg, C, zero_pad, scale, dtable);
    ASR_CHECK_LAUNCH("embed_bwd");
    return ASR_OK;
}

extern "C" int asr_embed_bwd_ids(const float* dout, const int32_t* ids, int rows, int V, int C, int zero_pad, float scale,
                                 float* dtable, void* stream) {
    if (!dout || !ids || !dtable || rows < 1 || V < 1 || C < 4) return ASR_ERR_BAD_ARG;
    if ((C & 3) || C > 1024 || (((uintptr_t)dout | (uintptr_t)dtable) & 15)) return ASR_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(embed_bwd_ids_kernel, dim3(asr_cdiv(V, 4)), dim3(256), 0, (hipStream_t)stream, dout, ids, rows, V, C, zero_pad, scale, dtable);
    ASR_CHECK_LAUNCH("embed_bwd_ids");
    return ASR_OK;
}

extern "C" int asr_smoothed_ce(const float* logits, int ld, const int32_t* target, int rows, int V, float eps, int pad_id,
                               float inv_count, float* loss_rows, int32_t* preds, float* stats, float* dlogits, void* stream) {
    if (!logits || !target || !loss_rows || !preds || !stats || rows < 1 || V < 1 || ld < V) return ASR_ERR_BAD_ARG;
    const bool aligned = (ld & 3) == 0 && (((uintptr_t)logits | (uintptr_t)dlogits) & 15) == 0;
    if (aligned && ld <= 128 * 4 * 4)
        hipLaunchKernelGGL(smoothed_ce_reg_kernel<4>, dim3(asr_cdiv(rows, 2)), dim3(256), 0, (hipStream_t)stream, logits, ld, target, rows, V, eps, pad_id, inv_count, loss_rows, preds, stats, dlogits);
    else if (aligned && ld <= 128 * 4 * 13)
        hipLaunchKernelGGL(smoothed_ce_reg_kernel<13>, dim3(asr_cdiv(rows, 2)), dim3(256), 0, (hipStream_t)stream, logits, ld, target, rows, V, eps, pad_id, inv_count, loss_rows, preds, stats, dlogits);
    else
        hipLaunchKernelGGL(smoothed_ce_kernel, dim3(asr_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, logits, ld, target, rows, V, eps, pad_id, inv_count, loss_rows, preds, stats, dlogits);
    ASR_CHECK_LAUNCH("smoothed_ce");
    return ASR_OK;
}

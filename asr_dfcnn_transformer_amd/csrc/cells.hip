// HBM-bound pieces of the DFCNN cells (everything that is not a contraction):
//   * first cell (Cin = 1): conv3x3 + bias + ReLU + frozen-BN affine + 2x2 pool in one pass,
//     backward recomputes the pre-pool activations from the spectrogram;
//   * pool forward, cell backward prologue (pool-bwd + BN-bwd + ReLU-bwd + channel sums);
//   * squeeze-excitation forward/backward;
// All channel reductions are block partials + a fixed-order column sum (reduce.h).
#include "asr_common.h"
#include "reduce.h"

namespace {

__device__ __forceinline__ bool interior(long p, int HPWP, int WP, int H, int W, int& b, int& hh, int& ww) {
    b = (int)(p / HPWP);
    const int r = (int)(p - (long)b * HPWP);
    hh = r / WP;
    ww = r - hh * WP;
    return hh >= 1 && hh <= H && ww >= 1 && ww <= W;
}

__device__ __forceinline__ float4 f4(float v) { return make_float4(v, v, v, v); }
__device__ __forceinline__ float4 ld4(const float* p) { return *(const float4*)p; }
__device__ __forceinline__ void st4(float* p, float4 v) { *(float4*)p = v; }
__device__ __forceinline__ float4 fma4(float4 a, float4 b, float4 c) {
    return make_float4(fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y), fmaf(a.z, b.z, c.z), fmaf(a.w, b.w, c.w));
}
__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 mul4(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }

// ------------------------------------------------------------------ first cell
// One lane owns a PAIR of output channels of one pooled pixel, so that the 36 multiply-adds of the four pre-pool positions
// are packed fp32 instructions (v_pk_fma_f32: the patch value is the splat operand).  This kernel is vector-ALU work and
// nothing else; beside the fp32 MFMA kernels of the other stream every instruction of it costs its full issue time
// (DESIGN section 4), so the backward also skips what max-pooling makes zero: only the winning position of a window
// carries gradient, its 3x3 patch is re-read from the staged rows (LDS, not the vector pipe) and nine multiply-adds per
// channel replace thirty-six.  (Adding the three zero terms left the sums bit-identical, so dropping them does too.)
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 splat2(float v) { return f2{v, v}; }
__device__ __forceinline__ f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }

template <int POOL, bool BWD>
__global__ __launch_bounds__(256) void cell1_kernel(const float* __restrict__ x, int B, int T, int F, int C,
                                                    const float* __restrict__ w, const float* __restrict__ bias,
                                                    const float* __restrict__ sc, const float* __restrict__ sh,
                                                    float* __restrict__ y, const float* __restrict__ dy,
                                                    float* __restrict__ partials, int RPB) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int FP = F + 2;
    const int H2 = T / 2, W2 = F / 2, WP2 = W2 + 1, HP2 = H2 + 1;     // smem: two buffers of [4][F+2]
    const int tid = threadIdx.x;
    const int CH = C >> 1;               // lanes per pixel
    const int c = 2 * (tid % CH), slot = tid / CH, nslots = 256 / CH;
    const int b = blockIdx.y;
    f2 wr[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) wr[k] = f2{w[k * C + c], w[k * C + c + 1]};
    const f2 bs = f2{bias[c], bias[c + 1]}, s = f2{sc[c], sc[c + 1]}, h = f2{sh[c], sh[c + 1]};
    f2 gsum[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) gsum[k] = splat2(0.f);

    const int h2beg = blockIdx.x * RPB;
    const int h2end = (h2beg + RPB < H2) ? h2beg + RPB : H2;
    // the four spectrogram rows of a pooled row are staged one pooled row ahead: fetched into registers before the
    // arithmetic of the current row, written to the other LDS buffer after it (one barrier per row, no exposed load)
    auto value = [&](int h2n, int i) {
        const int r = i / FP, col = i - r * FP - 1;
        const int row = 2 * h2n - 1 + r;
        return (row >= 0 && row < T && col >= 0 && col < F) ? x[((long)b * T + row) * F + col] : 0.f;
    };
    auto fetch = [&](int h2n, float (&nv)[4]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { const int i = tid + 256 * j; nv[j] = (i < 4 * FP) ? value(h2n, i) : 0.f; }
    };
    auto put = [&](int h2n, float* dst, const float (&nv)[4]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { const int i = tid + 256 * j; if (i < 4 * FP) dst[i] = nv[j]; }
        for (int i = tid + 1024; i < 4 * FP; i += 256) dst[i] = value(h2n, i);     // F > 254 only
    };
    float nv[4];
    if (h2beg < h2end) { fetch(h2beg, nv); put(h2beg, smem, nv); }
    __syncthreads();
    int cur = 0;
    for (int h2 = h2beg; h2 < h2end; ++h2, cur ^= 1) {
        const float* xs = smem + cur * 4 * FP;
        const bool more = h2 + 1 < h2end;
        if (more) fetch(h2 + 1, nv);
        for (int w2 = slot; w2 < W2; w2 += nslots) {
            float p[4][4];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int q = 0; q < 4; ++q) p[r][q] = xs[r * FP + 2 * w2 + q];
            f2 a[4], yv[4];
#pragma unroll
            for (int pos = 0; pos < 4; ++pos) {
                const int dyy = pos >> 1, dxx = pos & 1;
                f2 z = bs;
#pragma unroll
                for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) z = fma2(splat2(p[dyy + kh][dxx + kw]), wr[kh * 3 + kw], z);
                a[pos] = f2{fmaxf(z.x, 0.f), fmaxf(z.y, 0.f)};
                yv[pos] = fma2(s, a[pos], h);
            }
            const long po = (((long)b * HP2 + h2 + 1) * WP2 + w2 + 1) * C + c;
            if (!BWD) {
                f2 o;
                if (POOL == 1) o = 0.25f * ((yv[0] + yv[1]) + (yv[2] + yv[3]));
                else o = f2{fmaxf(fmaxf(yv[0].x, yv[1].x), fmaxf(yv[2].x, yv[3].x)), fmaxf(fmaxf(yv[0].y, yv[1].y), fmaxf(yv[2].y, yv[3].y))};
                *(f2*)(y + po) = o;
            } else if (POOL == 1) {
                const f2 gy = 0.25f * *(const f2*)(dy + po);
                const f2 gs = gy * s;
#pragma unroll
                for (int pos = 0; pos < 4; ++pos) {
                    const int dyy = pos >> 1, dxx = pos & 1;
                    gsum[11] += gy;                          // dshift
                    gsum[10] = fma2(gy, a[pos], gsum[10]);   // dscale
                    const f2 dz = f2{a[pos].x > 0.f ? gs.x : 0.f, a[pos].y > 0.f ? gs.y : 0.f};
                    gsum[9] += dz;                           // dbias
#pragma unroll
                    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                        for (int kw = 0; kw < 3; ++kw)
                            gsum[kh * 3 + kw] = fma2(splat2(p[dyy + kh][dxx + kw]), dz, gsum[kh * 3 + kw]);
                }
            } else {
                const f2 gy = *(const f2*)(dy + po);
                // the first maximum of the window (strict >, as the forward's max and the oracle's argmax pick it)
                float m0 = yv[0].x, m1 = yv[0].y, a0 = a[0].x, a1 = a[0].y;
                int q0 = 0, q1 = 0;
#pragma unroll
                for (int pos = 1; pos < 4; ++pos) {
                    const int off = (pos >> 1) * FP + (pos & 1);
                    if (yv[pos].x > m0) { m0 = yv[pos].x; a0 = a[pos].x; q0 = off; }
                    if (yv[pos].y > m1) { m1 = yv[pos].y; a1 = a[pos].y; q1 = off; }
                }
                gsum[11] += gy;
                gsum[10] = fma2(gy, f2{a0, a1}, gsum[10]);
                const f2 gs = gy * s;
                const f2 dz = f2{a0 > 0.f ? gs.x : 0.f, a1 > 0.f ? gs.y : 0.f};
                gsum[9] += dz;
                const float* p0 = xs + 2 * w2 + q0;
                const float* p1 = xs + 2 * w2 + q1;
#pragma unroll
                for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw)
                        gsum[kh * 3 + kw] = fma2(f2{p0[kh * FP + kw], p1[kh * FP + kw]}, dz, gsum[kh * 3 + kw]);
            }
        }
        if (more) put(h2 + 1, smem + (cur ^ 1) * 4 * FP, nv);
        __syncthreads();
    }
    if (BWD) {
        // fewer than 64 lanes per pixel: the slots of a wave meet by shuffles first, so the scratch is one row set per
        // wave (6 KB at C = 32) -- small enough to run beside a one-workgroup-per-CU weight-gradient kernel that holds
        // 150 KB of the CU's LDS
        const bool fold = CH < 64;       // 256 % CH == 0, so CH is a power of two
        if (fold) {
            for (int off = CH; off < 64; off <<= 1)
#pragma unroll
                for (int k = 0; k < 12; ++k) {
                    gsum[k].x += __shfl_xor(gsum[k].x, off, 64);
                    gsum[k].y += __shfl_xor(gsum[k].y, off, 64);
                }
        }
        const int rslot = fold ? tid >> 6 : slot, rslots = fold ? 4 : nslots;
        __syncthreads();
        float* red = smem;               // [rslots][12][C]
        if (!fold || (tid & 63) < CH) {
#pragma unroll
            for (int k = 0; k < 12; ++k) *(f2*)(red + (rslot * 12 + k) * C + c) = gsum[k];
        }
        __syncthreads();
        float* out = partials + ((long)blockIdx.y * gridDim.x + blockIdx.x) * 12 * C;
        for (int i = tid; i < 12 * C; i += 256) {
            float sum = 0.f;
            for (int sl = 0; sl < rslots; ++sl) sum += red[sl * 12 * C + i];
            out[i] = sum;
        }
    }
}

constexpr int kCell1RPB = 8;

// ------------------------------------------------------------------ pool forward
template <int POOL>
__global__ void pool_fwd_kernel(const float* __restrict__ a, int B, int H, int W, int C,
                                const float* __restrict__ sc, const float* __restrict__ sh,
                                float* __restrict__ y) {
    const int C4 = C >> 2, H2 = H >> 1, W2 = W >> 1;
    const long total = (long)B * H2 * W2 * C4;
    const int WP = W + 1, HP = H + 1, WP2 = W2 + 1, HP2 = H2 + 1;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int cg = (int)(idx % C4);
        long pix = idx / C4;
        const int w2 = (int)(pix % W2); pix /= W2;
        const int h2 = (int)(pix % H2);
        const int b = (int)(pix / H2);
        const float4 s = ld4(sc + cg * 4), h = ld4(sh + cg * 4);
        const long pin = ((long)b * HP + 2 * h2 + 1) * WP + 2 * w2 + 1;
        const float4 v0 = fma4(s, ld4(a + pin * C + cg * 4), h);
        const float4 v1 = fma4(s, ld4(a + (pin + 1) * C + cg * 4), h);
        const float4 v2 = fma4(s, ld4(a + (pin + WP) * C + cg * 4), h);
        const float4 v3 = fma4(s, ld4(a + (pin + WP + 1) * C + cg * 4), h);
        float4 o;
        if (POOL == 1) {
            o = make_float4(0.25f * ((v0.x + v1.x) + (v2.x + v3.x)), 0.25f * ((v0.y + v1.y) + (v2.y + v3.y)),
                            0.25f * ((v0.z + v1.z) + (v2.z + v3.z)), 0.25f * ((v0.w + v1.w) + (v2.w + v3.w)));
        } else {
            o = make_float4(fmaxf(fmaxf(v0.x, v1.x), fmaxf(v2.x, v3.x)), fmaxf(fmaxf(v0.y, v1.y), fmaxf(v2.y, v3.y)),
                            fmaxf(fmaxf(v0.z, v1.z), fmaxf(v2.z, v3.z)), fmaxf(fmaxf(v0.w, v1.w), fmaxf(v2.w, v3.w)));
        }
        st4(y + (((long)b * HP2 + h2 + 1) * WP2 + w2 + 1) * C + cg * 4, o);
    }
}

// ------------------------------------------------------------------ cell backward prologue
__device__ __forceinline__ float pick_max_grad(float y0, float y1, float y2, float y3, int mine, float dp) {
    int arg = 0; float m = y0;
    if (y1 > m) { m = y1; arg = 1; }
    if (y2 > m) { m = y2; arg = 2; }
    if (y3 > m) { m = y3; arg = 3; }
    return arg == mine ? dp : 0.f;
}

// padded pixels per block: sized so that even the 200x25 layers launch >= ~1300 blocks
inline int pre_ppb(long NP) { return NP >= (4L << 20) ? 1024 : (NP >= (1L << 20) ? 512 : 128); }

template <int POOL>
__global__ __launch_bounds__(256) void cell_bwd_pre_kernel(const float* __restrict__ dy, int layout,
                                                           const float* __restrict__ a, int B, int H, int W, int C,
                                                           const float* __restrict__ sc, const float* __restrict__ sh,
                                                           float* __restrict__ dz, float* __restrict__ partials, int PPB) {
    __shared__ float red[256 * 12];
    const int C4 = C >> 2;
    const int tid = threadIdx.x;
    const int cg = tid % C4, slot = tid / C4, nslots = 256 / C4;
    const int WP = W + 1, HPWP = (H + 1) * WP;
    const long NP = (long)B * HPWP;
    const long pbeg = (long)blockIdx.x * PPB;
    const long pend = (pbeg + PPB < NP) ? pbeg + PPB : NP;
    const float4 s = ld4(sc + cg * 4), h = ld4(sh + cg * 4);
    float4 s_shift = f4(0.f), s_scale = f4(0.f), s_bias = f4(0.f);
    const int H2 = H >> 1, W2 = W >> 1;
    for (long p = pbeg + slot; p < pend; p += nslots) {
        int b, hh, ww;
        if (!interior(p, HPWP, WP, H, W, b, hh, ww)) continue;
        const float4 av = ld4(a + p * C + cg * 4);
        float4 g;
        if (POOL == 0) {
            g = (layout == 0) ? ld4(dy + p * C + cg * 4)
                              : ld4(dy + (((long)b * H + hh - 1) * W + ww - 1) * C + cg * 4);
        } else {
            const int hi = hh - 1, wi = ww - 1, h2 = hi >> 1, w2 = wi >> 1;
            if (h2 < H2 && w2 < W2) {
                const float4 dp = ld4(dy + (((long)b * (H2 + 1) + h2 + 1) * (W2 + 1) + w2 + 1) * C + cg * 4);
                if (POOL == 1) {
                    g = make_float4(0.25f * dp.x, 0.25f * dp.y, 0.25f * dp.z, 0.25f * dp.w);
                } else {
                    const long q = ((long)b * (H + 1) + 2 * h2 + 1) * WP + 2 * w2 + 1;
                    const float4 y0 = fma4(s, ld4(a + q * C + cg * 4), h);
                    const float4 y1 = fma4(s, ld4(a + (q + 1) * C + cg * 4), h);
                    const float4 y2 = fma4(s, ld4(a + (q + WP) * C + cg * 4), h);
                    const float4 y3 = fma4(s, ld4(a + (q + WP + 1) * C + cg * 4), h);
                    const int mine = (hi & 1) * 2 + (wi & 1);
                    g = make_float4(pick_max_grad(y0.x, y1.x, y2.x, y3.x, mine, dp.x),
                                    pick_max_grad(y0.y, y1.y, y2.y, y3.y, mine, dp.y),
                                    pick_max_grad(y0.z, y1.z, y2.z, y3.z, mine, dp.z),
                                    pick_max_grad(y0.w, y1.w, y2.w, y3.w, mine, dp.w));
                }
            } else {
                g = f4(0.f);
            }
        }
        s_shift = add4(s_shift, g);
        s_scale = fma4(g, av, s_scale);
        float4 d = mul4(g, s);
        d.x = av.x > 0.f ? d.x : 0.f; d.y = av.y > 0.f ? d.y : 0.f;
        d.z = av.z > 0.f ? d.z : 0.f; d.w = av.w > 0.f ? d.w : 0.f;
        s_bias = add4(s_bias, d);
        st4(dz + p * C + cg * 4, d);
    }
    // block reduction over the pixel slots: red[tid][12]
    float* mine = red + tid * 12;
    mine[0] = s_scale.x; mine[1] = s_scale.y; mine[2] = s_scale.z; mine[3] = s_scale.w;
    mine[4] = s_shift.x; mine[5] = s_shift.y; mine[6] = s_shift.z; mine[7] = s_shift.w;
    mine[8] = s_bias.x;  mine[9] = s_bias.y;  mine[10] = s_bias.z; mine[11] = s_bias.w;
    __syncthreads();
    float* out = partials + (long)blockIdx.x * 3 * C;
    for (int i = tid; i < 3 * C; i += 256) {
        const int which = i / C, c = i - which * C;
        const int g4 = c >> 2, e = c & 3;
        float sum = 0.f;
        for (int sl = 0; sl < nslots; ++sl) sum += red[(sl * C4 + g4) * 12 + which * 4 + e];
        out[i] = sum;
    }
}

// ------------------------------------------------------------------ squeeze-excitation
// One 16-byte load in flight per lane and 512 pixels per workgroup ON PURPOSE: a form with four unconditional loads per trip and
// plane-sized pixel ranges runs the small planes at 5 instead of 2.8 TB/s alone (se_reduce<0> 45 -> 31 us per launch), but in the
// two-stream step, beside the weight-gradient kernels of the other stream, it made the SE-DFCNN step 0.15 ms SLOWER (same box,
// three rounds: 14.43 -> 14.59 ms) -- the trickle hides under the matrix kernels, the burst does not.
constexpr int kSePPB = 512;

// partial[b][split][C] = sum over a pixel range of image b of x (MODE 0) or dout*(sc*x+sh) (MODE 1)
template <int MODE>
__global__ __launch_bounds__(256) void se_reduce_kernel(const float* __restrict__ x, const float* __restrict__ dout,
                                                        int H, int W, int C, const float* __restrict__ sc,
                                                        const float* __restrict__ sh, float* __restrict__ partials) {
    __shared__ float red[256 * 4];
    const int C4 = C >> 2;
    const int tid = threadIdx.x;
    const int cg = tid % C4, slot = tid / C4, nslots = 256 / C4;
    const int WP = W + 1, HPWP = (H + 1) * WP;
    const int b = blockIdx.y;
    const long base = (long)b * HPWP;
    const int pbeg = blockIdx.x * kSePPB;
    const int pend = (pbeg + kSePPB < HPWP) ? pbeg + kSePPB : HPWP;
    float4 acc = f4(0.f);
    float4 s = f4(1.f), h = f4(0.f);
    if (MODE == 1) { s = ld4(sc + cg * 4); h = ld4(sh + cg * 4); }
    for (int r = pbeg + slot; r < pend; r += nslots) {
        const int hh = r / WP, ww = r - hh * WP;
        if (hh < 1 || hh > H || ww < 1 || ww > W) continue;
        const float4 xv = ld4(x + (base + r) * C + cg * 4);
        if (MODE == 0) acc = add4(acc, xv);
        else acc = fma4(ld4(dout + (base + r) * C + cg * 4), fma4(s, xv, h), acc);
    }
    float* mine = red + tid * 4;
    mine[0] = acc.x; mine[1] = acc.y; mine[2] = acc.z; mine[3] = acc.w;
    __syncthreads();
    float* out = partials + ((long)b * gridDim.x + blockIdx.x) * C;
    for (int c = tid; c < C; c += 256) {
        float sum = 0.f;
        for (int sl = 0; sl < nslots; ++sl) sum += red[(sl * C4 + (c >> 2)) * 4 + (c & 3)];
        out[c] = sum;
    }
}

// fold of the split partials of image b, all 256 threads: thread (g, c) adds partials g, g + G, ... of channel c (G = 256 / C
// groups, four loads in flight), the groups are then added in the order g = 0 .. G-1 -- a fixed order, bitwise reproducible.
// Result in dst[C] (LDS); scratch [256] floats.  Ends with a barrier.
__device__ __forceinline__ void se_fold_partials(const float* __restrict__ part, int nsplit, int C, float* scratch, float* dst) {
    const int tid = threadIdx.x;
    if (C <= 256) {
        const int G = 256 / C, c = tid % C, g = tid / C;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        if (g < G) {
            int k = g;
            for (; k + 3 * G < nsplit; k += 4 * G) {
                const float v0 = part[(long)k * C + c], v1 = part[(long)(k + G) * C + c];
                const float v2 = part[(long)(k + 2 * G) * C + c], v3 = part[(long)(k + 3 * G) * C + c];
                a0 += v0; a1 += v1; a2 += v2; a3 += v3;
            }
            for (; k < nsplit; k += G) a0 += part[(long)k * C + c];
            scratch[g * C + c] = (a0 + a1) + (a2 + a3);
        }
        __syncthreads();
        if (tid < C) {
            float sum = 0.f;
            for (int gg = 0; gg < G; ++gg) sum += scratch[gg * C + tid];
            dst[tid] = sum;
        }
    } else {
        for (int c = tid; c < C; c += 256) {
            float sum = 0.f;
            for (int k = 0; k < nsplit; ++k) sum += part[(long)k * C + c];
            dst[c] = sum;
        }
    }
    __syncthreads();
}

// out[j] = sum_c in[c] * Wm[c * ldr + j * ldc]  for j < nout, contraction over nin, all 256 threads: thread (g, j) takes the
// terms c = g, g + G, ...; groups folded in fixed order.  scratch [256]; result via the callback on threads j < nout after
// the barrier.  (nout <= 256; otherwise one thread per output walks the whole contraction.)
template <class F>
__device__ __forceinline__ void se_matvec(const float* in, const float* __restrict__ Wm, int nin, int nout, long ldr, long ldc,
                                          float* scratch, F&& done) {
    const int tid = threadIdx.x;
    if (nout <= 256) {
        const int G = 256 / nout, j = tid % nout, g = tid / nout;
        if (g < G) {
            float a0 = 0.f, a1 = 0.f;
            int c = g;
            for (; c + G < nin; c += 2 * G) {
                a0 = fmaf(in[c], Wm[(long)c * ldr + (long)j * ldc], a0);
                a1 = fmaf(in[c + G], Wm[(long)(c + G) * ldr + (long)j * ldc], a1);
            }
            if (c < nin) a0 = fmaf(in[c], Wm[(long)c * ldr + (long)j * ldc], a0);
            scratch[g * nout + j] = a0 + a1;
        }
        __syncthreads();
        if (tid < nout) {
            float sum = 0.f;
            for (int gg = 0; gg < G; ++gg) sum += scratch[gg * nout + tid];
            done(tid, sum);
        }
    } else {
        for (int j = tid; j < nout; j += 256) {
            float sum = 0.f;
            for (int c = 0; c < nin; ++c) sum = fmaf(in[c], Wm[(long)c * ldr + (long)j * ldc], sum);
            done(j, sum);
        }
    }
    __syncthreads();
}

// one block per image: fold the split partials, run the 2-layer excitation MLP
__global__ __launch_bounds__(256) void se_excite_kernel(const float* __restrict__ partials, int nsplit, int H, int W,
                                                        int C, int hid, const float* __restrict__ sc,
                                                        const float* __restrict__ sh, const float* __restrict__ w1,
                                                        const float* __restrict__ b1, const float* __restrict__ w2,
                                                        const float* __restrict__ b2, float* __restrict__ st_s,
                                                        float* __restrict__ st_r, float* __restrict__ st_e) {
    extern __shared__ float sm[];
    float* s = sm;            // [C]
    float* r = sm + C;        // [hid]
    float* scratch = r + hid; // [256]
    const int b = blockIdx.x, tid = threadIdx.x;
    const float inv = 1.f / (float)(H * W);
    se_fold_partials(partials + (long)b * nsplit * C, nsplit, C, scratch, s);
    for (int c = tid; c < C; c += 256) {
        const float v = fmaf(sc[c], s[c] * inv, sh[c]);
        s[c] = v; st_s[(long)b * C + c] = v;
    }
    __syncthreads();
    se_matvec(s, w1, C, hid, hid, 1, scratch, [&](int j, float sum) {
        const float u = fmaxf(sum + b1[j], 0.f);
        r[j] = u; st_r[(long)b * hid + j] = u;
    });
    se_matvec(r, w2, hid, C, C, 1, scratch, [&](int c, float sum) {
        st_e[(long)b * C + c] = 1.f / (1.f + expf(-(sum + b2[c])));
    });
}

__global__ void se_apply_kernel(const float* __restrict__ main_in, const float* __restrict__ x, int B, int H, int W,
                                int C, const float* __restrict__ sc, const float* __restrict__ sh,
                                const float* __restrict__ e, float* __restrict__ out) {
    const int C4 = C >> 2, WP = W + 1, HPWP = (H + 1) * WP;
    const long total = (long)B * HPWP * C4;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int cg = (int)(idx % C4);
        const long p = idx / C4;
        int b, hh, ww;
        if (!interior(p, HPWP, WP, H, W, b, hh, ww)) continue;
        const float4 xt = fma4(ld4(sc + cg * 4), ld4(x + p * C + cg * 4), ld4(sh + cg * 4));
        st4(out + p * C + cg * 4, fma4(xt, ld4(e + (long)b * C + cg * 4), ld4(main_in + p * C + cg * 4)));
    }
}

// one block per image: backward of the excitation MLP.
// out_b (per image, later summed over b): [dw1 C*hid][db1 hid][dw2 hid*C][db2 C];  dsb[b][C] = ds/(H*W)
__global__ __launch_bounds__(256) void se_bwd_mlp_kernel(const float* __restrict__ partials, int nsplit, int H, int W,
                                                         int C, int hid, const float* __restrict__ w1,
                                                         const float* __restrict__ w2, const float* __restrict__ st_s,
                                                         const float* __restrict__ st_r, const float* __restrict__ st_e,
                                                         float* __restrict__ out_b, float* __restrict__ dsb) {
    extern __shared__ float sm[];
    float* dv = sm;            // [C]
    float* du = sm + C;        // [hid]
    float* scratch = du + hid; // [256]
    const int b = blockIdx.x, tid = threadIdx.x;
    const long per = (long)C * hid + hid + (long)hid * C + C;
    float* o = out_b + (long)b * per;
    float* o_dw1 = o; float* o_db1 = o + (long)C * hid; float* o_dw2 = o_db1 + hid; float* o_db2 = o_dw2 + (long)hid * C;
    se_fold_partials(partials + (long)b * nsplit * C, nsplit, C, scratch, dv);
    for (int c = tid; c < C; c += 256) {
        const float e = st_e[(long)b * C + c];
        const float v = dv[c] * e * (1.f - e);
        dv[c] = v; o_db2[c] = v;
    }
    __syncthreads();
    se_matvec(dv, w2, C, hid, 1, C, scratch, [&](int j, float dr) {
        const float rj = st_r[(long)b * hid + j];
        const float u = rj > 0.f ? dr : 0.f;
        du[j] = u; o_db1[j] = u;
    });
    for (int i = tid; i < hid * C; i += 256) {
        const int j = i / C, c = i - j * C;
        o_dw2[i] = st_r[(long)b * hid + j] * dv[c];
    }
    for (int i = tid; i < C * hid; i += 256) {
        const int c = i / hid, j = i - c * hid;
        o_dw1[i] = st_s[(long)b * C + c] * du[j];
    }
    const float inv = 1.f / (float)(H * W);
    se_matvec(du, w1, hid, C, 1, hid, scratch, [&](int c, float ds) { dsb[(long)b * C + c] = ds * inv; });
}

inline int se_apply_ppb(long NP) { return NP >= (4L << 20) ? 1024 : (NP >= (1L << 20) ? 512 : 128); }

// dxt = dout*e + dsb; dx = dxt*sc (+ dout when add_dout); channel sums dshift = sum dxt, dscale = sum dxt*x
// FUSE (asr_se_bwd_cell): x is the BN output of a conv cell that nothing else reads, so dx is that cell's complete output gradient
// and its backward prologue (cell_bwd_pre_kernel<0>: BN and ReLU backward + the cell's three channel sums) runs here on the value
// in the register -- dx is not written and read back: four planes of traffic instead of six.  Same blocks, slots and order as the
// two kernels it replaces: dZ and every partial sum bit for bit.
template <bool FUSE>
__global__ __launch_bounds__(256) void se_bwd_apply_kernel(const float* __restrict__ dout, const float* __restrict__ x,
                                                           int B, int H, int W, int C, const float* __restrict__ sc,
                                                           const float* __restrict__ e, const float* __restrict__ dsb,
                                                           int add_dout, float* __restrict__ dx,
                                                           float* __restrict__ partials, int PPB,
                                                           const float* __restrict__ cell_a, const float* __restrict__ cell_sc,
                                                           float* __restrict__ cell_dz, float* __restrict__ cell_partials) {
    __shared__ float red[256 * (FUSE ? 20 : 8)];
    const int C4 = C >> 2;
    const int tid = threadIdx.x;
    const int cg = tid % C4, slot = tid / C4, nslots = 256 / C4;
    const int WP = W + 1, HPWP = (H + 1) * WP;
    const long NP = (long)B * HPWP;
    const long pbeg = (long)blockIdx.x * PPB;
    const long pend = (pbeg + PPB < NP) ? pbeg + PPB : NP;
    const float4 s = ld4(sc + cg * 4);
    float4 s_shift = f4(0.f), s_scale = f4(0.f);
    float4 cs = f4(0.f), c_shift = f4(0.f), c_scale = f4(0.f), c_bias = f4(0.f);
    if (FUSE) cs = ld4(cell_sc + cg * 4);
    for (long p = pbeg + slot; p < pend; p += nslots) {
        int b, hh, ww;
        if (!interior(p, HPWP, WP, H, W, b, hh, ww)) continue;
        const float4 g = ld4(dout + p * C + cg * 4);
        const float4 xv = ld4(x + p * C + cg * 4);
        const float4 dxt = fma4(g, ld4(e + (long)b * C + cg * 4), ld4(dsb + (long)b * C + cg * 4));
        s_shift = add4(s_shift, dxt);
        s_scale = fma4(dxt, xv, s_scale);
        float4 d = mul4(dxt, s);
        if (add_dout) d = add4(d, g);
        if (!FUSE) {
            st4(dx + p * C + cg * 4, d);
        } else {                                         // cell_bwd_pre_kernel<0> on d = dL/dy of the cell
            const float4 av = ld4(cell_a + p * C + cg * 4);
            c_shift = add4(c_shift, d);
            c_scale = fma4(d, av, c_scale);
            float4 dd = mul4(d, cs);
            dd.x = av.x > 0.f ? dd.x : 0.f; dd.y = av.y > 0.f ? dd.y : 0.f;
            dd.z = av.z > 0.f ? dd.z : 0.f; dd.w = av.w > 0.f ? dd.w : 0.f;
            c_bias = add4(c_bias, dd);
            st4(cell_dz + p * C + cg * 4, dd);
        }
    }
    constexpr int RW = FUSE ? 20 : 8;
    float* mine = red + tid * RW;
    mine[0] = s_scale.x; mine[1] = s_scale.y; mine[2] = s_scale.z; mine[3] = s_scale.w;
    mine[4] = s_shift.x; mine[5] = s_shift.y; mine[6] = s_shift.z; mine[7] = s_shift.w;
    if (FUSE) {
        mine[8] = c_scale.x; mine[9] = c_scale.y; mine[10] = c_scale.z; mine[11] = c_scale.w;
        mine[12] = c_shift.x; mine[13] = c_shift.y; mine[14] = c_shift.z; mine[15] = c_shift.w;
        mine[16] = c_bias.x; mine[17] = c_bias.y; mine[18] = c_bias.z; mine[19] = c_bias.w;
    }
    __syncthreads();
    float* out = partials + (long)blockIdx.x * 2 * C;
    for (int i = tid; i < 2 * C; i += 256) {
        const int which = i / C, c = i - which * C;
        float sum = 0.f;
        for (int sl = 0; sl < nslots; ++sl) sum += red[(sl * C4 + (c >> 2)) * RW + which * 4 + (c & 3)];
        out[i] = sum;
    }
    if (FUSE) {
        float* outc = cell_partials + (long)blockIdx.x * 3 * C;
        for (int i = tid; i < 3 * C; i += 256) {
            const int which = i / C, c = i - which * C;
            float sum = 0.f;
            for (int sl = 0; sl < nslots; ++sl) sum += red[(sl * C4 + (c >> 2)) * RW + 8 + which * 4 + (c & 3)];
            outc[i] = sum;
        }
    }
}

__global__ void axpy_kernel(float* __restrict__ dst, const float* __restrict__ src, size_t n, float alpha, int accumulate) {
    for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += (size_t)gridDim.x * blockDim.x * 4) {
        if (i + 4 <= n) {
            float4 v = ld4(src + i);
            v = make_float4(alpha * v.x, alpha * v.y, alpha * v.z, alpha * v.w);
            if (accumulate) v = add4(v, ld4(dst + i));
            st4(dst + i, v);
        } else {
            for (size_t j = i; j < n; ++j) dst[j] = alpha * src[j] + (accumulate ? dst[j] : 0.f);
        }
    }
}

__global__ void relu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ h, size_t n, float* __restrict__ dz) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        dz[i] = h[i] > 0.f ? dy[i] : 0.f;
}

// 16-byte version (n4 = n / 4 float4 elements, 16-byte aligned pointers)
__global__ void relu_bwd4_kernel(const float* __restrict__ dy, const float* __restrict__ h, size_t n4, float* __restrict__ dz) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 d = ld4(dy + i * 4), a = ld4(h + i * 4);
        st4(dz + i * 4, make_float4(a.x > 0.f ? d.x : 0.f, a.y > 0.f ? d.y : 0.f, a.z > 0.f ? d.z : 0.f, a.w > 0.f ? d.w : 0.f));
    }
}

// dz = (h > 0) ? dy * scale : 0  (n4 float4 elements): the ReLU backward of an activation that was dropped out IN PLACE --
// its zeros already carry the dropout mask, so the backward of the dropout is only its 1 / (1 - rate) factor
__global__ void relu_bwd4_scaled_kernel(const float* __restrict__ dy, const float* __restrict__ h, size_t n4, float scale,
                                        float* __restrict__ dz) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 d = ld4(dy + i * 4), a = ld4(h + i * 4);
        st4(dz + i * 4, make_float4(a.x > 0.f ? d.x * scale : 0.f, a.y > 0.f ? d.y * scale : 0.f, a.z > 0.f ? d.z * scale : 0.f,
                                    a.w > 0.f ? d.w * scale : 0.f));
    }
}

inline bool chan_ok(int C) { return C >= 4 && (C & 3) == 0 && (256 % (C / 4)) == 0 && C <= 1024; }
inline int grid_for(long total, int threads) {
    long b = (total + threads - 1) / threads;
    return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

}  // namespace

// ===================================================================== C ABI
extern "C" int asr_cell1_fwd(const float* x, int B, int T, int F, int C, const float* w, const float* bias,
                             const float* bn_scale, const float* bn_shift, int pool, float* y, void* stream) {
    if (!x || !w || !bias || !bn_scale || !bn_shift || !y) return ASR_ERR_BAD_ARG;
    // two channels per lane: C even, C / 2 lanes per pixel dividing the 256-thread workgroup; y rows read / written as pairs
    if (C < 2 || C > 256 || (256 % C) != 0 || T < 2 || F < 2 || (pool != 1 && pool != 2) || ((size_t)y & 7)) return ASR_ERR_BAD_ARG;
    const int H2 = T / 2;
    dim3 grid(asr_cdiv(H2, kCell1RPB), B);
    const size_t lds = (size_t)8 * (F + 2) * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
    if (pool == 1)
        hipLaunchKernelGGL((cell1_kernel<1, false>), grid, dim3(256), lds, st, x, B, T, F, C, w, bias, bn_scale, bn_shift, y, (const float*)nullptr, (float*)nullptr, kCell1RPB);
    else
        hipLaunchKernelGGL((cell1_kernel<2, false>), grid, dim3(256), lds, st, x, B, T, F, C, w, bias, bn_scale, bn_shift, y, (const float*)nullptr, (float*)nullptr, kCell1RPB);
    ASR_CHECK_LAUNCH("cell1_fwd");
    return ASR_OK;
}

extern "C" size_t asr_cell1_bwd_workspace(int B, int T, int F, int C) {
    const size_t nblk = (size_t)asr_cdiv(T / 2, kCell1RPB) * B;
    return (nblk * 12 * C + asr_reduce::colsum_tmp_floats((int)nblk, 12 * C)) * sizeof(float);
}

extern "C" int asr_cell1_bwd(const float* x, int B, int T, int F, int C, const float* w, const float* bias,
                             const float* bn_scale, const float* bn_shift, int pool, const float* dy,
                             float* dw, float* db, float* dscale, float* dshift, float* partials, void* stream) {
    if (!x || !w || !bias || !bn_scale || !bn_shift || !dy || !dw || !db || !dscale || !dshift || !partials)
        return ASR_ERR_BAD_ARG;
    if (C < 2 || C > 256 || (256 % C) != 0 || T < 2 || F < 2 || (pool != 1 && pool != 2) || ((size_t)dy & 7)) return ASR_ERR_BAD_ARG;
    const int H2 = T / 2;
    dim3 grid(asr_cdiv(H2, kCell1RPB), B);
    const int nblk = grid.x * grid.y;
    size_t lds = (size_t)8 * (F + 2) * sizeof(float);
    const size_t red = (size_t)(C < 128 ? 4 : 512 / C) * 12 * C * sizeof(float);
    if (red > lds) lds = red;
    hipStream_t st = (hipStream_t)stream;
    if (pool == 1)
        hipLaunchKernelGGL((cell1_kernel<1, true>), grid, dim3(256), lds, st, x, B, T, F, C, w, bias, bn_scale, bn_shift, (float*)nullptr, dy, partials, kCell1RPB);
    else
        hipLaunchKernelGGL((cell1_kernel<2, true>), grid, dim3(256), lds, st, x, B, T, F, C, w, bias, bn_scale, bn_shift, (float*)nullptr, dy, partials, kCell1RPB);
    ASR_CHECK_LAUNCH("cell1_bwd");
    float* tmp = partials + (size_t)nblk * 12 * C;
    asr_reduce::Multi m;
    m.nseg = 4; m.width[0] = 9 * C; m.width[1] = C; m.width[2] = C; m.width[3] = C;
    m.out[0] = dw; m.out[1] = db; m.out[2] = dscale; m.out[3] = dshift;
    return asr_reduce::colsum_multi(partials, nblk, 12 * C, m, tmp, st);
}

extern "C" int asr_pool_fwd(const float* a, int B, int H, int W, int C, const float* bn_scale,
                            const float* bn_shift, int pool, float* y, void* stream) {
    if (!a || !bn_scale || !bn_shift || !y || (C & 3) || H < 2 || W < 2 || (pool != 1 && pool != 2)) return ASR_ERR_BAD_ARG;
    const long total = (long)B * (H / 2) * (W / 2) * (C / 4);
    hipStream_t st = (hipStream_t)stream;
    if (pool == 1) hipLaunchKernelGGL(pool_fwd_kernel<1>, dim3(grid_for(total, 256)), dim3(256), 0, st, a, B, H, W, C, bn_scale, bn_shift, y);
    else hipLaunchKernelGGL(pool_fwd_kernel<2>, dim3(grid_for(total, 256)), dim3(256), 0, st, a, B, H, W, C, bn_scale, bn_shift, y);
    ASR_CHECK_LAUNCH("pool_fwd");
    return ASR_OK;
}

extern "C" size_t asr_cell_bwd_pre_workspace(int B, int H, int W, int C) {
    const long NP = (long)B * (H + 1) * (W + 1);
    const size_t nblk = (size_t)asr_cdiv(NP, pre_ppb(NP));
    return (nblk * 3 * C + asr_reduce::colsum_tmp_floats((int)nblk, 3 * C)) * sizeof(float);
}

extern "C" int asr_cell_bwd_pre(const float* dy, int dy_layout, const float* a, int B, int H, int W, int C,
                                const float* bn_scale, const float* bn_shift, int pool,
                                float* dz, float* dscale, float* dshift, float* dbias,
                                float* partials, void* stream) {
    if (!dy || !a || !bn_scale || !bn_shift || !dz || !dscale || !dshift || !dbias || !partials) return ASR_ERR_BAD_ARG;
    if (!chan_ok(C) || pool < 0 || pool > 2) return ASR_ERR_BAD_ARG;
    if ((pool != 0) != (dy_layout == 1)) return ASR_ERR_BAD_ARG;
    if (dy_layout < 0 || dy_layout > 2) return ASR_ERR_BAD_ARG;
    const long NP = (long)B * (H + 1) * (W + 1);
    const int ppb = pre_ppb(NP);
    const int nblk = asr_cdiv(NP, ppb);
    hipStream_t st = (hipStream_t)stream;
    if (pool == 0) hipLaunchKernelGGL(cell_bwd_pre_kernel<0>, dim3(nblk), dim3(256), 0, st, dy, dy_layout, a, B, H, W, C, bn_scale, bn_shift, dz, partials, ppb);
    else if (pool == 1) hipLaunchKernelGGL(cell_bwd_pre_kernel<1>, dim3(nblk), dim3(256), 0, st, dy, dy_layout, a, B, H, W, C, bn_scale, bn_shift, dz, partials, ppb);
    else hipLaunchKernelGGL(cell_bwd_pre_kernel<2>, dim3(nblk), dim3(256), 0, st, dy, dy_layout, a, B, H, W, C, bn_scale, bn_shift, dz, partials, ppb);
    ASR_CHECK_LAUNCH("cell_bwd_pre");
    float* tmp = partials + (size_t)nblk * 3 * C;
    asr_reduce::Multi m;
    m.nseg = 3; m.width[0] = C; m.width[1] = C; m.width[2] = C; m.width[3] = 0;
    m.out[0] = dscale; m.out[1] = dshift; m.out[2] = dbias; m.out[3] = nullptr;
    return asr_reduce::colsum_multi(partials, nblk, 3 * C, m, tmp, st);
}

extern "C" size_t asr_se_state_floats(int B, int C, int hid) { return (size_t)B * (2 * C + hid); }

static inline int se_nsplit(int H, int W) { return asr_cdiv((long)(H + 1) * (W + 1), kSePPB); }

extern "C" size_t asr_se_fwd_workspace(int B, int H, int W, int C) {
    return ((size_t)B * se_nsplit(H, W) * C + 64) * sizeof(float);
}

extern "C" size_t asr_se_bwd_workspace(int B, int H, int W, int C, int hid) {
    const size_t per = (size_t)C * hid + hid + (size_t)hid * C + C;
    const long NP = (long)B * (H + 1) * (W + 1);
    const size_t nblk = (size_t)asr_cdiv(NP, se_apply_ppb(NP));
    const size_t fl = (size_t)B * se_nsplit(H, W) * C + (size_t)B * per + (size_t)B * C + nblk * 2 * C
                    + asr_reduce::colsum_tmp_floats((int)nblk, 2 * C) + 64;
    return fl * sizeof(float);
}

extern "C" int asr_se_fwd(const float* main_in, const float* x, int B, int H, int W, int C, int hid,
                          const float* bn_scale, const float* bn_shift, const float* w1, const float* b1,
                          const float* w2, const float* b2, float* state, float* partials, float* out,
                          void* stream) {
    if (!main_in || !x || !bn_scale || !bn_shift || !w1 || !b1 || !w2 || !b2 || !state || !partials || !out)
        return ASR_ERR_BAD_ARG;
    if (!chan_ok(C) || hid < 1 || hid > 1024) return ASR_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int ns = se_nsplit(H, W);
    float* st_s = state; float* st_r = st_s + (size_t)B * C; float* st_e = st_r + (size_t)B * hid;
    hipLaunchKernelGGL(se_reduce_kernel<0>, dim3(ns, B), dim3(256), 0, st, x, (const float*)nullptr, H, W, C, bn_scale, bn_shift, partials);
    hipLaunchKernelGGL(se_excite_kernel, dim3(B), dim3(256), (size_t)(C + hid + 256) * sizeof(float), st, (const float*)partials, ns, H, W, C, hid, bn_scale, bn_shift, w1, b1, w2, b2, st_s, st_r, st_e);
    const long total = (long)B * (H + 1) * (W + 1) * (C / 4);
    hipLaunchKernelGGL(se_apply_kernel, dim3(grid_for(total, 256)), dim3(256), 0, st, main_in, x, B, H, W, C, bn_scale, bn_shift, (const float*)st_e, out);
    ASR_CHECK_LAUNCH("se_fwd");
    return ASR_OK;
}

// asr_se_fwd with the squeeze sums already made by the launch that wrote x (asr_tap_gemm_wino_sums): sums [B][nsplit][C] partial rows
extern "C" int asr_se_fwd_sums(const float* main_in, const float* x, int B, int H, int W, int C, int hid,
                               const float* bn_scale, const float* bn_shift, const float* w1, const float* b1,
                               const float* w2, const float* b2, float* state, const float* sums, int nsplit, float* out, void* stream) {
    if (!main_in || !x || !bn_scale || !bn_shift || !w1 || !b1 || !w2 || !b2 || !state || !sums || !out || nsplit < 1) return ASR_ERR_BAD_ARG;
    if (!chan_ok(C) || hid < 1 || hid > 1024) return ASR_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    float* st_s = state; float* st_r = st_s + (size_t)B * C; float* st_e = st_r + (size_t)B * hid;
    hipLaunchKernelGGL(se_excite_kernel, dim3(B), dim3(256), (size_t)(C + hid + 256) * sizeof(float), st, sums, nsplit, H, W, C, hid, bn_scale, bn_shift, w1, b1, w2, b2, st_s, st_r, st_e);
    const long total = (long)B * (H + 1) * (W + 1) * (C / 4);
    hipLaunchKernelGGL(se_apply_kernel, dim3(grid_for(total, 256)), dim3(256), 0, st, main_in, x, B, H, W, C, bn_scale, bn_shift, (const float*)st_e, out);
    ASR_CHECK_LAUNCH("se_fwd_sums");
    return ASR_OK;
}

struct SeCell { const float* a; const float* scale; float* dz; float* dscale; float* dshift; float* dbias; };

static int se_bwd_impl(const float* dout, const float* x, int B, int H, int W, int C, int hid,
                       const float* bn_scale, const float* bn_shift, const float* w1, const float* w2,
                       const float* state, int add_dout, float* dx, float* dscale, float* dshift,
                       float* dw1, float* db1, float* dw2, float* db2, float* partials, void* stream, const SeCell* cell,
                       const float* xsums = nullptr, int xsplit = 0) {
    if (!dout || !x || !bn_scale || !bn_shift || !w1 || !w2 || !state || (!dx && !cell) || !dscale || !dshift || !dw1 ||
        !db1 || !dw2 || !db2 || !partials)
        return ASR_ERR_BAD_ARG;
    if (!chan_ok(C) || hid < 1 || hid > 1024) return ASR_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int ns = se_nsplit(H, W);
    const size_t per = (size_t)C * hid + hid + (size_t)hid * C + C;
    const long NP = (long)B * (H + 1) * (W + 1);
    const int appb = se_apply_ppb(NP);
    const int nblk = asr_cdiv(NP, appb);
    float* part_red = partials;
    float* mlp_out = part_red + (size_t)B * ns * C;
    float* dsb = mlp_out + (size_t)B * per;
    float* part_apply = dsb + (size_t)B * C;
    float* tmp = part_apply + (size_t)nblk * 2 * C;
    const float* st_s = state; const float* st_r = st_s + (size_t)B * C; const float* st_e = st_r + (size_t)B * hid;
    // sum over the pixels of dout * (sc * x + sh) per image and channel: a pass of its own, or the partial rows the data-gradient launch
    // that wrote dout left (asr_tap_gemm_wino_sesum)
    if (!xsums) hipLaunchKernelGGL(se_reduce_kernel<1>, dim3(ns, B), dim3(256), 0, st, x, dout, H, W, C, bn_scale, bn_shift, part_red);
    hipLaunchKernelGGL(se_bwd_mlp_kernel, dim3(B), dim3(256), (size_t)(C + hid + 256) * sizeof(float), st, xsums ? xsums : (const float*)part_red,
                       xsums ? xsplit : ns, H, W, C, hid, w1, w2, st_s, st_r, st_e, mlp_out, dsb);
    ASR_CHECK_LAUNCH("se_bwd_mlp");
    int rc;
    {
        asr_reduce::Multi m;
        m.nseg = 4; m.width[0] = C * hid; m.width[1] = hid; m.width[2] = hid * C; m.width[3] = C;
        m.out[0] = dw1; m.out[1] = db1; m.out[2] = dw2; m.out[3] = db2;
        if ((rc = asr_reduce::colsum_multi(mlp_out, B, (long)per, m, tmp, st))) return rc;
    }
    float* part_cell = tmp + asr_reduce::colsum_tmp_floats(nblk, 2 * C) + 64;      // asr_se_bwd_cell_workspace: [nblk][3 C] + its scratch behind the rest
    if (cell)
        hipLaunchKernelGGL(se_bwd_apply_kernel<true>, dim3(nblk), dim3(256), 0, st, dout, x, B, H, W, C, bn_scale, st_e, (const float*)dsb, add_dout, dx,
                           part_apply, appb, cell->a, cell->scale, cell->dz, part_cell);
    else
        hipLaunchKernelGGL(se_bwd_apply_kernel<false>, dim3(nblk), dim3(256), 0, st, dout, x, B, H, W, C, bn_scale, st_e, (const float*)dsb, add_dout, dx,
                           part_apply, appb, (const float*)nullptr, (const float*)nullptr, (float*)nullptr, (float*)nullptr);
    ASR_CHECK_LAUNCH("se_bwd_apply");
    {
        asr_reduce::Multi m;
        m.nseg = 2; m.width[0] = C; m.width[1] = C; m.width[2] = 0; m.width[3] = 0;
        m.out[0] = dscale; m.out[1] = dshift; m.out[2] = nullptr; m.out[3] = nullptr;
        if ((rc = asr_reduce::colsum_multi(part_apply, nblk, 2 * C, m, tmp, st))) return rc;
    }
    if (cell) {                                          // the reduction asr_cell_bwd_pre runs on the same block partials
        asr_reduce::Multi m;
        m.nseg = 3; m.width[0] = C; m.width[1] = C; m.width[2] = C; m.width[3] = 0;
        m.out[0] = cell->dscale; m.out[1] = cell->dshift; m.out[2] = cell->dbias; m.out[3] = nullptr;
        if ((rc = asr_reduce::colsum_multi(part_cell, nblk, 3 * C, m, part_cell + (size_t)nblk * 3 * C, st))) return rc;
    }
    return ASR_OK;
}

extern "C" int asr_se_bwd(const float* dout, const float* x, int B, int H, int W, int C, int hid,
                          const float* bn_scale, const float* bn_shift, const float* w1, const float* w2,
                          const float* state, int add_dout, float* dx, float* dscale, float* dshift,
                          float* dw1, float* db1, float* dw2, float* db2, float* partials, void* stream) {
    if (!dx) return ASR_ERR_BAD_ARG;
    return se_bwd_impl(dout, x, B, H, W, C, hid, bn_scale, bn_shift, w1, w2, state, add_dout, dx, dscale, dshift, dw1, db1, dw2, db2,
                       partials, stream, nullptr);
}

extern "C" size_t asr_se_bwd_cell_workspace(int B, int H, int W, int C, int hid) {
    const long NP = (long)B * (H + 1) * (W + 1);
    const size_t nblk = (size_t)asr_cdiv(NP, se_apply_ppb(NP));
    return asr_se_bwd_workspace(B, H, W, C, hid) + (nblk * 3 * C + asr_reduce::colsum_tmp_floats((int)nblk, 3 * C) + 64) * sizeof(float);
}

extern "C" int asr_se_bwd_cell(const float* dout, const float* x, int B, int H, int W, int C, int hid,
                               const float* bn_scale, const float* bn_shift, const float* w1, const float* w2,
                               const float* state, int add_dout, float* dscale, float* dshift,
                               float* dw1, float* db1, float* dw2, float* db2,
                               const float* cell_a, const float* cell_scale, float* cell_dz, float* cell_dscale, float* cell_dshift,
                               float* cell_dbias, float* partials, void* stream) {
    if (!cell_a || !cell_scale || !cell_dz || !cell_dscale || !cell_dshift || !cell_dbias) return ASR_ERR_BAD_ARG;
    static_assert(sizeof(float) == 4, "");
    SeCell c; c.a = cell_a; c.scale = cell_scale; c.dz = cell_dz; c.dscale = cell_dscale; c.dshift = cell_dshift; c.dbias = cell_dbias;
    return se_bwd_impl(dout, x, B, H, W, C, hid, bn_scale, bn_shift, w1, w2, state, add_dout, nullptr, dscale, dshift, dw1, db1, dw2, db2,
                       partials, stream, &c);
}

// asr_se_bwd_cell with the block's first reduction already made by the launch that wrote dout (asr_tap_gemm_wino_sesum)
extern "C" int asr_se_bwd_cell_sums(const float* dout, const float* x, int B, int H, int W, int C, int hid,
                                    const float* bn_scale, const float* bn_shift, const float* w1, const float* w2,
                                    const float* state, int add_dout, float* dscale, float* dshift,
                                    float* dw1, float* db1, float* dw2, float* db2,
                                    const float* cell_a, const float* cell_scale, float* cell_dz, float* cell_dscale, float* cell_dshift,
                                    float* cell_dbias, const float* xsums, int nsplit, float* partials, void* stream) {
    if (!cell_a || !cell_scale || !cell_dz || !cell_dscale || !cell_dshift || !cell_dbias || !xsums || nsplit < 1) return ASR_ERR_BAD_ARG;
    SeCell c; c.a = cell_a; c.scale = cell_scale; c.dz = cell_dz; c.dscale = cell_dscale; c.dshift = cell_dshift; c.dbias = cell_dbias;
    return se_bwd_impl(dout, x, B, H, W, C, hid, bn_scale, bn_shift, w1, w2, state, add_dout, nullptr, dscale, dshift, dw1, db1, dw2, db2,
                       partials, stream, &c, xsums, nsplit);
}

extern "C" int asr_axpy(float* dst, const float* src, size_t n, float alpha, int accumulate, void* stream) {
    if (!dst || !src) return ASR_ERR_BAD_ARG;
    if (n == 0) return ASR_OK;
    hipLaunchKernelGGL(axpy_kernel, dim3(grid_for((long)((n + 3) / 4), 256)), dim3(256), 0, (hipStream_t)stream, dst, src, n, alpha, accumulate);
    ASR_CHECK_LAUNCH("axpy");
    return ASR_OK;
}

extern "C" int asr_relu_bwd(const float* dy, const float* h, size_t n, float* dz, void* stream) {
    if (!dy || !h || !dz) return ASR_ERR_BAD_ARG;
    if (n == 0) return ASR_OK;
    if ((n & 3) == 0 && ((((uintptr_t)dy | (uintptr_t)h | (uintptr_t)dz)) & 15) == 0) {
        long b = (long)((n / 4 + 255) / 256);
        if (b > 16384) b = 16384;
        hipLaunchKernelGGL(relu_bwd4_kernel, dim3((int)b), dim3(256), 0, (hipStream_t)stream, dy, h, n / 4, dz);
    } else {
        hipLaunchKernelGGL(relu_bwd_kernel, dim3(grid_for((long)n, 256)), dim3(256), 0, (hipStream_t)stream, dy, h, n, dz);
    }
    ASR_CHECK_LAUNCH("relu_bwd");
    return ASR_OK;
}

extern "C" int asr_relu_bwd_scaled(const float* dy, const float* h, size_t n, float scale, float* dz, void* stream) {
    if (!dy || !h || !dz || n == 0 || (n & 3) || (((uintptr_t)dy | (uintptr_t)h | (uintptr_t)dz) & 15)) return ASR_ERR_BAD_ARG;
    long b = (long)((n / 4 + 255) / 256);
    if (b > 16384) b = 16384;
    hipLaunchKernelGGL(relu_bwd4_scaled_kernel, dim3((int)b), dim3(256), 0, (hipStream_t)stream, dy, h, n / 4, scale, dz);
    ASR_CHECK_LAUNCH("relu_bwd_scaled");
    return ASR_OK;
}

extern "C" size_t asr_colsum_workspace(int rows, int cols) {
    return (asr_reduce::colsum_tmp_floats(rows, cols) + 4) * sizeof(float);
}

extern "C" int asr_colsum(const float* x, int rows, int cols, int ldx, float* out, float* partials, void* stream) {
    if (!x || !out || rows < 1 || cols < 1) return ASR_ERR_BAD_ARG;
    if (rows > asr_reduce::kSplits && !partials) return ASR_ERR_BAD_ARG;
    return asr_reduce::colsum(x, rows, cols, ldx, out, partials, (hipStream_t)stream);
}

// end2end pre-net (end2end/model.py:214-264) pieces that are not tap-GEMMs:
//   * first conv (1 -> 64 channels, 3x3, stride 2, tanh): direct, HBM-bound, forward + weight gradient;
//   * batch-statistics BatchNorm (tf.layers.batch_normalization(training=True)): statistics in float64,
//     apply (optionally + residual + ReLU), backward (fused with the activation derivative of the conv in
//     front of it), all with fixed-order reductions (no float atomics);
//   * the "phase split" view that turns the 64 -> 64 stride-2 conv into a stride-1 2x2-tap tap-GEMM over
//     256 channels (weights expanded / gradients gathered here);
//   * [B][T][F][c] <-> [B][T][c][F] transposes in front of / behind the 2-D attention;
//   * the frequency-axis attention (80 x 80 scores per (batch, channel), contraction over time) on MFMA;
//   * residual add + LayerNorm over the 64 channels of a padded plane, forward / backward.
// Pixel addressing goes through asr_pixmap (include/asr_hip.h): padded plane, plain NHWC, or phase-split plane.
#include "asr_common.h"
#include "reduce.h"
#include <math.h>

namespace {

__device__ __forceinline__ long pix_off(const asr_pixmap& m, int b, int h, int w) {
    if (m.kind == 0) return ((long)(b * (m.H + 1) + h + 1) * (m.W + 1) + w + 1) * m.ld;
    if (m.kind == 1) return ((long)(b * m.H + h) * m.W + w) * m.ld;
    // kind 2: tensor [B][H][W][C] (H, W even) held as a padded plane of (H/2, W/2) pixels x 4C channels
    return ((long)(b * (m.H / 2 + 1) + (h >> 1) + 1) * (m.W / 2 + 1) + (w >> 1) + 1) * m.ld + ((h & 1) * 2 + (w & 1)) * m.C;
}

// pixel counts are checked on the host to fit 31 bits: 32-bit division only (a 64-bit one per pixel costs more
// than the memory access it addresses)
__device__ __forceinline__ void pix_decode(const asr_pixmap& m, int p, int& b, int& h, int& w) {
    const unsigned hw = (unsigned)(m.H * m.W), up = (unsigned)p;
    const unsigned ub = up / hw, r = up - ub * hw;
    const unsigned uh = r / (unsigned)m.W;
    b = (int)ub; h = (int)uh; w = (int)(r - uh * (unsigned)m.W);
}

// ------------------------------------------------------------------ first conv: 1 -> CO channels, 3x3, stride 2, tanh
// TF 'same' for even sizes and stride 2: no padding in front, one zero row / column behind.
constexpr int CO = 64;

__global__ __launch_bounds__(256) void conv1_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ bias, int B, int T, int F,
                                                        float* __restrict__ a1) {
    __shared__ float ws[9 * CO + CO];
    for (int i = threadIdx.x; i < 9 * CO; i += 256) ws[i] = w[i];
    for (int i = threadIdx.x; i < CO; i += 256) ws[9 * CO + i] = bias[i];
    __syncthreads();
    const int H1 = T / 2, W1 = F / 2;
    const int npix = B * H1 * W1;
    const int c4 = threadIdx.x & 15;
    for (int p = blockIdx.x * 16 + (threadIdx.x >> 4); p < npix; p += gridDim.x * 16) {
        const int b = p / (H1 * W1);
        const int r = p - b * H1 * W1;
        const int i = r / W1, j = r - i * W1;
        float4 acc = *(const float4*)(ws + 9 * CO + c4 * 4);
#pragma unroll
        for (int dh = 0; dh < 3; ++dh)
#pragma unroll
            for (int dw = 0; dw < 3; ++dw) {
                const int t = 2 * i + dh, f = 2 * j + dw;
                const float xv = (t < T && f < F) ? x[((long)b * T + t) * F + f] : 0.f;
                const float4 wv = *(const float4*)(ws + (dh * 3 + dw) * CO + c4 * 4);
                acc.x += xv * wv.x; acc.y += xv * wv.y; acc.z += xv * wv.z; acc.w += xv * wv.w;
            }
        *(float4*)(a1 + (long)p * CO + c4 * 4) = make_float4(tanhf(acc.x), tanhf(acc.y), tanhf(acc.z), tanhf(acc.w));
    }
}

// dW[tap][c] = sum_pixels x[2i+dh][2j+dw] * dz[pixel][c], db[c] = sum dz: block partials [blocks][10*CO]
__global__ __launch_bounds__(256) void conv1_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dz,
                                                        int B, int T, int F, float* __restrict__ partials) {
    __shared__ float red[16][10 * CO + 4];
    const int H1 = T / 2, W1 = F / 2;
    const int npix = B * H1 * W1;
    const int c4 = threadIdx.x & 15, pl = threadIdx.x >> 4;
    float4 acc[10];
#pragma unroll
    for (int t = 0; t < 10; ++t) acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int p = blockIdx.x * 16 + pl; p < npix; p += gridDim.x * 16) {
        const int b = p / (H1 * W1);
        const int r = p - b * H1 * W1;
        const int i = r / W1, j = r - i * W1;
        const float4 g = *(const float4*)(dz + (long)p * CO + c4 * 4);
#pragma unroll
        for (int dh = 0; dh < 3; ++dh)
#pragma unroll
            for (int dw = 0; dw < 3; ++dw) {
                const int t = 2 * i + dh, f = 2 * j + dw;
                const float xv = (t < T && f < F) ? x[((long)b * T + t) * F + f] : 0.f;
                float4& a = acc[dh * 3 + dw];
                a.x += xv * g.x; a.y += xv * g.y; a.z += xv * g.z; a.w += xv * g.w;
            }
        acc[9].x += g.x; acc[9].y += g.y; acc[9].z += g.z; acc[9].w += g.w;
    }
#pragma unroll
    for (int t = 0; t < 10; ++t) *(float4*)(&red[pl][t * CO + c4 * 4]) = acc[t];
    __syncthreads();
    for (int i = threadIdx.x; i < 10 * CO; i += 256) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) s += red[k][i];
        partials[(long)blockIdx.x * (10 * CO) + i] = s;
    }
}

// ------------------------------------------------------------------ batch-statistics BatchNorm
constexpr int kUn = 4;      // pixels per lane per loop trip: with one 16-byte load in flight per lane these kernels sat at 3.7 TB/s
// One thread owns 4 channels of a pixel; C/4 threads per pixel, 256/(C/4) pixels per block iteration.
// Sums are kept in float64 end to end (the oracle is float64; E[x^2]-mean^2 in float32 would not do).
__global__ __launch_bounds__(256) void bn_stats_kernel(const float* __restrict__ src, asr_pixmap m,
                                                       double* __restrict__ partials) {
    extern __shared__ double sh[];                     // [ppb][2*C]
    const int cpp = m.C / 4, ppb = 256 / cpp;
    const int c4 = threadIdx.x % cpp, pl = threadIdx.x / cpp;
    const int npix = m.B * m.H * m.W;
    double s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
    const int stride = gridDim.x * ppb;
    for (int p0 = blockIdx.x * ppb + pl; p0 < npix; p0 += kUn * stride) {       // kUn independent loads in flight per lane
        float4 v[kUn];
#pragma unroll
        for (int u = 0; u < kUn; ++u) {
            const int p = p0 + u * stride;
            v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p < npix) { int b, h, w; pix_decode(m, p, b, h, w); v[u] = *(const float4*)(src + pix_off(m, b, h, w) + c4 * 4); }
        }
#pragma unroll
        for (int u = 0; u < kUn; ++u) {
            s[0] += v[u].x; s[1] += v[u].y; s[2] += v[u].z; s[3] += v[u].w;
            q[0] += (double)v[u].x * v[u].x; q[1] += (double)v[u].y * v[u].y;
            q[2] += (double)v[u].z * v[u].z; q[3] += (double)v[u].w * v[u].w;
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        sh[pl * 2 * m.C + c4 * 4 + k] = s[k];
        sh[pl * 2 * m.C + m.C + c4 * 4 + k] = q[k];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * m.C; i += 256) {
        double t = 0;
        for (int k = 0; k < ppb; ++k) t += sh[k * 2 * m.C + i];
        partials[(long)blockIdx.x * 2 * m.C + i] = t;
    }
}

// one wave per channel: lane l folds partial blocks l, l+64, ... in float64, then a fixed shuffle tree
__global__ __launch_bounds__(64) void bn_stats_final_kernel(const double* __restrict__ partials, int nblocks, int C, double count,
                                                            float eps, float* __restrict__ mean, float* __restrict__ rstd) {
    const int c = blockIdx.x, l = threadIdx.x;
    double s = 0, q = 0;
    for (int k = l; k < nblocks; k += 64) { s += partials[(long)k * 2 * C + c]; q += partials[(long)k * 2 * C + C + c]; }
    s = asr_wave_sum_d(s); q = asr_wave_sum_d(q);
    if (l == 0) {
        const double mu = s / count;
        double var = q / count - mu * mu;
        if (var < 0) var = 0;
        mean[c] = (float)mu;
        rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    }
}

// y = gamma * (a - mean) * rstd + beta; optionally y = relu(y + res)
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ src, asr_pixmap sm,
                                                       const float* __restrict__ mean, const float* __restrict__ rstd,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const float* __restrict__ res, asr_pixmap rm, int relu,
                                                       float* __restrict__ dst, asr_pixmap dm) {
    const int cpp = sm.C / 4, ppb = 256 / cpp;
    const int c4 = threadIdx.x % cpp, pl = threadIdx.x / cpp;
    const int npix = sm.B * sm.H * sm.W;
    const float4 mu = *(const float4*)(mean + c4 * 4), rs = *(const float4*)(rstd + c4 * 4);
    const float4 g = *(const float4*)(gamma + c4 * 4), be = *(const float4*)(beta + c4 * 4);
    const int stride = gridDim.x * ppb;
    for (int p0 = blockIdx.x * ppb + pl; p0 < npix; p0 += kUn * stride) {
        float4 v[kUn], r[kUn];
        int bb[kUn], hh[kUn], ww[kUn];
#pragma unroll
        for (int u = 0; u < kUn; ++u) {
            const int p = p0 + u * stride;
            v[u] = make_float4(0.f, 0.f, 0.f, 0.f); r[u] = v[u]; bb[u] = -1; hh[u] = 0; ww[u] = 0;
            if (p < npix) {
                pix_decode(sm, p, bb[u], hh[u], ww[u]);
                v[u] = *(const float4*)(src + pix_off(sm, bb[u], hh[u], ww[u]) + c4 * 4);
                if (res) r[u] = *(const float4*)(res + pix_off(rm, bb[u], hh[u], ww[u]) + c4 * 4);
            }
        }
#pragma unroll
        for (int u = 0; u < kUn; ++u) {
            if (bb[u] < 0) continue;
            float4 y = make_float4(g.x * ((v[u].x - mu.x) * rs.x) + be.x + r[u].x, g.y * ((v[u].y - mu.y) * rs.y) + be.y + r[u].y,
                                   g.z * ((v[u].z - mu.z) * rs.z) + be.z + r[u].z, g.w * ((v[u].w - mu.w) * rs.w) + be.w + r[u].w);
            if (relu) { y.x = fmaxf(y.x, 0.f); y.y = fmaxf(y.y, 0.f); y.z = fmaxf(y.z, 0.f); y.w = fmaxf(y.w, 0.f); }
            *(float4*)(dst + pix_off(dm, bb[u], hh[u], ww[u]) + c4 * 4) = y;
        }
    }
}

// per-channel sum(dy) and sum(dy * xhat), float64 partials
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float* __restrict__ dy, asr_pixmap ym,
                                                            const float* __restrict__ a, asr_pixmap am,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            double* __restrict__ partials) {
    extern __shared__ double sh[];
    const int C = am.C, cpp = C / 4, ppb = 256 / cpp;
    const int c4 = threadIdx.x % cpp, pl = threadIdx.x / cpp;
    const int npix = am.B * am.H * am.W;
    const float4 mu = *(const float4*)(mean + c4 * 4), rs = *(const float4*)(rstd + c4 * 4);
    double s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
    const int stride = gridDim.x * ppb;
    for (int p0 = blockIdx.x * ppb + pl; p0 < npix; p0 += kUn * stride) {
        float4 g[kUn], v[kUn];
#pragma unroll
        for (int u = 0; u < kUn; ++u) {
            const int p = p0 + u * stride;
            g[u] = make_float4(0.f, 0.f, 0.f, 0.f); v[u] = g[u];
            if (p < npix) {
                int b, h, w;
                pix_decode(am, p, b, h, w);
                g[u] = *(const float4*)(dy + pix_off(ym, b, h, w) + c4 * 4);
                v[u] = *(const float4*)(a + pix_off(am, b, h, w) + c4 * 4);
            }
        }
#pragma unroll
        for (int u = 0; u < kUn; ++u) {          // out-of-range lanes hold dy = 0: they add nothing
            s[0] += g[u].x; s[1] += g[u].y; s[2] += g[u].z; s[3] += g[u].w;
            q[0] += (double)g[u].x * ((v[u].x - mu.x) * rs.x); q[1] += (double)g[u].y * ((v[u].y - mu.y) * rs.y);
            q[2] += (double)g[u].z * ((v[u].z - mu.z) * rs.z); q[3] += (double)g[u].w * ((v[u].w - mu.w) * rs.w);
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        sh[pl * 2 * C + c4 * 4 + k] = s[k];
        sh[pl * 2 * C + C + c4 * 4 + k] = q[k];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * C; i += 256) {
        double t = 0;
        for (int k = 0; k < ppb; ++k) t += sh[k * 2 * C + i];
        partials[(long)blockIdx.x * 2 * C + i] = t;
    }
}

// dgamma = sum(dy*xhat), dbeta = sum(dy); also the two means the apply pass needs (as floats in `sums`);
// one wave per channel as above
__global__ __launch_bounds__(64) void bn_bwd_final_kernel(const double* __restrict__ partials, int nblocks, int C, double count,
                                                          float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                          float* __restrict__ sums) {
    const int c = blockIdx.x, l = threadIdx.x;
    double s = 0, q = 0;
    for (int k = l; k < nblocks; k += 64) { s += partials[(long)k * 2 * C + c]; q += partials[(long)k * 2 * C + C + c]; }
    s = asr_wave_sum_d(s); q = asr_wave_sum_d(q);
    if (l == 0) {
        dbeta[c] = (float)s;
        dgamma[c] = (float)q;
        sums[c] = (float)(s / count);
        sums[C + c] = (float)(q / count);
    }
}

// d(a) = gamma*rstd*(dy - mean(dy) - xhat*mean(dy*xhat));  dz = d(a) * act'(z) written through a:
//   act 0: identity, 1: relu (a > 0), 2: tanh (1 - a^2)
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ dy, asr_pixmap ym,
                                                           const float* __restrict__ a, asr_pixmap am,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           const float* __restrict__ gamma, const float* __restrict__ sums,
                                                           int act, float* __restrict__ dz, asr_pixmap zm) {
    const int C = am.C, cpp = C / 4, ppb = 256 / cpp;
    const int c4 = threadIdx.x % cpp, pl = threadIdx.x / cpp;
    const int npix = am.B * am.H * am.W;
    const float4 mu = *(const float4*)(mean + c4 * 4), rs = *(const float4*)(rstd + c4 * 4);
    const float4 g = *(const float4*)(gamma + c4 * 4);
    const float4 m1 = *(const float4*)(sums + c4 * 4), m2 = *(const float4*)(sums + C + c4 * 4);
    const int stride = gridDim.x * ppb;
    for (int p0 = blockIdx.x * ppb + pl; p0 < npix; p0 += kUn * stride) {
      float4 dd[kUn], vv[kUn];
      int bb[kUn], hh[kUn], ww[kUn];
#pragma unroll
      for (int u = 0; u < kUn; ++u) {
        const int p = p0 + u * stride;
        dd[u] = make_float4(0.f, 0.f, 0.f, 0.f); vv[u] = dd[u]; bb[u] = -1; hh[u] = 0; ww[u] = 0;
        if (p < npix) {
            pix_decode(am, p, bb[u], hh[u], ww[u]);
            dd[u] = *(const float4*)(dy + pix_off(ym, bb[u], hh[u], ww[u]) + c4 * 4);
            vv[u] = *(const float4*)(a + pix_off(am, bb[u], hh[u], ww[u]) + c4 * 4);
        }
      }
#pragma unroll
      for (int u = 0; u < kUn; ++u) {
        if (bb[u] < 0) continue;
        const int b = bb[u], h = hh[u], w = ww[u];
        const float4 d = dd[u], v = vv[u];
        float4 o;
        o.x = g.x * rs.x * (d.x - m1.x - (v.x - mu.x) * rs.x * m2.x);
        o.y = g.y * rs.y * (d.y - m1.y - (v.y - mu.y) * rs.y * m2.y);
        o.z = g.z * rs.z * (d.z - m1.z - (v.z - mu.z) * rs.z * m2.z);
        o.w = g.w * rs.w * (d.w - m1.w - (v.w - mu.w) * rs.w * m2.w);
        if (act == 1) { o.x = v.x > 0.f ? o.x : 0.f; o.y = v.y > 0.f ? o.y : 0.f; o.z = v.z > 0.f ? o.z : 0.f; o.w = v.w > 0.f ? o.w : 0.f; }
        else if (act == 2) { o.x *= 1.f - v.x * v.x; o.y *= 1.f - v.y * v.y; o.z *= 1.f - v.z * v.z; o.w *= 1.f - v.w * v.w; }
        *(float4*)(dz + pix_off(zm, b, h, w) + c4 * 4) = o;
      }
    }
}

// dst = dy * (y > 0)   (gradient of the final relu(f2 + out), written into a plane)
__global__ __launch_bounds__(256) void relu_mask_kernel(const float* __restrict__ dy, asr_pixmap ym,
                                                        const float* __restrict__ y, asr_pixmap om,
                                                        float* __restrict__ dst, asr_pixmap dm) {
    const int cpp = ym.C / 4, ppb = 256 / cpp;
    const int c4 = threadIdx.x % cpp, pl = threadIdx.x / cpp;
    const int npix = ym.B * ym.H * ym.W;
    for (int p = blockIdx.x * ppb + pl; p < npix; p += gridDim.x * ppb) {
        int b, h, w;
        pix_decode(ym, p, b, h, w);
        const float4 d = *(const float4*)(dy + pix_off(ym, b, h, w) + c4 * 4);
        const float4 v = *(const float4*)(y + pix_off(om, b, h, w) + c4 * 4);
        *(float4*)(dst + pix_off(dm, b, h, w) + c4 * 4) =
            make_float4(v.x > 0.f ? d.x : 0.f, v.y > 0.f ? d.y : 0.f, v.z > 0.f ? d.z : 0.f, v.w > 0.f ? d.w : 0.f);
    }
}

// ------------------------------------------------------------------ 2x2 max-pool backward on padded planes
// (Keras MaxPooling2D behind a batch-stat BN, lm_and_am/model/cnn_ctc.py:108-131: unlike the frozen-BN cells of
// cells.hip there is no ReLU / affine to fuse here).  The first maximum in the order (0,0),(0,1),(1,0),(1,1) takes
// the gradient, as in cells.hip and oracle/nn.py.
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                          int B, int H, int W, int C, float* __restrict__ dx) {
    const int C4 = C >> 2, H2 = H >> 1, W2 = W >> 1;
    const int total = B * H2 * W2 * C4;
    const int WP = W + 1, HP = H + 1, WP2 = W2 + 1, HP2 = H2 + 1;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
        const int cg = idx % C4;
        int pix = idx / C4;
        const int w2 = pix % W2; pix /= W2;
        const int h2 = pix % H2;
        const int b = pix / H2;
        const long pin = ((long)b * HP + 2 * h2 + 1) * WP + 2 * w2 + 1;
        const float4 g = *(const float4*)(dy + (((long)b * HP2 + h2 + 1) * WP2 + w2 + 1) * C + cg * 4);
        const long offs[4] = {pin, pin + 1, pin + WP, pin + WP + 1};
        float4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = *(const float4*)(y + offs[k] * C + cg * 4);
        int ax = 0, ay = 0, az = 0, aw = 0;
        float mx = v[0].x, my = v[0].y, mz = v[0].z, mw = v[0].w;
#pragma unroll
        for (int k = 1; k < 4; ++k) {
            if (v[k].x > mx) { mx = v[k].x; ax = k; }
            if (v[k].y > my) { my = v[k].y; ay = k; }
            if (v[k].z > mz) { mz = v[k].z; az = k; }
            if (v[k].w > mw) { mw = v[k].w; aw = k; }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
            *(float4*)(dx + offs[k] * C + cg * 4) =
                make_float4(ax == k ? g.x : 0.f, ay == k ? g.y : 0.f, az == k ? g.z : 0.f, aw == k ? g.w : 0.f);
    }
}

// ------------------------------------------------------------------ stride-2 conv as a 2x2-tap conv on the phase-split plane
// W4[t = th*2+tw][(ph*2+pw)*Cin + c][n] = w[2*th+ph][2*tw+pw][c][n]  (zero where 2*th+ph or 2*tw+pw exceeds 2)
__global__ void s2_expand_kernel(const float* __restrict__ w, int Cin, int Cout, float* __restrict__ W4) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = 4L * 4 * Cin * Cout;
    if (i >= total) return;
    const int n = (int)(i % Cout);
    long r = i / Cout;
    const int c = (int)(r % Cin); r /= Cin;
    const int ph = (int)(r % 4) >> 1, pw = (int)(r % 4) & 1;
    const int t = (int)(r / 4), th = t >> 1, tw = t & 1;
    const int dh = 2 * th + ph, dw = 2 * tw + pw;
    W4[i] = (dh <= 2 && dw <= 2) ? w[((long)(dh * 3 + dw) * Cin + c) * Cout + n] : 0.f;
}

__global__ void s2_gather_kernel(const float* __restrict__ dW4, int Cin, int Cout, float* __restrict__ dw_out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = 9L * Cin * Cout;
    if (i >= total) return;
    const int n = (int)(i % Cout);
    long r = i / Cout;
    const int c = (int)(r % Cin);
    const int tap = (int)(r / Cin), dh = tap / 3, dw = tap % 3;
    const int th = dh >> 1, ph = dh & 1, tw = dw >> 1, pw = dw & 1;
    dw_out[i] = dW4[(((long)(th * 2 + tw) * 4 + ph * 2 + pw) * Cin + c) * Cout + n];
}

// ------------------------------------------------------------------ [B][H][W][c] plane slice <-> [B][H][CT][W]
constexpr int CT = 64;          // channels of the attention tensors
constexpr int WD = 80;          // frequency bins after the two stride-2 convs (4 * dimension / 4)

// one block per (b, h): 2-D transpose of [W][CT] through LDS
__global__ __launch_bounds__(256) void plane_to_T_kernel(const float* __restrict__ src, int H, int W, int ld, int choff,
                                                         float* __restrict__ dst) {
    __shared__ float tile[WD][CT + 1];
    const int b = blockIdx.x / H, h = blockIdx.x - b * H;
    const float* s = src + ((long)(b * (H + 1) + h + 1) * (W + 1) + 1) * ld + choff;
    for (int i = threadIdx.x; i < W * (CT / 4); i += 256) {
        const int w = i / (CT / 4), c4 = i - w * (CT / 4);
        const float4 v = *(const float4*)(s + (long)w * ld + c4 * 4);
        tile[w][c4 * 4 + 0] = v.x; tile[w][c4 * 4 + 1] = v.y; tile[w][c4 * 4 + 2] = v.z; tile[w][c4 * 4 + 3] = v.w;
    }
    __syncthreads();
    float* d = dst + (long)blockIdx.x * CT * W;
    for (int i = threadIdx.x; i < CT * W; i += 256) {
        const int c = i / W, w = i - c * W;
        d[i] = tile[w][c];
    }
}

// plane[..., choff + c] = srcA[b][h][c][w] (+ srcB)
__global__ __launch_bounds__(256) void T_to_plane_kernel(const float* __restrict__ srcA, const float* __restrict__ srcB,
                                                         int H, int W, int ld, int choff, float* __restrict__ dst) {
    __shared__ float tile[WD][CT + 1];
    const int b = blockIdx.x / H, h = blockIdx.x - b * H;
    const float* a = srcA + (long)blockIdx.x * CT * W;
    const float* bb = srcB ? srcB + (long)blockIdx.x * CT * W : nullptr;
    for (int i = threadIdx.x; i < CT * W; i += 256) {
        const int c = i / W, w = i - c * W;
        tile[w][c] = a[i] + (bb ? bb[i] : 0.f);
    }
    __syncthreads();
    float* d = dst + ((long)(b * (H + 1) + h + 1) * (W + 1) + 1) * ld + choff;
    for (int i = threadIdx.x; i < W * (CT / 4); i += 256) {
        const int w = i / (CT / 4), c4 = i - w * (CT / 4);
        *(float4*)(d + (long)w * ld + c4 * 4) = make_float4(tile[w][c4 * 4], tile[w][c4 * 4 + 1], tile[w][c4 * 4 + 2], tile[w][c4 * 4 + 3]);
    }
}

// ------------------------------------------------------------------ frequency-axis attention (model.py:242-256)
// Per (batch, channel) the operands are the [T][WD] matrices X_c[t][i] = X[b][t][c][i] (row pitch CT*WD).
//   gram:  G[i][j] = sum_t X[t][i] * Y[t][j]            (32x32x2 MFMA, contraction = time, 3x3 tiles of 32)
//          MODE 0: out = softmax_j(scale * G)            (the attention weights P)
//          MODE 1: out = P o (G - rowsum(G o P)) * scale (dS from dP = G; P given)
//   apply: Z[t][n] = sum_k X[t][k] * M[n][k]  (TRANS 0)  or  sum_k X[t][k] * M[k][n]  (TRANS 1)
constexpr int GP = 97;          // LDS pitch of the 96 x 96 score tile

template <int MODE>
__global__ __launch_bounds__(256) void gram_kernel(const float* __restrict__ X, const float* __restrict__ Y, int T,
                                                   float scale, const float* __restrict__ P, float* __restrict__ out) {
    __shared__ float S[96 * GP];
    const int b = blockIdx.x / CT, c = blockIdx.x - b * CT;
    const long base = (long)b * T * CT * WD + (long)c * WD;
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, lh = lane >> 5, wave = tid >> 6;
    // 9 output tiles over 4 waves: wave w owns tiles w, w+4, w+8 (row-major over (it, jt))
    floatx16 acc[3];
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
    const int ntile = (wave == 0) ? 3 : 2;
    for (int t0 = 0; t0 < T; t0 += 8) {
        float av[4][3], bv[4][3];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int t = t0 + 2 * u + lh;
            const float* xr = X + base + (long)t * CT * WD;
            const float* yr = Y + base + (long)t * CT * WD;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int tile = wave + 4 * k;
                const int it = tile / 3, jt = tile - it * 3;
                const int i = it * 32 + li, j = jt * 32 + li;
                av[u][k] = (k < ntile && t < T && i < WD) ? xr[i] : 0.f;
                bv[u][k] = (k < ntile && t < T && j < WD) ? yr[j] : 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int k = 0; k < 3; ++k)
                if (k < ntile) acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][k], bv[u][k], acc[k], 0, 0, 0);
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        if (k >= ntile) continue;
        const int tile = wave + 4 * k;
        const int it = tile / 3, jt = tile - it * 3;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = it * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            S[row * GP + jt * 32 + li] = acc[k][r];
        }
    }
    __syncthreads();
    float* o = out + (long)blockIdx.x * WD * WD;
    if (MODE == 0) {
        // softmax over j for each row i: 3 threads... one thread per row keeps it simple (80 x 80 values)
        if (tid < WD) {
            float* row = S + tid * GP;
            float mx = -INFINITY;
            for (int j = 0; j < WD; ++j) { row[j] *= scale; mx = fmaxf(mx, row[j]); }
            float sum = 0.f;
            for (int j = 0; j < WD; ++j) { row[j] = expf(row[j] - mx); sum += row[j]; }
            const float inv = 1.f / sum;
            for (int j = 0; j < WD; ++j) row[j] *= inv;
        }
    } else {
        const float* p = P + (long)blockIdx.x * WD * WD;
        if (tid < WD) {
            float* row = S + tid * GP;
            const float* pr = p + tid * WD;
            float dot = 0.f;
            for (int j = 0; j < WD; ++j) dot += row[j] * pr[j];
            for (int j = 0; j < WD; ++j) row[j] = pr[j] * (row[j] - dot) * scale;
        }
    }
    __syncthreads();
    for (int i = tid; i < WD * WD; i += 256) o[i] = S[(i / WD) * GP + (i % WD)];
}

template <int TRANS>
__global__ __launch_bounds__(256) void apply_kernel(const float* __restrict__ X, const float* __restrict__ M, int T,
                                                    float* __restrict__ Z) {
    // Ms[n][k] (B operand read as Ms[n*GP + k]); zero-padded to 96 x 96
    __shared__ float Ms[96 * GP];
    const int b = blockIdx.x / CT, c = blockIdx.x - b * CT;
    const long base = (long)b * T * CT * WD + (long)c * WD;
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, lh = lane >> 5, wave = tid >> 6;
    const float* m = M + (long)blockIdx.x * WD * WD;
    for (int i = tid; i < 96 * 96; i += 256) {
        const int n = i / 96, k = i - n * 96;
        float v = 0.f;
        if (n < WD && k < WD) v = TRANS ? m[k * WD + n] : m[n * WD + k];
        Ms[n * GP + k] = v;
    }
    __syncthreads();
    for (int t0 = wave * 32; t0 < T; t0 += 128) {
        floatx16 acc[3];
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
        const int t = t0 + li;
        const float* xr = X + base + (long)t * CT * WD;
#pragma unroll
        for (int g = 0; g < WD / 8; ++g) {
            float4 xv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (t < T) xv = *(const float4*)(xr + g * 8 + 4 * lh);
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) {
                const float a = (s2 == 0) ? xv.x : (s2 == 1) ? xv.y : (s2 == 2) ? xv.z : xv.w;
                const int k = g * 8 + 4 * lh + s2;
#pragma unroll
                for (int nt = 0; nt < 3; ++nt)
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, Ms[(nt * 32 + li) * GP + k], acc[nt], 0, 0, 0);
            }
        }
#pragma unroll
        for (int nt = 0; nt < 3; ++nt) {
            const int n = nt * 32 + li;
            if (n >= WD) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int tt = t0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (tt < T) Z[base + (long)tt * CT * WD + n] = acc[nt][r];
            }
        }
    }
}

// ------------------------------------------------------------------ residual add + LayerNorm over the channels of a plane
// 16 lanes own one pixel (4 channels each, C = 64); model.py:261 layer_norm(conv(...) + residual), eps 1e-8
__device__ __forceinline__ float sum16(float v) {
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__global__ __launch_bounds__(256) void pix_add_ln_fwd_kernel(const float* __restrict__ a, const float* __restrict__ r,
                                                             asr_pixmap m, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float eps,
                                                             float* __restrict__ y, float* __restrict__ xhat,
                                                             float* __restrict__ rstd_out) {
    const int c4 = threadIdx.x & 15, pl = threadIdx.x >> 4;
    const int npix = m.B * m.H * m.W;
    const float4 g = *(const float4*)(gamma + c4 * 4), be = *(const float4*)(beta + c4 * 4);
    for (int p0 = blockIdx.x * 16; p0 < npix; p0 += gridDim.x * 16) {
        const int p = p0 + pl;
        const bool ok = p < npix;
        int b = 0, h = 0, w = 0;
        if (ok) pix_decode(m, p, b, h, w);
        const long off = ok ? pix_off(m, b, h, w) + c4 * 4 : 0;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok) {
            v = *(const float4*)(a + off);
            const float4 rr = *(const float4*)(r + off);
            v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
        }
        const float mu = sum16(v.x + v.y + v.z + v.w) * (1.f / 64.f);
        const float4 d = make_float4(v.x - mu, v.y - mu, v.z - mu, v.w - mu);
        const float var = sum16(d.x * d.x + d.y * d.y + d.z * d.z + d.w * d.w) * (1.f / 64.f);
        const float rs = 1.f / sqrtf(var + eps);
        if (ok) {
            const float4 xh = make_float4(d.x * rs, d.y * rs, d.z * rs, d.w * rs);
            *(float4*)(xhat + off) = xh;
            *(float4*)(y + off) = make_float4(g.x * xh.x + be.x, g.y * xh.y + be.y, g.z * xh.z + be.z, g.w * xh.w + be.w);
            if (c4 == 0) rstd_out[p] = rs;
        }
    }
}

// dx = rstd * (g*dy - mean(g*dy) - xhat * mean(g*dy*xhat)); block partials of dgamma / dbeta: [blocks][128]
__global__ __launch_bounds__(256) void pix_ln_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ xhat,
                                                         const float* __restrict__ rstd, asr_pixmap m,
                                                         const float* __restrict__ gamma, float* __restrict__ dx,
                                                         float* __restrict__ partials) {
    __shared__ float red[16][128 + 4];
    const int c4 = threadIdx.x & 15, pl = threadIdx.x >> 4;
    const int npix = m.B * m.H * m.W;
    const float4 g = *(const float4*)(gamma + c4 * 4);
    float4 dg = make_float4(0.f, 0.f, 0.f, 0.f), db = dg;
    for (int p0 = blockIdx.x * 16; p0 < npix; p0 += gridDim.x * 16) {
        const int p = p0 + pl;
        const bool ok = p < npix;
        int b = 0, h = 0, w = 0;
        if (ok) pix_decode(m, p, b, h, w);
        const long off = ok ? pix_off(m, b, h, w) + c4 * 4 : 0;
        float4 d = make_float4(0.f, 0.f, 0.f, 0.f), xh = d;
        if (ok) { d = *(const float4*)(dy + off); xh = *(const float4*)(xhat + off); }
        dg.x += d.x * xh.x; dg.y += d.y * xh.y; dg.z += d.z * xh.z; dg.w += d.w * xh.w;
        db.x += d.x; db.y += d.y; db.z += d.z; db.w += d.w;
        const float4 gd = make_float4(g.x * d.x, g.y * d.y, g.z * d.z, g.w * d.w);
        const float m1 = sum16(gd.x + gd.y + gd.z + gd.w) * (1.f / 64.f);
        const float m2 = sum16(gd.x * xh.x + gd.y * xh.y + gd.z * xh.z + gd.w * xh.w) * (1.f / 64.f);
        if (ok) {
            const float rs = rstd[p];
            *(float4*)(dx + off) = make_float4(rs * (gd.x - m1 - xh.x * m2), rs * (gd.y - m1 - xh.y * m2),
                                               rs * (gd.z - m1 - xh.z * m2), rs * (gd.w - m1 - xh.w * m2));
        }
    }
    *(float4*)(&red[pl][c4 * 4]) = dg;
    *(float4*)(&red[pl][64 + c4 * 4]) = db;
    __syncthreads();
    if (threadIdx.x < 128) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) s += red[k][threadIdx.x];
        partials[(long)blockIdx.x * 128 + threadIdx.x] = s;
    }
}

inline int pix_blocks(long npix, int ppb) {
    long b = (npix + ppb - 1) / ppb;
    return (int)(b > 2048 ? 2048 : (b < 1 ? 1 : b));
}

inline bool map_ok(const asr_pixmap* m) {
    if (!m || m->B < 1 || m->H < 1 || m->W < 1 || m->C < 4 || (m->C & 3) || (m->ld & 3)) return false;
    if ((long)m->B * m->H * m->W >= (1L << 30)) return false;         // 32-bit pixel arithmetic in the kernels
    const int cpp = m->C / 4;
    if (cpp > 256 || (256 % cpp) != 0) return false;
    if (m->kind == 2) return (m->H % 2 == 0) && (m->W % 2 == 0) && m->ld >= 4 * m->C;
    return (m->kind == 0 || m->kind == 1) && m->ld >= m->C;
}

inline bool same_shape(const asr_pixmap* a, const asr_pixmap* b) {
    return a->B == b->B && a->H == b->H && a->W == b->W && a->C == b->C;
}

}  // namespace

// ===================================================================== C ABI
extern "C" int asr_prenet_conv1_fwd(const float* x, const float* w, const float* bias, int B, int T, int F,
                                    float* a1, void* stream) {
    if (!x || !w || !bias || !a1 || B < 1 || T < 2 || F < 2 || (T & 1) || (F & 1)) return ASR_ERR_BAD_ARG;
    if ((long)B * T * F >= (1L << 31)) return ASR_ERR_UNSUPPORTED;
    const long npix = (long)B * (T / 2) * (F / 2);
    hipLaunchKernelGGL(conv1_fwd_kernel, dim3(pix_blocks(npix, 16)), dim3(256), 0, (hipStream_t)stream, x, w, bias, B, T, F, a1);
    ASR_CHECK_LAUNCH("prenet_conv1_fwd");
    return ASR_OK;
}

static int conv1_bwd_blocks(int B, int T, int F) {
    const long npix = (long)B * (T / 2) * (F / 2);
    long b = (npix + 16 * 64 - 1) / (16 * 64);
    return (int)(b > 1024 ? 1024 : (b < 1 ? 1 : b));
}

extern "C" size_t asr_prenet_conv1_bwd_workspace(int B, int T, int F) {
    const int nb = conv1_bwd_blocks(B, T, F);
    return ((size_t)nb * 10 * CO + asr_reduce::colsum_tmp_floats(nb, 10 * CO)) * sizeof(float);
}

extern "C" int asr_prenet_conv1_bwd(const float* x, const float* dz, int B, int T, int F, float* dw, float* db,
                                    float* workspace, void* stream) {
    if (!x || !dz || !dw || !db || !workspace || B < 1 || T < 2 || F < 2 || (T & 1) || (F & 1)) return ASR_ERR_BAD_ARG;
    if ((long)B * T * F >= (1L << 31)) return ASR_ERR_UNSUPPORTED;
    const int nb = conv1_bwd_blocks(B, T, F);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(conv1_bwd_kernel, dim3(nb), dim3(256), 0, st, x, dz, B, T, F, workspace);
    ASR_CHECK_LAUNCH("prenet_conv1_bwd");
    asr_reduce::Multi m;
    m.nseg = 2; m.width[0] = 9 * CO; m.out[0] = dw; m.width[1] = CO; m.out[1] = db;
    m.width[2] = m.width[3] = 0; m.out[2] = m.out[3] = nullptr;
    return asr_reduce::colsum_multi(workspace, nb, 10 * CO, m, workspace + (size_t)nb * 10 * CO, st);
}

extern "C" size_t asr_bn_workspace(const asr_pixmap* m) {
    if (!map_ok(m)) return 0;
    const int ppb = 256 / (m->C / 4);
    const int nb = pix_blocks((long)m->B * m->H * m->W, ppb * 8);
    return (size_t)nb * 2 * m->C * sizeof(double) + 2 * m->C * sizeof(float) + 64;
}

extern "C" int asr_bn_stats(const float* src, const asr_pixmap* m, float eps, float* mean, float* rstd,
                            void* workspace, void* stream) {
    if (!src || !map_ok(m) || !mean || !rstd || !workspace) return ASR_ERR_BAD_ARG;
    const int ppb = 256 / (m->C / 4);
    const long npix = (long)m->B * m->H * m->W;
    const int nb = pix_blocks(npix, ppb * 8);
    hipStream_t st = (hipStream_t)stream;
    double* part = (double*)workspace;
    hipLaunchKernelGGL(bn_stats_kernel, dim3(nb), dim3(256), (size_t)ppb * 2 * m->C * sizeof(double), st, src, *m, part);
    hipLaunchKernelGGL(bn_stats_final_kernel, dim3(m->C), dim3(64), 0, st, (const double*)part, nb, m->C,
                       (double)npix, eps, mean, rstd);
    ASR_CHECK_LAUNCH("bn_stats");
    return ASR_OK;
}

namespace {
// moving statistics of a Keras BatchNormalization (keras/layers/normalization.py 2.3.1, call(): the batch variance is made
// unbiased with sample_size / (sample_size - (1 + epsilon)) before K.moving_average_update, which is
// variable -= (variable - value) * (1 - momentum)), and the inference-mode 1 / sqrt(moving_var + eps)
__global__ void bn_moving_kernel(const float* __restrict__ mean, const float* __restrict__ rstd, int C, float eps, float count,
                                 float momentum, float* __restrict__ mov_mean, float* __restrict__ mov_var,
                                 float* __restrict__ inf_rstd) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float mm = mov_mean[c], mv = mov_var[c];
    if (mean && rstd) {
        const float r = rstd[c];
        float var = 1.f / (r * r) - eps;
        if (var < 0.f) var = 0.f;
        var *= count / (count - (1.f + eps));
        mm -= (mm - mean[c]) * (1.f - momentum);
        mv -= (mv - var) * (1.f - momentum);
        mov_mean[c] = mm; mov_var[c] = mv;
    }
    if (inf_rstd) inf_rstd[c] = 1.f / sqrtf(mv + eps);
}
}  // namespace

extern "C" int asr_bn_moving(const float* mean, const float* rstd, int C, float eps, float count, float momentum,
                             float* mov_mean, float* mov_var, float* inf_rstd, void* stream) {
    if (!mov_mean || !mov_var || C < 1 || (!!mean != !!rstd) || (!mean && !inf_rstd)) return ASR_ERR_BAD_ARG;
    if (mean && !(count > 1.f + eps)) return ASR_ERR_BAD_ARG;
    hipLaunchKernelGGL(bn_moving_kernel, dim3(asr_cdiv(C, 64)), dim3(64), 0, (hipStream_t)stream, mean, rstd, C, eps, count, momentum,
                       mov_mean, mov_var, inf_rstd);
    ASR_CHECK_LAUNCH("bn_moving");
    return ASR_OK;
}

extern "C" int asr_bn_apply(const float* src, const asr_pixmap* sm, const float* mean, const float* rstd,
                            const float* gamma, const float* beta, const float* res, const asr_pixmap* rm, int relu,
                            float* dst, const asr_pixmap* dm, void* stream) {
    if (!src || !map_ok(sm) || !map_ok(dm) || !mean || !rstd || !gamma || !beta || !dst) return ASR_ERR_BAD_ARG;
    if (!same_shape(sm, dm) || (res && (!map_ok(rm) || !same_shape(sm, rm)))) return ASR_ERR_BAD_ARG;
    const int ppb = 256 / (sm->C / 4);
    const long npix = (long)sm->B * sm->H * sm->W;
    const asr_pixmap rmap = res ? *rm : *sm;
    hipLaunchKernelGGL(bn_apply_kernel, dim3(pix_blocks(npix, ppb * 4)), dim3(256), 0, (hipStream_t)stream, src, *sm, mean, rstd,
                       gamma, beta, res, rmap, relu, dst, *dm);
    ASR_CHECK_LAUNCH("bn_apply");
    return ASR_OK;
}

extern "C" int asr_bn_bwd(const float* dy, const asr_pixmap* ym, const float* a, const asr_pixmap* am,
                          const float* mean, const float* rstd, const float* gamma, int act,
                          float* dz, const asr_pixmap* zm, float* dgamma, float* dbeta, void* workspace, void* stream) {
    if (!dy || !a || !map_ok(ym) || !map_ok(am) || !map_ok(zm) || !mean || !rstd || !gamma || !dz || !dgamma || !dbeta ||
        !workspace) return ASR_ERR_BAD_ARG;
    if (!same_shape(ym, am) || !same_shape(zm, am) || act < 0 || act > 2) return ASR_ERR_BAD_ARG;
    const int C = am->C, ppb = 256 / (C / 4);
    const long npix = (long)am->B * am->H * am->W;
    const int nb = pix_blocks(npix, ppb * 8);
    hipStream_t st = (hipStream_t)stream;
    double* part = (double*)workspace;
    float* sums = (float*)(part + (size_t)nb * 2 * C);
    hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(nb), dim3(256), (size_t)ppb * 2 * C * sizeof(double), st, dy, *ym, a, *am,
                       mean, rstd, part);
    hipLaunchKernelGGL(bn_bwd_final_kernel, dim3(C), dim3(64), 0, st, (const double*)part, nb, C, (double)npix,
                       dgamma, dbeta, sums);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(pix_blocks(npix, ppb * 4)), dim3(256), 0, st, dy, *ym, a, *am, mean, rstd,
                       gamma, (const float*)sums, act, dz, *zm);
    ASR_CHECK_LAUNCH("bn_bwd");
    return ASR_OK;
}

extern "C" int asr_relu_mask(const float* dy, const asr_pixmap* ym, const float* y, const asr_pixmap* om,
                             float* dst, const asr_pixmap* dm, void* stream) {
    if (!dy || !y || !dst || !map_ok(ym) || !map_ok(om) || !map_ok(dm) || !same_shape(ym, om) || !same_shape(ym, dm))
        return ASR_ERR_BAD_ARG;
    const int ppb = 256 / (ym->C / 4);
    const long npix = (long)ym->B * ym->H * ym->W;
    hipLaunchKernelGGL(relu_mask_kernel, dim3(pix_blocks(npix, ppb * 4)), dim3(256), 0, (hipStream_t)stream, dy, *ym, y, *om, dst, *dm);
    ASR_CHECK_LAUNCH("relu_mask");
    return ASR_OK;
}

extern "C" int asr_maxpool_bwd(const float* dy, const float* y, int B, int H, int W, int C, float* dx, void* stream) {
    if (!dy || !y || !dx || B < 1 || H < 2 || W < 2 || (H & 1) || (W & 1) || C < 4 || (C & 3)) return ASR_ERR_BAD_ARG;
    const long total = (long)B * (H / 2) * (W / 2) * (C / 4);
    if (total >= (1L << 31)) return ASR_ERR_UNSUPPORTED;
    long nb = (total + 255) / 256;
    if (nb > 8192) nb = 8192;
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3((int)nb), dim3(256), 0, (hipStream_t)stream, dy, y, B, H, W, C, dx);
    ASR_CHECK_LAUNCH("maxpool_bwd");
    return ASR_OK;
}

extern "C" int asr_conv_s2_expand(const float* w, int Cin, int Cout, float* W4, void* stream) {
    if (!w || !W4 || Cin < 1 || Cout < 1) return ASR_ERR_BAD_ARG;
    const long total = 16L * Cin * Cout;
    hipLaunchKernelGGL(s2_expand_kernel, dim3(asr_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, w, Cin, Cout, W4);
    ASR_CHECK_LAUNCH("conv_s2_expand");
    return ASR_OK;
}

extern "C" int asr_conv_s2_gather(const float* dW4, int Cin, int Cout, float* dw, void* stream) {
    if (!dW4 || !dw || Cin < 1 || Cout < 1) return ASR_ERR_BAD_ARG;
    const long total = 9L * Cin * Cout;
    hipLaunchKernelGGL(s2_gather_kernel, dim3(asr_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, dW4, Cin, Cout, dw);
    ASR_CHECK_LAUNCH("conv_s2_gather");
    return ASR_OK;
}

extern "C" int asr_plane_to_T(const float* plane, int B, int H, int W, int ld, int choff, float* dst, void* stream) {
    if (!plane || !dst || B < 1 || H < 1 || W != WD || (ld & 3) || (choff & 3) || choff + CT > ld) return ASR_ERR_BAD_ARG;
    hipLaunchKernelGGL(plane_to_T_kernel, dim3(B * H), dim3(256), 0, (hipStream_t)stream, plane, H, W, ld, choff, dst);
    ASR_CHECK_LAUNCH("plane_to_T");
    return ASR_OK;
}

extern "C" int asr_T_to_plane(const float* srcA, const float* srcB, int B, int H, int W, int ld, int choff, float* plane,
                              void* stream) {
    if (!srcA || !plane || B < 1 || H < 1 || W != WD || (ld & 3) || (choff & 3) || choff + CT > ld) return ASR_ERR_BAD_ARG;
    hipLaunchKernelGGL(T_to_plane_kernel, dim3(B * H), dim3(256), 0, (hipStream_t)stream, srcA, srcB, H, W, ld, choff, plane);
    ASR_CHECK_LAUNCH("T_to_plane");
    return ASR_OK;
}

extern "C" int asr_freq_attention_fwd(const float* Q, const float* K, const float* V, int B, int T, float* P, float* O,
                                      void* stream) {
    if (!Q || !K || !V || !P || !O || B < 1 || T < 1) return ASR_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    const float scale = 1.0f / sqrtf((float)T);
    hipLaunchKernelGGL(gram_kernel<0>, dim3(B * CT), dim3(256), 0, st, Q, K, T, scale, (const float*)nullptr, P);
    hipLaunchKernelGGL(apply_kernel<0>, dim3(B * CT), dim3(256), 0, st, V, (const float*)P, T, O);
    ASR_CHECK_LAUNCH("freq_attention_fwd");
    return ASR_OK;
}

extern "C" int asr_freq_attention_bwd(const float* Q, const float* K, const float* V, const float* P, const float* dO,
                                      int B, int T, float* dQ, float* dK, float* dV, float* dS_ws, void* stream) {
    if (!Q || !K || !V || !P || !dO || !dQ || !dK || !dV || !dS_ws || B < 1 || T < 1) return ASR_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    const float scale = 1.0f / sqrtf((float)T);
    hipLaunchKernelGGL(apply_kernel<1>, dim3(B * CT), dim3(256), 0, st, dO, P, T, dV);                 // dV = dO . P
    hipLaunchKernelGGL(gram_kernel<1>, dim3(B * CT), dim3(256), 0, st, dO, V, T, scale, P, dS_ws);     // dS from dP = dO^T V
    hipLaunchKernelGGL(apply_kernel<0>, dim3(B * CT), dim3(256), 0, st, K, (const float*)dS_ws, T, dQ);  // dQ[t][i] = sum_j K[t][j] dS[i][j]
    hipLaunchKernelGGL(apply_kernel<1>, dim3(B * CT), dim3(256), 0, st, Q, (const float*)dS_ws, T, dK);  // dK[t][j] = sum_i Q[t][i] dS[i][j]
    ASR_CHECK_LAUNCH("freq_attention_bwd");
    return ASR_OK;
}

extern "C" int asr_pix_add_ln_fwd(const float* a, const float* r, const asr_pixmap* m, const float* gamma,
                                  const float* beta, float eps, float* y, float* xhat, float* rstd, void* stream) {
    if (!a || !r || !map_ok(m) || m->C != 64 || m->kind == 2 || !gamma || !beta || !y || !xhat || !rstd) return ASR_ERR_BAD_ARG;
    const long npix = (long)m->B * m->H * m->W;
    hipLaunchKernelGGL(pix_add_ln_fwd_kernel, dim3(pix_blocks(npix, 16 * 4)), dim3(256), 0, (hipStream_t)stream, a, r, *m, gamma,
                       beta, eps, y, xhat, rstd);
    ASR_CHECK_LAUNCH("pix_add_ln_fwd");
    return ASR_OK;
}

static int ln_bwd_blocks(const asr_pixmap* m) { return pix_blocks((long)m->B * m->H * m->W, 16 * 16); }

extern "C" size_t asr_pix_ln_bwd_workspace(const asr_pixmap* m) {
    if (!map_ok(m)) return 0;
    const int nb = ln_bwd_blocks(m);
    return ((size_t)nb * 128 + asr_reduce::colsum_tmp_floats(nb, 128)) * sizeof(float);
}

extern "C" int asr_pix_ln_bwd(const float* dy, const float* xhat, const float* rstd, const asr_pixmap* m,
                              const float* gamma, float* dx, float* dgamma, float* dbeta, float* workspace, void* stream) {
    if (!dy || !xhat || !rstd || !map_ok(m) || m->C != 64 || m->kind == 2 || !gamma || !dx || !dgamma || !dbeta || !workspace)
        return ASR_ERR_BAD_ARG;
    const int nb = ln_bwd_blocks(m);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(pix_ln_bwd_kernel, dim3(nb), dim3(256), 0, st, dy, xhat, rstd, *m, gamma, dx, workspace);
    ASR_CHECK_LAUNCH("pix_ln_bwd");
    asr_reduce::Multi mm;
    mm.nseg = 2; mm.width[0] = 64; mm.out[0] = dgamma; mm.width[1] = 64; mm.out[1] = dbeta;
    mm.width[2] = mm.width[3] = 0; mm.out[2] = mm.out[3] = nullptr;
    return asr_reduce::colsum_multi(workspace, nb, 128, mm, workspace + (size_t)nb * 128, st);
}

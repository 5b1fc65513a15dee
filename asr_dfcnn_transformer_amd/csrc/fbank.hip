// Log-mel filterbank features + per-utterance column standardisation on device, float64
// arithmetic (the reference runs python_speech_features + sklearn in float64 on the host
// and only casts to float32 when feeding the model: util/wav_util.py:22-31, train.py:41).
//
// Kernel 1: one wave per 25 ms frame: pre-emphasis on load, 512-point radix-2 FFT in LDS,
//           power spectrum, banded mel filterbank (353 non-zeros for nfilt = 200), log.
// Kernel 2: per (utterance, 32 columns): mean / population-std over time in fixed order,
//           sklearn.preprocessing.scale's centre -> divide -> re-centre sequence, float32
//           output zero-padded to t_pad rows.
// HBM-bound (0.64 MB in, 0.8 MB out per 10 s utterance); the float64 math is noise.
#include "asr_common.h"
#include <math.h>
#include <stdlib.h>

namespace {

__device__ __forceinline__ int num_frames_of(int ns, int frame_len, int frame_step) {
    if (ns <= frame_len) return 1;
    return 1 + (ns - frame_len + frame_step - 1) / frame_step;
}

__global__ __launch_bounds__(256) void fbank_logmel_kernel(const float* __restrict__ signal, const int32_t* __restrict__ nsamples,
                                                           int max_samples, int frame_len, int frame_step, int nfft, int log2n,
                                                           double preemph, int nfilt, const int32_t* __restrict__ fb_start,
                                                           const int32_t* __restrict__ fb_count, const double* __restrict__ fb_weight,
                                                           int fb_width, const double* __restrict__ twiddle,
                                                           double* __restrict__ logfb, int max_frames, int32_t* __restrict__ frames_out) {
    extern __shared__ __attribute__((aligned(16))) double smd[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y;
    const int ns = nsamples[b];
    int nf = num_frames_of(ns, frame_len, frame_step);
    if (nf > max_frames) nf = max_frames;
    if (blockIdx.x == 0 && tid == 0) frames_out[b] = nf;
    const int f = blockIdx.x * 4 + wave;
    const bool active = f < nf;
    double* re = smd + (size_t)wave * (2 * nfft + nfft / 2 + 2);
    double* im = re + nfft;
    double* ps = im + nfft;
    const float* sig = signal + (long)b * max_samples;

    for (int i = lane; i < nfft; i += 64) {
        double v = 0.0;
        const long idx = (long)f * frame_step + i;
        if (active && i < frame_len && idx < ns) {
            const double x = (double)sig[idx];
            v = (idx == 0) ? x : x - preemph * (double)sig[idx - 1];
        }
        const int r = (int)(__brev((unsigned)i) >> (32 - log2n));
        re[r] = v; im[r] = 0.0;
    }
    __syncthreads();
    for (int s = 1; s <= log2n; ++s) {
        const int half = 1 << (s - 1);
        const int tstep = nfft >> s;
        for (int j = lane; j < nfft / 2; j += 64) {
            const int grp = j >> (s - 1), k = j & (half - 1);
            const int i0 = (grp << s) + k, i1 = i0 + half;
            const double wr = twiddle[2 * (k * tstep)], wi = twiddle[2 * (k * tstep) + 1];
            const double xr = re[i1], xi = im[i1];
            const double vr = xr * wr - xi * wi, vi = xr * wi + xi * wr;
            const double ur = re[i0], ui = im[i0];
            re[i0] = ur + vr; im[i0] = ui + vi;
            re[i1] = ur - vr; im[i1] = ui - vi;
        }
        __syncthreads();
    }
    const double invn = 1.0 / (double)nfft;
    for (int k = lane; k <= nfft / 2; k += 64) ps[k] = (re[k] * re[k] + im[k] * im[k]) * invn;
    __syncthreads();
    if (active) {
        double* o = logfb + ((long)b * max_frames + f) * nfilt;
        for (int j = lane; j < nfilt; j += 64) {
            const int st = fb_start[j], cnt = fb_count[j];
            double e = 0.0;
            for (int i = 0; i < cnt; ++i) e = fma(ps[st + i], fb_weight[(long)j * fb_width + i], e);
            if (e == 0.0) e = 2.220446049250313e-16;
            o[j] = log(e);
        }
    }
}

__global__ __launch_bounds__(256) void fbank_scale_kernel(const double* __restrict__ logfb, const int32_t* __restrict__ frames,
                                                          int max_frames, int nfilt, float* __restrict__ out, int t_pad) {
    __shared__ double red[8][32];
    const int tid = threadIdx.x, cl = tid & 31, ts = tid >> 5;
    const int b = blockIdx.y;
    const int col = blockIdx.x * 32 + cl;
    const bool ok = col < nfilt;
    int nf = frames[b];
    if (nf > t_pad) nf = t_pad;
    const double* x = logfb + (long)b * max_frames * nfilt + col;
    const double dn = (double)nf;

    auto block_sum = [&](double v) -> double {
        red[ts][cl] = v;
        __syncthreads();
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < 8; ++k) s += red[k][cl];
        __syncthreads();
        return s;
    };

    double acc = 0.0;
    if (ok) for (int t = ts; t < nf; t += 8) acc += x[(long)t * nfilt];
    const double mean = block_sum(acc) / dn;
    double a2 = 0.0, a1 = 0.0;
    if (ok) for (int t = ts; t < nf; t += 8) { const double d = x[(long)t * nfilt] - mean; a2 += d * d; a1 += d; }
    const double var = block_sum(a2) / dn;
    double mean1 = block_sum(a1) / dn;
    if (!(fabs(mean1) > 1e-8)) mean1 = 0.0;           // sklearn re-centres only when not allclose(mean_1, 0)
    double sd = sqrt(var);
    if (sd < 10.0 * 2.220446049250313e-16) sd = 1.0;  // _handle_zeros_in_scale
    double a3 = 0.0;
    if (ok) for (int t = ts; t < nf; t += 8) a3 += (x[(long)t * nfilt] - mean - mean1) / sd;
    const double mean2 = block_sum(a3) / dn;
    if (ok) {
        float* o = out + (long)b * t_pad * nfilt + col;
        for (int t = ts; t < nf; t += 8) o[(long)t * nfilt] = (float)((x[(long)t * nfilt] - mean - mean1) / sd - mean2);
        for (int t = nf + ts; t < t_pad; t += 8) o[(long)t * nfilt] = 0.f;
    }
}

// ---- v2 (nfft = 512, the only size the reference uses): the frame is real, so its spectrum comes from ONE 256-point
// complex FFT of z[n] = x[2n] + i x[2n+1] plus an untangling pass -- half the butterflies -- run as four radix-4 Stockham
// passes (natural order out, no bit reversal): every lane does one radix-4 butterfly per pass, 4 LDS exchanges per frame
// instead of 9, conflict-free 16-byte reads, and NO workgroup barrier: a wave owns its frame, and the LDS executes one
// wave's requests in order.  The 256 twiddles sit in LDS once per workgroup; a wave handles FPW frames in a row.
// Still float64 throughout (python_speech_features computes in float64): results equal the radix-2 kernel's to ~1e-15
// relative, i.e. the same float32 features.
struct dcomplex { double x, y; };

__device__ __forceinline__ void wave_lds_sync() {
    // same-wave LDS hand-over: ordering is the hardware's (DS ops of one wave execute in issue order); this only stops
    // the compiler from moving LDS accesses across it
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int FPW>
__global__ __launch_bounds__(256) void fbank_logmel_v2_kernel(const float* __restrict__ signal, const int32_t* __restrict__ nsamples,
                                                              int max_samples, int frame_len, int frame_step, double preemph, int nfilt,
                                                              const int32_t* __restrict__ fb_start, const int32_t* __restrict__ fb_count,
                                                              const double* __restrict__ fb_weight, int fb_width,
                                                              const double* __restrict__ twiddle, double* __restrict__ logfb,
                                                              int max_frames, int32_t* __restrict__ frames_out) {
    constexpr int N = 256;                                    // complex FFT length = nfft / 2
    __shared__ __attribute__((aligned(16))) dcomplex tw[N];   // W_512^k = exp(-2 pi i k / 512), k < 256
    __shared__ __attribute__((aligned(16))) dcomplex buf[4][2][N];
    extern __shared__ __attribute__((aligned(16))) double fbw[];      // [nfilt][fb_width]: the banded filterbank, once per workgroup
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y;
    const int ns = nsamples[b];
    int nf = num_frames_of(ns, frame_len, frame_step);
    if (nf > max_frames) nf = max_frames;
    if (blockIdx.x == 0 && tid == 0) frames_out[b] = nf;
    tw[tid].x = twiddle[2 * tid]; tw[tid].y = twiddle[2 * tid + 1];
    for (int i = tid; i < nfilt * fb_width; i += 256) fbw[i] = fb_weight[i];
    // This lane's filters do not change from frame to frame.  Round 5: the filters that HAVE taps are dealt to the lanes densely (lane,
    // lane + 64, ... of their ascending list) -- 157 of 200 at nfilt 200, i.e. three groups of 64 instead of four: a float64 log per
    // group and frame is half of this kernel's vector work -- and the empty ones only get their constant stored.
    constexpr int JMAX = 4;                                            // nfilt <= 256
    __shared__ short nzl[256], ezl[256];
    __shared__ int wcnt[2][4];
    const bool isf = tid < nfilt;
    const bool nzf = isf && fb_count[tid] > 0;
    {
        const unsigned long long mnz = __ballot(nzf), mez = __ballot(isf && !nzf);
        if (lane == 0) { wcnt[0][wave] = __popcll(mnz); wcnt[1][wave] = __popcll(mez); }
        __syncthreads();
        int onz = 0, oez = 0;
        for (int w = 0; w < wave; ++w) { onz += wcnt[0][w]; oez += wcnt[1][w]; }
        const unsigned long long below = (1ull << lane) - 1ull;
        if (nzf) nzl[onz + __popcll(mnz & below)] = (short)tid;
        else if (isf) ezl[oez + __popcll(mez & below)] = (short)tid;
    }
    __syncthreads();
    const int nnz = wcnt[0][0] + wcnt[0][1] + wcnt[0][2] + wcnt[0][3], nez = wcnt[1][0] + wcnt[1][1] + wcnt[1][2] + wcnt[1][3];
    int fj[JMAX], fst[JMAX], fcn[JMAX];
#pragma unroll
    for (int r = 0; r < JMAX; ++r) {
        const int idx = lane + 64 * r;
        fj[r] = idx < nnz ? (int)nzl[idx] : -1;
        fst[r] = fj[r] >= 0 ? fb_start[fj[r]] : 0;
        fcn[r] = fj[r] >= 0 ? fb_count[fj[r]] : 0;
    }
    const double log_eps = log(2.220446049250313e-16);                 // what an empty filter yields (43 of 200 at nfilt 200)
    __syncthreads();
    const float* sig = signal + (long)b * max_samples;
    dcomplex* A = buf[wave][0];
    dcomplex* Bf = buf[wave][1];
    double* ps = (double*)Bf;                                 // the power spectrum reuses the buffer the last pass left free

    auto twid = [&](int idx) -> dcomplex {                    // W_512^idx for 0 <= idx < 512
        dcomplex w = tw[idx & 255];
        if (idx & 256) { w.x = -w.x; w.y = -w.y; }
        return w;
    };

    // framing: lane l packs samples (2n, 2n+1), n = l, l+64, l+128, l+192, into z[n].  All 16 loads of a lane are
    // unconditional (clamped addresses) and issued together, and the NEXT frame's samples are fetched while this one is
    // transformed (a wave walks its FPW frames one after the other)
    float cur[8], prv[8], ncur[8], nprv[8];
    auto fetch = [&](int f, float (&c)[8], float (&pv)[8]) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                long idx = (long)f * frame_step + 2 * (lane + 64 * r) + h;
                if (idx > max_samples - 1) idx = max_samples - 1;
                c[2 * r + h] = sig[idx];
                pv[2 * r + h] = sig[idx > 0 ? idx - 1 : 0];
            }
    };
    const int f0 = (blockIdx.x * 4 + wave) * FPW;
    fetch(f0, ncur, nprv);
    for (int q = 0; q < FPW; ++q) {
        const int f = f0 + q;
        if (f >= nf) break;                                   // wave-uniform
#pragma unroll
        for (int i = 0; i < 8; ++i) { cur[i] = ncur[i]; prv[i] = nprv[i]; }
        if (q + 1 < FPW) fetch(f + 1, ncur, nprv);
        // z[n], n = lane + 64 r, are exactly the four inputs of this lane's first butterfly: pass 0 runs from registers
        dcomplex zin[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = lane + 64 * r;
            double v[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int i = 2 * n + h;
                const long idx = (long)f * frame_step + i;
                const double x = (idx > 0) ? (double)cur[2 * r + h] - preemph * (double)prv[2 * r + h] : (double)cur[2 * r + h];
                v[h] = (i < frame_len && idx < ns) ? x : 0.0;
            }
            zin[r].x = v[0]; zin[r].y = v[1];
        }
        // four radix-4 Stockham passes, Ns = 1, 4, 16, 64.  The buffer between pass 0 and pass 1 is XOR-swizzled inside
        // aligned groups of four elements (element p sits at (p & ~3) | ((p ^ (p >> 3)) & 3)): pass 0 writes with a
        // 4-element lane stride, which would be a 4-way bank conflict on 16-byte stores; pass 1 reads consecutive elements,
        // for which a permutation inside 64-byte groups changes nothing.
        dcomplex* src = A; dcomplex* dst = Bf;
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            const int Ns = 1 << (2 * pass);
            const int k = lane & (Ns - 1);
            const int tstep = 128 >> (2 * pass);              // W_{4 Ns}^{k} = W_512^{k * 128 / Ns}
            dcomplex v0, v1, v2, v3;
            if (pass == 0) { v0 = zin[0]; v1 = zin[1]; v2 = zin[2]; v3 = zin[3]; }
            else if (pass == 1) {
                const int sw = (lane >> 3) & 3;               // (p >> 3) & 3 for p = lane + 64 r
                const int q = (lane & ~3) | ((lane ^ sw) & 3);
                v0 = src[q]; v1 = src[q + 64]; v2 = src[q + 128]; v3 = src[q + 192];
            } else { v0 = src[lane]; v1 = src[lane + 64]; v2 = src[lane + 128]; v3 = src[lane + 192]; }
            if (pass > 0) {
                const dcomplex w1 = twid(k * tstep), w2 = twid(2 * k * tstep), w3 = twid(3 * k * tstep);
                dcomplex t;
                t.x = v1.x * w1.x - v1.y * w1.y; t.y = v1.x * w1.y + v1.y * w1.x; v1 = t;
                t.x = v2.x * w2.x - v2.y * w2.y; t.y = v2.x * w2.y + v2.y * w2.x; v2 = t;
                t.x = v3.x * w3.x - v3.y * w3.y; t.y = v3.x * w3.y + v3.y * w3.x; v3 = t;
            }
            const double ax = v0.x + v2.x, ay = v0.y + v2.y, bx = v0.x - v2.x, by = v0.y - v2.y;
            const double cx = v1.x + v3.x, cy = v1.y + v3.y;
            const double dx = v1.y - v3.y, dy = -(v1.x - v3.x);          // (v1 - v3) * (-i)
            const int j0 = ((lane >> (2 * pass)) << (2 * pass + 2)) + k;
            dcomplex o;
            if (pass == 0) {                                   // positions 4 lane + r, swizzled: r ^ ((lane >> 1) & 3)
                const int sw = (lane >> 1) & 3;
                o.x = ax + cx; o.y = ay + cy; dst[j0 + (0 ^ sw)] = o;
                o.x = bx + dx; o.y = by + dy; dst[j0 + (1 ^ sw)] = o;
                o.x = ax - cx; o.y = ay - cy; dst[j0 + (2 ^ sw)] = o;
                o.x = bx - dx; o.y = by - dy; dst[j0 + (3 ^ sw)] = o;
            } else {
                o.x = ax + cx; o.y = ay + cy; dst[j0] = o;
                o.x = bx + dx; o.y = by + dy; dst[j0 + Ns] = o;
                o.x = ax - cx; o.y = ay - cy; dst[j0 + 2 * Ns] = o;
                o.x = bx - dx; o.y = by - dy; dst[j0 + 3 * Ns] = o;
            }
            wave_lds_sync();
            dcomplex* t2 = src; src = dst; dst = t2;
        }
        // src = Z (256-point spectrum of z).  Untangle: X[k] = E[k] + W_512^k O[k], E = (Z[k] + conj Z[N-k]) / 2,
        // O = (Z[k] - conj Z[N-k]) / (2 i); power spectrum |X|^2 / 512 for k = 0 .. 256
        const double invn = 1.0 / 512.0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k = lane + 64 * r;
            const dcomplex zk = src[k], zc = src[(N - k) & (N - 1)];
            const double ex = 0.5 * (zk.x + zc.x), ey = 0.5 * (zk.y - zc.y);
            const double ox = 0.5 * (zk.y + zc.y), oy = -0.5 * (zk.x - zc.x);
            const dcomplex w = tw[k];
            const double xr = ex + (w.x * ox - w.y * oy), xi = ey + (w.x * oy + w.y * ox);
            ps[k] = (xr * xr + xi * xi) * invn;
            if (k == 0) { const double nr = ex - ox, ni = ey - oy; ps[N] = (nr * nr + ni * ni) * invn; }
        }
        wave_lds_sync();
        double* o = logfb + ((long)b * max_frames + f) * nfilt;
#pragma unroll
        for (int r = 0; r < JMAX; ++r) {
            if (64 * r >= nnz) break;                          // (uniform) no filter in this group
            const int j = fj[r];
            if (j >= 0) {
                // the table rows are zero beyond a filter's own taps: a zero weight leaves e unchanged (fma(x, 0, e) == e),
                // so all fb_width taps are taken, with the bin index clamped to the last one
                const double* wrow = fbw + j * fb_width;
                double e = 0.0;
                for (int i = 0; i < fcn[r]; ++i) e = fma(ps[fst[r] + i], wrow[i], e);
                o[j] = (e == 0.0) ? log_eps : log(e);
            }
        }
        for (int idx = lane; idx < nez; idx += 64) o[ezl[idx]] = log_eps;
        wave_lds_sync();                                       // ps / A are rewritten by the next frame
    }
}

// scale v2: one workgroup per (utterance, 8 columns): 8 columns x 32 time slices; a thread keeps its <= RMAX rows of the
// column in registers, so the float64 log-energies are read ONCE and the four sklearn passes (mean, variance + first-order
// correction, re-centre, write) run from registers.  Same arithmetic and summation order per slice as the v1 kernel.
// (round 5: SIXTEEN columns per workgroup, 512 threads -- a row segment is a whole 128-byte line instead of half of one that a second
//  workgroup fetched again: 38 -> ~25 us; per column the same slices, rows and summation order as with eight)
template <int RMAX>
__global__ __launch_bounds__(512) void fbank_scale_v2_kernel(const double* __restrict__ logfb, const int32_t* __restrict__ frames,
                                                             int max_frames, int nfilt, float* __restrict__ out, int t_pad) {
    __shared__ double red[32][16];
    const int tid = threadIdx.x, cl = tid & 15, ts = tid >> 4;
    const int b = blockIdx.y;
    const int col = blockIdx.x * 16 + cl;
    const bool ok = col < nfilt;
    int nf = frames[b];
    if (nf > t_pad) nf = t_pad;
    const double* x = logfb + (long)b * max_frames * nfilt + col;
    const double dn = (double)nf;
    double v[RMAX];
#pragma unroll
    for (int i = 0; i < RMAX; ++i) {
        const int t = ts + 32 * i;
        v[i] = (ok && t < nf) ? x[(long)t * nfilt] : 0.0;
    }
    auto block_sum = [&](double s) -> double {
        red[ts][cl] = s;
        __syncthreads();
        double tot = 0.0;
#pragma unroll
        for (int k = 0; k < 32; ++k) tot += red[k][cl];
        __syncthreads();
        return tot;
    };
    double acc = 0.0;
#pragma unroll
    for (int i = 0; i < RMAX; ++i) if (ts + 32 * i < nf) acc += v[i];
    const double mean = block_sum(acc) / dn;
    double a2 = 0.0, a1 = 0.0;
#pragma unroll
    for (int i = 0; i < RMAX; ++i) if (ts + 32 * i < nf) { const double d = v[i] - mean; a2 += d * d; a1 += d; }
    const double var = block_sum(a2) / dn;
    double mean1 = block_sum(a1) / dn;
    if (!(fabs(mean1) > 1e-8)) mean1 = 0.0;
    double sd = sqrt(var);
    if (sd < 10.0 * 2.220446049250313e-16) sd = 1.0;
    double a3 = 0.0;
#pragma unroll
    for (int i = 0; i < RMAX; ++i) if (ts + 32 * i < nf) { v[i] = (v[i] - mean - mean1) / sd; a3 += v[i]; }
    const double mean2 = block_sum(a3) / dn;
    if (ok) {
        float* o = out + (long)b * t_pad * nfilt + col;
#pragma unroll
        for (int i = 0; i < RMAX; ++i) {
            const int t = ts + 32 * i;
            if (t < nf) o[(long)t * nfilt] = (float)(v[i] - mean2);
        }
        for (int t = nf + ts; t < t_pad; t += 32) o[(long)t * nfilt] = 0.f;
    }
}

// Low-frame-rate stacking (util/utils.py:7-31): out[b][i][j*D + d] = feat[b][min(i*n + j, frames[b] - 1)][d] for
// i < ceil(frames[b] / n), zero rows after that.  One thread per output float4; pure gather, HBM-bound.
__global__ __launch_bounds__(256) void lfr_kernel(const float* __restrict__ feat, const int32_t* __restrict__ frames,
                                                  int t_pad, int D, int m, int n, int t_out, float* __restrict__ out, long total4) {
    const long i4 = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i4 >= total4) return;
    const int row4 = (m * D) >> 2;
    const long r = i4 / row4;
    const int c = (int)(i4 - r * row4) << 2;
    const int b = (int)(r / t_out), i = (int)(r - (long)b * t_out);
    const int nf = frames[b];
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if ((long)i * n < nf) {
        const int j = c / D, d = c - j * D;
        int t = i * n + j;
        if (t > nf - 1) t = nf - 1;
        v = *(const float4*)(feat + ((long)b * t_pad + t) * D + d);
    }
    *(float4*)(out + i4 * 4) = v;
}

}  // namespace

extern "C" int asr_lfr(const float* feat, const int32_t* frames, int B, int t_pad, int D, int m, int n, int t_out,
                       float* out, void* stream) {
    if (!feat || !frames || !out) return ASR_ERR_BAD_ARG;
    if (B < 1 || t_pad < 1 || D < 4 || (D & 3) || m < 1 || n < 1 || t_out < 1) return ASR_ERR_BAD_ARG;
    const long total4 = (long)B * t_out * m * D / 4;
    hipLaunchKernelGGL(lfr_kernel, dim3(asr_cdiv(total4, 256)), dim3(256), 0, (hipStream_t)stream, feat, frames, t_pad, D, m, n,
                       t_out, out, total4);
    ASR_CHECK_LAUNCH("lfr");
    return ASR_OK;
}

extern "C" int asr_fbank(const float* signal, const int32_t* nsamples, int B, int max_samples,
                         int frame_len, int frame_step, int nfft, double preemph, int nfilt,
                         const int32_t* fb_start, const int32_t* fb_count, const double* fb_weight, int fb_width,
                         const double* twiddle, double* logfb, int max_frames,
                         float* out, int t_pad, int32_t* frames, void* stream) {
    if (!signal || !nsamples || !fb_start || !fb_count || !fb_weight || !twiddle || !logfb || !out || !frames)
        return ASR_ERR_BAD_ARG;
    if (B < 1 || max_samples < 1 || frame_len < 1 || frame_step < 1 || nfilt < 1 || max_frames < 1 || t_pad < 1)
        return ASR_ERR_BAD_ARG;
    int log2n = 0;
    while ((1 << log2n) < nfft) ++log2n;
    if ((1 << log2n) != nfft || nfft < 128 || nfft > 2048 || frame_len > nfft) return ASR_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    const size_t lds = (size_t)4 * (2 * nfft + nfft / 2 + 2) * sizeof(double);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)fbank_logmel_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    const size_t lds2 = (size_t)nfilt * fb_width * sizeof(double);
    if (nfft == 512 && frame_len <= 512 && nfilt <= 256 && lds2 <= 96 * 1024) {
        constexpr int FPW = 5;
        static bool attr2 = false;
        if (!attr2) {
            (void)hipFuncSetAttribute((const void*)fbank_logmel_v2_kernel<FPW>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
            attr2 = true;
        }
        hipLaunchKernelGGL(fbank_logmel_v2_kernel<FPW>, dim3(asr_cdiv(max_frames, 4 * FPW), B), dim3(256), lds2, st, signal, nsamples,
                           max_samples, frame_len, frame_step, preemph, nfilt, fb_start, fb_count, fb_weight, fb_width, twiddle,
                           logfb, max_frames, frames);
    } else {
        hipLaunchKernelGGL(fbank_logmel_kernel, dim3(asr_cdiv(max_frames, 4), B), dim3(256), lds, st, signal, nsamples, max_samples,
                           frame_len, frame_step, nfft, log2n, preemph, nfilt, fb_start, fb_count, fb_weight, fb_width, twiddle,
                           logfb, max_frames, frames);
    }
    const int tmax = max_frames < t_pad ? max_frames : t_pad;          // rows a column can have
    if (tmax <= 32 * 32) {
        hipLaunchKernelGGL(fbank_scale_v2_kernel<32>, dim3(asr_cdiv(nfilt, 16), B), dim3(512), 0, st, (const double*)logfb,
                           (const int32_t*)frames, max_frames, nfilt, out, t_pad);
    } else if (tmax <= 32 * 64) {
        hipLaunchKernelGGL(fbank_scale_v2_kernel<64>, dim3(asr_cdiv(nfilt, 16), B), dim3(512), 0, st, (const double*)logfb,
                           (const int32_t*)frames, max_frames, nfilt, out, t_pad);
    } else {
        hipLaunchKernelGGL(fbank_scale_kernel, dim3(asr_cdiv(nfilt, 32), B), dim3(256), 0, st, (const double*)logfb,
                           (const int32_t*)frames, max_frames, nfilt, out, t_pad);
    }
    ASR_CHECK_LAUNCH("fbank");
    return ASR_OK;
}

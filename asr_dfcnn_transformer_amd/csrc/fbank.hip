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
    hipLaunchKernelGGL(fbank_logmel_kernel, dim3(asr_cdiv(max_frames, 4), B), dim3(256), lds, st, signal, nsamples, max_samples,
                       frame_len, frame_step, nfft, log2n, preemph, nfilt, fb_start, fb_count, fb_weight, fb_width, twiddle,
                       logfb, max_frames, frames);
    hipLaunchKernelGGL(fbank_scale_kernel, dim3(asr_cdiv(nfilt, 32), B), dim3(256), 0, st, (const double*)logfb,
                       (const int32_t*)frames, max_frames, nfilt, out, t_pad);
    ASR_CHECK_LAUNCH("fbank");
    return ASR_OK;
}

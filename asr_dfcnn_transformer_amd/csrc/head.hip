// Loss head of the acoustic model: softmax -> log(p + eps) (time-major), CTC loss and
// gradient (alpha/beta in float64, one workgroup per utterance, lattice inputs gathered
// into LDS once), greedy decoding, normalised edit distance, and the TF-form Adam update.
// All of these are HBM/latency-bound row kernels: one wave per (t, b) row, wave-level
// reductions, no MFMA.
#include "asr_common.h"
#include <math.h>
#include <type_traits>

namespace {

// ------------------------------------------------------------------ softmax + log, fwd / bwd
__global__ __launch_bounds__(256) void softmax_log_fwd_kernel(const float* __restrict__ d, int B, int T, int V,
                                                              float eps, float* __restrict__ out) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);      // row = b*T + t
    if (row >= B * T) return;
    const int lane = threadIdx.x & 63;
    const int b = row / T, t = row - b * T;
    const float* x = d + (long)row * V;
    float m = -INFINITY;
    for (int k = lane; k < V; k += 64) m = fmaxf(m, x[k]);
    m = asr_wave_max(m);
    float s = 0.f;
    for (int k = lane; k < V; k += 64) s += expf(x[k] - m);
    s = asr_wave_sum(s);
    const float inv = 1.f / s;
    float* o = out + ((long)t * B + b) * V;
    for (int k = lane; k < V; k += 64) o[k] = logf(expf(x[k] - m) * inv + eps);
}

__global__ __launch_bounds__(256) void softmax_log_bwd_kernel(const float* __restrict__ li, const float* __restrict__ g,
                                                              int B, int T, int V, float eps, float gscale,
                                                              float* __restrict__ dd) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);      // row = t*B + b
    if (row >= B * T) return;
    const int lane = threadIdx.x & 63;
    const int t = row / B, b = row - t * B;
    const float* l = li + (long)row * V;
    const float* gr = g + (long)row * V;
    float inner = 0.f;
    for (int k = lane; k < V; k += 64) {
        const float pe = expf(l[k]);               // p + eps
        inner += (pe - eps) * (gr[k] / pe);
    }
    inner = asr_wave_sum(inner);
    float* o = dd + ((long)b * T + t) * V;
    for (int k = lane; k < V; k += 64) {
        const float pe = expf(l[k]);
        o[k] = (pe - eps) * (gr[k] / pe - inner) * gscale;
    }
}

// The same two passes for rows of up to 64 x 4 x NV floats (V % 4 == 0): a lane keeps its NV float4 of the row in registers -- one
// 16-byte load per 16 bytes of input instead of three (forward) / two (backward) 4-byte loads; the per-element formulas are the ones
// above, the row sums are taken in another (fixed) order.
template <int NV>
__global__ __launch_bounds__(256) void softmax_log_fwd_vec_kernel(const float* __restrict__ d, int B, int T, int V,
                                                                  float eps, float* __restrict__ out) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);      // row = b*T + t
    if (row >= B * T) return;
    const int lane = threadIdx.x & 63;
    const int b = row / T, t = row - b * T;
    const int n4 = V >> 2;
    const float4* x4 = (const float4*)(d + (long)row * V);
    float4 v[NV];
    float m = -INFINITY;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int k4 = lane + 64 * i;
        v[i] = k4 < n4 ? x4[k4] : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        m = fmaxf(fmaxf(m, fmaxf(v[i].x, v[i].y)), fmaxf(v[i].z, v[i].w));
    }
    m = asr_wave_max(m);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        v[i].x = expf(v[i].x - m); v[i].y = expf(v[i].y - m); v[i].z = expf(v[i].z - m); v[i].w = expf(v[i].w - m);
        s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    s = asr_wave_sum(s);
    const float inv = 1.f / s;
    float4* o4 = (float4*)(out + ((long)t * B + b) * V);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int k4 = lane + 64 * i;
        if (k4 < n4) o4[k4] = make_float4(logf(v[i].x * inv + eps), logf(v[i].y * inv + eps), logf(v[i].z * inv + eps), logf(v[i].w * inv + eps));
    }
}

template <int NV>
__global__ __launch_bounds__(256) void softmax_log_bwd_vec_kernel(const float* __restrict__ li, const float* __restrict__ g,
                                                                  int B, int T, int V, float eps, float gscale,
                                                                  float* __restrict__ dd) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);      // row = t*B + b
    if (row >= B * T) return;
    const int lane = threadIdx.x & 63;
    const int t = row / B, b = row - t * B;
    const int n4 = V >> 2;
    const float4* l4 = (const float4*)(li + (long)row * V);
    const float4* g4 = (const float4*)(g + (long)row * V);
    float4 pm[NV], q[NV];          // p = (p + eps) - eps and the quotient g / (p + eps)
    float inner = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int k4 = lane + 64 * i;
        const bool ok = k4 < n4;
        const float4 l = ok ? l4[k4] : make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 gr = ok ? g4[k4] : make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 pe = make_float4(expf(l.x), expf(l.y), expf(l.z), expf(l.w));
        pm[i] = ok ? make_float4(pe.x - eps, pe.y - eps, pe.z - eps, pe.w - eps) : make_float4(0.f, 0.f, 0.f, 0.f);
        q[i] = make_float4(gr.x / pe.x, gr.y / pe.y, gr.z / pe.z, gr.w / pe.w);
        inner += (pm[i].x * q[i].x + pm[i].y * q[i].y) + (pm[i].z * q[i].z + pm[i].w * q[i].w);
    }
    inner = asr_wave_sum(inner);
    float4* o4 = (float4*)(dd + ((long)b * T + t) * V);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int k4 = lane + 64 * i;
        if (k4 < n4) o4[k4] = make_float4(pm[i].x * (q[i].x - inner) * gscale, pm[i].y * (q[i].y - inner) * gscale,
                                          pm[i].z * (q[i].z - inner) * gscale, pm[i].w * (q[i].w - inner) * gscale);
    }
}

// ------------------------------------------------------------------ CTC
// one wave per (t,b) row: lse of the row (TF's own log-softmax), the dense part of the gradient, softmax(logits), and the emission
// probabilities of the utterance's lattice states p_t(s) = exp(logit[ext(s)] - lse) in float64 (ext = blank, l1, blank, l2, ...: two
// exps per lane here instead of eighteen per thread in front of the lattice's serial part); rows at t >= seq_len or of infeasible
// utterances are zero.  prob rows: [b][t][SW].
// asr_ctc_loss's status of utterance b (0 ok, 1 infeasible: no valid alignment / bad lengths, 2 a padding row of a short batch), by every row's
// wave for itself -- two scalar loads and a ballot per 64 labels instead of a launch of its own in front of the row pass (the lattice
// kernel reads what the t = 0 row wrote)
__device__ __forceinline__ int ctc_row_status(const int32_t* __restrict__ labels, int max_label, const int32_t* __restrict__ label_len,
                                              const int32_t* __restrict__ seq_len, int T, int b, int lane) {
    const int L = label_len[b], Tb = seq_len[b];
    int bad = (L < 0 || L > max_label || Tb <= 0 || Tb > T) ? 1 : 0;
    if (Tb == 0 && L == 0) bad = 2;          // padding row of a short batch: loss 0, gradient 0
    if (!bad) {
        int rep = 0;
        for (int i0 = 1; i0 < L; i0 += 64) {
            const int i = i0 + lane;
            const bool same = i < L && labels[(long)b * max_label + i] == labels[(long)b * max_label + i - 1];
            rep += __popcll(__ballot(same));
        }
        if (Tb < L + rep) bad = 1;
    }
    return bad;
}

__device__ __forceinline__ void ctc_row_probs(const float* __restrict__ x, double l, int b, int t, int T, int lane,
                                               const int32_t* __restrict__ labels, int max_label, int L, int blank, int SW,
                                               double* __restrict__ prob) {
    const int S = 2 * L + 1;
    double* pr = prob + ((long)b * T + t) * SW;
    for (int q = lane; q < S; q += 64) {
        const int e = (q & 1) ? labels[(long)b * max_label + (q >> 1)] : blank;
        pr[q] = exp((double)x[e] - l);
    }
}

__global__ __launch_bounds__(256) void ctc_rows_kernel(const float* __restrict__ logits, int T, int B, int V,
                                                       const int32_t* __restrict__ seq_len, int32_t* __restrict__ status,
                                                       const int32_t* __restrict__ labels, int max_label,
                                                       const int32_t* __restrict__ label_len, int blank, int SW,
                                                       double* __restrict__ lse, float* __restrict__ grad, double* __restrict__ prob) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);      // row = t*B + b
    if (row >= T * B) return;
    const int lane = threadIdx.x & 63;
    const int t = row / B, b = row - t * B;
    float* gr = grad + (long)row * V;
    const int st = ctc_row_status(labels, max_label, label_len, seq_len, T, b, lane);
    if (t == 0 && lane == 0) status[b] = st;
    if (st != 0 || t >= seq_len[b]) {
        for (int k = lane; k < V; k += 64) gr[k] = 0.f;
        if (lane == 0) lse[row] = 0.0;
        return;
    }
    const float* x = logits + (long)row * V;
    float m = -INFINITY;
    for (int k = lane; k < V; k += 64) m = fmaxf(m, x[k]);
    m = asr_wave_max(m);
    double s = 0.0;
    for (int k = lane; k < V; k += 64) s += (double)expf(x[k] - m);
    s = asr_wave_sum_d(s);
    const double l = (double)m + log(s);
    if (lane == 0) lse[row] = l;
    const float lf = (float)l;
    for (int k = lane; k < V; k += 64) gr[k] = expf(x[k] - lf);
    ctc_row_probs(x, l, b, t, T, lane, labels, max_label, label_len[b], blank, SW, prob);
}

// the same with the row in registers (V % 4 == 0, V <= 64 x 4 x NV): one 16-byte load per 16 bytes instead of three 4-byte loads
template <int NV>
__global__ __launch_bounds__(256) void ctc_rows_vec_kernel(const float* __restrict__ logits, int T, int B, int V,
                                                           const int32_t* __restrict__ seq_len, int32_t* __restrict__ status,
                                                           const int32_t* __restrict__ labels, int max_label,
                                                           const int32_t* __restrict__ label_len, int blank, int SW,
                                                           double* __restrict__ lse, float* __restrict__ grad, double* __restrict__ prob) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);      // row = t*B + b
    if (row >= T * B) return;
    const int lane = threadIdx.x & 63;
    const int t = row / B, b = row - t * B;
    const int n4 = V >> 2;
    float4* g4 = (float4*)(grad + (long)row * V);
    const int st = ctc_row_status(labels, max_label, label_len, seq_len, T, b, lane);
    if (t == 0 && lane == 0) status[b] = st;
    if (st != 0 || t >= seq_len[b]) {
        for (int k4 = lane; k4 < n4; k4 += 64) g4[k4] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (lane == 0) lse[row] = 0.0;
        return;
    }
    const float* x = logits + (long)row * V;
    const float4* x4 = (const float4*)x;
    float4 v[NV];
    float m = -INFINITY;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int k4 = lane + 64 * i;
        v[i] = k4 < n4 ? x4[k4] : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        m = fmaxf(fmaxf(m, fmaxf(v[i].x, v[i].y)), fmaxf(v[i].z, v[i].w));
    }
    m = asr_wave_max(m);
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < NV; ++i)
        s += ((double)expf(v[i].x - m) + (double)expf(v[i].y - m)) + ((double)expf(v[i].z - m) + (double)expf(v[i].w - m));
    s = asr_wave_sum_d(s);
    const double l = (double)m + log(s);
    if (lane == 0) lse[row] = l;
    const float lf = (float)l;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int k4 = lane + 64 * i;
        if (k4 < n4) g4[k4] = make_float4(expf(v[i].x - lf), expf(v[i].y - lf), expf(v[i].z - lf), expf(v[i].w - lf));
    }
    ctc_row_probs(x, l, b, t, T, lane, labels, max_label, label_len[b], blank, SW, prob);
}

// value of the lane below / above across the whole wave (lane 0 / lane 63: zero) by DPP wave shifts, two 32-bit moves per double
__device__ __forceinline__ double wave_shr1_d(double x) {
    const unsigned lo = (unsigned)__double2loint(x), hi = (unsigned)__double2hiint(x);
    const int l = __builtin_amdgcn_update_dpp(0, (int)lo, 0x138, 0xF, 0xF, false);      // wave_shr:1
    const int h = __builtin_amdgcn_update_dpp(0, (int)hi, 0x138, 0xF, 0xF, false);
    return __hiloint2double(h, l);
}
__device__ __forceinline__ double wave_shl1_d(double x) {
    const unsigned lo = (unsigned)__double2loint(x), hi = (unsigned)__double2hiint(x);
    const int l = __builtin_amdgcn_update_dpp(0, (int)lo, 0x130, 0xF, 0xF, false);      // wave_shl:1
    const int h = __builtin_amdgcn_update_dpp(0, (int)hi, 0x130, 0xF, 0xF, false);
    return __hiloint2double(h, l);
}

// Wave sums of TWO float64 values at once (all 64 lanes active; every lane gets both totals): four DPP butterfly steps inside the
// rows of sixteen (quad_perm, row_half_mirror, row_mirror: no LDS crossbar round trip as ds_bpermute's, and the two chains fill
// each other's wait states), then the four row totals through scalar registers.  A fixed order.
template <int CTRL>
__device__ __forceinline__ double dpp_perm_d(double x) {
    const int l = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, 0xF, 0xF, true);
    const int h = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, 0xF, 0xF, true);
    return __hiloint2double(h, l);
}
__device__ __forceinline__ double readlane_d(double x, int lane) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), lane), __builtin_amdgcn_readlane(__double2loint(x), lane));
}
__device__ __forceinline__ void wave_sum2_d(double& a, double& b) {
    a += dpp_perm_d<0xB1>(a); b += dpp_perm_d<0xB1>(b);        // quad_perm [1, 0, 3, 2]
    a += dpp_perm_d<0x4E>(a); b += dpp_perm_d<0x4E>(b);        // quad_perm [2, 3, 0, 1]
    a += dpp_perm_d<0x141>(a); b += dpp_perm_d<0x141>(b);      // row_half_mirror
    a += dpp_perm_d<0x140>(a); b += dpp_perm_d<0x140>(b);      // row_mirror
    a = (readlane_d(a, 0) + readlane_d(a, 16)) + (readlane_d(a, 32) + readlane_d(a, 48));
    b = (readlane_d(b, 0) + readlane_d(b, 16)) + (readlane_d(b, 32) + readlane_d(b, 48));
}

// One workgroup of 512 threads per utterance.  Phase 1: wave 0 runs the alpha recursion while wave 1 runs the beta recursion,
// both lattices go to the float64 workspace.  Phase 2 has no sequential dependency: the eight waves walk the time
// steps in parallel, form the state occupancies and subtract their per-label sums from
// the dense softmax term that ctc_rows_kernel already wrote.
__global__ __launch_bounds__(512) void ctc_lattice_kernel(const float* __restrict__ logits, int T, int B, int V,
                                                          const int32_t* __restrict__ labels, int max_label,
                                                          const int32_t* __restrict__ label_len,
                                                          const int32_t* __restrict__ seq_len, int blank,
                                                          const int32_t* __restrict__ status, const double* __restrict__ lse,
                                                          double* __restrict__ alpha_ws, double* __restrict__ beta_ws,
                                                          double* __restrict__ prob_ws,
                                                          float* __restrict__ loss, float* __restrict__ grad, int p_cap) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smraw[];
    const int b = blockIdx.x, tid = threadIdx.x;
    if (status[b] != 0) {
        if (tid == 0) loss[b] = (status[b] == 2) ? 0.f : INFINITY;
        return;
    }
    const int L = label_len[b], Tb = seq_len[b], S = 2 * L + 1;
    const int SMAX = 2 * max_label + 1;
    const int half = tid >> 8, s = tid & 255;           // half 0: alpha, half 1: beta
    double* abuf0 = (double*)smraw;                     // alpha ping/pong with two leading zero sentinels
    double* abuf1 = abuf0 + SMAX + 4;
    double* bbuf0 = abuf1 + SMAX + 4;                   // beta ping/pong with two trailing zero sentinels
    double* bbuf1 = bbuf0 + SMAX + 4;
    double* occ = bbuf1 + SMAX + 4;                     // [16: the first four are zeros] + [8 waves][SMAX + 1]
    double* lse_t = occ + 16 + 8 * (SMAX + 1);          // [T]
    double* sca = lse_t + T;                            // [T] cumulative log scale of alpha per frame; later the frame's occupancy factor
    double* scb = sca + T;                              // [T] the same for beta
    double* llp = scb + T;                              // [1] log-likelihood
    int* ext = (int*)(llp + 2);                         // [SMAX + 1]
    int* nxt = ext + SMAX + 1;                          // [SMAX + 1] next state with the same label (or -1)
    int* lead = nxt + SMAX + 1;                         // [SMAX + 1] 1 = first odd state of its label
    // When the utterance's lattice fits (p_cap doubles): its emission probabilities [Tb][SPL] in float64 (16-byte aligned rows of SPL =
    // S rounded up to 4, + 4: zeros behind the last state).  The log-domain form puts the gathered lattice inputs [Tb][S] (float) here.
    // (the offset is rounded, not the pointer: a pointer that went through an integer is a generic one, and its loads are flat loads that
    //  share the vector-memory counter with the recursion's stores)
    const int lg_off = (int)(((unsigned char*)(lead + SMAX + 1) - smraw + 15) & ~15);
    float* lg = (float*)(smraw + lg_off);
    double* pl = (double*)(smraw + lg_off);
    const int SW = ((SMAX + 3) & ~3) + 4;               // row pitch of the float64 workspaces
    const int SP = (S + 3) & ~3, SPL = SP + 4;
    const bool plds = S <= 256 && (long)Tb * SPL <= (long)p_cap;
    double* aw = alpha_ws + (long)b * T * SW;
    double* bw = beta_ws + (long)b * T * SW;
    double* pw = prob_ws + (long)b * T * SW;            // p_t(s), written by ctc_rows_kernel

    if (tid < S) ext[tid] = (tid & 1) ? labels[(long)b * max_label + (tid >> 1)] : blank;
    for (int t = tid; t < Tb; t += 512) lse_t[t] = lse[(long)t * B + b];
    // ---- phase 0: the probabilities into LDS; the columns behind state S - 1 are zeros, so that the recursion below runs whole groups
    // of four states without a test
    if (plds) {
        for (int i0 = tid; i0 < Tb * SPL; i0 += 512 * 8) {
            double gv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u * 512;
                const int t = i / SPL, q = i - t * SPL;
                gv[u] = (i < Tb * SPL && q < S) ? pw[(long)t * SW + q] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u * 512;
                if (i < Tb * SPL) pl[i] = gv[u];
            }
        }
    }
    __syncthreads();
    int my = blank; bool skip_f = false, skip_b = false;
    if (s < S) {
        my = ext[s];
        skip_f = (s >= 2) && (my != blank) && (my != ext[s - 2]);
        skip_b = (s + 2 < S) && (my != blank) && (my != ext[s + 2]);
        if (half == 0) {
            int n = -1, ld = 0;
            if (s & 1) {
                ld = 1;
                for (int j = 1; j < s; j += 2) if (ext[j] == my) { ld = 0; break; }
                for (int j = s + 2; j < S; j += 2) if (ext[j] == my) { n = j; break; }
            }
            nxt[s] = n; lead[s] = ld;
        }
    }
    // the lattice inputs as floats, for the log-domain form only: gathered in batches of eight independent loads (one load per iteration
    // left its whole latency exposed: 50 round trips)
    auto gather_lg = [&]() {
    for (int i0 = tid; i0 < Tb * S; i0 += 512 * 8) {
        float gv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + u * 512;
            const int t = i / S, q = i - t * S;
            gv[u] = (i < Tb * S) ? logits[((long)t * B + b) * V + ext[q]] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + u * 512;
            if (i < Tb * S) lg[i] = gv[u];
        }
    }
    };
    if (tid < 2) { abuf0[tid] = 0.0; abuf1[tid] = 0.0; }                 // linear domain: the sentinels are zeros
    if (tid >= 64 && tid < 68) occ[tid - 64] = 0.0;                      // a block of four zeros (phase 1's idle lanes read it)
    if (tid >= 2 && tid < 4) { bbuf0[2 + S + tid - 2] = 0.0; bbuf1[2 + S + tid - 2] = 0.0; }
    __threadfence_block();
    __syncthreads();
    // ---- phase 1: alpha forward in time (wave 0), beta backward in time (wave 1), concurrently, in the LINEAR domain with
    // rescaling: a_t(s) = (a_{t-1}(s) + a_{t-1}(s-1) + [skip] a_{t-1}(s-2)) p_t(s) is two adds and a multiply where the
    // log-domain form (the oracle's) pays three exps and a log per state and step on the serial path.  ONE wave per
    // direction holds the whole vector in registers, four states per lane, neighbours by lane shifts: no barrier and no LDS
    // round trip per frame (the 512-thread form with a barrier per frame spent 0.25 us per frame on exactly those).
    // Every RS frames the vector is divided by its sum and the log of the divisor is carried along: true alpha_t = stored
    // alpha_t * exp(sca[t]); between two rescalings a state loses at most RS emission factors.
    constexpr int RS = 16, NPL = 4, PD = 8;             // rescale period, states per lane, prefetch distance (frames)
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), ln = tid & 63;
    const bool fast = S <= 64 * NPL;                    // the register-resident form holds 256 states; longer label rows go to the log-domain form
    int* badp = (int*)(llp + 1);
    if (tid == 0) *badp = fast ? 0 : 1;
    if (fast && plds && wv < 2) {
        // p from LDS: a frame is two 16-byte LDS reads per lane (issued two frames ahead, two register sets used alternately), two
        // 16-byte stores of the new vector, the shifts and twelve float64 operations -- no test, no address arithmetic beyond running
        // offsets.  All 64 lanes run (the wave sums and the DPP shifts read every lane); the lanes behind the last state's run read a
        // block of zeros and store nothing.  The beta wave stores q_t(s) = beta_t(s) / p_t(s) -- the sum BEFORE the multiplication by
        // p_t(s) -- which is what the occupancy alpha_t(s) beta_t(s) / p_t(s) needs: phase 2 has no division.
        // States per lane NL: two while the lattice has at most 128 states (half the float64 operations per frame on the serial path:
        // the wave issues ~25 instead of ~45 instructions per frame), four above.
        auto run = [&](auto dir, auto npl) {
            constexpr bool fwd = decltype(dir)::value;
            constexpr int NL = decltype(npl)::value;
            const int s0 = ln * NL;
            const bool act = s0 < SPL;
            double* scl = fwd ? sca : scb;
            bool sk[NL];
#pragma unroll
            for (int j = 0; j < NL; ++j) {
                const int q = s0 + j;
                sk[j] = false;
                if (q < S) {
                    const int e = ext[q];
                    sk[j] = fwd ? ((q >= 2) && (e != blank) && (e != ext[q - 2])) : ((q + 2 < S) && (e != blank) && (e != ext[q + 2]));
                }
            }
            const int t0 = fwd ? 0 : Tb - 1;
            const int dp = act ? (fwd ? SPL : -SPL) : 0, dx = fwd ? SW : -SW;
            const double* pp = act ? pl + t0 * SPL + s0 : occ;          // occ[0..3]: zeros
            double* xp = (fwd ? aw : bw) + (long)t0 * SW + s0;
            double v[NL], pa[NL], pb[NL];
            auto put = [&](const double (&x)[NL]) {
                if (act) {
#pragma unroll
                    for (int j = 0; j < NL; j += 2) *(double2*)(xp + j) = double2{x[j], x[j + 1]};
                }
            };
#pragma unroll
            for (int j = 0; j < NL; ++j) {
                const int q = s0 + j;
                const bool on = fwd ? (q < 2) : (q >= S - 2);
                const double p0 = pp[j];                 // (zeros behind S - 1)
                v[j] = on ? p0 : 0.0;
                pa[j] = 0.0; pb[j] = 0.0;
            }
            if (fwd) put(v);
            else {
                double q1[NL];                           // q at the last frame: 1 on the two final states
#pragma unroll
                for (int j = 0; j < NL; ++j) q1[j] = (s0 + j >= S - 2 && s0 + j < S) ? 1.0 : 0.0;
                put(q1);
            }
            if (ln == 0) scl[t0] = 0.0;
            xp += dx;
            if (Tb > 1) {
#pragma unroll
                for (int j = 0; j < NL; ++j) pa[j] = pp[dp + j];
            }
            if (Tb > 2) {
#pragma unroll
                for (int j = 0; j < NL; ++j) pb[j] = pp[2 * dp + j];
            }
            pp += 3 * dp;                                // the row of frame k + 2 when frame k is at work
            double lsc = 0.0;
            int t = t0;
            // one frame: the vector times the frame's probabilities pc; then pc is refilled with the probabilities of frame k + 2
            auto frame = [&](double (&pc)[NL], int k) {
                t += fwd ? 1 : -1;
                double inv = 1.0;
                const bool resc = (k % RS) == 0;         // uniform: rescale by the sum of the previous frame's vector
                if (resc) {
                    double tot = NL == 4 ? (v[0] + v[1]) + (v[NL - 2] + v[NL - 1]) : v[0] + v[1], dummy = 0.0;
                    wave_sum2_d(tot, dummy);
                    if (tot > 0.0) { inv = 1.0 / tot; lsc += log(tot); }
                }
                double sm[NL];
                if (fwd) {
                    // neighbours below: states s0 - 1, s0 - 2 live in lane - 1 (its last two states); lane 0 has none
                    const double m1 = wave_shr1_d(v[NL - 1]), m2 = wave_shr1_d(v[NL - 2]);
#pragma unroll
                    for (int j = 0; j < NL; ++j) {
                        const double lo1 = j >= 1 ? v[j >= 1 ? j - 1 : 0] : m1;
                        const double lo2 = j >= 2 ? v[j >= 2 ? j - 2 : 0] : (j == 1 ? m1 : m2);
                        sm[j] = (v[j] + lo1) + (sk[j] ? lo2 : 0.0);
                    }
                } else {
                    const double p1 = wave_shl1_d(v[0]), p2 = wave_shl1_d(v[1]);
#pragma unroll
                    for (int j = NL - 1; j >= 0; --j) {
                        const double hi1 = j + 1 < NL ? v[j + 1 < NL ? j + 1 : 0] : p1;
                        const double hi2 = j + 2 < NL ? v[j + 2 < NL ? j + 2 : 0] : (j + 1 < NL ? p1 : p2);
                        sm[j] = (v[j] + hi1) + (sk[j] ? hi2 : 0.0);
                    }
                }
                if (resc) {
#pragma unroll
                    for (int j = 0; j < NL; ++j) sm[j] *= inv;
                }
#pragma unroll
                for (int j = 0; j < NL; ++j) v[j] = sm[j] * pc[j];
                if (fwd) put(v); else put(sm);
                xp += dx;
                if (ln == 0) scl[t] = lsc;
                if (k + 2 < Tb) {                        // uniform
#pragma unroll
                    for (int j = 0; j < NL; ++j) pc[j] = pp[j];
                }
                pp += dp;
            };
            int k = 1;
            for (; k + 1 < Tb; k += 2) { frame(pa, k); frame(pb, k + 1); }
            if (k < Tb) frame(pa, k);
            if (fwd) {
#pragma unroll
                for (int j = 0; j < NL; ++j) if (s0 + j < S) abuf0[2 + s0 + j] = v[j];
            }
        };
        if (S <= 128) { if (wv == 0) run(std::true_type{}, std::integral_constant<int, 2>{}); else run(std::false_type{}, std::integral_constant<int, 2>{}); }
        else { if (wv == 0) run(std::true_type{}, std::integral_constant<int, 4>{}); else run(std::false_type{}, std::integral_constant<int, 4>{}); }
    } else if (fast && wv < 2) {
        const bool fwd = wv == 0;
        double* scl = fwd ? sca : scb;
        double* xw = fwd ? aw : bw;
        auto frame = [&](int k) { return fwd ? k : Tb - 1 - k; };
        const int s0 = ln * NPL;
        bool sk[NPL];
#pragma unroll
        for (int j = 0; j < NPL; ++j) {
            const int q = s0 + j;
            sk[j] = false;
            if (q < S) {
                const int e = ext[q];
                sk[j] = fwd ? ((q >= 2) && (e != blank) && (e != ext[q - 2])) : ((q + 2 < S) && (e != blank) && (e != ext[q + 2]));
            }
        }
        double v[NPL], pr[PD][NPL];
        auto fetch = [&](double (&dst)[NPL], int k) {
#pragma unroll
            for (int j = 0; j < NPL; ++j) dst[j] = (k < Tb && s0 + j < S) ? pw[(long)frame(k) * SW + s0 + j] : 0.0;
        };
        {
            const int t = frame(0);
#pragma unroll
            for (int j = 0; j < NPL; ++j) {
                const int q = s0 + j;
                const bool on = fwd ? (q < 2) : (q >= S - 2);
                v[j] = (q < S && on) ? pw[(long)t * SW + q] : 0.0;
                if (q < S) xw[(long)t * SW + q] = v[j];
            }
            if (ln == 0) scl[t] = 0.0;
        }
#pragma unroll
        for (int u = 0; u < PD; ++u) fetch(pr[u], 1 + u);
        double lsc = 0.0;
        for (int k0 = 1; k0 < Tb; k0 += PD) {
#pragma unroll
            for (int u = 0; u < PD; ++u) {
                const int k = k0 + u;
                if (k < Tb) {                            // uniform
                    double inv = 1.0;
                    if (k % RS == 0) {                   // uniform: rescale by the sum of the previous frame's vector
                        const double tot = asr_wave_sum_d((v[0] + v[1]) + (v[2] + v[3]));
                        if (tot > 0.0) { inv = 1.0 / tot; lsc += log(tot); }
                    }
                    double nv[NPL];
                    if (fwd) {
                        // neighbours below: states s0 - 1, s0 - 2 live in lane - 1 (its v[3], v[2]); lane 0 has none
                        // (DPP wave shifts -- lane i takes lane i - 1's value, lane 0 the zero -- instead of ds_bpermute round trips: the
                        //  shift sits on the recursion's serial path, once per frame)
                        const double m1 = wave_shr1_d(v[NPL - 1]), m2 = wave_shr1_d(v[NPL - 2]);
                        nv[0] = ((v[0] + m1) + (sk[0] ? m2 : 0.0)) * inv * pr[u][0];
                        nv[1] = ((v[1] + v[0]) + (sk[1] ? m1 : 0.0)) * inv * pr[u][1];
                        nv[2] = ((v[2] + v[1]) + (sk[2] ? v[0] : 0.0)) * inv * pr[u][2];
                        nv[3] = ((v[3] + v[2]) + (sk[3] ? v[1] : 0.0)) * inv * pr[u][3];
                    } else {
                        const double p1 = wave_shl1_d(v[0]), p2 = wave_shl1_d(v[1]);
                        nv[3] = ((v[3] + p1) + (sk[3] ? p2 : 0.0)) * inv * pr[u][3];
                        nv[2] = ((v[2] + v[3]) + (sk[2] ? p1 : 0.0)) * inv * pr[u][2];
                        nv[1] = ((v[1] + v[2]) + (sk[1] ? v[3] : 0.0)) * inv * pr[u][1];
                        nv[0] = ((v[0] + v[1]) + (sk[0] ? v[2] : 0.0)) * inv * pr[u][0];
                    }
                    const int t = frame(k);
#pragma unroll
                    for (int j = 0; j < NPL; ++j) {
                        v[j] = nv[j];
                        if (s0 + j < S) xw[(long)t * SW + s0 + j] = nv[j];
                    }
                    if (ln == 0) scl[t] = lsc;
                    fetch(pr[u], k + PD);
                }
            }
        }
        // the last frame's vector of alpha, for the likelihood
        if (fwd) {
#pragma unroll
            for (int j = 0; j < NPL; ++j) if (s0 + j < S) abuf0[2 + s0 + j] = v[j];
        }
    }
    __syncthreads();
    if (fast && tid == 0) {       // abuf0 holds the scaled alpha at the last frame
        const double* prev = abuf0 + 2;
        const double tail = (S > 1) ? prev[S - 1] + prev[S - 2] : prev[0];
        const double ll = log(tail) + sca[Tb - 1];
        llp[0] = ll;
        loss[b] = (float)(-ll);
        // A state more than ~708 nats below the frame's dominant state flushes to zero in float64 (the vector is rescaled by
        // its SUM): with the softmax saturated at the eps floor (16 nats per label emission) and >= ~45 labels the final
        // states do.  Then the log-domain form below redoes this utterance (tf.nn.ctc_loss returns a finite loss there).
        if (!(tail > 0.0) || !(fabs(ll) < 1e300)) *badp = 1;
    }
    __threadfence_block();
    __syncthreads();
    const int wave = tid >> 6, lane = tid & 63;
    double* oc = occ + 16 + wave * (SMAX + 1);
    // subtracts the per-label sums of one frame's occupancies oc[0..S) from the dense softmax term of its gradient row
    auto scatter_frame = [&](int t) {
        float* gr = grad + ((long)t * B + b) * V;
        double bsum = 0.0;
        for (int q = 2 * lane; q < S; q += 128) bsum += oc[q];
        bsum = asr_wave_sum_d(bsum);
        if (lane == 0) gr[blank] -= (float)bsum;
        for (int q = 2 * lane + 1; q < S; q += 128) {
            if (lead[q]) {
                double sum = oc[q];
                for (int j = nxt[q]; j >= 0; j = nxt[j]) sum += oc[j];
                gr[ext[q]] -= (float)sum;
            }
        }
    };
    if (*badp == 0) {
        // ---- phase 2: occupancies alpha_t(s) beta_t(s) / (p_t(s) P) and the sparse part of the gradient, time steps in
        // parallel over the waves; the scales of a frame enter as ONE factor exp(sca[t] + scb[t] - ll) (formed for all frames at once,
        // a frame per thread, instead of one exp per frame on every wave's path).  The occupancies of
        // a frame are a probability distribution over its states: a sum off 1 means that states the gradient needs were
        // flushed to zero in one of the two recursions -> the utterance is redone in the log domain.
        const double ll = llp[0];
        for (int t = tid; t < Tb; t += 512) sca[t] = exp(sca[t] + scb[t] - ll);
        __syncthreads();
        constexpr int QL = 4;                               // states per lane: S <= 256
        const int nj = (S + 63) >> 6;                       // (uniform) 64-state groups in use
        double ra[QL], rb[QL], rp[QL];
        auto fetch3 = [&](int t) {
#pragma unroll
            for (int j = 0; j < QL; ++j) {
                const int q = lane + 64 * j;
                const bool ok = j < nj && t < Tb && q < S;
                ra[j] = ok ? aw[(long)t * SW + q] : 0.0;
                rb[j] = ok ? bw[(long)t * SW + q] : 0.0;
                rp[j] = (ok && !plds) ? pw[(long)t * SW + q] : 1.0;
            }
        };
        // the gradient entries a frame's scatter updates (lane 0: the blank; lanes 0..L-1: a label, if it is the first state of its
        // label) are fetched a frame ahead as well: the read-modify-write's load was a global round trip per frame on every wave
        const int ql = 2 * lane + 1;
        const bool has_l = ql < S && lead[ql];
        const int el = has_l ? ext[ql] : 0;
        float gb_n = 0.f, gl_n = 0.f;
        auto gfetch = [&](int t) {
            if (t < Tb) {
                const float* gr = grad + ((long)t * B + b) * V;
                if (lane == 0) gb_n = gr[blank];
                if (has_l) gl_n = gr[el];
            }
        };
        bool off = false;
        fetch3(wave);
        gfetch(wave);
        for (int t = wave; t < Tb; t += 8) {
            const float gb = gb_n, gl = gl_n;
            const double ft = sca[t];
            double tot = 0.0;
#pragma unroll
            for (int j = 0; j < QL; ++j) {
                const int q = lane + 64 * j;
                double o;
                if (plds) o = ra[j] * rb[j] * ft;           // (uniform) rb = beta / p
                else { const double ab = ra[j] * rb[j]; o = (ab == 0.0) ? 0.0 : ab / rp[j] * ft; }
                if (j < nj && q < S) { oc[q] = o; tot += o; }
            }
            fetch3(t + 8);                                  // the next frame's values travel while this frame's sums are formed
            gfetch(t + 8);
            // the frame's total and its blank states' total (state q = lane + 64 j is a blank iff the lane is even) in one pass
            double bsum = (lane & 1) ? 0.0 : tot;
            wave_sum2_d(tot, bsum);
            off = off || !(fabs(tot - 1.0) <= 1e-6);
            // (the wave's LDS writes are ordered before its own later reads)
            {
                float* gr = grad + ((long)t * B + b) * V;
                if (lane == 0) gr[blank] = gb - (float)bsum;
                if (has_l) {                                // (2 * lane + 1 + 128 >= S: one state per lane)
                    double sum = oc[ql];
                    for (int j = nxt[ql]; j >= 0; j = nxt[j]) sum += oc[j];
                    gr[el] = gl - (float)sum;
                }
            }
        }
        if (off && lane == 0) *badp = 1;
    }
    __syncthreads();
    if (*badp == 0) return;

    // ---- log-domain form (the oracle's recursion; Appendix A7): alpha by threads 0-255, beta by threads 256-511, a state per
    // thread (and stride), one barrier per frame; the float64 workspaces now hold log alpha / log beta.  Slow (a barrier per
    // frame) and rare: only utterances whose linear-domain lattice lost states it needs.
    {
        const double NEG = -INFINITY;
        auto lse3 = [&](double a, double b2, double c) {
            const double m = fmax(a, fmax(b2, c));
            if (m == NEG) return NEG;
            return m + log(exp(a - m) + exp(b2 - m) + exp(c - m));
        };
        gather_lg();                                         // the float inputs (in LDS mode they take the place of the probabilities)
        __syncthreads();
        auto lpr = [&](int t, int q) { return (double)lg[t * S + q] - lse_t[t]; };
        const int h = tid >> 8, s1 = tid & 255;
        for (int k = 0; k < Tb; ++k) {
            const int t = h == 0 ? k : Tb - 1 - k;
            double* xw = h == 0 ? aw : bw;
            for (int q = s1; q < S; q += 256) {
                double val;
                if (k == 0) {
                    const bool on = h == 0 ? (q < 2) : (q >= S - 2);
                    val = on ? lpr(t, q) : NEG;
                } else {
                    const double* pv = xw + (long)(h == 0 ? t - 1 : t + 1) * SW;
                    const int e = ext[q];
                    if (h == 0) {
                        const bool skp = (q >= 2) && (e != blank) && (e != ext[q - 2]);
                        val = lpr(t, q) + lse3(pv[q], q >= 1 ? pv[q - 1] : NEG, skp ? pv[q - 2] : NEG);
                    } else {
                        const bool skp = (q + 2 < S) && (e != blank) && (e != ext[q + 2]);
                        val = lpr(t, q) + lse3(pv[q], q + 1 < S ? pv[q + 1] : NEG, skp ? pv[q + 2] : NEG);
                    }
                }
                xw[(long)t * SW + q] = val;
            }
            __threadfence_block();
            __syncthreads();
        }
        if (tid == 0) {
            const double* la = aw + (long)(Tb - 1) * SW;
            const double ll = (S > 1) ? lse3(la[S - 1], la[S - 2], NEG) : la[0];
            llp[0] = ll;
            loss[b] = (float)(-ll);
        }
        __threadfence_block();
        __syncthreads();
        const double ll = llp[0];
        for (int t = wave; t < Tb; t += 8) {
            // the dense softmax term of the entries the linear-domain pass may have touched, exactly as ctc_rows_kernel wrote them
            float* gr = grad + ((long)t * B + b) * V;
            const float* x = logits + ((long)t * B + b) * V;
            const float lf = (float)lse_t[t];
            if (lane == 0) gr[blank] = expf(x[blank] - lf);
            for (int q = 2 * lane + 1; q < S; q += 128)
                if (lead[q]) gr[ext[q]] = expf(x[ext[q]] - lf);
            for (int q = lane; q < S; q += 64) {
                const double e = aw[(long)t * SW + q] + bw[(long)t * SW + q] - lpr(t, q) - ll;
                oc[q] = (e == e && e > -745.0) ? exp(e) : 0.0;
            }
            scatter_frame(t);
        }
    }
}


// ------------------------------------------------------------------ greedy decode
// pass 1: one wave per (t,b) row over the whole chip: argmax with the lowest index on ties
__global__ __launch_bounds__(256) void ctc_argmax_kernel(const float* __restrict__ logits, int T, int B, int V,
                                                         int32_t* __restrict__ best_k, float* __restrict__ best_v) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);      // row = t*B + b
    if (row >= T * B) return;
    const int lane = threadIdx.x & 63;
    const float* x = logits + (long)row * V;
    float bv = -INFINITY; int bk = 0x7fffffff;
    for (int k = lane; k < V; k += 64) {
        const float v = x[k];
        if (v > bv || bk == 0x7fffffff) { bv = v; bk = k; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bv, o, 64);
        const int ok = __shfl_xor(bk, o, 64);
        if (ov > bv || (ov == bv && ok < bk)) { bv = ov; bk = ok; }
    }
    if (lane == 0) { best_k[row] = bk; best_v[row] = bv; }
}

// pass 2: per utterance, merge repeats / drop blanks in time order.  One wave per utterance, 64 frames at a time: lane t holds frame
// t's arg-max, its left neighbour comes by a lane shift (the chunk's first lane takes the previous chunk's last id), the survivors
// are counted with a ballot + popcount prefix and scattered.  (Round 4: thread 0 walked the frames alone, two dependent global
// loads per frame -- 56 us for 125 frames.)  neg_sum adds the frames' scores in time order, one at a time, as before: same bits.
__global__ __launch_bounds__(64) void ctc_compact_kernel(const int32_t* __restrict__ best_k, const float* __restrict__ best_v,
                                                         int T, int B, const int32_t* __restrict__ seq_len, int blank,
                                                         int32_t* __restrict__ out_ids, int32_t* __restrict__ out_len,
                                                         float* __restrict__ neg_sum) {
    const int b = blockIdx.x, lane = threadIdx.x;
    int Tb = seq_len[b];
    if (Tb > T) Tb = T;
    if (Tb < 0) Tb = 0;
    int n = 0, carry = -1;
    float acc = 0.f;
    for (int t0 = 0; t0 < Tb; t0 += 64) {
        const int t = t0 + lane;
        const bool in = t < Tb;
        const int k = in ? best_k[(long)t * B + b] : blank;
        const float v = in ? best_v[(long)t * B + b] : 0.f;
        int prev = __shfl_up(k, 1, 64);
        if (lane == 0) prev = carry;
        const bool keep = in && k != blank && k != prev;
        const unsigned long long m = __ballot(keep);
        const int pos = n + __popcll(m & ((1ull << lane) - 1ull));
        if (keep) out_ids[(long)b * T + pos] = k;
        n += __popcll(m);
        carry = __shfl(k, 63, 64);
        const int cnt = Tb - t0 < 64 ? Tb - t0 : 64;
        for (int i = 0; i < cnt; ++i) acc += -__shfl(v, i, 64);          // (uniform: every lane carries the same running sum)
    }
    for (int i = n + lane; i < T; i += 64) out_ids[(long)b * T + i] = -1;
    if (lane == 0) { out_len[b] = n; neg_sum[b] = acc; }
}

// ------------------------------------------------------------------ edit distance (one wave per pair)
__global__ __launch_bounds__(64) void edit_distance_kernel(const int32_t* __restrict__ hyp, int hyp_pitch,
                                                           const int32_t* __restrict__ hyp_len,
                                                           const int32_t* __restrict__ truth, int truth_pitch,
                                                           const int32_t* __restrict__ truth_len, int B,
                                                           float* __restrict__ dist) {
    extern __shared__ int sm_e[];
    const int b = blockIdx.x, lane = threadIdx.x;
    const int n = truth_len[b], m = hyp_len[b];
    int* prev = sm_e;                 // [n+1]
    int* tr = sm_e + truth_pitch + 1; // [n]
    for (int j = lane; j <= n; j += 64) prev[j] = j;
    for (int j = lane; j < n; j += 64) tr[j] = truth[(long)b * truth_pitch + j];
    __syncthreads();
    // Row i of the Levenshtein table from row i-1 (prev):  D[i][0] = i,
    //   t[j] = min(prev[j] + 1, prev[j-1] + cost(i,j)),  D[i][j] = min(t[j], D[i][j-1] + 1)
    // which unrolls to D[i][j] = j + min(i, min_{1<=k<=j}(t[k] - k)): a prefix-min per row.
    for (int i = 1; i <= m; ++i) {
        const int hi = hyp[(long)b * hyp_pitch + i - 1];
        int run = i;          // running min of (t[k] - k), k = 0 term is D[i][0] - 0
        int diag = i - 1;     // old prev[j0 - 1] for the first lane of the chunk (D[i-1][0] = i-1)
        for (int j0 = 1; j0 <= n; j0 += 64) {
            const int j = j0 + lane;
            const int pj = (j <= n) ? prev[j] : 0;
            int pjm1 = __shfl_up(pj, 1, 64);
            if (lane == 0) pjm1 = diag;
            diag = __shfl(pj, 63, 64);
            int u = 0x3fffffff;
            if (j <= n) u = min(pj + 1, pjm1 + (tr[j - 1] != hi ? 1 : 0)) - j;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int v = __shfl_up(u, o, 64);
                if (lane >= o) u = min(u, v);
            }
            u = min(u, run);
            run = __shfl(u, 63, 64);
            if (j <= n) prev[j] = u + j;
        }
        __syncthreads();
    }
    if (lane == 0) {
        const int d = (n == 0) ? m : prev[n];
        dist[b] = (n == 0) ? (m > 0 ? INFINITY : 0.f) : (float)d / (float)n;
    }
}

// ------------------------------------------------------------------ Adam (TF form)
__global__ void adam_tf_kernel(float* __restrict__ theta, const float* __restrict__ grad, float* __restrict__ m,
                               float* __restrict__ v, size_t n, float lr_t, float b1, float b2, float eps, float gscale) {
    for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += (size_t)gridDim.x * blockDim.x * 4) {
        if (i + 4 <= n) {
            const float4 g4 = *(const float4*)(grad + i);
            float4 m4 = *(const float4*)(m + i), v4 = *(const float4*)(v + i), t4 = *(const float4*)(theta + i);
            const float g[4] = {g4.x * gscale, g4.y * gscale, g4.z * gscale, g4.w * gscale};
            float mm[4] = {m4.x, m4.y, m4.z, m4.w}, vv[4] = {v4.x, v4.y, v4.z, v4.w}, tt[4] = {t4.x, t4.y, t4.z, t4.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                mm[j] = b1 * mm[j] + (1.f - b1) * g[j];
                vv[j] = b2 * vv[j] + (1.f - b2) * g[j] * g[j];
                tt[j] = tt[j] - lr_t * mm[j] / (sqrtf(vv[j]) + eps);
            }
            *(float4*)(m + i) = make_float4(mm[0], mm[1], mm[2], mm[3]);
            *(float4*)(v + i) = make_float4(vv[0], vv[1], vv[2], vv[3]);
            *(float4*)(theta + i) = make_float4(tt[0], tt[1], tt[2], tt[3]);
        } else {
            for (size_t j = i; j < n; ++j) {
                const float g = grad[j] * gscale;
                const float mj = b1 * m[j] + (1.f - b1) * g;
                const float vj = b2 * v[j] + (1.f - b2) * g * g;
                m[j] = mj; v[j] = vj;
                theta[j] = theta[j] - lr_t * mj / (sqrtf(vj) + eps);
            }
        }
    }
}

// bytes in front of the region that holds the gathered inputs or the probabilities (rounded up to 16), and the whole request:
// room for the float inputs [T][SMAX] at least (the form every utterance can fall back to), for the float64 probabilities of a
// full lattice [T][SMAX rounded up to 4, + 4] at most, capped at 144 KB -- an utterance whose own Tb x (S rounded up to 4, + 4)
// doubles fit keeps its probabilities in LDS (10 s of audio, 32 labels: 125 x 72 x 8 = 72 KB)
inline size_t ctc_lds_fixed(int T, int max_label) {
    const size_t SMAX = 2 * (size_t)max_label + 1;
    return ((4 * (SMAX + 4) + 16 + 8 * (SMAX + 1) + 3 * (size_t)T + 2) * sizeof(double) + 3 * (SMAX + 1) * sizeof(int) + 15 + 16) & ~(size_t)15;
}
inline size_t ctc_ws_pitch(int max_label) { return ((2 * (size_t)max_label + 1 + 3) & ~(size_t)3) + 4; }
inline size_t ctc_lds_bytes(int T, int max_label) {
    const size_t SMAX = 2 * (size_t)max_label + 1;
    const size_t fixed = ctc_lds_fixed(T, max_label);
    const size_t lo = fixed + (size_t)T * SMAX * sizeof(float) + 16, hi = fixed + (size_t)T * ctc_ws_pitch(max_label) * sizeof(double);
    const size_t cap = 144 * 1024;
    return lo > cap ? lo : (hi < cap ? hi : cap);
}

}  // namespace

extern "C" int asr_softmax_log_fwd(const float* d, int B, int T, int V, float eps, float* logits_tm, void* stream) {
    if (!d || !logits_tm || B < 1 || T < 1 || V < 1) return ASR_ERR_BAD_ARG;
    const bool vec = (V & 3) == 0 && V <= 64 * 4 * 8 && ((((uintptr_t)d) | ((uintptr_t)logits_tm)) & 15) == 0;
    if (vec && V <= 64 * 4 * 4) hipLaunchKernelGGL(softmax_log_fwd_vec_kernel<4>, dim3(asr_cdiv((long)B * T, 4)), dim3(256), 0, (hipStream_t)stream, d, B, T, V, eps, logits_tm);
    else if (vec && V <= 64 * 4 * 6) hipLaunchKernelGGL(softmax_log_fwd_vec_kernel<6>, dim3(asr_cdiv((long)B * T, 4)), dim3(256), 0, (hipStream_t)stream, d, B, T, V, eps, logits_tm);
    else if (vec) hipLaunchKernelGGL(softmax_log_fwd_vec_kernel<8>, dim3(asr_cdiv((long)B * T, 4)), dim3(256), 0, (hipStream_t)stream, d, B, T, V, eps, logits_tm);
    else hipLaunchKernelGGL(softmax_log_fwd_kernel, dim3(asr_cdiv((long)B * T, 4)), dim3(256), 0, (hipStream_t)stream, d, B, T, V, eps, logits_tm);
    ASR_CHECK_LAUNCH("softmax_log_fwd");
    return ASR_OK;
}

extern "C" int asr_softmax_log_bwd(const float* logits_tm, const float* g_tm, int B, int T, int V, float eps,
                                   float gscale, float* dd, void* stream) {
    if (!logits_tm || !g_tm || !dd || B < 1 || T < 1 || V < 1) return ASR_ERR_BAD_ARG;
    const bool vec = (V & 3) == 0 && V <= 64 * 4 * 8 && ((((uintptr_t)logits_tm) | ((uintptr_t)g_tm) | ((uintptr_t)dd)) & 15) == 0;
    if (vec && V <= 64 * 4 * 4) hipLaunchKernelGGL(softmax_log_bwd_vec_kernel<4>, dim3(asr_cdiv((long)B * T, 4)), dim3(256), 0, (hipStream_t)stream, logits_tm, g_tm, B, T, V, eps, gscale, dd);
    else if (vec && V <= 64 * 4 * 6) hipLaunchKernelGGL(softmax_log_bwd_vec_kernel<6>, dim3(asr_cdiv((long)B * T, 4)), dim3(256), 0, (hipStream_t)stream, logits_tm, g_tm, B, T, V, eps, gscale, dd);
    else if (vec) hipLaunchKernelGGL(softmax_log_bwd_vec_kernel<8>, dim3(asr_cdiv((long)B * T, 4)), dim3(256), 0, (hipStream_t)stream, logits_tm, g_tm, B, T, V, eps, gscale, dd);
    else hipLaunchKernelGGL(softmax_log_bwd_kernel, dim3(asr_cdiv((long)B * T, 4)), dim3(256), 0, (hipStream_t)stream, logits_tm, g_tm, B, T, V, eps, gscale, dd);
    ASR_CHECK_LAUNCH("softmax_log_bwd");
    return ASR_OK;
}

extern "C" size_t asr_ctc_workspace(int T, int B, int max_label) {
    return (((size_t)T * B + 1) / 2 * 2 + 3 * (size_t)B * T * ctc_ws_pitch(max_label)) * sizeof(double) + 64;
}

extern "C" int asr_ctc_loss(const float* logits_tm, int T, int B, int V, const int32_t* labels, int max_label,
                            const int32_t* label_len, const int32_t* seq_len, int blank,
                            float* loss, float* grad, int32_t* status, void* workspace, void* stream) {
    if (!logits_tm || !labels || !label_len || !seq_len || !loss || !grad || !status || !workspace) return ASR_ERR_BAD_ARG;
    if (T < 1 || B < 1 || V < 1 || blank < 0 || blank >= V) return ASR_ERR_BAD_ARG;
    if (max_label < 1 || max_label > 127) return ASR_ERR_UNSUPPORTED;
    const size_t lds = ctc_lds_bytes(T, max_label);
    if (lds > 160 * 1024) return ASR_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    double* lse = (double*)workspace;
    if (((uintptr_t)workspace) & 15) return ASR_ERR_BAD_ARG;
    double* alpha_ws = lse + ((size_t)T * B + 1) / 2 * 2;          // 16-byte aligned rows (pitch: a multiple of four doubles)
    double* beta_ws = alpha_ws + (size_t)B * T * ctc_ws_pitch(max_label);
    double* prob_ws = beta_ws + (size_t)B * T * ctc_ws_pitch(max_label);
    const int p_cap = (int)((lds - ctc_lds_fixed(T, max_label)) / sizeof(double));
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)ctc_lattice_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    const int SW = (int)ctc_ws_pitch(max_label);
    const dim3 rg(asr_cdiv((long)T * B, 4)), rb(256);
    const bool vec = (V & 3) == 0 && V <= 64 * 4 * 8 && ((((uintptr_t)logits_tm) | ((uintptr_t)grad)) & 15) == 0;
#define ASR_CTC_ROWS_ARGS logits_tm, T, B, V, seq_len, status, labels, max_label, label_len, blank, SW, lse, grad, prob_ws
    if (vec && V <= 64 * 4 * 4) hipLaunchKernelGGL(ctc_rows_vec_kernel<4>, rg, rb, 0, st, ASR_CTC_ROWS_ARGS);
    else if (vec && V <= 64 * 4 * 6) hipLaunchKernelGGL(ctc_rows_vec_kernel<6>, rg, rb, 0, st, ASR_CTC_ROWS_ARGS);
    else if (vec) hipLaunchKernelGGL(ctc_rows_vec_kernel<8>, rg, rb, 0, st, ASR_CTC_ROWS_ARGS);
    else hipLaunchKernelGGL(ctc_rows_kernel, rg, rb, 0, st, ASR_CTC_ROWS_ARGS);
#undef ASR_CTC_ROWS_ARGS
    hipLaunchKernelGGL(ctc_lattice_kernel, dim3(B), dim3(512), lds, st, logits_tm, T, B, V, labels, max_label, label_len, seq_len, blank, (const int32_t*)status, (const double*)lse, alpha_ws, beta_ws, prob_ws, loss, grad, p_cap);
    ASR_CHECK_LAUNCH("ctc_loss");
    return ASR_OK;
}

extern "C" size_t asr_ctc_greedy_workspace(int T, int B) { return (size_t)T * B * 8 + 64; }

extern "C" int asr_ctc_greedy(const float* logits_tm, int T, int B, int V, const int32_t* seq_len, int blank,
                              int32_t* out_ids, int32_t* out_len, float* neg_sum_logits, void* workspace, void* stream) {
    if (!logits_tm || !seq_len || !out_ids || !out_len || !neg_sum_logits || !workspace) return ASR_ERR_BAD_ARG;
    if (T < 1 || B < 1 || V < 1) return ASR_ERR_BAD_ARG;
    int32_t* best_k = (int32_t*)workspace;
    float* best_v = (float*)(best_k + (size_t)T * B);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(ctc_argmax_kernel, dim3(asr_cdiv((long)T * B, 4)), dim3(256), 0, st, logits_tm, T, B, V, best_k, best_v);
    hipLaunchKernelGGL(ctc_compact_kernel, dim3(B), dim3(64), 0, st, (const int32_t*)best_k, (const float*)best_v, T, B, seq_len, blank, out_ids, out_len, neg_sum_logits);
    ASR_CHECK_LAUNCH("ctc_greedy");
    return ASR_OK;
}

extern "C" int asr_edit_distance(const int32_t* hyp, int hyp_pitch, const int32_t* hyp_len,
                                 const int32_t* truth, int truth_pitch, const int32_t* truth_len,
                                 int B, float* dist, void* stream) {
    if (!hyp || !hyp_len || !truth || !truth_len || !dist || B < 1 || truth_pitch < 1 || truth_pitch > 4096) return ASR_ERR_BAD_ARG;
    const size_t lds = (size_t)(2 * truth_pitch + 2) * sizeof(int);
    hipLaunchKernelGGL(edit_distance_kernel, dim3(B), dim3(64), lds, (hipStream_t)stream, hyp, hyp_pitch, hyp_len, truth, truth_pitch, truth_len, B, dist);
    ASR_CHECK_LAUNCH("edit_distance");
    return ASR_OK;
}

extern "C" int asr_adam_tf(float* theta, const float* grad, float* m, float* v, size_t n,
                           float lr_t, float beta1, float beta2, float eps, float gscale, void* stream) {
    if (!theta || !grad || !m || !v) return ASR_ERR_BAD_ARG;
    if (n == 0) return ASR_OK;
    long blocks = (long)((n + 3) / 4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(adam_tf_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, theta, grad, m, v, n, lr_t, beta1, beta2, eps, gscale);
    ASR_CHECK_LAUNCH("adam_tf");
    return ASR_OK;
}

// Unmasked fused attention for the time axis of the end2end pre-net (end2end/model.py:134-172 called with
// mask=False at :252): per (batch, channel) softmax(Q K^T / sqrt(d)) V over [T', d] matrices, d = F' = 80.
// Same structure as the Transformer kernels of transformer.hip (scores transposed, S^T = K.Q^T, so the online
// softmax is lane-local and P is already the B operand of the P.V MFMA; fp32 MFMA; scores never reach HBM),
// templated on the head width: d = 80 is 10 contraction groups of 8 and 2.5 output tiles of 32 -- the third
// tile's upper half reads past the row (padding / next row) into accumulator rows that are never stored.
// Tensors are [N][T][H*d] with head h in columns h*d .. h*d+d-1 (the [B][T'][c][F'] layout of asr_plane_to_T).
#include "asr_common.h"
#include <math.h>

namespace {

__device__ __forceinline__ int rowidx(int r, int lh) { return (r & 3) + 8 * (r >> 2) + 4 * lh; }

// scores are kept in base-2 units (the recomputed operand is pre-scaled by log2(e) / sqrt(d)): a probability is one
// v_sub + one v_exp_f32 -- vector instructions beside fp32 MFMAs cost their full issue time (transformer.hip)
constexpr float LOG2E = 1.44269504088896340736f;
__device__ __forceinline__ float ex2(float x) { return __builtin_amdgcn_exp2f(x); }

template <int DH>
struct Cfg {
    static constexpr int KP = DH + 4;             // LDS pitch (floats): conflict-free ds_read_b128
    static constexpr int GK = DH / 8;             // contraction groups of the score MFMAs
    static constexpr int DT = (DH + 31) / 32;     // 32-wide output tiles
    static constexpr int PADF = 32;               // slack behind a tile for the partial last output tile
    static_assert(DH % 8 == 0, "head width must be a multiple of 8");
};

template <int DH>
__device__ __forceinline__ void stage(float* dst, const float* __restrict__ X, long base_row, int row0, int nrows, int T,
                                      int C, int hoff, int tid) {
    constexpr int KP = Cfg<DH>::KP, F4 = DH / 4;
    for (int f = tid; f < nrows * F4; f += 256) {
        const int row = f / F4, c4 = f - row * F4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row0 + row < T) v = *(const float4*)(X + (base_row + row0 + row) * C + hoff + c4 * 4);
        *(float4*)(dst + row * KP + c4 * 4) = v;
    }
}

// transposed accumulator tiles (rows = d in registers, column = token on the lane) -> X[tok][hoff + d], whole rows
template <int DH>
__device__ __forceinline__ void store_T(float* __restrict__ X, float* scratch, const floatx16 (&acc)[Cfg<DH>::DT], float mul_lane,
                                        long base_row, int tok0, int T, int C, int hoff, int lane) {
    constexpr int DT = Cfg<DH>::DT;
    const int li = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int c0 = 0; c0 < DT; c0 += 2) {                 // 64 columns per pass through the 32 x 65 scratch
#pragma unroll
        for (int dt = c0; dt < c0 + 2 && dt < DT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) scratch[li * 65 + (dt - c0) * 32 + rowidx(r, lh)] = acc[dt][r] * mul_lane;
        __builtin_amdgcn_s_waitcnt(0xc07f);              // lgkmcnt(0): wave-private scratch
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int row = it * 4 + (lane >> 4), c4 = lane & 15;
            const int col = c0 * 32 + c4 * 4;
            if (tok0 + row < T && col < DH) {
                const float* s = scratch + row * 65 + c4 * 4;
                *(float4*)(X + (base_row + tok0 + row) * C + hoff + col) = make_float4(s[0], s[1], s[2], s[3]);
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
    }
}

template <int DH>
__global__ __launch_bounds__(256, 2) void nm_fwd_kernel(const float* __restrict__ Q, const float* __restrict__ K,
                                                        const float* __restrict__ V, float* __restrict__ O,
                                                        float* __restrict__ lse, int Tq, int Tk, int C, int H, float scale) {
    const float scale2 = scale * LOG2E;
    constexpr int KP = Cfg<DH>::KP, GK = Cfg<DH>::GK, DT = Cfg<DH>::DT;
    __shared__ __attribute__((aligned(16))) float kv_lds[2 * (64 * KP + Cfg<DH>::PADF)];
    static_assert(4 * 32 * 65 <= 2 * (64 * KP + Cfg<DH>::PADF), "transpose scratch must fit in the K/V tiles");
    float* Ks = kv_lds;
    float* Vs = kv_lds + 64 * KP + Cfg<DH>::PADF;
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, lh = lane >> 5, wave = tid >> 6;
    const int head = blockIdx.y, n = blockIdx.z, hoff = head * DH;
    const int q0 = blockIdx.x * 128 + wave * 32, q = q0 + li;
    const long qbase = (long)n * Tq, kbase = (long)n * Tk;

    float qreg[GK * 4];
#pragma unroll
    for (int g = 0; g < GK; ++g) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (q < Tq) v = *(const float4*)(Q + (qbase + q) * C + hoff + 8 * g + 4 * lh);
        qreg[g * 4 + 0] = v.x * scale2; qreg[g * 4 + 1] = v.y * scale2; qreg[g * 4 + 2] = v.z * scale2; qreg[g * 4 + 3] = v.w * scale2;
    }
    floatx16 oacc[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[dt][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    for (int k0 = 0; k0 < Tk; k0 += 64) {
        __syncthreads();
        stage<DH>(Ks, K, kbase, k0, 64, Tk, C, hoff, tid);
        stage<DH>(Vs, V, kbase, k0, 64, Tk, C, hoff, tid);
        __syncthreads();
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            if (k0 + sub * 32 >= Tk) continue;
            floatx16 s;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
            for (int g = 0; g < GK; ++g) {
                const float4 kv = *(const float4*)(Ks + (sub * 32 + li) * KP + 8 * g + 4 * lh);
                s = __builtin_amdgcn_mfma_f32_32x32x2f32(kv.x, qreg[g * 4 + 0], s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x2f32(kv.y, qreg[g * 4 + 1], s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x2f32(kv.z, qreg[g * 4 + 2], s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x2f32(kv.w, qreg[g * 4 + 3], s, 0, 0, 0);
            }
            float mt = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = k0 + sub * 32 + rowidx(r, lh);
                s[r] = (key < Tk) ? s[r] : -INFINITY;
                mt = fmaxf(mt, s[r]);
            }
            mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
            const float m_new = fmaxf(m_run, mt);
            const float alpha = ex2(m_run - m_new);
            float lt = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = ex2(s[r] - m_new); lt += s[r]; }
            lt += __shfl_xor(lt, 32, 64);
            l_run = l_run * alpha + lt;
            m_run = m_new;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[dt][r] *= alpha;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float* vrow = Vs + (sub * 32 + rowidx(r, lh)) * KP + li;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
                    oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(vrow[dt * 32], s[r], oacc[dt], 0, 0, 0);
            }
        }
    }
    __syncthreads();
    if (q < Tq && lh == 0) lse[((long)n * H + head) * Tq + q] = m_run + log2f(l_run);      // base-2 units
    store_T<DH>(O, Ks + wave * (32 * 65), oacc, 1.f / l_run, qbase, q0, Tq, C, hoff, lane);
}

// delta[n][head][q] = sum_d dO * O   (32 lanes per row)
__global__ void nm_delta_kernel(const float* __restrict__ O, const float* __restrict__ dO, float* __restrict__ delta,
                                int N, int Tq, int C, int H, int DH) {
    const long gid = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 5;
    const int l = threadIdx.x & 31;
    const long total = (long)N * Tq * H;
    if (gid >= total) return;
    const int head = (int)(gid % H);
    const long tok = gid / H;
    float s = 0.f;
    if (l * 4 < DH) {
        const float4 a = *(const float4*)(O + tok * C + head * DH + l * 4);
        const float4 b = *(const float4*)(dO + tok * C + head * DH + l * 4);
        s = a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (l == 0) {
        const int n = (int)(tok / Tq), q = (int)(tok - (long)n * Tq);
        delta[((long)n * H + head) * Tq + q] = s;
    }
}

// dK, dV: one wave owns 32 keys (K, V rows in registers), the block walks the queries in tiles of 32
template <int DH>
__global__ __launch_bounds__(256, 2) void nm_bwd_kv_kernel(const float* __restrict__ Q, const float* __restrict__ K,
                                                           const float* __restrict__ V, const float* __restrict__ dO,
                                                           const float* __restrict__ lse, const float* __restrict__ delta,
                                                           float* __restrict__ dK, float* __restrict__ dV,
                                                           int Tq, int Tk, int C, int H, float scale) {
    const float scale2 = scale * LOG2E;
    constexpr int KP = Cfg<DH>::KP, GK = Cfg<DH>::GK, DT = Cfg<DH>::DT;
    __shared__ __attribute__((aligned(16))) float Qs[32 * KP + Cfg<DH>::PADF];
    __shared__ __attribute__((aligned(16))) float Ds[32 * KP + Cfg<DH>::PADF];
    __shared__ float lse_s[32], del_s[32];
    __shared__ float scratch[4 * 32 * 65];
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, lh = lane >> 5, wave = tid >> 6;
    const int head = blockIdx.y, n = blockIdx.z, hoff = head * DH;
    const int k0 = blockIdx.x * 128 + wave * 32, key = k0 + li;
    const long qbase = (long)n * Tq, kbase = (long)n * Tk;

    float kreg[GK * 4], vreg[GK * 4];
#pragma unroll
    for (int g = 0; g < GK; ++g) {
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
        if (key < Tk) {
            a = *(const float4*)(K + (kbase + key) * C + hoff + 8 * g + 4 * lh);
            b = *(const float4*)(V + (kbase + key) * C + hoff + 8 * g + 4 * lh);
        }
        kreg[g * 4 + 0] = a.x * scale2; kreg[g * 4 + 1] = a.y * scale2; kreg[g * 4 + 2] = a.z * scale2; kreg[g * 4 + 3] = a.w * scale2;
        vreg[g * 4 + 0] = b.x; vreg[g * 4 + 1] = b.y; vreg[g * 4 + 2] = b.z; vreg[g * 4 + 3] = b.w;
    }
    floatx16 dk[DT], dv[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk[dt][r] = 0.f; dv[dt][r] = 0.f; }

    for (int q0 = 0; q0 < Tq; q0 += 32) {
        __syncthreads();
        // (the thread index re-derived per trip: the staging addresses hipcc hoists out of this loop otherwise live across its 250
        //  registers of accumulators and operands and come back as twelve scratch reloads per trip, each behind an s_waitcnt vmcnt(0))
        int tid = (int)threadIdx.x;
        asm volatile("" : "+v"(tid));
        stage<DH>(Qs, Q, qbase, q0, 32, Tq, C, hoff, tid);
        stage<DH>(Ds, dO, qbase, q0, 32, Tq, C, hoff, tid);
        if (tid < 32) {
            const int q = q0 + tid;
            lse_s[tid] = (q < Tq) ? lse[((long)n * H + head) * Tq + q] : INFINITY;
            del_s[tid] = (q < Tq) ? delta[((long)n * H + head) * Tq + q] : 0.f;
        }
        __syncthreads();
        floatx16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
        for (int g = 0; g < GK; ++g) {
            const float4 qv = *(const float4*)(Qs + li * KP + 8 * g + 4 * lh);
            const float4 dv4 = *(const float4*)(Ds + li * KP + 8 * g + 4 * lh);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(qv.x, kreg[g * 4 + 0], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(qv.y, kreg[g * 4 + 1], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(qv.z, kreg[g * 4 + 2], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(qv.w, kreg[g * 4 + 3], s, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(dv4.x, vreg[g * 4 + 0], dp, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(dv4.y, vreg[g * 4 + 1], dp, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(dv4.z, vreg[g * 4 + 2], dp, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(dv4.w, vreg[g * 4 + 3], dp, 0, 0, 0);
        }
        // rows of s/dp = queries rowidx(r, lh), column = this lane's key
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ql = rowidx(r, lh);
            const float p = (key < Tk) ? ex2(s[r] - lse_s[ql]) : 0.f;        // lse = +inf for rows past Tq -> p = 0
            const float ds = p * (dp[r] - del_s[ql]) * scale;
            s[r] = p; dp[r] = ds;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float* drow = Ds + rowidx(r, lh) * KP + li;
            const float* qrow = Qs + rowidx(r, lh) * KP + li;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                dv[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(drow[dt * 32], s[r], dv[dt], 0, 0, 0);
                dk[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(qrow[dt * 32], dp[r], dk[dt], 0, 0, 0);
            }
        }
    }
    store_T<DH>(dK, scratch + wave * (32 * 65), dk, 1.f, kbase, k0, Tk, C, hoff, lane);
    store_T<DH>(dV, scratch + wave * (32 * 65), dv, 1.f, kbase, k0, Tk, C, hoff, lane);
}

// dQ: one wave owns 32 queries (Q, dO rows in registers), the block walks the keys in tiles of 64
template <int DH>
__global__ __launch_bounds__(256, 2) void nm_bwd_q_kernel(const float* __restrict__ Q, const float* __restrict__ K,
                                                          const float* __restrict__ V, const float* __restrict__ dO,
                                                          const float* __restrict__ lse, const float* __restrict__ delta,
                                                          float* __restrict__ dQ, int Tq, int Tk, int C, int H, float scale) {
    const float scale2 = scale * LOG2E;
    constexpr int KP = Cfg<DH>::KP, GK = Cfg<DH>::GK, DT = Cfg<DH>::DT;
    __shared__ __attribute__((aligned(16))) float kv_lds[2 * (64 * KP + Cfg<DH>::PADF)];
    float* Ks = kv_lds;
    float* Vs = kv_lds + 64 * KP + Cfg<DH>::PADF;
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, lh = lane >> 5, wave = tid >> 6;
    const int head = blockIdx.y, n = blockIdx.z, hoff = head * DH;
    const int q0 = blockIdx.x * 128 + wave * 32, q = q0 + li;
    const long qbase = (long)n * Tq, kbase = (long)n * Tk;

    float qreg[GK * 4], doreg[GK * 4];
#pragma unroll
    for (int g = 0; g < GK; ++g) {
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
        if (q < Tq) {
            a = *(const float4*)(Q + (qbase + q) * C + hoff + 8 * g + 4 * lh);
            b = *(const float4*)(dO + (qbase + q) * C + hoff + 8 * g + 4 * lh);
        }
        qreg[g * 4 + 0] = a.x * scale2; qreg[g * 4 + 1] = a.y * scale2; qreg[g * 4 + 2] = a.z * scale2; qreg[g * 4 + 3] = a.w * scale2;
        doreg[g * 4 + 0] = b.x; doreg[g * 4 + 1] = b.y; doreg[g * 4 + 2] = b.z; doreg[g * 4 + 3] = b.w;
    }
    const float my_lse = (q < Tq) ? lse[((long)n * H + head) * Tq + q] : INFINITY;
    const float my_del = (q < Tq) ? delta[((long)n * H + head) * Tq + q] : 0.f;
    floatx16 dq[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[dt][r] = 0.f;

    for (int k0 = 0; k0 < Tk; k0 += 64) {
        __syncthreads();
        stage<DH>(Ks, K, kbase, k0, 64, Tk, C, hoff, tid);
        stage<DH>(Vs, V, kbase, k0, 64, Tk, C, hoff, tid);
        __syncthreads();
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            if (k0 + sub * 32 >= Tk) continue;
            floatx16 s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
            for (int g = 0; g < GK; ++g) {
                const float4 kv = *(const float4*)(Ks + (sub * 32 + li) * KP + 8 * g + 4 * lh);
                const float4 vv = *(const float4*)(Vs + (sub * 32 + li) * KP + 8 * g + 4 * lh);
                s = __builtin_amdgcn_mfma_f32_32x32x2f32(kv.x, qreg[g * 4 + 0], s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x2f32(kv.y, qreg[g * 4 + 1], s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x2f32(kv.z, qreg[g * 4 + 2], s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x2f32(kv.w, qreg[g * 4 + 3], s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x2f32(vv.x, doreg[g * 4 + 0], dp, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x2f32(vv.y, doreg[g * 4 + 1], dp, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x2f32(vv.z, doreg[g * 4 + 2], dp, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x2f32(vv.w, doreg[g * 4 + 3], dp, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = k0 + sub * 32 + rowidx(r, lh);
                const float p = (key < Tk) ? ex2(s[r] - my_lse) : 0.f;
                dp[r] = p * (dp[r] - my_del) * scale;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float* krow = Ks + (sub * 32 + rowidx(r, lh)) * KP + li;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
                    dq[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(krow[dt * 32], dp[r], dq[dt], 0, 0, 0);
            }
        }
    }
    __syncthreads();
    store_T<DH>(dQ, Ks + wave * (32 * 65), dq, 1.f, qbase, q0, Tq, C, hoff, lane);
}

}  // namespace

// ===================================================================== C ABI
extern "C" int asr_attention_nomask_fwd(const float* Q, const float* K, const float* V, int N, int Tq, int Tk, int C, int H,
                                        float* O, float* lse, void* stream) {
    if (!Q || !K || !V || !O || !lse || N < 1 || Tq < 1 || Tk < 1 || H < 1 || C != H * 80) return ASR_ERR_BAD_ARG;
    dim3 grid(asr_cdiv(Tq, 128), H, N);
    const float scale = 1.0f / sqrtf(80.0f);
    hipLaunchKernelGGL(nm_fwd_kernel<80>, grid, dim3(256), 0, (hipStream_t)stream, Q, K, V, O, lse, Tq, Tk, C, H, scale);
    ASR_CHECK_LAUNCH("attention_nomask_fwd");
    return ASR_OK;
}

extern "C" int asr_attention_nomask_bwd(const float* Q, const float* K, const float* V, const float* O, const float* dO,
                                        const float* lse, int N, int Tq, int Tk, int C, int H,
                                        float* dQ, float* dK, float* dV, float* delta_ws, void* stream) {
    if (!Q || !K || !V || !O || !dO || !lse || !dQ || !dK || !dV || !delta_ws) return ASR_ERR_BAD_ARG;
    if (N < 1 || Tq < 1 || Tk < 1 || H < 1 || C != H * 80) return ASR_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    const float scale = 1.0f / sqrtf(80.0f);
    const long rows = (long)N * Tq * H;
    hipLaunchKernelGGL(nm_delta_kernel, dim3(asr_cdiv(rows * 32, 256)), dim3(256), 0, st, O, dO, delta_ws, N, Tq, C, H, 80);
    dim3 gkv(asr_cdiv(Tk, 128), H, N), gq(asr_cdiv(Tq, 128), H, N);
    hipLaunchKernelGGL(nm_bwd_kv_kernel<80>, gkv, dim3(256), 0, st, Q, K, V, dO, lse, (const float*)delta_ws, dK, dV, Tq, Tk, C, H, scale);
    hipLaunchKernelGGL(nm_bwd_q_kernel<80>, gq, dim3(256), 0, st, Q, K, V, dO, lse, (const float*)delta_ws, dQ, Tq, Tk, C, H, scale);
    ASR_CHECK_LAUNCH("attention_nomask_bwd");
    return ASR_OK;
}

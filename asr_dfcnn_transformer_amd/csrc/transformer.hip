// Transformer pieces of end2end/transformer.py on gfx950:
//   * fused multi-head attention forward / backward on the fp32 MFMA pipe with the reference's
//     mask semantics (key mask = zero rows of the per-head K, fill value -2^32+1, optional
//     lower-triangular mask, query mask applied after the softmax): scores never touch HBM;
//   * (residual add +) LayerNorm forward / backward (eps 1e-8, biased variance);
//   * embedding gather (+ sqrt(d) scale, zero_pad row, learned position table) and its
//     deterministic scatter (sorted index lists from the host: no float atomics);
//   * label-smoothed softmax cross-entropy with the reference's masking (-1 targets count).
// The projection / FFN / vocabulary GEMMs are the 1-tap tap_gemm / tap_wgrad kernels.
//
// Attention layout trick: the score tile is computed TRANSPOSED, S^T = K.Q^T (keys in
// registers, query on the lane), so the row max / row sum of the online softmax are
// lane-local (+1 exchange between the two half-waves), and the P tile sits in exactly the
// register layout the next MFMA needs as its B operand (contraction index = register
// index): P never moves between lanes or through LDS.
#include "asr_common.h"
#include "reduce.h"
#include <math.h>

namespace {

constexpr float MASK_FILL = -4294967296.0f;     // float32(-2**32 + 1)
constexpr int DH = 64;                          // head width (hidden_units / num_heads = 512 / 8)
constexpr int KP = DH + 4;                      // LDS pitch of tiles read with ds_read_b128
// The fp32 MFMA shares the SIMD's vector issue with every other vector instruction (tools/mfma_valu.hip: an MFMA-only
// wave and a VALU-only wave on one SIMD take the SUM of their times), so each vector instruction of the softmax costs
// its full issue time.  The scores are therefore kept in base-2 units (q pre-scaled by log2(e) / sqrt(d)): one v_sub +
// one v_exp_f32 per element instead of expf's 13 instructions, and the key mask is one compare + select against a per-key
// bias staged with the tile (a wave-uniform branch around the mask code cost 300 spilled registers instead).
constexpr float LOG2E = 1.44269504088896340736f;
constexpr float QSCALE2 = 0.125f * LOG2E;        // 1 / sqrt(64), in base-2 units
constexpr float FILL2 = MASK_FILL * LOG2E;       // the fill value in the same units
__device__ __forceinline__ float ex2(float x) { return __builtin_amdgcn_exp2f(x); }

__device__ __forceinline__ int rowidx(int r, int lh) { return (r & 3) + 8 * (r >> 2) + 4 * lh; }

// Counter-based dropout mask (tf.layers.dropout, transformer.py:111,154,226; model.py:290): element `idx` of the tensor
// drawn for `seed` is kept when the top 24 bits of a murmur3-finalised hash reach the threshold rate * 2^24; kept values
// are scaled by 1 / (1 - rate).  The same function regenerates the mask in the backward pass (nothing is stored) and
// in oracle/transformer.py.  TensorFlow's own random stream cannot be reproduced; parity is against this generator.
__device__ __forceinline__ bool drop_keep(uint32_t idx, uint32_t seed, uint32_t thr) {
    uint32_t h = idx * 0x9E3779B1u + seed;
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return (h >> 8) >= thr;
}

// stage a [rows x 64] head slice of X[n][t][C] into LDS with pitch KP; rows beyond T are zero.
// 16 consecutive lanes own one row, so a 16-lane xor-reduction gives per-row statistics: the row sum (stat_mode 0), the
// sum of magnitudes (1), or the key bias of the row (2): 0 for a real key, the fill value for a key-masked one (zero row
// sum), -inf for a row past the end -- what a score is replaced by when the bias is not 0.
__device__ __forceinline__ void stage_tile(float* dst, const float* __restrict__ X, long base_row, int row0, int nrows,
                                           int T, int C, int hoff, int tid, float scale, float* rowstat, int stat_mode) {
    for (int f = tid; f < nrows * 16; f += 256) {
        const int row = f >> 4, c4 = f & 15;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row0 + row < T) v = *(const float4*)(X + (base_row + row0 + row) * C + hoff + c4 * 4);
        if (rowstat) {
            float s = (stat_mode != 1) ? (v.x + v.y + v.z + v.w) : (fabsf(v.x) + fabsf(v.y) + fabsf(v.z) + fabsf(v.w));
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
            if (stat_mode == 2) s = (row0 + row < T) ? (s != 0.f ? 0.f : FILL2) : -INFINITY;
            if (c4 == 0) rowstat[row] = s;
        }
        v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
        *(float4*)(dst + row * KP + c4 * 4) = v;
    }
}

// write a wave's transposed accumulator tile (rows = d in registers, column = token on the lane)
// to X[n][tok][hoff + d] through an LDS transpose so the global stores are whole rows.
// relu_src (optional): the post-ReLU tensor this gradient belongs to; the stored value is masked by (relu_src > 0).
__device__ __forceinline__ void store_tile_T(float* __restrict__ X, float* scratch, const floatx16 (&acc)[2], float mul_lane,
                                             long base_row, int tok0, int T, int C, int hoff, int lane,
                                             const float* __restrict__ relu_src = nullptr) {
    const int li = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) scratch[li * 65 + dt * 32 + rowidx(r, lh)] = acc[dt][r] * mul_lane;
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): this wave's LDS writes are done (wave-private region)
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int row = it * 4 + (lane >> 4), c4 = lane & 15;
        if (tok0 + row < T) {
            const float* s = scratch + row * 65 + c4 * 4;
            float4 o = make_float4(s[0], s[1], s[2], s[3]);
            if (relu_src) {
                const float4 h = *(const float4*)(relu_src + (base_row + tok0 + row) * C + hoff + c4 * 4);
                o.x = h.x > 0.f ? o.x : 0.f; o.y = h.y > 0.f ? o.y : 0.f; o.z = h.z > 0.f ? o.z : 0.f; o.w = h.w > 0.f ? o.w : 0.f;
            }
            *(float4*)(X + (base_row + tok0 + row) * C + hoff + c4 * 4) = o;
        }
    }
}

// Workgroup -> (tile, head, sample).  Plain launches use grid (tiles, H, N): the tiles of one (sample, head) run
// together and share its K/V in L2.  Causal launches use grid (N*H, tiles): their tiles do unequal work (the key
// range ends at the diagonal), so the longest tiles are dispatched first across all (sample, head) pairs --
// longest-first keeps the last round full; with the plain order the skipped tiles bought no time at all.
template <bool CAUSAL>
__device__ __forceinline__ void attn_block_coords(int H, bool longest_is_last, int& tile, int& head, int& n, int& N) {
    if (CAUSAL) {
        head = blockIdx.x % H; n = blockIdx.x / H; N = gridDim.x / H;
        tile = longest_is_last ? (int)(gridDim.y - 1 - blockIdx.y) : (int)blockIdx.y;
    } else {
        tile = blockIdx.x; head = blockIdx.y; n = blockIdx.z; N = gridDim.z;
    }
}

// ------------------------------------------------------------------ attention forward
template <bool CAUSAL, bool DROP>
__global__ __launch_bounds__(256, CAUSAL ? 2 : 3) void attn_fwd_kernel(const float* __restrict__ Q, const float* __restrict__ K,
                                                       const float* __restrict__ V, float* __restrict__ O,
                                                       float* __restrict__ lse, int Tq, int Tk, int C, int H, int ldq, int ldk,
                                                       uint32_t drop_thr, uint32_t drop_seed, float drop_scale) {
    __shared__ __attribute__((aligned(16))) float kv_lds[2 * 64 * KP];   // also the epilogue's transpose scratch
    float* Ks = kv_lds;
    float* Vs = kv_lds + 64 * KP;
    __shared__ float kstat[64];
    static_assert(4 * 32 * 65 <= 2 * 64 * KP, "transpose scratch must fit in the K/V tiles");
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, lh = lane >> 5, wave = tid >> 6;
    int qtile, head, n, Nn;
    attn_block_coords<CAUSAL>(H, true, qtile, head, n, Nn);
    const int hoff = head * DH;
    const int q0 = qtile * 128 + wave * 32, q = q0 + li;
    const long qbase = (long)n * Tq, kbase = (long)n * Tk;

    float qreg[32];
    float qabs = 0.f;
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (q < Tq) v = *(const float4*)(Q + (qbase + q) * ldq + hoff + 8 * g + 4 * lh);
        qabs += fabsf(v.x) + fabsf(v.y) + fabsf(v.z) + fabsf(v.w);
        qreg[g * 4 + 0] = v.x * QSCALE2; qreg[g * 4 + 1] = v.y * QSCALE2;
        qreg[g * 4 + 2] = v.z * QSCALE2; qreg[g * 4 + 3] = v.w * QSCALE2;
    }
    qabs += __shfl_xor(qabs, 32, 64);
    const float qmask = (qabs != 0.f) ? 1.f : 0.f;

    floatx16 oacc[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { oacc[0][r] = 0.f; oacc[1][r] = 0.f; }
    float m_run = -INFINITY, l_run = 0.f;

    const int qlast_blk = qtile * 128 + 127;
    for (int k0 = 0; k0 < Tk; k0 += 64) {
        // A tile that lies entirely in the future of every query of this workgroup holds only the fill value:
        // it contributes exp(fill - max) = 0 to a row that has already met a real score.  It may be skipped only
        // when that holds for EVERY row (a row whose keys so far were all key-masked must still see it: TF's
        // softmax is uniform over all fill entries, future ones included).
        if (CAUSAL && k0 > qlast_blk) {
            if (__syncthreads_and((q >= Tq) || (m_run > -1.0e9f))) break;
        }
        __syncthreads();
        stage_tile(Ks, K, kbase, k0, 64, Tk, ldk, hoff, tid, 1.f, kstat, 2);
        stage_tile(Vs, V, kbase, k0, 64, Tk, ldk, hoff, tid, 1.f, nullptr, 0);
        __syncthreads();
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            // same rule per wave: 32 keys that are future to all 32 queries of this wave (no barrier in this loop)
            if (CAUSAL && k0 + sub * 32 > q0 + 31 && __all((q >= Tq) || (m_run > -1.0e9f))) continue;
            floatx16 s;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const float4 kv = *(const float4*)(Ks + (sub * 32 + li) * KP + 8 * g + 4 * lh);
                s = __builtin_amdgcn_mfma_f32_32x32x2f32(kv.x, qreg[g * 4 + 0], s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x2f32(kv.y, qreg[g * 4 + 1], s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x2f32(kv.z, qreg[g * 4 + 2], s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x2f32(kv.w, qreg[g * 4 + 3], s, 0, 0, 0);
            }
            float mt = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kl = sub * 32 + rowidx(r, lh);
                const float kb = kstat[kl];                      // key bias (stage_tile, mode 2)
                float v = s[r];
                if (CAUSAL) v = (k0 + kl <= q) ? v : FILL2;
                v = (kb == 0.f) ? v : kb;
                s[r] = v;
                mt = fmaxf(mt, v);
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
            for (int r = 0; r < 16; ++r) { oacc[0][r] *= alpha; oacc[1][r] *= alpha; }
            if (DROP) {            // dropout of the attention weights (transformer.py:111): after the row sum, before P.V
                const uint32_t base = (uint32_t)(((n * H + head) * Tq + q) * Tk + k0 + sub * 32);
#pragma unroll
                for (int r = 0; r < 16; ++r) s[r] = drop_keep(base + rowidx(r, lh), drop_seed, drop_thr) ? s[r] * drop_scale : 0.f;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float* vrow = Vs + (sub * 32 + rowidx(r, lh)) * KP + li;
                oacc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(vrow[0], s[r], oacc[0], 0, 0, 0);
                oacc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(vrow[32], s[r], oacc[1], 0, 0, 0);
            }
        }
    }
    __syncthreads();
    if (q < Tq && lh == 0) {      // kept as (max, log-sum) pair, both in base-2 units: max may be the fill, which would swallow log(l)
        lse[((long)n * H + head) * Tq + q] = m_run;
        lse[(long)Nn * H * Tq + ((long)n * H + head) * Tq + q] = log2f(l_run);
    }
    store_tile_T(O, Ks + wave * (32 * 65), oacc, qmask / l_run, qbase, q0, Tq, C, hoff, lane);
}

// delta[n][head][q] = sum_d dO*O  (one 16-lane group per (q, head))
__global__ void attn_delta_kernel(const float* __restrict__ O, const float* __restrict__ dO, float* __restrict__ delta,
                                  int N, int Tq, int C, int H) {
    const long gid = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const int c4 = threadIdx.x & 15;
    const long total = (long)N * Tq * H;
    float s = 0.f;
    if (gid < total) {
        const int head = (int)(gid % H);
        const long row = gid / H;                       // n*Tq + q
        const float4 a = *(const float4*)(O + row * C + head * DH + c4 * 4);
        const float4 b = *(const float4*)(dO + row * C + head * DH + c4 * 4);
        s = a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (gid < total && c4 == 0) {
        const int head = (int)(gid % H);
        const long row = gid / H;
        const int n = (int)(row / Tq), q = (int)(row - (long)n * Tq);
        delta[((long)n * H + head) * Tq + q] = s;
    }
}

// ------------------------------------------------------------------ attention backward: dK, dV
// One wave owns 32 keys (K, V in registers); the block walks the queries in tiles of 64.
template <bool CAUSAL, bool DROP>
__global__ __launch_bounds__(256, 2) void attn_bwd_kv_kernel(const float* __restrict__ Q, const float* __restrict__ K,
                                                          const float* __restrict__ V, const float* __restrict__ dO,
                                                          const float* __restrict__ lse, const float* __restrict__ delta,
                                                          float* __restrict__ dK, float* __restrict__ dV,
                                                          int Tq, int Tk, int C, int H, int ldq, int ldk, int relu_grad,
                                                          uint32_t drop_thr, uint32_t drop_seed, float drop_scale) {
    constexpr int QT = 64;                       // queries staged per barrier round (two 32-row MFMA sub-tiles)
    __shared__ __attribute__((aligned(16))) float Qs[QT * KP];
    __shared__ __attribute__((aligned(16))) float Ds[QT * KP];
    __shared__ float qstat[QT];
    __shared__ float lse_s[QT], lsl_s[QT], del_s[QT];
    __shared__ float scratch[4 * 32 * 65];
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, lh = lane >> 5, wave = tid >> 6;
    int ktile, head, n, Nn;
    attn_block_coords<CAUSAL>(H, false, ktile, head, n, Nn);      // causal: the first key tile sees every query
    const int hoff = head * DH;
    const int k0 = ktile * 128 + wave * 32, key = k0 + li;
    const long qbase = (long)n * Tq, kbase = (long)n * Tk;

    float kreg[32], vreg[32];
    float ksum = 0.f;
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
        if (key < Tk) {
            a = *(const float4*)(K + (kbase + key) * ldk + hoff + 8 * g + 4 * lh);
            b = *(const float4*)(V + (kbase + key) * ldk + hoff + 8 * g + 4 * lh);
        }
        ksum += a.x + a.y + a.z + a.w;
        // K only feeds the score recomputation here: pre-scaled to base-2 units of the scaled scores
        kreg[g * 4 + 0] = a.x * QSCALE2; kreg[g * 4 + 1] = a.y * QSCALE2; kreg[g * 4 + 2] = a.z * QSCALE2; kreg[g * 4 + 3] = a.w * QSCALE2;
        vreg[g * 4 + 0] = b.x; vreg[g * 4 + 1] = b.y; vreg[g * 4 + 2] = b.z; vreg[g * 4 + 3] = b.w;
    }
    ksum += __shfl_xor(ksum, 32, 64);
    const bool kkeep = (ksum != 0.f) && (key < Tk);
    const float kfill = (key < Tk) ? FILL2 : -INFINITY;

    floatx16 dk[2], dv[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[0][r] = 0.f; dk[1][r] = 0.f; dv[0][r] = 0.f; dv[1][r] = 0.f; }

    const int kfirst_blk = ktile * 128;
    const long lrow = ((long)n * H + head) * Tq;
    for (int q0 = 0; q0 < Tq; q0 += QT) {
        if (CAUSAL && q0 + QT - 1 < kfirst_blk) {
            // every score of this query tile against this key block is future-masked: dS = 0 (no dK), and P is
            // exp(fill - max) = 0 unless a row's max IS the fill value (all of its keys masked) -- only then dV sees it
            const int q = q0 + (tid & (QT - 1));
            const bool degenerate = (q < Tq) && (lse[lrow + q] < -1.0e9f);
            if (!__syncthreads_or(degenerate)) continue;
        }
        __syncthreads();
        stage_tile(Qs, Q, qbase, q0, QT, Tq, ldq, hoff, tid, 1.f, qstat, 1);
        if (tid < QT) {
            const int q = q0 + tid;
            lse_s[tid] = (q < Tq) ? lse[lrow + q] : INFINITY;
            lsl_s[tid] = (q < Tq) ? lse[(long)Nn * H * Tq + lrow + q] : 0.f;
            del_s[tid] = (q < Tq) ? delta[lrow + q] : 0.f;
        }
        __syncthreads();
        // dO' = qmask * dO  (the query mask multiplies the post-softmax matrix)
        for (int f = tid; f < QT * 16; f += 256) {
            const int row = f >> 4, c4 = f & 15;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (q0 + row < Tq && qstat[row] != 0.f) v = *(const float4*)(dO + (qbase + q0 + row) * C + hoff + c4 * 4);
            *(float4*)(Ds + row * KP + c4 * 4) = v;
        }
        __syncthreads();
#pragma unroll
        for (int sub = 0; sub < QT / 32; ++sub) {
            const float* Qt = Qs + sub * 32 * KP;
            const float* Dt = Ds + sub * 32 * KP;
            const float* lse_t = lse_s + sub * 32; const float* lsl_t = lsl_s + sub * 32; const float* del_t = del_s + sub * 32;
            // the same rule per wave: these 32 queries all precede this wave's 32 keys (no barrier inside this loop)
            if (CAUSAL && q0 + sub * 32 + 31 < k0 && !__any(lse_t[li] < -1.0e9f)) continue;
            floatx16 s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const float4 qv = *(const float4*)(Qt + li * KP + 8 * g + 4 * lh);
                const float4 dv4 = *(const float4*)(Dt + li * KP + 8 * g + 4 * lh);
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
            // s <- P (as dV sees it), dp <- dS / 0.125 (the 1 / sqrt(d) is applied when dK is stored)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ql = rowidx(r, lh), q = q0 + sub * 32 + ql;
                const bool keep = kkeep && (!CAUSAL || key <= q);
                const float sv = keep ? s[r] : kfill;            // fill value, or -inf (p = 0) for a key past the end
                const float p = ex2((sv - lse_t[ql]) - lsl_t[ql]);
                float pd = p, dpe = dp[r];
                if (DROP) {        // O = (P o M / (1-rate)) V: dV sees the dropped weights, dP arrives through the same mask
                    const bool dm = drop_keep((uint32_t)(((n * H + head) * Tq + q) * Tk + key), drop_seed, drop_thr);
                    pd = dm ? p * drop_scale : 0.f;
                    dpe = dm ? dpe * drop_scale : 0.f;
                }
                const float ds = keep ? p * (dpe - del_t[ql]) : 0.f;
                s[r] = pd; dp[r] = ds;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float* drow = Dt + rowidx(r, lh) * KP + li;
                const float* qrow = Qt + rowidx(r, lh) * KP + li;
                dv[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(drow[0], s[r], dv[0], 0, 0, 0);
                dv[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(drow[32], s[r], dv[1], 0, 0, 0);
                dk[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(qrow[0], dp[r], dk[0], 0, 0, 0);
                dk[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(qrow[32], dp[r], dk[1], 0, 0, 0);
            }
        }
    }
    store_tile_T(dK, scratch + wave * (32 * 65), dk, 0.125f, kbase, k0, Tk, ldk, hoff, lane, relu_grad ? K : nullptr);
    store_tile_T(dV, scratch + wave * (32 * 65), dv, 1.f, kbase, k0, Tk, ldk, hoff, lane, relu_grad ? V : nullptr);
}

// ------------------------------------------------------------------ attention backward: dQ
template <bool CAUSAL, bool DROP>
__global__ __launch_bounds__(256, 3) void attn_bwd_q_kernel(const float* __restrict__ Q, const float* __restrict__ K,
                                                         const float* __restrict__ V, const float* __restrict__ dO,
                                                         const float* __restrict__ lse, const float* __restrict__ delta,
                                                         float* __restrict__ dQ, int Tq, int Tk, int C, int H, int ldq, int ldk, int relu_grad,
                                                         uint32_t drop_thr, uint32_t drop_seed, float drop_scale) {
    __shared__ __attribute__((aligned(16))) float kv_lds[2 * 64 * KP];   // also the epilogue's transpose scratch
    float* Ks = kv_lds;
    float* Vs = kv_lds + 64 * KP;
    __shared__ float kstat[64];
    static_assert(4 * 32 * 65 <= 2 * 64 * KP, "transpose scratch must fit in the K/V tiles");
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, lh = lane >> 5, wave = tid >> 6;
    int qtile, head, n, Nn;
    attn_block_coords<CAUSAL>(H, true, qtile, head, n, Nn);
    const int hoff = head * DH;
    const int q0 = qtile * 128 + wave * 32, q = q0 + li;
    const long qbase = (long)n * Tq, kbase = (long)n * Tk;

    float qreg[32], doreg[32];
    float qabs = 0.f;
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
        if (q < Tq) {
            a = *(const float4*)(Q + (qbase + q) * ldq + hoff + 8 * g + 4 * lh);
            b = *(const float4*)(dO + (qbase + q) * C + hoff + 8 * g + 4 * lh);
        }
        qabs += fabsf(a.x) + fabsf(a.y) + fabsf(a.z) + fabsf(a.w);
        qreg[g * 4 + 0] = a.x * QSCALE2; qreg[g * 4 + 1] = a.y * QSCALE2; qreg[g * 4 + 2] = a.z * QSCALE2; qreg[g * 4 + 3] = a.w * QSCALE2;
        doreg[g * 4 + 0] = b.x; doreg[g * 4 + 1] = b.y; doreg[g * 4 + 2] = b.z; doreg[g * 4 + 3] = b.w;
    }
    qabs += __shfl_xor(qabs, 32, 64);
    const float qmask = (qabs != 0.f) ? 1.f : 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) doreg[i] *= qmask;
    const float my_lse = (q < Tq) ? lse[((long)n * H + head) * Tq + q] : INFINITY;
    const float my_lsl = (q < Tq) ? lse[(long)Nn * H * Tq + ((long)n * H + head) * Tq + q] : 0.f;
    const float my_del = (q < Tq) ? delta[((long)n * H + head) * Tq + q] : 0.f;

    floatx16 dq[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { dq[0][r] = 0.f; dq[1][r] = 0.f; }

    for (int k0 = 0; k0 < Tk; k0 += 64) {
        if (CAUSAL && k0 > qtile * 128 + 127) break;     // masked scores get no gradient: nothing for dQ there
        __syncthreads();
        stage_tile(Ks, K, kbase, k0, 64, Tk, ldk, hoff, tid, 1.f, kstat, 2);
        stage_tile(Vs, V, kbase, k0, 64, Tk, ldk, hoff, tid, 1.f, nullptr, 0);
        __syncthreads();
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            if (CAUSAL && k0 + sub * 32 > q0 + 31) continue;       // future to this whole wave: dS = 0
            floatx16 s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
            for (int g = 0; g < 8; ++g) {
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
            // dp <- dS / 0.125 (the 1 / sqrt(d) is applied when dQ is stored)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kl = sub * 32 + rowidx(r, lh);
                const float kb = kstat[kl];                      // key bias (stage_tile, mode 2)
                const bool keep = (kb == 0.f) && (!CAUSAL || k0 + kl <= q);
                float sv = s[r];
                if (CAUSAL) sv = (k0 + kl <= q) ? sv : FILL2;
                sv = (kb == 0.f) ? sv : kb;
                const float p = ex2((sv - my_lse) - my_lsl);
                float dpe = dp[r];
                if (DROP) dpe = drop_keep((uint32_t)(((n * H + head) * Tq + q) * Tk + k0 + kl), drop_seed, drop_thr) ? dpe * drop_scale : 0.f;
                dp[r] = keep ? p * (dpe - my_del) : 0.f;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float* krow = Ks + (sub * 32 + rowidx(r, lh)) * KP + li;
                dq[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(krow[0], dp[r], dq[0], 0, 0, 0);
                dq[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(krow[32], dp[r], dq[1], 0, 0, 0);
            }
        }
    }
    __syncthreads();
    store_tile_T(dQ, Ks + wave * (32 * 65), dq, 0.125f, qbase, q0, Tq, ldq, hoff, lane, relu_grad ? Q : nullptr);
}

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
                                                             float* __restrict__ xhat, float* __restrict__ rstd) {
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

constexpr int kLnRows = 64;     // rows per block in the backward (block partials for dgamma/dbeta)

// dx = rstd * (g - mean(g) - xhat * mean(g*xhat)),  g = dy*gamma;  partials[blk][2][C] = sum dy*xhat, sum dy
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ xhat,
                                                     const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                     int rows, int C, float* __restrict__ dx, int accumulate,
                                                     float* __restrict__ partials) {
    extern __shared__ float sm[];        // [4 waves][2][C]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float* mine = sm + wave * 2 * C;
    for (int c = lane; c < 2 * C; c += 64) mine[c] = 0.f;
    const int r0 = blockIdx.x * kLnRows;
    for (int rr = wave; rr < kLnRows; rr += 4) {
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
template <int NV>
__global__ __launch_bounds__(256) void ln_bwd_vec_kernel(const float* __restrict__ dy, const float* __restrict__ xhat,
                                                         const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                         int rows, int C, float* __restrict__ dx, int accumulate,
                                                         float* __restrict__ partials) {
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
    const int r0 = blockIdx.x * kLnRows;
    for (int rr = wave; rr < kLnRows; rr += 4) {
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
            float4 v = make_float4(rs * (d[i].x * g4[i].x - s1 - xh[i].x * s2), rs * (d[i].y * g4[i].y - s1 - xh[i].y * s2),
                                   rs * (d[i].z * g4[i].z - s1 - xh[i].z * s2), rs * (d[i].w * g4[i].w - s1 - xh[i].w * s2));
            float* o = dx + (long)row * C + q * 4;
            if (accumulate) { const float4 p = *(const float4*)o; v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w; }
            *(float4*)o = v;
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
static inline uint32_t drop_threshold(float rate) {
    double t = (double)rate * 16777216.0;
    if (t < 0) t = 0;
    if (t > 16777215.0) t = 16777215.0;
    return (uint32_t)(t + 0.5);
}

extern "C" int asr_attention_fwd_p(const float* Q, const float* K, const float* V, int N, int Tq, int Tk, int C, int H,
                                   int ldq, int ldk, int causal, float dropout_rate, unsigned int seed, float* O, float* lse,
                                   void* stream) {
    if (!Q || !K || !V || !O || !lse || N < 1 || Tq < 1 || Tk < 1 || H < 1 || C != H * DH) return ASR_ERR_BAD_ARG;
    if (ldq < C || ldk < C || (ldq & 3) || (ldk & 3)) return ASR_ERR_BAD_ARG;
    if (dropout_rate < 0.f || dropout_rate >= 1.f || (double)N * H * Tq * Tk >= 4294967296.0) return ASR_ERR_BAD_ARG;
    dim3 grid(asr_cdiv(Tq, 128), H, N), grid_causal(N * H, asr_cdiv(Tq, 128), 1);      // see attn_block_coords
    hipStream_t st = (hipStream_t)stream;
    const uint32_t thr = drop_threshold(dropout_rate);
    const float sc = 1.0f / (1.0f - dropout_rate);
    if (dropout_rate > 0.f) {
        if (causal) hipLaunchKernelGGL((attn_fwd_kernel<true, true>), grid_causal, dim3(256), 0, st, Q, K, V, O, lse, Tq, Tk, C, H, ldq, ldk, thr, seed, sc);
        else hipLaunchKernelGGL((attn_fwd_kernel<false, true>), grid, dim3(256), 0, st, Q, K, V, O, lse, Tq, Tk, C, H, ldq, ldk, thr, seed, sc);
    } else {
        if (causal) hipLaunchKernelGGL((attn_fwd_kernel<true, false>), grid_causal, dim3(256), 0, st, Q, K, V, O, lse, Tq, Tk, C, H, ldq, ldk, thr, seed, sc);
        else hipLaunchKernelGGL((attn_fwd_kernel<false, false>), grid, dim3(256), 0, st, Q, K, V, O, lse, Tq, Tk, C, H, ldq, ldk, thr, seed, sc);
    }
    ASR_CHECK_LAUNCH("attention_fwd");
    return ASR_OK;
}

extern "C" int asr_attention_fwd(const float* Q, const float* K, const float* V, int N, int Tq, int Tk, int C, int H,
                                 int causal, float dropout_rate, unsigned int seed, float* O, float* lse, void* stream) {
    return asr_attention_fwd_p(Q, K, V, N, Tq, Tk, C, H, C, C, causal, dropout_rate, seed, O, lse, stream);
}

extern "C" int asr_attention_bwd_p(const float* Q, const float* K, const float* V, const float* O, const float* dO,
                                   const float* lse, int N, int Tq, int Tk, int C, int H, int ldq, int ldk, int causal,
                                   int relu_grad, float dropout_rate, unsigned int seed,
                                   float* dQ, float* dK, float* dV, float* delta_ws, void* stream) {
    if (ldq < C || ldk < C || (ldq & 3) || (ldk & 3)) return ASR_ERR_BAD_ARG;
    if (!Q || !K || !V || !O || !dO || !lse || !dQ || !dK || !dV || !delta_ws) return ASR_ERR_BAD_ARG;
    if (N < 1 || Tq < 1 || Tk < 1 || H < 1 || C != H * DH) return ASR_ERR_BAD_ARG;
    if (dropout_rate < 0.f || dropout_rate >= 1.f || (double)N * H * Tq * Tk >= 4294967296.0) return ASR_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    const long groups = (long)N * Tq * H;
    hipLaunchKernelGGL(attn_delta_kernel, dim3(asr_cdiv(groups * 16, 256)), dim3(256), 0, st, O, dO, delta_ws, N, Tq, C, H);
    dim3 gkv(asr_cdiv(Tk, 128), H, N), gq(asr_cdiv(Tq, 128), H, N);
    dim3 ckv(N * H, asr_cdiv(Tk, 128), 1), cq(N * H, asr_cdiv(Tq, 128), 1);
    const float* dl = (const float*)delta_ws;
    const uint32_t thr = drop_threshold(dropout_rate);
    const float sc = 1.0f / (1.0f - dropout_rate);
#define ASR_ATTN_BWD(CA, DR, GKV, GQ)                                                                                          \
    do {                                                                                                                       \
        hipLaunchKernelGGL((attn_bwd_kv_kernel<CA, DR>), GKV, dim3(256), 0, st, Q, K, V, dO, lse, dl, dK, dV, Tq, Tk, C, H,    \
                           ldq, ldk, relu_grad, thr, seed, sc);                                                                          \
        hipLaunchKernelGGL((attn_bwd_q_kernel<CA, DR>), GQ, dim3(256), 0, st, Q, K, V, dO, lse, dl, dQ, Tq, Tk, C, H,           \
                           ldq, ldk, relu_grad, thr, seed, sc);                                                                          \
    } while (0)
    if (dropout_rate > 0.f) {
        if (causal) ASR_ATTN_BWD(true, true, ckv, cq); else ASR_ATTN_BWD(false, true, gkv, gq);
    } else {
        if (causal) ASR_ATTN_BWD(true, false, ckv, cq); else ASR_ATTN_BWD(false, false, gkv, gq);
    }
#undef ASR_ATTN_BWD
    ASR_CHECK_LAUNCH("attention_bwd");
    return ASR_OK;
}

extern "C" int asr_attention_bwd(const float* Q, const float* K, const float* V, const float* O, const float* dO,
                                 const float* lse, int N, int Tq, int Tk, int C, int H, int causal, int relu_grad,
                                 float dropout_rate, unsigned int seed,
                                 float* dQ, float* dK, float* dV, float* delta_ws, void* stream) {
    return asr_attention_bwd_p(Q, K, V, O, dO, lse, N, Tq, Tk, C, H, C, C, causal, relu_grad, dropout_rate, seed, dQ, dK, dV,
                               delta_ws, stream);
}

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

extern "C" int asr_copy2d(float* dst, int ldd, const float* src, int lds, int rows, int cols, int accumulate, void* stream) {
    if (!dst || !src || rows < 1 || cols < 4 || (cols & 3) || (ldd & 3) || (lds & 3) || ldd < cols || lds < cols) return ASR_ERR_BAD_ARG;
    if (((uintptr_t)dst | (uintptr_t)src) & 15) return ASR_ERR_BAD_ARG;
    hipLaunchKernelGGL(copy2d_kernel, dim3(grid_for((long)rows * (cols / 4), 256)), dim3(256), 0, (hipStream_t)stream, dst, ldd, src,
                       lds, rows, cols / 4, accumulate);
    ASR_CHECK_LAUNCH("copy2d");
    return ASR_OK;
}

// several strided copies in ONE launch: the table (asr_copy2d_item, device memory) is built once by the caller
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

extern "C" int asr_add_layernorm_fwd(const float* a, const float* b, const float* gamma, const float* beta, int rows, int C,
                                     float eps, float* y, float* xhat, float* rstd, void* stream) {
    if (!a || !gamma || !beta || !y || !xhat || !rstd || rows < 1 || C < 1) return ASR_ERR_BAD_ARG;
    const bool vec = (C & 3) == 0 &&
        ((((uintptr_t)a | (uintptr_t)b | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)y | (uintptr_t)xhat)) & 15) == 0;
    const dim3 grid(asr_cdiv(rows, 4));
    hipStream_t st = (hipStream_t)stream;
    if (vec && C <= 256) hipLaunchKernelGGL(add_ln_fwd_vec_kernel<1>, grid, dim3(256), 0, st, a, b, gamma, beta, rows, C, eps, y, xhat, rstd);
    else if (vec && C <= 512) hipLaunchKernelGGL(add_ln_fwd_vec_kernel<2>, grid, dim3(256), 0, st, a, b, gamma, beta, rows, C, eps, y, xhat, rstd);
    else if (vec && C <= 2048) hipLaunchKernelGGL(add_ln_fwd_vec_kernel<8>, grid, dim3(256), 0, st, a, b, gamma, beta, rows, C, eps, y, xhat, rstd);
    else hipLaunchKernelGGL(add_ln_fwd_kernel, grid, dim3(256), 0, st, a, b, gamma, beta, rows, C, eps, y, xhat, rstd);
    ASR_CHECK_LAUNCH("add_layernorm_fwd");
    return ASR_OK;
}

extern "C" size_t asr_layernorm_bwd_workspace(int rows, int C) {
    const size_t nblk = (size_t)asr_cdiv(rows, kLnRows);
    return (nblk * 2 * C + asr_reduce::colsum_tmp_floats((int)nblk, 2 * C) + 16) * sizeof(float);
}

extern "C" int asr_layernorm_bwd(const float* dy, const float* xhat, const float* rstd, const float* gamma, int rows, int C,
                                 float* dx, int accumulate, float* dgamma, float* dbeta, float* partials, void* stream) {
    if (!dy || !xhat || !rstd || !gamma || !dx || !dgamma || !dbeta || !partials || rows < 1 || C < 1) return ASR_ERR_BAD_ARG;
    if ((size_t)8 * C * sizeof(float) > 64 * 1024) return ASR_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const int nblk = asr_cdiv(rows, kLnRows);
    const size_t lds = (size_t)8 * C * sizeof(float);
    const bool vec = (C & 3) == 0 && ((((uintptr_t)dy | (uintptr_t)xhat | (uintptr_t)dx | (uintptr_t)gamma)) & 15) == 0;
    if (vec && C <= 256) hipLaunchKernelGGL(ln_bwd_vec_kernel<1>, dim3(nblk), dim3(256), lds, st, dy, xhat, rstd, gamma, rows, C, dx, accumulate, partials);
    else if (vec && C <= 512) hipLaunchKernelGGL(ln_bwd_vec_kernel<2>, dim3(nblk), dim3(256), lds, st, dy, xhat, rstd, gamma, rows, C, dx, accumulate, partials);
    else if (vec && C <= 2048) hipLaunchKernelGGL(ln_bwd_vec_kernel<8>, dim3(nblk), dim3(256), lds, st, dy, xhat, rstd, gamma, rows, C, dx, accumulate, partials);
    else hipLaunchKernelGGL(ln_bwd_kernel, dim3(nblk), dim3(256), lds, st, dy, xhat, rstd, gamma, rows, C, dx, accumulate, partials);
    ASR_CHECK_LAUNCH("layernorm_bwd");
    float* tmp = partials + (size_t)nblk * 2 * C;
    asr_reduce::Multi m;
    m.nseg = 2; m.width[0] = C; m.width[1] = C; m.width[2] = 0; m.width[3] = 0;
    m.out[0] = dgamma; m.out[1] = dbeta; m.out[2] = nullptr; m.out[3] = nullptr;
    return asr_reduce::colsum_multi(partials, nblk, 2 * C, m, tmp, st);
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

// Fused multi-head attention for SHORT sequences (Tq, Tk <= 128) -- Language_Model's causal self-attention (lm_and_am/model/language_model.py:39-52:
// T <= 100 by the position table of util/hparams.py:23) and every other call of that size; the semantics are attention.hip's
// (end2end/transformer.py:89-115,144-151: key mask = zero rows of the per-head K, fill value -2^32+1, optional lower-triangular mask, query
// mask after the softmax, dropout of the weights) and so are the lse pair (reference, log2 row sum) and the dropout counter: the kernels of the
// two files can be mixed freely (forward of one, backward of the other).
//
// Why a second family (round 6): attention.hip gives a wave 32 queries and streams 32-key tiles through a workgroup of four waves.  At
// T = 100 that is ONE workgroup per (sample, head) whose fourth wave (4 live rows) walks four tiles one after the other while the first walks
// one -- the launch runs at the pace of that wave (28.9 us forward for 3.4 GFLOP, mfma_busy 0.22).  Here the unit is 16 rows:
// v_mfma_f32_16x16x4_f32 (same rate as 32x32x2), a wave owns 16 queries (keys in the backward), a workgroup = ceil(T / 16) waves holds the WHOLE
// K and V (Q and dO) head slices of its (sample, head) in LDS -- one DMA burst, one barrier, then no synchronisation at all -- and the causal
// mask skips 16 x 16 blocks: 28 instead of 40 (of 32 x 32 size: 10 x 4) block products at T = 100, the longest wave 7 short steps instead of 4 long.
//
// Layouts (g = lane >> 4, c = lane & 15):
//   D = A.B, 16x16x4: A value of lane = A[row c][k g], B value = B[k g][col c], accumulator register r of lane = D[row 4 g + r][col c].
//   Scores transposed, S^T = K.Q^T (key on the accumulator row, query on the lane column): softmax statistics of a query are lane-local + two
//   lane exchanges (xor 16, 32), and register r of the score tile IS the B operand of the next product's instruction r (contraction index =
//   keys {4 g' + r}), so P never moves.  Instruction t of a score tile contracts d = {16 g' + t}: a lane's 16 operand values of a row are 64
//   contiguous bytes (four ds_read_b128 / four global float4).
//   LDS rows are 256 bytes without padding (LDS-DMA writes 1 KB pieces); the 16-byte chunks of row r are stored at chunk ^ f(r):
//   row-read tiles (ds_read_b128 of one row per lane, 16 rows x 4 chunk groups per instruction) f(r) = ((r & 15) - 4) & 15 -- conflict-free on the
//   instruction's lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, ...; column-read tiles (ds_read_b32 of 16 consecutive floats of rows 4 g + j)
//   f(r) = r & 4 -- the two rows of a 32-lane group land on different bank halves.
#include "asr_common.h"
#include "attn_common.h"

namespace {

typedef __attribute__((address_space(3))) float as_lds_f;

__device__ __forceinline__ int as_fk(int r) { return ((r & 15) - 4) & 15; }     // row-read tiles
__device__ __forceinline__ int as_fv(int r) { return r & 4; }                   // column-read tiles

// all pieces (4 rows x 256 B) of the head slice X[base_row .. base_row + T)[hoff .. hoff + 64) into dst[rows16][64], chunk-swizzled by F on the
// GLOBAL side; rows >= T are sent out of the buffer's range and read zeros.  Wave w of nw issues pieces w, w + nw, ...
template <int F, class R>
__device__ __forceinline__ void as_stage(R rs, float* dst, long base_row, int T, int rows16, int ld, int hoff, int wave, int nw, int lane) {
    const int pr = lane >> 4, pc = lane & 15;
    for (int p = wave; p < rows16 / 4; p += nw) {
        const int r = 4 * p + pr;
        const int ch = pc ^ (F == 0 ? as_fk(r) : as_fv(r));
        unsigned v = (unsigned)((((long)pr) * ld + hoff + ch * 4) * 4);
        if (r >= T) v = 0xFFFFFFF0u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (as_lds_f*)(dst + p * 256), 16, v, (int)((base_row + 4 * p) * ld * 4), 0, 0);
    }
}

// key bias in min() form (+inf: a real key, the fill value: a key-masked one = zero row sum of the head slice, -inf: past the end) for keys
// [0, T16): precomputed (asr_attention_stats) or from K -- sixteen lanes per row, the summation order of attention.hip's at_row_stats
__device__ __forceinline__ void as_key_bias(float* kb, const float* __restrict__ pre, const float* __restrict__ K, long kbase, int Tk, int T16,
                                            int ldk, int hoff, int tid, int nthreads) {
    if (pre) {
        for (int r = tid; r < T16; r += nthreads) kb[r] = r < Tk ? pre[r] : -INFINITY;
        return;
    }
    const int c4 = tid & 15;
    for (int r0 = 0; r0 < T16; r0 += nthreads >> 4) {
        const int r = r0 + (tid >> 4);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < Tk) v = *(const float4*)(K + (kbase + r) * ldk + hoff + c4 * 4);
        float s = v.x + v.y + v.z + v.w;
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if (c4 == 0 && r < T16) kb[r] = (r < Tk) ? (s != 0.f ? INFINITY : FILL2) : -INFINITY;
    }
}

__device__ __forceinline__ float as_xmax(float v) { v = vmax(v, __shfl_xor(v, 16, 64)); return vmax(v, __shfl_xor(v, 32, 64)); }
__device__ __forceinline__ float as_xsum(float v) { v += __shfl_xor(v, 16, 64); return v + __shfl_xor(v, 32, 64); }

// ------------------------------------------------------------------ forward
constexpr int AS_MAXT = 128, AS_MAXKT = AS_MAXT / 16;

template <bool CAUSAL, bool DROP>
__global__ __launch_bounds__(512) void attn_small_fwd_kernel(const float* __restrict__ Q, const float* __restrict__ K, const float* __restrict__ V,
                                                             float* __restrict__ O, float* __restrict__ lse, int Nn, int Tq, int Tk, int C, int H,
                                                             int ldq, int ldk, uint32_t drop_thr, uint32_t drop_seed, float drop_scale,
                                                             const float* __restrict__ kstat, uint32_t wmap) {
#if __HIP_DEVICE_COMPILE__
    extern __shared__ __attribute__((aligned(16))) float as_smem[];
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, c = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), nw = blockDim.x >> 6;
    const int head = blockIdx.x % H, n = blockIdx.x / H;
    const int hoff = head * DH;
    const int nkt = (Tk + 15) >> 4, T16 = nkt * 16;
    float* Ks = as_smem;                       // [T16][64], chunks ^ as_fk(row)
    float* Vs = as_smem + T16 * 64;            // [T16][64], chunks ^ as_fv(row)
    float* kb = Vs + T16 * 64;                 // [T16]
    const long qbase = (long)n * Tq, kbase = (long)n * Tk;
    auto rk = __builtin_amdgcn_make_buffer_rsrc((void*)K, 0, (int)((((long)Nn * Tk - 1) * ldk + C) * 4), 0x00020000);
    auto rv = __builtin_amdgcn_make_buffer_rsrc((void*)V, 0, (int)((((long)Nn * Tk - 1) * ldk + C) * 4), 0x00020000);
    as_stage<0>(rk, Ks, kbase, Tk, T16, ldk, hoff, wave, nw, lane);
    as_stage<1>(rv, Vs, kbase, Tk, T16, ldk, hoff, wave, nw, lane);

    const int q0 = (int)((wmap >> (4 * wave)) & 15u) * 16, q = q0 + c;        // the 16-query block of this wave (as_wave_map)
    float qreg[16];
    float qabs = 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (q < Tq) v = *(const float4*)(Q + (qbase + q) * ldq + hoff + 16 * g + 4 * u);
        qabs += fabsf(v.x) + fabsf(v.y) + fabsf(v.z) + fabsf(v.w);
        qreg[4 * u + 0] = v.x * QSCALE2; qreg[4 * u + 1] = v.y * QSCALE2; qreg[4 * u + 2] = v.z * QSCALE2; qreg[4 * u + 3] = v.w * QSCALE2;
    }
    qabs = as_xsum(qabs);
    const float qmask = (qabs != 0.f) ? 1.f : 0.f;
    as_key_bias(kb, kstat ? kstat + ((long)n * H + head) * Tk : nullptr, K, kbase, Tk, T16, ldk, hoff, tid, blockDim.x);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#if defined(ASR_DEV_HOOKS) && defined(AS_PROLOGUE_ONLY)      // development probe (tools/build_variant.sh): what a launch costs up to its one barrier
    return;
#endif                            // the only barrier: K, V and the key bias are complete

    // scores of the whole row: s[kt][r] = key 16 kt + 4 g + r, query q (base-2 units)
    floatx4 s[AS_MAXKT];
    float m = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < AS_MAXKT; ++kt) {
        if (kt < nkt) {
            floatx4 a = {0.f, 0.f, 0.f, 0.f};
            const bool future = CAUSAL && 16 * kt > q0 + 15;          // every key of the tile is in the future of every query of this wave
            if (!future) {
                const float* krow = Ks + (16 * kt + c) * 64;
                const int f = as_fk(c);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float4 kv = *(const float4*)(krow + (((4 * g + u) ^ f) << 2));
                    a = __builtin_amdgcn_mfma_f32_16x16x4f32(kv.x, qreg[4 * u + 0], a, 0, 0, 0);
                    a = __builtin_amdgcn_mfma_f32_16x16x4f32(kv.y, qreg[4 * u + 1], a, 0, 0, 0);
                    a = __builtin_amdgcn_mfma_f32_16x16x4f32(kv.z, qreg[4 * u + 2], a, 0, 0, 0);
                    a = __builtin_amdgcn_mfma_f32_16x16x4f32(kv.w, qreg[4 * u + 3], a, 0, 0, 0);
                }
            }
            const float4 kbv = *(const float4*)(kb + 16 * kt + 4 * g);
            const float kbr[4] = {kbv.x, kbv.y, kbv.z, kbv.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = a[r];
                if (CAUSAL) v = (!future && 16 * kt + 4 * g + r <= q) ? v : FILL2;
                v = vmin(v, kbr[r]);
                a[r] = v;
                m = vmax(m, v);
            }
            s[kt] = a;
        }
    }
    m = as_xmax(m);
    float l = 0.f;
#pragma unroll
    for (int kt = 0; kt < AS_MAXKT; ++kt)
        if (kt < nkt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { s[kt][r] = ex2(s[kt][r] - m); l += s[kt][r]; }
        }
    l = as_xsum(l);
    if (DROP) {                  // dropout of the attention weights (transformer.py:111): after the row sum, before P.V
        const uint32_t base = (uint32_t)(((n * H + head) * Tq + q) * Tk);
#pragma unroll
        for (int kt = 0; kt < AS_MAXKT; ++kt)
            if (kt < nkt) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    s[kt][r] = drop_keep(base + (uint32_t)(16 * kt + 4 * g + r), drop_seed, drop_thr) ? s[kt][r] * drop_scale : 0.f;
            }
    }
    // a future tile holds exp2(fill - m) = 0 for every row that has met a real score: skipped when that holds for every row of the wave
    // (a row whose visible keys are all masked is uniform over ALL Tk fill entries, future ones included: attention.hip, attn_fwd_tile)
    const bool all_real = CAUSAL && __all((q >= Tq) || (m > -1.0e9f));
    floatx4 oacc[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) oacc[dt] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < AS_MAXKT; ++kt) {
        if (kt < nkt && !(CAUSAL && 16 * kt > q0 + 15 && all_real)) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = 16 * kt + 4 * g + j;
                const float* vrow = Vs + row * 64 + (c & 3);
                const int f = as_fv(row), c4 = c >> 2;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt)
                    oacc[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vrow[((4 * dt + c4) ^ f) << 2], s[kt][j], oacc[dt], 0, 0, 0);
            }
        }
    }
    if (q < Tq) {
        if (g == 0) {
            lse[((long)n * H + head) * Tq + q] = m;
            lse[(long)Nn * H * Tq + ((long)n * H + head) * Tq + q] = log2f(l);
        }
        const float mul = qmask / l;
        float* orow = O + (qbase + q) * C + hoff + 4 * g;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
            *(float4*)(orow + 16 * dt) = make_float4(oacc[dt][0] * mul, oacc[dt][1] * mul, oacc[dt][2] * mul, oacc[dt][3] * mul);
    }
#endif
}


// ------------------------------------------------------------------ backward
// Shared prologue pieces.  Per-query statistics of the whole (sample, head) in LDS: the softmax reference with the query mask folded in (+inf for a
// masked query or one past the end: P = 0, hence no gradient -- what multiplying dO by the mask gave), the log2 row sum, and
// delta[q] = sum_d dO O, formed HERE (sixteen lanes per row, the summation order of attention.hip's attn_delta_kernel: no separate launch).
__device__ __forceinline__ void as_query_stats(float* lse_a, float* lsl_a, float* del_a, const float* __restrict__ qstat, const float* __restrict__ Q,
                                               const float* __restrict__ O, const float* __restrict__ dO, const float* __restrict__ lse, long lrow,
                                               long lsl_off, long qbase, int Tq, int T16, int ldq, int C, int hoff, int tid, int nthreads) {
    const int c4 = tid & 15;
    for (int r0 = 0; r0 < T16; r0 += nthreads >> 4) {
        const int r = r0 + (tid >> 4);
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a, qv = a;
        const bool in = r < Tq;
        if (in) {
            a = *(const float4*)(O + (qbase + r) * C + hoff + c4 * 4);
            b = *(const float4*)(dO + (qbase + r) * C + hoff + c4 * 4);
            if (!qstat) qv = *(const float4*)(Q + (qbase + r) * ldq + hoff + c4 * 4);
        }
        float dl = a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
        float qa = fabsf(qv.x) + fabsf(qv.y) + fabsf(qv.z) + fabsf(qv.w);
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) { dl += __shfl_xor(dl, o, 64); qa += __shfl_xor(qa, o, 64); }
        if (c4 == 0 && r < T16) {
            const bool live = in && (qstat ? qstat[lrow + r] != 0.f : qa != 0.f);
            lse_a[r] = live ? lse[lrow + r] : INFINITY;
            lsl_a[r] = in ? lse[lsl_off + lrow + r] : 0.f;
            del_a[r] = in ? dl : 0.f;
        }
    }
}

// dK, dV: a wave owns 16 keys and ONE of the two gradients, and walks the 16-query tiles of Q / dO, which sit in LDS whole.
// Two roles per key block (round 6, second form): the dV wave recomputes S -> P and accumulates dV += P^T dO (K rows in registers: 32 MFMAs per
// tile), the dK wave recomputes S and dP -> dS and accumulates dK += dS^T Q (K and V rows in registers: 48 MFMAs per tile).  One wave doing
// both (64 MFMAs per tile) needed K, V and two accumulator sets = 148 registers -- one 7-wave workgroup per CU, 1.75 waves per SIMD, its
// LDS round trips exposed (49 us per launch at T = 100); the split pays a second S product (+25 % MFMAs) for twice the waves per CU at under
// 128 registers each.
#if __HIP_DEVICE_COMPILE__
template <bool CAUSAL, bool DROP, int ROLE>          // ROLE 0: dV, 1: dK
__device__ __forceinline__ void as_bwd_kv_body(const float* __restrict__ K, const float* __restrict__ V, float* __restrict__ dOut,
                                               const float* __restrict__ Qs, const float* __restrict__ Ds, const float* __restrict__ lse_a,
                                               const float* __restrict__ lsl_a, const float* __restrict__ del_a, const float* __restrict__ kb,
                                               int n, int head, int H, int Tq, int Tk, int nqt, int ldk, int hoff, long kbase, int k0, int lane,
                                               int relu_grad, uint32_t drop_thr, uint32_t drop_seed, float drop_scale) {
    const int g = lane >> 4, c = lane & 15, key = k0 + c;
    float kreg[16], vreg[ROLE == 1 ? 16 : 1];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
        if (key < Tk) {
            a = *(const float4*)(K + (kbase + key) * ldk + hoff + 16 * g + 4 * u);
            if (ROLE == 1) b = *(const float4*)(V + (kbase + key) * ldk + hoff + 16 * g + 4 * u);
        }
        kreg[4 * u + 0] = a.x * QSCALE2; kreg[4 * u + 1] = a.y * QSCALE2; kreg[4 * u + 2] = a.z * QSCALE2; kreg[4 * u + 3] = a.w * QSCALE2;
        if (ROLE == 1) { vreg[4 * u + 0] = b.x; vreg[4 * u + 1] = b.y; vreg[4 * u + 2] = b.z; vreg[4 * u + 3] = b.w; }
    }
    const bool kkeep = kb[key] > 1.0e38f && key < Tk;      // +inf: a real key  (k0 < Tk, so key < Tk16)
    const float kfill = (key < Tk) ? FILL2 : -INFINITY;
    floatx4 acc[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) acc[dt] = floatx4{0.f, 0.f, 0.f, 0.f};
    const int f = as_fk(c);
    for (int qt = 0; qt < nqt; ++qt) {
        const int q0 = 16 * qt;
        // these 16 queries all precede this wave's 16 keys: dS = 0, and P = exp(fill - max) = 0 unless a row's max IS the fill value
        if (CAUSAL && q0 + 15 < k0 && !__any(lse_a[q0 + c] < -1.0e9f)) continue;
        floatx4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
        const float* qrow = Qs + (q0 + c) * 64;
        const float* drow = Ds + (q0 + c) * 64;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float4 qv = *(const float4*)(qrow + (((4 * g + u) ^ f) << 2));
            s = __builtin_amdgcn_mfma_f32_16x16x4f32(qv.x, kreg[4 * u + 0], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_16x16x4f32(qv.y, kreg[4 * u + 1], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_16x16x4f32(qv.z, kreg[4 * u + 2], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_16x16x4f32(qv.w, kreg[4 * u + 3], s, 0, 0, 0);
            if (ROLE == 1) {
                const float4 dv4 = *(const float4*)(drow + (((4 * g + u) ^ f) << 2));
                dp = __builtin_amdgcn_mfma_f32_16x16x4f32(dv4.x, vreg[4 * u + 0], dp, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_16x16x4f32(dv4.y, vreg[4 * u + 1], dp, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_16x16x4f32(dv4.z, vreg[4 * u + 2], dp, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_16x16x4f32(dv4.w, vreg[4 * u + 3], dp, 0, 0, 0);
            }
        }
        // rows of s / dp = queries q0 + 4 g + r, column = this lane's key;  dV wave: s <- P as dV sees it;  dK wave: s <- dS / 0.125
        const float4 t0 = *(const float4*)(lse_a + q0 + 4 * g), t1 = *(const float4*)(lsl_a + q0 + 4 * g);
        const float lsv[4] = {t0.x, t0.y, t0.z, t0.w}, llv[4] = {t1.x, t1.y, t1.z, t1.w};
        float dlv[4] = {0.f, 0.f, 0.f, 0.f};
        if (ROLE == 1) { const float4 t2 = *(const float4*)(del_a + q0 + 4 * g); dlv[0] = t2.x; dlv[1] = t2.y; dlv[2] = t2.z; dlv[3] = t2.w; }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int qq = q0 + 4 * g + r;
            const bool keep = kkeep && (!CAUSAL || key <= qq);
            const float sv = keep ? s[r] : kfill;
            const float p = ex2((sv - lsv[r]) - llv[r]);
            bool dm = true;
            if (DROP) dm = drop_keep((uint32_t)(((n * H + head) * Tq + qq) * Tk + key), drop_seed, drop_thr);
            if (ROLE == 0) s[r] = DROP ? (dm ? p * drop_scale : 0.f) : p;
            else {
                float dpe = dp[r];
                if (DROP) dpe = dm ? dpe * drop_scale : 0.f;
                s[r] = keep ? p * (dpe - dlv[r]) : 0.f;
            }
        }
        const float* Xs = ROLE == 0 ? Ds : Qs;         // dV += P^T dO (dO by column);  dK += dS^T Q (Q by column)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = q0 + 4 * g + j;
            const int fr = as_fk(row), c4 = c >> 2;
            const float* xc = Xs + row * 64 + (c & 3);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) acc[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(xc[((4 * dt + c4) ^ fr) << 2], s[j], acc[dt], 0, 0, 0);
        }
    }
    if (key < Tk) {
        float* orow = dOut + (kbase + key) * ldk + hoff + 4 * g;
        const float* src = (ROLE == 0 ? V : K) + (kbase + key) * ldk + hoff + 4 * g;
        const float mul = ROLE == 0 ? 1.f : 0.125f;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            float4 a = make_float4(acc[dt][0] * mul, acc[dt][1] * mul, acc[dt][2] * mul, acc[dt][3] * mul);
            if (relu_grad) {        // the gradient of the pre-ReLU projection: masked by (projection > 0)
                const float4 h4 = *(const float4*)(src + 16 * dt);
                a.x = h4.x > 0.f ? a.x : 0.f; a.y = h4.y > 0.f ? a.y : 0.f; a.z = h4.z > 0.f ? a.z : 0.f; a.w = h4.w > 0.f ? a.w : 0.f;
            }
            *(float4*)(orow + 16 * dt) = a;
        }
    }
}
#endif

template <bool CAUSAL, bool DROP>
__global__ __launch_bounds__(1024) void attn_small_bwd_kv_kernel(const float* __restrict__ Q, const float* __restrict__ K, const float* __restrict__ V,
                                                                 const float* __restrict__ O, const float* __restrict__ dO, const float* __restrict__ lse,
                                                                 float* __restrict__ dK, float* __restrict__ dV, int Nn, int Tq, int Tk, int C, int H,
                                                                 int ldq, int ldk, int relu_grad, uint32_t drop_thr, uint32_t drop_seed, float drop_scale,
                                                                 const float* __restrict__ qstat, const float* __restrict__ kstat, unsigned long long wmap) {
#if __HIP_DEVICE_COMPILE__
    extern __shared__ __attribute__((aligned(16))) float as_smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), nw = blockDim.x >> 6;
    const int head = blockIdx.x % H, n = blockIdx.x / H;
    const int hoff = head * DH;
    const int nqt = (Tq + 15) >> 4, Tq16 = nqt * 16, Tk16 = ((Tk + 15) >> 4) * 16;
    float* Qs = as_smem;                       // [Tq16][64], chunks ^ as_fk(row): read by row (S) and by column (dK)
    float* Ds = Qs + Tq16 * 64;                // dO, the same
    float* lse_a = Ds + Tq16 * 64;             // [Tq16] each
    float* lsl_a = lse_a + Tq16;
    float* del_a = lsl_a + Tq16;
    float* kb = del_a + Tq16;                  // [Tk16]
    const long qbase = (long)n * Tq, kbase = (long)n * Tk;
    const long lrow = ((long)n * H + head) * Tq;
    auto rq = __builtin_amdgcn_make_buffer_rsrc((void*)Q, 0, (int)((((long)Nn * Tq - 1) * ldq + C) * 4), 0x00020000);
    auto rd = __builtin_amdgcn_make_buffer_rsrc((void*)dO, 0, (int)((((long)Nn * Tq - 1) * C + C) * 4), 0x00020000);
    as_stage<0>(rq, Qs, qbase, Tq, Tq16, ldq, hoff, wave, nw, lane);
    as_stage<0>(rd, Ds, qbase, Tq, Tq16, C, hoff, wave, nw, lane);
    as_query_stats(lse_a, lsl_a, del_a, qstat ? qstat : nullptr, Q, O, dO, lse, lrow, (long)Nn * H * Tq, qbase, Tq, Tq16, ldq, C, hoff, tid, blockDim.x);
    as_key_bias(kb, kstat ? kstat + ((long)n * H + head) * Tk : nullptr, K, kbase, Tk, Tk16, ldk, hoff, tid, blockDim.x);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#if defined(ASR_DEV_HOOKS) && defined(AS_PROLOGUE_ONLY)      // development probe (tools/build_variant.sh): what a launch costs up to its one barrier
    return;
#endif
    // this wave's unit (as_wave_map over 2 x key blocks units, one per wave): nibble = 16-key block | role << 3 (0 dV, 1 dK)
    const int unit = (int)((wmap >> (4 * wave)) & 15ull);
    const int k0 = (unit & 7) * 16, role = unit >> 3;
    if (k0 >= Tk) return;
    if (role == 0)
        as_bwd_kv_body<CAUSAL, DROP, 0>(K, V, dV, Qs, Ds, lse_a, lsl_a, del_a, kb, n, head, H, Tq, Tk, nqt, ldk, hoff, kbase, k0, lane, relu_grad,
                                        drop_thr, drop_seed, drop_scale);
    else
        as_bwd_kv_body<CAUSAL, DROP, 1>(K, V, dK, Qs, Ds, lse_a, lsl_a, del_a, kb, n, head, H, Tq, Tk, nqt, ldk, hoff, kbase, k0, lane, relu_grad,
                                        drop_thr, drop_seed, drop_scale);
#endif
}

// dQ: a wave owns 16 queries (Q, dO rows in registers) and walks the 16-key tiles of K / V, which sit in LDS whole.
template <bool CAUSAL, bool DROP>
__global__ __launch_bounds__(512, 4) void attn_small_bwd_q_kernel(const float* __restrict__ Q, const float* __restrict__ K, const float* __restrict__ V,
                                                               const float* __restrict__ O, const float* __restrict__ dO, const float* __restrict__ lse,
                                                               float* __restrict__ dQ, int Nn, int Tq, int Tk, int C, int H, int ldq, int ldk,
                                                               int relu_grad, uint32_t drop_thr, uint32_t drop_seed, float drop_scale,
                                                               const float* __restrict__ qstat, const float* __restrict__ kstat, uint32_t wmap) {
#if __HIP_DEVICE_COMPILE__
    extern __shared__ __attribute__((aligned(16))) float as_smem[];
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, c = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), nw = blockDim.x >> 6;
    const int head = blockIdx.x % H, n = blockIdx.x / H;
    const int hoff = head * DH;
    const int nkt = (Tk + 15) >> 4, Tk16 = nkt * 16, Tq16 = ((Tq + 15) >> 4) * 16;
    float* Ks = as_smem;                       // [Tk16][64], chunks ^ as_fk(row): read by row (S) and by column (dQ)
    float* Vs = Ks + Tk16 * 64;                // read by row (dP)
    float* lse_a = Vs + Tk16 * 64;             // [Tq16] each
    float* lsl_a = lse_a + Tq16;
    float* del_a = lsl_a + Tq16;
    float* kb = del_a + Tq16;                  // [Tk16]
    const long qbase = (long)n * Tq, kbase = (long)n * Tk;
    const long lrow = ((long)n * H + head) * Tq;
    auto rk = __builtin_amdgcn_make_buffer_rsrc((void*)K, 0, (int)((((long)Nn * Tk - 1) * ldk + C) * 4), 0x00020000);
    auto rv = __builtin_amdgcn_make_buffer_rsrc((void*)V, 0, (int)((((long)Nn * Tk - 1) * ldk + C) * 4), 0x00020000);
    as_stage<0>(rk, Ks, kbase, Tk, Tk16, ldk, hoff, wave, nw, lane);
    as_stage<0>(rv, Vs, kbase, Tk, Tk16, ldk, hoff, wave, nw, lane);

    const int q0 = (int)((wmap >> (4 * wave)) & 15u) * 16, q = q0 + c;
    float qreg[16], doreg[16];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
        if (q < Tq) {
            a = *(const float4*)(Q + (qbase + q) * ldq + hoff + 16 * g + 4 * u);
            b = *(const float4*)(dO + (qbase + q) * C + hoff + 16 * g + 4 * u);
        }
        qreg[4 * u + 0] = a.x * QSCALE2; qreg[4 * u + 1] = a.y * QSCALE2; qreg[4 * u + 2] = a.z * QSCALE2; qreg[4 * u + 3] = a.w * QSCALE2;
        doreg[4 * u + 0] = b.x; doreg[4 * u + 1] = b.y; doreg[4 * u + 2] = b.z; doreg[4 * u + 3] = b.w;
    }
    as_query_stats(lse_a, lsl_a, del_a, qstat ? qstat : nullptr, Q, O, dO, lse, lrow, (long)Nn * H * Tq, qbase, Tq, Tq16, ldq, C, hoff, tid, blockDim.x);
    as_key_bias(kb, kstat ? kstat + ((long)n * H + head) * Tk : nullptr, K, kbase, Tk, Tk16, ldk, hoff, tid, blockDim.x);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#if defined(ASR_DEV_HOOKS) && defined(AS_PROLOGUE_ONLY)      // development probe (tools/build_variant.sh): what a launch costs up to its one barrier
    return;
#endif
    if (q0 >= Tq) return;
    // (a masked query: reference +inf -> P = 0 -> dS = 0: the mask on dO of attention.hip's dQ kernel, folded into the statistic)
    const float my_lse = lse_a[q0 + c], my_lsl = lsl_a[q0 + c], my_del = del_a[q0 + c];

    floatx4 dq[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) dq[dt] = floatx4{0.f, 0.f, 0.f, 0.f};
    const int f = as_fk(c);
    int nlive = nkt;
    if (CAUSAL) { const int lim = (q0 + 15) / 16 + 1; if (lim < nlive) nlive = lim; }      // masked scores get no gradient
    for (int kt = 0; kt < nlive; ++kt) {
        floatx4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
        const float* krow = Ks + (16 * kt + c) * 64;
        const float* vrow = Vs + (16 * kt + c) * 64;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float4 kv = *(const float4*)(krow + (((4 * g + u) ^ f) << 2));
            const float4 vv = *(const float4*)(vrow + (((4 * g + u) ^ f) << 2));
            s = __builtin_amdgcn_mfma_f32_16x16x4f32(kv.x, qreg[4 * u + 0], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_16x16x4f32(kv.y, qreg[4 * u + 1], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_16x16x4f32(kv.z, qreg[4 * u + 2], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_16x16x4f32(kv.w, qreg[4 * u + 3], s, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_16x16x4f32(vv.x, doreg[4 * u + 0], dp, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_16x16x4f32(vv.y, doreg[4 * u + 1], dp, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_16x16x4f32(vv.z, doreg[4 * u + 2], dp, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_16x16x4f32(vv.w, doreg[4 * u + 3], dp, 0, 0, 0);
        }
        const float4 kbv = *(const float4*)(kb + 16 * kt + 4 * g);
        const float kbr[4] = {kbv.x, kbv.y, kbv.z, kbv.w};
        // rows of s / dp = keys 16 kt + 4 g + r, column = this lane's query;  dp <- dS / 0.125
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int kk = 16 * kt + 4 * g + r;
            const bool keep = (kbr[r] > 1.0e38f) && (!CAUSAL || kk <= q);
            float sv = s[r];
            if (CAUSAL) sv = (kk <= q) ? sv : FILL2;
            sv = vmin(sv, kbr[r]);
            const float p = ex2((sv - my_lse) - my_lsl);
            float dpe = dp[r];
            if (DROP) dpe = drop_keep((uint32_t)(((n * H + head) * Tq + q) * Tk + kk), drop_seed, drop_thr) ? dpe * drop_scale : 0.f;
            dp[r] = keep ? p * (dpe - my_del) : 0.f;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = 16 * kt + 4 * g + j;
            const int fr = as_fk(row), c4 = c >> 2;
            const float* kc = Ks + row * 64 + (c & 3);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
                dq[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kc[((4 * dt + c4) ^ fr) << 2], dp[j], dq[dt], 0, 0, 0);
        }
    }
    if (q < Tq) {
        float* qr = dQ + (qbase + q) * ldq + hoff + 4 * g;
        const float* qs = Q + (qbase + q) * ldq + hoff + 4 * g;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            float4 a = make_float4(dq[dt][0] * 0.125f, dq[dt][1] * 0.125f, dq[dt][2] * 0.125f, dq[dt][3] * 0.125f);
            if (relu_grad) {
                const float4 hq = *(const float4*)(qs + 16 * dt);
                a.x = hq.x > 0.f ? a.x : 0.f; a.y = hq.y > 0.f ? a.y : 0.f; a.z = hq.z > 0.f ? a.z : 0.f; a.w = hq.w > 0.f ? a.w : 0.f;
            }
            *(float4*)(qr + 16 * dt) = a;
        }
    }
#endif
}

}  // namespace

// Which unit of work a wave owns, one nibble per wave (wave w: bits 4 w .. 4 w + 3): a 16-row block for the forward and dQ kernels, block | role << 3
// for the dK / dV kernel.  A workgroup's waves go to the CU's four SIMDs cyclically (MI355X_MICROARCH.md, LDS section), so waves w and w + 4 share
// a SIMD; under the causal mask block b walks cost[b] tiles, and with wave w on block w the longest waves met on one SIMD (7 + 3 steps against 4 at
// T = 100).  Longest-processing-time-first over the four wave classes: at T = 100 (costs 1 .. 7) every SIMD gets 7 steps.
static unsigned long long as_wave_map(int nw, int nlive, const int* cost, const int* ids = nullptr) {
    // nlive units (unit i has id ids[i], default i, and cost cost[i]) over nw <= 16 waves; waves without a unit get id 15
    int load[4] = {0, 0, 0, 0}, used[4] = {0, 0, 0, 0}, slots[4] = {0, 0, 0, 0}, order[16], id_of[16];
    for (int w = 0; w < nw; ++w) { ++slots[w & 3]; id_of[w] = 15; }
    for (int b = 0; b < nlive; ++b) order[b] = b;
    for (int i = 1; i < nlive; ++i)              // insertion sort, descending cost (stable: ties keep the lower unit first)
        for (int j = i; j > 0 && cost[order[j]] > cost[order[j - 1]]; --j) { const int t = order[j]; order[j] = order[j - 1]; order[j - 1] = t; }
    for (int i = 0; i < nlive; ++i) {
        int best = -1;
        for (int k = 0; k < 4; ++k)
            if (used[k] < slots[k] && (best < 0 || load[k] < load[best])) best = k;
        id_of[best + 4 * used[best]] = ids ? ids[order[i]] : order[i];
        load[best] += cost[order[i]]; ++used[best];
    }
    unsigned long long m = 0;
    for (int w = 0; w < 16; ++w) m |= (unsigned long long)(w < nw ? id_of[w] : 15) << (4 * w);
    return m;
}

// 1 when the short-sequence kernels take the call (attention.hip's entry points ask)
int asr_attention_small_takes(int N, int Tq, int Tk, int C, int ldq, int ldk) {
    // (a function of the sequence lengths; the byte bounds are those of the 32-bit buffer offsets)
    return Tq >= 1 && Tk >= 1 && Tq <= AS_MAXT && Tk <= AS_MAXT && (long)N * Tk * ldk * 4 < (1L << 31) && (long)N * Tq * ldq * 4 < (1L << 31) &&
           (long)N * Tq * C * 4 < (1L << 31);
}

int asr_attention_small_fwd_launch(const float* Q, const float* K, const float* V, int N, int Tq, int Tk, int C, int H, int ldq, int ldk, int causal,
                                   int drop, uint32_t thr, uint32_t seed, float scale, float* O, float* lse, const float* kstat, void* stream) {
    const int nkt = asr_cdiv(Tk, 16);
    const size_t lds = (size_t)(2 * nkt * 16 * 64 + nkt * 16) * sizeof(float);
    const int nqt = asr_cdiv(Tq, 16);
    dim3 grid(N * H), block(64 * nqt);
    int cost[8];
    for (int b = 0; b < nqt; ++b) cost[b] = causal ? (b + 1 < nkt ? b + 1 : nkt) : nkt;          // key tiles block b walks
    const uint32_t wmap = (uint32_t)as_wave_map(nqt, nqt, cost);
    hipStream_t st = (hipStream_t)stream;
#define ASR_AS_FWD(CA, DR)                                                                                                      \
    do {                                                                                                                       \
        auto kern = attn_small_fwd_kernel<CA, DR>;                                                                             \
        static size_t have = 0;                                                                                                \
        if (lds > have) { if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) { (void)hipGetLastError(); return ASR_ERR_UNSUPPORTED; } have = lds; } \
        hipLaunchKernelGGL(kern, grid, block, lds, st, Q, K, V, O, lse, N, Tq, Tk, C, H, ldq, ldk, thr, seed, scale, kstat, wmap); \
    } while (0)
    if (drop) { if (causal) ASR_AS_FWD(true, true); else ASR_AS_FWD(false, true); }
    else { if (causal) ASR_AS_FWD(true, false); else ASR_AS_FWD(false, false); }
#undef ASR_AS_FWD
    ASR_CHECK_LAUNCH("attention_small_fwd");
    return ASR_OK;
}

int asr_attention_small_bwd_launch(const float* Q, const float* K, const float* V, const float* O, const float* dO, const float* lse, int N, int Tq, int Tk,
                                   int C, int H, int ldq, int ldk, int causal, int relu_grad, int drop, uint32_t thr, uint32_t seed, float scale,
                                   float* dQ, float* dK, float* dV, const float* qstat, const float* kstat, void* stream) {
    const int nqt = asr_cdiv(Tq, 16), nkt = asr_cdiv(Tk, 16);
    const size_t ldskv = (size_t)(2 * nqt * 16 * 64 + 3 * nqt * 16 + nkt * 16) * sizeof(float);
    const size_t ldsq = (size_t)(2 * nkt * 16 * 64 + 3 * nqt * 16 + nkt * 16) * sizeof(float);
    // dK / dV: two waves per 16-key block (roles dV: 32 MFMAs per tile, dK: 48); dQ: one wave per 16-query block
    dim3 grid(N * H), blockkv(64 * 2 * nkt), blockq(64 * nqt);
    int ckv[16], idkv[16], cq[8];
    for (int b = 0; b < nkt; ++b) {
        const int tiles = causal ? (nqt - b > 0 ? nqt - b : 0) : nqt;          // query tiles key block b walks
        ckv[2 * b] = 32 * tiles + 8; idkv[2 * b] = b;                          // dV wave
        ckv[2 * b + 1] = 48 * tiles + 8; idkv[2 * b + 1] = b | 8;              // dK wave
    }
    for (int b = 0; b < nqt; ++b) cq[b] = causal ? (b + 1 < nkt ? b + 1 : nkt) : nkt;
    const unsigned long long mapkv = as_wave_map(2 * nkt, 2 * nkt, ckv, idkv);
    const uint32_t mapq = (uint32_t)as_wave_map(nqt, nqt, cq);
    hipStream_t st = (hipStream_t)stream;
#define ASR_AS_BWD(CA, DR)                                                                                                     \
    do {                                                                                                                       \
        auto kkv = attn_small_bwd_kv_kernel<CA, DR>;                                                                           \
        static size_t havekv = 0;                                                                                              \
        if (ldskv > havekv) { if (hipFuncSetAttribute((const void*)kkv, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldskv) != hipSuccess) { (void)hipGetLastError(); return ASR_ERR_UNSUPPORTED; } havekv = ldskv; } \
        hipLaunchKernelGGL(kkv, grid, blockkv, ldskv, st, Q, K, V, O, dO, lse, dK, dV, N, Tq, Tk, C, H, ldq, ldk, relu_grad, thr, seed, scale, qstat, kstat, mapkv); \
        auto kq = attn_small_bwd_q_kernel<CA, DR>;                                                                             \
        static size_t haveq = 0;                                                                                               \
        if (ldsq > haveq) { if (hipFuncSetAttribute((const void*)kq, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsq) != hipSuccess) { (void)hipGetLastError(); return ASR_ERR_UNSUPPORTED; } haveq = ldsq; } \
        hipLaunchKernelGGL(kq, grid, blockq, ldsq, st, Q, K, V, O, dO, lse, dQ, N, Tq, Tk, C, H, ldq, ldk, relu_grad, thr, seed, scale, qstat, kstat, mapq); \
    } while (0)
    if (drop) { if (causal) ASR_AS_BWD(true, true); else ASR_AS_BWD(false, true); }
    else { if (causal) ASR_AS_BWD(true, false); else ASR_AS_BWD(false, false); }
#undef ASR_AS_BWD
    ASR_CHECK_LAUNCH("attention_small_bwd");
    return ASR_OK;
}

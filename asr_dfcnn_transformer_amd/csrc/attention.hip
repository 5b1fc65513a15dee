// Fused multi-head attention forward / backward on the fp32 MFMA pipe with the reference's mask semantics
// (end2end/transformer.py:89-115,144-151): key mask = zero rows of the per-head K, fill value -2^32+1, optional
// lower-triangular mask, query mask applied after the softmax; scores never touch HBM.
//
// Layout trick: the score tile is computed TRANSPOSED, S^T = K.Q^T (keys in registers, query on the lane), so the row max /
// row sum of the online softmax are lane-local (+1 exchange between the two half-waves), and the P tile sits in exactly the
// register layout the next MFMA needs as its B operand (contraction index = register index): P never moves between lanes
// or through LDS.
#include "asr_common.h"
#include "reduce.h"
#include "attn_common.h"

namespace {

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

// ------------------------------------------------------------------ LDS-DMA of the streamed tiles (round 3)
// The K / V (forward, dQ) and Q / dO (dK, dV) tiles are 64 rows x 64 floats of a head slice.  They arrive by buffer-form
// LDS-DMA (buffer_load_dwordx4 ... lds: resource and row offset scalar, per-lane offset constant -- no vector instruction per
// piece, no staging registers, no ds_write) into TWO tile sets; the pieces of tile j + 1 are issued at the top of tile j and land
// while it is computed: one barrier per tile (the register-staged form paid two barriers and an exposed global round trip per
// tile).  The forward streams 32-key tiles (32 KB of LDS for the two sets: three workgroups per CU; 64-key tiles at two per CU
// are 3 % slower without and 11 % slower with the causal mask, whose skipping then works on 64-key steps); measured on
// B 64 x T 512 x 8 heads: 0.371 -> 0.350 ms, causal 0.230 -> 0.207 ms.  Ablation of the 64-key form (0.367 ms): without the DMA
// 0.341, without the barrier 0.331, with neither 0.320 -- the rest is the MFMA + softmax core at two or three waves per SIMD.
// Pieces issued between the MFMA groups instead of in a block at the tile start were 2.5 % SLOWER here (unlike gemm1.hip).
// A piece = 4 rows x 256 bytes; wave w issues pieces w, w + 4, ... of a tile.
// The DMA writes 1 KB contiguously, so rows have no padding; tiles that are read with ds_read_b128 (32 consecutive rows at one
// logical 16-byte chunk) store chunk c of row r at position c ^ f(r), f(r) = (r & 3) | ((r >> 3) & 3) << 2 -- a bijection on
// each of the instruction's 16-lane service groups {0-3,12-15,20-27}, {4-11,16-19,28-31}: conflict-free -- and independent of
// bit 2 of r, so that for the OTHER read pattern (ds_read_b32 of row rowidx(reg, half): 32 consecutive floats of one row) f is
// the compile-time register index.  Rows past the end of a sequence are sent out of the buffer's range and read zeros.
typedef __attribute__((address_space(3))) float at_lds_f;
typedef __attribute__((address_space(3))) char at_lds_c;

// The barrier that publishes a DMA'd tile: every wave first waits for ITS OWN pieces, explicitly -- hipcc does not add the
// vmcnt(0) to this __syncthreads() by itself here (its alias analysis decides the LDS-DMA cannot matter) --, then the barrier
// makes all pieces visible to all waves.
__device__ __forceinline__ void at_dma_barrier() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}

// dynamic LDS rounded up to 128 bytes (the launchers allocate 128 bytes more)
__device__ __forceinline__ float* at_lds_align(float* p) {
    const unsigned a = (unsigned)(uintptr_t)(at_lds_c*)p;
    return p + (((128u - (a & 127u)) & 127u) >> 2);
}

__device__ __forceinline__ int at_swz(int row) { return (row & 3) | (((row >> 3) & 3) << 2); }

// per-lane byte offset of piece index i (pieces wave + 4 i) of a tile whose rows have pitch ld floats; SWZ: see above
template <bool SWZ>
__device__ __forceinline__ unsigned at_voff(int i, int wave, int lane, int ld, int hoff) {
    const int r = 4 * (wave + 4 * i) + (lane >> 4);
    const int c = SWZ ? ((lane & 15) ^ at_swz(r)) : (lane & 15);
    return (unsigned)((((long)(lane >> 4)) * ld + hoff + c * 4) * 4);
}

#if __HIP_DEVICE_COMPILE__
// the 4 pieces this wave owns of the tile that starts at absolute row `row0` (`valid` rows of it exist)
template <bool SWZ, int NPW = 4, class R>
__device__ __forceinline__ void at_tile_dma(R rs, float* __restrict__ dst, int wave, int lane, const unsigned (&vo)[4], long row0,
                                            int valid, int ld) {
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
        const int p = wave + 4 * i;
        unsigned v = vo[SWZ ? i : 0];
        if (valid < 16 * NPW && 4 * p + (lane >> 4) >= valid) v = 0xFFFFFFF0u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (at_lds_f*)(dst + p * 256), 16, v, (int)((row0 + 4 * p) * ld * 4), 0, 0);
    }
}
#endif

// per-row statistic of a head slice X[n][t][hoff .. hoff + 63] for every t < T, into LDS: MODE 0 the key bias in min() form
// (+inf for a real key, the fill value for a key-masked one = zero row sum, -inf past the end: score = min(score, bias)),
// MODE 1 the query mask (1 / 0: sum of magnitudes != 0).  16 lanes per row, the summation order of the register-staged form.
template <int MODE>
__device__ __forceinline__ void at_row_stats(float* stat, const float* __restrict__ X, long base_row, int T, int T64, int ld,
                                             int hoff, int tid) {
    const int c4 = tid & 15;
    for (int r0 = 0; r0 < T64; r0 += 64) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int r = r0 + u * 16 + (tid >> 4);
            v[u] = (r < T) ? *(const float4*)(X + (base_row + r) * ld + hoff + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int r = r0 + u * 16 + (tid >> 4);
            float s = MODE == 0 ? (v[u].x + v[u].y + v[u].z + v[u].w) : (fabsf(v[u].x) + fabsf(v[u].y) + fabsf(v[u].z) + fabsf(v[u].w));
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
            if (c4 == 0 && r < T64) {
                if (MODE == 0) stat[r] = (r < T) ? (s != 0.f ? INFINITY : FILL2) : -INFINITY;
                else stat[r] = (r < T && s != 0.f) ? 1.f : 0.f;
            }
        }
    }
}

// The same two statistics for EVERY (sample, head, row) in one pass (asr_attention_stats, round 5): the kernels above recompute them per
// workgroup -- 4 to 16 workgroups of a (sample, head) each read its whole K (or Q) slice again, 6-12 % of their time
// (profiles/r05_attention_row_stats_probe.txt) -- or take them from here.  blockIdx.y = 0: query mask of Q, 1: key bias of K; sixteen
// lanes per (row, head) slice, the reduction order of at_row_stats: the same values.
__global__ __launch_bounds__(256) void attn_stats_kernel(const float* __restrict__ Q, const float* __restrict__ K, int N, int Tq, int Tk,
                                                         int H, int ldq, int ldk, float* __restrict__ qm, float* __restrict__ kb) {
    const bool keys = blockIdx.y == 1;
    const float* X = keys ? K : Q;
    const int T = keys ? Tk : Tq, ld = keys ? ldk : ldq;
    const long rows = (long)N * T;
    // a sixteen-lane group takes the head slice of FOUR consecutive rows: four independent 16-byte loads per lane in flight
    const long g = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const int c4 = threadIdx.x & 15;
    const int head = (int)(g % H);
    const long row0 = (g / H) * 4;
    float4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
        v[u] = (row0 + u < rows) ? *(const float4*)(X + (row0 + u) * ld + head * DH + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    float* out = keys ? kb : qm;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        float s = keys ? (v[u].x + v[u].y + v[u].z + v[u].w) : (fabsf(v[u].x) + fabsf(v[u].y) + fabsf(v[u].z) + fabsf(v[u].w));
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        const long row = row0 + u;
        if (row < rows && c4 == 0) {
            const int n = (int)(row / T), t = (int)(row - (long)n * T);
            out[((long)n * H + head) * T + t] = keys ? (s != 0.f ? INFINITY : FILL2) : (s != 0.f ? 1.f : 0.f);
        }
    }
}

// a (sample, head)'s precomputed statistics into LDS, padded like at_row_stats pads (MODE 0: -inf past the end, MODE 1: 0)
template <int MODE>
__device__ __forceinline__ void at_row_stats_load(float* stat, const float* __restrict__ pre, int T, int T64, int tid) {
    for (int r = tid; r < T64; r += 256) stat[r] = r < T ? pre[r] : (MODE == 0 ? -INFINITY : 0.f);
}

// ------------------------------------------------------------------ attention forward
#if __HIP_DEVICE_COMPILE__
// one 64-key tile out of (Kc, Vc); first the DMA of the next tile into (Kn, Vn).  __restrict__ parameters of an inlined
// function on purpose: without the alias scopes hipcc orders every LDS read behind the DMA in flight.
template <bool CAUSAL, bool DROP, int TK, class R>
__device__ __forceinline__ void attn_fwd_tile(const float* __restrict__ Kc, const float* __restrict__ Vc, float* __restrict__ Kn,
                                              float* __restrict__ Vn, const float* __restrict__ kb, R rk, R rv, bool more,
                                              const unsigned (&vok)[4], const unsigned (&vov)[4], long krow_next, int valid_next,
                                              int ldk, int wave, int lane, int k0, int Tk, int q0, int q, int Tq,
                                              const float (&qreg)[32], floatx16 (&oacc)[2], float& m_run, float& l_run,
                                              uint32_t drop_base, uint32_t drop_thr, uint32_t drop_seed, float drop_scale) {
    const int li = lane & 31, lh = lane >> 5;
    if (more) {
        at_tile_dma<true, TK / 16>(rk, Kn, wave, lane, vok, krow_next, valid_next, ldk);
        at_tile_dma<false, TK / 16>(rv, Vn, wave, lane, vov, krow_next, valid_next, ldk);
    }
    const int fk = at_swz(li);
#pragma unroll
    for (int sub = 0; sub < TK / 32; ++sub) {
        if (k0 + sub * 32 >= Tk) continue;               // these 32 keys do not exist
        // 32 keys that are future to all 32 queries of this wave hold only the fill value: exp(fill - max) = 0 for a row that has
        // met a real score -- skipped when that holds for EVERY row of the wave (a row whose keys so far were all key-masked must
        // still see them: TF's softmax is uniform over all fill entries, future ones included)
        if (CAUSAL && k0 + sub * 32 > q0 + 31 && __all((q >= Tq) || (m_run > -1.0e9f))) continue;
        floatx16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const float4 kv = *(const float4*)(Kc + (sub * 32 + li) * 64 + (((2 * g + lh) ^ fk) << 2));
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(kv.x, qreg[g * 4 + 0], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(kv.y, qreg[g * 4 + 1], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(kv.z, qreg[g * 4 + 2], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(kv.w, qreg[g * 4 + 3], s, 0, 0, 0);
        }
        // key bias of this half-wave's 16 keys: rows rowidx(r, lh) = {0-3, 8-11, 16-19, 24-27} + 4 lh
        float kbv[16];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float4 t = *(const float4*)(kb + k0 + sub * 32 + 8 * j + 4 * lh);
            kbv[4 * j + 0] = t.x; kbv[4 * j + 1] = t.y; kbv[4 * j + 2] = t.z; kbv[4 * j + 3] = t.w;
        }
        float mt = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float v = s[r];
            if (CAUSAL) v = (k0 + sub * 32 + rowidx(r, lh) <= q) ? v : FILL2;
            s[r] = vmin(v, kbv[r]);                      // +inf: a real key; the fill value: key-masked; -inf: past the end
            if (r & 1) mt = vmax3(mt, s[r - 1], s[r]);
        }
        mt = vmax(mt, __shfl_xor(mt, 32, 64));
        // Online softmax with a LAZY reference: the running reference m_run only moves when a score exceeds it by more than 2^8
        // (probabilities then stay <= 256: exact in fp32 range), so the rescaling of the 32 output accumulators -- vector
        // instructions that add to the MFMA time on this pipe -- runs for a few tiles per row instead of for every tile.  Any
        // consistent (reference, log2 row-sum) pair describes the same softmax; the backward only uses their sum.
        const bool grow = mt > m_run + 8.0f;
        if (__any(grow)) {
            const float m_new = grow ? mt : m_run;
            const float alpha = grow ? ex2(m_run - m_new) : 1.0f;
            l_run *= alpha;
#pragma unroll
            for (int r = 0; r < 16; ++r) { oacc[0][r] *= alpha; oacc[1][r] *= alpha; }
            m_run = m_new;
        }
        float lt = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = ex2(s[r] - m_run); lt += s[r]; }
        lt += __shfl_xor(lt, 32, 64);
        l_run += lt;
        if (DROP) {            // dropout of the attention weights (transformer.py:111): after the row sum, before P.V
            const uint32_t base = drop_base + (uint32_t)(k0 + sub * 32);
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = drop_keep(base + rowidx(r, lh), drop_seed, drop_thr) ? s[r] * drop_scale : 0.f;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float* vrow = Vc + (sub * 32 + rowidx(r, lh)) * 64 + li;
            oacc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(vrow[0], s[r], oacc[0], 0, 0, 0);
            oacc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(vrow[32], s[r], oacc[1], 0, 0, 0);
        }
    }
}
#endif

template <bool CAUSAL, bool DROP, int TK>
__global__ __launch_bounds__(256, TK == 64 ? 2 : 3) void attn_fwd_kernel(const float* __restrict__ Q, const float* __restrict__ K,
                                                       const float* __restrict__ V, float* __restrict__ O,
                                                       float* __restrict__ lse, int Tq, int Tk, int C, int H, int ldq, int ldk,
                                                       uint32_t drop_thr, uint32_t drop_seed, float drop_scale,
                                                       const float* __restrict__ kstat) {
#if __HIP_DEVICE_COMPILE__
    extern __shared__ __attribute__((aligned(16))) float at_smem[];
    constexpr int TF = TK * 64;                          // floats of one tile
    float* tiles = at_smem;                              // [2 sets][K TK x 64 | V TK x 64], later the epilogue's transpose scratch
    float* kb = at_smem + (4 * TF > 4 * 32 * 65 ? 4 * TF : 4 * 32 * 65);      // [Tk rounded up to 64] key bias
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, lh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int qtile, head, n, Nn;
    attn_block_coords<CAUSAL>(H, true, qtile, head, n, Nn);
    const int hoff = head * DH;
    const int q0 = qtile * 128 + wave * 32, q = q0 + li;
    const long qbase = (long)n * Tq, kbase = (long)n * Tk;
    const int ntiles = (Tk + TK - 1) / TK;

    auto rk = __builtin_amdgcn_make_buffer_rsrc((void*)K, 0, (int)((((long)Nn * Tk - 1) * ldk + C) * 4), 0x00020000);
    auto rv = __builtin_amdgcn_make_buffer_rsrc((void*)V, 0, (int)((((long)Nn * Tk - 1) * ldk + C) * 4), 0x00020000);
    unsigned vok[4], vov[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { vok[i] = at_voff<true>(i, wave, lane, ldk, hoff); vov[i] = at_voff<false>(i, wave, lane, ldk, hoff); }
    at_tile_dma<true, TK / 16>(rk, tiles, wave, lane, vok, kbase, Tk, ldk);
    at_tile_dma<false, TK / 16>(rv, tiles + TF, wave, lane, vov, kbase, Tk, ldk);

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
    if (kstat) at_row_stats_load<0>(kb, kstat + ((long)n * H + head) * Tk, Tk, (Tk + 63) / 64 * 64, tid);
    else at_row_stats<0>(kb, K, kbase, Tk, (Tk + 63) / 64 * 64, ldk, hoff, tid);

    floatx16 oacc[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { oacc[0][r] = 0.f; oacc[1][r] = 0.f; }
    float m_run = -INFINITY, l_run = 0.f;
    const uint32_t drop_base = (uint32_t)(((n * H + head) * Tq + q) * Tk);

    const int qlast_blk = qtile * 128 + 127;
    for (int j = 0; j < ntiles; ++j) {
        const int k0 = j * TK;
        // A tile that lies entirely in the future of every query of this workgroup holds only the fill value: it may be skipped
        // only when EVERY row has met a real score (see attn_fwd_tile).  (The tile's DMA is in flight: harmless, the final
        // barrier waits for it.)
        if (CAUSAL && k0 > qlast_blk) {
            if (__syncthreads_and((q >= Tq) || (m_run > -1.0e9f))) break;
        }
        at_dma_barrier();                                // tile j has landed, nobody reads the other set any more
        float* cur = tiles + (j & 1) * 2 * TF;
        float* nxt = tiles + ((j + 1) & 1) * 2 * TF;
        const int vnext = Tk - (k0 + TK);
        attn_fwd_tile<CAUSAL, DROP, TK>(cur, cur + TF, nxt, nxt + TF, kb, rk, rv, j + 1 < ntiles, vok, vov, kbase + k0 + TK,
                                        vnext < TK ? vnext : TK, ldk, wave, lane, k0, Tk, q0, q, Tq, qreg, oacc, m_run, l_run,
                                        drop_base, drop_thr, drop_seed, drop_scale);
    }
    at_dma_barrier();
    if (q < Tq && lh == 0) {      // kept as (reference, log-sum) pair, both in base-2 units: the reference may be the fill, which would swallow log(l)
        lse[((long)n * H + head) * Tq + q] = m_run;
        lse[(long)Nn * H * Tq + ((long)n * H + head) * Tq + q] = log2f(l_run);
    }
    store_tile_T(O, tiles + wave * (32 * 65), oacc, qmask / l_run, qbase, q0, Tq, C, hoff, lane);
#endif
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
// One wave owns 32 keys (K, V in registers); the block walks the queries in tiles of 32.  Round 3: the Q and dO tiles arrive like
// the forward's K / V tiles (buffer-form LDS-DMA, two tile sets, one barrier per tile); both are read as MFMA A operands by row
// (ds_read_b128, swizzled chunks) AND transposed (dV += dO^T.P, dK += Q^T.dS: ds_read_b32 with the lane's LDS address XOR
// ((r & 7) << 4), see attn_bwd_q_tile).  The per-query statistics of a whole (sample, head) -- row reference, log2 row sum,
// delta -- sit in LDS once; the query mask (zero rows of Q) is folded into the reference (+inf: P = 0, hence dS = 0 and no dV --
// what multiplying dO by the mask gave).
#if __HIP_DEVICE_COMPILE__
template <bool CAUSAL, bool DROP, class R>
__device__ __forceinline__ void attn_bwd_kv_tile(const float* __restrict__ Qc, const float* __restrict__ Dc, float* __restrict__ Qn,
                                                 float* __restrict__ Dn, const float* __restrict__ lse_a, const float* __restrict__ lsl_a,
                                                 const float* __restrict__ del_a, R rq, R rd, bool more, const unsigned (&voq)[4],
                                                 const unsigned (&vod)[4], long qrow_next, int valid_next, int ldq, int ldd, int wave,
                                                 int lane, int q0, int k0, int key, bool kkeep, float kfill, const float (&kreg)[32],
                                                 const float (&vreg)[32], floatx16 (&dk)[2], floatx16 (&dv)[2], uint32_t drop_row0,
                                                 int Tk, uint32_t drop_thr, uint32_t drop_seed, float drop_scale) {
    const int li = lane & 31, lh = lane >> 5;
    if (more) {
        at_tile_dma<true, 2>(rq, Qn, wave, lane, voq, qrow_next, valid_next, ldq);
        at_tile_dma<true, 2>(rd, Dn, wave, lane, vod, qrow_next, valid_next, ldd);
    }
    // these 32 queries all precede this wave's 32 keys: dS = 0, and P = exp(fill - max) = 0 unless a row's max IS the fill value
    if (CAUSAL && q0 + 31 < k0 && !__any(lse_a[q0 + li] < -1.0e9f)) return;
    const int fq = at_swz(li);
    floatx16 s, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        const float4 qv = *(const float4*)(Qc + li * 64 + (((2 * g + lh) ^ fq) << 2));
        const float4 dv4 = *(const float4*)(Dc + li * 64 + (((2 * g + lh) ^ fq) << 2));
        s = __builtin_amdgcn_mfma_f32_32x32x2f32(qv.x, kreg[g * 4 + 0], s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_32x32x2f32(qv.y, kreg[g * 4 + 1], s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_32x32x2f32(qv.z, kreg[g * 4 + 2], s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_32x32x2f32(qv.w, kreg[g * 4 + 3], s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x2f32(dv4.x, vreg[g * 4 + 0], dp, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x2f32(dv4.y, vreg[g * 4 + 1], dp, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x2f32(dv4.z, vreg[g * 4 + 2], dp, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x2f32(dv4.w, vreg[g * 4 + 3], dp, 0, 0, 0);
    }
    // per-query statistics of this half-wave's 16 queries: rows rowidx(r, lh) = {0-3, 8-11, 16-19, 24-27} + 4 lh
    float lsv[16], llv[16], dlv[16];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float4 t0 = *(const float4*)(lse_a + q0 + 8 * j + 4 * lh);
        const float4 t1 = *(const float4*)(lsl_a + q0 + 8 * j + 4 * lh);
        const float4 t2 = *(const float4*)(del_a + q0 + 8 * j + 4 * lh);
        lsv[4 * j + 0] = t0.x; lsv[4 * j + 1] = t0.y; lsv[4 * j + 2] = t0.z; lsv[4 * j + 3] = t0.w;
        llv[4 * j + 0] = t1.x; llv[4 * j + 1] = t1.y; llv[4 * j + 2] = t1.z; llv[4 * j + 3] = t1.w;
        dlv[4 * j + 0] = t2.x; dlv[4 * j + 1] = t2.y; dlv[4 * j + 2] = t2.z; dlv[4 * j + 3] = t2.w;
    }
    // rows of s / dp = queries rowidx(r, lh), column = this lane's key
    // s <- P (as dV sees it), dp <- dS / 0.125 (the 1 / sqrt(d) is applied when dK is stored)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int q = q0 + rowidx(r, lh);
        const bool keep = kkeep && (!CAUSAL || key <= q);
        const float sv = keep ? s[r] : kfill;            // fill value, or -inf (p = 0) for a key past the end
        const float p = ex2((sv - lsv[r]) - llv[r]);
        float pd = p, dpe = dp[r];
        if (DROP) {        // O = (P o M / (1-rate)) V: dV sees the dropped weights, dP arrives through the same mask
            const bool dm = drop_keep(drop_row0 + (uint32_t)rowidx(r, lh) * (uint32_t)Tk, drop_seed, drop_thr);
            pd = dm ? p * drop_scale : 0.f;
            dpe = dm ? dpe * drop_scale : 0.f;
        }
        const float ds = keep ? p * (dpe - dlv[r]) : 0.f;
        s[r] = pd; dp[r] = ds;
    }
    const unsigned qx = (unsigned)(uintptr_t)(const at_lds_c*)((const char*)Qc + lh * 1024 + li * 4);
    const unsigned dx = (unsigned)(uintptr_t)(const at_lds_c*)((const char*)Dc + lh * 1024 + li * 4);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const unsigned off = (unsigned)(((r & 3) + 8 * (r >> 2)) * 256), x = (unsigned)((r & 7) << 4);
        const unsigned a0 = (r & 8) ? 128 : 0, a1 = (r & 8) ? 0 : 128;
        dv[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(*(const at_lds_f*)(uintptr_t)((dx ^ x) + off + a0), s[r], dv[0], 0, 0, 0);
        dv[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(*(const at_lds_f*)(uintptr_t)((dx ^ x) + off + a1), s[r], dv[1], 0, 0, 0);
        dk[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(*(const at_lds_f*)(uintptr_t)((qx ^ x) + off + a0), dp[r], dk[0], 0, 0, 0);
        dk[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(*(const at_lds_f*)(uintptr_t)((qx ^ x) + off + a1), dp[r], dk[1], 0, 0, 0);
    }
}
#endif

template <bool CAUSAL, bool DROP>
__global__ __launch_bounds__(256, 2) void attn_bwd_kv_kernel(const float* __restrict__ Q, const float* __restrict__ K,
                                                          const float* __restrict__ V, const float* __restrict__ dO,
                                                          const float* __restrict__ lse, const float* __restrict__ delta,
                                                          float* __restrict__ dK, float* __restrict__ dV,
                                                          int Tq, int Tk, int C, int H, int ldq, int ldk, int relu_grad,
                                                          uint32_t drop_thr, uint32_t drop_seed, float drop_scale,
                                                          const float* __restrict__ qstat) {
#if __HIP_DEVICE_COMPILE__
    constexpr int QT = 32;                       // queries per tile
    constexpr int TF = QT * 64;
    extern __shared__ __attribute__((aligned(16))) float at_smem[];
    // (128-byte aligned by hand: the transposed reads XOR bits 4-6 of the LDS address, and static LDS in front of the dynamic
    // segment -- __syncthreads_or -- may shift its start)
    float* tiles = (float*)(at_lds_align(at_smem));                      // [2 sets][Q QT x 64 | dO QT x 64], later the epilogue's transpose scratch
    const int Tq64 = (Tq + 63) / 64 * 64;
    float* lse_a = tiles + (4 * TF > 4 * 32 * 65 ? 4 * TF : 4 * 32 * 65);      // [Tq64] each: reference (+inf: masked / past the end),
    float* lsl_a = lse_a + Tq64;                                                 // log2 row sum, delta
    float* del_a = lsl_a + Tq64;
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, lh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int ktile, head, n, Nn;
    attn_block_coords<CAUSAL>(H, false, ktile, head, n, Nn);      // causal: the first key tile sees every query
    const int hoff = head * DH;
    const int k0 = ktile * 128 + wave * 32, key = k0 + li;
    const long qbase = (long)n * Tq, kbase = (long)n * Tk;
    const long lrow = ((long)n * H + head) * Tq;

    bool degenerate = false;                     // a query before this key block whose visible keys were all masked
    const int kfirst_blk = ktile * 128;
    if (qstat) {
        // the precomputed masks: the three statistics in ONE pass of independent loads (no staging of the mask in LDS, no barrier between)
        for (int q = tid; q < Tq64; q += 256) {
            const bool in = q < Tq;
            const float mk = in ? qstat[lrow + q] : 0.f;
            const float l0 = in ? lse[lrow + q] : 0.f;
            const float l1 = in ? lse[(long)Nn * H * Tq + lrow + q] : 0.f;
            const float dl = in ? delta[lrow + q] : 0.f;
            if (CAUSAL && in && q < kfirst_blk && l0 < -1.0e9f) degenerate = true;
            lse_a[q] = (in && mk != 0.f) ? l0 : INFINITY;
            lsl_a[q] = l1;
            del_a[q] = dl;
        }
    } else {
        // query mask (1 / 0) into del_a for the moment, then the three statistics
        at_row_stats<1>(del_a, Q, qbase, Tq, Tq64, ldq, hoff, tid);
        __syncthreads();
        for (int q = tid; q < Tq64; q += 256) {
            const bool live = q < Tq && del_a[q] != 0.f;
            const float l0 = q < Tq ? lse[lrow + q] : 0.f;
            if (CAUSAL && q < Tq && q < kfirst_blk && l0 < -1.0e9f) degenerate = true;
            lse_a[q] = live ? l0 : INFINITY;
            lsl_a[q] = q < Tq ? lse[(long)Nn * H * Tq + lrow + q] : 0.f;
        }
        __syncthreads();                             // del_a (the mask) has been read by every thread
        for (int q = tid; q < Tq64; q += 256) del_a[q] = q < Tq ? delta[lrow + q] : 0.f;
    }
    // causal: query tiles wholly before this key block only matter for such degenerate rows (dV; the common case has none)
    int j0 = 0;
    if (CAUSAL) { const bool any = __syncthreads_or(degenerate); if (!any) j0 = kfirst_blk / QT; }
    const int ntiles = (Tq + QT - 1) / QT;

    auto rq = __builtin_amdgcn_make_buffer_rsrc((void*)Q, 0, (int)((((long)Nn * Tq - 1) * ldq + C) * 4), 0x00020000);
    auto rd = __builtin_amdgcn_make_buffer_rsrc((void*)dO, 0, (int)((((long)Nn * Tq - 1) * C + C) * 4), 0x00020000);
    unsigned voq[4], vod[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { voq[i] = at_voff<true>(i, wave, lane, ldq, hoff); vod[i] = at_voff<true>(i, wave, lane, C, hoff); }
    if (j0 < ntiles) {
        const int v0 = Tq - j0 * QT;
        at_tile_dma<true, 2>(rq, tiles, wave, lane, voq, qbase + j0 * QT, v0 < QT ? v0 : QT, ldq);
        at_tile_dma<true, 2>(rd, tiles + TF, wave, lane, vod, qbase + j0 * QT, v0 < QT ? v0 : QT, C);
    }

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

    for (int j = j0; j < ntiles; ++j) {
        const int q0 = j * QT;
        at_dma_barrier();                        // tile j has landed (and, the first time, the statistics are complete)
        float* cur = tiles + ((j - j0) & 1) * 2 * TF;
        float* nxt = tiles + ((j - j0 + 1) & 1) * 2 * TF;
        const int vnext = Tq - (q0 + QT);
        const uint32_t drop_row0 = (uint32_t)(((n * H + head) * Tq + q0) * Tk + key);
        attn_bwd_kv_tile<CAUSAL, DROP>(cur, cur + TF, nxt, nxt + TF, lse_a, lsl_a, del_a, rq, rd, j + 1 < ntiles, voq, vod,
                                       qbase + q0 + QT, vnext < QT ? vnext : QT, ldq, C, wave, lane, q0, k0, key, kkeep, kfill, kreg, vreg,
                                       dk, dv, drop_row0, Tk, drop_thr, drop_seed, drop_scale);
    }
    at_dma_barrier();
    store_tile_T(dK, tiles + wave * (32 * 65), dk, 0.125f, kbase, k0, Tk, ldk, hoff, lane, relu_grad ? K : nullptr);
    store_tile_T(dV, tiles + wave * (32 * 65), dv, 1.f, kbase, k0, Tk, ldk, hoff, lane, relu_grad ? V : nullptr);
#endif
}

// ------------------------------------------------------------------ attention backward: dQ
// Round 3: the K / V tiles arrive like the forward's (buffer-form LDS-DMA, two tile sets, 32-key tiles, one barrier per tile, three
// workgroups per CU).  Both tiles are stored with the swizzle of at_swz: V is only read as the A operand of dP = V.dO^T
// (ds_read_b128, row = lane); K also as the A operand of dQ += K^T.dS (ds_read_b32 of 32 consecutive floats of row
// rowidx(reg, half)): element (R, c) of a swizzled row sits at chunk (c >> 2) ^ f(R), and f(rowidx(r, half)) = r -- so the lane's
// byte address is its plain one XOR ((r & 7) << 4), the columns li and li + 32 are 128 bytes apart in the order bit 3 of r gives.
#if __HIP_DEVICE_COMPILE__
template <bool CAUSAL, bool DROP, int TK, class R>
__device__ __forceinline__ void attn_bwd_q_tile(const float* __restrict__ Kc, const float* __restrict__ Vc, float* __restrict__ Kn,
                                                float* __restrict__ Vn, const float* __restrict__ kb, R rk, R rv, bool more,
                                                const unsigned (&vok)[4], long krow_next, int valid_next, int ldk, int wave, int lane,
                                                int k0, int q0, int q, const float (&qreg)[32], const float (&doreg)[32],
                                                floatx16 (&dq)[2], float my_lse, float my_lsl, float my_del, uint32_t drop_base,
                                                uint32_t drop_thr, uint32_t drop_seed, float drop_scale) {
    const int li = lane & 31, lh = lane >> 5;
    if (more) {
        at_tile_dma<true, TK / 16>(rk, Kn, wave, lane, vok, krow_next, valid_next, ldk);
        at_tile_dma<true, TK / 16>(rv, Vn, wave, lane, vok, krow_next, valid_next, ldk);
    }
    const int fk = at_swz(li);
#pragma unroll
    for (int sub = 0; sub < TK / 32; ++sub) {
        if (CAUSAL && k0 + sub * 32 > q0 + 31) continue;       // future to this whole wave: dS = 0
        floatx16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const float4 kv = *(const float4*)(Kc + (sub * 32 + li) * 64 + (((2 * g + lh) ^ fk) << 2));
            const float4 vv = *(const float4*)(Vc + (sub * 32 + li) * 64 + (((2 * g + lh) ^ fk) << 2));
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(kv.x, qreg[g * 4 + 0], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(kv.y, qreg[g * 4 + 1], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(kv.z, qreg[g * 4 + 2], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(kv.w, qreg[g * 4 + 3], s, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(vv.x, doreg[g * 4 + 0], dp, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(vv.y, doreg[g * 4 + 1], dp, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(vv.z, doreg[g * 4 + 2], dp, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(vv.w, doreg[g * 4 + 3], dp, 0, 0, 0);
        }
        float kbv[16];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float4 t = *(const float4*)(kb + k0 + sub * 32 + 8 * j + 4 * lh);
            kbv[4 * j + 0] = t.x; kbv[4 * j + 1] = t.y; kbv[4 * j + 2] = t.z; kbv[4 * j + 3] = t.w;
        }
        // dp <- dS / 0.125 (the 1 / sqrt(d) is applied when dQ is stored)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int kl = sub * 32 + rowidx(r, lh);
            const bool keep = (kbv[r] > 1.0e38f) && (!CAUSAL || k0 + kl <= q);       // +inf: a real key
            float sv = s[r];
            if (CAUSAL) sv = (k0 + kl <= q) ? sv : FILL2;
            sv = vmin(sv, kbv[r]);
            const float p = ex2((sv - my_lse) - my_lsl);
            float dpe = dp[r];
            if (DROP) dpe = drop_keep(drop_base + (uint32_t)(k0 + kl), drop_seed, drop_thr) ? dpe * drop_scale : 0.f;
            dp[r] = keep ? p * (dpe - my_del) : 0.f;
        }
        // (the XOR on the 32-bit LDS address: on a generic pointer it turns the reads into flat loads)
        const unsigned kx = (unsigned)(uintptr_t)(const at_lds_c*)((const char*)Kc + sub * 32 * 256 + lh * 1024 + li * 4);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const unsigned krow = (kx ^ (unsigned)((r & 7) << 4)) + ((r & 3) + 8 * (r >> 2)) * 256;
            dq[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(*(const at_lds_f*)(uintptr_t)(krow + ((r & 8) ? 128 : 0)), dp[r], dq[0], 0, 0, 0);
            dq[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(*(const at_lds_f*)(uintptr_t)(krow + ((r & 8) ? 0 : 128)), dp[r], dq[1], 0, 0, 0);
        }
    }
}
#endif

template <bool CAUSAL, bool DROP, int TK>
__global__ __launch_bounds__(256, 3) void attn_bwd_q_kernel(const float* __restrict__ Q, const float* __restrict__ K,
                                                         const float* __restrict__ V, const float* __restrict__ dO,
                                                         const float* __restrict__ lse, const float* __restrict__ delta,
                                                         float* __restrict__ dQ, int Tq, int Tk, int C, int H, int ldq, int ldk, int relu_grad,
                                                         uint32_t drop_thr, uint32_t drop_seed, float drop_scale,
                                                         const float* __restrict__ kstat) {
#if __HIP_DEVICE_COMPILE__
    extern __shared__ __attribute__((aligned(16))) float at_smem[];
    constexpr int TF = TK * 64;                          // floats of one tile
    float* tiles = at_lds_align(at_smem);                // [2 sets][K TK x 64 | V TK x 64], later the epilogue's transpose scratch (128-byte aligned: the XOR reads)
    float* kb = tiles + (4 * TF > 4 * 32 * 65 ? 4 * TF : 4 * 32 * 65);      // [Tk rounded up to 64] key bias (min form)
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, lh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int qtile, head, n, Nn;
    attn_block_coords<CAUSAL>(H, true, qtile, head, n, Nn);
    const int hoff = head * DH;
    const int q0 = qtile * 128 + wave * 32, q = q0 + li;
    const long qbase = (long)n * Tq, kbase = (long)n * Tk;
    int ntiles = (Tk + TK - 1) / TK;
    if (CAUSAL) { const int lim = (qtile * 128 + 127) / TK + 1; if (lim < ntiles) ntiles = lim; }     // masked scores get no gradient

    auto rk = __builtin_amdgcn_make_buffer_rsrc((void*)K, 0, (int)((((long)Nn * Tk - 1) * ldk + C) * 4), 0x00020000);
    auto rv = __builtin_amdgcn_make_buffer_rsrc((void*)V, 0, (int)((((long)Nn * Tk - 1) * ldk + C) * 4), 0x00020000);
    unsigned vok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) vok[i] = at_voff<true>(i, wave, lane, ldk, hoff);
    at_tile_dma<true, TK / 16>(rk, tiles, wave, lane, vok, kbase, Tk, ldk);
    at_tile_dma<true, TK / 16>(rv, tiles + TF, wave, lane, vok, kbase, Tk, ldk);

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
    if (kstat) at_row_stats_load<0>(kb, kstat + ((long)n * H + head) * Tk, Tk, (Tk + 63) / 64 * 64, tid);
    else at_row_stats<0>(kb, K, kbase, Tk, (Tk + 63) / 64 * 64, ldk, hoff, tid);

    floatx16 dq[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { dq[0][r] = 0.f; dq[1][r] = 0.f; }
    const uint32_t drop_base = (uint32_t)(((n * H + head) * Tq + q) * Tk);

    for (int j = 0; j < ntiles; ++j) {
        const int k0 = j * TK;
        at_dma_barrier();                                // tile j has landed, nobody reads the other set any more
        float* cur = tiles + (j & 1) * 2 * TF;
        float* nxt = tiles + ((j + 1) & 1) * 2 * TF;
        const int vnext = Tk - (k0 + TK);
        attn_bwd_q_tile<CAUSAL, DROP, TK>(cur, cur + TF, nxt, nxt + TF, kb, rk, rv, j + 1 < ntiles, vok, kbase + k0 + TK,
                                          vnext < TK ? vnext : TK, ldk, wave, lane, k0, q0, q, qreg, doreg, dq, my_lse, my_lsl, my_del,
                                          drop_base, drop_thr, drop_seed, drop_scale);
    }
    at_dma_barrier();
    store_tile_T(dQ, tiles + wave * (32 * 65), dq, 0.125f, qbase, q0, Tq, ldq, hoff, lane, relu_grad ? Q : nullptr);
#endif
}

}  // namespace

// short sequences (Tq, Tk <= 128): attention_small.hip
int asr_attention_small_takes(int N, int Tq, int Tk, int C, int ldq, int ldk);
int asr_attention_small_bwd_launch(const float* Q, const float* K, const float* V, const float* O, const float* dO, const float* lse, int N, int Tq, int Tk,
                                   int C, int H, int ldq, int ldk, int causal, int relu_grad, int drop, uint32_t thr, uint32_t seed, float scale,
                                   float* dQ, float* dK, float* dV, const float* qstat, const float* kstat, void* stream);
int asr_attention_small_fwd_launch(const float* Q, const float* K, const float* V, int N, int Tq, int Tk, int C, int H, int ldq, int ldk, int causal,
                                   int drop, uint32_t thr, uint32_t seed, float scale, float* O, float* lse, const float* kstat, void* stream);

// ===================================================================== C ABI
// stats: [N][H][Tq] query masks, then [N][H][Tk] key biases (asr_attention_stats)
extern "C" size_t asr_attention_stats_floats(int N, int Tq, int Tk, int H) { return (size_t)N * H * ((size_t)Tq + Tk); }

extern "C" int asr_attention_stats(const float* Q, const float* K, int N, int Tq, int Tk, int C, int H, int ldq, int ldk, float* stats,
                                   void* stream) {
    if (!Q || !K || !stats || N < 1 || Tq < 1 || Tk < 1 || H < 1 || C != H * DH) return ASR_ERR_BAD_ARG;
    if (ldq < C || ldk < C || (ldq & 3) || (ldk & 3) || ((((uintptr_t)Q) | ((uintptr_t)K)) & 15)) return ASR_ERR_BAD_ARG;
    const long groups = (((long)N * (Tq > Tk ? Tq : Tk) + 3) / 4) * H;
    hipLaunchKernelGGL(attn_stats_kernel, dim3(asr_cdiv(groups * 16, 256), 2), dim3(256), 0, (hipStream_t)stream, Q, K, N, Tq, Tk, H, ldq, ldk,
                       stats, stats + (size_t)N * H * Tq);
    ASR_CHECK_LAUNCH("attention_stats");
    return ASR_OK;
}

extern "C" int asr_attention_fwd_s(const float* Q, const float* K, const float* V, int N, int Tq, int Tk, int C, int H,
                                   int ldq, int ldk, int causal, float dropout_rate, unsigned int seed, float* O, float* lse,
                                   const float* stats, void* stream) {
    if (!Q || !K || !V || !O || !lse || N < 1 || Tq < 1 || Tk < 1 || H < 1 || C != H * DH) return ASR_ERR_BAD_ARG;
    const float* kst = stats ? stats + (size_t)N * H * Tq : nullptr;
    if (ldq < C || ldk < C || (ldq & 3) || (ldk & 3)) return ASR_ERR_BAD_ARG;
    if (dropout_rate < 0.f || dropout_rate >= 1.f || (double)N * H * Tq * Tk >= 4294967296.0) return ASR_ERR_BAD_ARG;
    dim3 grid(asr_cdiv(Tq, 128), H, N), grid_causal(N * H, asr_cdiv(Tq, 128), 1);      // see attn_block_coords
    hipStream_t st = (hipStream_t)stream;
    const uint32_t thr = drop_threshold(dropout_rate);
    const float sc = 1.0f / (1.0f - dropout_rate);
    if (asr_attention_small_takes(N, Tq, Tk, C, ldq, ldk))    // a function of the sequence lengths only
        return asr_attention_small_fwd_launch(Q, K, V, N, Tq, Tk, C, H, ldq, ldk, causal, dropout_rate > 0.f, thr, seed, sc, O, lse, kst, stream);
    // the tile sets (or the epilogue's scratch) + the key bias of every key of a (sample, head); 32-bit buffer offsets
    constexpr int ATK = 32;
    const size_t lds = (size_t)((4 * ATK * 64 > 4 * 32 * 65 ? 4 * ATK * 64 : 4 * 32 * 65) + asr_cdiv(Tk, 64) * 64) * sizeof(float);
    if (lds > 160 * 1024 || (long)N * Tk * ldk * 4 >= (1L << 31)) return ASR_ERR_UNSUPPORTED;
#define ASR_ATTN_FWD(CA, DR, G)                                                                                                 \
    do {                                                                                                                       \
        auto kern = attn_fwd_kernel<CA, DR, ATK>;                                                                                \
        static size_t have = 0;           /* a kernel with static LDS too (__syncthreads_and) rejects the 160 KB blanket request */ \
        if (lds > have) { if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) { (void)hipGetLastError(); return ASR_ERR_UNSUPPORTED; } have = lds; } \
        hipLaunchKernelGGL(kern, G, dim3(256), lds, st, Q, K, V, O, lse, Tq, Tk, C, H, ldq, ldk, thr, seed, sc, kst);           \
    } while (0)
    if (dropout_rate > 0.f) {
        if (causal) ASR_ATTN_FWD(true, true, grid_causal); else ASR_ATTN_FWD(false, true, grid);
    } else {
        if (causal) ASR_ATTN_FWD(true, false, grid_causal); else ASR_ATTN_FWD(false, false, grid);
    }
#undef ASR_ATTN_FWD
    ASR_CHECK_LAUNCH("attention_fwd");
    return ASR_OK;
}

extern "C" int asr_attention_fwd_p(const float* Q, const float* K, const float* V, int N, int Tq, int Tk, int C, int H,
                                   int ldq, int ldk, int causal, float dropout_rate, unsigned int seed, float* O, float* lse,
                                   void* stream) {
    return asr_attention_fwd_s(Q, K, V, N, Tq, Tk, C, H, ldq, ldk, causal, dropout_rate, seed, O, lse, nullptr, stream);
}

extern "C" int asr_attention_fwd(const float* Q, const float* K, const float* V, int N, int Tq, int Tk, int C, int H,
                                 int causal, float dropout_rate, unsigned int seed, float* O, float* lse, void* stream) {
    return asr_attention_fwd_p(Q, K, V, N, Tq, Tk, C, H, C, C, causal, dropout_rate, seed, O, lse, stream);
}

extern "C" int asr_attention_bwd_s(const float* Q, const float* K, const float* V, const float* O, const float* dO,
                                   const float* lse, int N, int Tq, int Tk, int C, int H, int ldq, int ldk, int causal,
                                   int relu_grad, float dropout_rate, unsigned int seed,
                                   float* dQ, float* dK, float* dV, float* delta_ws, const float* stats, void* stream) {
    if (ldq < C || ldk < C || (ldq & 3) || (ldk & 3)) return ASR_ERR_BAD_ARG;
    const float* qst = stats;
    const float* kst = stats ? stats + (size_t)N * H * Tq : nullptr;
    if (!Q || !K || !V || !O || !dO || !lse || !dQ || !dK || !dV || !delta_ws) return ASR_ERR_BAD_ARG;
    if (N < 1 || Tq < 1 || Tk < 1 || H < 1 || C != H * DH) return ASR_ERR_BAD_ARG;
    if (dropout_rate < 0.f || dropout_rate >= 1.f || (double)N * H * Tq * Tk >= 4294967296.0) return ASR_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (asr_attention_small_takes(N, Tq, Tk, C, ldq, ldk))    // short sequences: two launches, delta formed inside them (delta_ws is not written)
        return asr_attention_small_bwd_launch(Q, K, V, O, dO, lse, N, Tq, Tk, C, H, ldq, ldk, causal, relu_grad, dropout_rate > 0.f,
                                              drop_threshold(dropout_rate), seed, 1.0f / (1.0f - dropout_rate), dQ, dK, dV, qst, kst, stream);
    const long groups = (long)N * Tq * H;
    hipLaunchKernelGGL(attn_delta_kernel, dim3(asr_cdiv(groups * 16, 256)), dim3(256), 0, st, O, dO, delta_ws, N, Tq, C, H);
    dim3 gkv(asr_cdiv(Tk, 128), H, N), gq(asr_cdiv(Tq, 128), H, N);
    dim3 ckv(N * H, asr_cdiv(Tk, 128), 1), cq(N * H, asr_cdiv(Tq, 128), 1);
    const float* dl = (const float*)delta_ws;
    const uint32_t thr = drop_threshold(dropout_rate);
    const float sc = 1.0f / (1.0f - dropout_rate);
    // dQ kernel: the tile sets (or the epilogue's scratch) + the key bias of every key of a (sample, head); 32-bit buffer offsets
    const size_t ldsq = (size_t)((4 * 32 * 64 > 4 * 32 * 65 ? 4 * 32 * 64 : 4 * 32 * 65) + asr_cdiv(Tk, 64) * 64 + 32) * sizeof(float);
    if (ldsq > 160 * 1024 || (long)N * Tk * ldk * 4 >= (1L << 31)) return ASR_ERR_UNSUPPORTED;
    // dK / dV kernel: the Q / dO tile sets (or the scratch) + three statistics per query of a (sample, head)
    const size_t ldskv = (size_t)((4 * 32 * 64 > 4 * 32 * 65 ? 4 * 32 * 64 : 4 * 32 * 65) + 3 * asr_cdiv(Tq, 64) * 64 + 32) * sizeof(float);
    if (ldskv > 160 * 1024 || (long)N * Tq * ldq * 4 >= (1L << 31) || (long)N * Tq * C * 4 >= (1L << 31)) return ASR_ERR_UNSUPPORTED;
#define ASR_ATTN_BWD(CA, DR, GKV, GQ)                                                                                          \
    do {                                                                                                                       \
        auto kkv = attn_bwd_kv_kernel<CA, DR>;                                                                                 \
        static size_t havekv = 0;                                                                                              \
        if (ldskv > havekv) { if (hipFuncSetAttribute((const void*)kkv, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldskv) != hipSuccess) { (void)hipGetLastError(); return ASR_ERR_UNSUPPORTED; } havekv = ldskv; } \
        hipLaunchKernelGGL(kkv, GKV, dim3(256), ldskv, st, Q, K, V, dO, lse, dl, dK, dV, Tq, Tk, C, H,                          \
                           ldq, ldk, relu_grad, thr, seed, sc, qst);                                                                     \
        auto kq = attn_bwd_q_kernel<CA, DR, 32>;                                                                               \
        static size_t have = 0;                                                                                                \
        if (ldsq > have) { if (hipFuncSetAttribute((const void*)kq, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsq) != hipSuccess) { (void)hipGetLastError(); return ASR_ERR_UNSUPPORTED; } have = ldsq; } \
        hipLaunchKernelGGL(kq, GQ, dim3(256), ldsq, st, Q, K, V, dO, lse, dl, dQ, Tq, Tk, C, H,                                  \
                           ldq, ldk, relu_grad, thr, seed, sc, kst);                                                                     \
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

extern "C" int asr_attention_bwd_p(const float* Q, const float* K, const float* V, const float* O, const float* dO,
                                   const float* lse, int N, int Tq, int Tk, int C, int H, int ldq, int ldk, int causal,
                                   int relu_grad, float dropout_rate, unsigned int seed,
                                   float* dQ, float* dK, float* dV, float* delta_ws, void* stream) {
    return asr_attention_bwd_s(Q, K, V, O, dO, lse, N, Tq, Tk, C, H, ldq, ldk, causal, relu_grad, dropout_rate, seed, dQ, dK, dV, delta_ws,
                               nullptr, stream);
}

extern "C" int asr_attention_bwd(const float* Q, const float* K, const float* V, const float* O, const float* dO,
                                 const float* lse, int N, int Tq, int Tk, int C, int H, int causal, int relu_grad,
                                 float dropout_rate, unsigned int seed,
                                 float* dQ, float* dK, float* dV, float* delta_ws, void* stream) {
    return asr_attention_bwd_p(Q, K, V, O, dO, lse, N, Tq, Tk, C, H, C, C, causal, relu_grad, dropout_rate, seed, dQ, dK, dV,
                               delta_ws, stream);
}

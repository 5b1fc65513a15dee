// 3x3 convolution (forward and data-gradient view) by Winograd F(2x2, 3x3) on the fp32 MFMA pipe: the engines' default for the
// 3x3 layers with N % 64 == 0 (asr_winograd_supported), 1.3-1.6x tap_gemm_kernel_v5 per layer.  DESIGN.md section 9 item 4 has
// the design decisions, every measurement and the dead ends; tests/test_wino_gpu.py, tools/bench_wino.py, tools/trace_wino.py.
//
//   Y = A^T [ sum_ci (G g G^T) (.) (B^T d B) ] A        16 multiplies per 2x2 output tile, input and output channel
//                                                        instead of 36; still fp32
//
// One GEMM per transform position xi = 0..15:  M_xi[tile][co] = sum_ci V_xi[tile][ci] * U_xi[ci][co].
//   * U = G g G^T is prepared once per optimiser step (asr_winograd_weights2: [16][K][N], forward or data-gradient view);
//   * a work item is 64 consecutive tiles x 64 output channels, the input channels come in chunks of 8: the raw 4x4 patches of
//     the tiles and the 16 weight matrices ([xi][ci 8][co 64]) arrive by LDS-DMA in two buffer sets, one barrier per chunk;
//   * the lanes transform their own patch (B^T d B) and V goes straight into the MFMAs as the A operand -- the transformed
//     input never exists in memory;
//   * the inverse transform A^T M A is lane-local per accumulator register and yields four 32-row blocks -- the four pixels
//     of the tiles -- which go through the SAME epilogues as the tap-GEMM kernels (tap_epilogue: bias / ReLU / BN affine /
//     accumulate / the gated backward prologues) via their row tables; for a pooled cell the 2x2 pool of the BN output is
//     formed from those four pixels in registers (asr_tap_gemm_wino_pool).
// wino8_kernel: eight waves, the sixteen positions split 8 + 8 over the two waves of a SIMD, persistent workgroups, DMA pieces
// between the MFMAs.  (The first, four-wave form -- sixteen accumulators per wave, one wave per SIMD -- kept the matrix pipe
// busy 34 % of the time and was removed in round 3.)
// Planes as everywhere else ([B][H+1][W+1][C], zero borders): patch rows / columns that stick out read the border; an odd
// plane width needs the fourth patch column of the last tile column zeroed (it would wrap into the next pixel row).
#include "asr_common.h"
#include "reduce.h"
#include "tap_epilogue.h"
#include <stdlib.h>

namespace {

typedef __attribute__((address_space(3))) float wn_lds_f;
typedef const __attribute__((address_space(1))) float wn_glb_f;

struct WinoArgs {
    TapGemmArgs g;
    const float* Ut;        // [16][K][N]
    int TH, TW;             // tile rows / columns per image
    long ntiles;            // B * TH * TW
    int wodd;               // plane width is odd
    float* pool_y;          // eight-wave kernel: pooled BN output plane [B][H/2+1][W/2+1][N] of a fused 2x2 pool, or null
    int pool_mode, H2, W2;  // 1 average, 2 maximum
    // compact maximum pool (wino11_kernel): instead of the pre-pool activation plane, the activation AT the maximum of each window
    // (pooled geometry) and the window position of that maximum, two bit planes of 32 channels per pooled pixel:
    // idx[(pixel * (N / 32) + channel block) * 2 + bit] -- all the maximum-pool backward needs (asr_tap_gemm_gated_poolmax)
    float* pool_amax;
    unsigned* pool_idx;     // forward: written; gated launch of mode 4: read (g.gate_a = the amax plane)
    // column-blocked tile order (wino11_kernel): the tile columns of an image are cut into ncb blocks of 11..15 columns; the tiles
    // of a block are numbered row-major WITHIN the block and a work item is 64 consecutive tiles of one block
    int ncb;                // number of column blocks (0: plain order, wino8_kernel)
    int cb_tj0[8];          // first tile column of the block
    int cb_w[8];            // tile columns of the block
    int cb_it0[9];          // first item of the block within an image; cb_it0[ncb] = items per image
    float inv_ipi;          // 1 / items per image and 1 / cb_w: the kernel's index divisions are a convert, a multiply-add and a convert
    float cb_invw[8];       // (exact for the index ranges here: floor((x + 0.5) / d) = floor(x / d), x < 2^22)
};

constexpr int WT = 64;                      // tiles per workgroup
constexpr int WC = 64;                      // output channels per workgroup
constexpr int WKC = 8;                      // input channels per chunk
constexpr int RAW_F = 16 * 2 * WT * 4;      // floats of one raw buffer   ([pixel][quad][tile] slots of 4 floats)
constexpr int U_F = 16 * WKC * WC;          // floats of one weight buffer ([xi][ci][co])

// U = G g G^T for every (k, n): g(kh, kw) = W[kh][kw][k][n] (forward) or W[2-kh][2-kw][n][k] (data-gradient view)
__global__ void wino_weights_kernel(const float* __restrict__ W, int K, int N, int ldw, int wmode, float* __restrict__ out, float* __restrict__ out2) {
    const long total = (long)K * N;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int k = (int)(i / N), n = (int)(i - (long)k * N);
        float g[3][3];
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw)
                g[kh][kw] = (wmode == 0) ? W[((long)(kh * 3 + kw) * K + k) * ldw + n]
                                         : W[((long)((2 - kh) * 3 + (2 - kw)) * N + n) * ldw + k];
        float t[4][3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            t[0][c] = g[0][c];
            t[1][c] = 0.5f * ((g[0][c] + g[1][c]) + g[2][c]);
            t[2][c] = 0.5f * ((g[0][c] - g[1][c]) + g[2][c]);
            t[3][c] = g[2][c];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float u0 = t[r][0], u1 = 0.5f * ((t[r][0] + t[r][1]) + t[r][2]), u2 = 0.5f * ((t[r][0] - t[r][1]) + t[r][2]), u3 = t[r][2];
            out[((long)(r * 4 + 0) * K + k) * N + n] = u0;
            out[((long)(r * 4 + 1) * K + k) * N + n] = u1;
            out[((long)(r * 4 + 2) * K + k) * N + n] = u2;
            out[((long)(r * 4 + 3) * K + k) * N + n] = u3;
            if (out2) {
                // second layout (wino11_kernel): [K / 8][xi 16][N / 32][k-pair pair][lane half][co 32][2]; channel kc * 8 + 4 lh + kp of
                // a chunk is the k-pair kp of lane half lh (the MFMA contracts channels kp and 4 + kp of the chunk together)
                const int kc = k >> 3, lhh = (k >> 2) & 1, kp = k & 3;
                const long base = (((long)kc * 16 + r * 4) * (N >> 5) + (n >> 5)) * 256 + (kp >> 1) * 128 + lhh * 64 + (n & 31) * 2 + (kp & 1);
                const long xs = (long)(N >> 5) * 256;
                out2[base] = u0; out2[base + xs] = u1; out2[base + 2 * xs] = u2; out2[base + 3 * xs] = u3;
            }
        }
    }
}

// ---- eight waves per workgroup: 64 tiles x 64 channels, the sixteen positions split between the two waves of a
// SIMD (wave xh owns transform columns 2 xh and 2 xh + 1: 8 accumulators = 128 registers), so that two waves cover each
// other's stalls -- with one wave per SIMD the pipe was busy 34 % of the time.  A wave needs three patch columns (12
// pixels, 20 adds per 8 MFMAs); the two halves meet once per item: each wave finishes one pixel ROW of the tiles and gets
// the two column sums it lacks from its partner through LDS.
// Raw layout of this kernel: [pixel 16][quad 2][tile 64] 16-byte slots.  Lane half lh reads quad lh of its tile with ONE
// conflict-free ds_read_b128 per pixel and chunk: the MFMA k-pair kp contracts channels kp (half 0) and 4 + kp (half 1) of the
// chunk -- the pairing of channels into k-pairs is free as long as the weights follow it -- so all four floats are used.
typedef float wn_f2 __attribute__((ext_vector_type(2)));
// workgroup barrier that orders LDS traffic only: __syncthreads() also waits for every outstanding global access of the wave
// (vmcnt(0)) -- the next item's DMA and the epilogue's stores, which nothing in the workgroup is waiting for.  Written as
// plain instructions: a release / acquire fence pair on the local address space still makes hipcc emit vmcnt(0) while an
// LDS-DMA is in flight (it counts the DMA as an LDS write), which is how the tail of wino8_kernel waited for the next item's
// first chunk until round 3.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
// two k-pairs at a time: components (x, y) or (z, w) of the float4s are aligned register pairs, so the 20 adds of a batch
// become 20 packed adds for TWO batches (v_pk_add_f32) -- on this pipe every vector instruction costs matrix time
template <int XH>
__device__ __forceinline__ void wino8_transform(const float4 (&d)[12], int half, wn_f2 (&v)[8]) {
    // d[r * 3 + cc] = patch pixel (r, XH + cc)
    auto comp = [&](const float4& q) { return half == 0 ? wn_f2{q.x, q.y} : wn_f2{q.z, q.w}; };
    wn_f2 t[4][3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const wn_f2 d0 = comp(d[0 + c]), d1 = comp(d[3 + c]), d2 = comp(d[6 + c]), d3 = comp(d[9 + c]);
        t[0][c] = d0 - d2; t[1][c] = d1 + d2; t[2][c] = d2 - d1; t[3][c] = d1 - d3;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        if (XH == 0) { v[r * 2 + 0] = t[r][0] - t[r][2]; v[r * 2 + 1] = t[r][1] + t[r][2]; }       // columns 0, 1 of V
        else         { v[r * 2 + 0] = t[r][1] - t[r][0]; v[r * 2 + 1] = t[r][0] - t[r][2]; }       // columns 2, 3 of V
    }
}

// DMA of one chunk by eight waves, eight pieces per wave: raw piece = pixel * 2 + quad (64 tiles), weight piece = xi * 2 +
// (ci >> 2).  A raw piece is 64 lanes on 64 different 128-byte lines: the texture addresser takes them one per clock, so
// the 32 raw pieces of a chunk occupy it for half the chunk's MFMA time -- and a wave whose DMA instruction waits for the
// addresser issues nothing else.  The pieces are therefore handed out ONE at a time (wino8_piece) and the callers put
// them between MFMAs, so that the other wave of the SIMD (and this wave's MFMAs in flight) keep the matrix pipe busy.
struct Wino8Dma {
    const char* abase; const char* ubase;
    unsigned off_t, off_u;
    int WPl, lda, N, wave;
    long ustride_xi;
};
__device__ __forceinline__ void wino8_piece(const Wino8Dma& q, float* __restrict__ raw, float* __restrict__ ub, int kc, int j) {
    const int p = q.wave + 8 * (j & 3);
    if (j < 4) {
        const int px = p >> 1, quad = p & 1;
        const int r = px >> 2, c = px & 3;
        const char* pb = q.abase + (((long)r * q.WPl + c) * q.lda + kc * WKC + quad * 4) * 4;      // uniform
        __builtin_amdgcn_global_load_lds((wn_glb_f*)(pb + q.off_t), (wn_lds_f*)(raw + p * 256), 16, 0, 0);
    } else {
        const int xi = p >> 1, cig = p & 1;
        const char* pb = q.ubase + (xi * q.ustride_xi + (long)(kc * WKC + cig * 4) * q.N) * 4;     // uniform
        __builtin_amdgcn_global_load_lds((wn_glb_f*)(pb + q.off_u), (wn_lds_f*)(ub + p * 256), 16, 0, 0);
    }
}
__device__ __forceinline__ void wino8_stage(const Wino8Dma& q, float* __restrict__ raw, float* __restrict__ ub, int kc) {
#pragma unroll
    for (int j = 0; j < 8; ++j) wino8_piece(q, raw, ub, kc, j);
}

template <int XH>
__device__ __forceinline__ void wino8_chunk(const float* __restrict__ raw, const float* __restrict__ ub, float* __restrict__ rawn,
                                            float* __restrict__ ubn, bool prefetch, const Wino8Dma& q, int kcn,
                                            int aoff, int boff, bool wodd, bool zero_c3, floatx16 (&acc)[8]) {
    auto load_u = [&](float (&u)[8], int kp) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int j = 0; j < 2; ++j) u[r * 2 + j] = ub[((r * 4 + 2 * XH + j) * WKC + kp) * WC + boff];
    };
    // eight MFMAs; behind MFMA 1, 3, 5, 7 one DMA piece of the next chunk (pieces first .. first + 3), if any
    auto mfmas = [&](const wn_f2 (&v)[8], int h, const float (&u)[8], int first) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(h ? v[i].y : v[i].x, u[i], acc[i], 0, 0, 0);
            if (first >= 0 && (i & 1)) {
                __builtin_amdgcn_sched_barrier(0);
                if (prefetch) wino8_piece(q, rawn, ubn, kcn, first + (i >> 1));
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    float4 d[12];
    float ua[8], ubb[8];
    wn_f2 v[8];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) d[r * 3 + c] = *(const float4*)(raw + (r * 4 + XH + c) * (2 * WT * 4) + aoff);
    load_u(ua, 0);
    load_u(ubb, 1);
    __builtin_amdgcn_sched_barrier(0);
    if (XH == 1 && wodd) {       // patch column 3 is this half's third column
#pragma unroll
        for (int r = 0; r < 4; ++r) if (zero_c3) d[r * 3 + 2] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // two double batches (k-pairs 0, 1 and 2, 3): the adds of a double batch all come before its sixteen MFMAs (an add behind
    // an fp32 MFMA waits for it, and the next MFMA pays again); the weights of the next double batch are requested as soon as
    // the MFMAs that read the current ones have issued.  Raw pieces (j 0..3) and weight pieces (4..7) alternate.
    wino8_transform<XH>(d, 0, v);
    __builtin_amdgcn_sched_barrier(0);
    mfmas(v, 0, ua, 0);
    __builtin_amdgcn_sched_barrier(0);
    load_u(ua, 2);
    __builtin_amdgcn_sched_barrier(0);
    mfmas(v, 1, ubb, 4);
    __builtin_amdgcn_sched_barrier(0);
    load_u(ubb, 3);
    __builtin_amdgcn_sched_barrier(0);
    wino8_transform<XH>(d, 1, v);
    __builtin_amdgcn_sched_barrier(0);
    mfmas(v, 0, ua, -1);
    __builtin_amdgcn_sched_barrier(0);
    mfmas(v, 1, ubb, -1);
}

// The item tail of wino8_kernel: exchange of the column sums between the two halves, the lane-local inverse
// transform, the fused pool and the epilogues.  rfree / ufree: the raw and the weight buffer set that no DMA is writing (8192
// floats each).
// `before_stores` runs after the exchange and before the first global store of the epilogues.
struct WinoNoHook { __device__ __forceinline__ void operator()() const {} };
// NWH = waves per position half (4); ws = this wave's index within its half.
template <int XH, class Hook = WinoNoHook, int NWH = 4>
__device__ __forceinline__ void wino_item_tail(const WinoArgs& args, floatx16 (&acc)[8], float* rfree, float* ufree, const int* rowa,
                                               const int* rowy, const int* prow, int wave, int lane, int wm, int wn, int n0, int blk,
                                               float pool_bs, float pool_sc, float pool_sh, Hook before_stores = Hook()) {
    const int ws = wave & (NWH - 1);
    const TapGemmArgs& g = args.g;
    // own column sums: m(r, j) = acc[r * 2 + j], transform column c = 2 XH + j
    floatx16 s0[2], s1[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s0[j][r] = (acc[0 + j][r] + acc[2 + j][r]) + acc[4 + j][r];
            s1[j][r] = (acc[2 + j][r] - acc[4 + j][r]) - acc[6 + j][r];
        }
    // half 0 finishes pixel row 0 and needs s0 of columns 2, 3; half 1 finishes pixel row 1 and needs s1 of columns 0, 1
    // each wave hands its partner 2 x 16 registers x 64 lanes = 2048 floats: waves 0-3 through the raw set, waves 4-7
    // through the weight set (8192 floats each)
    float* xch = (XH == 0 ? rfree : ufree) + ws * 2048;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) xch[(j * 16 + r) * 64 + lane] = XH == 0 ? s1[j][r] : s0[j][r];
    lds_barrier();
    const float* pch = (XH == 0 ? ufree : rfree) + ws * 2048;
    floatx16 out[2][1];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float c0, c1, c2, c3;
        if (XH == 0) { c0 = s0[0][r]; c1 = s0[1][r]; c2 = pch[(0 * 16 + r) * 64 + lane]; c3 = pch[(1 * 16 + r) * 64 + lane]; }
        else         { c0 = pch[(0 * 16 + r) * 64 + lane]; c1 = pch[(1 * 16 + r) * 64 + lane]; c2 = s1[0][r]; c3 = s1[1][r]; }
        out[0][0][r] = (c0 + c1) + c2;
        out[1][0][r] = (c1 - c2) - c3;
    }
    lds_barrier();                               // every wave has read its partner's values: the weight set becomes scratch
    before_stores();
    if (args.pool_y) {
        // Fused 2x2 pool (a Winograd tile IS a pooling window): this wave holds pixels (XH, 0) and (XH, 1) of its tiles as
        // out[0] / out[1], lane = output channel.  BN(ReLU(x + bias)) and the row's pair lane-locally, then half 1 hands its
        // pair to half 0 through LDS (behind the scratch areas) and half 0 stores the pooled pixel: 128-byte rows of 32
        // channels.  Same arithmetic and association as asr_pool_fwd on the stored activation: bit-identical.
        const float bs = pool_bs, scv = pool_sc, shv = pool_sh;
        floatx16 pm;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float x0 = out[0][0][r] + bs, x1 = out[1][0][r] + bs;
            if (g.relu == 1) { x0 = fmaxf(x0, 0.f); x1 = fmaxf(x1, 0.f); }
            else if (g.relu == 2) { x0 = tanhf(x0); x1 = tanhf(x1); }
            const float v0 = fmaf(scv, x0, shv), v1 = fmaf(scv, x1, shv);
            pm[r] = args.pool_mode == 1 ? v0 + v1 : fmaxf(v0, v1);
        }
        float* pex = NWH == 4 ? (wave & 2 ? ufree : rfree) + 4 * (32 * 33) + (wave & 1) * 1024      // pair (wm, wn)
                              : rfree + 2 * (32 * 33) + ws * 1024;                                 // pair wm
        if (XH == 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) pex[r * 64 + lane] = pm[r];
        }
        lds_barrier();
        if (XH == 0) {
            floatx16 po[1][1];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float qv = pex[r * 64 + lane];
                po[0][0][r] = args.pool_mode == 1 ? 0.25f * (pm[r] + qv) : fmaxf(pm[r], qv);
            }
            // stored through the shared epilogue (one 32 x 32 block, float4 rows: 4 store instructions instead of 16 scalar
            // ones -- the tail is store-issue bound), row table = the pooled rows of this wave's 32 tiles
            TapGemmArgs gp = g;
            gp.out_a = args.pool_y; gp.ldo_a = g.N; gp.out_y = nullptr; gp.bias = nullptr; gp.relu = 0; gp.accumulate = 0; gp.gate_mode = 0;
            float* pscr = (XH == 0 ? rfree : ufree) + ws * (32 * 33);
            tap_epilogue<1, 1>(gp, po, pscr, prow, prow, wm * 32, n0 + wn * 32, lane, 0);
        }
    }
    // transpose scratch (32 x 33 floats per wave): waves 0-3 in the raw set, waves 4-7 in the weight set
    float* scratch = (XH == 0 ? rfree : ufree) + ws * (32 * 33);
    tap_epilogue<2, 1>(g, out, scratch, rowa, rowy, wm * 128 + XH * 64, n0 + wn * 32, lane, (blk * 2 + wm) * 2 + XH);
}

// per-lane DMA offsets of one work item of the eight-wave kernel: tile t0 + lane (clamped): byte offset of the top-left pixel
// of its patch (planes are < 4 GB); weights: row lane / 16 of a 4-row piece, column quad lane % 16
__device__ __forceinline__ void wino8_offsets(const WinoArgs& args, long t0, int n0, int lane, unsigned& off_t, unsigned& off_u) {
    const TapGemmArgs& g = args.g;
    long t = t0 + lane;
    if (t > args.ntiles - 1) t = args.ntiles - 1;
    const int per = args.TH * args.TW;
    const int b = (int)(t / per);
    const int rr = (int)(t - (long)b * per);
    const int ti = rr / args.TW, tj = rr - ti * args.TW;
    off_t = (unsigned)((((long)b * g.HPWP + (long)(2 * ti) * g.WP + 2 * tj) * g.lda) * 4);
    off_u = (unsigned)(((long)(lane >> 4) * g.N + n0 + (lane & 15) * 4) * 4);
}

// Persistent workgroups (one per CU; work item = (tile block, channel block), channel blocks of a tile block adjacent).  The
// LAST chunk of an item requests chunk 0 of the NEXT item where the other chunks request their successor (between its
// MFMAs), so after it one buffer set is filling for the next item and the other is free for the exchange of the two halves
// and the epilogue's transpose scratch; the barriers of that tail order LDS only (lds_barrier), they do not wait for the
// DMA in flight or the epilogue's stores.  In-kernel stamps (WINO_TRACE, 64 -> 128 channels at 400 x 50): 8 chunks 26.4 us
// (2.9 each; 4.2-4.6 where the fetch crossed into new 128-byte lines or a new item), exchange 1.1 us, epilogue 3.2 us
// (store issue), 0.6 us set-up -- before the DMA moved between the MFMAs a chunk took 3.2 us and the tail 9 us.
template <int XH>
__device__ __forceinline__ void wino8_body(const WinoArgs& args, float* smem) {
    const TapGemmArgs& g = args.g;
    int* rowa = (int*)smem;                          // [2 wave rows][4 pixels][32 tiles]
    int* rowy = rowa + 256;
    int* prow = rowy + 256;                          // [64 tiles]: row of the tile's pooled pixel (fused 2x2 pool), or -1
    float* bufs = smem + 576;                        // raw0 | raw1 | u0 | u1
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, lh = lane >> 5;
    // the wave id is wave-uniform, but only the hardware knows: read it into a scalar register so that piece numbers and LDS
    // destinations of the DMA are scalar arithmetic (else every piece costs a v_readfirstlane and vector adds)
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = (wave >> 1) & 1, wn = wave & 1;                       // wave = xh * 4 + wm * 2 + wn
    const char* abase = (const char*)g.A;
    const char* ubase = (const char*)args.Ut;
    const long ustride_xi = (long)g.K * g.N;
    const int aoff = (lh * WT + wm * 32 + li) * 4;   // slot (quad = lh, tile) of a pixel
    const int boff = lh * 4 * WC + wn * 32 + li;     // half 1 contracts channels 4 .. 7 of the chunk
    const int nkc = g.K / WKC;
    const int nnb = g.ntn, nwork = g.ntm * nnb;

    int w = blockIdx.x;
    if (w >= nwork) return;
    // Which item a workgroup takes in a round: workgroup b runs on XCD b % 8 (round-robin dispatch), each XCD has its own L2.
    // Within a full round of G items XCD x takes the CONTIGUOUS items x G/8 .. (x + 1) G/8 - 1, so the channel blocks of a
    // tile block (adjacent items: the same input tiles) and neighbouring tile blocks (shared patch rows) are fetched through
    // one L2 at about the same time.  A partial last round keeps the plain order.
    const int G = gridDim.x;
    auto item_of = [&](int wl) {
        const int r0 = (wl / G) * G;
        if ((G & 7) || r0 + G > nwork) return wl;
        const int p = wl - r0;
        return r0 + (p & 7) * (G >> 3) + (p >> 3);
    };
    Wino8Dma q;
    q.abase = abase; q.ubase = ubase; q.WPl = g.WP; q.lda = g.lda; q.N = g.N; q.wave = wave; q.ustride_xi = ustride_xi;
    { const int it = item_of(w); wino8_offsets(args, (long)(it / nnb) * WT, (it % nnb) * WC, lane, q.off_t, q.off_u); }
    int cur = 0;
    wino8_stage(q, bufs, bufs + 2 * RAW_F, 0);

    for (; w < nwork; w += gridDim.x) {
        const int item = item_of(w);
        const int blk = item / nnb, nb = item - blk * nnb;
        const long t0 = (long)blk * WT;
        const int n0 = nb * WC;
        if (tid < 256) {
            const int tl = tid & 63, pl = tid >> 6;      // 64 tiles x 4 pixels = 256 entries
            const long t = t0 + tl;
            int ra = -1, ry = -1;
            if (t < args.ntiles) {
                const int per = args.TH * args.TW;
                const int b = (int)(t / per);
                const int rr = (int)(t - (long)b * per);
                const int ti = rr / args.TW, tj = rr - ti * args.TW;
                const int hh = 2 * ti + 1 + (pl >> 1), ww = 2 * tj + 1 + (pl & 1);
                if (ww <= g.Wd && hh <= g.H) {
                    ra = (int)((long)b * g.HPWP + (long)hh * g.WP + ww);
                    ry = g.y_unpadded ? ((b * g.H + hh - 1) * g.Wd + ww - 1) : ra;
                }
            }
            const int m = (tl >> 5) * 128 + pl * 32 + (tl & 31);
            rowa[m] = ra; rowy[m] = ry;
            if (pl == 0) {
                int pr = -1;
                if (args.pool_y && t < args.ntiles) {
                    const int per = args.TH * args.TW;
                    const int b = (int)(t / per);
                    const int rr = (int)(t - (long)b * per);
                    const int ti = rr / args.TW, tj = rr - ti * args.TW;
                    if (tj < args.W2 && ti < args.H2) pr = (b * (args.H2 + 1) + ti + 1) * (args.W2 + 1) + tj + 1;
                }
                prow[tl] = pr;
            }
        }
        bool zero_c3 = false;
        if (args.wodd) {
            long t = t0 + wm * 32 + li;
            if (t > args.ntiles - 1) t = args.ntiles - 1;
            zero_c3 = (int)(t % args.TW) == args.TW - 1;
        }
        // fused pool: this lane's channel constants, requested now so that the tail does not wait for them
        const int pool_n = n0 + wn * 32 + li;
        float pool_bs = 0.f, pool_sc = 1.f, pool_sh = 0.f;
        if (args.pool_y && pool_n < g.N) {
            if (g.bias) pool_bs = g.bias[pool_n];
            if (g.scale) pool_sc = g.scale[pool_n];
            if (g.shift) pool_sh = g.shift[pool_n];
        }
        floatx16 acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

        __builtin_amdgcn_s_waitcnt(0x0F70);          // chunk 0 of this item (requested before the previous epilogue) has landed
        __syncthreads();
        // the last chunk of an item requests chunk 0 of the NEXT item instead of a next chunk (same places between the MFMAs):
        // after it one set is being filled for the next item, the other is free for the exchange and the epilogue
        const int wnext = w + gridDim.x;
        unsigned otn = q.off_t, oun = q.off_u;
        if (wnext < nwork) { const int it = item_of(wnext); wino8_offsets(args, (long)(it / nnb) * WT, (it % nnb) * WC, lane, otn, oun); }
        for (int kc = 0; kc < nkc; ++kc) {
            const bool last = kc + 1 == nkc;
            if (last) { q.off_t = otn; q.off_u = oun; }
            wino8_chunk<XH>(bufs + cur * RAW_F, bufs + 2 * RAW_F + cur * U_F, bufs + (cur ^ 1) * RAW_F, bufs + 2 * RAW_F + (cur ^ 1) * U_F,
                            last ? wnext < nwork : true, q, last ? 0 : kc + 1, aoff, boff, args.wodd != 0, zero_c3, acc);
            if (!last) { __builtin_amdgcn_s_waitcnt(0x0F70); __syncthreads(); }
            else lds_barrier();
            cur ^= 1;
        }
        wino_item_tail<XH>(args, acc, bufs + (cur ^ 1) * RAW_F, bufs + 2 * RAW_F + (cur ^ 1) * U_F, rowa, rowy, prow, wave, lane, wm, wn, n0,
                           blk, pool_bs, pool_sc, pool_sh);
        lds_barrier();                               // row tables and buffer set `cur ^ 1` are reused by the next item
    }
}

// ------------------------------------------------------------------------------------------------ column-blocked tile order
// The tile columns of an image are cut into column blocks of w = 11..15 columns and the tiles are numbered row-major WITHIN a block,
// so that the 64 consecutive tiles of an item cover <= 7 tile rows x w columns: a region of <= 16 pixel rows x (2 w + 2) pixel
// columns holds every patch of the item once (round 3: wino9_kernel / wino10_kernel, superseded by wino11_kernel in round 4, which
// keeps their raw-region fetch):
//   * a DMA piece is one PAIR of pixel rows and one column parity -- <= 16 pixels (columns 2 idx + par) x 2 rows x two 16-byte channel
//     quads: [row 2][quad 2][position 16] 16-byte slots;
//   * the pixel idx of row pair m sits at position (idx + rho(m)) & 15, rho(m) = ((tr0 + m) w) & 15 -- so that the ds_read_b128 of
//     patch pixel (r, c) by the 32 lanes of a half-wave (tile l of the block) lands on position (l + (r >> 1) w + (c >> 1)) & 15:
//     consecutive tiles on consecutive positions, whatever w is -- conflict-free on the instruction's 16-lane service groups;
//   * columns past the plane's width and rows past the image are sent out of the buffer's range and read zeros.
// Items do not span column blocks or images: the last item of a block may be partly empty (1-3 % of the tiles).
#if __HIP_DEVICE_COMPILE__
struct WinoGeo { int b, tj0, w, l0, tr0, npieces; float invw; };

__device__ __forceinline__ int wino_fdiv(int x, float inv) { return (int)(((float)x + 0.5f) * inv); }      // floor(x / d) for 0 <= x < 2^22

__device__ __forceinline__ WinoGeo wino_geo(const WinoArgs& args, int blk) {
    WinoGeo e;
    const int ipi = args.cb_it0[args.ncb];
    e.b = wino_fdiv(blk, args.inv_ipi);
    const int rem = blk - e.b * ipi;
    int cb = 0;
#pragma unroll
    for (int c = 1; c < 8; ++c) cb += (c < args.ncb && rem >= args.cb_it0[c]) ? 1 : 0;      // blocks are in ascending item order
    e.tj0 = args.cb_tj0[cb]; e.w = args.cb_w[cb]; e.invw = args.cb_invw[cb];
    e.l0 = (rem - args.cb_it0[cb]) * WT;
    e.tr0 = wino_fdiv(e.l0, e.invw);
    const int rows = wino_fdiv(e.l0 + WT - 1, e.invw) - e.tr0 + 1;       // tile rows the item touches (<= 7)
    e.npieces = 2 * (2 * rows + 2);                           // pixel rows x column parities
    return e;
}
#endif

// ------------------------------------------------------------------------------------------------ wino11_kernel (round 4)
// The same computation with a wave owning ONE ROW of the 4 x 4 transform (4 positions, 64 accumulator registers): EIGHT waves of
// 128 registers per workgroup (wave = row * 2 + wm: 32 tiles x 32 channels x 4 positions each), 64 tiles x 32 output channels per
// item, two workgroups per CU -- FOUR waves per SIMD (round 3's wino10_kernel: the same item on four waves of 256 registers).  What that buys on this pipe, where a
// vector instruction of any wave of a SIMD costs matrix time:
//   * row r of V = B^T d B needs two pixel rows of the patch (0: rows 0, 2; 1 and 2: rows 1, 2; 3: rows 1, 3) and all four columns:
//     8 ds_read_b128 and 16 v_pk_add_f32 per 16 MFMAs (a wave with two transform columns: 12 reads and 40 packed adds per 32, i.e. 20 per 16);
//   * four instruction streams per SIMD: an LDS wait, a chunk barrier or an item tail of one wave leaves three others (two of
//     them of the other workgroup) to feed the matrix pipe;
//   * the item tail is light: the column combination (M A) is lane-local, the row combination (A^T .) is ONE exchange through LDS
//     in which wave r receives the register quarter 4 r .. 4 r + 3 of all four rows -- i.e. all four pixels of eight tiles -- so
//     the fused 2 x 2 pool is lane-local as well and every wave finishes ONE 32 x 32 block (pixel-major row table).
// Raw pieces and tile order: the section above.  The weights come from the SECOND layout asr_winograd_weights2
// writes ([K / 8][xi 16][N / 32][k-pair pair 2][lane half 2][co 32][2]): a weight piece is 1 KB of contiguous memory and a lane
// reads the two k-pairs of a position with one conflict-free ds_read_b64 at a 16-bit immediate of ONE base register.
// The chunk loop is unrolled by two so that both buffer sets are immediate offsets of per-lane base registers (no address
// arithmetic in the loop); an odd chunk count (K % 16 == 8) takes the variant with run-time set offsets.  Only the chunk loop is
// instantiated per transform row; set-up, tail and epilogues exist once.
// The argument block is read through the kernarg segment pointer, re-derived ("laundered") where a phase starts: hipcc otherwise
// keeps the ~100 argument dwords of the epilogues live in scalar registers across the chunk loop and spills them to vector-register
// lanes (round 3's wino10_kernel: 117 scalar spills, four waterfall loops per chunk around DMA pieces whose offsets ended up in
// vector registers).  A spill is worse than its instruction here: a scratch reload is a vector-memory load, and the s_waitcnt vmcnt(0) in
// front of its use also waits for every DMA piece and every epilogue store in flight.
// Which kernel runs is decided by the layer's widths and plane geometry alone -- never by the batch or the CU count -- so an
// utterance gives the same bits alone and inside a batch (tests/test_fullsize_gpu.py).
#ifdef ASR_DEV_HOOKS                 // development builds only (tools/trace_wino11.sh); the product build never defines it
#include "dev_hooks.h"
#endif
#ifndef W11T                         // phase stamps: nothing in the product
#define W11T(k) do { } while (0)
#define W11T_REALTIME(k) do { } while (0)
#define W11_TRACE_DUMP
#endif
constexpr int W11_C = 32;                       // output channels per item
constexpr int W11_SETF = 4224;                  // floats of a raw / weight set: 16 pieces of 256, padded so that four 32 x 33 transpose scratches fit
constexpr int W11_TABF = 768;                   // ints of a table set: rowa 256 | rowy 256 | prow 256
constexpr int W11_RAW0 = 2 * W11_TABF * 4;      // byte offsets of the four sets in the dynamic LDS segment
constexpr int W11_U0 = W11_RAW0 + 2 * W11_SETF * 4;
// epilogue instantiations (see wino11_body)
constexpr int W11_EPI_GENERIC = 0, W11_EPI_FWD = 1, W11_EPI_POOLMAX = 2, W11_EPI_POOLAVG = 3, W11_EPI_DGRAD = 4, W11_EPI_DGRAD_ACC = 5,
              W11_EPI_GATE1 = 6, W11_EPI_GATE2 = 7, W11_EPI_GATE3 = 8, W11_EPI_POOLMAXC = 9, W11_EPI_GATE4 = 10, W11_EPI_FWD_SUM = 11, W11_EPI_DGRAD_SESUM = 12, W11_EPI_POOLAVGC = 13, W11_EPI_GATE5 = 14;

#if __HIP_DEVICE_COMPILE__
typedef const __attribute__((address_space(4))) WinoArgs* wino_kernarg_p;
// the launch's argument block, through a pointer the optimiser cannot connect with earlier loads (s_load_dword on demand)
__device__ __forceinline__ const WinoArgs& wino_args_fresh() {
    wino_kernarg_p p = (wino_kernarg_p)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return *(const WinoArgs*)p;
}

struct W11Dma {
    unsigned voff[2];        // raw pieces wave + 8 j (per lane)
    unsigned l16;            // lane * 16: a weight piece is contiguous
    unsigned sbase;          // byte offset of the region's first pixel (channel 0)                    (scalar)
    unsigned pairbytes;      // bytes of two pixel rows of the plane                                   (scalar)
    unsigned ubytes;         // byte offset of chunk 0 of this item's channel block in the weights      (scalar)
    unsigned xibytes;        // bytes between two positions of a chunk: (N / 32) KB                     (scalar)
    unsigned chunkbytes;     // bytes of a chunk of the weights: 16 positions                           (scalar)
    int wave;
};

// per-lane offsets of the raw pieces this wave issues (piece p = wave + 8 j: row pair p >> 1 of the region, column parity p & 1;
// lane = [row of the pair][channel quad][position]); positions past the block / plane and rows past the image read zeros
__device__ __forceinline__ void wino11_offsets(const WinoArgs& args, const WinoGeo& e, int lane, W11Dma& q) {
    const TapGemmArgs& g = args.g;
    const int np = e.npieces >> 1;                              // (rows + 1) row pairs x 2 parities
    const int rowbit = lane >> 5, quad = (lane >> 4) & 1;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int p = q.wave + 8 * j, m = p >> 1, par = p & 1;
        const int y = 2 * m + rowbit;
        const int rho = ((e.tr0 + m) * e.w) & 15;
        const int idx = ((lane & 15) - rho) & 15;
        const int x = 2 * idx + par;
        const bool ok = idx <= e.w && 2 * e.tj0 + x < g.WP && 2 * e.tr0 + y <= g.H + 1 && p < np;
        q.voff[j] = ok ? (unsigned)(((rowbit * g.WP + x) * g.lda + quad * 4) * 4) : 0xFFFFFFF0u;
    }
    q.sbase = (unsigned)((((long)e.b * g.HPWP + (long)(2 * e.tr0) * g.WP + 2 * e.tj0) * g.lda) * 4);
}
template <class R>
__device__ __forceinline__ void wino11_raw_piece(const W11Dma& q, R ra, int lds_set, int j, int kc) {
    const int p = q.wave + 8 * j;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (wn_lds_f*)(uintptr_t)(unsigned)(lds_set + p * 1024), 16, q.voff[j],
                                             (int)(q.sbase + (unsigned)(p >> 1) * q.pairbytes + (unsigned)kc * (WKC * 4)), 0, 0);
}
// weight piece xi = wave + 8 j of chunk kc: [k-pair pair 2][lane half 2][32 co][2], contiguous in memory
template <class R>
__device__ __forceinline__ void wino11_u_piece(const W11Dma& q, R ru, int lds_set, int j, int kc) {
    const int xi = q.wave + 8 * j;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(ru, (wn_lds_f*)(uintptr_t)(unsigned)(lds_set + xi * 1024), 16, q.l16,
                                             (int)(q.ubytes + (unsigned)kc * q.chunkbytes + (unsigned)xi * q.xibytes), 0, 0);
}

// One 8-channel chunk of transform row RR out of buffer set `cur` (CUR 0 / 1: compile-time, all LDS addresses are immediates of the
// per-lane bases lb / ubase; CUR 2: run-time `cur`): 8 patch reads, 8 weight reads (two k-pairs each; the second four requested
// behind the first transform: 8 registers at a time), 16 packed adds, 16 MFMAs (k-pair major); behind the four MFMAs of a k-pair
// one DMA piece of the next chunk into the other set (weights first: they are read first).  FIRST: the accumulators start from
// the MFMA's zero operand (no 64 register moves per item).
template <int RR, int CUR, bool FIRST, class R>
__device__ __forceinline__ void wino11_chunk(const char* __restrict__ lds, int lds0, int cur, bool pre, const W11Dma& q, R ra, R ru, int kcn,
                                             const unsigned (&lb)[4], unsigned ubase, floatx16 (&acc)[4]) {
    constexpr int RA = RR == 0 ? 0 : 1, RB = RR == 3 ? 3 : 2;           // the two patch rows this transform row combines
    const int setoff = (CUR == 2 ? cur : CUR) * (W11_SETF * 4);
    const int nxtoff = (CUR == 2 ? cur ^ 1 : CUR ^ 1) * (W11_SETF * 4);
    float4 da[4], db[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        da[c] = *(const float4*)(lds + lb[(RA >> 1) * 2 + (c >> 1)] + setoff + (RA >> 1) * 2048 + (c & 1) * 1024 + (RA & 1) * 512);
        db[c] = *(const float4*)(lds + lb[(RB >> 1) * 2 + (c >> 1)] + setoff + (RB >> 1) * 2048 + (c & 1) * 1024 + (RB & 1) * 512);
    }
    wn_f2 u[4][2];
    auto load_u = [&](int kpp) {
#pragma unroll
        for (int c = 0; c < 4; ++c) u[c][kpp] = *(const wn_f2*)(lds + ubase + setoff + (RR * 4 + c) * 1024 + kpp * 512);
    };
    load_u(0);
    __builtin_amdgcn_sched_barrier(0);
    // stage 1, both k-pair halves at once: t = row RR of B^T d (the sixteen patch registers of the second row are free afterwards --
    // the chunk's peak is 40 registers beside the 64 accumulators); stage 2 per half: V = t B
    wn_f2 t[4][2];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const wn_f2 a0 = {da[c].x, da[c].y}, a1 = {da[c].z, da[c].w}, b0 = {db[c].x, db[c].y}, b1 = {db[c].z, db[c].w};
        t[c][0] = RR == 1 ? a0 + b0 : RR == 2 ? b0 - a0 : a0 - b0;
        t[c][1] = RR == 1 ? a1 + b1 : RR == 2 ? b1 - a1 : a1 - b1;
    }
    wn_f2 v[4];
    auto transform = [&](int h) { v[0] = t[0][h] - t[2][h]; v[1] = t[1][h] + t[2][h]; v[2] = t[2][h] - t[1][h]; v[3] = t[1][h] - t[3][h]; };
    auto mfmas = [&](int kp, int piece) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const floatx16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32((kp & 1) ? v[c].y : v[c].x, (kp & 1) ? u[c][kp >> 1].y : u[c][kp >> 1].x,
                                                          (FIRST && kp == 0) ? zero : acc[c], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (pre) {
            if (piece < 2) wino11_u_piece(q, ru, lds0 + W11_U0 + nxtoff, piece, kcn);
            else wino11_raw_piece(q, ra, lds0 + W11_RAW0 + nxtoff, piece - 2, kcn);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    transform(0);
    __builtin_amdgcn_sched_barrier(0);
    load_u(1);
    __builtin_amdgcn_sched_barrier(0);
    mfmas(0, 0);
    mfmas(1, 1);
    transform(1);
    __builtin_amdgcn_sched_barrier(0);
    mfmas(2, 2);
    mfmas(3, 3);
}

// the chunk loop of one item for transform row RR -- the only code instantiated per row.  The last chunk requests nothing: the
// next item's first chunk is requested at the top of the item tail (shared code), where its offsets are computed.
template <int RR, class R>
__device__ __forceinline__ void wino11_item_chunks(const char* lds, int lds0, int nkc, int& cur, const W11Dma& q, R ra, R ru,
                                                   const unsigned (&lb)[4], unsigned ubase, floatx16 (&acc)[4]) {
    if ((nkc & 1) == 0 && cur == 0) {
        wino11_chunk<RR, 0, true>(lds, lds0, 0, true, q, ra, ru, 1, lb, ubase, acc);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        for (int kc = 2; kc < nkc; kc += 2) {
            wino11_chunk<RR, 1, false>(lds, lds0, 1, true, q, ra, ru, kc, lb, ubase, acc);
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            wino11_chunk<RR, 0, false>(lds, lds0, 0, true, q, ra, ru, kc + 1, lb, ubase, acc);
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        wino11_chunk<RR, 1, false>(lds, lds0, 1, false, q, ra, ru, 0, lb, ubase, acc);
        lds_barrier();
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        for (int kc = 0; kc < nkc; ++kc) {
            const bool last = kc + 1 == nkc;
            wino11_chunk<RR, 2, false>(lds, lds0, cur, !last, q, ra, ru, kc + 1, lb, ubase, acc);
            if (last) lds_barrier();
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            cur ^= 1;
        }
    }
}

// row tables of one item (threads 0..255: 64 tiles x 4 pixels).  Wave (row r, half wm) finishes tiles wm * 32 + 8 r .. + 7 with
// all four pixels: its 32 x 32 epilogue block has row p * 8 + t' = pixel p of tile t' (the MFMA register q = 4 p + i holds tile
// i + 4 lh, the epilogues' register-to-row rule).  prow: rows 0..7 of a wave's block = the pooled pixel of its tiles, else -1.
__device__ __forceinline__ void wino11_tables(const WinoArgs& args, const WinoGeo& e, int tid, int* rowa, int* rowy, int* prow) {
    const TapGemmArgs& g = args.g;
    if (tid < 256) {
        const int tl = tid & 63, pl = tid >> 6;
        const int l = e.l0 + tl;
        const int ti = wino_fdiv(l, e.invw), tj = e.tj0 + l - ti * e.w;
        int ra_ = -1, ry = -1;
        if (ti < args.TH) {
            const int hh = 2 * ti + 1 + (pl >> 1), ww = 2 * tj + 1 + (pl & 1);
            if (ww <= g.Wd && hh <= g.H) {
                ra_ = (int)((long)e.b * g.HPWP + (long)hh * g.WP + ww);
                ry = g.y_unpadded ? ((e.b * g.H + hh - 1) * g.Wd + ww - 1) : ra_;
            }
        }
        // gated launches with a pooled cell in front: rowy = the full-resolution row of window position 0 of this output pixel
        if (g.gate_mode >= 2 && ra_ >= 0) {
            const int hh = 2 * ti + 1 + (pl >> 1), ww = 2 * tj + 1 + (pl & 1);
            ry = (e.b * (g.gate_H + 1) + 2 * hh - 1) * (g.gate_W + 1) + 2 * ww - 1;
        }
        const int wv = (tl >> 5) + 2 * ((tl & 31) >> 3);             // wave = row * 2 + wm
        const int m = wv * 32 + pl * 8 + (tl & 7);
        rowa[m] = ra_; rowy[m] = ry;
        int pr = -1;
        if (pl == 0 && args.pool_y && ti < args.TH && tj < args.W2 && ti < args.H2) pr = (e.b * (args.H2 + 1) + ti + 1) * (args.W2 + 1) + tj + 1;
        prow[m] = pr;                                                // rows 8..31 of a block (pl > 0): -1
    }
}

// ---- epilogues of one 32 x 32 block per wave.  The launcher picks the kernel instantiation EPI from the descriptor, so that what
// an epilogue does is known at compile time: with the options as run-time flags (tap_epilogue: ReLU / tanh, out_a, out_y, accumulate,
// pool mode, gate mode, each tested per ROW inside the unrolled loops) the item tail was a chain of ~100 scalar branches, and a taken
// branch costs a wave its instruction buffer -- in-kernel stamps: 50-60 cycles per tail instruction, tails as long as the whole
// chunk loop at K = 32.
//   EPI 0  generic: tap_epilogue on the run-time flags (tanh, unusual output combinations)
//   EPI 1  forward, ReLU, out_a and out_y            EPI 2 / 3  forward of a pooled cell: ReLU, out_a, fused max / average pool
//   EPI 4  data-gradient: out_y (no bias, no affine)  EPI 5  ... accumulating
//   EPI 6 / 7 / 8  gated data-gradient (fused backward prologue of the cell in front: no pool / average / max), accumulate at run time
typedef unsigned int w11_u4 __attribute__((ext_vector_type(4)));
constexpr unsigned W11_OOR = 0xFFFFFFF0u;        // a buffer offset past num_records: the load returns zeros, the store is dropped

// Output / gate planes go through buffer resources with 32-bit byte offsets (wino_impl refuses planes of 4 GiB or 2^24 rows and
// more): an address is one 24-bit multiply-add, and a row outside the plane is an out-of-range offset instead of a branch.
template <class R>
__device__ __forceinline__ float4 w11_load4(R r, unsigned off) {
    const w11_u4 v = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
template <class R>
__device__ __forceinline__ void w11_store4_nt(R r, unsigned off, float x, float y, float z, float w) {
    __builtin_amdgcn_raw_buffer_store_b128(w11_u4{__float_as_uint(x), __float_as_uint(y), __float_as_uint(z), __float_as_uint(w)}, r, off, 0, 2);
}

// the block through the per-wave 32 x 33 transpose scratch: afterwards lane (rsub = lane >> 3, c4 = lane & 7) reads row it * 8 + rsub,
// channels 4 c4 .. 4 c4 + 3 -- tile rsub of the wave, pixel it (wino11_tables)
__device__ __forceinline__ void wino11_transpose(const floatx16& blk, float* scratch, int lane) {
    const int li = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int r = 0; r < 16; ++r) scratch[((r & 3) + 8 * (r >> 2) + 4 * lh) * 33 + li] = blk[r];
}

// plain epilogue: v = act(x + bias) -> out_a; y = scale v + shift (+ previous) -> out_y.  The lane's channel constants are handed in:
// requested at the top of the item tail, they arrive behind the exchange.  pa / py: bytes per row of the two outputs, n4: byte
// offset of the lane's channel quad (W11_OOR beyond N).
// SUMS (round 5, asr_tap_gemm_wino_sums): the block's per-channel sums of y over its rows inside the plane go to ysum_row[channel] --
// one partial row per (tile block, wave), which is one image's: the squeeze of a squeeze-excitation block whose branch this cell is
// (acoustic_model2.py:135-148 Global_Average_Pooling) without a pass of its own over the plane.
// SUMS 2 (asr_tap_gemm_wino_sesum): the data-gradient that completes dL/d(output of an SE block) also leaves, per (tile block, wave),
// sum over its rows of y * (sc * x + sh) -- x the block's branch plane (rX, the geometry of y), sc / sh the block's BN affine: the
// reduction the block's backward needs for its excitation gradient (se_reduce_kernel<1>), without a pass over both planes.
template <bool RELU, bool HAS_A, bool HAS_Y, bool ACC, int SUMS = 0, class R>
__device__ __forceinline__ void wino11_epilogue_plain(R rA, R rY, unsigned pa, unsigned py, const floatx16& blk, float* scratch, const int* rowa,
                                                      const int* rowy, unsigned n4, const float4& bs, const float4& sc, const float4& sh, int lane,
                                                      float* ysum_row = nullptr, R rX = R()) {
    const int c4 = lane & 7, rsub = lane >> 3;
    wino11_transpose(blk, scratch, lane);
    unsigned oa[4], oy[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int ra = rowa[it * 8 + rsub];
        const bool ok = ra >= 0 && n4 != W11_OOR;
        oa[it] = ok ? __umul24((unsigned)ra, pa) + n4 : W11_OOR;
        if (HAS_Y) { const int ry = rowy[it * 8 + rsub]; oy[it] = ok ? __umul24((unsigned)ry, py) + n4 : W11_OOR; }
    }
    float4 prev[4], xv[4];
    float4 ys = make_float4(0.f, 0.f, 0.f, 0.f);
    if (HAS_Y && ACC) {
#pragma unroll
        for (int it = 0; it < 4; ++it) prev[it] = w11_load4(rY, oy[it]);
    }
    if (SUMS == 2) {
#pragma unroll
        for (int it = 0; it < 4; ++it) xv[it] = w11_load4(rX, oy[it]);         // (rows outside the plane read zeros and are masked below)
    }
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const float* sp = scratch + (it * 8 + rsub) * 33 + c4 * 4;
        float4 v = make_float4(sp[0] + bs.x, sp[1] + bs.y, sp[2] + bs.z, sp[3] + bs.w);
        if (RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        if (HAS_A) w11_store4_nt(rA, oa[it], v.x, v.y, v.z, v.w);
        if (HAS_Y) {
            // (SUMS 2: a data-gradient -- no affine of its own, sc / sh are the SE block's)
            float4 y = SUMS == 2 ? v : make_float4(sc.x * v.x + sh.x, sc.y * v.y + sh.y, sc.z * v.z + sh.z, sc.w * v.w + sh.w);
            if (ACC) { y.x += prev[it].x; y.y += prev[it].y; y.z += prev[it].z; y.w += prev[it].w; }
            w11_store4_nt(rY, oy[it], y.x, y.y, y.z, y.w);
            if (SUMS == 1) {
                const bool in = oy[it] != W11_OOR;
                ys.x += in ? y.x : 0.f; ys.y += in ? y.y : 0.f; ys.z += in ? y.z : 0.f; ys.w += in ? y.w : 0.f;
            }
            if (SUMS == 2) {
                const bool in = oy[it] != W11_OOR;
                ys.x = fmaf(in ? y.x : 0.f, fmaf(sc.x, xv[it].x, sh.x), ys.x); ys.y = fmaf(in ? y.y : 0.f, fmaf(sc.y, xv[it].y, sh.y), ys.y);
                ys.z = fmaf(in ? y.z : 0.f, fmaf(sc.z, xv[it].z, sh.z), ys.z); ys.w = fmaf(in ? y.w : 0.f, fmaf(sc.w, xv[it].w, sh.w), ys.w);
            }
        }
    }
    if (SUMS) {
        // rows of this wave: lanes with equal c4 differ in bits 3..5 of the lane id; fixed shuffle order
#pragma unroll
        for (int o = 8; o < 64; o <<= 1) {
            ys.x += __shfl_xor(ys.x, o, 64); ys.y += __shfl_xor(ys.y, o, 64); ys.z += __shfl_xor(ys.z, o, 64); ys.w += __shfl_xor(ys.w, o, 64);
        }
        if (rsub == 0 && n4 != W11_OOR) *(float4*)(ysum_row + (n4 >> 2)) = ys;
    }
}

// gated epilogue (GM 1 no pool, 2 average, 3 max): tap_epilogue_gated's arithmetic in its order -- per output pixel and channel quad
// g_k = the gradient routed to pre-pool position k, dZ_k = g_k scale where a_k > 0, the three channel sums -- with every load of a
// pixel pair requested before its arithmetic (the shared form waits for four dependent round trips per block) and the full-resolution
// row of window position 0 taken from the row table (rowf = rowy: wino11_tables writes it for the gated launches, no divisions here).
// rGA / rGD: gate activations and dZ (C channels per pixel), rY: dL/dy of the cell in front (accumulate), py its row bytes.
// GM 4 = the maximum pool in its compact form (asr_tap_gemm_gated_poolmax): rGA is the plane of the activations AT the window maxima
// (pooled geometry = this launch's output pixels), rGI the two bit planes with the position of each maximum; one float4 + 8 bytes per
// pixel and channel quad instead of four float4.  Every sum receives the terms GM 3 gives it, in the same order (the three positions
// that are not the maximum contribute exact zeros there too): the same bits.
template <int GM, class R>
__device__ __forceinline__ void wino11_epilogue_gated(R rGA, R rGD, R rY, R rGI, bool accumulate, unsigned py, unsigned c4b, unsigned wpf4, float* gate_part, int C,
                                                      const floatx16& blk, float* scratch, const int* rowa, const int* rowf,
                                                      int n, unsigned n4, const float4& sc, const float4& sh, int lane, int part_row) {
    const int c4 = lane & 7, rsub = lane >> 3;
    wino11_transpose(blk, scratch, lane);
    float s_scale[4] = {0.f, 0.f, 0.f, 0.f}, s_shift[4] = {0.f, 0.f, 0.f, 0.f}, s_bias[4] = {0.f, 0.f, 0.f, 0.f};
    const float scv[4] = {sc.x, sc.y, sc.z, sc.w}, shv[4] = {sh.x, sh.y, sh.z, sh.w};
    // GM 5 = the AVERAGE pool in its compact form (asr_tap_gemm_gated_poolavg, round 5): rGA is the plane of the window SUMS of the
    // activations (pooled geometry), rGI four bit planes with the ReLU sign of each window position.  dZ and the bias / shift sums are those of
    // GM 2 bit for bit (a position's gradient is 0.25 dy * scale where its activation was positive); the scale sum takes 0.25 dy times the
    // window's activation sum in ONE multiply-add instead of four (the same value, rounded once).
    constexpr int NK = (GM == 1 || GM == 4 || GM == 5) ? 1 : 4;
    // One pixel per trip of a ROLLED loop: unrolled, hipcc interleaves the four pixels (and vectorises across them), holds 100 values
    // more than the 128-register budget has and parks them in scratch -- whose reloads wait for every store in flight.  The other
    // three waves of the SIMD cover the round trip of a pixel's loads.
#pragma nounroll
    for (int p = 0; p < 4; ++p) {
        const int row = p * 8 + rsub;
        const int ra = rowa[row];
        const bool ok = ra >= 0 && n4 != W11_OOR;
        const unsigned pf = GM == 1 ? (unsigned)ra : (unsigned)rowf[row];
        const unsigned o0 = ok ? __umul24(pf, c4b) + n4 : W11_OOR;          // c4b: bytes of a gate pixel (4 C)
        unsigned off[GM == 1 ? 1 : 4];
        off[0] = o0;
        if (GM != 1) { off[1] = ok ? o0 + c4b : W11_OOR; off[2] = ok ? o0 + wpf4 : W11_OOR; off[3] = ok ? o0 + wpf4 + c4b : W11_OOR; }
        float4 a4[NK];
        unsigned iw0 = 0, iw1 = 0;
        w11_u4 sg = {0u, 0u, 0u, 0u};
        if (GM == 4) {
            a4[0] = w11_load4(rGA, ok ? __umul24((unsigned)ra, c4b) + n4 : W11_OOR);
            const unsigned io = ok ? __umul24((unsigned)ra, (unsigned)(C >> 5) * 8u) + (unsigned)(n >> 5) * 8u : W11_OOR;
            iw0 = __builtin_amdgcn_raw_buffer_load_b32(rGI, io, 0, 0);
            iw1 = __builtin_amdgcn_raw_buffer_load_b32(rGI, io == W11_OOR ? W11_OOR : io + 4u, 0, 0);
        } else if (GM == 5) {
            a4[0] = w11_load4(rGA, ok ? __umul24((unsigned)ra, c4b) + n4 : W11_OOR);
            sg = __builtin_amdgcn_raw_buffer_load_b128(rGI, ok ? __umul24((unsigned)ra, (unsigned)(C >> 5) * 16u) + (unsigned)(n >> 5) * 16u : W11_OOR, 0, 0);
        } else {
#pragma unroll
            for (int k = 0; k < NK; ++k) a4[k] = w11_load4(rGA, off[k]);
        }
        const float4 prev = accumulate ? w11_load4(rY, ok ? __umul24((unsigned)ra, py) + n4 : W11_OOR) : make_float4(0.f, 0.f, 0.f, 0.f);
        const float* sp = scratch + row * 33 + c4 * 4;
        // a row outside the plane: its activations read as zeros and its value is dropped (v = 0: no contribution to the sums)
        const float r0 = sp[0], r1 = sp[1], r2 = sp[2], r3 = sp[3];          // (read unconditionally: a load under a select becomes a branch)
        const float v[4] = {ok ? r0 + prev.x : 0.f, ok ? r1 + prev.y : 0.f, ok ? r2 + prev.z : 0.f, ok ? r3 + prev.w : 0.f};
        float av[NK][4];
#pragma unroll
        for (int k = 0; k < NK; ++k) { av[k][0] = a4[k].x; av[k][1] = a4[k].y; av[k][2] = a4[k].z; av[k][3] = a4[k].w; }
        if (GM == 1) {
            float d[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                s_shift[e] += v[e];
                s_scale[e] = fmaf(v[e], av[0][e], s_scale[e]);
                d[e] = av[0][e] > 0.f ? v[e] * scv[e] : 0.f;
                s_bias[e] += d[e];
            }
            w11_store4_nt(rGD, off[0], d[0], d[1], d[2], d[3]);
        } else if (GM == 5) {
            const int cb = n & 31;
            const unsigned sgw[4] = {sg.x, sg.y, sg.z, sg.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) s_scale[e] = fmaf(0.25f * v[e], av[0][e], s_scale[e]);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float d[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float gk = 0.25f * v[e];
                    s_shift[e] += gk;
                    d[e] = ((sgw[k] >> (cb + e)) & 1u) ? gk * scv[e] : 0.f;
                    s_bias[e] += d[e];
                }
                w11_store4_nt(rGD, off[k], d[0], d[1], d[2], d[3]);
            }
        } else if (GM == 4) {
            const int cb = n & 31;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float d[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int ag = (int)((iw0 >> (cb + e)) & 1u) | (int)(((iw1 >> (cb + e)) & 1u) << 1);
                    const bool here = ag == k;
                    const float gk = here ? v[e] : 0.f;
                    s_shift[e] += gk;
                    s_scale[e] = fmaf(gk, here ? av[0][e] : 0.f, s_scale[e]);
                    d[e] = (here && av[0][e] > 0.f) ? gk * scv[e] : 0.f;
                    s_bias[e] += d[e];
                }
                w11_store4_nt(rGD, off[k], d[0], d[1], d[2], d[3]);
            }
        } else {
            int arg[4];
            if (GM == 3) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {          // first maximum of scale a + shift in row-major window order (gate_pick)
                    const float y0 = fmaf(scv[e], av[0][e], shv[e]), y1 = fmaf(scv[e], av[1][e], shv[e]);
                    const float y2 = fmaf(scv[e], av[2][e], shv[e]), y3 = fmaf(scv[e], av[3][e], shv[e]);
                    int ag = 0; float m = y0;
                    if (y1 > m) { m = y1; ag = 1; }
                    if (y2 > m) { m = y2; ag = 2; }
                    if (y3 > m) { m = y3; ag = 3; }
                    arg[e] = ag;
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float d[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float gk = GM == 2 ? 0.25f * v[e] : (arg[e] == k ? v[e] : 0.f);
                    s_shift[e] += gk;
                    s_scale[e] = fmaf(gk, av[k][e], s_scale[e]);
                    d[e] = av[k][e] > 0.f ? gk * scv[e] : 0.f;
                    s_bias[e] += d[e];
                }
                w11_store4_nt(rGD, off[k], d[0], d[1], d[2], d[3]);
            }
        }
    }
    // rows of this wave: lanes with equal c4 differ in bits 3..5 of the lane id; fixed shuffle order
#pragma unroll
    for (int e = 0; e < 4; ++e) {
#pragma unroll
        for (int o = 8; o < 64; o <<= 1) {
            s_scale[e] += __shfl_xor(s_scale[e], o, 64);
            s_shift[e] += __shfl_xor(s_shift[e], o, 64);
            s_bias[e] += __shfl_xor(s_bias[e], o, 64);
        }
    }
    if (rsub == 0 && n4 != W11_OOR) {
        float* pr = gate_part + (long)part_row * 3 * C + n;
        *(float4*)(pr) = make_float4(s_scale[0], s_scale[1], s_scale[2], s_scale[3]);
        *(float4*)(pr + C) = make_float4(s_shift[0], s_shift[1], s_shift[2], s_shift[3]);
        *(float4*)(pr + 2 * C) = make_float4(s_bias[0], s_bias[1], s_bias[2], s_bias[3]);
    }
}

template <int EPI>
__device__ __forceinline__ void wino11_body(float* smem) {
    constexpr bool GATED = (EPI >= W11_EPI_GATE1 && EPI <= W11_EPI_GATE3) || EPI == W11_EPI_GATE4 || EPI == W11_EPI_GATE5;
    constexpr bool POOLED = EPI == W11_EPI_POOLMAX || EPI == W11_EPI_POOLAVG || EPI == W11_EPI_POOLMAXC || EPI == W11_EPI_POOLAVGC;
    const char* lds = (const char*)smem;
    int* tables = (int*)smem;                        // two sets of [rowa 256 | rowy 256 | prow 256]
    float* bufs = smem + 2 * W11_TABF;               // raw0 | raw1 | u0 | u1
    float* pconst = bufs + 4 * W11_SETF;             // fused pool: [bias N | scale N | shift N]
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, lh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);           // = row * 2 + wm
    const int wm = wave & 1, rr = wave >> 1;
    unsigned ubase = (unsigned)(W11_U0 + lh * 256 + li * 8);             // [k-pair pair][lane half][co][2]: half 1 contracts channels 4 .. 7
    asm volatile("" : "+v"(ubase));                  // opaque: the set and position offsets stay 16-bit immediates of this base
    const int lds0 = (int)(unsigned)(uintptr_t)(wn_lds_f*)smem;          // LDS address of the dynamic segment (DMA destinations are absolute)
    int nkc, nnb;
    W11Dma q;
    q.wave = wave;
    q.l16 = (unsigned)lane * 16u;
    // Work: the two workgroups of a CU (w and w + G / 2 under the observed round-robin placement: speed only) share a CONTIGUOUS range
    // of items (an item = tile block x 32-channel block, the channel blocks of a tile block adjacent) and split it in the middle:
    // the CUs are balanced to one item, and consecutive items of a workgroup mostly share the tile block, so that everything which
    // depends on it alone -- region geometry, the lanes' patch offsets, DMA offsets, the row tables -- is set up once per tile block.
    // The ranges of the CUs of one XCD (w % 8) are adjacent: neighbouring tile blocks go through one L2.
    int it, it_end;
    {
        const WinoArgs& args = wino_args_fresh();
        const TapGemmArgs& g = args.g;
        nkc = g.K / WKC; nnb = g.ntn;
        const long nwork = (long)g.ntm * nnb;
        const int G = gridDim.x, w = blockIdx.x;
        if ((G & 1) == 0) {
            const int P = G >> 1, p = w % P, half = w / P;
            const int r = (P & 7) ? p : (p & 7) * (P >> 3) + (p >> 3);
            const int lo = (int)(nwork * r / P), hi = (int)(nwork * (r + 1) / P), mid = lo + (hi - lo + 1) / 2;
            it = half ? mid : lo; it_end = half ? hi : mid;
        } else {
            it = (int)(nwork * w / G); it_end = (int)(nwork * (w + 1) / G);
        }
        if (it >= it_end) return;
        q.pairbytes = (unsigned)(2 * g.WP * g.lda * 4);
        q.xibytes = (unsigned)nnb * 1024u; q.chunkbytes = 16u * (unsigned)nnb * 1024u;
    }
    // the two buffer resources: plane and transformed weights (second layout: 16 K N floats into the buffer)
    const WinoArgs& args0 = wino_args_fresh();
    auto ra = __builtin_amdgcn_make_buffer_rsrc((void*)args0.g.A, 0, 0x7FFFFFF0, 0x00020000);
    auto ru = __builtin_amdgcn_make_buffer_rsrc((void*)(args0.Ut + (size_t)16 * args0.g.K * args0.g.N), 0, 0x7FFFFFF0, 0x00020000);
    int tcur = 0, cur = 0;
    unsigned lb[4];
    // per-lane LDS offsets of the patch pixels of a tile block (tile l of the block, its row relative to the region)
    auto patch_offsets = [&](const WinoGeo& e) {
        const int l = e.l0 + wm * 32 + li;
        const int trl = wino_fdiv(l, e.invw) - e.tr0;
        const unsigned fix = (unsigned)(W11_RAW0 + trl * 2048 + lh * 256);
#pragma unroll
        for (int j = 0; j < 4; ++j) { lb[j] = fix + (unsigned)(((l + (j >> 1) * e.w + (j & 1)) & 15) * 16); asm volatile("" : "+v"(lb[j])); }
    };
    {
        const WinoArgs& args = wino_args_fresh();
        const TapGemmArgs& g = args.g;
        const WinoGeo e = wino_geo(args, it / nnb);
        wino11_offsets(args, e, lane, q);
        patch_offsets(e);
        q.ubytes = (unsigned)(it % nnb) * 1024u;
#pragma unroll
        for (int j = 0; j < 2; ++j) { wino11_raw_piece(q, ra, lds0 + W11_RAW0, j, 0); wino11_u_piece(q, ru, lds0 + W11_U0, j, 0); }
        wino11_tables(args, e, tid, tables, tables + 256, tables + 512);
        if (args.pool_y)
            for (int n = tid; n < g.N; n += 512) {
                pconst[n] = g.bias ? g.bias[n] : 0.f;
                pconst[g.N + n] = g.scale ? g.scale[n] : 1.f;
                pconst[2 * g.N + n] = g.shift ? g.shift[n] : 0.f;
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lds_barrier();
    }
    int titem = -2;
    // (tile block, channel block) of the item, stepped along: a division by the run-time nnb at the top of every item was ~40
    // instructions that all eight waves executed in lock-step behind the tail's last barrier
    int blk = it / nnb, nb = it - blk * nnb;
    for (; it < it_end; ++it) {
        ++titem;
        W11T(0);
        W11T_REALTIME(12);
        const int n0 = nb * W11_C;
        const bool more = it + 1 < it_end;
        const bool newblk = more && nb + 1 == nnb;   // the next item starts a new tile block
        floatx16 acc[4];
        __builtin_amdgcn_s_setprio(0);
        switch (rr) {
            case 0: wino11_item_chunks<0>(lds, lds0, nkc, cur, q, ra, ru, lb, ubase, acc); break;
            case 1: wino11_item_chunks<1>(lds, lds0, nkc, cur, q, ra, ru, lb, ubase, acc); break;
            case 2: wino11_item_chunks<2>(lds, lds0, nkc, cur, q, ra, ru, lb, ubase, acc); break;
            default: wino11_item_chunks<3>(lds, lds0, nkc, cur, q, ra, ru, lb, ubase, acc); break;
        }
        W11T(1);
        // the tail above the chunk phase of the other workgroup's waves on this SIMD (s_setprio 2 here, 0 in the chunk loop): the tail is
        // the workgroup's critical path -- nothing of it overlaps its own matrix work -- while the other workgroup's MFMAs wait a few
        // instructions at most (plain DFCNN step -0.3 % on two boxes, the reverse assignment: no change)
        __builtin_amdgcn_s_setprio(2);
        // ---- item tail.  The set the last chunk read (cur ^ 1) is free: exchange area, then transpose scratch; the other one (cur)
        // receives chunk 0 of the next item, requested FIRST so that it travels behind the whole tail.
        // (The lane id is re-derived here: what the tail computes from it -- scratch and table addresses -- must not be hoisted out of
        // the item loop, where it would live across the chunk loop's full register budget and come back as scratch reloads, each with
        // an s_waitcnt vmcnt(0) that also waits for the stores in flight.)
        int lane_t = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        asm volatile("" : "+v"(lane_t));
        const int lane = lane_t, li = lane & 31, lh = lane >> 5;
        WinoGeo en;                                  // the next tile block (where the next item starts one)
        if (more) {
            if (newblk) {
                const WinoArgs& args = wino_args_fresh();
                en = wino_geo(args, blk + 1);
                wino11_offsets(args, en, lane, q);
                q.ubytes = 0;
            } else q.ubytes += 1024u;
#pragma unroll
            for (int j = 0; j < 2; ++j) { wino11_raw_piece(q, ra, lds0 + W11_RAW0 + cur * (W11_SETF * 4), j, 0); wino11_u_piece(q, ru, lds0 + W11_U0 + cur * (W11_SETF * 4), j, 0); }
        }
        // this lane's channel constants for the epilogue (channels n0 + 4 c4 ..): requested now, used behind the exchange
        const int ne = n0 + (lane & 7) * 4;
        float4 cbs = make_float4(0.f, 0.f, 0.f, 0.f), csc = make_float4(1.f, 1.f, 1.f, 1.f), csh = cbs;
        bool ncol;
        if (EPI != W11_EPI_GENERIC) {
            const WinoArgs& args = wino_args_fresh();
            const TapGemmArgs& g = args.g;
            ncol = ne < g.N;
            if (ncol) {
                if (EPI == W11_EPI_FWD || EPI == W11_EPI_FWD_SUM || POOLED) cbs = *(const float4*)(g.bias + ne);
                if (EPI == W11_EPI_FWD || EPI == W11_EPI_FWD_SUM || EPI == W11_EPI_DGRAD_SESUM || GATED) { csc = *(const float4*)(g.scale + ne); csh = *(const float4*)(g.shift + ne); }
            }
        }
        float* rfree = bufs + (cur ^ 1) * W11_SETF;
        float* ufree = bufs + (2 + (cur ^ 1)) * W11_SETF;
        float* xch = wm == 0 ? rfree : ufree;        // the four row waves of a tile half meet in one set (4096 floats per phase)
        floatx16 out[1][1];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            // column combination (M A)[row][j], lane-local, two registers per packed add
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const wn_f2 a0 = {acc[0][r], acc[0][r + 1]}, a1 = {acc[1][r], acc[1][r + 1]}, a2 = {acc[2][r], acc[2][r + 1]}, a3 = {acc[3][r], acc[3][r + 1]};
                const wn_f2 pj = j == 0 ? (a0 + a1) + a2 : (a1 - a2) - a3;
                xch[(rr * 16 + r) * 64 + lane] = pj.x;
                xch[(rr * 16 + r + 1) * 64 + lane] = pj.y;
            }
            if (j == 0) W11T(2);
            lds_barrier();
            if (j == 0) W11T(3);
            // row combination for this wave's register quarter: Y[i][j] = sum_r A^T[i][r] (M A)[r][j]
#pragma unroll
            for (int i = 0; i < 4; i += 2) {
                const wn_f2 p0 = {xch[(0 * 16 + 4 * rr + i) * 64 + lane], xch[(0 * 16 + 4 * rr + i + 1) * 64 + lane]};
                const wn_f2 p1 = {xch[(1 * 16 + 4 * rr + i) * 64 + lane], xch[(1 * 16 + 4 * rr + i + 1) * 64 + lane]};
                const wn_f2 p2 = {xch[(2 * 16 + 4 * rr + i) * 64 + lane], xch[(2 * 16 + 4 * rr + i + 1) * 64 + lane]};
                const wn_f2 p3 = {xch[(3 * 16 + 4 * rr + i) * 64 + lane], xch[(3 * 16 + 4 * rr + i + 1) * 64 + lane]};
                const wn_f2 y0 = (p0 + p1) + p2, y1 = (p1 - p2) - p3;
                out[0][0][4 * (0 * 2 + j) + i] = y0.x; out[0][0][4 * (0 * 2 + j) + i + 1] = y0.y;
                out[0][0][4 * (1 * 2 + j) + i] = y1.x; out[0][0][4 * (1 * 2 + j) + i + 1] = y1.y;
            }
            lds_barrier();                           // the set is rewritten (phase 1) / becomes transpose scratch
        }
        W11T(4);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the next item's chunk 0 and the channel constants (no store is in flight yet)
        W11T(5);
        const WinoArgs& args = wino_args_fresh();
        const TapGemmArgs& g = args.g;
        int* rowa = tables + tcur * W11_TABF;
        int* rowy = rowa + 256;
        int* prow = rowy + 256;
        if (newblk) {                                // the next tile block: its patch offsets and (second table set) its row tables
            patch_offsets(en);
            // (the thread index from the re-derived lane id: `tid` itself would live across the chunk loop and come back from scratch)
            wino11_tables(args, en, wave * 64 + lane, tables + (tcur ^ 1) * W11_TABF, tables + (tcur ^ 1) * W11_TABF + 256, tables + (tcur ^ 1) * W11_TABF + 512);
        }
        W11T(6);
        float* scratch = (wave < 4 ? rfree : ufree) + (wave & 3) * (32 * 33);
        const int* ra_w = rowa + wave * 32;
        const int* ry_w = rowy + wave * 32;
        if (EPI == W11_EPI_GENERIC) {
            if (args.pool_y) {
                // fused 2 x 2 pool, run-time options (see the specialised form below)
                const int pool_n = n0 + li;
                const bool pcol = pool_n < g.N;
                float bs = 0.f, scv = 1.f, shv = 0.f;
                if (pcol) { bs = pconst[pool_n]; scv = pconst[g.N + pool_n]; shv = pconst[2 * g.N + pool_n]; }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float vv[4];
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        float x = out[0][0][4 * p + i] + bs;
                        if (g.relu == 1) x = fmaxf(x, 0.f);
                        else if (g.relu == 2) x = tanhf(x);
                        vv[p] = fmaf(scv, x, shv);
                    }
                    const float pv = args.pool_mode == 1 ? 0.25f * ((vv[0] + vv[1]) + (vv[2] + vv[3])) : fmaxf(fmaxf(vv[0], vv[1]), fmaxf(vv[2], vv[3]));
                    const int pr = prow[wave * 32 + i + 4 * lh];
                    if (pr >= 0 && pcol) __builtin_nontemporal_store(pv, args.pool_y + (long)pr * g.N + pool_n);
                }
            }
            W11T(7);
            tap_epilogue<1, 1>(g, out, scratch, rowa, rowy, wave * 32, n0, lane, blk * 8 + wave);
        } else {
            // everything the specialised epilogues need from the argument block, loaded once
            const int N = g.N;
            const unsigned n4 = ncol ? (unsigned)ne * 4u : W11_OOR;
            const unsigned pa = (unsigned)g.ldo_a * 4u, py = (unsigned)g.ldo_y * 4u;
            auto rA = __builtin_amdgcn_make_buffer_rsrc((void*)(POOLED || EPI == W11_EPI_FWD || EPI == W11_EPI_FWD_SUM ? g.out_a : g.out_y), 0, 0xFFFFFFF0, 0x00020000);
            auto rY = __builtin_amdgcn_make_buffer_rsrc((void*)g.out_y, 0, 0xFFFFFFF0, 0x00020000);
            if (POOLED) {
                // fused 2 x 2 pool: register 4 p + i = pixel p of tile i + 4 lh -- the window is lane-local.  Same arithmetic and
                // association as asr_pool_fwd on the stored activation (bit-identical): BN(ReLU(x + bias)), row pairs, then the rows.
                // Stored straight from the registers: a half-wave writes the 128 bytes of one pooled pixel's 32 channels.
                auto rP = __builtin_amdgcn_make_buffer_rsrc((void*)args.pool_y, 0, 0xFFFFFFF0, 0x00020000);
                const int pool_n = n0 + li;
                const bool pcol = pool_n < N;
                float bs = 0.f, scv = 1.f, shv = 0.f;
                if (pcol) { bs = pconst[pool_n]; scv = pconst[N + pool_n]; shv = pconst[2 * N + pool_n]; }
                unsigned po[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int pr = prow[wave * 32 + i + 4 * lh];
                    po[i] = (pr >= 0 && pcol) ? __umul24((unsigned)pr, (unsigned)N * 4u) + (unsigned)pool_n * 4u : W11_OOR;
                }
                if (EPI == W11_EPI_POOLMAXC) {
                    // Compact form: the pre-pool activation plane is not written at all.  Per window: the pooled value, the activation
                    // AT the first maximum (row-major window order, asr_cell_bwd_pre's rule) and its position -- two ballots per tile
                    // slot give the bit planes of 32 channels x 2 tiles, stored by lane 0 of each half-wave.
                    auto rPA = __builtin_amdgcn_make_buffer_rsrc((void*)args.pool_amax, 0, 0xFFFFFFF0, 0x00020000);
                    auto rPI = __builtin_amdgcn_make_buffer_rsrc((void*)args.pool_idx, 0, 0xFFFFFFF0, 0x00020000);
                    const unsigned nb8 = (unsigned)(N >> 5) * 8u, cb8 = (unsigned)(n0 >> 5) * 8u;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        // (scalars and selects on purpose: written with arrays, hipcc turns "the activation at the arg-max" into an indexed
                        //  load from a private array in scratch -- a store and a dependent load per window, behind every store in flight)
                        const float x0 = fmaxf(out[0][0][i] + bs, 0.f), x1 = fmaxf(out[0][0][4 + i] + bs, 0.f);
                        const float x2 = fmaxf(out[0][0][8 + i] + bs, 0.f), x3 = fmaxf(out[0][0][12 + i] + bs, 0.f);
                        const float v0 = fmaf(scv, x0, shv), v1 = fmaf(scv, x1, shv), v2 = fmaf(scv, x2, shv), v3 = fmaf(scv, x3, shv);
                        float m = v0, am = x0; int ag = 0;
                        const bool c1 = v1 > m; m = c1 ? v1 : m; am = c1 ? x1 : am; ag = c1 ? 1 : ag;
                        asm volatile("" : "+v"(am), "+v"(ag));
                        const bool c2 = v2 > m; m = c2 ? v2 : m; am = c2 ? x2 : am; ag = c2 ? 2 : ag;
                        asm volatile("" : "+v"(am), "+v"(ag));
                        const bool c3 = v3 > m; m = c3 ? v3 : m; am = c3 ? x3 : am; ag = c3 ? 3 : ag;
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(m), rP, po[i], 0, 2);
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(am), rPA, po[i], 0, 2);
                        const unsigned long long b0 = __ballot(ag & 1), b1 = __ballot(ag >> 1);
                        const unsigned w0 = lh ? (unsigned)(b0 >> 32) : (unsigned)b0, w1 = lh ? (unsigned)(b1 >> 32) : (unsigned)b1;
                        const int pr = prow[wave * 32 + i + 4 * lh];
                        const unsigned io = (pr >= 0 && li == 0) ? __umul24((unsigned)pr, nb8) + cb8 : W11_OOR;
                        __builtin_amdgcn_raw_buffer_store_b32(w0, rPI, io, 0, 0);
                        __builtin_amdgcn_raw_buffer_store_b32(w1, rPI, io == W11_OOR ? W11_OOR : io + 4u, 0, 0);
                    }
                } else if (EPI == W11_EPI_POOLAVGC) {
                    // Compact form of the AVERAGE pool (round 5): no activation plane either.  Per window: the pooled value (POOLAVG's
                    // arithmetic), the SUM of the four activations (all the backward's scale sum needs of them) and the ReLU sign of each
                    // position -- four ballots per tile slot, one 16-byte store per pooled pixel and 32-channel block.
                    auto rPA = __builtin_amdgcn_make_buffer_rsrc((void*)args.pool_amax, 0, 0xFFFFFFF0, 0x00020000);
                    auto rPI = __builtin_amdgcn_make_buffer_rsrc((void*)args.pool_idx, 0, 0xFFFFFFF0, 0x00020000);
                    const unsigned nb16 = (unsigned)(N >> 5) * 16u, cb16 = (unsigned)(n0 >> 5) * 16u;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float x0 = fmaxf(out[0][0][i] + bs, 0.f), x1 = fmaxf(out[0][0][4 + i] + bs, 0.f);
                        const float x2 = fmaxf(out[0][0][8 + i] + bs, 0.f), x3 = fmaxf(out[0][0][12 + i] + bs, 0.f);
                        const float v0 = fmaf(scv, x0, shv), v1 = fmaf(scv, x1, shv), v2 = fmaf(scv, x2, shv), v3 = fmaf(scv, x3, shv);
                        const float pv = 0.25f * ((v0 + v1) + (v2 + v3));
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(pv), rP, po[i], 0, 2);
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint((x0 + x1) + (x2 + x3)), rPA, po[i], 0, 2);
                        const unsigned long long b0 = __ballot(x0 > 0.f), b1 = __ballot(x1 > 0.f), b2 = __ballot(x2 > 0.f), b3 = __ballot(x3 > 0.f);
                        const w11_u4 wd = {lh ? (unsigned)(b0 >> 32) : (unsigned)b0, lh ? (unsigned)(b1 >> 32) : (unsigned)b1,
                                           lh ? (unsigned)(b2 >> 32) : (unsigned)b2, lh ? (unsigned)(b3 >> 32) : (unsigned)b3};
                        const int pr = prow[wave * 32 + i + 4 * lh];
                        __builtin_amdgcn_raw_buffer_store_b128(wd, rPI, (pr >= 0 && li == 0) ? __umul24((unsigned)pr, nb16) + cb16 : W11_OOR, 0, 0);
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float vv[4];
#pragma unroll
                        for (int p = 0; p < 4; ++p) vv[p] = fmaf(scv, fmaxf(out[0][0][4 * p + i] + bs, 0.f), shv);
                        const float pv = EPI == W11_EPI_POOLAVG ? 0.25f * ((vv[0] + vv[1]) + (vv[2] + vv[3])) : fmaxf(fmaxf(vv[0], vv[1]), fmaxf(vv[2], vv[3]));
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(pv), rP, po[i], 0, 2);
                    }
                }
            }
            W11T(7);
            if (EPI == W11_EPI_FWD) wino11_epilogue_plain<true, true, true, false>(rA, rY, pa, py, out[0][0], scratch, ra_w, ry_w, n4, cbs, csc, csh, lane);
            else if (EPI == W11_EPI_FWD_SUM) wino11_epilogue_plain<true, true, true, false, 1>(rA, rY, pa, py, out[0][0], scratch, ra_w, ry_w, n4, cbs, csc, csh, lane,
                                                                                                g.gate_part + (long)(blk * 8 + wave) * N);
            else if (EPI == W11_EPI_DGRAD_SESUM) {
                auto rX = __builtin_amdgcn_make_buffer_rsrc((void*)g.gate_a, 0, 0xFFFFFFF0, 0x00020000);
                wino11_epilogue_plain<false, false, true, false, 2>(rA, rY, pa, py, out[0][0], scratch, ra_w, ry_w, n4, cbs, csc, csh, lane,
                                                                    g.gate_part + (long)(blk * 8 + wave) * N, rX);
            }
            else if (EPI == W11_EPI_POOLMAXC || EPI == W11_EPI_POOLAVGC) { }            // nothing else to store: no activation plane in the compact forms
            else if (POOLED) wino11_epilogue_plain<true, true, false, false>(rA, rY, pa, py, out[0][0], scratch, ra_w, ry_w, n4, cbs, csc, csh, lane);
            else if (EPI == W11_EPI_DGRAD) wino11_epilogue_plain<false, false, true, false>(rA, rY, pa, py, out[0][0], scratch, ra_w, ry_w, n4, cbs, csc, csh, lane);
            else if (EPI == W11_EPI_DGRAD_ACC) wino11_epilogue_plain<false, false, true, true>(rA, rY, pa, py, out[0][0], scratch, ra_w, ry_w, n4, cbs, csc, csh, lane);
            else {
                auto rGA = __builtin_amdgcn_make_buffer_rsrc((void*)g.gate_a, 0, 0xFFFFFFF0, 0x00020000);
                auto rGD = __builtin_amdgcn_make_buffer_rsrc((void*)g.gate_dz, 0, 0xFFFFFFF0, 0x00020000);
                const unsigned c4b = (unsigned)N * 4u, wpf4 = (unsigned)(g.gate_W + 1) * c4b;
                const bool accu = g.accumulate != 0;
                float* gpart = g.gate_part;
                constexpr int GM = EPI == W11_EPI_GATE1 ? 1 : EPI == W11_EPI_GATE2 ? 2 : EPI == W11_EPI_GATE3 ? 3 : EPI == W11_EPI_GATE4 ? 4 : 5;
                auto rGI = __builtin_amdgcn_make_buffer_rsrc((void*)args.pool_idx, 0, 0xFFFFFFF0, 0x00020000);
                wino11_epilogue_gated<GM>(rGA, rGD, rY, rGI, accu, py, c4b, wpf4, gpart, N, out[0][0], scratch, ra_w, ry_w, ne, n4, csc, csh, lane, blk * 8 + wave);
            }
        }
        W11T(8);
        lds_barrier();                               // scratch sets and this item's tables are free; the next item's tables and data visible
        W11T(11);
        if (newblk) tcur ^= 1;
        if (++nb == nnb) { nb = 0; ++blk; }
    }
}
#endif

template <int DIR, int EPI>
__global__ __launch_bounds__(512, 4) void wino11_kernel(WinoArgs args) {
#if __HIP_DEVICE_COMPILE__
    extern __shared__ __attribute__((aligned(16))) float smem[];
    wino11_body<EPI>(smem);
#endif
}


template <int DIR>
__global__ __launch_bounds__(512) void wino8_kernel(WinoArgs args) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    if ((threadIdx.x >> 8) == 0) wino8_body<0>(args, smem);
    else wino8_body<1>(args, smem);
}

}  // namespace

W11_TRACE_DUMP

// two layouts side by side: [16][K][N] (wino8_kernel) and, 16 K N floats further, the chunk-major one of wino11_kernel
extern "C" size_t asr_winograd_weights_bytes(int K, int N) { return (size_t)2 * 16 * K * N * sizeof(float); }

// (round 5: `asr_winograd_weights2` with the size of `out` -- round 4 doubled what the entry point writes under its old name;
// a caller that still allocates 16 K N floats now gets ASR_ERR_BAD_ARG, and one built against the old name fails to link)
extern "C" int asr_winograd_weights2(const float* W, int K, int N, int ldw, int wmode, float* out, size_t out_bytes, void* stream) {
    if (!W || !out || K < 1 || N < 1 || ldw < 1 || out_bytes < asr_winograd_weights_bytes(K, N)) return ASR_ERR_BAD_ARG;
    long nb = ((long)K * N + 255) / 256;
    if (nb > 4096) nb = 4096;
    float* out2 = ((K & 7) == 0 && (N & 31) == 0) ? out + (size_t)16 * K * N : nullptr;
    hipLaunchKernelGGL(wino_weights_kernel, dim3((int)nb), dim3(256), 0, (hipStream_t)stream, W, K, N, ldw, wmode, out, out2);
    ASR_CHECK_LAUNCH("winograd_weights");
    return ASR_OK;
}

// column blocks of 11..15 tile columns (wino11_kernel); 0: no such split
static int wino_column_blocks(int TW, int* tj0, int* cw) {
    for (int nb = asr_cdiv(TW, 15); nb <= 8 && nb * 11 <= TW; ++nb) {
        if (asr_cdiv(TW, nb) > 15) continue;
        int tj = 0;
        for (int c = 0; c < nb; ++c) {
            const int w = TW / nb + (c < TW % nb ? 1 : 0);
            if (tj0) tj0[c] = tj;
            if (cw) cw[c] = w;
            tj += w;
        }
        return nb;
    }
    return 0;
}

// wino11_kernel takes a shape when its tile columns split into column blocks, N is a multiple of 32 and every plane it touches fits
// its 32-bit addressing: outputs and gate planes go through buffer resources with 24-bit row x 24-bit pitch offsets (< 4 GiB,
// < 2^24 rows).  A function of the descriptor's widths and geometry only (plus the launch's output pitches).
static bool wino11_takes(const asr_gemm_desc* d, bool pooled, int gate_mode) {
    if ((d->N % W11_C) != 0 || d->N > 2048 || wino_column_blocks((d->W + 1) / 2, nullptr, nullptr) == 0) return false;
    if (pooled && d->N > 640) return false;                                  // the pool's channel constants live in LDS (3 N floats)
    const long rows = d->M, grows = gate_mode >= 2 ? (long)d->B * (2 * d->H + 1) * (2 * d->W + 1) : rows;
    const long pitch = d->ldo_a > d->ldo_y ? d->ldo_a : d->ldo_y;
    if (grows >= (1L << 24) || pitch * 4 >= (1L << 24) || (long)d->N * 4 >= (1L << 24)) return false;
    if (rows * (pitch > d->N ? pitch : d->N) * 4 >= 0xFFFFFFF0L || grows * d->N * 4 >= 0xFFFFFFF0L) return false;
    return true;
}

// An odd plane height is one half-filled last tile row: its second pixel row is the zero border under the image (rows beyond it
// are sent out of range / never enter a stored pixel) and the row tables carry no entry for it, as for the last column of an
// odd width.  The input plane is read through 32-bit offsets (buffer-form DMA with num_records 0x7FFFFFF0 in wino11_kernel --
// offsets past it read zeros by design --, 32-bit lane offsets in wino8_kernel): planes of 2 GiB or more are refused here and stay
// on the direct kernels (asr_tap_gemm_pw), which address with 64 bits.
extern "C" int asr_winograd_supported(const asr_gemm_desc* d) {
    if (!(d && d->ntaps == 9 && d->H > 0 && d->W >= 2 && (d->K % WKC) == 0 && (d->lda & 3) == 0 &&
          d->M == d->B * (d->H + 1) * (d->W + 1))) return 0;
    if ((long)d->M * d->lda * 4 >= 0x7FFFFFF0L || (long)16 * d->K * d->N * 4 >= 0x7FFFFFF0L) return 0;
    if ((d->N % WC) == 0) return 1;                     // wino8_kernel (or wino11_kernel)
    return wino11_takes(d, false, 3) ? 1 : 0;           // 32-wide channel blocks: wino11_kernel only (any epilogue the ABI can ask for)
}

struct WinoGate { int mode, H, W; const float* a; float* dz; float* part; int* rows; };

// Tile-block count of a launch (work items along the pixel axis): B x items per image in the column-blocked order, else
// ceil(tiles / 64).  A gated launch writes up to 8 partial rows per tile block (asr_tap_gemm_gated_workspace sizes its buffer with it:
// planes of a few tile rows have mostly empty items, so this can exceed the direct kernels' M / 32 rows).
static int wino_tile_blocks(const asr_gemm_desc* d) {
    const int TH = (d->H + 1) / 2, TW = (d->W + 1) / 2;
    int cw[8], tj0[8];
    const int ncb = wino_column_blocks(TW, tj0, cw);
    if (!ncb) return asr_cdiv((long)d->B * TH * TW, WT);
    int it = 0;
    for (int c = 0; c < ncb; ++c) it += asr_cdiv(TH * cw[c], WT);
    const int blocked = d->B * it, plain = asr_cdiv((long)d->B * TH * TW, WT);
    return blocked > plain ? blocked : plain;            // (wino8_kernel's plain order is the fallback of a blocked geometry)
}
extern "C" ASR_INTERNAL int asr_winograd_gate_rows(const asr_gemm_desc* d) { return asr_winograd_supported(d) ? 8 * wino_tile_blocks(d) : 0; }      // wino11_kernel: one per wave

static int wino_impl(const asr_gemm_desc* d, const float* A, const float* Ut, const float* bias, const float* scale, const float* shift,
                     float* out_a, float* out_y, void* stream, const WinoGate* gs, float* pool_y = nullptr, int pool_mode = 0,
                     float* pool_amax = nullptr, unsigned* pool_idx = nullptr, float* y_sums = nullptr, const float* se_x = nullptr) {
    if (!d || !A || !Ut || (!out_a && !out_y && !gs && !pool_amax)) return ASR_ERR_BAD_ARG;
    if (!asr_winograd_supported(d)) return ASR_ERR_UNSUPPORTED;
    if ((((uintptr_t)A) | ((uintptr_t)Ut)) & 15) return ASR_ERR_BAD_ARG;
    WinoArgs w;
    TapGemmArgs& a = w.g;
    a.A = A; a.W = nullptr; a.bias = bias; a.scale = scale; a.shift = shift;
    a.out_a = out_a; a.out_y = out_y;
    a.M = d->M; a.K = d->K; a.N = d->N; a.lda = d->lda; a.ldw = d->ldw;
    a.ldo_a = d->ldo_a; a.ldo_y = d->ldo_y;
    a.H = d->H; a.Wd = d->W; a.WP = d->W + 1; a.HPWP = (d->H + 1) * (d->W + 1);
    a.halo = a.WP + 1;
    a.rmin = -(long)a.halo; a.rmax = (long)d->M + a.halo;
    a.relu = d->relu; a.accumulate = d->accumulate; a.y_unpadded = d->y_unpadded;
    a.nt_store = 1;       // streaming stores: -2..3 % per layer (the planes do not fit L2 anyway)
    a.gate_mode = 0; a.gate_H = a.gate_W = 0; a.gate_a = nullptr; a.gate_dz = nullptr; a.gate_part = nullptr; a.gate_rows = nullptr;
    if (gs) { a.gate_mode = gs->mode; a.gate_H = gs->H; a.gate_W = gs->W; a.gate_a = gs->a; a.gate_dz = gs->dz; a.gate_part = gs->part; a.gate_rows = gs->rows; }
    if (y_sums) { if (gs) return ASR_ERR_BAD_ARG; a.gate_part = y_sums; a.gate_a = se_x; }      // W11_EPI_FWD_SUM / _DGRAD_SESUM: partial rows [tile block x 8][N]
    w.Ut = Ut;
    w.TH = (d->H + 1) / 2; w.TW = (d->W + 1) / 2;
    w.ntiles = (long)d->B * w.TH * w.TW;
    w.wodd = d->W & 1;
    w.pool_y = pool_y; w.pool_mode = pool_mode; w.H2 = d->H / 2; w.W2 = d->W / 2;
    w.pool_amax = pool_amax; w.pool_idx = pool_idx;
    const bool compact = pool_amax != nullptr || (gs && (gs->mode == 4 || gs->mode == 6));
    static int ncu8 = 0;
    if (!ncu8) {
        int dev = 0; hipDeviceProp_t pr;
        ncu8 = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) ? pr.multiProcessorCount : 256;
    }
    w.ncb = wino_column_blocks(w.TW, w.cb_tj0, w.cb_w);
    int it = 0;
    for (int c = 0; c < w.ncb; ++c) { w.cb_it0[c] = it; it += asr_cdiv(w.TH * w.cb_w[c], WT); }
    w.cb_it0[w.ncb] = it;
    w.inv_ipi = it > 0 ? 1.0f / (float)it : 0.f;
    for (int c = 0; c < 8; ++c) w.cb_invw[c] = c < w.ncb ? 1.0f / (float)w.cb_w[c] : 0.f;
    hipStream_t st = (hipStream_t)stream;
    // wino11_kernel (eight waves of 128 registers, one transform row per wave, two workgroups per CU) takes every column-blocked shape
    // with 32-wide channel blocks; the choice depends on widths and plane geometry only, never on the batch or the CU count.
    if (wino11_takes(d, pool_y != nullptr, a.gate_mode)) {
        const int nblk11 = d->B * w.cb_it0[w.ncb];
        a.ntm = nblk11; a.ntn = d->N / W11_C;
        if (!asr_gate_rows_fit(a.gate_rows, nblk11 * 8)) return ASR_ERR_UNSUPPORTED;
        const long nwork11 = (long)nblk11 * a.ntn;
#if defined(ASR_DEV_HOOKS) && defined(W11_ONE_PER_CU)      // development probe (tools/build_variant.sh one wino.hip -DASR_DEV_HOOKS -DW11_ONE_PER_CU): ONE workgroup per CU,
                                                           // the LDS request padded so that two cannot share one (profiles/r05_wino11_one_wg_probe.txt)
        const int grid11 = nwork11 > (long)ncu8 ? ncu8 : (int)nwork11;
        const size_t lds11 = 82 * 1024;
#else
        const int grid11 = nwork11 > 2L * ncu8 ? 2 * ncu8 : (int)nwork11;     // persistent: two workgroups per CU
        const size_t lds11 = (size_t)(2 * W11_TABF + 4 * W11_SETF + (pool_y ? 3 * d->N : 0)) * sizeof(float);
#endif
        static_assert(4096 <= W11_SETF && 4 * 32 * 33 <= W11_SETF, "exchange phase / four transpose scratches must fit in one buffer set");
        // the epilogue instantiation (wino11_body): decided by the descriptor
        int epi = W11_EPI_GENERIC;
        const bool aff = scale && shift;
        if (gs) epi = gs->mode == 1 ? W11_EPI_GATE1 : gs->mode == 2 ? W11_EPI_GATE2 : gs->mode == 3 ? W11_EPI_GATE3 : gs->mode == 4 ? W11_EPI_GATE4 : W11_EPI_GATE5;
        else if (pool_amax) {
            if (d->wmode || d->relu != 1 || !bias || !aff || !pool_y || !pool_idx || (pool_mode != 2 && pool_mode != 1)) return ASR_ERR_BAD_ARG;
            epi = pool_mode == 2 ? W11_EPI_POOLMAXC : W11_EPI_POOLAVGC;
        }
        else if (!d->wmode && d->relu == 1 && bias && aff && out_a && pool_y) epi = pool_mode == 2 ? W11_EPI_POOLMAX : W11_EPI_POOLAVG;
        else if (!d->wmode && d->relu == 1 && bias && aff && out_a && out_y && !d->accumulate && !pool_y) epi = y_sums ? W11_EPI_FWD_SUM : W11_EPI_FWD;
        else if (d->wmode && d->relu == 0 && !bias && !scale && !shift && !out_a && out_y && !pool_y) epi = d->accumulate ? W11_EPI_DGRAD_ACC : W11_EPI_DGRAD;
        else if (d->wmode && d->relu == 0 && !bias && aff && !out_a && out_y && !pool_y && !d->accumulate && y_sums && se_x && d->ldo_y == d->N) epi = W11_EPI_DGRAD_SESUM;
        if (y_sums && epi != W11_EPI_FWD_SUM && epi != W11_EPI_DGRAD_SESUM) return ASR_ERR_UNSUPPORTED;
        typedef void (*w11_fn)(WinoArgs);
        static const w11_fn fns[15] = {nullptr, wino11_kernel<0, 1>, wino11_kernel<0, 2>, wino11_kernel<0, 3>, wino11_kernel<1, 4>, wino11_kernel<1, 5>,
                                       wino11_kernel<1, 6>, wino11_kernel<1, 7>, wino11_kernel<1, 8>, wino11_kernel<0, 9>, wino11_kernel<1, 10>,
                                       wino11_kernel<0, 11>, wino11_kernel<1, 12>, wino11_kernel<0, 13>, wino11_kernel<1, 14>};
        const w11_fn fn = epi ? fns[epi] : (d->wmode ? (w11_fn)wino11_kernel<1, 0> : (w11_fn)wino11_kernel<0, 0>);
        static bool attr[16] = {false, false, false, false, false, false, false, false, false, false, false, false, false, false, false, false};
        const int slot = epi ? (epi >= W11_EPI_FWD_SUM ? epi + 1 : epi) : (d->wmode ? 11 : 0);
        if (!attr[slot]) { (void)hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds11 > 80 * 1024 ? (int)lds11 : 80 * 1024); attr[slot] = true; }
        hipLaunchKernelGGL(fn, dim3(grid11), dim3(512), lds11, st, w);
        ASR_CHECK_LAUNCH("tap_gemm_wino11");
        static const char* const names[16] = {"wino11_kernel<0, 0>", "wino11_kernel<0, 1>", "wino11_kernel<0, 2>", "wino11_kernel<0, 3>", "wino11_kernel<1, 4>",
                                              "wino11_kernel<1, 5>", "wino11_kernel<1, 6>", "wino11_kernel<1, 7>", "wino11_kernel<1, 8>", "wino11_kernel<0, 9>",
                                              "wino11_kernel<1, 10>", "wino11_kernel<1, 0>", "wino11_kernel<0, 11>", "wino11_kernel<1, 12>", "wino11_kernel<0, 13>", "wino11_kernel<1, 14>"};
        asr_set_last_kernel(names[slot]);
        return ASR_OK;
    }
    // wino8_kernel: 64 x 64 items in the plain tile order (planes whose tile columns do not split into column blocks, N % 64 == 0)
    if ((d->N % WC) != 0 || compact || y_sums) return ASR_ERR_UNSUPPORTED;
    w.ncb = 0;
    const int nblk = asr_cdiv(w.ntiles, WT);
    a.ntm = nblk; a.ntn = d->N / WC;
    if (!asr_gate_rows_fit(a.gate_rows, nblk * 4)) return ASR_ERR_UNSUPPORTED;
    const size_t lds8 = (size_t)(576 + 2 * RAW_F + 2 * U_F) * sizeof(float);
    static_assert(4 * 2048 <= RAW_F && 4 * 2048 <= U_F && 4 * 32 * 33 + 2 * 1024 <= RAW_F && 4 * 32 * 33 + 2 * 1024 <= U_F, "exchange / scratch + pool hand-over must fit in one buffer set");
    const int nwork8 = nblk * a.ntn;
    const int grid8 = nwork8 > ncu8 ? ncu8 : nwork8;        // persistent: one workgroup per CU
    auto q0 = wino8_kernel<0>;
    auto q1 = wino8_kernel<1>;
    static bool b0 = false, b1 = false;
    if (d->wmode) {
        if (!b1) { (void)hipFuncSetAttribute((const void*)q1, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds8); b1 = true; }
        hipLaunchKernelGGL(q1, dim3(grid8), dim3(512), lds8, st, w);
    } else {
        if (!b0) { (void)hipFuncSetAttribute((const void*)q0, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds8); b0 = true; }
        hipLaunchKernelGGL(q0, dim3(grid8), dim3(512), lds8, st, w);
    }
    ASR_CHECK_LAUNCH("tap_gemm_wino8");
    if (d->wmode) ASR_NOTE_KERNEL("wino8_kernel<1>"); else ASR_NOTE_KERNEL("wino8_kernel<0>");     // one name per call site
    return ASR_OK;
}

extern "C" int asr_tap_gemm_wino(const asr_gemm_desc* d, const float* A, const float* Ut,
                                 const float* bias, const float* scale, const float* shift,
                                 float* out_a, float* out_y, void* stream) {
    return wino_impl(d, A, Ut, bias, scale, shift, out_a, out_y, stream, nullptr);
}

// Forward conv of a cell that is the BRANCH of a squeeze-excitation block (round 5): asr_tap_gemm_wino + the channel sums of its
// BN output y per image, as partial rows y_sums[rows][N] with rows = asr_winograd_sum_rows(d), the rows of image b being
// [b * rows / B, (b + 1) * rows / B) -- what the block's global average pool needs (asr_se_fwd_sums), without its pass over the plane.
// ASR_ERR_UNSUPPORTED where wino11_kernel's forward epilogue does not apply (then: asr_tap_gemm_wino + asr_se_fwd).
extern "C" int asr_winograd_sum_rows(const asr_gemm_desc* d) {
    if (!d || !asr_winograd_supported(d) || !wino11_takes(d, false, 0)) return 0;
    return 8 * wino_tile_blocks(d);
}
extern "C" int asr_tap_gemm_wino_sums(const asr_gemm_desc* d, const float* A, const float* Ut, const float* bias, const float* scale,
                                      const float* shift, float* out_a, float* out_y, float* y_sums, void* stream) {
    if (!d || !bias || !scale || !shift || !out_a || !out_y || !y_sums || d->wmode || d->relu != 1 || d->accumulate || d->y_unpadded) return ASR_ERR_BAD_ARG;
    if (asr_winograd_sum_rows(d) <= 0) return ASR_ERR_UNSUPPORTED;
    return wino_impl(d, A, Ut, bias, scale, shift, out_a, out_y, stream, nullptr, nullptr, 0, nullptr, nullptr, y_sums);
}

// Data-gradient (wmode 1, no accumulate) that completes dL/d(output of an SE block) + the reduction the block's backward starts with:
// xsums[rows][N] partial rows (rows = asr_winograd_sum_rows(d): 8 per tile block, image-contiguous) of sum dy * (se_scale * x + se_shift),
// x = the block's branch plane [B][H + 1][W + 1][N].  asr_se_bwd_cell_sums takes them.
extern "C" int asr_tap_gemm_wino_sesum(const asr_gemm_desc* d, const float* dZ, const float* Ut, const float* x, const float* se_scale,
                                       const float* se_shift, float* dy, float* xsums, void* stream) {
    if (!d || !dZ || !Ut || !x || !se_scale || !se_shift || !dy || !xsums || !d->wmode || d->relu || d->accumulate || d->y_unpadded || d->ldo_y != d->N)
        return ASR_ERR_BAD_ARG;
    if (!asr_winograd_supported(d) || !wino11_takes(d, false, 0)) return ASR_ERR_UNSUPPORTED;
    return wino_impl(d, dZ, Ut, nullptr, se_scale, se_shift, nullptr, dy, stream, nullptr, nullptr, 0, nullptr, nullptr, xsums, x);
}

extern "C" int asr_pool_fwd(const float* a, int B, int H, int W, int C, const float* bn_scale, const float* bn_shift, int pool,
                            float* y, void* stream);

// Forward conv of a POOLED cell in one launch: out_a = ReLU(conv + bias) as asr_tap_gemm_wino writes it, and
// y_pooled = 2x2 pool (1 average / 2 maximum) of bn_scale * out_a + bn_shift -- what asr_pool_fwd would compute from out_a,
// bit for bit, without reading it back (the four pixels of a Winograd tile are one pooling window).
extern "C" int asr_tap_gemm_wino_pool(const asr_gemm_desc* d, const float* A, const float* Ut, const float* bias, const float* scale,
                                      const float* shift, float* out_a, int pool, float* y_pooled, void* stream) {
    if (!d || !out_a || !y_pooled || !scale || !shift || (pool != 1 && pool != 2) || d->ldo_a != d->N) return ASR_ERR_BAD_ARG;
    return wino_impl(d, A, Ut, bias, scale, shift, out_a, nullptr, stream, nullptr, y_pooled, pool);
}

// Launch of the gated data-gradient (asr_tap_gemm_gated with prearranged == 2; the caller folds the partial rows)
// ---- the maximum pool in its compact form (round 4): the forward launch of a max-pooled cell writes, instead of the pre-pool
// activation plane (4 x the pooled size, read back once by the backward pass), the pooled output, the activation AT each window's
// maximum and the maximum's position (2 bits); the gated data-gradient that completes the cell's output gradient reads those.
extern "C" size_t asr_poolmax_index_bytes(int B, int H2, int W2, int N) {
    if (B < 1 || H2 < 1 || W2 < 1 || N < 32 || (N & 31)) return 0;
    return (size_t)B * (H2 + 1) * (W2 + 1) * (N >> 5) * 2 * sizeof(unsigned);
}

// fwd: the pooled cell's forward descriptor (its pre-pool plane H x W); bwd: the data-gradient descriptor whose OUTPUT is the cell's
// pooled plane (H / 2 x W / 2, N = the cell's channels).  Widths and geometry only.
extern "C" int asr_winograd_poolmax_supported(const asr_gemm_desc* fwd, const asr_gemm_desc* bwd) {
    if (!fwd || !bwd || fwd->wmode != 0 || bwd->wmode != 1) return 0;
    if ((fwd->H & 1) || (fwd->W & 1) || bwd->H != fwd->H / 2 || bwd->W != fwd->W / 2 || bwd->N != fwd->N || bwd->B != fwd->B) return 0;
    if (!asr_winograd_supported(fwd) || !asr_winograd_supported(bwd)) return 0;
    return wino11_takes(fwd, true, 0) && wino11_takes(bwd, false, 4) ? 1 : 0;
}

extern "C" int asr_tap_gemm_wino_poolmax(const asr_gemm_desc* d, const float* A, const float* Ut, const float* bias, const float* scale,
                                         const float* shift, float* y_pooled, float* a_max, unsigned* index, void* stream) {
    if (!d || !A || !Ut || !bias || !scale || !shift || !y_pooled || !a_max || !index) return ASR_ERR_BAD_ARG;
    if ((d->H & 1) || (d->W & 1) || !asr_winograd_supported(d) || !wino11_takes(d, true, 0)) return ASR_ERR_UNSUPPORTED;
    return wino_impl(d, A, Ut, bias, scale, shift, nullptr, nullptr, stream, nullptr, y_pooled, 2, a_max, index);
}

extern "C" ASR_INTERNAL int asr_gated_partial_rows(const asr_gemm_desc* d);       // tap_gemm.hip: rows the partials buffer of asr_tap_gemm_gated_workspace holds

extern "C" int asr_tap_gemm_gated_poolmax(const asr_gemm_desc* d, const float* dZ, const float* Ut, int gate_H, int gate_W, const float* a_max,
                                          const unsigned* index, const float* bn_scale, const float* bn_shift, const float* dy_prev,
                                          float* dz_out, float* dscale, float* dshift, float* dbias, float* partials, void* stream) {
    if (!d || !dZ || !Ut || !a_max || !index || !bn_scale || !bn_shift || !dz_out || !dscale || !dshift || !dbias || !partials) return ASR_ERR_BAD_ARG;
    if (d->wmode != 1 || d->relu != 0 || d->y_unpadded || d->H <= 0 || d->ldo_y != d->N || (d->accumulate && !dy_prev)) return ASR_ERR_BAD_ARG;
    if (gate_H != 2 * d->H || gate_W != 2 * d->W) return ASR_ERR_BAD_ARG;
    if (!asr_winograd_supported(d) || !wino11_takes(d, false, 4)) return ASR_ERR_UNSUPPORTED;
    int rows = asr_gated_partial_rows(d);                     // capacity of `partials`; wino_impl refuses before its launch when it needs more
    WinoGate gs;
    gs.mode = 4; gs.H = gate_H; gs.W = gate_W; gs.a = a_max; gs.dz = dz_out; gs.part = partials; gs.rows = &rows;
    const int rc = wino_impl(d, dZ, Ut, nullptr, bn_scale, bn_shift, nullptr, (float*)dy_prev, stream, &gs, nullptr, 0, nullptr, (unsigned*)index);
    if (rc != ASR_OK) return rc;
    if (rows <= 0) return ASR_ERR_UNSUPPORTED;
    asr_reduce::Multi m;
    m.nseg = 3; m.width[0] = d->N; m.width[1] = d->N; m.width[2] = d->N; m.width[3] = 0;
    m.out[0] = dscale; m.out[1] = dshift; m.out[2] = dbias; m.out[3] = nullptr;
    return asr_reduce::colsum_multi(partials, rows, 3L * d->N, m, partials + (size_t)rows * 3 * d->N, (hipStream_t)stream);
}

// ---- the AVERAGE pool in its compact form (round 5; acoustic_model2.py:116-124 "maxpool" = average pooling): the forward launch of an
// average-pooled cell writes the pooled output, the SUM of each window's four activations and the ReLU sign of each position (four bit planes
// of 32 channels per pooled pixel) instead of the pre-pool activation plane; the gated data-gradient that completes the cell's output gradient
// reads those.  dZ, dshift and dbias are the bits of the activation-plane form, dscale the same value rounded once instead of four times.
extern "C" size_t asr_poolavg_index_bytes(int B, int H2, int W2, int N) {
    if (B < 1 || H2 < 1 || W2 < 1 || N < 32 || (N & 31)) return 0;
    return (size_t)B * (H2 + 1) * (W2 + 1) * (N >> 5) * 4 * sizeof(unsigned);
}

extern "C" int asr_tap_gemm_wino_poolavg(const asr_gemm_desc* d, const float* A, const float* Ut, const float* bias, const float* scale,
                                         const float* shift, float* y_pooled, float* a_sum, unsigned* index, void* stream) {
    if (!d || !A || !Ut || !bias || !scale || !shift || !y_pooled || !a_sum || !index) return ASR_ERR_BAD_ARG;
    if ((d->H & 1) || (d->W & 1) || !asr_winograd_supported(d) || !wino11_takes(d, true, 0)) return ASR_ERR_UNSUPPORTED;
    return wino_impl(d, A, Ut, bias, scale, shift, nullptr, nullptr, stream, nullptr, y_pooled, 1, a_sum, index);
}

extern "C" int asr_tap_gemm_gated_poolavg(const asr_gemm_desc* d, const float* dZ, const float* Ut, int gate_H, int gate_W, const float* a_sum,
                                          const unsigned* index, const float* bn_scale, const float* bn_shift, const float* dy_prev,
                                          float* dz_out, float* dscale, float* dshift, float* dbias, float* partials, void* stream) {
    if (!d || !dZ || !Ut || !a_sum || !index || !bn_scale || !bn_shift || !dz_out || !dscale || !dshift || !dbias || !partials) return ASR_ERR_BAD_ARG;
    if (d->wmode != 1 || d->relu != 0 || d->y_unpadded || d->H <= 0 || d->ldo_y != d->N || (d->accumulate && !dy_prev)) return ASR_ERR_BAD_ARG;
    if (gate_H != 2 * d->H || gate_W != 2 * d->W) return ASR_ERR_BAD_ARG;
    if (!asr_winograd_supported(d) || !wino11_takes(d, false, 6)) return ASR_ERR_UNSUPPORTED;
    int rows = asr_gated_partial_rows(d);                     // capacity of `partials`; wino_impl refuses before its launch when it needs more
    WinoGate gs;
    gs.mode = 6; gs.H = gate_H; gs.W = gate_W; gs.a = a_sum; gs.dz = dz_out; gs.part = partials; gs.rows = &rows;
    const int rc = wino_impl(d, dZ, Ut, nullptr, bn_scale, bn_shift, nullptr, (float*)dy_prev, stream, &gs, nullptr, 0, nullptr, (unsigned*)index);
    if (rc != ASR_OK) return rc;
    if (rows <= 0) return ASR_ERR_UNSUPPORTED;
    asr_reduce::Multi m;
    m.nseg = 3; m.width[0] = d->N; m.width[1] = d->N; m.width[2] = d->N; m.width[3] = 0;
    m.out[0] = dscale; m.out[1] = dshift; m.out[2] = dbias; m.out[3] = nullptr;
    return asr_reduce::colsum_multi(partials, rows, 3L * d->N, m, partials + (size_t)rows * 3 * d->N, (hipStream_t)stream);
}

extern "C" ASR_INTERNAL int asr_tap_gemm_wino_gated_launch(const asr_gemm_desc* d, const float* dZ, const float* Ut, int mode, int gate_H, int gate_W,
                                              const float* gate_a, const float* scale, const float* shift, float* dy_prev,
                                              float* dz_out, float* partials, int* rows, void* stream) {
    WinoGate gs;
    gs.mode = mode; gs.H = gate_H; gs.W = gate_W; gs.a = gate_a; gs.dz = dz_out; gs.part = partials; gs.rows = rows;
    return wino_impl(d, dZ, Ut, nullptr, scale, shift, nullptr, dy_prev, stream, &gs);
}

// Weight gradient of the tap-GEMM family on the fp32 MFMA pipe:
//
//   dW[tap][k][n] = sum_m A[m + off(tap)][k] * dZ[m][n]
//
// The contraction runs over pixels (hundreds of thousands), the output is tiny
// (ntaps*K*N), so the grid splits the pixel axis into chunks; every block keeps ALL
// nine tap accumulators of its (32-channel k-tile x n-tile) in registers (9 x 16 fp32
// per lane) and walks its chunk in LDS-staged runs of PS pixels: one staged run of A
// (with its halo) feeds all nine taps, one staged run of dZ feeds all k-tiles.
// Chunk partials go to a slab and are summed by a second, fixed-order pass, so the
// result is bitwise reproducible (no float atomics).
#include "asr_common.h"
#include <stdlib.h>
#include <stdio.h>
#include <stdint.h>

namespace {

struct WgradArgs {
    const float* A; const float* Z; float* out;   // out: dW (nchunks == 1) or the partial slab
    int M, K, N, lda, ldz;
    int WP, halo;
    long rmin, rmax;
    int pch;                                      // pixels per chunk (multiple of PS)
    long slab;                                    // floats per chunk partial = ntaps*K*N
};

// ---- v1: direct staging, two barriers per run, two blocks per CU
template <int NTAPS, int TKW, int WAVES_N, int TNW, int PS>
__global__ __launch_bounds__(256, 2) void tap_wgrad_kernel_v1(WgradArgs g) {
    constexpr int WAVES_P = 4 / WAVES_N;
    constexpr int KT = TKW * 32, NT = WAVES_N * TNW * 32;
    constexpr int NACC = NTAPS * TKW * TNW;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int halo = g.halo;
    const int arows = PS + 2 * halo;
    float* As = smem;                 // [arows][KT]
    float* Zs = As + arows * KT;      // [PS][NT]

    const int tid = threadIdx.x;
    const int lane = tid & 63, li = lane & 31, lh = lane >> 5;
    const int wave = tid >> 6, wn = wave % WAVES_N, wp = wave / WAVES_N;
    const int chunk = blockIdx.x;
    const int k0 = blockIdx.y * KT, n0 = blockIdx.z * NT;
    const long cbeg = (long)chunk * g.pch;
    const long cend = (cbeg + g.pch < g.M) ? cbeg + g.pch : g.M;

    floatx16 acc[NTAPS][TKW][TNW];
#pragma unroll
    for (int t = 0; t < NTAPS; ++t)
#pragma unroll
        for (int a = 0; a < TKW; ++a)
#pragma unroll
            for (int b = 0; b < TNW; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][a][b][r] = 0.f;

    // 4-tap (phase-split stride-2 conv): taps whose weight block is structurally zero for this tile's input-channel
    // phase are skipped (see tap_gemm.hip step_valid); their accumulators stay zero.
    bool tv[NTAPS];
#pragma unroll
    for (int t = 0; t < NTAPS; ++t) {
        const int phase = (NTAPS == 4) ? k0 / (g.K >> 2) : 0;
        tv[t] = (NTAPS != 4) || !(((t >> 1) & (phase >> 1)) | ((t & 1) & (phase & 1)));
    }

    for (long ps0 = cbeg; ps0 < cend; ps0 += PS) {
        __syncthreads();
        // batches of SB independent loads before the LDS writes: a plain "load; store" loop makes hipcc wait
        // for every load before it issues the next (one exposed memory latency per float4)
        constexpr int SB = 4;
        for (int base = 0; base < arows * (KT / 4); base += SB * 256) {
            float4 t[SB];
#pragma unroll
            for (int i = 0; i < SB; ++i) {
                const int f = base + tid + i * 256;
                const int row = f / (KT / 4), c4 = f - row * (KT / 4);
                const long grow = ps0 - halo + row;
                const int kk = k0 + c4 * 4;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (f < arows * (KT / 4) && grow >= g.rmin && grow < g.rmax && kk < g.K)
                    v = *(const float4*)(g.A + grow * g.lda + kk);
                t[i] = v;
            }
#pragma unroll
            for (int i = 0; i < SB; ++i) {
                const int f = base + tid + i * 256;
                const int row = f / (KT / 4), c4 = f - row * (KT / 4);
                if (f < arows * (KT / 4)) *(float4*)(As + row * KT + c4 * 4) = t[i];
            }
        }
        for (int base = 0; base < PS * (NT / 4); base += SB * 256) {
            float4 t[SB];
#pragma unroll
            for (int i = 0; i < SB; ++i) {
                const int f = base + tid + i * 256;
                const int row = f / (NT / 4), n4 = f - row * (NT / 4);
                const long grow = ps0 + row;
                const int nn = n0 + n4 * 4;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (f < PS * (NT / 4) && grow < cend && nn < g.N)
                    v = *(const float4*)(g.Z + grow * g.ldz + nn);
                t[i] = v;
            }
#pragma unroll
            for (int i = 0; i < SB; ++i) {
                const int f = base + tid + i * 256;
                const int row = f / (NT / 4), n4 = f - row * (NT / 4);
                if (f < PS * (NT / 4)) *(float4*)(Zs + row * NT + n4 * 4) = t[i];
            }
        }
        __syncthreads();
        // Software-pipelined operand fetch, unrolled by two with two named register sets: the LDS reads of
        // the next pixel pair are issued BEFORE the MFMAs of the current pair and consumed after them, so
        // their latency hides under 9 x 64 MFMA cycles.
        {
            float a0[NTAPS][TKW], b0[TNW], a1[NTAPS][TKW], b1[TNW];
            auto fetch = [&](float (&an)[NTAPS][TKW], float (&bn)[TNW], int r) {
#pragma unroll
                for (int b = 0; b < TNW; ++b) bn[b] = Zs[(r + lh) * NT + (wn * TNW + b) * 32 + li];
#pragma unroll
                for (int t = 0; t < NTAPS; ++t) {
                    if (!tv[t]) continue;
                    // 9: 3x3 SAME window; 4: forward-looking 2x2 window of the phase-split stride-2 conv (see tap_gemm.hip)
                    const int off = (NTAPS == 9) ? ((t / 3) - 1) * g.WP + (t % 3) - 1
                                  : (NTAPS == 4) ? (t >> 1) * g.WP + (t & 1) : 0;
#pragma unroll
                    for (int a = 0; a < TKW; ++a) an[t][a] = As[(r + lh + halo + off) * KT + li + a * 32];
                }
            };
            auto fma_all = [&](const float (&ac)[NTAPS][TKW], const float (&bc)[TNW]) {
#pragma unroll
                for (int t = 0; t < NTAPS; ++t) {
                    if (!tv[t]) continue;
#pragma unroll
                    for (int a = 0; a < TKW; ++a)
#pragma unroll
                        for (int b = 0; b < TNW; ++b)
                            acc[t][a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[t][a], bc[b], acc[t][a][b], 0, 0, 0);
                }
            };
            constexpr int RS = 2 * WAVES_P;             // pixel-pair stride of this wave (PS / RS is even)
            static_assert((PS / RS) % 2 == 0, "unroll by two");
            fetch(a0, b0, 2 * wp);
            for (int r = 2 * wp; r < PS; r += 2 * RS) {
                fetch(a1, b1, r + RS);
                fma_all(a0, b0);
                if (r + 2 * RS < PS) fetch(a0, b0, r + 2 * RS);
                fma_all(a1, b1);
            }
        }
    }

    // fold the pixel-split waves of this block through LDS, one tap at a time (keeps the
    // transfer at 16 registers per lane) and in a fixed order wp = 1, 2, ...
    if (WAVES_P > 1) {
        float* red = smem;   // [WAVES_P-1][WAVES_N][TKW*TNW*16][64]
#pragma unroll
        for (int t = 0; t < NTAPS; ++t) {
            __syncthreads();
            if (wp > 0) {
                float* dst = red + (((wp - 1) * WAVES_N + wn) * (TKW * TNW * 16)) * 64 + lane;
#pragma unroll
                for (int a = 0; a < TKW; ++a)
#pragma unroll
                    for (int b = 0; b < TNW; ++b)
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            dst[((a * TNW + b) * 16 + r) * 64] = acc[t][a][b][r];
            }
            __syncthreads();
            if (wp == 0) {
                for (int src = 1; src < WAVES_P; ++src) {
                    const float* sp = red + (((src - 1) * WAVES_N + wn) * (TKW * TNW * 16)) * 64 + lane;
#pragma unroll
                    for (int a = 0; a < TKW; ++a)
#pragma unroll
                        for (int b = 0; b < TNW; ++b)
#pragma unroll
                            for (int r = 0; r < 16; ++r)
                                acc[t][a][b][r] += sp[((a * TNW + b) * 16 + r) * 64];
                }
            }
        }
    }
    if (wp != 0) return;

    float* out = g.out + (long)chunk * g.slab;
#pragma unroll
    for (int t = 0; t < NTAPS; ++t)
#pragma unroll
        for (int a = 0; a < TKW; ++a)
#pragma unroll
            for (int b = 0; b < TNW; ++b) {
                const int n = n0 + (wn * TNW + b) * 32 + li;
                if (n >= g.N) continue;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int k = k0 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (k < g.K) out[((long)t * g.K + k) * g.N + n] = acc[t][a][b][r];
                }
            }
}

// ---- v3: as v1, but dZ never touches LDS.  A wave only needs its own 32 output channels of dZ, and for a
// pixel pair that is one coalesced 2 x 128-byte row segment per wave: it is loaded straight into registers
// (all pairs of the run up front, before the A staging and its barriers, so the latency is covered) and used
// as the MFMA B operand.  LDS traffic per run drops from 47 KB to 15 KB (N = 128), the barrier phase shrinks.
template <int NTAPS, int TKW, int WAVES_N, int TNW, int PS>
__global__ __launch_bounds__(256, 2) void tap_wgrad_kernel_v3(WgradArgs g) {
    constexpr int WAVES_P = 4 / WAVES_N;
    constexpr int KT = TKW * 32, NT = WAVES_N * TNW * 32;
    constexpr int NACC = NTAPS * TKW * TNW;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int halo = g.halo;
    const int arows = PS + 2 * halo;
    float* As = smem;                 // [arows][KT]
    constexpr int RS = 2 * WAVES_P;   // pixel-pair stride of this wave
    constexpr int NPW = PS / RS;      // pixel pairs per wave and run
    static_assert(NPW % 2 == 0, "unroll by two");

    const int tid = threadIdx.x;
    const int lane = tid & 63, li = lane & 31, lh = lane >> 5;
    const int wave = tid >> 6, wn = wave % WAVES_N, wp = wave / WAVES_N;
    const int chunk = blockIdx.x;
    const int k0 = blockIdx.y * KT, n0 = blockIdx.z * NT;
    const long cbeg = (long)chunk * g.pch;
    const long cend = (cbeg + g.pch < g.M) ? cbeg + g.pch : g.M;

    floatx16 acc[NTAPS][TKW][TNW];
#pragma unroll
    for (int t = 0; t < NTAPS; ++t)
#pragma unroll
        for (int a = 0; a < TKW; ++a)
#pragma unroll
            for (int b = 0; b < TNW; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][a][b][r] = 0.f;

    for (long ps0 = cbeg; ps0 < cend; ps0 += PS) {
        float zr[NPW][TNW];
#pragma unroll
        for (int j = 0; j < NPW; ++j)
#pragma unroll
            for (int b = 0; b < TNW; ++b) {
                const long grow = ps0 + 2 * wp + j * RS + lh;
                const int nn = n0 + (wn * TNW + b) * 32 + li;
                zr[j][b] = (grow < cend && nn < g.N) ? g.Z[grow * g.ldz + nn] : 0.f;
            }
        __syncthreads();
        // batches of SB independent loads before the LDS writes: a plain "load; store" loop makes hipcc wait
        // for every load before it issues the next (one exposed memory latency per float4)
        constexpr int SB = 4;
        for (int base = 0; base < arows * (KT / 4); base += SB * 256) {
            float4 t[SB];
#pragma unroll
            for (int i = 0; i < SB; ++i) {
                const int f = base + tid + i * 256;
                const int row = f / (KT / 4), c4 = f - row * (KT / 4);
                const long grow = ps0 - halo + row;
                const int kk = k0 + c4 * 4;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (f < arows * (KT / 4) && grow >= g.rmin && grow < g.rmax && kk < g.K)
                    v = *(const float4*)(g.A + grow * g.lda + kk);
                t[i] = v;
            }
#pragma unroll
            for (int i = 0; i < SB; ++i) {
                const int f = base + tid + i * 256;
                const int row = f / (KT / 4), c4 = f - row * (KT / 4);
                if (f < arows * (KT / 4)) *(float4*)(As + row * KT + c4 * 4) = t[i];
            }
        }
        __syncthreads();
        // A operands: LDS reads of the next pixel pair are issued before the MFMAs of the current pair (two named
        // register sets, statically unrolled); B operands are the zr registers.
        {
            float a0[NTAPS][TKW], a1[NTAPS][TKW];
            auto fetch = [&](float (&an)[NTAPS][TKW], int r) {
#pragma unroll
                for (int t = 0; t < NTAPS; ++t) {
                    const int off = (NTAPS == 9) ? ((t / 3) - 1) * g.WP + (t % 3) - 1 : 0;
#pragma unroll
                    for (int a = 0; a < TKW; ++a) an[t][a] = As[(r + lh + halo + off) * KT + li + a * 32];
                }
            };
            auto fma_all = [&](const float (&ac)[NTAPS][TKW], const float (&bc)[TNW]) {
#pragma unroll
                for (int t = 0; t < NTAPS; ++t)
#pragma unroll
                    for (int a = 0; a < TKW; ++a)
#pragma unroll
                        for (int b = 0; b < TNW; ++b)
                            acc[t][a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[t][a], bc[b], acc[t][a][b], 0, 0, 0);
            };
            fetch(a0, 2 * wp);
#pragma unroll
            for (int j = 0; j < NPW; j += 2) {
                fetch(a1, 2 * wp + (j + 1) * RS);
                fma_all(a0, zr[j]);
                if (j + 2 < NPW) fetch(a0, 2 * wp + (j + 2) * RS);
                fma_all(a1, zr[j + 1]);
            }
        }
    }

    // fold the pixel-split waves of this block through LDS, one tap at a time (keeps the
    // transfer at 16 registers per lane) and in a fixed order wp = 1, 2, ...
    if (WAVES_P > 1) {
        float* red = smem;   // [WAVES_P-1][WAVES_N][TKW*TNW*16][64]
#pragma unroll
        for (int t = 0; t < NTAPS; ++t) {
            __syncthreads();
            if (wp > 0) {
                float* dst = red + (((wp - 1) * WAVES_N + wn) * (TKW * TNW * 16)) * 64 + lane;
#pragma unroll
                for (int a = 0; a < TKW; ++a)
#pragma unroll
                    for (int b = 0; b < TNW; ++b)
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            dst[((a * TNW + b) * 16 + r) * 64] = acc[t][a][b][r];
            }
            __syncthreads();
            if (wp == 0) {
                for (int src = 1; src < WAVES_P; ++src) {
                    const float* sp = red + (((src - 1) * WAVES_N + wn) * (TKW * TNW * 16)) * 64 + lane;
#pragma unroll
                    for (int a = 0; a < TKW; ++a)
#pragma unroll
                        for (int b = 0; b < TNW; ++b)
#pragma unroll
                            for (int r = 0; r < 16; ++r)
                                acc[t][a][b][r] += sp[((a * TNW + b) * 16 + r) * 64];
                }
            }
        }
    }
    if (wp != 0) return;

    float* out = g.out + (long)chunk * g.slab;
#pragma unroll
    for (int t = 0; t < NTAPS; ++t)
#pragma unroll
        for (int a = 0; a < TKW; ++a)
#pragma unroll
            for (int b = 0; b < TNW; ++b) {
                const int n = n0 + (wn * TNW + b) * 32 + li;
                if (n >= g.N) continue;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int k = k0 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (k < g.K) out[((long)t * g.K + k) * g.N + n] = acc[t][a][b][r];
                }
            }
}

// ---- v6: as v1, but the runs arrive by LDS-DMA (global_load_lds_dwordx4: global -> LDS with no register in between)
// into TWO run buffers: the DMA of run i+1 is issued before the MFMAs of run i and has the whole run (9-18k matrix-pipe
// cycles) to land, so a wave never waits on memory -- v1 spends a third of its time in its two staging barriers with the
// matrix pipe idle (PMC: busy 0.62-0.67), and 144 accumulator registers leave no room for a register prefetch.  One barrier
// per run.  A DMA piece is one wave instruction = 64 lanes x 16 B = 1 KiB, contiguous in LDS (destination = wave-uniform
// base + 16 * lane) with a per-lane SOURCE address: 8 rows of 32 channels, or 2 rows of 128, per piece; the images are the
// plain [row][channel] tiles v1 reads (conflict-free ds_read_b32: 32 consecutive floats per half wave).  Needs whole
// k / n tiles (K % KT == 0, N % NT == 0) and sources whose rows are all inside their allocation: true for every full run
// (planes carry W + 3 zero guard pixels >= the halo); the ONE ragged run at the end of the pixel axis is staged through
// registers with masks, like v1.
typedef __attribute__((address_space(3))) float wg_lds_f;
typedef const __attribute__((address_space(1))) float wg_glb_f;

// Body of tap_wgrad_kernel_v6 for the taps T0 .. T1 - 1 of a workgroup of NW waves (4: all nine taps per wave, the round-2 form;
// 8: tap_wgrad_kernel_v6s, where the two waves of a SIMD share a weight block and split its taps 5 + 4).
template <int NTAPS, int TKW, int WAVES_N, int TNW, int PS, int T0, int T1, int NW>
__device__ __forceinline__ void v6_body(const WgradArgs& g, float* smem) {
    constexpr int NTL = T1 - T0;                    // taps of this wave
    constexpr int WAVES_P = 4 / WAVES_N;
    constexpr int KT = TKW * 32, NT = WAVES_N * TNW * 32;

    const int halo = g.halo;
    const int arows = PS + 2 * halo;
    const int atotal = arows * (KT / 4);            // float4 of the A image
    const int asz = (arows * KT + 255) & ~255;      // floats, rounded up to whole 1-KiB pieces
    const int bufsz = asz + PS * NT;

    const int tid = threadIdx.x;
    const int lane = tid & 63, li = lane & 31, lh = lane >> 5;
    const int wave8 = tid >> 6;                      // DMA piece owner (0 .. NW - 1)
    const int wave = wave8 & 3, wn = wave % WAVES_N, wp = wave / WAVES_N;
    const int chunk = blockIdx.x;
    const int k0 = blockIdx.y * KT, n0 = blockIdx.z * NT;
    const long cbeg = (long)chunk * g.pch;
    const long cend = (cbeg + g.pch < g.M) ? cbeg + g.pch : g.M;

    floatx16 acc[NTL][TKW][TNW];
#pragma unroll
    for (int t = T0; t < T1; ++t)
#pragma unroll
        for (int a = 0; a < TKW; ++a)
#pragma unroll
            for (int b = 0; b < TNW; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t - T0][a][b][r] = 0.f;

    constexpr int ARP = 256 / KT, ZRP = 256 / NT;   // rows per DMA piece
    const int apieces = (atotal + 63) >> 6;
    constexpr int zpieces = PS * (NT / 4) / 64;
    static_assert((PS * (NT / 4)) % 64 == 0, "whole dZ pieces");
    const int arow_l = lane / (KT / 4), ac4 = lane % (KT / 4);
    const int zrow_l = lane / (NT / 4), zc4 = lane % (NT / 4);

    auto stage_dma = [&](long ps0, float* buf) {
        // channels past K / N (ragged last tile) are not fetched: their LDS words keep whatever they held and only feed
        // accumulators of weight rows / columns that are never stored
        const float* asrc = g.A + (ps0 - halo + arow_l) * (long)g.lda + k0 + ac4 * 4;
        const bool aok = k0 + ac4 * 4 < g.K, zok = n0 + zc4 * 4 < g.N;
        for (int p = wave8; p < apieces; p += NW)
            if (aok && p * 64 + lane < atotal)
                __builtin_amdgcn_global_load_lds((wg_glb_f*)(asrc + (long)p * ARP * g.lda), (wg_lds_f*)(buf + p * 256), 16, 0, 0);
        const float* zsrc = g.Z + (ps0 + zrow_l) * (long)g.ldz + n0 + zc4 * 4;
        float* zb = buf + asz;
#pragma unroll
        for (int p = wave8; p < zpieces; p += NW)
            if (zok)
                __builtin_amdgcn_global_load_lds((wg_glb_f*)(zsrc + (long)p * ZRP * g.ldz), (wg_lds_f*)(zb + p * 256), 16, 0, 0);
    };
    auto stage_masked = [&](long ps0, float* buf) {     // ragged last run: rows past the end read as zero
        float* Zs = buf + asz;
        for (int f = tid; f < atotal; f += 64 * NW) {
            const int row = f / (KT / 4), c4 = f - row * (KT / 4);
            const long grow = ps0 - halo + row;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (grow >= g.rmin && grow < g.rmax && k0 + c4 * 4 < g.K) v = *(const float4*)(g.A + grow * g.lda + k0 + c4 * 4);
            *(float4*)(buf + row * KT + c4 * 4) = v;
        }
        for (int f = tid; f < PS * (NT / 4); f += 64 * NW) {
            const int row = f / (NT / 4), n4 = f - row * (NT / 4);
            const long grow = ps0 + row;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (grow < cend && n0 + n4 * 4 < g.N) v = *(const float4*)(g.Z + grow * g.ldz + n0 + n4 * 4);
            *(float4*)(Zs + row * NT + n4 * 4) = v;
        }
    };
    auto stage = [&](long ps0, float* buf) {
        if (ps0 + PS <= g.M) stage_dma(ps0, buf); else stage_masked(ps0, buf);
    };

    stage(cbeg, smem);
    __syncthreads();                     // vmcnt(0) + barrier: run 0 has landed for every wave
    int cur = 0;
    for (long ps0 = cbeg; ps0 < cend; ps0 += PS) {
        const float* As = smem + cur * bufsz;
        const float* Zs = As + asz;
        // the other buffer was last read in the previous iteration, which every wave has left (barrier below)
        if (ps0 + PS < cend) stage(ps0 + PS, smem + (cur ^ 1) * bufsz);
        {
            // fully unrolled run (compile-time trip count): pixel offsets become immediates and the two operand sets are
            // plain renamed registers -- the rolled form of v1 carries them through v_mov copies behind an lgkmcnt(0)
            float a0[NTL][TKW], b0[TNW], a1[NTL][TKW], b1[TNW];
            constexpr int RS = 2 * WAVES_P;             // pixel stride between this wave's pairs
            constexpr int NP = PS / RS;                 // pairs per wave and run
            static_assert(NP % 2 == 0, "unroll by two");
            const float* Zw = Zs + (2 * wp + lh) * NT + wn * TNW * 32 + li;
            const float* Aw = As + (2 * wp + lh + halo) * KT + li;
            int toff[NTL];
#pragma unroll
            for (int t = T0; t < T1; ++t) toff[t - T0] = ((NTAPS == 9) ? ((t / 3) - 1) * g.WP + (t % 3) - 1 : 0) * KT;
            auto fetch = [&](float (&an)[NTL][TKW], float (&bn)[TNW], int r) {
#pragma unroll
                for (int b = 0; b < TNW; ++b) bn[b] = Zw[r * NT + b * 32];
#pragma unroll
                for (int t = T0; t < T1; ++t)
#pragma unroll
                    for (int a = 0; a < TKW; ++a) an[t - T0][a] = Aw[toff[t - T0] + r * KT + a * 32];
            };
            auto fma_all = [&](const float (&ac)[NTL][TKW], const float (&bc)[TNW]) {
#pragma unroll
                for (int t = T0; t < T1; ++t)
#pragma unroll
                    for (int a = 0; a < TKW; ++a)
#pragma unroll
                        for (int b = 0; b < TNW; ++b)
                            acc[t - T0][a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[t - T0][a], bc[b], acc[t - T0][a][b], 0, 0, 0);
            };
            fetch(a0, b0, 0);
#pragma unroll
            for (int i = 0; i < NP; i += 2) {
                fetch(a1, b1, (i + 1) * RS);
                fma_all(a0, b0);
                if (i + 2 < NP) fetch(a0, b0, (i + 2) * RS);
                fma_all(a1, b1);
            }
        }
        __syncthreads();                 // this wave's DMA of the next run has landed (vmcnt(0)), then all waves meet
        cur ^= 1;
    }

    if (WAVES_P > 1) {
        // every wave takes every barrier; a wave of the tap-split form only moves data for its own taps (tq)
        float* red = smem;   // [WAVES_P-1][WAVES_N][TKW*TNW*16][64]
#pragma unroll
        for (int tq = 0; tq < NTAPS; ++tq) {
            const bool mine = tq >= T0 && tq < T1;
            const int t = mine ? tq : T0;
            __syncthreads();
            if (mine && wp > 0) {
                float* dst = red + (((wp - 1) * WAVES_N + wn) * (TKW * TNW * 16)) * 64 + lane;
#pragma unroll
                for (int a = 0; a < TKW; ++a)
#pragma unroll
                    for (int b = 0; b < TNW; ++b)
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            dst[((a * TNW + b) * 16 + r) * 64] = acc[t - T0][a][b][r];
            }
            __syncthreads();
            if (mine && wp == 0) {
                for (int src = 1; src < WAVES_P; ++src) {
                    const float* sp = red + (((src - 1) * WAVES_N + wn) * (TKW * TNW * 16)) * 64 + lane;
#pragma unroll
                    for (int a = 0; a < TKW; ++a)
#pragma unroll
                        for (int b = 0; b < TNW; ++b)
#pragma unroll
                            for (int r = 0; r < 16; ++r)
                                acc[t - T0][a][b][r] += sp[((a * TNW + b) * 16 + r) * 64];
                }
            }
        }
    }
    if (wp != 0) return;

    float* out = g.out + (long)chunk * g.slab;
#pragma unroll
    for (int t = T0; t < T1; ++t)
#pragma unroll
        for (int a = 0; a < TKW; ++a)
#pragma unroll
            for (int b = 0; b < TNW; ++b) {
                const int n = n0 + (wn * TNW + b) * 32 + li;
                if (n >= g.N) continue;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int k = k0 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (k < g.K) out[((long)t * g.K + k) * g.N + n] = acc[t - T0][a][b][r];
                }
            }
}

template <int NTAPS, int TKW, int WAVES_N, int TNW, int PS, int MINB>
__global__ __launch_bounds__(256, MINB) void tap_wgrad_kernel_v6(WgradArgs g) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    v6_body<NTAPS, TKW, WAVES_N, TNW, PS, 0, NTAPS, 4>(g, smem);
}

// Eight waves: waves w and w + 4 share a SIMD and a 32 x 32 weight block and split its nine taps 5 + 4 (80 / 64 accumulator
// registers), so that a wave held at a DMA piece or an LDS read leaves the matrix pipe to its partner -- with one wave per
// SIMD the DMA of the next run costs 5-19 % of the kernel and cannot be moved out of the way.
template <int TKW, int WAVES_N, int TNW, int PS>
__global__ __launch_bounds__(512, 1) void tap_wgrad_kernel_v6s(WgradArgs g) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    if ((threadIdx.x >> 8) == 0) v6_body<9, TKW, WAVES_N, TNW, PS, 0, 5, 8>(g, smem);
    else v6_body<9, TKW, WAVES_N, TNW, PS, 5, 9, 8>(g, smem);
}

__global__ void sum_chunks_kernel(const float* __restrict__ part, float* __restrict__ out,
                                  long n, int nchunks) {
    const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= n) return;
    // four independent chains over the chunks (loads in flight together), folded in a fixed order
    float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0, s2 = s0, s3 = s0;
    int c = 0;
    for (; c + 4 <= nchunks; c += 4) {
        const float4 v0 = *(const float4*)(part + (long)c * n + i);
        const float4 v1 = *(const float4*)(part + (long)(c + 1) * n + i);
        const float4 v2 = *(const float4*)(part + (long)(c + 2) * n + i);
        const float4 v3 = *(const float4*)(part + (long)(c + 3) * n + i);
        s0.x += v0.x; s0.y += v0.y; s0.z += v0.z; s0.w += v0.w;
        s1.x += v1.x; s1.y += v1.y; s1.z += v1.z; s1.w += v1.w;
        s2.x += v2.x; s2.y += v2.y; s2.z += v2.z; s2.w += v2.w;
        s3.x += v3.x; s3.y += v3.y; s3.z += v3.z; s3.w += v3.w;
    }
    for (; c < nchunks; ++c) {
        const float4 v = *(const float4*)(part + (long)c * n + i);
        s0.x += v.x; s0.y += v.y; s0.z += v.z; s0.w += v.w;
    }
    *(float4*)(out + i) = make_float4((s0.x + s1.x) + (s2.x + s3.x), (s0.y + s1.y) + (s2.y + s3.y),
                                      (s0.z + s1.z) + (s2.z + s3.z), (s0.w + s1.w) + (s2.w + s3.w));
}

struct Plan { int ktile, ntile, ps, nchunks, pch; };

Plan make_plan(const asr_gemm_desc* d, int target_blocks = 768, long cap_mb = 64, int ps_override = 0) {
    Plan p;
    if (d->ntaps != 1) {
        p.ktile = 32;
        p.ntile = d->N > 64 ? 128 : (d->N > 32 ? 64 : 32);
        p.ps = d->N > 32 ? 64 : 128;
    } else {
        p.ktile = 128; p.ntile = 128; p.ps = 64;
    }
    if (ps_override) p.ps = ps_override;
    const long tiles = (long)asr_cdiv(d->K, p.ktile) * asr_cdiv(d->N, p.ntile);
    long want = target_blocks / tiles;
    if (want < 1) want = 1;
    const long slab_bytes = (long)d->ntaps * d->K * d->N * 4;
    const long cap = (cap_mb << 20) / slab_bytes;      // fewer, longer chunks beat more slabs: 64 MB measured better than 128 for fp32
    if (want > cap) want = cap < 1 ? 1 : cap;
    // at least four runs per chunk -- except for the few-tile outputs of narrow dense layers (the 128-wide language half of
    // the joint graph: 1-4 tiles), where that leaves 25 workgroups on 256 CUs: one run per chunk there
    const long maxc = asr_cdiv(d->M, (d->ntaps == 1 && tiles <= 4 ? 1 : 4) * p.ps);
    if (want > maxc) want = maxc;
    if (want < 1) want = 1;
    long pch = ((long)asr_cdiv(d->M, want) + p.ps - 1) / p.ps * p.ps;
    p.pch = (int)pch;
    p.nchunks = asr_cdiv(d->M, pch);
    return p;
}

template <int NTAPS, int TKW, int WAVES_N, int TNW, int PS>
int launch_wgrad(const WgradArgs& a, const Plan& p, int K, int N, hipStream_t st) {
    // The register-staged generation, for what the LDS-DMA kernels below do not take (4-tap phase-split convs, operands
    // that are not 16-byte aligned, dense outputs of fewer than four tiles): 3x3 / 4-tap on v1 (batched direct staging,
    // 2 workgroups per CU), dense / 1x1 on v3 (dZ in registers, half-length runs: 10-30 % faster than v1 there).
    constexpr bool V3 = (NTAPS == 1);
    constexpr int PS3 = PS / 2;          // v3 keeps the run's dZ in registers: half the run length
    constexpr int KT = TKW * 32, NT = WAVES_N * TNW * 32;
    size_t lds = V3 ? (size_t)(PS3 + 2 * a.halo) * KT * sizeof(float)
                    : ((size_t)(PS + 2 * a.halo) * KT + (size_t)PS * NT) * sizeof(float);
    const size_t red = (size_t)(4 / WAVES_N - 1) * WAVES_N * TKW * TNW * 16 * 64 * sizeof(float);
    if (red > lds) lds = red;
    if (lds > 160 * 1024) return ASR_ERR_UNSUPPORTED;
    const dim3 grid(p.nchunks, asr_cdiv(K, KT), asr_cdiv(N, NT));
    static bool attr_set = false;
    if constexpr (V3) {
        auto kern = tap_wgrad_kernel_v3<NTAPS, TKW, WAVES_N, TNW, PS3>;
        if (!attr_set) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr_set = true; }
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, a);
        ASR_NOTE_KERNEL("tap_wgrad_kernel_v3<%d, %d, %d, %d, %d>", NTAPS, TKW, WAVES_N, TNW, PS3);
    } else {
        auto kern = tap_wgrad_kernel_v1<NTAPS, TKW, WAVES_N, TNW, PS>;
        if (!attr_set) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr_set = true; }
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, a);
        ASR_NOTE_KERNEL("tap_wgrad_kernel_v1<%d, %d, %d, %d, %d>", NTAPS, TKW, WAVES_N, TNW, PS);
    }
    ASR_CHECK_LAUNCH("tap_wgrad");
    return ASR_OK;
}

// v6 (LDS-DMA, two run buffers): run length and workgroups per CU from what fits the 160 KB of LDS -- two workgroups per
// CU with 32-pixel runs when two double-buffered images fit in 80 KB each, else one with 64- or 32-pixel runs -- and a
// grid of exactly one round of resident workgroups (512 or 256): every workgroup runs beside the same neighbours from
// start to end, no partial round.
struct Plan6 { Plan p; int ps, minb; size_t lds; };

template <int KT, int NT>
Plan6 make_plan6(const asr_gemm_desc* d) {
    const int halo = (d->ntaps != 1) ? d->W + 2 : 0;
    auto lds_for = [&](int ps) { return (size_t)2 * ((((size_t)(ps + 2 * halo) * KT + 255) & ~(size_t)255) + (size_t)ps * NT) * sizeof(float); };
    Plan6 q;
    auto fits = [&](int ps) { return lds_for(ps) <= 160 * 1024; };
    // measured on the DFCNN layers (tools/bench_layers.py wgrad, gpurun_out/r02b): with <= 64 output channels the A image and
    // its halo (two plane rows: 2 x (W + 2) pixels re-staged per run) dominate, so longer runs: 64 pixels, or 128 on planes
    // wider than 64 (800x100: 88 / 114 TFLOP/s at N = 32 / 64 against 78 / 103 with 32-pixel runs)
    // >= 128 outputs, 3x3: 64-pixel runs at ONE workgroup per CU tie with 32-pixel runs at two in isolation (127 vs 126
    // TFLOP/s) but leave half of every CU's registers and wave slots to the HBM-bound kernels that run beside the weight
    // gradient on the other stream, and halve the partial slab (256 workgroups): whole step -1 % (gpurun_out/r02g/ps.log)
    int ps = (d->ntaps == 1) ? 32 : (NT > 64 ? 64 : (d->W > 64 ? 128 : 64));
    while (ps > 32 && !fits(ps)) ps >>= 1;
    q.ps = ps;
    q.minb = (2 * lds_for(ps) <= 160 * 1024) ? 2 : 1;
    q.lds = lds_for(q.ps);
    q.p = make_plan(d, 256 * q.minb, 128, q.ps);
    q.p.ktile = KT; q.p.ntile = NT;
    if (d->ntaps == 1) {
        // Dense layers: the chunk count is picked from the beat model (a CU finishes its workgroups at a fixed aggregate
        // rate, so a grid costs ceil(workgroups / 256) beats of one chunk each) plus the price of summing the slabs:
        //   t(nc) = ceil(tiles * nc / 256) / nc * t_rows + nc * slab traffic
        // e.g. 6400 x 6400 x 1536 (600 tiles): one chunk = 3 beats, two chunks = 5 half-beats (1388 -> 1156 us measured);
        // 32768 x 512 x 6348 (200 tiles): 2 chunks 2266 us, 15 chunks 1894 us; the 16- and 64-tile projection / FFN shapes
        // keep 512 workgroups (tools/bench_wgrad1.py)
        const long tiles = (long)asr_cdiv(d->K, KT) * asr_cdiv(d->N, NT);
        const double slab = (double)d->K * d->N * 4.0;
        const double t_rows = (double)d->M * KT * NT * 2.0 / (0.75 * 157.3e12 / 256);
        long cap = (long)((128.0 * (1 << 20)) / slab); if (cap < 1) cap = 1;
        long maxc = asr_cdiv(d->M, (tiles <= 4 ? 1 : 4) * q.ps); if (maxc < 1) maxc = 1;
        long lim = cap < maxc ? cap : maxc; if (lim > 512) lim = 512;
        auto cost = [&](long nc) { return (double)asr_cdiv(tiles * nc, 256) / nc * t_rows + (nc > 1 ? (nc + 1) * slab / 3.0e12 : 0.0); };
        long best = 1; double tbest = 1e30;
        for (long nc = 1; nc <= lim; ++nc) {
            const double t = cost(nc);
            if (t < tbest * 0.995) { tbest = t; best = nc; }      // fewer chunks on (near) ties
        }
        // ... but not fewer workgroups than fit side by side (two per CU cover each other's staging) when more cost the same
        while (tiles * best < 256L * q.minb && 2 * best <= lim && cost(2 * best) <= 1.06 * tbest) best *= 2;
        long pch = ((long)asr_cdiv(d->M, best) + q.ps - 1) / q.ps * q.ps;
        q.p.pch = (int)pch;
        q.p.nchunks = asr_cdiv(d->M, pch);
    }
    return q;
}

template <int NTAPS, int TKW, int WAVES_N, int TNW>
bool v6_ok(const asr_gemm_desc* d, int ldz) {
    constexpr int KT = TKW * 32, NT = WAVES_N * TNW * 32;
    if ((d->K & 3) || (d->N & 3) || (d->lda & 3) || (ldz & 3)) return false;
    const Plan6 q = make_plan6<KT, NT>(d);
    const size_t red = (size_t)(4 / WAVES_N - 1) * WAVES_N * TKW * TNW * 16 * 64 * sizeof(float);
    return q.lds <= 160 * 1024 && red <= q.lds;
}

template <int NTAPS, int TKW, int WAVES_N, int TNW>
int launch_wgrad6(WgradArgs a, const asr_gemm_desc* d, float* dW, float* partials, hipStream_t st, Plan* used) {
    constexpr int KT = TKW * 32, NT = WAVES_N * TNW * 32;
    const Plan6 q = make_plan6<KT, NT>(d);
    *used = q.p;
    if (q.p.nchunks > 1 && !partials) return ASR_ERR_BAD_ARG;
    a.out = (q.p.nchunks > 1) ? partials : dW;
    a.pch = q.p.pch;
    const dim3 grid(q.p.nchunks, asr_cdiv(d->K, KT), asr_cdiv(d->N, NT));
#define ASR_V6_LAUNCH(PSV, MB)                                                                                            \
    do {                                                                                                                 \
        auto kern = tap_wgrad_kernel_v6<NTAPS, TKW, WAVES_N, TNW, PSV, MB>;                                               \
        static bool attr = false;                                                                                        \
        if (!attr) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; } \
        hipLaunchKernelGGL(kern, grid, dim3(256), q.lds, st, a);                                                         \
        ASR_NOTE_KERNEL("tap_wgrad_kernel_v6<%d, %d, %d, %d, %d, %d>", NTAPS, TKW, WAVES_N, TNW, PSV, MB);                \
    } while (0)
    if constexpr (NTAPS == 9) {
#define ASR_V6S_LAUNCH(PSV)                                                                                               \
        do {                                                                                                             \
            auto kern = tap_wgrad_kernel_v6s<TKW, WAVES_N, TNW, PSV>;                                                     \
            static bool attr = false;                                                                                    \
            if (!attr) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; } \
            hipLaunchKernelGGL(kern, grid, dim3(512), q.lds, st, a);                                                     \
            ASR_NOTE_KERNEL("tap_wgrad_kernel_v6s<%d, %d, %d, %d>", TKW, WAVES_N, TNW, PSV);                              \
            ASR_CHECK_LAUNCH("tap_wgrad_v6s");                                                                           \
            return ASR_OK;                                                                                               \
        } while (0)
        // one workgroup per CU (minb 1): eight waves instead of four, two per SIMD
        if (q.minb == 1 && q.ps == 64) ASR_V6S_LAUNCH(64);
        if (q.minb == 1 && q.ps == 128) ASR_V6S_LAUNCH(128);
#undef ASR_V6S_LAUNCH
    }
    if (q.ps == 128) { if (q.minb == 2) ASR_V6_LAUNCH(128, 2); else ASR_V6_LAUNCH(128, 1); }
    else if (q.ps == 32 && q.minb == 2) ASR_V6_LAUNCH(32, 2);
    else if (q.ps == 32) ASR_V6_LAUNCH(32, 1);
    else if (q.minb == 2) ASR_V6_LAUNCH(64, 2);
    else ASR_V6_LAUNCH(64, 1);
#undef ASR_V6_LAUNCH
    ASR_CHECK_LAUNCH("tap_wgrad_v6");
    return ASR_OK;
}

}  // namespace

// gemm1.hip: the dense (1-tap) weight gradient on buffer-form LDS-DMA; this file keeps the chunk plan and the slab sum
bool asr_wgrad1_eligible(const asr_gemm_desc* d, const float* A, const float* dZ, int ldz);
int asr_wgrad1_launch(const asr_gemm_desc* d, const float* A, const float* dZ, int ldz, float* out, int pch, int nchunks, void* stream);

size_t asr_wino_wgrad_workspace(const asr_gemm_desc* d, int ldz);
int asr_wino_wgrad_launch(const asr_gemm_desc* d, const float* A, const float* dZ, int ldz, float* dW, float* partials, void* stream);

static size_t direct_wgrad_workspace(const asr_gemm_desc* d);
extern "C" size_t asr_tap_wgrad_workspace(const asr_gemm_desc* d) {
    if (!d) return 0;
    const size_t a = direct_wgrad_workspace(d), b = d->ntaps == 9 ? asr_wino_wgrad_workspace(d, d->N) : 0;
    return a > b ? a : b;
}
static size_t direct_wgrad_workspace(const asr_gemm_desc* d) {
    const Plan p = make_plan(d), q = make_plan(d, 1024, 128);  // q: an upper bound on the chunks of any variant
    int nc = p.nchunks > q.nchunks ? p.nchunks : q.nchunks;
    if (d->ntaps == 9) {                                       // the LDS-DMA variant plans its own (shorter) runs
        const int n6 = d->N > 64 ? make_plan6<32, 128>(d).p.nchunks : d->N > 32 ? make_plan6<32, 64>(d).p.nchunks
                                                                                 : make_plan6<32, 32>(d).p.nchunks;
        if (n6 > nc) nc = n6;
    } else if (d->ntaps == 1) {
        const int n6 = d->N <= 32 ? make_plan6<128, 32>(d).p.nchunks : make_plan6<128, 128>(d).p.nchunks;
        if (n6 > nc) nc = n6;
    }
    if (nc <= 1) return 16;
    return (size_t)nc * d->ntaps * d->K * d->N * sizeof(float);
}

static int tap_wgrad_impl(const asr_gemm_desc* d, const float* A, const float* dZ, int ldz, float* dW, float* partials, void* stream,
                          bool allow_winograd);
extern "C" int asr_tap_wgrad(const asr_gemm_desc* d, const float* A, const float* dZ, int ldz,
                             float* dW, float* partials, void* stream) {
    return tap_wgrad_impl(d, A, dZ, ldz, dW, partials, stream, true);
}
// the direct (36 multiplies per tile) kernels only: the second, independent implementation the Winograd path is tested against
extern "C" int asr_tap_wgrad_direct(const asr_gemm_desc* d, const float* A, const float* dZ, int ldz,
                                    float* dW, float* partials, void* stream) {
    return tap_wgrad_impl(d, A, dZ, ldz, dW, partials, stream, false);
}
static int tap_wgrad_impl(const asr_gemm_desc* d, const float* A, const float* dZ, int ldz, float* dW, float* partials, void* stream,
                          bool allow_winograd) {
    if (!d || !A || !dZ || !dW) return ASR_ERR_BAD_ARG;
    if (d->ntaps != 1 && d->ntaps != 9 && d->ntaps != 4) return ASR_ERR_BAD_ARG;
    if ((d->K & 3) || (d->N & 3) || (d->lda & 3) || (ldz & 3)) return ASR_ERR_BAD_ARG;
    if (d->ntaps != 1 && d->H <= 0) return ASR_ERR_BAD_ARG;
    if (d->ntaps == 4 && (d->K & 255)) return ASR_ERR_UNSUPPORTED;      // phase blocks of K/4 channels, whole 32-wide k-tiles
    // 3x3 layers with 32 or a multiple of 64 input channels and a multiple of 64 output channels: Winograd F(3x3, 2x2),
    // wino_wgrad.hip (16 instead of 36 multiplies per tile and channel pair); ASR_ERR_UNSUPPORTED = not that shape
    if (allow_winograd && d->ntaps == 9 && partials) {
        const int rw = asr_wino_wgrad_launch(d, A, dZ, ldz, dW, partials, stream);
        if (rw != ASR_ERR_UNSUPPORTED) return rw;
    }
    const Plan p = make_plan(d);
    if (p.nchunks > 1 && !partials) return ASR_ERR_BAD_ARG;
    WgradArgs a;
    a.A = A; a.Z = dZ; a.out = (p.nchunks > 1) ? partials : dW;
    a.M = d->M; a.K = d->K; a.N = d->N; a.lda = d->lda; a.ldz = ldz;
    a.WP = d->W + 1;
    a.halo = (d->ntaps != 1) ? a.WP + 1 : 0;
    a.rmin = -(long)a.halo; a.rmax = (long)d->M + a.halo;
    a.pch = p.pch;
    a.slab = (long)d->ntaps * d->K * d->N;
    hipStream_t st = (hipStream_t)stream;
    int rc;
    Plan p6 = p;
    // the LDS-DMA kernels fetch 16 bytes per lane: both operands must be 16-byte aligned (row pitches are checked above)
    if (d->ntaps == 1 && asr_wgrad1_eligible(d, A, dZ, ldz)) {
        // dense layers and 1x1 convs: wgrad1_kernel (gemm1.hip), 128 x 128 tiles, or 128 x 32 for outputs of up to 32 channels
        const Plan6 q = d->N <= 32 ? make_plan6<128, 32>(d) : make_plan6<128, 128>(d);
        if (q.p.nchunks > 1 && !partials) return ASR_ERR_BAD_ARG;
        rc = asr_wgrad1_launch(d, A, dZ, ldz, q.p.nchunks > 1 ? partials : dW, q.p.pch, q.p.nchunks, stream);
        if (rc != ASR_OK) return rc;
        if (q.p.nchunks > 1) {
            hipLaunchKernelGGL(sum_chunks_kernel, dim3(asr_cdiv(asr_cdiv(a.slab, 4), 64)), dim3(64), 0, st, partials, dW, a.slab, q.p.nchunks);
            ASR_CHECK_LAUNCH("sum_chunks");
        }
        return ASR_OK;
    }
    bool is6 = ((((uintptr_t)A) | ((uintptr_t)dZ)) & 15) == 0;
    if (!is6) { /* fall through to the register-staged kernels */ }
    else if (d->ntaps == 9 && d->N > 64 && v6_ok<9, 1, 4, 1>(d, ldz)) rc = launch_wgrad6<9, 1, 4, 1>(a, d, dW, partials, st, &p6);
    else if (d->ntaps == 9 && d->N > 32 && d->N <= 64 && v6_ok<9, 1, 2, 1>(d, ldz)) rc = launch_wgrad6<9, 1, 2, 1>(a, d, dW, partials, st, &p6);
    else if (d->ntaps == 9 && d->N <= 32 && v6_ok<9, 1, 1, 1>(d, ldz)) rc = launch_wgrad6<9, 1, 1, 1>(a, d, dW, partials, st, &p6);
    else if (d->ntaps == 1 && (long)asr_cdiv(d->K, 128) * asr_cdiv(d->N, 128) >= 4 && v6_ok<1, 4, 4, 1>(d, ldz)) rc = launch_wgrad6<1, 4, 4, 1>(a, d, dW, partials, st, &p6);
    else is6 = false;
    if (is6) {
        if (rc != ASR_OK) return rc;
        if (p6.nchunks > 1) {
            const long n = a.slab;
            hipLaunchKernelGGL(sum_chunks_kernel, dim3(asr_cdiv(asr_cdiv(n, 4), 64)), dim3(64), 0, st, partials, dW, n, p6.nchunks);
            ASR_CHECK_LAUNCH("sum_chunks");
        }
        return ASR_OK;
    }
    if (d->ntaps == 9) {
        if (d->N > 64) rc = launch_wgrad<9, 1, 4, 1, 64>(a, p, d->K, d->N, st);
        else if (d->N > 32) rc = launch_wgrad<9, 1, 2, 1, 64>(a, p, d->K, d->N, st);
        else rc = launch_wgrad<9, 1, 1, 1, 128>(a, p, d->K, d->N, st);
    } else if (d->ntaps == 4) {
        if (d->N > 64) rc = launch_wgrad<4, 1, 4, 1, 64>(a, p, d->K, d->N, st);
        else if (d->N > 32) rc = launch_wgrad<4, 1, 2, 1, 64>(a, p, d->K, d->N, st);
        else rc = launch_wgrad<4, 1, 1, 1, 128>(a, p, d->K, d->N, st);
    } else {
        rc = launch_wgrad<1, 4, 4, 1, 64>(a, p, d->K, d->N, st);
    }
    if (rc != ASR_OK) return rc;
    if (p.nchunks > 1) {
        const long n = a.slab;
        const int threads = 64;
        const int blocks = asr_cdiv(asr_cdiv(n, 4), threads);
        hipLaunchKernelGGL(sum_chunks_kernel, dim3(blocks), dim3(threads), 0, st, partials, dW, n, p.nchunks);
        ASR_CHECK_LAUNCH("sum_chunks");
    }
    return ASR_OK;
}

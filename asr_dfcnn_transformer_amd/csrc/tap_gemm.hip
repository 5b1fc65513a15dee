// Tap-GEMM on the fp32 MFMA pipe (v_mfma_f32_32x32x2_f32): conv3x3 / conv1x1 forward,
// their data-gradients and the dense layers of the DFCNN head, in one kernel family.
//
//   acc[m][n] = sum_tap sum_k A[m + off(tap)][k] * Wt(tap,k,n)
//
// Design (MI355X-first, see DESIGN.md "tap-GEMM"):
//   * activations live in "padded planes" so a 3x3 tap is a constant row offset in the
//     flattened pixel index: a block stages ONE contiguous run of (MT + 2*(W+3)) pixel
//     rows x 32 channels into LDS and reuses it for all 9 taps (9x less A traffic than
//     an im2col GEMM); neighbouring tiles run on the same XCD so halo rows hit its L2;
//   * the contraction index inside an 8-wide group is permuted (k = 4*half + step) so a
//     lane fetches the A operands of four MFMAs with one ds_read_b128;
//   * weights are read straight from the model's HWIO tensor; the data-gradient reads
//     the same tensor transposed through a (32+1)-pitch LDS tile -- no packed copies;
//   * the next weight tile is prefetched into registers while the MFMAs of the current
//     tap run (issue-early / write-late staging); 2-3 blocks per CU cover the rest;
//   * epilogue fuses bias + ReLU (+ the frozen-BN affine) and skips border pixels.
#include "asr_common.h"
#include "reduce.h"
#include "tap_epilogue.h"
#include <stdlib.h>

namespace {

// ---- v1: single-buffered W, direct A staging, two barriers per tap, 3 blocks per CU
template <int MT, int NT, int WM, int WN, int NTAPS, int WMODE, int KCV = 32>
__global__ __launch_bounds__(256) void tap_gemm_kernel_v1(TapGemmArgs g) {
    constexpr int KC = KCV, AP = KCV + 4;   // shadow the file-level chunk size: this generation is instantiated at 16 and 32
    constexpr int TM = MT / WM / 32, TN = NT / WN / 32;
    constexpr int WREG = KC * NT / 4 / 256;            // float4 per thread per weight tile
    constexpr int AREG = (NTAPS == 1) ? MT * (KC / 4) / 256 : 1;
    constexpr int WPITCH = (WMODE == 0) ? NT : (KC + 1);
    static_assert(WREG >= 1 && TM >= 1 && TN >= 1, "tile");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int halo = g.halo;
    const int arows = MT + 2 * halo;
    int* rowa = (int*)smem;
    int* rowy = rowa + MT;
    float* tile_lds = smem + 2 * MT;
    float* As = tile_lds;
    float* Ws = As + arows * AP;

    const int tid = threadIdx.x;
    const int lane = tid & 63, li = lane & 31, lh = lane >> 5;
    const int wave = tid >> 6, wm = wave / WN, wn = wave % WN;

    if (NTAPS == 1 && WMODE == 0 && g.ksplit > 1) {      // split-K: this grid row's slice of the contraction, raw partial sums out
        const int z = blockIdx.y, kper = g.K / g.ksplit;
        g.A += (long)z * kper; g.W += (long)z * kper * g.ldw; g.K = kper;
        g.out_a = g.split_out + (long)z * g.M * g.N; g.ldo_a = g.N; g.out_y = nullptr;
        g.bias = nullptr; g.scale = nullptr; g.shift = nullptr; g.relu = 0; g.accumulate = 0;
    }
    const int swz = asr_xcd_swizzle(blockIdx.x, gridDim.x);
    const int tile_m = swz / g.ntn, tile_n = swz - tile_m * g.ntn;
    const long p0 = (long)tile_m * MT;
    const int n0 = tile_n * NT;
    const int K = g.K, N = g.N;

    if (tid < MT) {
        const long p = p0 + tid;
        int ra = -1, ry = -1;
        if (p < g.M) {
            if (g.H == 0) {
                ra = (int)p; ry = (int)p;
            } else {
                const int b = (int)(p / g.HPWP);
                const int r = (int)(p - (long)b * g.HPWP);
                const int hh = r / g.WP, ww = r - hh * g.WP;
                if (hh >= 1 && hh <= g.H && ww >= 1 && ww <= g.Wd) {
                    ra = (int)p;
                    ry = g.y_unpadded ? ((b * g.H + hh - 1) * g.Wd + ww - 1) : (int)p;
                }
            }
        }
        rowa[tid] = ra; rowy[tid] = ry;
    }

    floatx16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int nkc = (K + KC - 1) / KC;
    const int nsteps = nkc * NTAPS;
    float4 wreg[WREG];
    float4 areg[AREG];

    auto load_w = [&](int step) {
        const int kc = step / NTAPS, tap = step - kc * NTAPS;
#pragma unroll
        for (int i = 0; i < WREG; ++i) {
            const int f = tid + i * 256;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (WMODE == 0) {
                const int k = f / (NT / 4), n4 = f - k * (NT / 4);
                const int kk = kc * KC + k, nn = n0 + n4 * 4;
                if (kk < K && nn < N)
                    v = *(const float4*)(g.W + ((long)tap * K + kk) * g.ldw + nn);
            } else {
                const int n = f / (KC / 4), k4 = f - n * (KC / 4);
                const int kk = kc * KC + k4 * 4, nn = n0 + n;
                if (kk < K && nn < N)
                    v = *(const float4*)(g.W + ((long)tap_weight_index<NTAPS, WMODE>(tap) * N + nn) * g.ldw + kk);
            }
            wreg[i] = v;
        }
    };
    auto store_w = [&]() {
#pragma unroll
        for (int i = 0; i < WREG; ++i) {
            const int f = tid + i * 256;
            if (WMODE == 0) {
                const int k = f / (NT / 4), n4 = f - k * (NT / 4);
                *(float4*)(Ws + k * NT + n4 * 4) = wreg[i];
            } else {
                const int n = f / (KC / 4), k4 = f - n * (KC / 4);
                float* d = Ws + n * (KC + 1) + k4 * 4;
                d[0] = wreg[i].x; d[1] = wreg[i].y; d[2] = wreg[i].z; d[3] = wreg[i].w;
            }
        }
    };
    auto load_a_row = [&](int f, int kc) -> float4 {
        const int row = f / (KC / 4), c4 = f - row * (KC / 4);
        const long grow = p0 - halo + row;
        const int kk = kc * KC + c4 * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (grow >= g.rmin && grow < g.rmax && kk < K)
            v = *(const float4*)(g.A + grow * g.lda + kk);
        return v;
    };

    // 4-tap (phase-split stride-2 conv): 7 of the 16 (tap, phase) weight blocks are structurally zero -- input pixel
    // (2i+dh, 2j+dw) exists only for dh, dw <= 2, i.e. not (tap row 1 & odd row phase) nor (tap col 1 & odd col
    // phase).  The phase is the 64-channel block of the contraction chunk (forward) or of this tile's output
    // columns (data-gradient, NT <= N/4); those steps are skipped altogether.
    auto step_valid = [&](int step) -> bool {
        if (NTAPS != 4) return true;
        const int kc = step / NTAPS, tap = step - kc * NTAPS;
        const int phase = (WMODE == 0) ? (kc * KC) / (K >> 2) : n0 / (N >> 2);
        return !(((tap >> 1) & (phase >> 1)) | ((tap & 1) & (phase & 1)));
    };
    auto next_step = [&](int step) -> int {
        do { ++step; } while (step < nsteps && !step_valid(step));
        return step;
    };

    load_w(0);
    if (NTAPS == 1) {
#pragma unroll
        for (int i = 0; i < AREG; ++i) areg[i] = load_a_row(tid + i * 256, 0);
    }

    for (int step = 0; step < nsteps; step = next_step(step)) {
        const int kc = step / NTAPS, tap = step - kc * NTAPS;
        __syncthreads();
        if (tap == 0) {
            if (NTAPS == 1) {
#pragma unroll
                for (int i = 0; i < AREG; ++i) {
                    const int f = tid + i * 256;
                    *(float4*)(As + (f / (KC / 4)) * AP + (f % (KC / 4)) * 4) = areg[i];
                }
            } else {
                // batches of SB independent loads, then SB LDS writes: a plain "load; store" loop makes hipcc
                // wait for every load before issuing the next one (one exposed memory latency per float4)
                constexpr int SB = (TM * TN >= 4) ? 1 : 4;   // the 128x128 tile has no registers to spare (3 waves per SIMD)
                for (int base = 0; base < arows * (KC / 4); base += SB * 256) {
                    float4 t[SB];
#pragma unroll
                    for (int i = 0; i < SB; ++i) {
                        const int f = base + tid + i * 256;
                        t[i] = (f < arows * (KC / 4)) ? load_a_row(f, kc) : make_float4(0.f, 0.f, 0.f, 0.f);
                    }
#pragma unroll
                    for (int i = 0; i < SB; ++i) {
                        const int f = base + tid + i * 256;
                        if (f < arows * (KC / 4)) *(float4*)(As + (f / (KC / 4)) * AP + (f % (KC / 4)) * 4) = t[i];
                    }
                }
            }
        }
        store_w();
        __syncthreads();
        if (next_step(step) < nsteps) {
            load_w(next_step(step));
            if (NTAPS == 1) {
#pragma unroll
                for (int i = 0; i < AREG; ++i) areg[i] = load_a_row(tid + i * 256, kc + 1);
            }
        }
        const int toff = halo + tap_row_offset<NTAPS, WMODE>(tap, g.WP);
        const float* abase = As + (wm * (TM * 32) + li + toff) * AP + 4 * lh;
        const float* wbase = (WMODE == 0) ? (Ws + (4 * lh) * NT + wn * (TN * 32) + li)
                                          : (Ws + (wn * (TN * 32) + li) * (KC + 1) + 4 * lh);
#pragma unroll
        for (int gk = 0; gk < KC / 8; ++gk) {
            float4 av[TM];
            float bv[TN][4];
#pragma unroll
            for (int a = 0; a < TM; ++a) av[a] = *(const float4*)(abase + a * 32 * AP + gk * 8);
#pragma unroll
            for (int b = 0; b < TN; ++b)
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    bv[b][s] = (WMODE == 0) ? wbase[(gk * 8 + s) * NT + b * 32]
                                            : wbase[b * 32 * (KC + 1) + gk * 8 + s];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
#pragma unroll
                for (int a = 0; a < TM; ++a) {
                    const float as = (s == 0) ? av[a].x : (s == 1) ? av[a].y : (s == 2) ? av[a].z : av[a].w;
#pragma unroll
                    for (int b = 0; b < TN; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(as, bv[b][s], acc[a][b], 0, 0, 0);
                }
            }
        }
    }

    // epilogue (tap_epilogue): transpose through LDS, float4 stores.  The staging tiles are dead by now.
    __syncthreads();
    tap_epilogue<TM, TN>(g, acc, tile_lds + wave * (32 * 33), rowa, rowy, wm * (TM * 32), n0 + wn * (TN * 32), lane, tile_m * WM + wm);
}

// ---- v5 ("pw"): the weight operand never touches LDS -- it is PRE-ARRANGED in MFMA fragment order (asr_arrange_weights, once per optimiser step):
// Wf [tap][K/8 groups][ceil(N/32) column blocks][64 lanes][4] fp32, lane 32h+i = column 32*block+i, contraction indices
// 8*group + 4h .. + 3 -- the B operands of four consecutive v_mfma_f32_32x32x2_f32 are ONE coalesced float4 per lane
// (1 KB per wave), identical for the forward and the data-gradient view.  All ring loads are branch-free (zero padding,
// clamped tails), so the compiler's s_waitcnt counts stay exact and the ring really runs D units ahead; the unit loop
// is fully unrolled.  Two barriers per chunk, no weight tile in LDS.
// toff: pixel-row offsets of the taps of a launch with a tap LIST of its own (NTAPS 2 or 4: the per-phase launches of the stride-2
// data-gradient, asr_conv_s2_dgrad); 9-tap and 1-tap launches use tap_row_offset
struct PwArgs { TapGemmArgs g; const float* Wf; int kg, nbt; int toff[4]; };

// DIR (0 forward, 1 data-gradient view) changes no code: it gives the two directions distinct symbols in a profile.
template <int MT, int NT, int WM, int WN, int NTAPS, int KCV, int D, int MINB, int DIR>
__global__ __launch_bounds__(256, MINB) void tap_gemm_kernel_v5(PwArgs args) {
    const TapGemmArgs& g = args.g;
    constexpr int KC = KCV, AP = KCV + 4;
    constexpr int TM = MT / WM / 32, TN = NT / WN / 32;
    constexpr int GK = KC / 8;               // 8-wide contraction groups per chunk
    constexpr int U = NTAPS * GK;            // pipeline units per chunk
    constexpr int SB = 4;
    static_assert(U % D == 0 && TM >= 1 && TN >= 1, "ring depth must divide the units of a chunk");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int halo = g.halo;
    const int arows = MT + 2 * halo;
    int* rowa = (int*)smem;
    int* rowy = rowa + MT;
    float* tile_lds = smem + 2 * MT;
    float* As = tile_lds;

    const int tid = threadIdx.x;
    const int lane = tid & 63, li = lane & 31, lh = lane >> 5;
    const int wave = tid >> 6, wm = wave / WN, wn = wave % WN;
    const int swz = asr_xcd_swizzle(blockIdx.x, gridDim.x);
    const int tile_m = swz / g.ntn, tile_n = swz - tile_m * g.ntn;
    const long p0 = (long)tile_m * MT;
    const int n0 = tile_n * NT;
    const int K = g.K;

    if (tid < MT) {
        const long p = p0 + tid;
        int ra = -1, ry = -1;
        if (p < g.M) {
            if (g.H == 0) {
                ra = (int)p; ry = (int)p;
            } else {
                const int b = (int)(p / g.HPWP);
                const int r = (int)(p - (long)b * g.HPWP);
                const int hh = r / g.WP, ww = r - hh * g.WP;
                if (hh >= 1 && hh <= g.H && ww >= 1 && ww <= g.Wd) {
                    ra = (int)p;
                    ry = g.y_unpadded ? ((b * g.H + hh - 1) * g.Wd + ww - 1) : (int)p;
                }
            }
        }
        rowa[tid] = ra; rowy[tid] = ry;
    }

    floatx16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int nkc = (K + KC - 1) / KC;
    float4 breg[D][TN];
    const int kg = args.kg, nbt = args.nbt;
    int nbb[TN];
#pragma unroll
    for (int b = 0; b < TN; ++b) { nbb[b] = (n0 >> 5) + wn * TN + b; if (nbb[b] >= nbt) nbb[b] = nbt - 1; }

    auto load_b = [&](float4 (&dst)[TN], int kc, int u) {
        const int tap = u / GK, gk = u - tap * GK;
        int grp = kc * GK + gk;
        if (grp >= kg) grp = kg - 1;                       // K tail of a 16/32-deep chunk: the A side is zero there
#pragma unroll
        for (int b = 0; b < TN; ++b)
            dst[b] = *(const float4*)(args.Wf + ((((long)tap * kg + grp) * nbt + nbb[b]) * 64 + lane) * 4);
    };
    auto load_a_row = [&](int f, int kc) -> float4 {
        const int row = f / (KC / 4), c4 = f - row * (KC / 4);
        const long grow = p0 - halo + row;
        const int kk = kc * KC + c4 * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (grow >= g.rmin && grow < g.rmax && kk < K) v = *(const float4*)(g.A + grow * g.lda + kk);
        return v;
    };

#pragma unroll
    for (int d = 0; d < D; ++d) load_b(breg[d], 0, d);

    for (int kc = 0; kc < nkc; ++kc) {
        __syncthreads();
        for (int base = 0; base < arows * (KC / 4); base += SB * 256) {
            float4 t[SB];
#pragma unroll
            for (int i = 0; i < SB; ++i) {
                const int f = base + tid + i * 256;
                t[i] = (f < arows * (KC / 4)) ? load_a_row(f, kc) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int i = 0; i < SB; ++i) {
                const int f = base + tid + i * 256;
                if (f < arows * (KC / 4)) *(float4*)(As + (f / (KC / 4)) * AP + (f % (KC / 4)) * 4) = t[i];
            }
        }
        __syncthreads();
        const int knext = (kc + 1 < nkc) ? kc + 1 : kc;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int tap = u / GK, gk = u - tap * GK;
            const int toff = halo + ((NTAPS == 2 || NTAPS == 4) ? args.toff[tap] : tap_row_offset<NTAPS, 0>(tap, g.WP));
            const float* abase = As + (wm * (TM * 32) + li + toff) * AP + 4 * lh + gk * 8;
            float4 av[TM];
#pragma unroll
            for (int a = 0; a < TM; ++a) av[a] = *(const float4*)(abase + a * 32 * AP);
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) {
#pragma unroll
                for (int a = 0; a < TM; ++a) {
                    const float as = (s2 == 0) ? av[a].x : (s2 == 1) ? av[a].y : (s2 == 2) ? av[a].z : av[a].w;
#pragma unroll
                    for (int b = 0; b < TN; ++b) {
                        const float4 bv = breg[u % D][b];
                        const float bs = (s2 == 0) ? bv.x : (s2 == 1) ? bv.y : (s2 == 2) ? bv.z : bv.w;
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(as, bs, acc[a][b], 0, 0, 0);
                    }
                }
            }
            // refill this ring slot with the unit D ahead (it may belong to the next chunk; the last chunk reloads itself)
            __builtin_amdgcn_sched_barrier(0);
            load_b(breg[u % D], (u + D) >= U ? knext : kc, (u + D) % U);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    __syncthreads();
    tap_epilogue<TM, TN>(g, acc, tile_lds + wave * (32 * 33), rowa, rowy, wm * (TM * 32), n0 + wn * (TN * 32), lane, tile_m * WM + wm);
}

template <int MT, int NT, int WM, int WN, int NTAPS, int KCV, int D, int MINB, int DIR>
int launch_v5d(const TapGemmArgs& a, const float* Wf, hipStream_t st, const int* toff = nullptr) {
    auto kern = tap_gemm_kernel_v5<MT, NT, WM, WN, NTAPS, KCV, D, MINB, DIR>;
    const int arows = MT + 2 * a.halo;
    size_t lds = (size_t)arows * (KCV + 4) * sizeof(float) + 2 * MT * sizeof(int);
    if (lds < kEpilogueLds + 2 * MT * sizeof(int)) lds = kEpilogueLds + 2 * MT * sizeof(int);
    if (lds > 160 * 1024) return ASR_ERR_UNSUPPORTED;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    PwArgs p;
    p.g = a; p.Wf = Wf; p.kg = (a.K + 7) / 8; p.nbt = (a.N + 31) / 32;
    for (int i = 0; i < 4; ++i) p.toff[i] = (toff && i < NTAPS) ? toff[i] : 0;
    p.g.ntm = asr_cdiv(a.M, MT);
    p.g.ntn = asr_cdiv(a.N, NT);
    if (!asr_gate_rows_fit(a.gate_rows, p.g.ntm * WM)) return ASR_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(kern, dim3(p.g.ntm * p.g.ntn), dim3(256), lds, st, p);
    ASR_CHECK_LAUNCH("tap_gemm_pw");
    ASR_NOTE_KERNEL("tap_gemm_kernel_v5<%d, %d, %d, %d, %d, %d, %d, %d, %d>", MT, NT, WM, WN, NTAPS, KCV, D, MINB, DIR);
    return ASR_OK;
}

template <int MT, int NT, int WM, int WN, int NTAPS, int KCV, int D, int MINB>
int launch_v5(const TapGemmArgs& a, const float* Wf, int dir, hipStream_t st) {
    return dir ? launch_v5d<MT, NT, WM, WN, NTAPS, KCV, D, MINB, 1>(a, Wf, st) : launch_v5d<MT, NT, WM, WN, NTAPS, KCV, D, MINB, 0>(a, Wf, st);
}

// W [ntaps][K][N] (HWIO; wmode 0) or its data-gradient view (wmode 1: taps mirrored, K and N = the GEMM's, i.e. swapped)
// -> Wf fp32 in fragment order (see tap_gemm_kernel_v5)
__global__ void arrange_weights_kernel(const float* __restrict__ W, int ntaps, int K, int N, int ldw, int wmode,
                                       float* __restrict__ out) {
    // one thread = one lane's float4 (4 consecutive contraction indices of one column): the forward view reads four
    // rows (each coalesced over the 32 columns of a block), the data-gradient view one float4; the store is coalesced
    const int KG = (K + 7) >> 3, NB = (N + 31) >> 5;
    const long total = (long)ntaps * KG * NB * 64;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int lane = (int)(i & 63);
        long r = i >> 6;
        const int nb = (int)(r % NB); r /= NB;
        const int g8 = (int)(r % KG);
        const int tap = (int)(r / KG);
        const int n = nb * 32 + (lane & 31), k = g8 * 8 + 4 * (lane >> 5);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (n < N) {
            if (wmode == 0) {
                const float* src = W + ((long)tap * K + k) * ldw + n;
                if (k < K) v.x = src[0];
                if (k + 1 < K) v.y = src[ldw];
                if (k + 2 < K) v.z = src[2 * (long)ldw];
                if (k + 3 < K) v.w = src[3 * (long)ldw];
            } else {
                const float* src = W + ((long)(ntaps - 1 - tap) * N + n) * ldw + k;
                if (k + 3 < K) v = *(const float4*)src;      // K (= ldw-contiguous) is a multiple of 4 (checked by the caller)
            }
        }
        *(float4*)(out + i * 4) = v;
    }
}

template <int MT, int NT, int WM, int WN, int NTAPS, int WMODE, int KCV>
int launch_v1(const TapGemmArgs& a, hipStream_t st) {
    auto kern = tap_gemm_kernel_v1<MT, NT, WM, WN, NTAPS, WMODE, KCV>;
    const int arows = MT + 2 * a.halo;
    const size_t wfl = ((WMODE == 0) ? KCV * NT : NT * (KCV + 1) + 3) / 4 * 4;
    size_t lds = ((size_t)arows * (KCV + 4) + wfl) * sizeof(float) + 2 * MT * sizeof(int);
    if (lds < kEpilogueLds + 2 * MT * sizeof(int)) lds = kEpilogueLds + 2 * MT * sizeof(int);
    if (lds > 160 * 1024) return ASR_ERR_UNSUPPORTED;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    TapGemmArgs g = a;
    g.ntm = asr_cdiv(a.M, MT);
    g.ntn = asr_cdiv(a.N, NT);
    if (!asr_gate_rows_fit(a.gate_rows, g.ntm * WM)) return ASR_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(kern, dim3(g.ntm * g.ntn, (NTAPS == 1 && WMODE == 0 && g.ksplit > 1) ? g.ksplit : 1), dim3(256), lds, st, g);
    ASR_CHECK_LAUNCH("tap_gemm");
    ASR_NOTE_KERNEL("tap_gemm_kernel_v1<%d, %d, %d, %d, %d, %d, %d>", MT, NT, WM, WN, NTAPS, WMODE, KCV);
    return ASR_OK;
}

template <int NTAPS, int WMODE>
int launch_n(const TapGemmArgs& a, hipStream_t st) {
    if constexpr (NTAPS == 4) {         // pre-net stride-2 conv on the phase-split plane: 4C -> C and its data-gradient
        // 64-wide column tiles: a data-gradient tile then lies inside one phase block (see step_valid)
        if (WMODE == 0 ? (a.K & 255) : (a.N & 255)) return ASR_ERR_UNSUPPORTED;        // C must be a multiple of 64
        if (a.N > 32) return launch_v1<128, 64, 2, 2, NTAPS, WMODE, 32>(a, st);
        return launch_v1<256, 32, 4, 1, NTAPS, WMODE, 32>(a, st);
    } else {
    if (NTAPS == 1 && a.N > 32) {
        // a grid of 128x128 tiles that leaves most CUs idle (e.g. the 6400->128 hidden dense of
        // acoustic_model.py: 50 tiles) runs on 64x64 tiles instead
        const long tiles = (long)asr_cdiv(a.M, 128) * asr_cdiv(a.N, 128);
        if (tiles < 160) return launch_v1<64, 64, 2, 2, NTAPS, WMODE, 32>(a, st);
        if constexpr (WMODE == 0) {
            // forward GEMMs whose 128x128 grid is about one round of the chip (tools/bench_layers.py): up to 768 tiles
            // (3 workgroups/CU at KC 32) leave CUs idle -> 128x64 tiles (+7 % on 6400x6400x1536); 769..1024 tiles fit
            // one round only at 4 workgroups/CU, which the 16-deep chunk allows (+14 % on 32768x512x512)
            if (a.N > 64 && tiles <= 768) return launch_v1<128, 64, 2, 2, NTAPS, WMODE, 32>(a, st);
            if (a.N > 64 && tiles <= 1024) return launch_v1<128, 128, 2, 2, NTAPS, WMODE, 16>(a, st);
        } else {
            // data-gradient GEMMs of under one round of 128x128 tiles (the 6 400-row language models: 200 tiles): 128x64
            // tiles (+1.3 % on the Language_Model step)
            if (a.N > 64 && tiles <= 512) return launch_v1<128, 64, 2, 2, NTAPS, WMODE, 32>(a, st);
        }
    }
    if (NTAPS == 9) {
        // measured on the DFCNN layer shapes (tools/bench_layers.py, profiles/r01b_layer_tiles.txt):
        // 128x64 tiles (more, smaller workgroups per CU) win up to 128 output channels; a 16-deep
        // K chunk wins while the A tile (plane width + halo) dominates LDS, 32-deep for K >= 128
        if (a.N > 32 && a.N <= 64) return launch_v1<128, 64, 2, 2, NTAPS, WMODE, 16>(a, st);
        if (a.N > 64 && a.N <= 128) {
            if (a.K >= 128) return launch_v1<128, 64, 2, 2, NTAPS, WMODE, 32>(a, st);
            return launch_v1<128, 64, 2, 2, NTAPS, WMODE, 16>(a, st);
        }
    }
    if (a.N > 64) return launch_v1<128, 128, 2, 2, NTAPS, WMODE, 32>(a, st);
    if (a.N > 32) return launch_v1<256, 64, 4, 1, NTAPS, WMODE, 32>(a, st);
    return launch_v1<256, 32, 4, 1, NTAPS, WMODE, 32>(a, st);
    }
}

}  // namespace

struct GateSpec { int mode, H, W; const float* a; float* dz; float* part; int* rows; };

// gemm1.hip: the LDS-DMA "NT" GEMM that takes the large 1-tap data-gradients (dX = dY . W^T: both operands K-contiguous)
struct Gemm1Gate { int mode, H, W; const float* a; float* dz; float* part; int* rows; int C = 0; };
bool asr_gemm1_eligible(const asr_gemm_desc* d, const float* A, const float* Bt, int ldb);
int asr_gemm1_launch(const asr_gemm_desc* d, const float* A, const float* Bt, int ldb, const float* bias, const float* scale,
                     const float* shift, float* out_a, float* out_y, int dir, void* stream, const Gemm1Gate* gate);

static void set_gate(TapGemmArgs& a, const GateSpec* gs) {
    a.gate_mode = 0; a.gate_H = a.gate_W = 0; a.gate_a = nullptr; a.gate_dz = nullptr; a.gate_part = nullptr; a.gate_rows = nullptr;
    if (gs) { a.gate_mode = gs->mode; a.gate_H = gs->H; a.gate_W = gs->W; a.gate_a = gs->a; a.gate_dz = gs->dz; a.gate_part = gs->part; a.gate_rows = gs->rows; }
}

static int tap_gemm_impl(const asr_gemm_desc* d, const float* A, const float* W,
                         const float* bias, const float* scale, const float* shift,
                         float* out_a, float* out_y, void* stream, const GateSpec* gs) {
    if (!d || !A || !W || (!out_a && !out_y && !gs)) return ASR_ERR_BAD_ARG;
    if (d->ntaps != 1 && d->ntaps != 9 && d->ntaps != 4) return ASR_ERR_BAD_ARG;
    if ((d->K & 3) || (d->N & 3) || (d->lda & 3) || (d->ldw & 3)) return ASR_ERR_BAD_ARG;
    if (d->ntaps != 1 && d->H <= 0) return ASR_ERR_BAD_ARG;
    if (d->M <= 0 || d->K <= 0 || d->N <= 0) return ASR_ERR_BAD_ARG;
    if (((uintptr_t)A | (uintptr_t)W) & 15) return ASR_ERR_BAD_ARG;
    if (d->ntaps == 1 && d->wmode == 1 && asr_gemm1_eligible(d, A, W, d->ldw)) {
        Gemm1Gate gg;
        if (gs) { gg.mode = gs->mode; gg.H = gs->H; gg.W = gs->W; gg.a = gs->a; gg.dz = gs->dz; gg.part = gs->part; gg.rows = gs->rows; }
        return asr_gemm1_launch(d, A, W, d->ldw, bias, scale, shift, out_a, out_y, 1, stream, gs ? &gg : nullptr);
    }
    TapGemmArgs a;
    a.A = A; a.W = W; a.bias = bias; a.scale = scale; a.shift = shift;
    a.out_a = out_a; a.out_y = out_y;
    a.M = d->M; a.K = d->K; a.N = d->N; a.lda = d->lda; a.ldw = d->ldw;
    a.ldo_a = d->ldo_a; a.ldo_y = d->ldo_y;
    a.H = d->H; a.Wd = d->W; a.WP = d->W + 1; a.HPWP = (d->H + 1) * (d->W + 1);
    if (d->H > 0 && d->M != d->B * a.HPWP) return ASR_ERR_BAD_ARG;
    a.halo = (d->ntaps != 1) ? a.WP + 1 : 0;
    a.rmin = -(long)a.halo; a.rmax = (long)d->M + a.halo;
    a.relu = d->relu; a.accumulate = d->accumulate; a.y_unpadded = d->y_unpadded;
    a.ntm = a.ntn = 0;
    set_gate(a, gs);
    hipStream_t st = (hipStream_t)stream;
    if (d->ntaps == 9) return d->wmode ? launch_n<9, 1>(a, st) : launch_n<9, 0>(a, st);
    if (d->ntaps == 4) return d->wmode ? launch_n<4, 1>(a, st) : launch_n<4, 0>(a, st);
    return d->wmode ? launch_n<1, 1>(a, st) : launch_n<1, 0>(a, st);
}

namespace {
// second pass of the split-K form: sum of the partial planes in a fixed order, then bias / ReLU|tanh / BN affine as tap_epilogue
__global__ void splitk_reduce_kernel(const float* __restrict__ part, int splits, long MN, int N, const float* __restrict__ bias,
                                     const float* __restrict__ scale, const float* __restrict__ shift, int relu, int accumulate,
                                     float* __restrict__ out_a, int ldo_a, float* __restrict__ out_y, int ldo_y) {
    const long n4 = MN / 4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        float4 v = *(const float4*)(part + i * 4);
        for (int z = 1; z < splits; ++z) {
            const float4 q = *(const float4*)(part + (long)z * MN + i * 4);
            v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
        }
        const long e = i * 4;
        const long row = e / N;
        const int n = (int)(e - row * N);
        if (bias) { const float4 b = *(const float4*)(bias + n); v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w; }
        if (relu == 1) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        else if (relu == 2) { v.x = tanhf(v.x); v.y = tanhf(v.y); v.z = tanhf(v.z); v.w = tanhf(v.w); }
        if (out_a) *(float4*)(out_a + row * ldo_a + n) = v;
        if (out_y) {
            float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
            if (scale) sc = *(const float4*)(scale + n);
            if (shift) sh = *(const float4*)(shift + n);
            float4 y = make_float4(sc.x * v.x + sh.x, sc.y * v.y + sh.y, sc.z * v.z + sh.z, sc.w * v.w + sh.w);
            float* o = out_y + row * ldo_y + n;
            if (accumulate) { const float4 p = *(const float4*)o; y.x += p.x; y.y += p.y; y.z += p.z; y.w += p.w; }
            *(float4*)o = y;
        }
    }
}
}  // namespace

// (also the second pass of asr_tap_gemm_nt_splitk, gemm1.hip)
ASR_INTERNAL int asr_splitk_reduce_launch(const float* slab, int splits, const asr_gemm_desc* d, const float* bias, const float* scale,
                                          const float* shift, float* out_a, float* out_y, void* stream) {
    const long MN = (long)d->M * d->N;
    int blocks = (int)((MN / 4 + 255) / 256); if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, slab, splits, MN, d->N, bias, scale, shift,
                       d->relu, d->accumulate, out_a, d->ldo_a, out_y, d->ldo_y);
    ASR_CHECK_LAUNCH("splitk_reduce");
    return ASR_OK;
}

extern "C" size_t asr_tap_gemm_splitk_workspace(const asr_gemm_desc* d, int splits) {
    return (d && splits > 1) ? (size_t)splits * d->M * d->N * sizeof(float) : 0;
}

// Dense forward GEMM (ntaps 1, wmode 0, no plane geometry) with the contraction split `splits` ways over the grid: for a deep K
// and few output tiles (6400 -> 128 hidden dense of acoustic_model.py:53: 200 tiles on 256 CUs, 200 chunk steps each).
// Same result up to the order of the K sum (partials are added in a fixed order: reproducible).
extern "C" int asr_tap_gemm_splitk(const asr_gemm_desc* d, const float* A, const float* W, const float* bias, const float* scale,
                                   const float* shift, float* out_a, float* out_y, int splits, void* workspace, void* stream) {
    if (!d || !A || !W || (!out_a && !out_y) || !workspace) return ASR_ERR_BAD_ARG;
    if (d->ntaps != 1 || d->wmode != 0 || d->H > 0 || d->y_unpadded || splits < 2 || splits > 16) return ASR_ERR_BAD_ARG;
    if ((d->K % (splits * 32)) || (d->N & 3) || (d->lda & 3) || (d->ldw & 3) || (((uintptr_t)A | (uintptr_t)W | (uintptr_t)workspace) & 15)) return ASR_ERR_BAD_ARG;
    TapGemmArgs a;
    a.A = A; a.W = W; a.bias = nullptr; a.scale = nullptr; a.shift = nullptr;
    a.out_a = (float*)workspace; a.out_y = nullptr;
    a.M = d->M; a.K = d->K; a.N = d->N; a.lda = d->lda; a.ldw = d->ldw; a.ldo_a = d->N; a.ldo_y = 0;
    a.H = 0; a.Wd = 0; a.WP = 1; a.HPWP = 1; a.halo = 0;
    a.rmin = 0; a.rmax = d->M;
    a.relu = 0; a.accumulate = 0; a.y_unpadded = 0;
    a.ntm = a.ntn = 0;
    set_gate(a, nullptr);
    a.ksplit = splits; a.split_out = (float*)workspace;
    hipStream_t st = (hipStream_t)stream;
    const int rc = launch_v1<64, 64, 2, 2, 1, 0, 32>(a, st);
    if (rc != ASR_OK) return rc;
    const long MN = (long)d->M * d->N;
    int blocks = (int)((MN / 4 + 255) / 256); if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, (const float*)workspace, splits, MN, d->N, bias, scale, shift,
                       d->relu, d->accumulate, out_a, d->ldo_a, out_y, d->ldo_y);
    ASR_CHECK_LAUNCH("tap_gemm_splitk");
    return ASR_OK;
}

extern "C" int asr_tap_gemm(const asr_gemm_desc* d, const float* A, const float* W,
                            const float* bias, const float* scale, const float* shift,
                            float* out_a, float* out_y, void* stream) {
    return tap_gemm_impl(d, A, W, bias, scale, shift, out_a, out_y, stream, nullptr);
}


// ---- fp32 contraction on pre-arranged weights (include/asr_hip.h)
extern "C" size_t asr_arrange_weights_bytes(int ntaps, int K, int N) {
    return (size_t)ntaps * ((K + 7) / 8) * ((N + 31) / 32) * 256 * sizeof(float);
}

extern "C" int asr_arrange_weights(const float* W, int ntaps, int K, int N, int ldw, int wmode, float* out, void* stream) {
    if (!W || !out || (ntaps != 1 && ntaps != 9) || K < 1 || N < 1) return ASR_ERR_BAD_ARG;
    if ((K & 3) || (ldw & 3) || (((uintptr_t)W | (uintptr_t)out) & 15)) return ASR_ERR_BAD_ARG;
    const long total = (long)ntaps * ((K + 7) / 8) * ((N + 31) / 32) * 64;
    long nb = (total + 255) / 256;
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(arrange_weights_kernel, dim3((int)nb), dim3(256), 0, (hipStream_t)stream, W, ntaps, K, N, ldw, wmode, out);
    ASR_CHECK_LAUNCH("arrange_weights");
    return ASR_OK;
}

static int tap_gemm_pw_impl(const asr_gemm_desc* d, const float* A, const float* Wf,
                            const float* bias, const float* scale, const float* shift,
                            float* out_a, float* out_y, void* stream, const GateSpec* gs) {
    if (!d || !A || !Wf || (!out_a && !out_y && !gs)) return ASR_ERR_BAD_ARG;
    if (d->ntaps != 1 && d->ntaps != 9) return ASR_ERR_BAD_ARG;
    if ((d->K & 3) || (d->N & 3) || (d->lda & 3)) return ASR_ERR_BAD_ARG;
    if (d->ntaps == 9 && d->H <= 0) return ASR_ERR_BAD_ARG;
    if (d->M <= 0 || d->K <= 0 || d->N <= 0) return ASR_ERR_BAD_ARG;
    if (((uintptr_t)A | (uintptr_t)Wf) & 15) return ASR_ERR_BAD_ARG;
    TapGemmArgs a;
    a.A = A; a.W = nullptr; a.bias = bias; a.scale = scale; a.shift = shift;
    a.out_a = out_a; a.out_y = out_y;
    a.M = d->M; a.K = d->K; a.N = d->N; a.lda = d->lda; a.ldw = d->ldw;
    a.ldo_a = d->ldo_a; a.ldo_y = d->ldo_y;
    a.H = d->H; a.Wd = d->W; a.WP = d->W + 1; a.HPWP = (d->H + 1) * (d->W + 1);
    if (d->H > 0 && d->M != d->B * a.HPWP) return ASR_ERR_BAD_ARG;
    a.halo = (d->ntaps != 1) ? a.WP + 1 : 0;
    a.rmin = -(long)a.halo; a.rmax = (long)d->M + a.halo;
    a.relu = d->relu; a.accumulate = d->accumulate; a.y_unpadded = d->y_unpadded;
    a.ntm = a.ntn = 0;
    set_gate(a, gs);
    a.nt_store = 0;       // streaming stores were measured on the direct kernels: no change
    hipStream_t st = (hipStream_t)stream;
    const int dir = d->wmode ? 1 : 0;       // labels the launch only (distinct kernel symbols per direction)
    // tile choice (tools/bench_pw.py, MI355X): 128x64 workgroup tiles with a 16-deep chunk and a 3-unit ring win on every
    // 3x3 layer (deeper rings, 32-deep chunks, 256-row or 128-column tiles: -1..-8 %); for the 1-tap GEMMs the LDS-staged
    // asr_tap_gemm stays faster (0.86-0.97x here: no tap reuse to pay for the direct weight loads), so the engines use
    // this entry point for ntaps = 9 only -- the 1-tap configurations exist for completeness and tests
    if (d->ntaps == 9) {
        if (d->N > 32) {
            // The kernel advances in beats of one tile per CU (tools/exp_tail.py: its duration is ceil(tiles / 256) x the time
            // of one tile, whatever the three co-resident workgroups overlap), so the partial last beat is lost: 2 614 tiles
            // of 128 rows = 10.2 beats cost 11.  Rows per tile are therefore picked per launch, 128 or 192 (2 x 2 waves of
            // 2 or 3 row blocks), whichever needs fewer row-beats; at equal cost 192 wins by its smaller halo share
            // (gpurun_out/r02e/mt.log: 128->128 and 128->256 at 200x25 +2..5 %, the 400x50 layers stay on 128).
            const long nt = asr_cdiv(d->N, 64);
            const long c128 = (long)asr_cdiv((long)asr_cdiv(d->M, 128) * nt, 256) * 128;
            const long c192 = (long)asr_cdiv((long)asr_cdiv(d->M, 192) * nt, 256) * 192;
            if (c192 <= c128) return launch_v5<192, 64, 2, 2, 9, 16, 3, 3>(a, Wf, dir, st);
            return launch_v5<128, 64, 2, 2, 9, 16, 3, 3>(a, Wf, dir, st);
        }
        return launch_v5<256, 32, 4, 1, 9, 16, 3, 3>(a, Wf, dir, st);
    }
    if (d->N > 64) return launch_v5<128, 128, 2, 2, 1, 32, 4, 2>(a, Wf, dir, st);
    if (d->N > 32) return launch_v5<128, 64, 2, 2, 1, 32, 4, 3>(a, Wf, dir, st);
    return launch_v5<256, 32, 4, 1, 1, 32, 4, 3>(a, Wf, dir, st);
}

extern "C" int asr_tap_gemm_pw(const asr_gemm_desc* d, const float* A, const float* Wf,
                               const float* bias, const float* scale, const float* shift,
                               float* out_a, float* out_y, void* stream) {
    return tap_gemm_pw_impl(d, A, Wf, bias, scale, shift, out_a, out_y, stream, nullptr);
}

// ---- the stride-2 convolution of the end2end pre-net (end2end/model.py:225-229), data-gradient, one launch per OUTPUT PHASE (round 6).
// On the phase-split plane the layer is a forward-looking 2 x 2-tap convolution 4C -> C whose (tap, phase) weight blocks are zero when the
// tap looks one pixel further in a direction whose phase is odd (tap_gemm_kernel_v1's step_valid): phase (rp, cp) of the input gradient
// receives (rp ? 1 : 2) x (cp ? 1 : 2) taps -- 4 / 2 / 2 / 1 for the phases 0 .. 3.  asr_tap_gemm ran it as ONE 4-tap GEMM with a 64-deep
// contraction per tap on the register-staged kernel (two barriers per tap, every staged tile re-staged for each of the four 64-column phase
// blocks: 31 TFLOP/s).  Here each phase is a C -> C convolution with its own tap list on tap_gemm_kernel_v5 (weights in fragment order
// straight from L2, two barriers per 32-deep chunk).
//   Wf9 = asr_conv_s2_arrange: the nine non-zero (phase, tap) blocks of W4 in data-gradient view, phase-major, taps in forward order.
namespace {
// taps of phase p as (dr, dc) of the FORWARD window, in the order asr_conv_s2_arrange stores their blocks
inline int s2_phase_taps(int p, int (&dr)[4], int (&dc)[4]) {
    const int rp = p >> 1, cp = p & 1;
    int n = 0;
    for (int t = 0; t < 4; ++t)
        if (!(((t >> 1) & rp) | ((t & 1) & cp))) { dr[n] = t >> 1; dc[n] = t & 1; ++n; }
    return n;
}
__global__ void s2_arrange_kernel(const float* __restrict__ W4, int C, float* __restrict__ out) {
    // one thread = one lane's float4 of one (phase, tap) block: B[k][n] = W4[tap][C * phase + n][k], fragment order of arrange_weights_kernel
    const int KG = C >> 3, NB = C >> 5;
    const long per = (long)KG * NB * 64;
    const long total = 9 * per;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int blk = (int)(i / per);
        long r = i - (long)blk * per;
        const int lane = (int)(r & 63); r >>= 6;
        const int nb = (int)(r % NB), g8 = (int)(r / NB);
        // block -> (phase, tap): phase 0 taps 0 1 2 3 | phase 1 taps 0 2 | phase 2 taps 0 1 | phase 3 tap 0
        const int phase = blk < 4 ? 0 : blk < 6 ? 1 : blk < 8 ? 2 : 3;
        const int tap = blk < 4 ? blk : blk < 6 ? 2 * (blk - 4) : blk < 8 ? blk - 6 : 0;
        const int n = nb * 32 + (lane & 31), k = g8 * 8 + 4 * (lane >> 5);
        *(float4*)(out + i * 4) = *(const float4*)(W4 + ((long)tap * 4 * C + (long)C * phase + n) * C + k);
    }
}
}  // namespace

extern "C" size_t asr_conv_s2_arrange_bytes(int C) { return (C < 64 || (C & 63)) ? 0 : (size_t)9 * (C / 8) * (C / 32) * 256 * sizeof(float); }

extern "C" int asr_conv_s2_arrange(const float* W4, int C, float* out, size_t out_bytes, void* stream) {
    if (!W4 || !out || C < 64 || (C & 63) || (((uintptr_t)W4 | (uintptr_t)out) & 15)) return ASR_ERR_BAD_ARG;
    if (out_bytes < asr_conv_s2_arrange_bytes(C)) return ASR_ERR_BAD_ARG;
    const long total = 9L * (C / 8) * (C / 32) * 64;
    hipLaunchKernelGGL(s2_arrange_kernel, dim3((int)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, W4, C, out);
    ASR_CHECK_LAUNCH("conv_s2_arrange");
    return ASR_OK;
}

extern "C" int asr_conv_s2_dgrad(const asr_gemm_desc* d, const float* dZ, const float* Wf9, float* dx, void* stream) {
    if (!d || !dZ || !Wf9 || !dx) return ASR_ERR_BAD_ARG;
    const int C = d->K;
    // the 4-tap data-gradient descriptor of asr_tap_gemm: K = C gradient channels, N = 4 C phase-split input channels
    if (d->ntaps != 4 || d->wmode != 1 || C < 64 || (C & 63) || d->N != 4 * C || d->H <= 0 || d->relu || d->accumulate || d->y_unpadded) return ASR_ERR_BAD_ARG;
    if ((d->lda & 3) || (d->ldo_y & 3) || d->ldo_y < 4 * C || d->M != d->B * (d->H + 1) * (d->W + 1)) return ASR_ERR_BAD_ARG;
    if (((uintptr_t)dZ | (uintptr_t)Wf9 | (uintptr_t)dx) & 15) return ASR_ERR_BAD_ARG;
    TapGemmArgs a;
    a.A = dZ; a.W = nullptr; a.bias = nullptr; a.scale = nullptr; a.shift = nullptr;
    a.out_a = nullptr;
    a.M = d->M; a.K = C; a.N = C; a.lda = d->lda; a.ldw = C;
    a.ldo_a = 0; a.ldo_y = d->ldo_y;
    a.H = d->H; a.Wd = d->W; a.WP = d->W + 1; a.HPWP = (d->H + 1) * (d->W + 1);
    a.rmin = -(long)(a.WP + 1); a.rmax = (long)d->M + a.WP + 1;
    a.relu = 0; a.accumulate = 0; a.y_unpadded = 0;
    a.ntm = a.ntn = 0;
    set_gate(a, nullptr);
    a.nt_store = 0;
    hipStream_t st = (hipStream_t)stream;
    const size_t blk = (size_t)(C / 8) * (C / 32) * 256;          // floats of one (phase, tap) block
    size_t off = 0;
    for (int p = 0; p < 4; ++p) {
        int dr[4], dc[4], toff[4] = {0, 0, 0, 0};
        const int nt = s2_phase_taps(p, dr, dc);
        for (int i = 0; i < nt; ++i) toff[i] = -(dr[i] * a.WP + dc[i]);      // the gradient pixel a forward tap came from
        a.out_y = dx + (size_t)p * C;
        a.halo = nt > 1 ? a.WP + 1 : 0;
        int rc;
        // tile: 192 rows x 64 columns, 32-deep chunks, ring of four units (tools/bench_s2_dgrad.py, B 64 x 512 x 80 x 64: 2.25 ms; 128 rows 2.37,
        // 16-deep chunks 2.31-2.45, 256 rows 2.44; the one 4-tap GEMM of asr_tap_gemm 3.37)
#ifndef S2_MT
#define S2_MT 192
#define S2_KC 32
#define S2_D 4
#define S2_MINB 3
#endif
        if (nt == 4) rc = launch_v5d<S2_MT, 64, 2, 2, 4, S2_KC, S2_D, S2_MINB, 1>(a, Wf9 + off, st, toff);
        else if (nt == 2) rc = launch_v5d<S2_MT, 64, 2, 2, 2, S2_KC, S2_D, S2_MINB, 1>(a, Wf9 + off, st, toff);
        else rc = launch_v5d<S2_MT, 64, 2, 2, 1, S2_KC, S2_D, S2_MINB, 1>(a, Wf9 + off, st);
        if (rc != ASR_OK) return rc;
        off += (size_t)nt * blk;
    }
    return ASR_OK;
}

extern "C" ASR_INTERNAL int asr_tap_gemm_wino_gated_launch(const asr_gemm_desc* d, const float* dZ, const float* Ut, int mode, int gate_H, int gate_W,
                                              const float* gate_a, const float* scale, const float* shift, float* dy_prev,
                                              float* dz_out, float* partials, int* rows, void* stream);      // wino.hip

// Data-gradient GEMM whose epilogue IS the backward prologue of the cell in front (tap_epilogue_gated).
extern "C" ASR_INTERNAL int asr_winograd_gate_rows(const asr_gemm_desc* d);                                              // wino.hip
extern "C" ASR_INTERNAL int asr_gated_partial_rows(const asr_gemm_desc* d) {      // rows the partials buffer of asr_tap_gemm_gated_workspace holds (wino.hip uses it too)
    int rows = asr_cdiv(d->M, 32) + 4;                        // every direct launch configuration has >= 32 tile rows per wave row (+ the ragged last tile)
    const int wr = asr_winograd_gate_rows(d);                 // the Winograd kernels: 4 per tile block (more on planes of few tile rows)
    return wr > rows ? wr : rows;
}
extern "C" size_t asr_tap_gemm_gated_workspace(const asr_gemm_desc* d) {
    if (!d) return 0;
    const int rows = asr_gated_partial_rows(d);
    return ((size_t)rows * 3 * d->N + asr_reduce::colsum_tmp_floats(rows, 3 * d->N)) * sizeof(float);
}

extern "C" int asr_tap_gemm_gated(const asr_gemm_desc* d, const float* dZ, const float* W, int prearranged,
                                  int pool, int gate_H, int gate_W, const float* gate_a,
                                  const float* bn_scale, const float* bn_shift, const float* dy_prev,
                                  float* dz_out, float* dscale, float* dshift, float* dbias, float* partials, void* stream) {
    if (!d || !dZ || !W || !gate_a || !bn_scale || !bn_shift || !dz_out || !dscale || !dshift || !dbias || !partials)
        return ASR_ERR_BAD_ARG;
    if (pool < 0 || pool > 2 || d->wmode != 1 || d->relu != 0 || d->y_unpadded) return ASR_ERR_BAD_ARG;
    if (d->accumulate && !dy_prev) return ASR_ERR_BAD_ARG;
    if (d->H <= 0) return ASR_ERR_UNSUPPORTED;                 // pixel-indexed outputs only
    if (pool == 0 ? (gate_H != d->H || gate_W != d->W) : (gate_H != 2 * d->H || gate_W != 2 * d->W)) return ASR_ERR_BAD_ARG;
    if (d->ldo_y != d->N) return ASR_ERR_BAD_ARG;
    int rows = asr_gated_partial_rows(d);                         // capacity of `partials` in rows; the launcher refuses to write more
    GateSpec gs;
    gs.mode = pool + 1; gs.H = gate_H; gs.W = gate_W; gs.a = gate_a; gs.dz = dz_out; gs.part = partials; gs.rows = &rows;
    const int rc = prearranged == 2 ? asr_tap_gemm_wino_gated_launch(d, dZ, W, gs.mode, gs.H, gs.W, gs.a, bn_scale, bn_shift, (float*)dy_prev,
                                                                     gs.dz, gs.part, gs.rows, stream)
                 : prearranged ? tap_gemm_pw_impl(d, dZ, W, nullptr, bn_scale, bn_shift, nullptr, (float*)dy_prev, stream, &gs)
                               : tap_gemm_impl(d, dZ, W, nullptr, bn_scale, bn_shift, nullptr, (float*)dy_prev, stream, &gs);
    if (rc != ASR_OK) return rc;
    if (rows <= 0) return ASR_ERR_UNSUPPORTED;
    asr_reduce::Multi m;
    m.nseg = 3; m.width[0] = d->N; m.width[1] = d->N; m.width[2] = d->N; m.width[3] = 0;
    m.out[0] = dscale; m.out[1] = dshift; m.out[2] = dbias; m.out[3] = nullptr;
    return asr_reduce::colsum_multi(partials, rows, 3L * d->N, m, partials + (size_t)rows * 3 * d->N, (hipStream_t)stream);
}

// ---- round 5: the same fusion where the cell in front hands its output to a DENSE layer (acoustic_model.py:48-50 reshape -> dense;
// acoustic_model2.py:62-66): dL/d(flat) = dZ . W^T is [B * H][W * C] in the dense layout -- row = (image, pixel row), column = pixel
// column * C + channel -- and used to be written out (164 MB at B = 32, 200 x 25 x 256), read back with the cell's activations by
// asr_cell_bwd_pre and turned into the cell's dZ plane and channel sums.  Here the GEMM's epilogue does that on the value in the
// register (gate mode 5 of tap_epilogue_gated: BN / ReLU backward of an un-pooled cell): no dL/d(flat) tensor, no separate pass.
//   d: the dense layer's data-gradient descriptor (ntaps 1, wmode 1, M = B * gate_H rows, K = its output width, N = gate_W * gate_C)
//   a_plane / dz_out: the cell's activation / dZ planes [B][gate_H + 1][gate_W + 1][gate_C]; dscale / dshift / dbias [gate_C]
// dZ is bit for bit what asr_tap_gemm + asr_cell_bwd_pre give; the three channel sums are folded in another fixed order.
extern "C" int asr_tap_gemm_gated_dense_supported(const asr_gemm_desc* d, int gate_H, int gate_W, int gate_C) {
    if (!d || d->ntaps != 1 || d->wmode != 1 || d->relu != 0 || d->accumulate || d->H > 0) return 0;
    if (gate_H < 1 || gate_W < 1 || gate_C < 32 || (gate_C & 31) || d->N != gate_W * gate_C || (d->M % gate_H) != 0) return 0;
    if ((long)(d->M / gate_H) * (gate_H + 1) * (gate_W + 1) >= (1L << 31) / (long)gate_C) return 0;          // plane pixels x channels in 31 bits
    return ((d->K & 3) || (d->lda & 3) || (d->ldw & 3) || d->K < 32 ||
            (long)asr_cdiv(d->M, 128) * asr_cdiv(d->N, 128) < 48) ? 0 : 1;                                      // asr_gemm1_eligible's shape rules
}
extern "C" size_t asr_tap_gemm_gated_dense_workspace(const asr_gemm_desc* d, int gate_W, int gate_C) {
    if (!d || gate_W < 1 || gate_C < 1) return 0;
    const int rows = asr_cdiv(d->M, 128) * 2 * gate_W;
    return ((size_t)rows * 3 * gate_C + asr_reduce::colsum_tmp_floats(rows, 3 * gate_C)) * sizeof(float);
}
extern "C" int asr_tap_gemm_gated_dense(const asr_gemm_desc* d, const float* dZ, const float* W, int gate_H, int gate_W, int gate_C,
                                        const float* a_plane, const float* bn_scale, const float* bn_shift, float* dz_out,
                                        float* dscale, float* dshift, float* dbias, float* partials, void* stream) {
    if (!d || !dZ || !W || !a_plane || !bn_scale || !bn_shift || !dz_out || !dscale || !dshift || !dbias || !partials) return ASR_ERR_BAD_ARG;
    if (!asr_tap_gemm_gated_dense_supported(d, gate_H, gate_W, gate_C) || !asr_gemm1_eligible(d, dZ, W, d->ldw)) return ASR_ERR_UNSUPPORTED;
    int rows = asr_cdiv(d->M, 128) * 2 * gate_W;              // capacity of `partials` (asr_tap_gemm_gated_dense_workspace); checked before the launch
    Gemm1Gate gg;
    gg.mode = 5; gg.H = gate_H; gg.W = gate_W; gg.C = gate_C; gg.a = a_plane; gg.dz = dz_out; gg.part = partials; gg.rows = &rows;
    const int rc = asr_gemm1_launch(d, dZ, W, d->ldw, nullptr, bn_scale, bn_shift, nullptr, nullptr, 1, stream, &gg);
    if (rc != ASR_OK) return rc;
    if (rows <= 0) return ASR_ERR_UNSUPPORTED;
    asr_reduce::Multi m;
    m.nseg = 3; m.width[0] = gate_C; m.width[1] = gate_C; m.width[2] = gate_C; m.width[3] = 0;
    m.out[0] = dscale; m.out[1] = dshift; m.out[2] = dbias; m.out[3] = nullptr;
    return asr_reduce::colsum_multi(partials, rows, 3L * gate_C, m, partials + (size_t)rows * 3 * gate_C, (hipStream_t)stream);
}

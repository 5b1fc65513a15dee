// EXPERIMENTAL split-bf16 dense contractions on PRE-SPLIT operands.
//
// tap_gemm_kernel_bx6 splits its A tile into hi / mid / lo bf16 pieces while staging it; a 3x3 convolution amortises that
// over nine taps, a plain GEMM (the Transformer projections and FFN of end2end/transformer.py:117-158, 204-231) does
// not and ends up slower than the fp32 MFMA kernel.  Here both operands arrive already split:
//   asr_split_rows:  X [M][K] fp32 (row pitch ldx)  ->  Xs bf16 [3][Kp/32][M][32]  (Kp = K rounded up to 32, zero filled)
//   asr_gemm_bx6s:   Y [M][N] = act(As . B + bias) with As [3][M][Kp] and B pre-split in fragment order (asr_split_weights), six
//                    v_mfma_f32_32x32x16_bf16 products per K-step, fp32 accumulation; optionally also writes Y split
//                    ([3][M][Np]) for the next GEMM.
// Staging is then a plain 16-byte copy (no VALU work); B fragments come straight from global memory / L2 into a
// register ring as in tap_gemm_kernel_bx6.
#include "asr_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void gsplit3(float x, __bf16& h, __bf16& m, __bf16& l) {
    h = (__bf16)x;
    const float r1 = x - (float)h;
    m = (__bf16)r1;
    l = (__bf16)(r1 - (float)m);
}

// Split planes are stored K-CHUNK-MAJOR: Xs[piece][k / 32][row][k % 32] (bf16), so that the 32-deep K chunk of a block of
// rows is one contiguous run (64 B per row): the GEMM's staging copy is fully coalesced and touches every cache line once
// (row-major planes put consecutive rows a power of two apart: half-used lines that thrash the L1 sets).
// one thread = 8 consecutive k of one row: two float4 loads, three 16-byte stores
__global__ __launch_bounds__(256) void split_rows_kernel(const float* __restrict__ X, long M, int K, int ldx, int Kp,
                                                         __bf16* __restrict__ out) {
    const int k8 = Kp >> 3;
    const long total = M * k8;
    const long ps = M * (long)Kp;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        // consecutive threads write consecutive 16-byte units of a plane: (chunk, row, unit) with the unit fastest
        const long r4 = i >> 2;
        const long row = r4 % M;
        const int k = (int)(r4 / M) * 32 + (int)(i & 3) * 8;
        float e[8];
        const float* src = X + row * ldx + k;
        if (k + 8 <= K) {
            const float4 a = *(const float4*)src, b = *(const float4*)(src + 4);
            e[0] = a.x; e[1] = a.y; e[2] = a.z; e[3] = a.w; e[4] = b.x; e[5] = b.y; e[6] = b.z; e[7] = b.w;
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) e[j] = (k + j < K) ? src[j] : 0.f;
        }
        bf16x8 h, m, l;
#pragma unroll
        for (int j = 0; j < 8; ++j) { __bf16 a, b, c; gsplit3(e[j], a, b, c); h[j] = a; m[j] = b; l[j] = c; }
        __bf16* d = out + ((long)(k >> 5) * M + row) * 32 + (k & 31);
        *(bf16x8*)d = h; *(bf16x8*)(d + ps) = m; *(bf16x8*)(d + 2 * ps) = l;
    }
}

struct GemmSArgs {
    const __bf16* As; const __bf16* Bs;
    long a_ps;                  // piece stride (elements)
    long M; int N, Kp;
    const float* bias; int relu, accumulate;
    float* Y; int ldy;
    __bf16* Ys; long y_ps; int Np;
    int ntm, ntn;
};

template <int MT, int NT, int WM, int WN, int KC, int MINB, bool PIPE>
__global__ __launch_bounds__(256, MINB) void gemm_bx6s_kernel(GemmSArgs g) {
    constexpr int APB = 3 * KC * 2 + 16;              // LDS bytes per A row: 3 pieces x KC bf16 + one 16-byte pad
    constexpr int TM = MT / WM / 32, TN = NT / WN / 32;
    constexpr int KS = KC / 16;
    constexpr int U = KC / 8;                         // 16-byte units per row and piece
    constexpr int NU = MT * 3 * U / 256;              // units per thread
    static_assert(MT * 3 * U % 256 == 0, "staging");
    extern __shared__ __attribute__((aligned(16))) char smem_c[];
    char* As = smem_c;

    const int tid = threadIdx.x;
    const int lane = tid & 63, li = lane & 31, lh = lane >> 5;
    const int wave = tid >> 6, wm = wave / WN, wn = wave % WN;
    const int swz = asr_xcd_swizzle(blockIdx.x, gridDim.x);
    const int tile_m = swz / g.ntn, tile_n = swz - tile_m * g.ntn;
    const long p0 = (long)tile_m * MT;
    const int n0 = tile_n * NT;
    const int Kp = g.Kp, N = g.N;

    floatx16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // staging map: unit u = tid + i * 256 -> (piece, row, c8) with c8 fastest, then row: a wave copies 64-byte row
    // segments of 16 consecutive rows of one piece per instruction
    const __bf16* asrc[NU];
    int adst[NU];
#pragma unroll
    for (int i = 0; i < NU; ++i) {
        const int u = tid + i * 256;
        const int pc = u / (MT * U), r2 = u - pc * (MT * U);
        const int row = r2 / U, c8 = r2 - row * U;
        long grow = p0 + row;
        if (grow >= g.M) grow = g.M - 1;              // rows past the end are computed and never stored
        asrc[i] = g.As + pc * g.a_ps + grow * 32 + c8 * 8;       // chunk kc adds kc * M * 32
        adst[i] = row * APB + pc * (KC * 2) + c8 * 16;
    }

    static_assert(KC == 32, "the split planes are stored in 32-deep K chunks");
    const int nkc = Kp / KC;
    const long cstride = g.M * 32;
    bf16x8 breg[KS][TN][3];
    // B arrives in fragment order (asr_split_weights): [3][Kp/16][ceil(N/32)][64 lanes][8], one coalesced 1 KB load each
    const int kst = Kp >> 4, nbt = (N + 31) >> 5;
    int nbb[TN];
#pragma unroll
    for (int b = 0; b < TN; ++b) { nbb[b] = (n0 >> 5) + wn * TN + b; if (nbb[b] >= nbt) nbb[b] = nbt - 1; }
    auto load_b = [&](bf16x8 (&dst)[TN][3], int kstep) {
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int pc = 0; pc < 3; ++pc)
                dst[b][pc] = *(const bf16x8*)(g.Bs + ((((long)pc * kst + kstep) * nbt + nbb[b]) * 64 + lane) * 8);
    };
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) load_b(breg[ks], ks);

    const char* abase0 = As + (wm * (TM * 32) + li) * APB + lh * 16;
    constexpr int ABUF = MT * APB;                    // PIPE: two A buffers, chunk kc lives in buffer kc & 1
    bf16x8 t[NU];
    if (PIPE) {
#pragma unroll
        for (int i = 0; i < NU; ++i) t[i] = *(const bf16x8*)(asrc[i]);
#pragma unroll
        for (int i = 0; i < NU; ++i) *(bf16x8*)(As + adst[i]) = t[i];
        __syncthreads();
    }
    for (int kc = 0; kc < nkc; ++kc) {
        const char* abase = abase0 + (PIPE ? (kc & 1) * ABUF : 0);
        const int kn = (kc + 1 < nkc) ? kc + 1 : kc;
        if (PIPE) {
            // the next chunk's global loads fly while this chunk is multiplied; they land in the other buffer, whose last
            // readers passed the barrier at the end of the previous iteration
            // (branch-free: the last iteration reloads its own chunk, so that the compiler's s_waitcnt counts stay exact)
#pragma unroll
            for (int i = 0; i < NU; ++i) t[i] = *(const bf16x8*)(asrc[i] + kn * cstride);
            __builtin_amdgcn_sched_barrier(0);         // keep the loads up here: the scheduler otherwise sinks them past the MFMAs
        } else {
#pragma unroll
            for (int i = 0; i < NU; ++i) t[i] = *(const bf16x8*)(asrc[i] + kc * cstride);
            __syncthreads();                           // the previous chunk's fragment reads are done
#pragma unroll
            for (int i = 0; i < NU; ++i) *(bf16x8*)(As + adst[i]) = t[i];
            __syncthreads();
        }
        // A fragments are read one (K-step, row block) ahead of the MFMAs that use them: with one or two waves per SIMD
        // nothing else hides the LDS latency (the compiler alone issues each read right before its first use)
        bf16x8 fa[2][3];
        auto read_a = [&](bf16x8 (&f)[3], int ks, int a) {
            f[0] = *(const bf16x8*)(abase + a * 32 * APB + ks * 32);
            f[1] = *(const bf16x8*)(abase + a * 32 * APB + ks * 32 + KC * 2);
            f[2] = *(const bf16x8*)(abase + a * 32 * APB + ks * 32 + 2 * KC * 2);
        };
        read_a(fa[0], 0, 0);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
            for (int a = 0; a < TM; ++a) {
                constexpr int dummy = 0; (void)dummy;
                const int cur = (ks * TM + a) & 1;
                if (a + 1 < TM) read_a(fa[cur ^ 1], ks, a + 1);
                else if (ks + 1 < KS) read_a(fa[cur ^ 1], ks + 1, 0);
                __builtin_amdgcn_sched_barrier(0);
                const bf16x8 ah = fa[cur][0], am = fa[cur][1], al = fa[cur][2];
#pragma unroll
                for (int b = 0; b < TN; ++b) {
                    floatx16 c = acc[a][b];
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, breg[ks][b][0], c, 0, 0, 0);      // small terms first
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, breg[ks][b][2], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, breg[ks][b][1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, breg[ks][b][0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, breg[ks][b][1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, breg[ks][b][0], c, 0, 0, 0);
                    acc[a][b] = c;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            load_b(breg[ks], kn * KS + ks);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (PIPE) {
            char* dstb = As + ((kc + 1) & 1) * ABUF;
#pragma unroll
            for (int i = 0; i < NU; ++i) *(bf16x8*)(dstb + adst[i]) = t[i];
            __syncthreads();
        }
    }

    // epilogue: transpose each 32x32 accumulator block through LDS, float4 stores (bias, ReLU, accumulate, split copy)
    __syncthreads();
    float* scratch = (float*)smem_c + wave * (32 * 33);
    const int c4 = lane & 7, rsub = lane >> 3;
#pragma unroll
    for (int b = 0; b < TN; ++b) {
        const int n = n0 + wn * (TN * 32) + b * 32 + c4 * 4;
        const bool ncolok = n < N;
        float4 bs = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ncolok && g.bias) bs = *(const float4*)(g.bias + n);
#pragma unroll
        for (int a = 0; a < TM; ++a) {
#pragma unroll
            for (int r = 0; r < 16; ++r) scratch[((r & 3) + 8 * (r >> 2) + 4 * lh) * 33 + li] = acc[a][b][r];
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int row = it * 8 + rsub;
                const long m = p0 + wm * (TM * 32) + a * 32 + row;
                const float* sp = scratch + row * 33 + c4 * 4;
                float4 v = make_float4(sp[0] + bs.x, sp[1] + bs.y, sp[2] + bs.z, sp[3] + bs.w);
                if (m >= g.M || !ncolok) continue;
                if (g.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                if (g.Y) {
                    float* o = g.Y + m * g.ldy + n;
                    if (g.accumulate) { const float4 p = *(const float4*)o; v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w; }
                    *(float4*)o = v;
                }
                if (g.Ys) {
                    bf16x4 h, md, l;
                    const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) { __bf16 x, y, z; gsplit3(e[j], x, y, z); h[j] = x; md[j] = y; l[j] = z; }
                    __bf16* d = g.Ys + ((long)(n >> 5) * g.M + m) * 32 + (n & 31);
                    *(bf16x4*)d = h; *(bf16x4*)(d + g.y_ps) = md; *(bf16x4*)(d + 2 * g.y_ps) = l;
                }
            }
        }
    }
}

template <int MT, int NT, int WM, int WN, int KC, int MINB, bool PIPE = false>
int launch_s(GemmSArgs g, hipStream_t st) {
    auto kern = gemm_bx6s_kernel<MT, NT, WM, WN, KC, MINB, PIPE>;
    size_t lds = (size_t)MT * (3 * KC * 2 + 16) * (PIPE ? 2 : 1);
    if (lds < 4 * 32 * 33 * sizeof(float)) lds = 4 * 32 * 33 * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    g.ntm = asr_cdiv(g.M, MT);
    g.ntn = asr_cdiv(g.N, NT);
    hipLaunchKernelGGL(kern, dim3(g.ntm * g.ntn), dim3(256), lds, st, g);
    ASR_CHECK_LAUNCH("gemm_bx6s");
    return ASR_OK;
}

}  // namespace

extern "C" size_t asr_split_rows_bytes(long M, int K) {
    const int Kp = (K + 31) / 32 * 32;
    return (size_t)3 * M * Kp * 2;
}

extern "C" int asr_split_rows(const float* X, long M, int K, int ldx, void* out, void* stream) {
    if (!X || !out || M < 1 || K < 1 || (ldx & 3) || (((uintptr_t)X | (uintptr_t)out) & 15)) return ASR_ERR_BAD_ARG;
    const int Kp = (K + 31) / 32 * 32;
    const long total = M * (Kp / 8);
    long nb = (total + 255) / 256;
    if (nb > 8192) nb = 8192;
    hipLaunchKernelGGL(split_rows_kernel, dim3((int)nb), dim3(256), 0, (hipStream_t)stream, X, M, K, ldx, Kp, (__bf16*)out);
    ASR_CHECK_LAUNCH("split_rows");
    return ASR_OK;
}

extern "C" int asr_gemm_bx6s(const void* As, const void* Bs, long M, int K, int N, const float* bias, int relu,
                             int accumulate, float* Y, int ldy, void* Ysplit, void* stream) {
    if (!As || !Bs || (!Y && !Ysplit) || M < 1 || K < 1 || N < 1 || (N & 3) || (Y && (ldy & 3))) return ASR_ERR_BAD_ARG;
    if (((uintptr_t)As | (uintptr_t)Bs | (uintptr_t)Y | (uintptr_t)Ysplit) & 15) return ASR_ERR_BAD_ARG;
    GemmSArgs g;
    g.Kp = (K + 31) / 32 * 32;
    g.As = (const __bf16*)As; g.Bs = (const __bf16*)Bs;
    g.a_ps = M * (long)g.Kp;
    g.M = M; g.N = N;
    g.bias = bias; g.relu = relu; g.accumulate = accumulate;
    g.Y = Y; g.ldy = ldy;
    g.Ys = (__bf16*)Ysplit; g.Np = (N + 31) / 32 * 32; g.y_ps = M * (long)g.Np;
    g.ntm = g.ntn = 0;
    hipStream_t st = (hipStream_t)stream;
    static int cfg = -1;
    if (cfg < 0) { const char* e = getenv("ASR_BX6S_CFG"); cfg = e ? atoi(e) : 0; }
    // tools/bench_bx6s.py (MI355X): 256x128 workgroup tiles (128x64 per wave, one wave per SIMD) with the double-buffered,
    // register-prefetched A tile win on every Transformer shape (1.43-1.55x the fp32 kernel); narrow outputs keep 2 waves
    if (N <= 32) return launch_s<256, 32, 4, 1, 32, 2>(g, st);
    if (N <= 64) return launch_s<256, 64, 2, 2, 32, 2>(g, st);
    switch (cfg) {
        case 1: return launch_s<256, 128, 2, 2, 32, 1>(g, st);
        case 2: return launch_s<128, 128, 2, 2, 32, 2, true>(g, st);
        case 3: return launch_s<256, 64, 2, 2, 32, 2>(g, st);
        default: return launch_s<256, 128, 2, 2, 32, 1, true>(g, st);
    }
}

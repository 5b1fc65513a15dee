// Dense GEMM of the Transformer / head layers on the fp32 MFMA pipe, "NT" form: both operands are K-contiguous,
//
//   C[m][n] = sum_k A[m][k] * Bt[n][k]           A [M][K] (pitch lda), Bt [N][K] (pitch ldb)
//
// which is what a data-gradient is as it stands (dX = dY . W^T with W [Kout][N] row-major: Bt = W) and what a forward
// layer becomes once its weights are transposed -- one batched launch per optimiser step (asr_transpose_batch).
// Design (round 3; replaces tap_gemm_kernel_v1 for these shapes, which staged both tiles through registers with two
// barriers per chunk and ran at 0.48-0.72 of the pipe on the Transformer's GEMMs):
//   * 128 x 128 tile, four waves of 64 x 64 (2 x 2 MFMA blocks of 32 x 32), K in chunks of 32;
//   * both tiles arrive by LDS-DMA (global_load_lds_dwordx4: no registers, no ds_write) into TWO buffer sets, the pieces of
//     chunk c + 1 issued between the MFMA groups of chunk c, ONE barrier per chunk; a piece is 8 rows x 128 bytes = 8 whole
//     cache lines;
//   * LDS rows are 128 bytes with no padding (the DMA writes 1 KB contiguously); the 16-byte chunks of a row are stored
//     XOR-swizzled by (row >> 1) & 7, chosen on the GLOBAL side of the DMA (a lane fetches the chunk that belongs in its LDS
//     slot), so that the ds_read_b128 of 32 consecutive rows at one logical chunk is conflict-free in each of the
//     instruction's 16-lane groups;
//   * the contraction index inside an 8-wide group is permuted (k = 4 * half + step) on BOTH operands, so one ds_read_b128
//     per operand block feeds four MFMAs;
//   * two workgroups per CU (65 KB of LDS each): the second wave of a SIMD covers barrier and DMA-issue stalls;
//   * the DMA is the BUFFER form (buffer_load_dwordx4 ... lds): resource and chunk offset are scalar, the per-lane offset a
//     constant register -- no vector instruction per piece -- and a 16-byte chunk past K in the ragged last chunk (K % 32, e.g.
//     the 6348-wide vocabulary) is sent out of the buffer's range, where the load returns zeros; row / column tails read a
//     clamped row and are dropped by the epilogue's row table / column test;
//   * epilogue = tap_epilogue (bias, ReLU / tanh, affine, accumulate, float4 stores through a per-wave LDS transpose).
#include "asr_common.h"
#include "reduce.h"
#include "tap_epilogue.h"

namespace {

typedef __attribute__((address_space(3))) float g1_lds_f;
typedef const __attribute__((address_space(1))) float g1_glb_f;

// The barrier that publishes a DMA'd tile: every wave first waits for ITS OWN pieces (explicitly -- whether hipcc adds the
// vmcnt(0) to a __syncthreads() by itself depends on what its alias analysis concluded about the LDS-DMA; in one kernel of this
// family it did not), then the barrier makes all pieces visible to all waves.
__device__ __forceinline__ void dma_barrier() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}

}  // namespace

struct Gemm1Gate { int mode, H, W; const float* a; float* dz; float* part; int* rows; int C = 0; };

namespace {

struct Gemm1Args {
    TapGemmArgs g;           // A, bias, scale, shift, outputs, M, K, N, lda, ldo_*, relu, accumulate
    const float* Bt;         // [N][K]
    int ldb;
};

constexpr int G1_KC = 32;                                // floats per chunk = one 128-byte row
constexpr int G1_TILE_F = 128 * G1_KC;                   // floats of one operand tile in LDS

// What a wave needs to issue its DMA pieces: two buffer resources (scalar registers) + 32-bit per-lane byte offsets that never
// change; the chunk's K offset is the instruction's scalar offset -- no vector instruction per piece (the flat form
// global_load_lds cost two 64-bit vector adds per piece here).  Piece p (0..15) of an operand tile = rows 8p .. 8p + 7; wave w
// issues pieces w, w + 4, w + 8, w + 12 of A and of B.  R = __amdgpu_buffer_rsrc_t (a type only the device pass can hold in a
// struct, hence the separate template parameters).
struct G1Dma {
    unsigned offa[4], offb[4];
    int wave, dchunk, K;
};

// (__amdgpu_buffer_rsrc_t is a type of the device pass only: the bodies that touch it are compiled there and nowhere else)
#if __HIP_DEVICE_COMPILE__
// one piece of chunk kc: j 0..3 = A pieces, 4..7 = B pieces.  TAIL: chunk kc may be the ragged last one, whose 16-byte chunks
// past K must read zeros: their offset is sent past the end of the buffer, where a buffer load returns 0 (0 * x is exact; the
// other operand is real tensor data, finite)
template <bool TAIL, class R>
__device__ __forceinline__ void g1_piece(R ra, R rb, const G1Dma& q, int kc, int j, float* __restrict__ set) {
    const int i = j & 3;
    const bool isb = j >= 4;
    float* dst = set + (isb ? G1_TILE_F : 0) + (q.wave + 4 * i) * 256;
    unsigned off = isb ? q.offb[i] : q.offa[i];
    if (TAIL) off = (kc * G1_KC + q.dchunk * 4 < q.K) ? off : 0xFFFFFFF0u;
    if (isb) __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (g1_lds_f*)dst, 16, off, kc * (G1_KC * 4), 0, 0);
    else __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (g1_lds_f*)dst, 16, off, kc * (G1_KC * 4), 0, 0);
}

// One chunk out of buffer set `set`.  MODE 1: the next chunk (a whole one) is fetched into `nset`, two DMA pieces per k-group
// behind the first and the third k-step's MFMAs; MODE 2: the same with run-time tests (is there a next chunk, is it the ragged
// one) -- only the last three chunks of a tile run this form; the steady-state loop body has no branch.  `set` / `nset` are
// __restrict__ parameters of an inlined function on purpose: without the alias scopes hipcc orders every LDS read behind the
// DMA in flight (s_waitcnt vmcnt(0) in front of each k-group) and nothing overlaps.
// NB = 32-column blocks per wave: 2 (128 x 128 tile) or 1 (128 x 64 tile, whose B tile is 8 pieces: slots 6 and 7 of a chunk's
// eight DMA slots stay empty).
template <int MODE, int NB, class R>
__device__ __forceinline__ void g1_chunk(const float* __restrict__ set, float* __restrict__ nset, R ra, R rb, const G1Dma& q, int kc, int nkc,
                                         bool ktail, int arow, int brow, const int (&xo)[4], floatx16 (&acc)[2][NB]) {
    const bool more = MODE == 1 || kc + 1 < nkc;
    const bool tail = MODE == 2 && ktail && kc + 2 == nkc;
    float4 av[2][2], bv[2][NB];
    auto load_frag = [&](int slot, int gk) {
#pragma unroll
        for (int a = 0; a < 2; ++a) av[slot][a] = *(const float4*)(set + arow + a * 32 * G1_KC + xo[gk]);
#pragma unroll
        for (int b = 0; b < NB; ++b) bv[slot][b] = *(const float4*)(set + brow + b * 32 * G1_KC + xo[gk]);
    };
    load_frag(0, 0);
#pragma unroll
    for (int gk = 0; gk < 4; ++gk) {
        const int cs = gk & 1;
        if (gk + 1 < 4) load_frag(cs ^ 1, gk + 1);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const float as = s == 0 ? av[cs][a].x : s == 1 ? av[cs][a].y : s == 2 ? av[cs][a].z : av[cs][a].w;
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    const float bs = s == 0 ? bv[cs][b].x : s == 1 ? bv[cs][b].y : s == 2 ? bv[cs][b].z : bv[cs][b].w;
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(as, bs, acc[a][b], 0, 0, 0);
                }
            }
            if (s == 0 || s == 2) {
                __builtin_amdgcn_sched_barrier(0);
                if (more && gk * 2 + (s >> 1) < 4 + 2 * NB) {
                    if (tail) g1_piece<true>(ra, rb, q, kc + 1, gk * 2 + (s >> 1), nset);
                    else g1_piece<false>(ra, rb, q, kc + 1, gk * 2 + (s >> 1), nset);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
}

#endif

// DIR (0 forward, 1 data-gradient) changes no code: it gives the two uses distinct symbols in a profile.
// NB = 2: 128 x 128 tile, two workgroups per CU.  NB = 1: 128 x 64 tile (waves 2 x 2 of 64 x 32), 49 KB of LDS, three
// workgroups per CU -- for the problems whose 128 x 128 grid is not a whole number of rounds over 512 slots (the launcher's
// rule); every output element sums its K products in the same order in both, so the choice changes no bit.
#if __HIP_DEVICE_COMPILE__
// SK: the contraction split over gridDim.y (asr_tap_gemm_nt_splitk): split z works on columns [z * K, (z + 1) * K) of both operands
// (K = the split's depth, a multiple of 32) and writes plane z of a [splits][M][N] slab through the row table.
template <int NB, bool GD, bool SK = false, bool RM = false>
__device__ __forceinline__ void gemm1_body(const Gemm1Args& args) {
    const TapGemmArgs& g = args.g;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int* rowa = (int*)smem;                              // [128] output row of out_a (or -1), [128] of out_y
    int* rowy = rowa + 128;
    float* bufs = smem + 256;                            // A0 | B0 | A1 | B1, later the epilogue's transpose scratch
    constexpr int SETF = G1_TILE_F + NB * 64 * G1_KC;    // floats of one buffer set: A tile | B tile
    float* set0 = bufs;
    float* set1 = bufs + SETF;

    const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, lh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int swz = asr_xcd_swizzle(blockIdx.x, gridDim.x);
    const int tile_m = swz / g.ntn, tile_n = swz - tile_m * g.ntn;
    const int m0 = tile_m * 128, n0 = tile_n * (64 * NB);
    const int K = g.K;

    if (tid < 128) {
        // rows are GEMM rows, or -- for a 1x1 convolution -- pixels of a padded plane, whose border pixels are not written
        const long p = (long)m0 + tid;
        int ra = -1, ry = -1;
        if (p < g.M) {
            if (GD) {
                // dense-layout gate: GEMM row = image b, pixel row h of the gated cell; the table holds its plane pixel at column 1
                const int b = (int)(p / g.gate_H), h = (int)(p - (long)b * g.gate_H);
                ra = (b * (g.gate_H + 1) + h + 1) * (g.gate_W + 1) + 1; ry = ra;
            } else if (g.H == 0) {
                ra = (int)p + (SK ? (int)blockIdx.y * g.M : 0); ry = ra;
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

    // ---- DMA side.  Lane l fills LDS slot (row 8p + l / 8, physical chunk l % 8) with the row's LOGICAL chunk
    // (l % 8) ^ ((row >> 1) & 7); (row >> 1) & 7 = (4 (p & 1) + l / 16) & 7 and p & 1 = w & 1 for all of a wave's pieces.
    // buffer resources over exactly the bytes the operands own: [0, ((rows - 1) * pitch + K) * 4)
    const long koff = SK ? (long)blockIdx.y * K : 0;
    auto rsa = __builtin_amdgcn_make_buffer_rsrc((void*)(g.A + koff), 0, (int)((((long)g.M - 1) * g.lda + K) * 4), 0x00020000);
    auto rsb = __builtin_amdgcn_make_buffer_rsrc((void*)(args.Bt + koff), 0, (int)((((long)g.N - 1) * args.ldb + K) * 4), 0x00020000);
    G1Dma q;
    q.wave = wave; q.K = K;
    q.dchunk = (lane & 7) ^ ((((wave & 1) << 2) + (lane >> 4)) & 7);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = 8 * (wave + 4 * i) + (lane >> 3);
        int ra = m0 + r; if (ra > g.M - 1) ra = g.M - 1;
        int rb = n0 + r; if (rb > g.N - 1) rb = g.N - 1;                 // (NB = 1: only i < 2 is used)
        q.offa[i] = (unsigned)(((long)ra * g.lda + q.dchunk * 4) * 4);
        q.offb[i] = (unsigned)(((long)rb * args.ldb + q.dchunk * 4) * 4);
    }
    const int nkc = (K + G1_KC - 1) / G1_KC;
    const bool ktail = (K & (G1_KC - 1)) != 0;

    // ---- MFMA side.  Lane (li, lh) of wave (wm, wn) reads, for k-group gk, logical chunk 2 gk + lh of rows
    // wm * 64 + a * 32 + li (A) and wn * 64 + b * 32 + li (B): physical chunk (2 gk + lh) ^ ((li >> 1) & 7).
    int xo[4];
#pragma unroll
    for (int gk = 0; gk < 4; ++gk) xo[gk] = ((2 * gk + lh) ^ ((li >> 1) & 7)) * 4;
    const int arow = (wm * 64 + li) * G1_KC, brow = G1_TILE_F + (wn * (32 * NB) + li) * G1_KC;

    floatx16 acc[2][NB];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    if (ktail && nkc == 1) {
#pragma unroll
        for (int j = 0; j < 4 + 2 * NB; ++j) g1_piece<true>(rsa, rsb, q, 0, j, set0);
    } else {
#pragma unroll
        for (int j = 0; j < 4 + 2 * NB; ++j) g1_piece<false>(rsa, rsb, q, 0, j, set0);
    }

    // a barrier per chunk: behind it chunk kc has landed (every wave waited for its own pieces: __syncthreads is also
    // s_waitcnt vmcnt(0)) and nobody reads the other set any more, which the chunk then refills
    const int last_plain = ktail ? nkc - 2 : nkc - 1;    // the last chunk that can be fetched without the K-tail test
    int kc = 0;
    for (; kc + 2 <= last_plain; kc += 2) {
        dma_barrier();
        g1_chunk<1, NB>(set0, set1, rsa, rsb, q, kc, nkc, ktail, arow, brow, xo, acc);
        dma_barrier();
        g1_chunk<1, NB>(set1, set0, rsa, rsb, q, kc + 1, nkc, ktail, arow, brow, xo, acc);
    }
    if (kc < nkc) { dma_barrier(); g1_chunk<2, NB>(set0, set1, rsa, rsb, q, kc, nkc, ktail, arow, brow, xo, acc); ++kc; }
    if (kc < nkc) { dma_barrier(); g1_chunk<2, NB>(set1, set0, rsa, rsb, q, kc, nkc, ktail, arow, brow, xo, acc); ++kc; }
    if (kc < nkc) { dma_barrier(); g1_chunk<2, NB>(set0, set1, rsa, rsb, q, kc, nkc, ktail, arow, brow, xo, acc); ++kc; }

    __syncthreads();                                     // the tiles are dead: their space is the epilogue's scratch
    if (GD) tap_epilogue_gated<2, NB, true>(g, acc, bufs + wave * (32 * 33), rowa, rowy, wm * 64, n0 + wn * (32 * NB), lane, tile_m * 2 + wm);
    else tap_epilogue<2, NB, RM>(g, acc, bufs + wave * (32 * 33), rowa, rowy, wm * 64, n0 + wn * (32 * NB), lane, tile_m * 2 + wm);
}
#endif

template <int DIR, int NB>
__global__ __launch_bounds__(256, NB == 2 ? 2 : 3) void gemm1_kernel(Gemm1Args args) {
#if __HIP_DEVICE_COMPILE__
    gemm1_body<NB, false>(args);
#endif
}

// the data-gradient of a dense head with the backward prologue of the cell in front in its epilogue (gate mode 5, asr_tap_gemm_gated_dense):
// a symbol of its own, so that gemm1_kernel's code does not carry the mode
template <int NB>
__global__ __launch_bounds__(256, NB == 2 ? 2 : 3) void gemm1_dense_gate_kernel(Gemm1Args args) {
#if __HIP_DEVICE_COMPILE__
    gemm1_body<NB, true>(args);
#endif
}



// the data-gradient of a dense layer whose input is a Dense(relu) output: the ReLU backward of that layer in the epilogue
// (asr_tap_gemm_relu_bwd; tap_epilogue's RMASK)
template <int NB>
__global__ __launch_bounds__(256, NB == 2 ? 2 : 3) void gemm1_relumask_kernel(Gemm1Args args) {
#if __HIP_DEVICE_COMPILE__
    gemm1_body<NB, false, false, true>(args);
#endif
}

template <int NB>
__global__ __launch_bounds__(256, NB == 2 ? 2 : 3) void gemm1_splitk_kernel(Gemm1Args args) {
#if __HIP_DEVICE_COMPILE__
    gemm1_body<NB, false, true>(args);
#endif
}

// ---- dense weight gradient, "TN" form:  dW[k][n] = sum_m A[m][k] * dZ[m][n]  (tf.layers.dense backward, and the 1x1 conv's)
// Same machinery as gemm1_kernel -- buffer-form LDS-DMA into two run buffers, pieces between the MFMAs, one barrier per run,
// two workgroups per CU -- on the operands as they lie: a run is 32 rows of the [row][channel] tiles (A: KT channels, dZ: NT),
// an MFMA k-step contracts the row pair (2s, 2s + 1) and its operands are conflict-free ds_read_b32 (32 consecutive floats
// per half wave, the two halves one row apart).  The rows of a piece go into the instruction's SCALAR offset, the per-lane
// offset (row within the piece, 16-byte chunk) never changes: no vector instruction per piece.  Ragged edges read zeros
// through out-of-range per-lane offsets: channels past K / N for the whole workgroup, rows past the end in the last run.
// The pixel axis is split into chunks over the grid (make_plan6's beat model); chunk partials go to a slab that
// sum_chunks_kernel folds in a fixed order (tap_wgrad.hip): bitwise reproducible, no float atomics.
//   <2, 2, 2, 2>: 128 x 128 tile, waves 2 x 2 of 64 x 64 -- the Transformer / head layers;
//   <4, 1, 1, 1>: 128 x 32 tile, waves 4 x 1 of 32 x 32 -- outputs of up to 32 channels (the NiN 1x1 conv).
struct Wgrad1Args {
    const float* A; const float* Z; float* out;   // out: dW (one chunk) or the partial slab [chunk][K][N]
    int M, K, N, lda, ldz;
    int pch;                                      // rows per chunk (multiple of 32)
};

#if __HIP_DEVICE_COMPILE__
template <int WM, int WN, int TKB, int TNB, bool TAIL, class R>
__device__ __forceinline__ void w1_piece(R ra, R rz, int j, float* __restrict__ set, int wave, unsigned voa, unsigned voz,
                                         long row0, long rend, int lda, int ldz, int lane) {
    constexpr int KT = WM * TKB * 32, NT = WN * TNB * 32;
    constexpr int ARP = 256 / KT, ZRP = 256 / NT, AP4 = 32 / ARP / 4;        // rows per piece; A pieces per wave
    const bool isz = j >= AP4;
    const int p = wave + 4 * (isz ? j - AP4 : j);
    float* dst = set + (isz ? 32 * KT : 0) + p * 256;
    const long r = row0 + (long)p * (isz ? ZRP : ARP);                        // first row of the piece (uniform)
    unsigned vo = isz ? voz : voa;
    if (TAIL) { if (r + lane / (isz ? NT / 4 : KT / 4) >= rend) vo = 0xFFFFFFF0u; }
    if (isz) __builtin_amdgcn_raw_ptr_buffer_load_lds(rz, (g1_lds_f*)dst, 16, vo, (int)(r * ldz * 4), 0, 0);
    else __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (g1_lds_f*)dst, 16, vo, (int)(r * lda * 4), 0, 0);
}

// one run of 32 rows out of `set`; MODE 1: the next run (a whole one) is fetched into `nset`; MODE 2: run-time tests (is there
// a next run, is it ragged) -- the last three runs of a chunk
template <int WM, int WN, int TKB, int TNB, int MODE, class R>
__device__ __forceinline__ void w1_run(const float* __restrict__ set, float* __restrict__ nset, R ra, R rz, int wave, unsigned voa,
                                       unsigned voz, long nrow0, long rend, int lda, int ldz, int lane, bool more, int aoff, int zoff,
                                       floatx16 (&acc)[TKB][TNB]) {
    constexpr int KT = WM * TKB * 32, NT = WN * TNB * 32;
    constexpr int NPW = (32 / (256 / KT) + 32 / (256 / NT)) / 4;              // pieces per wave and run
    const bool tail = MODE == 2 && nrow0 + 32 > rend;
    const bool go = MODE == 1 || more;
    int issued = 0;
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        float av[TKB], bv[TNB];
#pragma unroll
        for (int a = 0; a < TKB; ++a) av[a] = set[aoff + s * 2 * KT + a * 32];
#pragma unroll
        for (int b = 0; b < TNB; ++b) bv[b] = set[zoff + s * 2 * NT + b * 32];
#pragma unroll
        for (int a = 0; a < TKB; ++a)
#pragma unroll
            for (int b = 0; b < TNB; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a], bv[b], acc[a][b], 0, 0, 0);
        // the next run's pieces, spread evenly over the sixteen k-steps
        if (issued < NPW && (s + 1) * NPW >= (issued + 1) * 16) {
            __builtin_amdgcn_sched_barrier(0);
            if (go) {
                if (tail) w1_piece<WM, WN, TKB, TNB, true>(ra, rz, issued, nset, wave, voa, voz, nrow0, rend, lda, ldz, lane);
                else w1_piece<WM, WN, TKB, TNB, false>(ra, rz, issued, nset, wave, voa, voz, nrow0, rend, lda, ldz, lane);
            }
            __builtin_amdgcn_sched_barrier(0);
            ++issued;
        }
    }
}
#endif

template <int WM, int WN, int TKB, int TNB>
__global__ __launch_bounds__(256, 2) void wgrad1_kernel(Wgrad1Args g) {
#if __HIP_DEVICE_COMPILE__
    constexpr int KT = WM * TKB * 32, NT = WN * TNB * 32;
    constexpr int NPW = (32 / (256 / KT) + 32 / (256 / NT)) / 4;
    constexpr int SET_F = 32 * (KT + NT);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* set0 = smem;
    float* set1 = smem + SET_F;
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, lh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int k0 = blockIdx.y * KT, n0 = blockIdx.z * NT;
    const long cbeg = (long)blockIdx.x * g.pch;
    const long cend = (cbeg + g.pch < g.M) ? cbeg + g.pch : g.M;
    const int nruns = (int)((cend - cbeg + 31) >> 5);

    auto ra = __builtin_amdgcn_make_buffer_rsrc((void*)g.A, 0, (int)((((long)g.M - 1) * g.lda + g.K) * 4), 0x00020000);
    auto rz = __builtin_amdgcn_make_buffer_rsrc((void*)g.Z, 0, (int)((((long)g.M - 1) * g.ldz + g.N) * 4), 0x00020000);
    // per-lane offset inside a piece: row lane / (T / 4), 16-byte chunk lane % (T / 4); channels past K / N read zeros
    const int ac = lane % (KT / 4), zc = lane % (NT / 4);
    const unsigned voa = (k0 + ac * 4 < g.K) ? (unsigned)((((long)(lane / (KT / 4)) * g.lda) + k0 + ac * 4) * 4) : 0xFFFFFFF0u;
    const unsigned voz = (n0 + zc * 4 < g.N) ? (unsigned)((((long)(lane / (NT / 4)) * g.ldz) + n0 + zc * 4) * 4) : 0xFFFFFFF0u;
    // operand reads: row pair member lh of k-step s = row 2s + lh of the run
    const int aoff = lh * KT + wm * (TKB * 32) + li, zoff = 32 * KT + lh * NT + wn * (TNB * 32) + li;

    floatx16 acc[TKB][TNB];
#pragma unroll
    for (int a = 0; a < TKB; ++a)
#pragma unroll
        for (int b = 0; b < TNB; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    if (cbeg + 32 > cend) {
#pragma unroll
        for (int j = 0; j < NPW; ++j) w1_piece<WM, WN, TKB, TNB, true>(ra, rz, j, set0, wave, voa, voz, cbeg, cend, g.lda, g.ldz, lane);
    } else {
#pragma unroll
        for (int j = 0; j < NPW; ++j) w1_piece<WM, WN, TKB, TNB, false>(ra, rz, j, set0, wave, voa, voz, cbeg, cend, g.lda, g.ldz, lane);
    }
    // runs 0 .. nfull - 1 are whole; the steady-state pair loop only fetches whole runs
    const int nfull = (int)((cend - cbeg) >> 5);
    int r = 0;
    for (; r + 2 <= nfull - 1; r += 2) {
        dma_barrier();
        w1_run<WM, WN, TKB, TNB, 1>(set0, set1, ra, rz, wave, voa, voz, cbeg + (long)(r + 1) * 32, cend, g.lda, g.ldz, lane, true, aoff, zoff, acc);
        dma_barrier();
        w1_run<WM, WN, TKB, TNB, 1>(set1, set0, ra, rz, wave, voa, voz, cbeg + (long)(r + 2) * 32, cend, g.lda, g.ldz, lane, true, aoff, zoff, acc);
    }
    if (r < nruns) { dma_barrier(); w1_run<WM, WN, TKB, TNB, 2>(set0, set1, ra, rz, wave, voa, voz, cbeg + (long)(r + 1) * 32, cend, g.lda, g.ldz, lane, r + 1 < nruns, aoff, zoff, acc); ++r; }
    if (r < nruns) { dma_barrier(); w1_run<WM, WN, TKB, TNB, 2>(set1, set0, ra, rz, wave, voa, voz, cbeg + (long)(r + 1) * 32, cend, g.lda, g.ldz, lane, r + 1 < nruns, aoff, zoff, acc); ++r; }
    if (r < nruns) { dma_barrier(); w1_run<WM, WN, TKB, TNB, 2>(set0, set1, ra, rz, wave, voa, voz, cbeg + (long)(r + 1) * 32, cend, g.lda, g.ldz, lane, r + 1 < nruns, aoff, zoff, acc); ++r; }
    if (r < nruns) { dma_barrier(); w1_run<WM, WN, TKB, TNB, 2>(set1, set0, ra, rz, wave, voa, voz, cbeg + (long)(r + 1) * 32, cend, g.lda, g.ldz, lane, r + 1 < nruns, aoff, zoff, acc); ++r; }

    float* out = g.out + (long)blockIdx.x * g.K * g.N;
#pragma unroll
    for (int a = 0; a < TKB; ++a)
#pragma unroll
        for (int b = 0; b < TNB; ++b) {
            const int n = n0 + wn * (TNB * 32) + b * 32 + li;
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) {
                const int k = k0 + wm * (TKB * 32) + a * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * lh;
                if (k < g.K && n < g.N) out[(long)k * g.N + n] = acc[a][b][rr];
            }
        }
#endif
}

// dst[c][r] = src[r][c], 32 x 32 tiles through LDS; several matrices per launch (blockIdx.y)
__global__ __launch_bounds__(256) void transpose_batch_kernel(const asr_copy2d_item* __restrict__ items) {
    __shared__ float t[32][33];
    const asr_copy2d_item it = items[blockIdx.y];
    const int tr = (it.rows + 31) >> 5, tc = (it.cols + 31) >> 5;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int tile = blockIdx.x; tile < tr * tc; tile += gridDim.x) {
        const int r0 = (tile / tc) * 32, c0 = (tile % tc) * 32;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = r0 + ty + 8 * j, c = c0 + tx;
            t[ty + 8 * j][tx] = (r < it.rows && c < it.cols) ? it.src[(long)r * it.lds + c] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = c0 + ty + 8 * j, r = r0 + tx;
            if (c < it.cols && r < it.rows) it.dst[(long)c * it.ldd + r] = t[tx][ty + 8 * j];
        }
        __syncthreads();
    }
}

}  // namespace

// ---- launcher shared by asr_tap_gemm (1 tap, wmode 1: the data-gradients) and asr_tap_gemm_nt (forward on transposed weights)
bool asr_gemm1_eligible(const asr_gemm_desc* d, const float* A, const float* Bt, int ldb) {
    if (d->ntaps != 1 || (d->H > 0 && d->M != d->B * (d->H + 1) * (d->W + 1))) return false;
    if ((d->K & 3) || (d->N & 3) || (d->lda & 3) || (ldb & 3) || d->K < 32 || d->N < 64) return false;
    if ((((uintptr_t)A) | ((uintptr_t)Bt)) & 15) return false;
    if ((long)d->M * d->lda * 4 >= (1L << 31) || (long)d->N * ldb * 4 >= (1L << 31)) return false;      // 32-bit buffer offsets / range
    // Small grids too: with one workgroup on a CU the register-staged kernels expose a global-memory round trip per chunk
    // (6400 x 512 x 512, 200 tiles: 102 us = 33 TFLOP/s on tap_gemm_kernel_v1), the DMA of the next chunk hides it.  Only
    // problems of a few dozen tiles keep the 64 x 64 tiles of tap_gemm_kernel_v1 (more workgroups than 128 x 128 tiles give).
    // (This rule and gemm1_blocks() look at the row count, i.e. at the batch.  What keeps "an utterance gives the same bits alone
    // and inside a batch" true is that gemm1_kernel with either tile and tap_gemm_kernel_v1 add the K products of an output element
    // in the same order: tests/test_gemm1_gpu.py::test_the_tile_choice_changes_no_bit holds all three to torch.equal.)
    return (long)asr_cdiv(d->M, 128) * asr_cdiv(d->N, 128) >= 48;
}

// 32-column blocks per wave (tile width / 64): see gemm1_kernel.  A function of the problem's shape alone: the 128 x 128 tile
// when its grid fills whole rounds of the chip's 512 workgroup slots (>= 0.96 of the last one), else the 128 x 64 tile, whose
// grid is twice as fine and runs three to a CU (tools/bench_gemm1_tiles.py, profiles/r04_gemm1_tiles.txt: 6400 x 512 -> 1536,
// 600 tiles: 90 -> 111 TFLOP/s; 6400 x 3200 -> 256, 100 tiles: 52 -> 102; 32768-row problems: unchanged, they keep 128 x 128).
static int gemm1_blocks(int M, int N) {
    const long tiles = (long)asr_cdiv(M, 128) * asr_cdiv(N, 128);
    const long rounds = (tiles + 511) / 512;
    return tiles * 100 >= rounds * 512 * 96 ? 2 : 1;
}

// gate: the fused backward prologue of asr_tap_gemm_gated (tap_epilogue_gated), or null
int asr_gemm1_launch(const asr_gemm_desc* d, const float* A, const float* Bt, int ldb, const float* bias, const float* scale,
                     const float* shift, float* out_a, float* out_y, int dir, void* stream, const Gemm1Gate* gate) {
    Gemm1Args ga;
    TapGemmArgs& a = ga.g;
    a.A = A; a.W = nullptr; a.bias = bias; a.scale = scale; a.shift = shift;
    a.out_a = out_a; a.out_y = out_y;
    a.M = d->M; a.K = d->K; a.N = d->N; a.lda = d->lda; a.ldw = ldb;
    a.ldo_a = d->ldo_a; a.ldo_y = d->ldo_y;
    a.H = d->H > 0 ? d->H : 0; a.Wd = d->W; a.WP = d->W + 1; a.HPWP = (d->H + 1) * (d->W + 1); a.halo = 0; a.rmin = 0; a.rmax = d->M;
    a.relu = d->relu; a.accumulate = d->accumulate; a.y_unpadded = d->H > 0 ? d->y_unpadded : 0;
    a.gate_mode = 0; a.gate_H = a.gate_W = 0; a.gate_a = nullptr; a.gate_dz = nullptr; a.gate_part = nullptr; a.gate_rows = nullptr;
    a.nt_store = 0;
    int nb = gemm1_blocks(d->M, d->N);
    // a dense gate over a short contraction (6400 x 6400 x 128 of acoustic_model.py:49: four chunks, then an epilogue that reads and writes
    // 64 KB per tile): three narrower workgroups per CU cover each other's epilogues better than two (194 -> 171 us; K = 1536: no difference)
    if (gate && gate->mode == 5 && d->K <= 256) nb = 1;
    a.ntm = asr_cdiv(d->M, 128); a.ntn = asr_cdiv(d->N, 64 * nb);
    if (gate) {
        a.gate_mode = gate->mode; a.gate_H = gate->H; a.gate_W = gate->W; a.gate_a = gate->a; a.gate_dz = gate->dz; a.gate_part = gate->part;
        if (gate->mode == 5) a.halo = gate->C;          // the gated cell's channels (TapGemmArgs: gate mode 5)
        // one partial row per (tile row, wave row[, pixel column])
        if (!asr_gate_rows_fit(gate->rows, a.ntm * 2 * (gate->mode == 5 ? gate->W : 1))) return ASR_ERR_UNSUPPORTED;
    }
    ga.Bt = Bt; ga.ldb = ldb;
    const size_t lds = (size_t)(256 + 2 * (G1_TILE_F + nb * 64 * G1_KC)) * sizeof(float);
    static_assert(4 * 32 * 33 <= 2 * (G1_TILE_F + 64 * G1_KC), "epilogue scratch fits the tile buffers");
    typedef void (*kern_t)(Gemm1Args);
    static const kern_t kerns[6] = {gemm1_kernel<0, 1>, gemm1_kernel<0, 2>, gemm1_kernel<1, 1>, gemm1_kernel<1, 2>, gemm1_dense_gate_kernel<1>, gemm1_dense_gate_kernel<2>};
    static const char* const names[6] = {"gemm1_kernel<0, 1>", "gemm1_kernel<0, 2>", "gemm1_kernel<1, 1>", "gemm1_kernel<1, 2>", "gemm1_dense_gate_kernel<1>",
                                         "gemm1_dense_gate_kernel<2>"};
    static bool attr[6] = {false, false, false, false, false, false};
    const int ki = (gate && gate->mode == 5) ? 4 + (nb - 1) : (dir ? 2 : 0) + (nb - 1);
    if (!attr[ki]) { (void)hipFuncSetAttribute((const void*)kerns[ki], hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr[ki] = true; }
    hipLaunchKernelGGL(kerns[ki], dim3(a.ntm * a.ntn), dim3(256), lds, (hipStream_t)stream, ga);
    asr_set_last_kernel(names[ki]);
    ASR_CHECK_LAUNCH("gemm1");
    return ASR_OK;
}

// Dense weight gradient on wgrad1_kernel: called by asr_tap_wgrad (tap_wgrad.hip), which owns the chunk plan and the slab sum.
bool asr_wgrad1_eligible(const asr_gemm_desc* d, const float* A, const float* dZ, int ldz) {
    if (d->ntaps != 1 || (d->K & 3) || (d->N & 3) || (d->lda & 3) || (ldz & 3)) return false;
    if ((((uintptr_t)A) | ((uintptr_t)dZ)) & 15) return false;
    if ((long)d->M * d->lda * 4 >= (1L << 31) || (long)d->M * ldz * 4 >= (1L << 31)) return false;       // 32-bit buffer offsets
    return d->M >= 64;
}

int asr_wgrad1_launch(const asr_gemm_desc* d, const float* A, const float* dZ, int ldz, float* out, int pch, int nchunks, void* stream) {
    Wgrad1Args a;
    a.A = A; a.Z = dZ; a.out = out; a.M = d->M; a.K = d->K; a.N = d->N; a.lda = d->lda; a.ldz = ldz; a.pch = pch;
    if (pch < 32 || (pch & 31)) return ASR_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (d->N <= 32) {
        auto kern = wgrad1_kernel<4, 1, 1, 1>;
        const size_t lds = (size_t)2 * 32 * (128 + 32) * sizeof(float);
        static bool attr = false;
        if (!attr) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr = true; }
        hipLaunchKernelGGL(kern, dim3(nchunks, asr_cdiv(d->K, 128), 1), dim3(256), lds, st, a);
        ASR_NOTE_KERNEL("wgrad1_kernel<4, 1, 1, 1>");
    } else {
        auto kern = wgrad1_kernel<2, 2, 2, 2>;
        const size_t lds = (size_t)2 * 32 * (128 + 128) * sizeof(float);
        static bool attr = false;
        if (!attr) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr = true; }
        hipLaunchKernelGGL(kern, dim3(nchunks, asr_cdiv(d->K, 128), asr_cdiv(d->N, 128)), dim3(256), lds, st, a);
        ASR_NOTE_KERNEL("wgrad1_kernel<2, 2, 2, 2>");
    }
    ASR_CHECK_LAUNCH("wgrad1");
    return ASR_OK;
}

extern "C" int asr_tap_gemm(const asr_gemm_desc* d, const float* A, const float* W, const float* bias, const float* scale,
                            const float* shift, float* out_a, float* out_y, void* stream);

extern "C" int asr_tap_gemm_nt(const asr_gemm_desc* d, const float* A, const float* W, const float* Wt, int ldwt,
                               const float* bias, const float* scale, const float* shift, float* out_a, float* out_y, void* stream) {
    if (!d || !A || !W || !Wt || (!out_a && !out_y) || d->wmode != 0 || d->ntaps != 1) return ASR_ERR_BAD_ARG;
    if (d->M <= 0 || d->K <= 0 || d->N <= 0 || ldwt < d->K) return ASR_ERR_BAD_ARG;
    if (asr_gemm1_eligible(d, A, Wt, ldwt)) return asr_gemm1_launch(d, A, Wt, ldwt, bias, scale, shift, out_a, out_y, 0, stream, nullptr);
    return asr_tap_gemm(d, A, W, bias, scale, shift, out_a, out_y, stream);
}

extern "C" int asr_relu_bwd(const float* dy, const float* h, size_t n, float* dz, void* stream);

// dX[rows][K_in] = (dY[rows][N_out] . W[K_in][N_out]^T) masked by (H[rows][K_in] > 0): the data-gradient of a dense layer (asr_tap_gemm with
// wmode 1: d->K = the layer's output width, d->N its input width, W the layer's kernel) followed by asr_relu_bwd against the activation H
// of the Dense(relu) layer in front -- in ONE launch where the LDS-DMA kernel takes the shape (the Transformer's feed-forward blocks:
// 32768 x 2048 <- 512: no 268 MB pass over the result), as the two calls otherwise.  The same bits either way.  d->accumulate must be 0.
extern "C" int asr_tap_gemm_relu_bwd(const asr_gemm_desc* d, const float* dY, const float* W, const float* H, float* dX, void* stream) {
    if (!d || !dY || !W || !H || !dX || d->ntaps != 1 || d->wmode != 1 || d->H > 0 || d->accumulate || d->ldo_y < d->N) return ASR_ERR_BAD_ARG;
    if (!asr_gemm1_eligible(d, dY, W, d->ldw) || ((((uintptr_t)H) | ((uintptr_t)dX)) & 15) || (d->ldo_y & 3)) {
        if (d->ldo_y != d->N) return ASR_ERR_UNSUPPORTED;          // the mask pass of this route is a flat one: refused BEFORE dX is touched
        const int rc = asr_tap_gemm(d, dY, W, nullptr, nullptr, nullptr, nullptr, dX, stream);
        if (rc != ASR_OK) return rc;
        return asr_relu_bwd(dX, H, (size_t)d->M * d->N, dX, stream);
    }
    Gemm1Args ga;
    TapGemmArgs& a = ga.g;
    a.A = dY; a.W = nullptr; a.bias = nullptr; a.scale = nullptr; a.shift = nullptr;
    a.out_a = nullptr; a.out_y = dX;
    a.M = d->M; a.K = d->K; a.N = d->N; a.lda = d->lda; a.ldw = d->ldw;
    a.ldo_a = 0; a.ldo_y = d->ldo_y;
    a.H = 0; a.Wd = 0; a.WP = 1; a.HPWP = 1; a.halo = 0; a.rmin = 0; a.rmax = d->M;
    a.relu = 0; a.accumulate = 0; a.y_unpadded = 0;
    a.gate_mode = 0; a.gate_H = a.gate_W = 0; a.gate_a = H; a.gate_dz = nullptr; a.gate_part = nullptr; a.gate_rows = nullptr;
    a.nt_store = 0;
    const int nb = gemm1_blocks(d->M, d->N);
    a.ntm = asr_cdiv(d->M, 128); a.ntn = asr_cdiv(d->N, 64 * nb);
    ga.Bt = W; ga.ldb = d->ldw;
    const size_t lds = (size_t)(256 + 2 * (G1_TILE_F + nb * 64 * G1_KC)) * sizeof(float);
    typedef void (*kern_t)(Gemm1Args);
    static const kern_t kerns[2] = {gemm1_relumask_kernel<1>, gemm1_relumask_kernel<2>};
    static bool attr[2] = {false, false};
    if (!attr[nb - 1]) { (void)hipFuncSetAttribute((const void*)kerns[nb - 1], hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr[nb - 1] = true; }
    hipLaunchKernelGGL(kerns[nb - 1], dim3(a.ntm * a.ntn), dim3(256), lds, (hipStream_t)stream, ga);
    asr_set_last_kernel(nb == 2 ? "gemm1_relumask_kernel<2>" : "gemm1_relumask_kernel<1>");
    ASR_CHECK_LAUNCH("gemm1_relumask");
    return ASR_OK;
}

// second pass of the split-K forms (tap_gemm.hip)
int asr_splitk_reduce_launch(const float* slab, int splits, const asr_gemm_desc* d, const float* bias, const float* scale, const float* shift,
                             float* out_a, float* out_y, void* stream);

// Dense GEMM, both operands K-contiguous (A [M][K], Bt [N][K]), the contraction split `splits` ways over the grid: for a deep K and few
// output tiles -- the 6400 -> 128 hidden dense of acoustic_model.py:53 (50 tiles of 128 x 128; ten splits of twenty chunks fill the
// chip's 512 workgroup slots once) and the 1536 -> 128 data-gradient of the layer behind it.  The LDS-DMA kernel of this file per split
// (gemm1_splitk_kernel), then the fixed-order second pass of asr_tap_gemm_splitk: reproducible; equal to asr_tap_gemm_nt up to the
// order of the K sum.
extern "C" size_t asr_tap_gemm_nt_splitk_workspace(const asr_gemm_desc* d, int splits) {
    return (d && splits > 1) ? (size_t)splits * d->M * d->N * sizeof(float) : 0;
}

extern "C" int asr_tap_gemm_nt_splitk(const asr_gemm_desc* d, const float* A, const float* Bt, int ldb, const float* bias, const float* scale,
                                      const float* shift, float* out_a, float* out_y, int splits, void* workspace, size_t workspace_bytes, void* stream) {
    if (!d || !A || !Bt || (!out_a && !out_y) || !workspace) return ASR_ERR_BAD_ARG;
    if (d->ntaps != 1 || d->H > 0 || d->y_unpadded || splits < 2 || splits > 16) return ASR_ERR_BAD_ARG;
    if (workspace_bytes < asr_tap_gemm_nt_splitk_workspace(d, splits)) return ASR_ERR_BAD_ARG;      // the slab is splits x M x N floats
    if ((d->K % (splits * 32)) || (d->N & 3) || (d->lda & 3) || (ldb & 3) || ldb < d->K || d->N < 64 ||
        (((uintptr_t)A | (uintptr_t)Bt | (uintptr_t)workspace) & 15)) return ASR_ERR_BAD_ARG;
    if ((long)d->M * d->lda * 4 >= (1L << 31) || (long)d->N * ldb * 4 >= (1L << 31) || (long)splits * d->M >= (1L << 31) / d->N) return ASR_ERR_UNSUPPORTED;
    Gemm1Args ga;
    TapGemmArgs& a = ga.g;
    a.A = A; a.W = nullptr; a.bias = nullptr; a.scale = nullptr; a.shift = nullptr;
    a.out_a = (float*)workspace; a.out_y = nullptr;
    a.M = d->M; a.K = d->K / splits; a.N = d->N; a.lda = d->lda; a.ldw = ldb;
    a.ldo_a = d->N; a.ldo_y = 0;
    a.H = 0; a.Wd = 0; a.WP = 1; a.HPWP = 1; a.halo = 0; a.rmin = 0; a.rmax = d->M;
    a.relu = 0; a.accumulate = 0; a.y_unpadded = 0;
    a.gate_mode = 0; a.gate_H = a.gate_W = 0; a.gate_a = nullptr; a.gate_dz = nullptr; a.gate_part = nullptr; a.gate_rows = nullptr;
    a.nt_store = 0;
    // the tile whose grid x splits comes closest to whole rounds of the chip's workgroup slots (512 of 128 x 128, 768 of 128 x 64)
    const long t2 = (long)asr_cdiv(d->M, 128) * asr_cdiv(d->N, 128) * splits, t1 = (long)asr_cdiv(d->M, 128) * asr_cdiv(d->N, 64) * splits;
    const double f2 = (double)t2 / (512.0 * ((t2 + 511) / 512)), f1 = (double)t1 / (768.0 * ((t1 + 767) / 768));
    const int nb = f2 >= f1 ? 2 : 1;
    a.ntm = asr_cdiv(d->M, 128); a.ntn = asr_cdiv(d->N, 64 * nb);
    ga.Bt = Bt; ga.ldb = ldb;
    const size_t lds = (size_t)(256 + 2 * (G1_TILE_F + nb * 64 * G1_KC)) * sizeof(float);
    typedef void (*kern_t)(Gemm1Args);
    static const kern_t kerns[2] = {gemm1_splitk_kernel<1>, gemm1_splitk_kernel<2>};
    static bool attr[2] = {false, false};
    if (!attr[nb - 1]) { (void)hipFuncSetAttribute((const void*)kerns[nb - 1], hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr[nb - 1] = true; }
    hipLaunchKernelGGL(kerns[nb - 1], dim3(a.ntm * a.ntn, splits), dim3(256), lds, (hipStream_t)stream, ga);
    asr_set_last_kernel(nb == 2 ? "gemm1_splitk_kernel<2>" : "gemm1_splitk_kernel<1>");
    ASR_CHECK_LAUNCH("gemm1_splitk");
    return asr_splitk_reduce_launch((const float*)workspace, splits, d, bias, scale, shift, out_a, out_y, stream);
}

extern "C" int asr_transpose_batch(const asr_copy2d_item* items_dev, int n_items, int max_elems, void* stream) {
    if (!items_dev || n_items < 1 || n_items > 65535 || max_elems < 1) return ASR_ERR_BAD_ARG;
    int blocks = asr_cdiv(max_elems, 1024);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(transpose_batch_kernel, dim3(blocks, n_items), dim3(256), 0, (hipStream_t)stream, items_dev);
    ASR_CHECK_LAUNCH("transpose_batch");
    return ASR_OK;
}

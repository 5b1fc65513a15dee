// Prototype of the CHUNK LOOP of a Winograd F(4x4,3x3) forward / data-gradient kernel in fp32 on the MFMA pipe (VERDICT r4 item 2;
// reference layers: lm_and_am/model/acoustic_model.py:44-46, acoustic_model2.py:53-62 -- the 3x3 convolutions with >= 128 input
// channels on 200 x 25 planes).  NO epilogue: the question this answers is how fast the loop alone runs against wino11_kernel's
// (wino.hip), before any inverse transform is written (profiles/r04_f4x4_page.txt ends with exactly this step).
//
// Same structure as wino11_kernel wherever the algorithm allows:
//   * a wave owns ONE ROW of the 6 x 6 transform: 6 positions = 96 accumulator registers; TWELVE waves per workgroup
//     (wave = row * 2 + tile half), three per SIMD (<= 168 registers), one workgroup per CU (156 KB of LDS: two sets of a
//     42-row pixel region x 8 channels + 36 weight positions);
//   * item = 64 tiles of 4 x 4 output pixels x 32 output channels, input channels in chunks of 8; a lane is a tile, the lane half
//     the channel quad, v_mfma_f32_32x32x2_f32 contracts channel e of both quads;
//   * region and weights by buffer-form LDS-DMA, one piece (1 KB) = one pixel row of the region: [quad 2][column residue 4][index 8]
//     16-byte slots, the index rotated per tile row so that the ds_read_b128 of consecutive tiles hit consecutive slots; pieces of
//     chunk c + 1 between the MFMA groups of chunk c, one barrier per chunk;
//   * input transform V = B^T d B for the wave's row: stage 1 over the 3-4 patch rows the row of B^T touches (streamed, 6 columns),
//     stage 2 per channel pair right in front of its MFMAs.
// The accumulators are folded into one float per lane at the end of an item (nothing else is stored): timing only.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/f4x4_loop.hip -o tools/f4x4_loop.bin && tools/f4x4_loop.bin [K N H W B]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) float lds_f;

struct Args {
    const float* A;      // padded plane [B][H + 1][W + 1][K] (+ one closing row), row 0 / column 0 of every image zero
    const float* U;      // transformed weights [K / 8][36][N / 32][1 KB]: [k-pair pair 2][lane half 2][co 32][2]
    float* out;
    int B, H, W, K, N, TH, TW, ipi, nkc, nnb, nitems;
};

constexpr int ROWS = 42;                 // pixel rows of a region: 64 tiles over >= 7 tile columns span <= 10 tile rows
constexpr int RAW_B = ROWS * 1024, U_B = 36 * 1024, SET_B = RAW_B + U_B;

// B^T of F(4x4, 3x3) (Lavin & Gray): row r of V = B^T d B combines patch rows / columns with these coefficients
__device__ constexpr float BT[6][6] = {{4.f, 0.f, -5.f, 0.f, 1.f, 0.f}, {0.f, -4.f, -4.f, 1.f, 1.f, 0.f}, {0.f, 4.f, -4.f, -1.f, 1.f, 0.f},
                                       {0.f, -2.f, -1.f, 2.f, 1.f, 0.f}, {0.f, 2.f, -1.f, -2.f, 1.f, 0.f}, {0.f, 4.f, 0.f, -5.f, 0.f, 1.f}};

#if __HIP_DEVICE_COMPILE__
struct Dma {
    unsigned voff[4];        // raw pieces wave + 12 j (per lane; out of range = zeros)
    unsigned l16;
    unsigned sbase, rowbytes, ubase_g, xibytes, chunkbytes;
    int wave;
};

template <class R>
__device__ __forceinline__ void raw_piece(const Dma& q, R ra, int lds_set, int j, int kc) {
    const int p = q.wave + 12 * j;
    if (p < ROWS)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_f*)(uintptr_t)(unsigned)(lds_set + p * 1024), 16, q.voff[j],
                                                 (int)(q.sbase + (unsigned)p * q.rowbytes + (unsigned)kc * 32u), 0, 0);
}
template <class R>
__device__ __forceinline__ void u_piece(const Dma& q, R ru, int lds_set, int j, int kc) {
    const int xi = q.wave + 12 * j;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(ru, (lds_f*)(uintptr_t)(unsigned)(lds_set + RAW_B + xi * 1024), 16, q.l16,
                                             (int)(q.ubase_g + (unsigned)kc * q.chunkbytes + (unsigned)xi * q.xibytes), 0, 0);
}

// one 8-channel chunk of transform row RR out of the set at byte offset `setoff`; seven DMA pieces of the next chunk between the groups.
// Per channel pair h (two of the lane's four channels: ds_read_b64, so that t[6] + one patch row are 24 registers beside the 96
// accumulators): stage 1 t[c] = sum_i BT[RR][i] d[i][c] over the 3-4 patch rows the row touches, stage 2 v[j] = sum_c BT[j][c] t[c]
// three positions at a time, each right in front of its MFMAs.
template <int RR, class R>
__device__ __forceinline__ void chunk(const char* __restrict__ lds, int lds0, int setoff, int nxtoff, bool pre, const Dma& q, R ra, R ru, int kcn,
                                      const unsigned (&lb)[4], unsigned ubase, floatx16 (&acc)[6]) {
    int piece = 0;
    auto dma = [&]() {
        if (pre) {
            if (piece < 3) u_piece(q, ru, lds0 + nxtoff, piece, kcn);
            else if (piece < 7) raw_piece(q, ra, lds0 + nxtoff, piece - 3, kcn);
        }
        ++piece;
    };
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        f2 t[6];
        bool first = true;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            if (BT[RR][i] == 0.f) continue;
            f2 d[6];
#pragma unroll
            for (int c = 0; c < 6; ++c)
                d[c] = *(const f2*)(lds + lb[(i >> 2) * 2 + (c >> 2)] + setoff + (i & 3) * 1024 + (c & 3) * 128 + h * 8);
            const float w = BT[RR][i];
#pragma unroll
            for (int c = 0; c < 6; ++c) t[c] = first ? w * d[c] : (w == 1.f ? t[c] + d[c] : w == -1.f ? t[c] - d[c] : w * d[c] + t[c]);
            first = false;
            __builtin_amdgcn_sched_barrier(0);
        }
        if (h == 0) dma();
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            f2 v[3], u[3];
#pragma unroll
            for (int jj = 0; jj < 3; ++jj) {
                const int j = g * 3 + jj;
                u[jj] = *(const f2*)(lds + ubase + setoff + (RR * 6 + j) * 1024 + h * 512);
                f2 sacc = {0.f, 0.f};
                bool f = true;
#pragma unroll
                for (int c = 0; c < 6; ++c) {
                    if (BT[j][c] == 0.f) continue;
                    const float w = BT[j][c];
                    sacc = f ? (w == 1.f ? t[c] : w * t[c]) : (w == 1.f ? sacc + t[c] : w == -1.f ? sacc - t[c] : w * t[c] + sacc);
                    f = false;
                }
                v[jj] = sacc;
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int jj = 0; jj < 3; ++jj) acc[g * 3 + jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[jj].x, u[jj].x, acc[g * 3 + jj], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            dma();
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int jj = 0; jj < 3; ++jj) acc[g * 3 + jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[jj].y, u[jj].y, acc[g * 3 + jj], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (!(h == 1 && g == 1)) dma();
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

template <int RR>
__device__ __forceinline__ void body(const Args& a, float* smem) {
    const char* lds = (const char*)smem;
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, lh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // = RR * 2 + wm
    const int wm = wave & 1;
    const int lds0 = (int)(unsigned)(uintptr_t)(lds_f*)smem;
    auto ra = __builtin_amdgcn_make_buffer_rsrc((void*)a.A, 0, 0x7FFFFFF0, 0x00020000);
    auto ru = __builtin_amdgcn_make_buffer_rsrc((void*)a.U, 0, 0x7FFFFFF0, 0x00020000);
    unsigned ubase = (unsigned)(RAW_B + lh * 256 + li * 8);
    asm volatile("" : "+v"(ubase));
    Dma q;
    q.wave = wave; q.l16 = (unsigned)lane * 16u;
    q.rowbytes = (unsigned)((a.W + 1) * a.K * 4);
    q.xibytes = (unsigned)a.nnb * 1024u; q.chunkbytes = 36u * (unsigned)a.nnb * 1024u;
    float sink = 0.f;
    for (int it = blockIdx.x; it < a.nitems; it += gridDim.x) {
        const int blk = it / a.nnb, nb = it - blk * a.nnb;
        const int b = blk / a.ipi, l0 = (blk - b * a.ipi) * 64;
        const int tr0 = l0 / a.TW;
        // this lane's tile (a tile past the image is clamped onto the last one: its reads stay inside the region)
        int l = l0 + wm * 32 + li;
        if (l >= a.TH * a.TW) l = a.TH * a.TW - 1;
        const int ti = l / a.TW, tj = l - ti * a.TW, m = ti - tr0;
        unsigned lb[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int rot = ((tr0 + m + (k >> 1)) * a.TW) & 7;
            lb[k] = (unsigned)((4 * (m + (k >> 1))) * 1024 + lh * 512 + ((tj + (k & 1) + rot) & 7) * 16);
            asm volatile("" : "+v"(lb[k]));
        }
        // DMA offsets of the raw pieces this wave issues: piece p = pixel row p of the region, lane = [quad][residue][slot]
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int p = wave + 12 * j;
            const int rot = ((tr0 + (p >> 2)) * a.TW) & 7;
            const int quad = lane >> 5, res = (lane >> 3) & 3, idx = ((lane & 7) - rot) & 7;
            const int col = 4 * idx + res, row = 4 * tr0 + p;
            const bool ok = col <= a.W && row <= a.H && p < ROWS;
            q.voff[j] = ok ? (unsigned)((col * a.K + quad * 4) * 4) : 0xFFFFFFF0u;
        }
        q.sbase = (unsigned)((((long)b * (a.H + 1) + 4 * tr0) * (a.W + 1)) * a.K * 4);
        q.ubase_g = (unsigned)nb * 1024u;
        // chunk 0 of this item into set 0
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 3; ++j) u_piece(q, ru, lds0, j, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) raw_piece(q, ra, lds0, j, 0);
        floatx16 acc[6];
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        int cur = 0;
        for (int kc = 0; kc < a.nkc; ++kc) {
            chunk<RR>(lds, lds0, cur * SET_B, (cur ^ 1) * SET_B, kc + 1 < a.nkc, q, ra, ru, kc + 1, lb, ubase, acc);
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            cur ^= 1;
        }
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) s += acc[i][r];
        sink += s;
    }
    a.out[(long)blockIdx.x * 768 + tid] = sink;
}

#endif

__global__ __launch_bounds__(768, 3) void f4_loop_kernel(Args a) {
#if __HIP_DEVICE_COMPILE__
    extern __shared__ __attribute__((aligned(16))) float smem[];
    switch (threadIdx.x >> 7) {
        case 0: body<0>(a, smem); break;
        case 1: body<1>(a, smem); break;
        case 2: body<2>(a, smem); break;
        case 3: body<3>(a, smem); break;
        case 4: body<4>(a, smem); break;
        default: body<5>(a, smem); break;
    }
#endif
}

int main(int argc, char** argv) {
    const int K = argc > 1 ? atoi(argv[1]) : 128, N = argc > 2 ? atoi(argv[2]) : 128, H = argc > 3 ? atoi(argv[3]) : 200,
              W = argc > 4 ? atoi(argv[4]) : 25, B = argc > 5 ? atoi(argv[5]) : 32;
    Args a;
    a.B = B; a.H = H; a.W = W; a.K = K; a.N = N;
    a.TH = (H + 3) / 4; a.TW = (W + 3) / 4;
    if (a.TW < 7 || a.TW > 8 || (K % 8) || (N % 32)) { printf("this prototype takes 7..8 tile columns (W 25..32), K %% 8 == 0, N %% 32 == 0\n"); return 1; }
    a.ipi = (a.TH * a.TW + 63) / 64; a.nkc = K / 8; a.nnb = N / 32; a.nitems = B * a.ipi * a.nnb;
    const size_t na = ((size_t)B * (H + 1) + 1) * (W + 1) * K + 4096, nu = (size_t)36 * K * N;
    float *dA, *dU, *dout;
    hipMalloc(&dA, na * 4); hipMalloc(&dU, nu * 4); hipMalloc(&dout, (size_t)1024 * 768 * 4);
    std::vector<float> h(na);
    for (size_t i = 0; i < na; ++i) h[i] = (float)((i * 2654435761u >> 8) & 0xFFFF) / 65536.f - 0.5f;
    hipMemcpy(dA, h.data(), na * 4, hipMemcpyHostToDevice);
    h.resize(nu);
    for (size_t i = 0; i < nu; ++i) h[i] = (float)((i * 40503u >> 4) & 0xFFFF) / 65536.f - 0.5f;
    hipMemcpy(dU, h.data(), nu * 4, hipMemcpyHostToDevice);
    a.A = dA; a.U = dU; a.out = dout;
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    const int ncu = pr.multiProcessorCount;
    const int grid = a.nitems < ncu ? a.nitems : ncu;
    const size_t ldsb = 2 * SET_B;
    hipFuncSetAttribute((const void*)f4_loop_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(f4_loop_kernel, dim3(grid), dim3(768), ldsb, 0, a);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
    const int iters = 20;
    hipEventRecord(e0);
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(f4_loop_kernel, dim3(grid), dim3(768), ldsb, 0, a);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us = 1e3 * ms / iters;
    const double mfma = (double)a.nitems * 12 * a.nkc * 24;                       // wave-level MFMAs
    const double pipe_us_100 = mfma * 64.0 / (4.0 * ncu) / 2400.0;                // at 2.4 GHz, all SIMDs
    const double direct_flops = 2.0 * B * H * W * 9.0 * K * N;
    printf("F(4x4,3x3) chunk loop only: K %d N %d %dx%d B %d: %d items of 64 tiles x 32 channels on %d CUs (%.2f per CU), %d chunks\n", K, N, H, W, B,
           a.nitems, ncu, (double)a.nitems / ncu, a.nkc);
    printf("  %.1f us per launch = %.1f TFLOP/s in direct-conv flops of the USEFUL pixels; matrix-pipe time at 2.4 GHz %.1f us -> share %.3f\n", us,
           direct_flops / us / 1e6, pipe_us_100, pipe_us_100 / us);
    printf("  tiles used: %d of %d per image (%.3f); columns used %d of %d\n", a.TH * a.TW, a.ipi * 64, (double)a.TH * a.TW / (a.ipi * 64), W, 4 * a.TW);
    return 0;
}

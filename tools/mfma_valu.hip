// Do VALU instructions hide under fp32 MFMAs?  (a) one wave per SIMD: NV independent v_fma between consecutive
// v_mfma_f32_32x32x2_f32; (b) two waves per SIMD, one MFMA-only and one VALU-only (even / odd workgroups of a CU share SIMDs).
// hipcc --offload-arch=gfx950 -O3 tools/mfma_valu.hip -o tools/mfma_valu.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

// the same pairing with v_mfma_f32_16x16x4_f32 (two independent accumulators = the flops of one 32x32x2)
template <int NV>
__global__ __launch_bounds__(512) void split16_loop(float* out, int iters, float a0, float b0, int mode) {
    floatx4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
    float a = a0 + threadIdx.x * 1e-6f, b = b0;
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = a + i;
    const int wv = threadIdx.x >> 6;
    const bool do_mfma = (mode == 0 || mode == 2) && wv < 4;
    const bool do_valu = (mode == 1 || mode == 2) && wv >= 4;
    if (do_mfma) {
        for (int it = 0; it < iters; ++it) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, acc1, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (do_valu) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < NV; ++i) v[i % 16] = __builtin_fmaf(v[i % 16], b, a);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = acc0[0] + acc1[3];
#pragma unroll
    for (int i = 0; i < 16; ++i) s += v[i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int NV, int MF>
__global__ __launch_bounds__(256) void mix_loop(float* out, int iters, float a0, float b0) {
    floatx16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float a = a0 + threadIdx.x * 1e-6f, b = b0;
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = a + i;
    for (int it = 0; it < iters; ++it) {
        if (MF) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i % 16] = __builtin_fmaf(v[i % 16], b, a);
        __builtin_amdgcn_sched_barrier(0);
    }
    float s = acc[0] + acc[15];
#pragma unroll
    for (int i = 0; i < 16; ++i) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// role by wave of a 512-thread workgroup: waves 0-3 MFMA-only, waves 4-7 VALU-only (NV v_fma per iteration); wave w and
// wave w + 4 share a SIMD
template <int NV>
__global__ __launch_bounds__(512) void split_loop(float* out, int iters, float a0, float b0, int mode) {
    floatx16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float a = a0 + threadIdx.x * 1e-6f, b = b0;
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = a + i;
    const int wv = threadIdx.x >> 6;
    const bool do_mfma = (mode == 0 || mode == 2) && wv < 4;
    const bool do_valu = (mode == 1 || mode == 2) && wv >= 4;
    if (do_mfma) {
        for (int it = 0; it < iters; ++it) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (do_valu) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < NV; ++i) v[i % 16] = __builtin_fmaf(v[i % 16], b, a);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = acc[0] + acc[15];
#pragma unroll
    for (int i = 0; i < 16; ++i) s += v[i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

// The issue pattern of wino8_kernel (wino.hip), eight waves = two per SIMD, every wave the same stream: per "double batch"
// NP packed adds on aligned register pairs (the input transform: 20 v_pk_add_f32 + a few selects per 16 MFMAs in the kernel),
// all of them BEFORE sixteen back-to-back v_mfma_f32_32x32x2_f32 on eight accumulators.  time(NP) / time(0) is the factor by
// which the transform's vector issue stretches the kernel's MFMA time: bench.py prices the Winograd kernels against
// 157.3 TFLOP/s x 2.25 x time(0) / time(NP).
typedef float f2 __attribute__((ext_vector_type(2)));
template <int NP>
__global__ __launch_bounds__(512) void wino_pattern(float* out, int iters, float a0, float b0) {
    floatx16 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const float a = a0 + threadIdx.x * 1e-6f, b = b0;
    f2 v[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) v[i] = f2{a + i, a - i};
    const f2 inc = f2{b, -b};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NP; ++i) v[i % 12] = v[i % 12] + (i % 3 == 0 ? inc : v[(i + 5) % 12]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i & 7] = __builtin_amdgcn_mfma_f32_32x32x2f32((i & 8) ? v[i & 7].y : v[i & 7].x, b, acc[i & 7], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][15];
#pragma unroll
    for (int i = 0; i < 12; ++i) s += v[i].x + v[i].y;
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

// Round 4: the same question for a wave that owns ONE ROW of the transform.  wino11_kernel: NM = 16 MFMAs on 4 accumulators, 16
// v_pk_add_f32, four waves per SIMD (1024 threads).  A would-be F(4x4,3x3) kernel: NM = 24 MFMAs on 6 accumulators, ~72 v_pk_fma_f32
// (B^T of F(4x4) has the constants 4, 5, 2: multiply-adds, not adds), three waves per SIMD (768 threads, 96 accumulator registers).
template <int NP, int NM, int NACC, int THREADS, bool FMA>
__global__ __launch_bounds__(THREADS) void row_pattern(float* out, int iters, float a0, float b0) {
    floatx16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const float a = a0 + threadIdx.x * 1e-6f, b = b0;
    f2 v[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) v[i] = f2{a + i, a - i};
    const f2 inc = f2{b, -b};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            if (FMA) v[i % 12] = __builtin_elementwise_fma(v[(i + 5) % 12], inc, v[i % 12]);
            else v[i % 12] = v[i % 12] + (i % 3 == 0 ? inc : v[(i + 5) % 12]);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NM; ++i) acc[i % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32((i & 8) ? v[i % 12].y : v[i % 12].x, b, acc[i % NACC], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][15];
#pragma unroll
    for (int i = 0; i < 12; ++i) s += v[i].x + v[i].y;
    out[blockIdx.x * THREADS + threadIdx.x] = s;
}

template <typename F>
float timeit(F launch) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 3; ++r) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / 3;
}

template <int NV>
void run_mix(float* out, int iters) {
    float t1 = timeit([&] { hipLaunchKernelGGL((mix_loop<NV, 1>), dim3(256), dim3(256), 0, 0, out, iters, 1.0f, 0.5f); });
    float t0 = timeit([&] { hipLaunchKernelGGL((mix_loop<NV, 0>), dim3(256), dim3(256), 0, 0, out, iters, 1.0f, 0.5f); });
    printf("one wave/SIMD: 1 MFMA + %2d v_fma per iteration: %.1f cyc/iter at 2.4 GHz   (v_fma alone %.1f)\n", NV, t1 * 1e-3 * 2.4e9 / iters, t0 * 1e-3 * 2.4e9 / iters);
}

template <int NV>
void run_split(float* out, int iters) {
    float tm = timeit([&] { hipLaunchKernelGGL((split_loop<NV>), dim3(256), dim3(512), 0, 0, out, iters, 1.0f, 0.5f, 0); });
    float tv = timeit([&] { hipLaunchKernelGGL((split_loop<NV>), dim3(256), dim3(512), 0, 0, out, iters, 1.0f, 0.5f, 1); });
    float tb = timeit([&] { hipLaunchKernelGGL((split_loop<NV>), dim3(256), dim3(512), 0, 0, out, iters, 1.0f, 0.5f, 2); });
    printf("two waves/SIMD, NV %2d: MFMA-only wave %.3f ms, VALU-only wave %.3f ms, both side by side %.3f ms\n", NV, tm, tv, tb);
}

template <int NV>
void run_split16(float* out, int iters) {
    float tm = timeit([&] { hipLaunchKernelGGL((split16_loop<NV>), dim3(256), dim3(512), 0, 0, out, iters, 1.0f, 0.5f, 0); });
    float tv = timeit([&] { hipLaunchKernelGGL((split16_loop<NV>), dim3(256), dim3(512), 0, 0, out, iters, 1.0f, 0.5f, 1); });
    float tb = timeit([&] { hipLaunchKernelGGL((split16_loop<NV>), dim3(256), dim3(512), 0, 0, out, iters, 1.0f, 0.5f, 2); });
    printf("16x16x4 x2, two waves/SIMD, NV %2d: MFMA-only wave %.3f ms, VALU-only wave %.3f ms, both side by side %.3f ms\n", NV, tm, tv, tb);
}

template <int NP>
float run_wino(float* out, int iters, float t0) {
    const float t = timeit([&] { hipLaunchKernelGGL((wino_pattern<NP>), dim3(256), dim3(512), 0, 0, out, iters, 1.0f, 0.5f); });
    printf("wino pattern, two waves/SIMD: %2d v_pk_add_f32 + 16 MFMA per iteration: %.3f ms  = %.1f cyc per 16 MFMAs and wave pair at 2.4 GHz;  time(0)/time(NP) = %.3f\n",
           NP, t, t * 1e-3 * 2.4e9 / iters, t0 > 0 ? t0 / t : 1.0);
    return t;
}

template <int NP, int NM, int NACC, int THREADS, bool FMA>
float run_row(float* out, int iters, float t0, const char* what) {
    const float t = timeit([&] { hipLaunchKernelGGL((row_pattern<NP, NM, NACC, THREADS, FMA>), dim3(256), dim3(THREADS), 0, 0, out, iters, 1.0f, 0.5f); });
    printf("%s, %d waves/SIMD: %2d %s + %d MFMA per iteration: %.3f ms;  time(0)/time(NP) = %.3f\n", what, THREADS / 256, NP,
           FMA ? "v_pk_fma_f32" : "v_pk_add_f32", NM, t, t0 > 0 ? t0 / t : 1.0);
    return t;
}

int main(int argc, char** argv) {
    float* out; hipMalloc(&out, 1024 * 1024 * sizeof(float));
    if (argc > 1 && argv[1][0] == 'r') {
        const int it = 30000;
        float t0 = run_row<0, 16, 4, 1024, false>(out, it, 0.f, "wino11 row pattern");
        run_row<16, 16, 4, 1024, false>(out, it, t0, "wino11 row pattern"); run_row<24, 16, 4, 1024, false>(out, it, t0, "wino11 row pattern");
        t0 = run_row<0, 24, 6, 768, true>(out, it, 0.f, "F(4x4) row pattern");
        run_row<48, 24, 6, 768, true>(out, it, t0, "F(4x4) row pattern"); run_row<72, 24, 6, 768, true>(out, it, t0, "F(4x4) row pattern");
        run_row<96, 24, 6, 768, true>(out, it, t0, "F(4x4) row pattern");
        return 0;
    }
    if (argc > 1 && argv[1][0] == 'w') {
        const int it = 50000;
        const float t0 = run_wino<0>(out, it, 0.f);
        run_wino<8>(out, it, t0); run_wino<16>(out, it, t0); run_wino<20>(out, it, t0); run_wino<23>(out, it, t0); run_wino<24>(out, it, t0);
        run_wino<32>(out, it, t0);
        return 0;
    }
    const int iters = 200000;
    run_mix<0>(out, iters); run_mix<4>(out, iters); run_mix<8>(out, iters); run_mix<12>(out, iters); run_mix<16>(out, iters); run_mix<24>(out, iters); run_mix<32>(out, iters);
    run_split<8>(out, iters); run_split<16>(out, iters); run_split<32>(out, iters);
    run_split16<8>(out, iters); run_split16<16>(out, iters); run_split16<32>(out, iters);
    return 0;
}

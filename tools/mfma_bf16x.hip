// Feasibility probe for a split-bf16 contraction path (DESIGN "what comes next"): an fp32 operand is written as a sum of
// bf16 pieces (hi + mid + lo, 8 mantissa bits each) and the product a*b as a few v_mfma_f32_32x32x16_bf16 products with
// fp32 accumulation.  Measures (1) the register-only rate of 1 / 3 / 6 bf16 MFMAs per fp32-equivalent step against the
// native v_mfma_f32_32x32x2_f32, and (2) the error of the 3- and 6-product schemes against float64 on random data.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_bf16x.hip -o /tmp/mfma_bf16x
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// NP bf16 MFMAs (K = 16 each) per "step"; a step of the fp32-equivalent GEMM covers K = 16
template <int NP, int NACC>
__global__ __launch_bounds__(256) void bf16_loop(float* out, int iters, float a0) {
    floatx16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 a, b;
#pragma unroll
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(a0 + threadIdx.x * 1e-3f + j); b[j] = (__bf16)(0.5f + j); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i)
#pragma unroll
            for (int p = 0; p < NP; ++p) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][15];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC>
__global__ __launch_bounds__(256) void f32_loop(float* out, int iters, float a0) {
    floatx16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = a0 + threadIdx.x * 1e-6f, b = 0.5f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i)
#pragma unroll
            for (int p = 0; p < 8; ++p) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);   // 8 x K=2 = K 16
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][15];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <typename F>
double time_ms(F launch) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / 5;
}

// numerical probe on the host: split and multiply in float arithmetic that mimics bf16 products with fp32 accumulation
static float bf16r(float x) {            // round-to-nearest-even to bf16
    unsigned u; memcpy(&u, &x, 4);
    unsigned r = u + 0x7FFFu + ((u >> 16) & 1u);
    r &= 0xFFFF0000u;
    float y; memcpy(&y, &r, 4);
    return y;
}

int main() {
    const int nblk = 256 * 2, iters = 4000;
    float* out; hipMalloc(&out, nblk * 256 * sizeof(float));
    const double steps = (double)nblk * 4 * iters * 4;          // waves * iters * NACC(4): fp32-equivalent K=16 steps of a 32x32 tile
    const double flop_step = 2.0 * 32 * 32 * 16;
    double t;
    t = time_ms([&] { hipLaunchKernelGGL((f32_loop<4>), dim3(nblk), dim3(256), 0, 0, out, iters, 1.0f); });
    printf("native fp32 MFMA (8 x 32x32x2)      : %7.3f ms  %7.1f fp32-equivalent TFLOP/s\n", t, steps * flop_step / t / 1e9);
    t = time_ms([&] { hipLaunchKernelGGL((bf16_loop<1, 4>), dim3(nblk), dim3(256), 0, 0, out, iters, 1.0f); });
    printf("1 bf16 MFMA per step (plain bf16)   : %7.3f ms  %7.1f\n", t, steps * flop_step / t / 1e9);
    t = time_ms([&] { hipLaunchKernelGGL((bf16_loop<3, 4>), dim3(nblk), dim3(256), 0, 0, out, iters, 1.0f); });
    printf("3 bf16 MFMAs per step (2-piece split): %7.3f ms  %7.1f\n", t, steps * flop_step / t / 1e9);
    t = time_ms([&] { hipLaunchKernelGGL((bf16_loop<6, 4>), dim3(nblk), dim3(256), 0, 0, out, iters, 1.0f); });
    printf("6 bf16 MFMAs per step (3-piece split): %7.3f ms  %7.1f\n", t, steps * flop_step / t / 1e9);
    hipFree(out);

    // accuracy: dot products of length 1152 (a 3x3 x 128-channel conv) of N(0,1) data
    srand(1);
    const int K = 1152, trials = 2000;
    double e32 = 0, e3 = 0, e6 = 0, scale = 0;
    for (int t2 = 0; t2 < trials; ++t2) {
        double ref = 0; float s32 = 0, s3 = 0, s6 = 0;
        for (int k = 0; k < K; ++k) {
            float a = (float)rand() / RAND_MAX * 2 - 1, b = (float)rand() / RAND_MAX * 2 - 1;
            ref += (double)a * b;
            s32 = fmaf(a, b, s32);
            float ah = bf16r(a), am = bf16r(a - ah), al = bf16r(a - ah - am);
            float bh = bf16r(b), bm = bf16r(b - bh), bl = bf16r(b - bh - bm);
            s3 += ah * bh + (ah * bm + am * bh);
            s6 += ah * bh + (ah * bm + am * bh) + (am * bm + ah * bl + al * bh);
        }
        e32 += fabs(s32 - ref); e3 += fabs(s3 - ref); e6 += fabs(s6 - ref); scale += fabs(ref);
    }
    printf("mean |error| of a K=%d dot product (mean |value| %.3f): fp32 fma %.3e, 3-product %.3e, 6-product %.3e\n",
           K, scale / trials, e32 / trials, e3 / trials, e6 / trials);
    return 0;
}

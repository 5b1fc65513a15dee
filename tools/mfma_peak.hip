// Measures the attainable fp32-MFMA rate on this device: register-only v_mfma_f32_32x32x2_f32 loops,
// W waves per SIMD, A independent accumulators per wave.  hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float floatx16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters, float a0, float b0) {
    floatx16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = a0 + threadIdx.x * 1e-6f, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][15];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC>
void run(int blocks_per_cu, int iters) {
    int nblk = 256 * blocks_per_cu;
    float* out; hipMalloc(&out, nblk * 256 * sizeof(float));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(mfma_loop<NACC>, dim3(nblk), dim3(256), 0, 0, out, iters, 1.0f, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(mfma_loop<NACC>, dim3(nblk), dim3(256), 0, 0, out, iters, 1.0f, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    double flops = (double)nblk * 4 * iters * NACC * 4096.0;
    printf("acc %d  waves/SIMD %d  iters %d : %.3f ms  %.1f TFLOP/s\n", NACC, blocks_per_cu, iters, ms, flops / ms / 1e9);
    hipFree(out);
}

int main() {
    run<1>(1, 20000); run<2>(1, 10000); run<4>(1, 5000); run<4>(2, 5000); run<2>(4, 5000); run<8>(1, 5000); run<9>(2, 2000);
    // long run to expose sustained clocks
    run<4>(2, 40000);
    return 0;
}

// What does feeding v_mfma_f32_32x32x2_f32 from LDS cost?  Mimics the tap-GEMM wave tile (TM=2, TN=1):
// per 8 MFMAs two ds_read_b128 (A) and four ds_read_b32 (B), optional barrier every NB groups.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float floatx16 __attribute__((ext_vector_type(16)));

template <int BAR>
__global__ __launch_bounds__(256) void k(float* out, int iters, int lds_floats) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    for (int i = threadIdx.x; i < lds_floats; i += 256) sm[i] = (float)(i & 15) * 0.01f;
    __syncthreads();
    const int lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5, wave = threadIdx.x >> 6;
    floatx16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    const float* abase = sm + (wave * 64 + li) * 36 + 4 * lh;          // A tile rows, pitch 36
    const float* wbase = sm + 184 * 36 + (4 * lh) * 64 + li;            // W tile [32][64]
    for (int it = 0; it < iters; ++it) {
        const int off = (it & 7) * 36;                                   // vary the row a little (tap offsets)
#pragma unroll
        for (int gk = 0; gk < 4; ++gk) {
            const float4 a0 = *(const float4*)(abase + off + gk * 8);
            const float4 a1 = *(const float4*)(abase + off + 32 * 36 + gk * 8);
            const float b0 = wbase[(gk * 8 + 0) * 64], b1 = wbase[(gk * 8 + 1) * 64], b2 = wbase[(gk * 8 + 2) * 64], b3 = wbase[(gk * 8 + 3) * 64];
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, b0, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, b0, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, b1, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, b1, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, b2, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, b2, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, b3, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, b3, acc1, 0, 0, 0);
        }
        if (BAR == 1) { __syncthreads(); }
        if (BAR == 2) { __syncthreads(); __syncthreads(); }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc0[0] + acc1[5];
}

template <int BAR>
void run(int bpc, int iters) {
    int nblk = 256 * bpc;
    float* out; (void)hipMalloc(&out, nblk * 256 * sizeof(float));
    const int lds_floats = 184 * 36 + 32 * 64 + 64;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<BAR>, dim3(nblk), dim3(256), lds_floats * 4, 0, out, iters, lds_floats);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<BAR>, dim3(nblk), dim3(256), lds_floats * 4, 0, out, iters, lds_floats);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    double flops = (double)nblk * 4 * iters * 32 * 4096.0;
    printf("barriers/32mfma %d  blocks/CU %d : %.3f ms  %.1f TFLOP/s\n", BAR, bpc, ms, flops / ms / 1e9);
    (void)hipFree(out);
}

int main() {
    for (int b = 1; b <= 4; ++b) run<0>(b, 2000);
    for (int b = 1; b <= 4; ++b) run<1>(b, 2000);
    for (int b = 1; b <= 4; ++b) run<2>(b, 2000);
    return 0;
}

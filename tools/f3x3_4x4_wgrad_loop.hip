// Prototype of the STAGE LOOP of a Winograd F(3x3,4x4) weight-gradient kernel in fp32 on the MFMA pipe (the page:
// profiles/r06_f3x3_4x4_page.txt; reference layers: lm_and_am/model/acoustic_model.py:42-46, acoustic_model2.py:47-62), timed against the stage
// loop of the product's F(3x3,2x2) kernel (wino_wgrad4_kernel<64>, csrc/wino_wgrad.hip) on the same planes.  No reduce pass, no partial store
// worth the name (one store per accumulator so that nothing is optimised away); the transforms are the real ones (B^T of F(4,3) on the input
// patch, A of F(4,3) on the 4 x 4 gradient tile), the ragged last tile of a block is multiplied like a whole one (its MFMAs are the ones a
// product kernel would issue with zeros in one lane half).
//
// What fits (the page, section b): 36 positions x 32 input x 64 output channels = 288 KB of accumulators -> twelve waves, wave = transform
// row x output-channel half, six 32 x 32 accumulators each, three waves per SIMD; the staged region of wino_wgrad4_kernel<32> (6 input rows x
// 32 pixels of 128 B, 4 gradient rows x 28 pixels of 256 B, two buffer sets) is ONE row of <= 7 tiles.
//
// build + run (GPU box):  hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include tools/f3x3_4x4_wgrad_loop.hip -o /tmp/f34 && /tmp/f34
#include "../asr_dfcnn_transformer_amd/csrc/wino_wgrad.hip"
#include <stdio.h>
#include <stdlib.h>
#include <vector>

void asr_set_error(const char*, hipError_t) {}
void asr_set_last_kernel(const char*) {}

namespace {

#if __HIP_DEVICE_COMPILE__
template <int RR>
__device__ __forceinline__ void ww6_body(const WwArgs& a, float* smem, float* sink) {
    typedef WwCfg<32, 64> C;
    constexpr int PXB = 128, ZXB = 256;
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, lh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);           // = RR * 2 + wn
    const int wn = wave & 1;
    const int wg = blockIdx.x;
    const int sl_lo = wg & 7, rest = wg >> 3;
    const int bp = rest % a.nbp, sl = (rest / a.nbp) * 8 + sl_lo;
    if (sl >= a.nsl) return;
    const int cib = bp / a.ncob, cob = bp - cib * a.ncob;
    const int cin0 = cib * 32, co0 = cob * 64;
    auto rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.A, 0, 0x7FFFFFF0, 0x00020000);
    auto rz = __builtin_amdgcn_make_buffer_rsrc((void*)a.Z, 0, 0x7FFFFFF0, 0x00020000);
    const int pxx = lane / 8, pxz = lane >> 4;
    const unsigned vox = (unsigned)((pxx * a.lda + (lane % 8) * 4) * 4);
    const unsigned voz = (unsigned)((pxz * a.ldz + (lane & 15) * 4) * 4);
    constexpr int NJ = (C::NXP + C::NZP + 11) / 12;                      // piece rounds per stage (pieces wave + 12 j)

    floatx16 acc[6];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    int g = sl;
    WwStage s = ww_stage<2>(a, g);
#pragma unroll
    for (int j = 0; j < NJ; ++j) { const int p = wave + 12 * j; ww_piece<32, 64>(a, s, rx, rz, smem, p & 7, p >> 3, cin0, co0, vox, voz, pxx, pxz); }
    int cur = 0;
    for (; g < a.nstages; g += a.nsl) {
        const int gn = g + a.nsl;
        const bool more = gn < a.nstages;
        WwStage sn = s;
        if (more) sn = ww_stage<2>(a, gn);
        ww_barrier_dma();
        const char* xs = (const char*)(smem + cur * C::SETF);
        const char* zs = xs + C::XF * 4;
        float* nxt = smem + (cur ^ 1) * C::SETF;
        int jn = 0;
        auto piece = [&]() {
            if (more && jn < NJ) { const int p = wave + 12 * jn; ww_piece<32, 64>(a, sn, rx, rz, nxt, p & 7, p >> 3, cin0, co0, vox, voz, pxx, pxz); }
            ++jn;
        };
        const int w4 = (s.w + 1) >> 1;                 // 4 x 4 tiles of this block row
        const int nsteps = (w4 + 3) >> 2;              // four tiles per step: lane half = tile of a pair, packed value = pair
        for (int st = 0; st < nsteps; ++st) {
            // this lane's tile of pair A: 4 st + lh; pair B two tiles (8 pixels) further
            const unsigned xe = (unsigned)(uintptr_t)(const ww_lds_c*)(xs + ((4 * (4 * st + lh)) * 32 + li) * 4);
            const unsigned xo = xe + PXB;
            const unsigned ze = (unsigned)(uintptr_t)(const ww_lds_c*)(zs + ((4 * (4 * st + lh)) * 64 + wn * 32 + li) * 4);
            ww_f2 t[6], v[6], yr[4], z[6];
#define LDX(dst, r, c) { constexpr int pix_ = (r) * C::XP + (c);                                                               \
        if ((pix_ & 1) == 0) asm volatile("ds_read2st64_b32 %0, %1 offset0:%2 offset1:%3" : "=&v"(dst) : "v"(xe), "i"(pix_ * PXB >> 8), "i"((pix_ + 8) * PXB >> 8)); \
        else asm volatile("ds_read2st64_b32 %0, %1 offset0:%2 offset1:%3" : "=&v"(dst) : "v"(xo), "i"((pix_ - 1) * PXB >> 8), "i"((pix_ + 7) * PXB >> 8)); }
#define LDZ(dst, r, c) { constexpr int pix_ = (r) * C::ZP + (c);                                                               \
        asm volatile("ds_read2st64_b32 %0, %1 offset0:%2 offset1:%3" : "=&v"(dst) : "v"(ze), "i"(pix_ * ZXB >> 8), "i"((pix_ + 8) * ZXB >> 8)); }
            // input: row RR of B^T d (3-4 patch rows per column), then the 6-point transform along the columns
#define XCOL(c) { ww_f2 d0, d1, d2, d3;                                                                                          \
                if constexpr (RR == 0) { LDX(d0, 0, c) LDX(d1, 2, c) LDX(d2, 4, c) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); t[c] = 4.f * d0 - 5.f * d1 + d2; } \
                else if constexpr (RR == 5) { LDX(d0, 1, c) LDX(d1, 3, c) LDX(d2, 5, c) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); t[c] = 4.f * d0 - 5.f * d1 + d2; } \
                else { LDX(d0, 1, c) LDX(d1, 2, c) LDX(d2, 3, c) LDX(d3, 4, c) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       \
                    if constexpr (RR == 1) t[c] = (d3 - 4.f * d1) + (d2 - 4.f * d0);                                                    \
                    else if constexpr (RR == 2) t[c] = (d3 - 4.f * d1) - (d2 - 4.f * d0);                                               \
                    else if constexpr (RR == 3) t[c] = (d3 - d1) + 2.f * (d2 - d0);                                                     \
                    else t[c] = (d3 - d1) - 2.f * (d2 - d0); } }
            XCOL(0) XCOL(1) XCOL(2) XCOL(3) XCOL(4) XCOL(5)
#undef XCOL
            v[0] = 4.f * t[0] - 5.f * t[2] + t[4];
            v[1] = (t[4] - 4.f * t[2]) + (t[3] - 4.f * t[1]);
            v[2] = (t[4] - 4.f * t[2]) - (t[3] - 4.f * t[1]);
            v[3] = (t[4] - t[2]) + 2.f * (t[3] - t[1]);
            v[4] = (t[4] - t[2]) - 2.f * (t[3] - t[1]);
            v[5] = 4.f * t[1] - 5.f * t[3] + t[5];
            // gradient: row RR of G' y (G' = A of F(4,3): [1 0 0 0; 1 1 1 1; 1 -1 1 -1; 1 2 4 8; 1 -2 4 -8; 0 0 0 1]), then along the columns
#define ZCOL(c) { ww_f2 y0, y1, y2, y3;                                                                                          \
                if constexpr (RR == 0) { LDZ(y0, 0, c) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); yr[c] = y0; }           \
                else if constexpr (RR == 5) { LDZ(y0, 3, c) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); yr[c] = y0; }      \
                else { LDZ(y0, 0, c) LDZ(y1, 1, c) LDZ(y2, 2, c) LDZ(y3, 3, c) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  \
                    if constexpr (RR == 1) yr[c] = (y0 + y2) + (y1 + y3);                                                         \
                    else if constexpr (RR == 2) yr[c] = (y0 + y2) - (y1 + y3);                                                    \
                    else if constexpr (RR == 3) yr[c] = (y0 + 4.f * y2) + (2.f * y1 + 8.f * y3);                                  \
                    else yr[c] = (y0 + 4.f * y2) - (2.f * y1 + 8.f * y3); } }
            ZCOL(0) ZCOL(1) ZCOL(2) ZCOL(3)
#undef ZCOL
            z[0] = yr[0];
            z[1] = (yr[0] + yr[2]) + (yr[1] + yr[3]);
            z[2] = (yr[0] + yr[2]) - (yr[1] + yr[3]);
            z[3] = (yr[0] + 4.f * yr[2]) + (2.f * yr[1] + 8.f * yr[3]);
            z[4] = (yr[0] + 4.f * yr[2]) - (2.f * yr[1] + 8.f * yr[3]);
            z[5] = yr[3];
#undef LDX
#undef LDZ
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 6; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[i].x, z[i].x, acc[i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            piece(); piece();
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 6; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[i].y, z[i].y, acc[i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            piece(); piece();
            __builtin_amdgcn_sched_barrier(0);
        }
        while (more && jn < NJ) piece();
        cur ^= 1;
        s = sn;
    }
    // one value per accumulator register leaves the kernel (nothing may be optimised away)
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) sum += acc[i][r];
    sink[(size_t)blockIdx.x * 768 + tid] = sum;
}
#endif

__global__ __launch_bounds__(768) void ww6_loop_kernel(WwArgs a, float* sink) {
#if __HIP_DEVICE_COMPILE__
    extern __shared__ __attribute__((aligned(16))) float smem[];
    switch (threadIdx.x >> 7) {
        case 0: ww6_body<0>(a, smem, sink); break;
        case 1: ww6_body<1>(a, smem, sink); break;
        case 2: ww6_body<2>(a, smem, sink); break;
        case 3: ww6_body<3>(a, smem, sink); break;
        case 4: ww6_body<4>(a, smem, sink); break;
        default: ww6_body<5>(a, smem, sink); break;
    }
#endif
}

WwArgs make_args(const asr_gemm_desc& d, const WwPlan& p, const float* A, const float* Z, float* part, int ldz) {
    WwArgs a;
    a.A = A; a.Z = Z; a.part = part;
    a.K = d.K; a.N = d.N; a.lda = d.lda; a.ldz = ldz; a.B = d.B; a.H = d.H; a.Wd = d.W; a.WP = d.W + 1; a.HPWP = (d.H + 1) * (d.W + 1);
    a.SR = p.SR; a.ncb = p.ncb;
    for (int c = 0; c < 8; ++c) { a.cb_tj0[c] = c < p.ncb ? p.cbt[c] : 0; a.cb_w[c] = c < p.ncb ? p.cbw[c] : 0; }
    a.nstages = p.nstages; a.nsl = p.nsl; a.nbp = p.nbp; a.ncob = p.ncob;
    a.inv_per = 1.0f / (float)(p.SR * p.ncb); a.inv_ncb = 1.0f / (float)p.ncb;
    a.cb_base = ((d.W + 1) / 2) / p.ncb; a.cb_rem = ((d.W + 1) / 2) % p.ncb;
    return a;
}

}  // namespace

int main() {
    struct Shape { const char* name; int K, N, H, W, B; } shapes[] = {
        {"h4  128 -> 128 @ 200 x 25", 128, 128, 200, 25, 32}, {"h5a 128 -> 256 @ 200 x 25", 128, 256, 200, 25, 32}, {"h3   64 -> 128 @ 400 x 50", 64, 128, 400, 50, 32}};
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    const int ncu = pr.multiProcessorCount;
    printf("F(3x3,4x4) weight-gradient STAGE LOOP against wino_wgrad4_kernel<64>'s launch (no reduce pass either side), %s, %d CUs\n", pr.gcnArchName, ncu);
    for (const Shape& sh : shapes) {
        const long M = (long)sh.B * (sh.H + 1) * (sh.W + 1);
        const long guard = sh.W + 3;
        float *x, *dz, *part, *sink;
        hipMalloc(&x, (M + 2 * guard) * sh.K * 4); hipMalloc(&dz, (M + 2 * guard) * sh.N * 4);
        std::vector<float> hx((M + 2 * guard) * sh.K), hz((M + 2 * guard) * sh.N);
        unsigned seed = 12345;
        auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return ((seed >> 9) & 0xFFFF) / 65536.0f - 0.5f; };
        for (auto& v : hx) v = rnd();
        for (auto& v : hz) v = rnd() * 1e-3f;
        hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dz, hz.data(), hz.size() * 4, hipMemcpyHostToDevice);
        asr_gemm_desc d = {};
        d.M = (int)M; d.K = sh.K; d.N = sh.N; d.lda = sh.K; d.ldw = sh.N; d.ntaps = 9; d.wmode = 0; d.B = sh.B; d.H = sh.H; d.W = sh.W;
        // the product kernel's plan and launch (stage loop + one partial store; its reduce kernel is NOT launched)
        const WwPlan p = ww_plan(&d, sh.N);
        if (!p.ok) { printf("%s: plan refused\n", sh.name); continue; }
        hipMalloc(&part, p.ws); hipMalloc(&sink, (size_t)8192 * 768 * 4);
        WwArgs a4 = make_args(d, p, x + guard * sh.K, dz + guard * sh.N, part, sh.N);
        const int grid4 = asr_cdiv(p.nsl, 8) * 8 * p.nbp;
        const size_t lds4 = (size_t)2 * WwCfg<64, 64>::SETF * sizeof(float);
        auto k4 = wino_wgrad4_kernel<64>;
        hipFuncSetAttribute((const void*)k4, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds4);
        // the prototype: 32 x 64 blocks -> twice the block pairs, half the slices; stages of ONE 4 x 4-tile row (the <32, 64> region)
        WwPlan p6 = p;
        p6.cinb = 32; p6.nbp = (sh.K / 32) * (sh.N / 64); p6.nsl = ncu / p6.nbp; if (p6.nsl < 1) p6.nsl = 1;
        if (p6.nsl > p6.nstages / 8) p6.nsl = p6.nstages / 8;
        WwArgs a6 = make_args(d, p6, x + guard * sh.K, dz + guard * sh.N, part, sh.N);
        const int grid6 = asr_cdiv(p6.nsl, 8) * 8 * p6.nbp;
        const size_t lds6 = (size_t)2 * WwCfg<32, 64>::SETF * sizeof(float);
        auto k6 = ww6_loop_kernel;
        hipFuncSetAttribute((const void*)k6, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds6);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        auto timeit = [&](auto fn) {
            for (int i = 0; i < 3; ++i) fn();
            hipDeviceSynchronize();
            hipEventRecord(e0);
            for (int i = 0; i < 20; ++i) fn();
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            return ms / 20 * 1e3f;
        };
        const float t4 = timeit([&]() { hipLaunchKernelGGL(k4, dim3(grid4), dim3(1024), lds4, 0, a4); });
        const float t6 = timeit([&]() { hipLaunchKernelGGL(k6, dim3(grid6), dim3(768), lds6, 0, a6, sink); });
        hipError_t e = hipGetLastError();
        const double gf = 2.0 * sh.B * sh.H * sh.W * 9.0 * sh.K * sh.N / 1e9;
        printf("%s, B %d: F(3x3,2x2) wino_wgrad4_kernel<64> %7.1f us (%d workgroups, %5.1f TFLOP/s direct-conv)   F(3x3,4x4) loop %7.1f us (%d workgroups of 12 waves)   ratio %.2fx   %s\n",
               sh.name, sh.B, t4, grid4, gf / t4 * 1e3, t6, grid6, t4 / t6, e == hipSuccess ? "" : hipGetErrorString(e));
        hipFree(x); hipFree(dz); hipFree(part); hipFree(sink);
    }
    return 0;
}

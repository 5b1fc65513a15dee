// Shared by the contraction kernels (tap_gemm.hip, wino.hip): the argument block, the tap geometry and the epilogues
// (bias / ReLU / frozen-BN affine stores, and the gated form that applies a cell's backward prologue).
#pragma once
#include "asr_common.h"
#include "reduce.h"

namespace {

constexpr size_t kEpilogueLds = 4 * 32 * 33 * sizeof(float);   // per-wave transpose scratch of the epilogue

typedef float ep_nt_f4 __attribute__((ext_vector_type(4)));

struct TapGemmArgs {
    const float* A; const float* W; const float* bias; const float* scale; const float* shift;
    float* out_a; float* out_y;
    int M, K, N, lda, ldw, ldo_a, ldo_y;
    int H, Wd, WP, HPWP;
    int halo;
    long rmin, rmax;
    int relu, accumulate, y_unpadded;
    int ntm, ntn;
    // fused backward prologue of the cell whose gradient this data-gradient completes (asr_tap_gemm_gated): instead of
    // writing dL/dy of that cell, the epilogue routes it through the cell's pool / BN / ReLU backward and writes dZ
    int gate_mode;       // 0 off, 1 no pool, 2 average pool 2x2, 3 max pool 2x2, 4 compact max pool (wino11_kernel),
                         // 5 no pool, DENSE layout (asr_tap_gemm_gated_dense): GEMM row = (image, pixel row), column = pixel column * gate_C + channel
    int gate_H, gate_W;  // the gated cell's pre-pool plane (= H, Wd here for mode 1, 2H x 2W for modes 2, 3)
    const float* gate_a; // its post-ReLU pre-BN activations, padded plane [B][gate_H+1][gate_W+1][N]
    float* gate_dz;      // out: dL/d(conv + bias) of that cell, same layout as gate_a (scale / shift = its BN affine)
    float* gate_part;    // out: [rows][3][N] per-(tile row, wave row) sums of dscale, dshift, dbias
    int* gate_rows;      // host out: rows of gate_part the launched configuration writes
    // split-K (asr_tap_gemm_splitk; 1-tap forward only): workgroup row blockIdx.y contracts K / ksplit channels and writes its raw
    // partial sums to split_out + y * M * N; a second pass adds them and applies the epilogue
    int ksplit = 1;
    float* split_out = nullptr;
    int nt_store = 0;    // non-temporal stores in the plain epilogue (the Winograd launches set it)
    // (gate mode 5 keeps the gated cell's channel count in `halo`, which a 1-tap GEMM does not use: the block stays what it was in round 4,
    //  so that no kernel's argument loads change -- DESIGN.md section 4 item 26)
};

// Row offset of tap `tap` in the flattened padded plane, and the tap of the weight tensor it multiplies.
//   9 taps: 3x3 SAME window, offsets (dh-1)*WP + (dw-1); the data-gradient walks the same offsets with the
//           mirrored weight tap (offset(8-t) = -offset(t)).
//   4 taps: forward-looking 2x2 window {0,1}x{0,1} -- the stride-2 3x3 conv of the end2end pre-net on a
//           phase-split plane (prenet.hip); its data-gradient looks backwards with the same weight tap.
template <int NTAPS, int WMODE>
__device__ __forceinline__ int tap_row_offset(int tap, int WP) {
    if (NTAPS == 9) return ((tap / 3) - 1) * WP + (tap % 3) - 1;
    if (NTAPS == 4) { const int o = (tap >> 1) * WP + (tap & 1); return WMODE == 0 ? o : -o; }
    return 0;
}
template <int NTAPS, int WMODE>
__device__ __forceinline__ int tap_weight_index(int tap) { return (NTAPS == 9 && WMODE == 1) ? NTAPS - 1 - tap : tap; }

// Epilogue shared by the three generations.  The 32x32 MFMA result has the output channel on the lane and
// the pixel row in the register, which makes the natural store 4 bytes per lane (two 128-byte rows per wave
// instruction, 32 instructions per tile and tensor).  Each wave instead transposes its tile through a private
// 32x33 LDS scratch and stores float4 rows: 4x fewer, 16-byte store instructions (+8 % on the conv kernels).
// A use of a just-loaded channel constant on EVERY path: a tile whose rows are all outside the plane never reads them, hipcc's
// wait-count pass then carries "load possibly in flight" to the top of a persistent kernel's next item, where the first write to the
// same register costs an s_waitcnt vmcnt(0) -- which on this ISA also waits for every store of the epilogue.
__device__ __forceinline__ void ep_touch(const float4& v) { asm volatile("" :: "v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w)); }

__device__ __forceinline__ float gate_pick(float y0, float y1, float y2, float y3, int k, float dp) {
    int arg = 0; float m = y0;                  // first maximum in row-major window order (TF's max-pool gradient)
    if (y1 > m) { m = y1; arg = 1; }
    if (y2 > m) { m = y2; arg = 2; }
    if (y3 > m) { m = y3; arg = 3; }
    return arg == k ? dp : 0.f;
}

// Gated epilogue (g.gate_mode != 0): the tile holds dL/dy of the cell in front (its own pixels: pooled resolution for
// modes 2, 3).  Per output pixel and channel quad: g_k = the gradient routed to pre-pool position k (the value itself /
// a quarter of it to each of four / all of it to the first maximum of scale * a + shift), dZ_k = g_k * scale where
// a_k > 0, plus the three per-channel sums of asr_cell_bwd_pre -- whose arithmetic this restates -- reduced over the
// wave's rows in a fixed order and written as one partial row per (tile row, wave row).
// DENSE (gate mode 5, asr_tap_gemm_gated_dense) is a template parameter -- only gemm1_dense_gate_kernel instantiates it --, so that the
// code of every other kernel that carries this epilogue is what it was before the mode existed.
template <int TM, int TN, bool DENSE = false>
__device__ __forceinline__ void tap_epilogue_gated(const TapGemmArgs& g, const floatx16 (&acc)[TM][TN], float* scratch,
                                                   const int* rowa, int* rowf, int row0, int col0, int lane, int part_row) {
    const int li = lane & 31, lh = lane >> 5;
    const int c4 = lane & 7, rsub = lane >> 3;
    constexpr bool dense = DENSE;                 // rowa[m] = the plane pixel of GEMM row m at pixel column 0 (the kernel's row table)
    const int C = dense ? g.halo : g.N;
    const int WPf = g.gate_W + 1;
    if (g.gate_mode >= 2 && !dense) {
        // full-resolution pixel of window position 0 for each of this wave's rows (wave-private slice of the table)
        for (int r = lane; r < TM * 32; r += 64) {
            const int m = row0 + r;
            const int ra = rowa[m];
            int pf = -1;
            if (ra >= 0) {
                const int b = ra / g.HPWP;
                const int rr = ra - b * g.HPWP;
                const int hq = rr / g.WP, wq = rr - hq * g.WP;
                pf = (b * (g.gate_H + 1) + 2 * hq - 1) * WPf + 2 * wq - 1;
            }
            rowf[m] = pf;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
#pragma unroll
    for (int b = 0; b < TN; ++b) {
        const int ng = col0 + b * 32 + c4 * 4;          // GEMM column
        const bool ncol = ng < g.N;
        // dense layout: the wave's 32 columns are 32 channels of ONE pixel column wq (gate_C % 32 == 0)
        const int wq = dense ? ng / C : 0;
        const int n = dense ? ng - wq * C : ng;          // channel
        float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ncol) { sc = *(const float4*)(g.scale + n); sh = *(const float4*)(g.shift + n); }
        ep_touch(sc); ep_touch(sh);
        float s_scale[4] = {0.f, 0.f, 0.f, 0.f}, s_shift[4] = {0.f, 0.f, 0.f, 0.f}, s_bias[4] = {0.f, 0.f, 0.f, 0.f};
        const float scv[4] = {sc.x, sc.y, sc.z, sc.w}, shv[4] = {sh.x, sh.y, sh.z, sh.w};
#pragma unroll
        for (int a = 0; a < TM; ++a) {
#pragma unroll
            for (int r = 0; r < 16; ++r) scratch[((r & 3) + 8 * (r >> 2) + 4 * lh) * 33 + li] = acc[a][b][r];
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int row = it * 8 + rsub;
                const int m = row0 + a * 32 + row;
                const int ra = dense ? (rowa[m] < 0 ? -1 : rowa[m] + wq) : rowa[m];
                const float* sp = scratch + row * 33 + c4 * 4;
                float v[4] = {sp[0], sp[1], sp[2], sp[3]};
                if (ra < 0 || !ncol) continue;
                if (g.accumulate) {
                    const float4 p = *(const float4*)(g.out_y + (long)ra * g.ldo_y + n);
                    v[0] += p.x; v[1] += p.y; v[2] += p.z; v[3] += p.w;
                }
                if (g.gate_mode == 1 || dense) {
                    const float4 a4 = *(const float4*)(g.gate_a + (long)ra * C + n);
                    const float av[4] = {a4.x, a4.y, a4.z, a4.w};
                    float d[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        s_shift[e] += v[e];
                        s_scale[e] = fmaf(v[e], av[e], s_scale[e]);
                        d[e] = av[e] > 0.f ? v[e] * scv[e] : 0.f;
                        s_bias[e] += d[e];
                    }
                    if (g.nt_store) __builtin_nontemporal_store(ep_nt_f4{d[0], d[1], d[2], d[3]}, (ep_nt_f4*)(g.gate_dz + (long)ra * C + n));
                    else *(float4*)(g.gate_dz + (long)ra * C + n) = make_float4(d[0], d[1], d[2], d[3]);
                } else {
                    const long pf = rowf[m];
                    const long off[4] = {pf, pf + 1, pf + WPf, pf + WPf + 1};
                    float av[4][4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float4 a4 = *(const float4*)(g.gate_a + off[k] * C + n);
                        av[k][0] = a4.x; av[k][1] = a4.y; av[k][2] = a4.z; av[k][3] = a4.w;
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        float d[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float gk;
                            if (g.gate_mode == 2) gk = 0.25f * v[e];
                            else gk = gate_pick(fmaf(scv[e], av[0][e], shv[e]), fmaf(scv[e], av[1][e], shv[e]),
                                                fmaf(scv[e], av[2][e], shv[e]), fmaf(scv[e], av[3][e], shv[e]), k, v[e]);
                            s_shift[e] += gk;
                            s_scale[e] = fmaf(gk, av[k][e], s_scale[e]);
                            d[e] = av[k][e] > 0.f ? gk * scv[e] : 0.f;
                            s_bias[e] += d[e];
                        }
                        if (g.nt_store) __builtin_nontemporal_store(ep_nt_f4{d[0], d[1], d[2], d[3]}, (ep_nt_f4*)(g.gate_dz + off[k] * C + n));
                        else *(float4*)(g.gate_dz + off[k] * C + n) = make_float4(d[0], d[1], d[2], d[3]);
                    }
                }
            }
        }
        // rows of this wave: lanes with equal c4 differ in bits 3..5 of the lane id; fixed shuffle order
#pragma unroll
        for (int e = 0; e < 4; ++e) {
#pragma unroll
            for (int o = 8; o < 64; o <<= 1) {
                s_scale[e] += __shfl_xor(s_scale[e], o, 64);
                s_shift[e] += __shfl_xor(s_shift[e], o, 64);
                s_bias[e] += __shfl_xor(s_bias[e], o, 64);
            }
        }
        if (rsub == 0 && ncol) {
            float* pr = g.gate_part + ((long)part_row * (dense ? g.gate_W : 1) + wq) * 3 * C + n;
            *(float4*)(pr) = make_float4(s_scale[0], s_scale[1], s_scale[2], s_scale[3]);
            *(float4*)(pr + C) = make_float4(s_shift[0], s_shift[1], s_shift[2], s_shift[3]);
            *(float4*)(pr + 2 * C) = make_float4(s_bias[0], s_bias[1], s_bias[2], s_bias[3]);
        }
    }
}

// RMASK (round 5, gemm1_relumask_kernel only -- a template parameter, so that no other kernel's code changes): out_y is the data-gradient
// of a Dense(relu) layer's OUTPUT side, i.e. it is zeroed where that layer's activation g.gate_a (same rows and pitch as out_y) is not
// positive -- asr_relu_bwd applied to the value in the register instead of a pass of its own over the result.
template <int TM, int TN, bool RMASK = false>
__device__ __forceinline__ void tap_epilogue(const TapGemmArgs& g, const floatx16 (&acc)[TM][TN], float* scratch,
                                             const int* rowa, const int* rowy, int row0, int col0, int lane, int part_row = 0) {
    if (!RMASK && g.gate_mode) {
        tap_epilogue_gated<TM, TN>(g, acc, scratch, rowa, const_cast<int*>(rowy), row0, col0, lane, part_row);
        return;
    }
    const int li = lane & 31, lh = lane >> 5;
    const int c4 = lane & 7, rsub = lane >> 3;
#pragma unroll
    for (int b = 0; b < TN; ++b) {
        const int n = col0 + b * 32 + c4 * 4;
        const bool ncol = n < g.N;
        float4 bs = make_float4(0.f, 0.f, 0.f, 0.f), sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = bs;
        if (ncol) {
            if (g.bias) bs = *(const float4*)(g.bias + n);
            if (g.scale) sc = *(const float4*)(g.scale + n);
            if (g.shift) sh = *(const float4*)(g.shift + n);
        }
        ep_touch(bs); ep_touch(sc); ep_touch(sh);
#pragma unroll
        for (int a = 0; a < TM; ++a) {
#pragma unroll
            for (int r = 0; r < 16; ++r) scratch[((r & 3) + 8 * (r >> 2) + 4 * lh) * 33 + li] = acc[a][b][r];
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int row = it * 8 + rsub;
                const int m = row0 + a * 32 + row;
                const int ra = rowa[m];
                const float* sp = scratch + row * 33 + c4 * 4;
                float4 v = make_float4(sp[0] + bs.x, sp[1] + bs.y, sp[2] + bs.z, sp[3] + bs.w);
                if (ra < 0 || !ncol) continue;
                if (g.relu == 1) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                else if (g.relu == 2) { v.x = tanhf(v.x); v.y = tanhf(v.y); v.z = tanhf(v.z); v.w = tanhf(v.w); }
                typedef float nt_f4 __attribute__((ext_vector_type(4)));
                if (g.out_a) {
                    if (g.nt_store) __builtin_nontemporal_store(nt_f4{v.x, v.y, v.z, v.w}, (nt_f4*)(g.out_a + (long)ra * g.ldo_a + n));
                    else *(float4*)(g.out_a + (long)ra * g.ldo_a + n) = v;
                }
                if (g.out_y) {
                    float4 y = make_float4(sc.x * v.x + sh.x, sc.y * v.y + sh.y, sc.z * v.z + sh.z, sc.w * v.w + sh.w);
                    float* o = g.out_y + (long)rowy[m] * g.ldo_y + n;
                    if (RMASK) {
                        const float4 h = *(const float4*)(g.gate_a + (long)rowy[m] * g.ldo_y + n);
                        y.x = h.x > 0.f ? y.x : 0.f; y.y = h.y > 0.f ? y.y : 0.f; y.z = h.z > 0.f ? y.z : 0.f; y.w = h.w > 0.f ? y.w : 0.f;
                    }
                    if (g.accumulate) { const float4 p = *(const float4*)o; y.x += p.x; y.y += p.y; y.z += p.z; y.w += p.w; }
                    if (g.nt_store) __builtin_nontemporal_store(nt_f4{y.x, y.y, y.z, y.w}, (nt_f4*)o);
                    else *(float4*)o = y;
                }
            }
        }
    }
}


}  // namespace

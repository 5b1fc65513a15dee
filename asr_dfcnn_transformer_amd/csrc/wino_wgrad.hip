// Weight gradient of the 3x3 convolutions by Winograd F(3x3, 2x2) on the fp32 MFMA pipe (round 3): per 2x2 tile of the output
// gradient, input channel and output channel 16 multiplies instead of 36, still fp32.
//
//   dW = A'^T [ sum_tiles (G' dy G'^T) (.) (B^T d B) ] A'        d  = 4x4 input patch of the tile (the forward's patch, wino.hip)
//                                                                 dy = its 2x2 output-gradient pixels
//   B^T = the forward's input transform, G' = [[1,0],[1,1],[1,-1],[0,1]] (the halves of G moved into A'),
//   A'^T = [[1,.5,.5,0],[0,.5,-.5,0],[0,.5,.5,-1]]
//
// One GEMM per transform position xi = 0..15 with the TILES as the contraction index:
//   M_xi[ci][co] = sum_tile V_xi[tile][ci] * Z_xi[tile][co]
// v_mfma_f32_32x32x2_f32 contracts two tiles per instruction: lane (l % 32, l / 32) supplies the A value of input channel
// l % 32 and the B value of output channel l % 32 for tile l / 32 of the pair -- so BOTH transforms are lane-local with the
// channel on the lane and the pixels in registers, and the planes' own [pixel][channel] layout is what LDS holds (rows of
// 128 / 256 contiguous bytes per pixel: every DMA piece is whole lines, every ds_read_b32 of a half-wave 128 contiguous bytes).
// There is no per-item tail: a workgroup owns a 64 (or 32) x 64 block of (ci, co) pairs for all sixteen positions -- half the
// register file of the CU -- walks its share of the tiles and writes its accumulators ONCE, as a partial
// [16][ci][co]; wino_wgrad_reduce_kernel adds the partials in a fixed order (deterministic) and applies A'^T . A'.
//
// Work: the tile columns of an image are cut into blocks of <= 13 columns; a STAGE is two tile rows of one block = 2 w tiles = w
// tile pairs.  Its input region (6 pixel rows x (2 w + 2) pixels) and gradient region (4 x 2 w pixels) arrive by buffer-form LDS-DMA
// in two buffer sets, one barrier per stage; pixels outside the plane are sent out of the buffer's range and read zeros.  Sixteen
// waves (round 4; round 3: eight, two transform columns each): wave = row * 4 + a * 2 + wn; `row` the transform row it owns (4
// accumulators), wn the 32-wide half of the 64 output channels, a the 32-wide half of 64 input channels (CINB = 64) or, for 32
// input channels (CINB = 32), every other tile row (two partials per workgroup).  Two tile pairs are transformed together with
// packed adds.
// Round 5: 32 input x 32 OUTPUT channels (acoustic_model2.py:39 `h1_1 = cnn_cell(32, h1)`, the 'small' graph): the block of (ci, co)
// pairs is a quarter of the 64 x 64 one, so a stage is FOUR tile rows (10 x 32 input pixels, 8 x 32 gradient pixels) and the four waves
// of a transform row take one tile row each (four partials per workgroup) -- per wave and stage the work of the CINB = 32 form.
#include "asr_common.h"
#include <stdint.h>

namespace {

typedef __attribute__((address_space(3))) float ww_lds_f;
typedef float ww_f2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) char ww_lds_c;

#ifdef ASR_DEV_HOOKS                 // development builds only (tools/ablate_wino_wgrad.sh); the product build never defines it
#include "dev_hooks.h"
#endif
#ifndef WW_SLOT
// where a DMA piece of the next stage goes between the eight MFMAs of a tile pair: two per pair (measured: four per pair -3 %,
// three -4 %, one -3..-5 %)
#define WW_SLOT(i) ((i) == 3 || (i) == 7)
#endif
constexpr int WW_MAXW = 13;          // tile columns of a column block

// CINB x NB = the workgroup's block of (input, output) channels: 64 x 64, 32 x 64 (stages of two tile rows) or 32 x 32 (four)
template <int CINB, int NB> struct WwCfg {
    static constexpr int TR = NB == 64 ? 2 : 4;                 // tile rows of a stage
    static constexpr int XP = CINB == 64 ? 28 : 32;             // pixel pitch of a staged input row (2 w + 2 <= 28), whole pieces
    static constexpr int XPX = 1024 / (CINB * 4);               // pixels per 1 KB piece
    static constexpr int XPPR = XP / XPX;                       // pieces per input row
    static constexpr int NXP = (2 * TR + 2) * XPPR;             // input pieces per stage
    static constexpr int ZP = NB == 64 ? 28 : 32;               // pixel pitch of a staged gradient row (2 w <= 26), whole pieces
    static constexpr int ZPX = 1024 / (NB * 4);                 // pixels per 1 KB gradient piece
    static constexpr int ZPPR = ZP / ZPX;
    static constexpr int NZP = 2 * TR * ZPPR;                   // gradient pieces per stage
    static constexpr int XF = (2 * TR + 2) * XP * CINB;         // floats of the input region
    static constexpr int ZF = 2 * TR * ZP * NB;
    static constexpr int SETF = XF + ZF;
    static constexpr int NTH = CINB == 64 ? 1 : NB == 64 ? 2 : 4;      // partials per workgroup (tile-row classes)
};

struct WwArgs {
    const float* A; const float* Z; float* part;
    int K, N, lda, ldz, B, H, Wd, WP, HPWP;
    int SR;                  // stage rows per image = ceil(TH / tile rows of a stage)
    int ncb; int cb_tj0[8]; int cb_w[8];
    int nstages, nsl, nbp, ncob;   // stages in all, slices (workgroups per block pair), block pairs, output-channel blocks (N / 64)
    float inv_per, inv_ncb;        // 1 / (SR * ncb), 1 / ncb: the stage index divisions are a convert, a multiply-add and a convert
    int cb_base, cb_rem;           // column block c is cb_base + (c < cb_rem) tile columns wide (cb_w / cb_tj0 without a table look-up)
};

__device__ __forceinline__ void ww_barrier_dma() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

#if __HIP_DEVICE_COMPILE__
struct WwStage { int b, row0, tj0, w; };

template <int TR>
__device__ __forceinline__ WwStage ww_stage(const WwArgs& a, int g) {
    WwStage s;
    // (exact for g < 2^22: floor((x + 0.5) / d) = floor(x / d); every wave of the workgroup pays this once per stage, in lock-step
    // behind the stage barrier -- two integer divisions here were ~70 vector instructions of idle matrix pipe per stage)
    const int per = a.SR * a.ncb;
    s.b = (int)(((float)g + 0.5f) * a.inv_per);
    const int rem = g - s.b * per;
    const int sr = (int)(((float)rem + 0.5f) * a.inv_ncb), cb = rem - sr * a.ncb;
    s.row0 = 2 * TR * sr;            // first padded pixel row of the input region; the gradient region starts one row lower
    // (from the two numbers the blocks are cut by, not from the tables: an indexed scalar load here sat on every wave's way to the barrier)
    s.w = a.cb_base + (cb < a.cb_rem ? 1 : 0);
    s.tj0 = cb * a.cb_base + (cb < a.cb_rem ? cb : a.cb_rem);
    return s;
}

// DMA piece j (pieces wave + 8 j) of stage s into buffer set `set`
template <int CINB, int NB, class R>
__device__ __forceinline__ void ww_piece(const WwArgs& a, const WwStage& s, R rx, R rz, float* __restrict__ set, int wave, int j, int cin0,
                                         int co0, unsigned vox, unsigned voz, int pxx, int pxz) {
    typedef WwCfg<CINB, NB> C;
    const int p = wave + 8 * j;
#ifdef WW_NO_DMA
    return;
#endif
    if (p < C::NXP) {
        const int r = p / C::XPPR, pc = p - r * C::XPPR;
        const int row = s.row0 + r, col0 = 2 * s.tj0 + pc * C::XPX;
        int lim = 2 * s.w + 2 - pc * C::XPX;                      // pixels of this piece inside the region ...
        const int limp = a.Wd + 1 - col0;                         // ... and inside the plane (columns 0 .. W)
        if (limp < lim) lim = limp;
        if (row > a.H) lim = 0;
        const unsigned v = pxx < lim ? vox : 0xFFFFFFF0u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (ww_lds_f*)(set + p * 256), 16, v,
                                                 (int)((((unsigned)(s.b * a.HPWP + row * a.WP + col0)) * (unsigned)a.lda + (unsigned)cin0) * 4u), 0, 0);
    } else if (p < C::NXP + C::NZP) {
        const int q = p - C::NXP;
        const int r = q / C::ZPPR, pc = q - r * C::ZPPR;
        const int row = s.row0 + 1 + r, col0 = 2 * s.tj0 + 1 + pc * C::ZPX;
        int lim = 2 * s.w - pc * C::ZPX;
        const int limp = a.Wd + 1 - col0;
        if (limp < lim) lim = limp;
        if (row > a.H) lim = 0;
        const unsigned v = pxz < lim ? voz : 0xFFFFFFF0u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rz, (ww_lds_f*)(set + C::XF + q * 256), 16, v,
                                                 (int)((((unsigned)(s.b * a.HPWP + row * a.WP + col0)) * (unsigned)a.ldz + (unsigned)co0) * 4u), 0, 0);
    }
}

#endif

// ------------------------------------------------------------------------------------------------ sixteen-wave form (round 4)
// The same work with a wave owning ONE ROW of the 4 x 4 transform (4 accumulators = 64 registers): sixteen waves of 128 registers,
// wave = row * 4 + a * 2 + wn -- FOUR waves per SIMD instead of two, as in wino11_kernel (wino.hip) and for the same reasons: row r
// of V = B^T d B needs two pixel rows of the patch (0: rows 0, 2; 1, 2: rows 1, 2; 3: rows 1, 3) and all four columns -- 8 packed
// adds per tile-pair pair instead of 20 for a column pair --, row r of Z = G' y G'^T needs one or both gradient rows (2-4 packed
// adds instead of 8), and four instruction streams per SIMD cover each other's LDS round trips and the stage barrier.
#if __HIP_DEVICE_COMPILE__
// wa: the tile-row class of the wave (CINB 64: unused -- every wave walks both rows; 32 x 64: 0 / 1; 32 x 32: 0 .. 3)
template <int RR, int CINB, int NB, class R>
__device__ __forceinline__ void ww4_compute(const WwArgs& a, const char* __restrict__ xs, const char* __restrict__ zs, float* __restrict__ nxt,
                                            const WwStage& sn, bool more, R rx, R rz, int w, int parity, int wave, int wa, int lh, int xch, int zch,
                                            int cin0, int co0, unsigned vox, unsigned voz, int pxx, int pxz, floatx16 (&acc)[4]) {
    typedef WwCfg<CINB, NB> C;
    constexpr int NJ = (C::NXP + C::NZP + 15) / 16;    // piece rounds per stage (pieces wave + 16 j)
    constexpr int PXB = CINB * 4;                      // bytes of an input pixel
    constexpr int ZXB = NB * 4;                        // bytes of a gradient pixel
    constexpr int RA = RR == 0 ? 0 : 1, RB = RR == 3 ? 3 : 2;           // the two patch rows this transform row combines
    int jn = 0;
    auto piece = [&]() {
        if (more && jn < NJ) {
            const int p = wave + 16 * jn;
            // (ww_piece numbers pieces wave + 8 j: hand it the equivalent (wave', j') of piece p)
            ww_piece<CINB, NB>(a, sn, rx, rz, nxt, p & 7, p >> 3, cin0, co0, vox, voz, pxx, pxz);
        }
        ++jn;
    };
    auto step = [&](int xa, int za, bool two) {
        // the two values of a packed register are the same pixel of the two tile pairs (one address register, two immediates)
        ww_f2 da[4], db[4], y0[2], y1[2];
        const unsigned xad = (unsigned)(uintptr_t)(const ww_lds_c*)(xs + xa), zad = (unsigned)(uintptr_t)(const ww_lds_c*)(zs + za);
        // (pixels of 256 bytes -- 64 channels --: the pair is ONE ds_read2st64_b32, whose two offsets count 256-byte units)
#define WW_LD2(dst, ad, off0, off1) {                                                                                      \
        if (((off0) & 255) == 0 && ((off1) & 255) == 0 && (off1) < 65536) {                                                \
            asm volatile("ds_read2st64_b32 %0, %1 offset0:%2 offset1:%3" : "=&v"(dst) : "v"(ad), "i"((off0) >> 8), "i"((off1) >> 8)); \
        } else { float lo_, hi_;                                                                                           \
            asm volatile("ds_read_b32 %0, %2 offset:%3\n\tds_read_b32 %1, %2 offset:%4" : "=&v"(lo_), "=&v"(hi_) : "v"(ad), "i"(off0), "i"(off1)); \
            dst = ww_f2{lo_, hi_}; } }
#define WW_LDX(dst, r, c) WW_LD2(dst, xad, ((r) * C::XP + (c)) * PXB, ((r) * C::XP + (c) + 4) * PXB)
#define WW_LDZ(dst, p) WW_LD2(dst, zad, (((p) >> 1) * C::ZP + ((p) & 1)) * ZXB, (((p) >> 1) * C::ZP + ((p) & 1) + 4) * ZXB)
        WW_LDX(da[0], RA, 0) WW_LDX(da[1], RA, 1) WW_LDX(da[2], RA, 2) WW_LDX(da[3], RA, 3)
        WW_LDX(db[0], RB, 0) WW_LDX(db[1], RB, 1) WW_LDX(db[2], RB, 2) WW_LDX(db[3], RB, 3)
        if (RR != 3) { WW_LDZ(y0[0], 0) WW_LDZ(y0[1], 1) }
        if (RR != 0) { WW_LDZ(y1[0], 2) WW_LDZ(y1[1], 3) }
#undef WW_LDX
#undef WW_LDZ
#undef WW_LD2
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        ww_f2 t[4], v[4], yr[2], z[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) t[c] = RR == 1 ? da[c] + db[c] : RR == 2 ? db[c] - da[c] : da[c] - db[c];
        v[0] = t[0] - t[2]; v[1] = t[1] + t[2]; v[2] = t[2] - t[1]; v[3] = t[1] - t[3];
#pragma unroll
        for (int b = 0; b < 2; ++b) yr[b] = RR == 0 ? y0[b] : RR == 1 ? y0[b] + y1[b] : RR == 2 ? y0[b] - y1[b] : y1[b];
        z[0] = yr[0]; z[1] = yr[0] + yr[1]; z[2] = yr[0] - yr[1]; z[3] = yr[1];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[i].x, z[i].x, acc[i], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        piece();
        __builtin_amdgcn_sched_barrier(0);
        if (two) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[i].y, z[i].y, acc[i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            piece();
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    const int np = w >> 1;                             // whole pairs per tile row
    const int xl = xch + lh * 2 * PXB, zl = zch + lh * 2 * ZXB;       // this lane's tile of a pair: two pixels further for the upper half
    // Wave priority falls as the wave advances through its stage (3 for the first quarter of its steps ... 0 for the last): the four
    // waves of a SIMD then take turns at the matrix pipe instead of finishing one after the other.  In-kernel stamps (s_memtime at the
    // stage barrier and around the steps, 64 -> 128 at 400 x 50): the arbiter serves the oldest wave first -- wave 0 was through a stage
    // after 8 400 cycles and waited 9 200 at the barrier, wave 15 needed 17 100 -- and the last wave of a SIMD, alone, cannot keep the
    // pipe busy; with the falling priority every wave takes 16 000-16 400.
    // (an odd block width: the last tiles of two neighbouring tile rows form one pair -- lane half = row; of the two waves that own
    // those rows the one whose class matches the stage parity takes it)
    const int odd = ((w & 1) && (CINB == 64 || (NB == 64 ? wa : (wa & 1)) == (parity & 1))) ? 1 : 0;
    const int nst = (CINB == 64 ? 2 : 1) * ((np + 1) >> 1) + odd;
    const int t1 = (nst + 3) >> 2, t2 = (nst + 1) >> 1, t3 = (3 * nst + 3) >> 2;
    int kst = 0;
    auto prio = [&]() {
        if (kst < t1) __builtin_amdgcn_s_setprio(3);
        else if (kst < t2) __builtin_amdgcn_s_setprio(2);
        else if (kst < t3) __builtin_amdgcn_s_setprio(1);
        else __builtin_amdgcn_s_setprio(0);
        ++kst;
    };
    for (int row = (CINB == 64 ? 0 : wa); row < (CINB == 64 ? 2 : wa + 1); ++row)
        for (int q = 0; q < np; q += 2) {
            prio();
            step(xl + row * (2 * C::XP * PXB) + q * (4 * PXB), zl + row * (2 * C::ZP * ZXB) + q * (4 * ZXB), q + 1 < np);
        }
    if (odd) {
        prio();
        const int r0 = NB == 64 ? 0 : (wa & ~1);        // the row pair whose last tiles meet (two-row stages: rows 0 and 1)
        step(xch + (r0 + lh) * (2 * C::XP * PXB) + (w - 1) * 2 * PXB, zch + (r0 + lh) * (2 * C::ZP * ZXB) + (w - 1) * 2 * ZXB, false);
    }
    // pieces the loop had no slot for (narrow blocks)
    while (more && jn < NJ) piece();
}

template <int RR, int CINB, int NB>
__device__ __forceinline__ void ww4_body(const WwArgs& a, float* smem) {
    typedef WwCfg<CINB, NB> C;
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, lh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);           // = RR * 4 + a * 2 + wn  (NB 32: RR * 4 + tile row)
    const int wa = NB == 64 ? (wave >> 1) & 1 : wave & 3, wn = NB == 64 ? wave & 1 : 0;
    // workgroup -> (block pair, slice): workgroups of one slice differ by multiples of 8 (the same XCD: they read the same pixels)
    const int wg = blockIdx.x;
    const int sl_lo = wg & 7, rest = wg >> 3;
    const int bp = rest % a.nbp, sl = (rest / a.nbp) * 8 + sl_lo;
    if (sl >= a.nsl) return;
    const int cib = bp / a.ncob, cob = bp - cib * a.ncob;
    const int cin0 = cib * CINB, co0 = cob * NB;

    auto rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.A, 0, 0x7FFFFFF0, 0x00020000);
    auto rz = __builtin_amdgcn_make_buffer_rsrc((void*)a.Z, 0, 0x7FFFFFF0, 0x00020000);
    constexpr int LPPX = CINB / 4, LPPZ = NB / 4;               // lanes (16 bytes each) per staged pixel
    const int pxx = lane / LPPX, pxz = lane >> (NB == 64 ? 4 : 3);
    const unsigned vox = (unsigned)((pxx * a.lda + (lane % LPPX) * 4) * 4);
    const unsigned voz = (unsigned)((pxz * a.ldz + (lane & (LPPZ - 1)) * 4) * 4);
    constexpr int NJ = (C::NXP + C::NZP + 15) / 16;

    floatx16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    int g = sl;
    {
        const WwStage s0 = ww_stage<C::TR>(a, g);
#pragma unroll
        for (int j = 0; j < NJ; ++j) { const int p = wave + 16 * j; ww_piece<CINB, NB>(a, s0, rx, rz, smem, p & 7, p >> 3, cin0, co0, vox, voz, pxx, pxz); }
    }
    int cur = 0;
    const int xch = ((CINB == 64 ? wa * 32 : 0) + li) * 4;      // byte offset of this lane's input channel inside a pixel
    const int zch = (wn * 32 + li) * 4;
    WwStage s = ww_stage<C::TR>(a, g);
    for (; g < a.nstages; g += a.nsl) {
        const int gn = g + a.nsl;
        const bool more = gn < a.nstages;
        WwStage sn = s;
        if (more) sn = ww_stage<C::TR>(a, gn);
        ww_barrier_dma();                            // this stage has landed; nobody reads the other set any more
        const char* xs = (const char*)(smem + cur * C::SETF);
        ww4_compute<RR, CINB, NB>(a, xs, xs + C::XF * 4, smem + (cur ^ 1) * C::SETF, sn, more, rx, rz, s.w, C::TR == 2 ? s.row0 >> 2 : s.row0 >> 3, wave, wa, lh, xch, zch, cin0, co0,
                              vox, voz, pxx, pxz, acc);
        cur ^= 1;
        s = sn;
    }
    // partial of this workgroup (CINB 32: one per tile-row class of the stages): [16][CINB][NB]; acc[c] = position RR * 4 + c,
    // register = input channel, lane = output channel
    constexpr int nth = C::NTH;
    float* P = a.part + ((long)(sl * a.nbp + bp) * nth + (CINB == 64 ? 0 : wa)) * 16 * CINB * NB;
    const int ci0 = CINB == 64 ? wa * 32 : 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int xi = RR * 4 + i;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ci = ci0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            P[((long)xi * CINB + ci) * NB + wn * 32 + li] = acc[i][r];
        }
    }
}
#endif

template <int CINB>
__global__ __launch_bounds__(1024, 4) void wino_wgrad4_kernel(WwArgs a) {
#if __HIP_DEVICE_COMPILE__
    extern __shared__ __attribute__((aligned(16))) float smem[];
    switch (threadIdx.x >> 8) {
        case 0: ww4_body<0, CINB, 64>(a, smem); break;
        case 1: ww4_body<1, CINB, 64>(a, smem); break;
        case 2: ww4_body<2, CINB, 64>(a, smem); break;
        default: ww4_body<3, CINB, 64>(a, smem); break;
    }
#endif
}

// 32 input x 32 output channels per workgroup, stages of four tile rows (a kernel name of its own: one rocprof row)
__global__ __launch_bounds__(1024, 4) void wino_wgrad4n32_kernel(WwArgs a) {
#if __HIP_DEVICE_COMPILE__
    extern __shared__ __attribute__((aligned(16))) float smem[];
    switch (threadIdx.x >> 8) {
        case 0: ww4_body<0, 32, 32>(a, smem); break;
        case 1: ww4_body<1, 32, 32>(a, smem); break;
        case 2: ww4_body<2, 32, 32>(a, smem); break;
        default: ww4_body<3, 32, 32>(a, smem); break;
    }
#endif
}

// dW[kh][kw][k][n] = A'^T (sum over the partials of M[.][k][n]) A'.  One workgroup per (input channel k, block of 64 output
// channels): thread (co, g) adds the partials of group g (a contiguous range, ascending) for its sixteen positions -- sixteen
// independent loads per partial, 128-byte rows per half-wave --, the groups' sums are added in ascending order through LDS: a fixed
// association, bitwise reproducible.  (The first form -- one thread per (k, n) walking all partials -- took 77 us per launch:
// 32-512 workgroups of dependent loads for 64 MB.)
constexpr int WW_RG = 16;            // groups of partials per workgroup
template <int NBW>                   // output channels of a block = blockDim.x: 64, or 32 for the 32 x 32 configuration (four partials per workgroup)
__global__ __launch_bounds__(64 * WW_RG) void wino_wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dW, int K, int N,
                                                                         int cinb, int nparts, int nbp, int ncob) {
    constexpr int nb = NBW;
    const int nth = NBW == 32 ? 4 : (cinb == 64 ? 1 : 2);
    __shared__ float red[WW_RG][16][64];
    const int co = threadIdx.x, g = threadIdx.y;
    const int k = blockIdx.x / ncob, cob = blockIdx.x - k * ncob;
    const int cib = k / cinb, ci = k - cib * cinb;
    const int bp = cib * ncob + cob;
    const long pstride = (long)16 * cinb * nb;
    const int PT = nparts * nth;                                  // partial (p, t) has index p * nth + t: (p * nbp + bp) * nth + t in memory
    const int lo = (int)((long)PT * g / WW_RG), hi = (int)((long)PT * (g + 1) / WW_RG);
    float m[16];
#pragma unroll
    for (int x = 0; x < 16; ++x) m[x] = 0.f;
    for (int i = lo; i < hi; ++i) {
        const int p = i / nth, t = i - p * nth;
        const float* q = part + ((long)(p * nbp + bp) * nth + t) * pstride + (long)ci * nb + co;
#pragma unroll
        for (int x = 0; x < 16; ++x) m[x] += q[(long)x * cinb * nb];
    }
#pragma unroll
    for (int x = 0; x < 16; ++x) red[g][x][co] = m[x];
    __syncthreads();
    if (g != 0) return;
#pragma unroll
    for (int x = 0; x < 16; ++x) {
        float s = red[0][x][co];
        for (int j = 1; j < WW_RG; ++j) s += red[j][x][co];
        m[x] = s;
    }
    const int n = cob * nb + co;
    // rows: t[kh][c] = A'^T m[.][c]
    float t[3][4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float m0 = m[0 * 4 + c], m1 = m[1 * 4 + c], m2 = m[2 * 4 + c], m3 = m[3 * 4 + c];
        t[0][c] = m0 + 0.5f * (m1 + m2);
        t[1][c] = 0.5f * (m1 - m2);
        t[2][c] = 0.5f * (m1 + m2) - m3;
    }
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
        const float w0 = t[kh][0] + 0.5f * (t[kh][1] + t[kh][2]);
        const float w1 = 0.5f * (t[kh][1] - t[kh][2]);
        const float w2 = 0.5f * (t[kh][1] + t[kh][2]) - t[kh][3];
        dW[((long)(kh * 3 + 0) * K + k) * N + n] = w0;
        dW[((long)(kh * 3 + 1) * K + k) * N + n] = w1;
        dW[((long)(kh * 3 + 2) * K + k) * N + n] = w2;
    }
}

struct WwPlan { bool ok; int cinb, nb, nth, ncb, cbw[8], cbt[8], SR, nstages, nbp, ncob, nsl, nparts; size_t ws; };

WwPlan ww_plan(const asr_gemm_desc* d, int ldz) {
    WwPlan p;
    p.ok = false; p.ws = 0;
    // (an odd height is a half-filled last tile row: the gradient row under the image is the zero border, rows past it are sent
    // out of range -- ww_piece's `row > a.H` -- so the phantom pixels contribute exact zeros)
    if (!d || d->ntaps != 9 || d->wmode != 0 || d->H < 2 || d->W < 2) return p;
    if (d->K != 32 && (d->K % 64) != 0) return p;
    const bool n32 = d->K == 32 && d->N == 32;                  // round 5: the 32 -> 32 layers (32 x 32 blocks, stages of four tile rows)
    if (((d->N % 64) != 0 && !n32) || (d->lda & 3) || (ldz & 3) || d->M != d->B * (d->H + 1) * (d->W + 1)) return p;
    if ((long)d->M * d->lda * 4 >= 0x7FFFFFF0L || (long)d->M * ldz * 4 >= 0x7FFFFFF0L) return p;
    const int TH = (d->H + 1) / 2, TW = (d->W + 1) / 2;
    p.cinb = d->K == 32 ? 32 : 64;
    p.nb = n32 ? 32 : 64;
    p.nth = p.cinb == 64 ? 1 : n32 ? 4 : 2;
    p.ncb = asr_cdiv(TW, WW_MAXW);
    if (p.ncb > 8) return p;
    int tj = 0;
    for (int c = 0; c < p.ncb; ++c) {
        p.cbw[c] = TW / p.ncb + (c < TW % p.ncb ? 1 : 0);
        p.cbt[c] = tj; tj += p.cbw[c];
    }
    p.SR = asr_cdiv(TH, n32 ? 4 : 2);
    p.nstages = d->B * p.SR * p.ncb;
    if ((long)d->B * p.SR * p.ncb >= (1L << 22)) return p;      // ww_stage's index divisions are exact below 2^22
    p.ncob = d->N / p.nb;
    p.nbp = (d->K / p.cinb) * p.ncob;
    static int ncu = 0;
    if (!ncu) {
        int dev = 0; hipDeviceProp_t pr;
        ncu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) ? pr.multiProcessorCount : 256;
    }
    // enough work per workgroup to amortise the partial it writes (256 KB: ~1.6 stages' worth of a CU's share of HBM): at least 8 stages --
    // or 4 where 8 would leave CUs without a workgroup (round 6: the per-GPU batches of a strong-scaling run, B <= 10 on the 200 x 25
    // planes; at B = 4 the 128 -> 128 launch ran 100 workgroups of 8 stages on 256 CUs: 84 us where an eighth of the B = 32 launch is 44)
    int nsl = ncu / p.nbp;
    if (nsl < 1) nsl = 1;
    if (nsl > p.nstages / 8) { const int n4 = p.nstages / 4; nsl = n4 < nsl ? n4 : nsl; if (nsl < p.nstages / 8) nsl = p.nstages / 8; }
    if (nsl < 1) return p;                       // too small a problem: the direct kernels
    p.nsl = nsl;
    p.nparts = nsl;
    p.ws = (size_t)nsl * p.nbp * p.nth * 16 * p.cinb * p.nb * sizeof(float);
    p.ok = true;
    return p;
}

}  // namespace

size_t asr_wino_wgrad_workspace(const asr_gemm_desc* d, int ldz) {      // (plain C++ linkage: used by tap_wgrad.hip only)
    const WwPlan p = ww_plan(d, ldz);
    return p.ok ? p.ws : 0;
}

// ASR_OK when the Winograd weight gradient ran; ASR_ERR_UNSUPPORTED when the caller should use the direct kernels
int asr_wino_wgrad_launch(const asr_gemm_desc* d, const float* A, const float* dZ, int ldz, float* dW, float* partials, void* stream) {
    const WwPlan p = ww_plan(d, ldz);
    if (!p.ok || !partials || ((((uintptr_t)A) | ((uintptr_t)dZ) | ((uintptr_t)partials)) & 15)) return ASR_ERR_UNSUPPORTED;
    WwArgs a;
    a.A = A; a.Z = dZ; a.part = partials;
    a.K = d->K; a.N = d->N; a.lda = d->lda; a.ldz = ldz; a.B = d->B; a.H = d->H; a.Wd = d->W; a.WP = d->W + 1; a.HPWP = (d->H + 1) * (d->W + 1);
    a.SR = p.SR; a.ncb = p.ncb;
    for (int c = 0; c < 8; ++c) { a.cb_tj0[c] = c < p.ncb ? p.cbt[c] : 0; a.cb_w[c] = c < p.ncb ? p.cbw[c] : 0; }
    a.nstages = p.nstages; a.nsl = p.nsl; a.nbp = p.nbp; a.ncob = p.ncob;
    a.inv_per = 1.0f / (float)(p.SR * p.ncb); a.inv_ncb = 1.0f / (float)p.ncb;
    a.cb_base = ((d->W + 1) / 2) / p.ncb; a.cb_rem = ((d->W + 1) / 2) % p.ncb;
    // grid: slices rounded up to whole groups of 8, times block pairs (see ww_body)
    const int grid = asr_cdiv(p.nsl, 8) * 8 * p.nbp;
    hipStream_t st = (hipStream_t)stream;
    if (p.nb == 32) {
        const size_t lds = (size_t)2 * WwCfg<32, 32>::SETF * sizeof(float);
        static_assert(2 * WwCfg<32, 32>::SETF * sizeof(float) <= 160 * 1024, "two buffer sets of the four-row stage must fit the CU's LDS");
        static bool t3232 = false;
        auto k = wino_wgrad4n32_kernel;
        if (!t3232) { (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); t3232 = true; }
        hipLaunchKernelGGL(k, dim3(grid), dim3(1024), lds, st, a);
        ASR_NOTE_KERNEL("wino_wgrad4n32_kernel");
    } else if (p.cinb == 64) {
        const size_t lds = (size_t)2 * WwCfg<64, 64>::SETF * sizeof(float);
        static bool t64 = false;
        auto k = wino_wgrad4_kernel<64>;
        if (!t64) { (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); t64 = true; }
        hipLaunchKernelGGL(k, dim3(grid), dim3(1024), lds, st, a);
        ASR_NOTE_KERNEL("wino_wgrad4_kernel<64>");
    } else {
        const size_t lds = (size_t)2 * WwCfg<32, 64>::SETF * sizeof(float);
        static bool t32 = false;
        auto k = wino_wgrad4_kernel<32>;
        if (!t32) { (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); t32 = true; }
        hipLaunchKernelGGL(k, dim3(grid), dim3(1024), lds, st, a);
        ASR_NOTE_KERNEL("wino_wgrad4_kernel<32>");
    }
    ASR_CHECK_LAUNCH("wino_wgrad");
    if (p.nb == 32)
        hipLaunchKernelGGL(wino_wgrad_reduce_kernel<32>, dim3(d->K * p.ncob), dim3(32, WW_RG), 0, st, partials, dW, d->K, d->N, p.cinb, p.nparts, p.nbp, p.ncob);
    else
        hipLaunchKernelGGL(wino_wgrad_reduce_kernel<64>, dim3(d->K * p.ncob), dim3(64, WW_RG), 0, st, partials, dW, d->K, d->N, p.cinb, p.nparts, p.nbp, p.ncob);
    ASR_CHECK_LAUNCH("wino_wgrad_reduce");
    return ASR_OK;
}

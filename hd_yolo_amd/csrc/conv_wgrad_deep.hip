// Deep-pipelined weight gradient for the wide layers (bf16, C % 64 == 0, K % 128 == 0, K >= 256): dW[k][q] = SUM_p dy[p][k] * x[p][q],
// q = (tap, c), p over the N*Ho*Wo output pixels, as a 256 (k) x 256 (q) output tile per 8-wave workgroup with the pixels streamed.
//
// Why.  conv_wgrad.hip's generic kernel stages 128 x 128 tiles (64 FLOP per staged byte) and drains its LDS-DMA at every 64-pixel block
// (`s_waitcnt vmcnt(0); __syncthreads()`): PMC on 128x256 3x3/s2 @80x80 (profiles/r03_layers_pmc_table.txt): 162 us, MFMA busy 15 %, waves
// parked 44 % — 0.15 of the layer's roofline.  This kernel is conv_deep.hip's 256 x 256 pipeline with the roles of the weight gradient:
//   * a STAGE is 64 pixels: dy [64][256 k] and x-gather [64][256 q] as four 16 KB units [dy-lo, x-lo, x-hi, dy-hi] of two [64 px][128 B]
//     sub-tiles each, in a 2-deep ring (128 KB), loaded by LDS-DMA two stages ahead with counted vmcnt(8) and raw barriers, the two waves
//     of a SIMD one barrier apart; 128 FLOP per staged byte;
//   * waves 2 (k) x 4 (q), wave tile 128 x 64 as 2 x 2 quadrants = the four phases of 16 MFMAs; both operands are TRANSPOSED LDS reads
//     (ds_read_b64_tr_b16: the reduction index is the pixel row of the [pixel][channel] image), read one segment ahead into the registers
//     the segment's MFMAs have just consumed;
//   * the gather: a thread owns ONE pixel row of every stage (its eight DMA pieces are that pixel's 16-byte chunk in each of the 4 + 4
//     sub-tiles); with C % 64 == 0 a 64-column x sub-tile lies inside one tap, so tap offset and channel base are scalars per sub-tile and the
//     per-lane work per stage is one pixel decomposition and one border test per sub-tile; out-of-image taps and rows past the split's end
//     carry an offset beyond the descriptor's range (zeros, conv_igemm.hip's trick);
//   * every workgroup writes ONE fp32 slab tile [256][256]; the slabs of a tile's pixel splits are summed by wgrad_reduce_kernel as before.
//
// Reference semantics replaced: autograd's conv backward-weight inside metayolo/models/layers.py:31 (Conv), reached from train.py:472.
#include <stdio.h>

#include "common.h"
#include "hdyolo_internal.h"
#include "hdyolo.h"

namespace {

constexpr int NTHR = 512;
constexpr int UNIT = 16384, STAGE = 4 * UNIT, RING = 2 * STAGE;
constexpr int U_DLO = 0, U_DHI = UNIT, U_XLO = 2 * UNIT, U_XHI = 3 * UNIT;

struct WDArgs {
    const void* x; const void* dy; float* partial;      // partial: [splits][K][Q]
    int N, Hin, Win, C, ldx;
    int Ho, Wo, K, lddy;
    int stride, dh0, dw0, TW;                            // input pixel of (output pixel, tap): oi * stride + dh0 + th, oj * stride + dw0 + tw
    int Q, P;
    int ktiles, qtiles, splits, pix_per_split;
    unsigned mg_howo, mg_wo;
    int sh_howo, sh_wo;
    unsigned mg_c;                                       // tap of a column block: (q / C)
    int sh_c;
    unsigned mg_tw;
    int sh_tw;
};

__device__ __forceinline__ unsigned fdiv(unsigned n, unsigned mg, int sh) { return __umulhi(n << 1, mg) >> sh; }
__device__ __forceinline__ int fsw(int row) { return ((row >> 1) & 1) | (((row >> 3) & 1) << 1); }

__device__ __forceinline__ void lds_dma16(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, unsigned lds_byte) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (void __attribute__((address_space(3)))*)(uintptr_t)lds_byte, 16, (int)voff, (int)soff, 0, 0);
}

typedef int i32x2 __attribute__((ext_vector_type(2)));
union F16 {          // one 16x16x32 operand = two transposed 8-byte reads (pixels +0..3, +4..7 of the lane group's 8)
    i32x2 d[2];
    bf16x8 h;
};
#define WD_TR(dst, addr, off) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off) : "memory")
#define WD_VMCNT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")

__global__ __launch_bounds__(NTHR, 2) void wgrad_deep_kernel(const WDArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned lds0 = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) unsigned char*)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wr = wave >> 2, wc = wave & 3;
    const int fr = lane & 15, fq = lane >> 4;

    // workgroup -> (output tile, pixel split): the splits of one tile read different pixels, the tiles of one split the same ones:
    // consecutive logical ids = the tiles of one split (one XCD's L2 serves their common dy / x rows)
    const int tiles = p.ktiles * p.qtiles;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int split = bid / tiles, tile = bid - split * tiles;
    const int kt = tile / p.qtiles, qt = tile - kt * p.qtiles;
    const int k0 = kt * 256, q0 = qt * 256;
    const int pbeg = split * p.pix_per_split;
    const int pend = min(pbeg + p.pix_per_split, p.P);
    const int nst = (pend - pbeg + 63) >> 6;
    if (nst <= 0) return;

    // ------------------------------------------------------------------ loader: this thread's pixel row r0 of every stage, chunk c8 of every sub-tile
    constexpr unsigned OOB = 0x80000000u;
    const int r0 = tid >> 3;
    const int c8 = ((((tid & 7) >> 1) ^ fsw(r0)) << 1) | (tid & 1);       // logical 16-byte chunk that the 32-byte-block swizzle puts in slot (tid & 7)
    const int ldyB = p.lddy * 2, ldxB = p.ldx * 2;
    // dy: rows are consecutive pixels -> offset = (pixel - pbeg) * pitch + chunk; channel base k0 + 64 * sub as the scalar offset
    const __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc((void*)((const unsigned char*)p.dy + (size_t)pbeg * ldyB), 0, OOB, 0x00020000);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, OOB, 0x00020000);
    // per sub-tile of the x operand: tap (th, tw) and channel base are workgroup constants (C % 64 == 0); columns past Q: never valid
    int x_th[4], x_tw[4];
    unsigned x_soff[4];
    bool x_col[4], d_col[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int q = q0 + 64 * s;
        x_col[s] = q < p.Q;
        const int tap = (int)fdiv((unsigned)min(q, p.Q - 1), p.mg_c, p.sh_c);
        const int c = q - tap * p.C;
        x_th[s] = (int)fdiv((unsigned)tap, p.mg_tw, p.sh_tw);
        x_tw[s] = tap - x_th[s] * p.TW;
        x_soff[s] = (unsigned)((x_th[s] * p.Win + x_tw[s]) * ldxB + c * 2);
        d_col[s] = k0 + 64 * s < p.K;
    }
    int ld_st = 0, ld_buf = 0;               // loader position: stage, ring buffer
    bool ld_live = true;
    unsigned dy_off = 0, x_off = 0;          // this thread's row offsets of the loader's stage (bit 31: the row is past the split's end)
    int x_hi = 0, x_wi = 0;                  // window origin of that pixel in the input
    const int HoWo = p.Ho * p.Wo;
    auto loader_set_stage = [&](int st) {
        const int pix = pbeg + st * 64 + r0;
        const bool live = pix < pend;
        dy_off = (live ? 0u : OOB) | (unsigned)((pix - pbeg) * ldyB + c8 * 16);
        const int pc = min(pix, p.P - 1);
        const int n = (int)fdiv((unsigned)pc, p.mg_howo, p.sh_howo), rem = pc - n * HoWo;
        const int oi = (int)fdiv((unsigned)rem, p.mg_wo, p.sh_wo), oj = rem - oi * p.Wo;
        x_hi = oi * p.stride + p.dh0;
        x_wi = oj * p.stride + p.dw0;
        // the window origin may lie outside the image (negative): the offset is taken modulo 2^32 and the tap offset brings it back; the
        // border test below decides validity, so a wrapped offset is never dereferenced
        x_off = (live ? 0u : OOB) | ((unsigned)(((n * p.Hin + x_hi) * p.Win + x_wi) * ldxB + c8 * 16) & 0x7FFFFFFFu);
    };
    // x addresses: (row origin + tap offset + chunk) must stay below 2^31: checked on the host (x bytes < 2^31)
    auto issue_d = [&](int half) {           // dy unit: sub-tiles 2 * half, 2 * half + 1
        if (!ld_live) return;
        const unsigned dst = lds0 + (unsigned)(ld_buf * STAGE + (half ? U_DHI : U_DLO) + wave * 1024);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int s = 2 * half + i;
            lds_dma16(rdy, d_col[s] ? dy_off : OOB, (unsigned)((k0 + 64 * s) * 2), dst + 8192 * i);
        }
    };
    auto issue_x = [&](int half) {
        if (!ld_live) return;
        const unsigned dst = lds0 + (unsigned)(ld_buf * STAGE + (half ? U_XHI : U_XLO) + wave * 1024);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int s = 2 * half + i;
            const bool ok = x_col[s] && (unsigned)(x_hi + x_th[s]) < (unsigned)p.Hin && (unsigned)(x_wi + x_tw[s]) < (unsigned)p.Win;
            // origin + tap offset: computed in 32 bits; valid taps give the true (non-negative) offset
            const unsigned off = (x_off & OOB) | ((x_off + x_soff[s]) & 0x7FFFFFFFu);
            lds_dma16(rx, ok ? off : OOB, 0, dst + 8192 * i);
        }
    };
    auto loader_advance = [&]() {
        if (!ld_live) return;
        ld_buf ^= 1;
        if (++ld_st < nst) loader_set_stage(ld_st);
        else ld_live = false;
    };

    // ------------------------------------------------------------------ accumulators, fragments, read addresses
    f32x4 acc[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto AI = [](int a, int b, int m, int n) constexpr { return ((a * 2 + b) * 4 + m) * 2 + n; };
    F16 af[4][2], bf0[2][2], bf1[2][2];
    // lane (g = lane >> 4, q4 = (lane & 15) >> 2, p4 = lane & 3) of a k-step reads pixel rows 8g + q4 (and + 4), 8 bytes at 16-channel block cb,
    // byte p4 * 8; the block swizzle f = fsw(row) is the same for both rows and both k-steps
    const int q4 = fr >> 2, p4 = fr & 3, rrow = 8 * fq + q4, f = fsw(rrow);
    unsigned ra[4], rb[2];
#pragma unroll
    for (int m = 0; m < 4; ++m) ra[m] = lds0 + (unsigned)(wr * 8192 + rrow * 128 + ((m ^ f) << 5) + p4 * 8);
#pragma unroll
    for (int n = 0; n < 2; ++n) rb[n] = lds0 + (unsigned)((wc >> 1) * 8192 + rrow * 128 + (((2 * (wc & 1) + n) ^ f) << 5) + p4 * 8);
    unsigned cbuf = 0;

#define WD_READ_A(UOFF)                                                                      \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                         \
        _Pragma("unroll") for (int m = 0; m < 4; ++m) {                                      \
            WD_TR(af[m][ks].d[0], ra[m] + cbuf, (UOFF) + ks * 4096);                         \
            WD_TR(af[m][ks].d[1], ra[m] + cbuf, (UOFF) + ks * 4096 + 512);                   \
        }
#define WD_READ_B(BF, UOFF)                                                                  \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                         \
        _Pragma("unroll") for (int n = 0; n < 2; ++n) {                                      \
            WD_TR(BF[n][ks].d[0], rb[n] + cbuf, (UOFF) + ks * 4096);                         \
            WD_TR(BF[n][ks].d[1], rb[n] + cbuf, (UOFF) + ks * 4096 + 512);                   \
        }
#define WD_MFMA(A_, B_, BF)                                                                                                            \
    {                                                                                                                                  \
        __builtin_amdgcn_s_setprio(1);                                                                                                 \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                                               \
            _Pragma("unroll") for (int m = 0; m < 4; ++m)                                                                              \
                _Pragma("unroll") for (int n = 0; n < 2; ++n)                                                                          \
                    acc[AI(A_, B_, m, n)] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[m][ks].h, BF[n][ks].h, acc[AI(A_, B_, m, n)], 0, 0, 0); \
        __builtin_amdgcn_s_setprio(0);                                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                                                             \
    }
#define WD_WAIT(n)                                           \
    {                                                        \
        if (ld_live) { WD_VMCNT(n); } else { WD_VMCNT(0); }  \
    }
#define WD_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0);

    // ------------------------------------------------------------------ prologue: stage 0 and dy-lo, x-lo, x-hi of stage 1
    loader_set_stage(0);
    issue_d(0); issue_x(0); issue_x(1); issue_d(1); loader_advance();
    issue_d(0); issue_x(0); issue_x(1);
    if (ld_live) { WD_VMCNT(8); } else { WD_VMCNT(0); }
    __syncthreads();
    __builtin_amdgcn_s_barrier();
    WD_READ_B(bf0, U_XLO)
    WD_READ_A(U_DLO)
    WD_LGKM0()
    if (grp == 1) __builtin_amdgcn_s_barrier();           // waves 4-7 run one barrier behind waves 0-3 from here on

    for (int st = 0; st < nst; ++st) {
        // the schedule of conv_deep.hip's 256-wide loop: A = dy (k rows of the output), B = x (q columns), fragments one segment ahead
        const unsigned cnext = cbuf ^ (unsigned)STAGE;
        issue_d(1);
        loader_advance();
        WD_WAIT(8)
        WD_LGKM0()
        __builtin_amdgcn_s_barrier();
        WD_MFMA(0, 0, bf0)
        WD_READ_B(bf1, U_XHI)
        __builtin_amdgcn_s_barrier();
        // ---- phase 2
        issue_d(0);
        WD_LGKM0()
        __builtin_amdgcn_s_barrier();
        WD_MFMA(0, 1, bf1)
        WD_READ_A(U_DHI)
        __builtin_amdgcn_s_barrier();
        // ---- phase 3
        issue_x(0);
        WD_WAIT(8)
        WD_LGKM0()
        __builtin_amdgcn_s_barrier();
        WD_MFMA(1, 1, bf1)
        __builtin_amdgcn_s_barrier();
        // ---- phase 4
        issue_x(1);
        WD_WAIT(8)
        __builtin_amdgcn_s_barrier();
        WD_MFMA(1, 0, bf0)
        cbuf = cnext;
        WD_READ_B(bf0, U_XLO)
        WD_READ_A(U_DLO)
        __builtin_amdgcn_s_barrier();
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();           // re-align the two wave groups
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#undef WD_READ_A
#undef WD_READ_B
#undef WD_MFMA
#undef WD_WAIT
#undef WD_LGKM0

    // ------------------------------------------------------------------ slab tile: D rows = k (4 consecutive per lane), columns = q (16 lanes = 64 contiguous bytes)
    float* out = p.partial + (size_t)split * p.K * p.Q;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k = k0 + a * 128 + wr * 64 + m * 16 + fq * 4 + r;
                if (k >= p.K) continue;
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int n = 0; n < 2; ++n) {
                        const int q = q0 + b * 128 + wc * 32 + n * 16 + fr;
                        if (q < p.Q) out[(size_t)k * p.Q + q] = acc[AI(a, b, m, n)][r];
                    }
            }
}

// pixel splits of the deep kernel: one workgroup per CU over (tiles x splits)
inline void deep_plan(int K, int Q, long long P, int* ktiles, int* qtiles, int* splits, int* pps) {
    *ktiles = cdiv(K, 256);
    *qtiles = cdiv(Q, 256);
    const int tiles = *ktiles * *qtiles;
    int s = 256 / tiles;
    if (s < 1) s = 1;
    const int maxs = cdiv(P, 512);                 // at least 8 stages per workgroup
    if (s > maxs) s = maxs;
    if (s < 1) s = 1;
    *pps = round_up(cdiv(P, s), 64);
    *splits = cdiv(P, *pps);
}

bool deep_ok(int N, int Hin, int Win, int Ho, int Wo, int C, int K, int R, int S, int stride, int ldx, int lddy, int dtype) {
    if (hdy_opt(HDY_OPT_NO_WGRAD_DEEP) || dtype != HDY_BF16) return false;
    // K: whole 64-channel sub-tiles (the loader's d_col[] test and the slab store's k < K test cover a last, partly filled 256-row tile); round 6: 192 and
    // up (yolov5m's 192-wide layers: K = 192 fills three quarters of one tile), before that K % 128 == 0 && K >= 256
    const int kmin = hdy_opt(HDY_OPT_WGRAD_DEEP_KMIN);
    if (C % 64 != 0 || K % 64 != 0 || K < kmin || ldx % 8 != 0 || lddy % 8 != 0) return false;
    // measured on yolov5s (B = 64): the multi-tap layers 158 -> 85 us (256x512 3x3/s2 @40x40), 158 -> 82 (128x256 @80x80), 96 -> 58 (256x256 s2); the
    // 1x1 layers at 20x20 / 40x40 lose (20 -> 28 us, 35 -> 39 us: a 256 x 256 tile leaves them 4-8 tiles, i.e. 32-64 pixel splits of 7-25 stages,
    // and twice the slab bytes): multi-tap layers only
    if (R * S == 1) return false;
    const long long P = (long long)N * Ho * Wo;
    if (P < 8192) return false;                                         // too few pixels to stream
    if ((long long)N * Hin * Win * ldx * 2 >= (1LL << 31) - (1LL << 24)) return false;    // 31-bit x offsets (bit 31 = out of range)
    if (P * lddy * 2 >= (1LL << 31)) return false;
    return true;
}

}  // namespace

// workspace bytes of the deep weight gradient for this shape; 0: not its shape (the caller's generic path applies)
size_t hdy_wgrad_deep_workspace_bytes(int N, int Hin, int Win, int Ho, int Wo, int C, int K, int R, int S, int stride, int dtype) {
    if (!deep_ok(N, Hin, Win, Ho, Wo, C, K, R, S, stride, 8, 8, dtype)) return 0;
    int kt, qt, sp, pps;
    deep_plan(K, R * S * C, (long long)N * Ho * Wo, &kt, &qt, &sp, &pps);
    return (size_t)sp * K * R * S * C * sizeof(float);
}

// launches the deep kernel when the shape is its; *splits = slabs written ([split][K][Q] in `partial`)
int hdy_wgrad_deep_try(const void* x, int ldx, const void* dy, int lddy, int N, int Hin, int Win, int Ho, int Wo, int C, int K, int R, int S, int stride,
                       int pad, float* partial, int dtype, hipStream_t st, int* splits, int* rc) {
    if (!deep_ok(N, Hin, Win, Ho, Wo, C, K, R, S, stride, ldx, lddy, dtype)) return 0;
    if ((((uintptr_t)x | (uintptr_t)dy) & 15) != 0) return 0;
    WDArgs a = {};
    a.x = x; a.dy = dy; a.partial = partial;
    a.N = N; a.Hin = Hin; a.Win = Win; a.C = C; a.ldx = ldx; a.Ho = Ho; a.Wo = Wo; a.K = K; a.lddy = lddy;
    a.stride = stride; a.dh0 = -pad; a.dw0 = -pad; a.TW = S;
    a.Q = R * S * C;
    a.P = N * Ho * Wo;
    deep_plan(K, a.Q, a.P, &a.ktiles, &a.qtiles, &a.splits, &a.pix_per_split);
    hdy_magic((unsigned)(Ho * Wo), &a.mg_howo, &a.sh_howo);
    hdy_magic((unsigned)Wo, &a.mg_wo, &a.sh_wo);
    hdy_magic((unsigned)C, &a.mg_c, &a.sh_c);
    hdy_magic((unsigned)S, &a.mg_tw, &a.sh_tw);
    const int grid = a.ktiles * a.qtiles * a.splits;
    static PerDeviceOnce attr_once;
    attr_once.run([&] {
        (void)hipFuncSetAttribute((const void*)wgrad_deep_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, RING);
    });
    hdy_note_dispatch("wgrad_deep");
    hipLaunchKernelGGL(wgrad_deep_kernel, dim3(grid), dim3(NTHR), RING, st, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        hdy_set_error("wgrad_deep: launch failed: %s", hipGetErrorString(e));
        *rc = (int)e;
        return 1;
    }
    *splits = a.splits;
    *rc = HDY_OK;
    return 1;
}

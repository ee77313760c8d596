// 3x3 / stride 1 / pad 1 convolution for 128 input channels (bf16), filter resident in REGISTERS: conv3x3.hip's design carried to the 128 -> 128
// bottleneck layers (yolov5s' @40x40 and their data gradients, yolov5l's @128x128 at inference) — round 6.
//
// Why.  On the deep-pipelined implicit GEMM (conv_deep.hip) the 128 -> 128 layer at 40 x 40, B = 64 is 400 row tiles of 18 K-tiles on 256 workgroups:
// 33 us of K-tiles (7 of them the half-empty second round), 6 us of tile boundaries and 6-9 us of launch + prologue + tail = 46-48 us for 30.2 GFLOP,
// 0.25 of the MFMA peak, where the 64 -> 64 layer holds 0.35 with its filter in registers (profiles/r06_deep_mfma_ab.txt: the MFMA shape is not the reason).
// The filter of 128 x 128 x 9 bf16 is 288 KB — more than half of a CU's register file — so a workgroup here owns HALF of the output channels:
//   * 4 waves, wave = 16 output channels of the workgroup's 64: their 9 x 128 filter slice is 36 MFMA row operands = 144 VGPRs for the wave's lifetime;
//   * items are 8 x 8 pixel tiles (40 = 5 x 8: the 16-wide tiles of conv3x3.hip do not divide it): a (8 + 2) x (8 + 2) patch of 256-byte pixel rows (25 KB) by
//     LDS-DMA, double buffered across items; an MFMA pixel tile is TWO image rows of 8 pixels, so the fragment of patch row pair q at column shift s feeds
//     (pixel tile q / 2, filter row 0) and (q / 2 - 1, filter row 2) for even q, ((q - 1) / 2, filter row 1) for odd q: 9 x 3 x 4 = 108 fragment reads for
//     144 MFMAs per wave and item;
//   * 2 patches + an 8 KB staging tile = 58 KB of LDS and 256 VGPRs: two workgroups per CU (2 waves per SIMD), independent, as in conv3x3.hip.  Workgroups
//     2 g and 2 g + 1 walk the SAME tiles for the two channel halves (neighbours on one XCD: the second patch fetch hits its L2);
//   * 256-byte rows cover all 64 banks once, so every lane group of a ds_read_b128 needs 16 different 16-byte chunks: chunk ^ ((patch column & 7) << 1) —
//     the key touches chunk bits 1-3 and leaves bit 0, which tells apart the two k-chunks a lane group mixes; conflict-free for the three column shifts
//     (exhaustive search, tests/test_abi.py::test_c128_swizzle_is_conflict_free restates it on the host).  LDS-DMA writes linearly: the XOR is on the source.
// The MFMAs run with the filter as the row operand: a lane's 4 accumulator values are 4 consecutive output channels of one pixel; the epilogue (8-byte
// staging writes keyed by the row, 16-byte row stores of 128 contiguous bytes per pixel, residual / accumulate on the way out, BatchNorm sums in registers
// across all of a workgroup's items, one slab per workgroup pair) is conv3x3.hip's.
//
// Requirements (checked by the launcher, otherwise the deep-pipelined / generic kernel runs): bf16 in / out, C == 128, K <= 128 and K % 8 == 0, R = S = 3,
// stride 1, pad 1, H % 8 == 0, W % 8 == 0, 16-byte aligned rows.
//
// Reference semantics replaced: nn.Conv2d(k = 3, s = 1, p = 1) inside metayolo/models/layers.py:92-93 (Bottleneck.cv2) and its autograd backward-data
// (train.py:472).
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "hdyolo_internal.h"

__device__ uint4 g_hdy_zero16_c128[4];   // zero page for out-of-image patch pixels

namespace {

__device__ __forceinline__ void glds16(const void* g, unsigned char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)g,
                                     (void __attribute__((address_space(3)))*)lds_wave_base, 16, 0, 0);
}

constexpr int NTHR = 256;
constexpr int C = 128, CB = C * 2, CPP = C / 8, KS = C / 32;          // 256-byte pixel rows, 16 chunks, 4 k-steps per tap
constexpr int TH = 8, TW = 8, PW = TW + 2, PH = TH + 2, PPIX = PH * PW;   // 100 patch pixels
constexpr int MT = TH * TW / 16;                                       // 4 MFMA pixel tiles (two image rows each)
constexpr int KH = 64;                                                 // output channels per workgroup
constexpr int PATCH_B = PPIX * CB;                                     // 25600
constexpr int STAGE_B = TH * TW * KH * 2;                              // 8192
constexpr int SMEM_B = 2 * PATCH_B + STAGE_B;                          // 59392: two workgroups per CU
constexpr int NPASS = (PPIX * CPP + NTHR - 1) / NTHR;                  // 7 loader passes (the last one a quarter full)
static_assert(NTHR % CPP == 0, "a thread's chunk slot is the same in every pass");

// EPI: 0 = raw convolution out (train-mode forward, dgrad), 1 = scale/shift, 2 = scale/shift + SiLU; STATS: BatchNorm partial sums
template <int EPI, bool STATS>
__global__ __launch_bounds__(NTHR, 2) void conv3x3_c128_kernel(const ConvArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sP = smem;                    // [2][100][256 B]
    unsigned char* sS = smem + 2 * PATCH_B;      // [64][128 B] staging tile

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int tiles_w = p.Wo / TW, tiles_h = p.Ho / TH;
    const int per_img = tiles_w * tiles_h;
    const int tiles_total = p.N * per_img;
    const int nkh = (p.K + KH - 1) / KH;                                 // channel halves (1 or 2)
    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int kh = wg % nkh, strm = wg / nkh, nstrm = (int)gridDim.x / nkh;     // grid % nkh == 0 (launcher)
    const int k0 = kh * KH;
    const int nitems = strm < tiles_total ? (tiles_total - strm + nstrm - 1) / nstrm : 0;

    const bf16_t* __restrict__ x = (const bf16_t*)p.x;
    const bf16_t* __restrict__ w = (const bf16_t*)p.w;
    const unsigned char* zero = (const unsigned char*)g_hdy_zero16_c128;

    // ---- filter slice -> registers: row operand of tap t, k-step ks = w[k0 + wave*16 + fr][t*128 + (ks*4 + fq)*8 .. +7]
    V16 bw[9][KS];
    {
        const int k = k0 + wave * 16 + fr;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const void* src = (k < p.K) ? (const void*)(w + (size_t)k * p.Kdp + t * C + (ks * 4 + fq) * 8) : (const void*)zero;
                bw[t][ks].i = *(const i32x4*)src;
            }
    }

    // ---- patch loader: a thread owns chunk slot (tid & 15) of patch pixels (tid >> 4) + 16 i; per item only the origin and the border test change
    // (registers are what this kernel is short of — 144 hold the filter: per pass only the packed (row, column) stays, the offset is recomputed per item)
    int pyx[NPASS];
#pragma unroll
    for (int i = 0; i < NPASS; ++i) {
        const int pix = (tid >> 4) + 16 * i;
        const int py = pix / PW, px = pix - py * PW;
        pyx[i] = py | (px << 8);
    }
    auto item_tile = [&](int idx, int& n, int& th, int& tw) {
        const int t = strm + idx * nstrm;
        n = t / per_img;
        const int rem = t - n * per_img;
        th = rem / tiles_w;
        tw = rem - th * tiles_w;
    };
    auto issue_patch = [&](int idx, int buf) {
        int n, th, tw;
        item_tile(idx, n, th, tw);
        const int h0 = th * TH - 1, w0 = tw * TW - 1;
        const bf16_t* org = x + (((long long)n * p.Hin + h0) * p.Win + w0) * p.ldx;
        unsigned char* dst = sP + buf * PATCH_B;
#pragma unroll
        for (int i = 0; i < NPASS; ++i) {
            const int py = pyx[i] & 255;
            if ((wave * 64 + NTHR * i) / CPP >= PPIX) break;              // wave-uniform: whole 1 KB pieces past the patch
            if (py >= PH) continue;                                       // lanes past the last pixel: no LDS write
            const int px = pyx[i] >> 8;
            const int h = h0 + py, ww = w0 + px;
            const int lcp = (tid & 15) ^ ((px & 7) << 1);                 // logical chunk that the swizzle places in this thread's physical slot
            const void* src = ((unsigned)h < (unsigned)p.Hin && (unsigned)ww < (unsigned)p.Win) ? (const void*)(org + ((py * p.Win + px) * p.ldx + lcp * 8)) : (const void*)zero;
            glds16(src, dst + (wave * 64 + NTHR * i) * 16);
        }
    };

    // fragment of (row pair q, shift s, k-step ks): lane (fr, fq) reads patch pixel (q + (fr >> 3), (fr & 7) + s), chunk (ks*4 + fq) ^ key(column)
    // (LDS byte addresses of patch buffer 0; moved to the other buffer and back by +- PATCH_B at the end of every item: one set of 12 registers)
    unsigned abase[3][KS];
    {
        const unsigned p0 = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) unsigned char*)sP;
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int px = (fr & 7) + s;
                abase[s][ks] = p0 + (unsigned)(((fr >> 3) * PW + px) * CB + (((ks * 4 + fq) ^ ((px & 7) << 1)) << 4));
            }
    }

    float sc[4], sh[4], s1[4], s2[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int c = k0 + wave * 16 + fq * 4 + r;
        sc[r] = (p.scale && c < p.K) ? p.scale[c] : 1.0f;
        sh[r] = (p.shift && c < p.K) ? p.shift[c] : 0.0f;
        s1[r] = 0.f;
        s2[r] = 0.f;
    }
    // store phase: thread = (16-byte chunk st_ch of the 128-byte staging row, rows st_rr and st_rr + 32); staging row = MFMA pixel tile * 16 + fr
    const int st_ch = tid & 7, st_rr = tid >> 3;
    const int st_lds = st_rr * 128 + ((st_ch ^ ((st_rr >> 1) & 7)) << 4);
    auto row_pixel = [&](int row) { return (2 * (row >> 4) + ((row >> 3) & 1)) * p.Wo + (row & 7); };      // offset of staging row's pixel from the tile origin
    const long long st_off0 = (long long)row_pixel(st_rr), st_off1 = (long long)row_pixel(st_rr + 32);
    // staging write: 8-byte slot (wave * 4 + fq) of pixel row fr, keyed by the row (16 consecutive lanes of a ds_write_b64 = the 16 rows fr of one slot)
    const int ep_off = fr * 128 + (((wave * 4 + fq) ^ (fr & 15)) << 3);

    if (nitems > 0) issue_patch(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int cur = 0;
    for (int idx = 0; idx < nitems; ++idx) {
        if (idx + 1 < nitems) issue_patch(idx + 1, cur ^ 1);
        f32x4 acc[MT];
#pragma unroll
        for (int a = 0; a < MT; ++a) acc[a] = f32x4{0.f, 0.f, 0.f, 0.f};
        // software pipeline by hand (conv3x3.hip): group g = (row pair g / KS, k-step g % KS) = 3 fragments; the next group's reads are issued before this
        // group's MFMAs, inline asm with counted lgkmcnt waits
        V16 fa[2][3];
#define C128_LOAD(G, F)                                                                                                           \
    {                                                                                                                             \
        _Pragma("unroll") for (int s_ = 0; s_ < 3; ++s_) {                                                                        \
            i32x4 v_;                                                                                                             \
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v_) : "v"(abase[s_][(G) % KS]), "n"(((G) / KS) * PW * CB));     \
            (F)[s_].i = v_;                                                                                                       \
        }                                                                                                                         \
    }
        C128_LOAD(0, fa[0])
#pragma unroll
        for (int g = 0; g < KS * (2 * MT + 1); ++g) {            // row pairs q = 0 .. 2 MT
            const int q = g / KS, ks = g % KS;
            if (g + 1 < KS * (2 * MT + 1)) {
                C128_LOAD(g + 1, fa[(g + 1) & 1])
                asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(fa[g & 1][0].i), "+v"(fa[g & 1][1].i), "+v"(fa[g & 1][2].i));
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[g & 1][0].i), "+v"(fa[g & 1][1].i), "+v"(fa[g & 1][2].i));
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < 3; ++s)
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const int a2 = q - r;                          // = 2 * pixel tile
                    if (a2 >= 0 && (a2 & 1) == 0 && a2 / 2 < MT) acc[a2 / 2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bw[r * 3 + s][ks].h, fa[g & 1][s].h, acc[a2 / 2], 0, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0);
        }
#undef C128_LOAD
        if (STATS) {
#pragma unroll
            for (int a = 0; a < MT; ++a)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = acc[a][r];
                    s1[r] += v;
                    s2[r] = __builtin_fmaf(v, v, s2[r]);
                }
        }
        // the staging tile is separate from the patches: no barrier between a wave's last MFMA and its staging writes
#pragma unroll
        for (int a = 0; a < MT; ++a) {
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[r] = acc[a][r];
                if (EPI >= 1) v[r] = v[r] * sc[r] + sh[r];
                if (EPI == 2) v[r] = v[r] * __builtin_amdgcn_rcpf(1.0f + __expf(-v[r]));
            }
            bf16x4 o = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
            *(bf16x4*)(sS + ep_off + a * 16 * 128) = o;
        }
        __syncthreads();                                   // staging complete; every wave is done with patch `cur`
        if (k0 + st_ch * 8 < p.K) {
            int n, th, tw;
            item_tile(idx, n, th, tw);
            const long long org = ((long long)n * p.Ho + th * TH) * p.Wo + tw * TW;
#pragma unroll
            for (int j = 0; j < 2; ++j) {                  // rows st_rr and st_rr + 32 of the 64 staged pixels
                const long long opix = org + (j ? st_off1 : st_off0);
                V16 v;
                v.i = *(const i32x4*)(sS + st_lds + j * 32 * 128);
                if (st_rr & 1) v.i = i32x4{v.i[2], v.i[3], v.i[0], v.i[1]};
                bf16_t* dst = (bf16_t*)p.y + opix * p.ldy + k0 + st_ch * 8;
                if (p.res || p.accumulate) {
                    float f[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] = (float)v.h[e];
                    if (p.res) {
                        V16 q;
                        q.i = *(const i32x4*)((const bf16_t*)p.res + opix * p.ldr + k0 + st_ch * 8);
#pragma unroll
                        for (int e = 0; e < 8; ++e) f[e] += (float)q.h[e];
                    }
                    if (p.accumulate) {
                        V16 q;
                        q.i = *(const i32x4*)dst;
#pragma unroll
                        for (int e = 0; e < 8; ++e) f[e] += (float)q.h[e];
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) v.h[e] = (bf16_t)f[e];
                }
                *(i32x4*)dst = v.i;
            }
        }
        // the next patch's DMA precedes these two stores in the wave's vm queue: wait for it, not for the stores
        asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        __syncthreads();                                   // next patch landed for everyone; staging tile free again
        {
            const unsigned step = cur ? (unsigned)-PATCH_B : (unsigned)PATCH_B;
#pragma unroll
            for (int s = 0; s < 3; ++s)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) abase[s][ks] += step;
        }
        cur ^= 1;
    }

    if (STATS) {                                           // a wave's channels are its own: 16 pixel lanes -> one value, no LDS; slab = the workgroup pair's stream
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float u = s1[r], q = s2[r];
            u = row_sum16(u);
            q = row_sum16(q);
            const int c = k0 + wave * 16 + fq * 4 + r;
            if (fr == 0 && c < p.K) {
                p.stats[((size_t)strm * 2 + 0) * p.K + c] = u;
                p.stats[((size_t)strm * 2 + 1) * p.K + c] = q;
            }
        }
    }
}

template <int EPI, bool STATS>
static void launch_c128(const ConvArgs& a, int grid, hipStream_t st) {
    static PerDeviceOnce attr_once;
    attr_once.run([&] {
        (void)hipFuncSetAttribute((const void*)conv3x3_c128_kernel<EPI, STATS>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_B);
    });
    hdy_note_dispatch("conv3x3_c128");
    hipLaunchKernelGGL((conv3x3_c128_kernel<EPI, STATS>), dim3(grid), dim3(NTHR), SMEM_B, st, a);
}

}  // namespace

static bool c128_shape_ok(int Cin, int K, int R, int S, int stride, int pad, int H, int W, int dtype) {
    return !hdy_opt(HDY_OPT_NO_CONV3X3_C128) && dtype == HDY_BF16 && R == 3 && S == 3 && stride == 1 && pad == 1 && Cin == C && K <= 2 * KH && K % 8 == 0 &&
           H % TH == 0 && W % TW == 0;
}

// workgroups: two resident per CU; a multiple of the channel halves; never more than (tiles x halves)
static int c128_grid(long long tiles, int K) {
    const int nkh = (K + KH - 1) / KH;
    long long g = 512 / nkh * nkh;
    if (g > tiles * nkh) g = tiles * nkh;
    return (int)g;
}

// statistic slabs this kernel writes (one per workgroup stream = grid / channel halves); 0 = not its shape
int hdy_conv3x3_c128_slabs(int N, int H, int W, int Cin, int K, int R, int S, int stride, int pad, int dtype) {
    if (!c128_shape_ok(Cin, K, R, S, stride, pad, H, W, dtype)) return 0;
    const int nkh = (K + KH - 1) / KH;
    return c128_grid((long long)N * (H / TH) * (W / TW), K) / nkh;
}

// Returns 1 and launches when the shape qualifies; 0 = not eligible (the caller goes on to the deep-pipelined / generic kernel).
int hdy_conv3x3_c128_try(const ConvArgs& a, int dtype, int out_f32, hipStream_t st, int* rc) {
    if (dtype != HDY_BF16 || out_f32 || a.act > 1) return 0;
    if (!(a.TH == 3 && a.TW == 3 && a.ih_mul == 1 && a.iw_mul == 1 && a.dh0 == -1 && a.dw0 == -1 && a.dense_out && !a.span_pixels)) return 0;
    if (!(a.Hin == a.Ho && a.Win == a.Wo && c128_shape_ok(a.C, a.K, 3, 3, 1, 1, a.Ho, a.Wo, dtype))) return 0;
    const bool aligned = a.ldx % 8 == 0 && a.ldy % 8 == 0 && ((uintptr_t)a.y & 15) == 0 && ((uintptr_t)a.x & 15) == 0 &&
                         (!a.res || (a.ldr % 8 == 0 && ((uintptr_t)a.res & 15) == 0));
    if (!aligned) {
        if (!a.stats) return 0;
        hdy_set_error("conv3x3_c128: statistics requested but x/y/res rows are not 16-byte aligned (ldx=%d ldy=%d)", a.ldx, a.ldy);
        *rc = HDY_EINVAL;
        return 1;
    }
    const int nkh = (a.K + KH - 1) / KH;
    const int grid = c128_grid((long long)a.N * (a.Ho / TH) * (a.Wo / TW), a.K);
    HDY_STAT_CAP(a, grid / nkh, "conv3x3_c128")
    const int epi = a.act == 1 ? 2 : ((a.scale || a.shift) ? 1 : 0);
    if (a.stats) {
        if (epi == 2) launch_c128<2, true>(a, grid, st);
        else if (epi == 1) launch_c128<1, true>(a, grid, st);
        else launch_c128<0, true>(a, grid, st);
    } else {
        if (epi == 2) launch_c128<2, false>(a, grid, st);
        else if (epi == 1) launch_c128<1, false>(a, grid, st);
        else launch_c128<0, false>(a, grid, st);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        hdy_set_error("conv3x3_c128: launch failed: %s", hipGetErrorString(e));
        *rc = (int)e;
        return 1;
    }
    *rc = HDY_OK;
    return 1;
}

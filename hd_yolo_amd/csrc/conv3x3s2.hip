// 3x3 / stride 2 / pad 1 convolution for 32 input channels and up to 64 output channels (bf16), patch-resident with the filter in
// registers: yolov5s' second layer (32 -> 64, 320x320 -> 160x160 at 640x640 tiles), which the generic implicit GEMM served at 0.4 of its
// HBM bound — 64-byte input pixels fetched once per tap (nine 64-byte segments per output pixel).
// Round 5: template <NP, NWV> — the input has 32 * NP channels, kept as NP separate 32-channel PLANES (each a patch laid out exactly as below, so
// loader, swizzle and fragment reads do not change), NWV waves = 16 * NWV output channels.  <1, 4> is the kernel described here; <2, 8> takes
// yolov5s' third down-sampling layer (64 -> 128, 160x160 -> 80x80), which the deep-pipelined kernel served at 0.39 of its HBM bound.
//
// Same structure as conv3x3.hip (the stride-1 kernel), with what stride 2 changes:
//   * a workgroup (4 waves, each 16 output channels with its 9 x 32 filter slice as nine MFMA row operands = 36 VGPRs) takes 4 x 16 output
//     pixels; their 9 x 33-pixel input patch (19 KB) arrives by LDS-DMA, double buffered across tiles; 47 KB of LDS per workgroup ->
//     three workgroups per CU;
//   * an MFMA column = 16 consecutive OUTPUT pixels = every other patch pixel.  The patch is stored as rows of 128 bytes holding a PAIR of
//     64-byte pixels; the 8 16-byte slots of pair P are XORed with (P >> 1) & 7, so that the 16 lanes of a fragment read — pairs P .. P+15, one
//     pixel of each — hit 16 different slots of the 256-byte bank row ((P & 1) picks the half, (P >> 1) & 7 moves the slot).  LDS-DMA writes
//     linearly, so the permutation is applied to the source address (linear destination + permuted source + permuted read);
//   * patch row q serves output row a = q / 2 with filter row 0 and a - 1 with filter row 2 (q even), or a = (q - 1) / 2 with filter row 1.
// Epilogue, BatchNorm sums (one slab per workgroup) and the store phase are conv3x3.hip's.
//
// Requirements (checked by the launcher, otherwise the generic kernel runs): bf16 in / out, C == 32, K <= 64 and a multiple of 8, R = S = 3,
// stride 2, pad 1, even H and W, Ho % 4 == 0, Wo % 16 == 0, 16-byte aligned rows.
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "hdyolo_internal.h"

__device__ uint4 g_hdy_zero16_s2[4];   // zero page for out-of-image patch pixels

namespace {

constexpr int TH = 4, TW = 16;
constexpr int PROWS = 2 * TH + 1, PPAIR = TW + 1;                    // 9 patch rows of 17 pixel pairs (34 pixels, 33 used)
constexpr int NPAIR = PROWS * PPAIR;                                 // 153 rows of 128 bytes
constexpr int PLANE_B = NPAIR * 128;                                 // 19584: one 32-channel plane of the patch
constexpr int C = 32;                                                // channels per plane

template <int NP, int NWV> struct Geo {
    static constexpr int NTHR = 64 * NWV;                            // wave = 16-channel group
    static constexpr int PATCH_B = NP * PLANE_B;
    static constexpr int SROW = NWV * 32;                            // bytes per staged output pixel (16 channels per wave)
    static constexpr int STAGE_B = TH * TW * SROW;                   // 8192 / 16384
    static constexpr int SMEM_B = 2 * PATCH_B + STAGE_B;             // <1, 4>: 47 360 (three workgroups per CU); <2, 8>: 94 720 (one)
    static constexpr int NPASS = (NPAIR * 8 + NTHR - 1) / NTHR;      // loader passes per plane (last partial)
    static constexpr int CPRW = SROW / 16;                           // 16-byte chunks per staged row
    static_assert(NTHR / CPRW == 32, "32 staged rows per store pass");
};

__device__ __forceinline__ void glds16(const void* g, unsigned char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)g, (void __attribute__((address_space(3)))*)lds_wave_base, 16, 0, 0);
}

// EPI: 0 = raw convolution out (train-mode forward), 1 = scale/shift, 2 = scale/shift + SiLU;  STATS: BatchNorm partial sums (one slab per workgroup)
// LDS fragment / staging reads as inline asm and raw barriers (round 5; see conv_dgrad_s2.hip): hipcc puts `s_waitcnt vmcnt(0)` in front of every
// compiler-visible LDS read that may alias a pending LDS-DMA and into `__syncthreads()` — the next patch's DMA was waited for at the tile's first fragment
// read and the tile's output stores at the barrier behind them.  The waits that are needed are counted by hand.
// HAND (= NP == 2) selects that form; the one-plane instance keeps compiler-visible reads and `__syncthreads()`: three of its workgroups share a CU and
// cover each other's waits, and the hand-counted form measured slower there (133 against 124 us).
#define S2_LDSR(dst, addr)                                                                             \
    do {                                                                                               \
        if constexpr (HAND) asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(addr) : "memory");    \
        else dst = *(const i32x4*)(smem + ((addr) - lds0));                                            \
    } while (0)
#define S2_LGKM(n)                                                                                                       \
    do {                                                                                                                 \
        if constexpr (HAND) { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(n) : "memory"); __builtin_amdgcn_sched_barrier(0); } \
    } while (0)
#define S2_BARRIER(lg)                                                                                                   \
    do {                                                                                                                 \
        if constexpr (HAND) { if (lg) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } \
        else __syncthreads();                                                                                            \
    } while (0)

template <int NP, int NWV, int EPI, bool STATS>
__global__ __launch_bounds__(64 * NWV, NWV == 4 ? 3 : 2) void conv3x3s2_kernel(const ConvArgs p) {
    using G = Geo<NP, NWV>;
    constexpr int NTHR = G::NTHR, PATCH_B = G::PATCH_B, NPASS = G::NPASS, SROW = G::SROW, CPRW = G::CPRW, KD = 32 * NP;
    constexpr bool HAND = NP == 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sP = smem;                    // [2][NP][153][128 B]
    unsigned char* sS = smem + 2 * PATCH_B;      // [64][SROW] staging tile

    const unsigned lds0 = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) unsigned char*)smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int tiles_w = p.Wo / TW, tiles_h = p.Ho / TH, per_img = tiles_w * tiles_h;
    const int tiles = p.N * per_img;
    const int wg = xcd_remap(blockIdx.x, gridDim.x);

    const bf16_t* __restrict__ x = (const bf16_t*)p.x;
    const bf16_t* __restrict__ w = (const bf16_t*)p.w;
    const unsigned char* zero = (const unsigned char*)g_hdy_zero16_s2;

    // ---- filter slice -> registers: row operand of (tap t, plane pl) = w[wave*16 + fr][t*KD + pl*32 + fq*8 .. +7]
    V16 bw[9][NP];
    {
        const int k = wave * 16 + fr;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) {
                const void* src = (k < p.K) ? (const void*)(w + (size_t)k * p.Kdp + t * KD + pl * C + fq * 8) : (const void*)zero;
                bw[t][pl].i = *(const i32x4*)src;
            }
    }

    auto issue_patch = [&](int t, int buf) {
        int tv = tid;                                 // opaque copy: keeps the per-pass address arithmetic inside the call instead of hoisted (and spilled) per-pass registers
        if constexpr (HAND) asm volatile("" : "+v"(tv));
        const int slot = tv & 7, wv = tv >> 6;
        const int n = t / per_img, rem = t - n * per_img;
        const int th = rem / tiles_w, tw = rem - th * tiles_w;
        const int h0 = 2 * th * TH - 1, w0 = 2 * tw * TW - 1;
        const bf16_t* org = x + (((long long)n * p.Hin + h0) * p.Win + w0) * p.ldx;      // patch pixel (0, 0); may lie outside the image
        unsigned char* dst = sP + buf * PATCH_B;
#pragma unroll
        for (int pl = 0; pl < NP; ++pl)
#pragma unroll
            for (int i = 0; i < NPASS; ++i) {
                if ((wave * 64 + NTHR * i) / 8 >= NPAIR) break;                // wave-uniform: whole 1 KB pieces past the patch
                const int pr = (tv + NTHR * i) >> 3;                           // pair row 0..152
                if (pr >= NPAIR) continue;
                const int py = (pr * 3856) >> 16, P = pr - py * PPAIR;          // pr / 17 for pr < 153
                const int L = slot ^ ((P >> 1) & 7);                           // logical 16-byte piece that belongs in this physical slot
                const int px = 2 * P + (L >> 2), h = h0 + py, ww = w0 + px;
                const void* src = ((unsigned)h < (unsigned)p.Hin && (unsigned)ww < (unsigned)p.Win && px < 2 * TW + 1)
                                      ? (const void*)(org + ((long long)py * p.Win + px) * p.ldx + pl * C + (L & 3) * 8) : (const void*)zero;
                glds16(src, dst + pl * PLANE_B + (wv * 64 + NTHR * i) * 16);
            }
    };

    int aoff[3];                                  // fragment byte offset inside a patch row for column tap s: pixel 2*fr + s, 16-byte piece fq
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        const int px = 2 * fr + s, P = px >> 1;
        aoff[s] = P * 128 + ((((px & 1) * 4 + fq) ^ ((P >> 1) & 7)) << 4);
    }

    float sc[4], sh[4], s1[4], s2[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int c = wave * 16 + fq * 4 + r;
        sc[r] = (EPI >= 1 && p.scale && c < p.K) ? p.scale[c] : 1.0f;
        sh[r] = (EPI >= 1 && p.shift && c < p.K) ? p.shift[c] : 0.0f;
        s1[r] = 0.f;
        s2[r] = 0.f;
    }
    const int st_ch = tid % CPRW, st_rr = tid / CPRW;                         // store phase: chunk, first row (rows + 32 j)
    const int st_lds = st_rr * SROW + ((st_ch ^ ((st_rr >> 1) & 7)) << 4);
    const long long st_off = ((long long)(st_rr >> 4) * p.Wo + (st_rr & 15)) * p.ldy + st_ch * 8;
    const long long st_step = (long long)2 * p.Wo * p.ldy;
    const long long rs_off = ((long long)(st_rr >> 4) * p.Wo + (st_rr & 15)) * p.ldr + st_ch * 8;
    const long long rs_step = (long long)2 * p.Wo * p.ldr;
    const int ep_off = fr * SROW + (((wave * 4 + fq) ^ (fr & 15)) << 3);      // key fr & 15: 16 rows -> 16 slots (see conv3x3.hip); odd rows swap a chunk's halves

    auto compute = [&](int t, int cur) {
        f32x4 acc[TH];
#pragma unroll
        for (int a = 0; a < TH; ++a) acc[a] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) {
            const unsigned pb = lds0 + cur * PATCH_B + pl * PLANE_B;
            V16 f[2][3];
#pragma unroll
            for (int s = 0; s < 3; ++s) S2_LDSR(f[0][s].i, pb + aoff[s]);
#pragma unroll
            for (int q = 0; q < PROWS; ++q) {
                if (q + 1 < PROWS) {
#pragma unroll
                    for (int s = 0; s < 3; ++s) S2_LDSR(f[(q + 1) & 1][s].i, pb + (q + 1) * PPAIR * 128 + aoff[s]);
                    S2_LGKM(3);                                   // this row's three fragments are back, the next row's are in flight
                } else {
                    S2_LGKM(0);
                }
#pragma unroll
                for (int s = 0; s < 3; ++s) {
                    const bf16x8 v = f[q & 1][s].h;
                    if (q & 1) {
                        acc[q >> 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bw[3 + s][pl].h, v, acc[q >> 1], 0, 0, 0);                      // filter row 1
                    } else {
                        if ((q >> 1) < TH) acc[q >> 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bw[s][pl].h, v, acc[q >> 1], 0, 0, 0);        // filter row 0
                        if (q >= 2) acc[(q >> 1) - 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bw[6 + s][pl].h, v, acc[(q >> 1) - 1], 0, 0, 0);   // filter row 2
                    }
                }
            }
        }
        if (STATS) {
#pragma unroll
            for (int a = 0; a < TH; ++a)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = acc[a][r];
                    s1[r] += v;
                    s2[r] = __builtin_fmaf(v, v, s2[r]);
                }
        }
#pragma unroll
        for (int a = 0; a < TH; ++a) {
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[r] = acc[a][r];
                if (EPI >= 1) v[r] = v[r] * sc[r] + sh[r];
                if (EPI == 2) v[r] = v[r] * __builtin_amdgcn_rcpf(1.0f + __expf(-v[r]));     // SiLU; 1 ulp, then rounded to bf16
            }
            bf16x4 o = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
            *(bf16x4*)(sS + ep_off + a * 16 * SROW) = o;
        }
        S2_BARRIER(true);                                  // staging complete; every wave is done with patch `cur`
        if (st_ch * 8 < p.K) {
            const int n = t / per_img, rem = t - n * per_img;
            const int th = rem / tiles_w, tw = rem - th * tiles_w;
            const long long org = (((long long)n * p.Ho + th * TH) * p.Wo + tw * TW);
            bf16_t* yb = (bf16_t*)p.y + org * p.ldy + st_off;
            const bf16_t* rb = p.res ? (const bf16_t*)p.res + org * p.ldr + rs_off : nullptr;
            V16 v[TH / 2];                                 // 32 pixel rows per pass: both passes' chunks by asm reads, then the stores back to back
#pragma unroll
            for (int j = 0; j < TH / 2; ++j) S2_LDSR(v[j].i, lds0 + 2 * PATCH_B + st_lds + j * 32 * SROW);
            S2_LGKM(0);
            if (p.res || p.accumulate) {                   // every row requested before the first store (a load behind a store waits for the store too)
                V16 qr[TH / 2], qa[TH / 2];
#pragma unroll
                for (int j = 0; j < TH / 2; ++j) {
                    if (p.res) qr[j].i = *(const i32x4*)(rb + j * rs_step);
                    if (p.accumulate) qa[j].i = *(const i32x4*)(yb + j * st_step);
                }
#pragma unroll
                for (int j = 0; j < TH / 2; ++j) {
                    if (st_rr & 1) v[j].i = i32x4{v[j].i[2], v[j].i[3], v[j].i[0], v[j].i[1]};
                    float g[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) g[e] = (float)v[j].h[e];
                    if (p.res) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) g[e] += (float)qr[j].h[e];
                    }
                    if (p.accumulate) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) g[e] += (float)qa[j].h[e];
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[j].h[e] = (bf16_t)g[e];
                }
            } else {
#pragma unroll
                for (int j = 0; j < TH / 2; ++j)
                    if (st_rr & 1) v[j].i = i32x4{v[j].i[2], v[j].i[3], v[j].i[0], v[j].i[1]};
            }
#pragma unroll
            for (int j = 0; j < TH / 2; ++j) *(i32x4*)(yb + j * st_step) = v[j].i;
        }
        // the next patch's DMA precedes these TH/2 stores in the wave's vm queue: wait for it, not for the stores
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(TH / 2) : "memory");
        S2_BARRIER(false);                                 // next patch landed for everyone; staging tile free again (its reads returned before the stores left)
    };

    if (wg < tiles) {
        issue_patch(wg, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int cur = 0;
        for (int t = wg; t < tiles; t += (int)gridDim.x, cur ^= 1) {
            if (t + (int)gridDim.x < tiles) issue_patch(t + (int)gridDim.x, cur ^ 1);
            compute(t, cur);
        }
    }

    if (STATS) {                                           // a wave's channels are its own: 16 pixel lanes -> one value, no LDS
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float u = s1[r], q = s2[r];
            u = row_sum16(u);
            q = row_sum16(q);
            const int c = wave * 16 + fq * 4 + r;
            if (fr == 0 && c < p.K) {
                p.stats[((size_t)wg * 2 + 0) * p.K + c] = u;
                p.stats[((size_t)wg * 2 + 1) * p.K + c] = q;
            }
        }
    }
}

template <int NP, int NWV, int EPI, bool STATS>
static void launch_s2(const ConvArgs& a, int grid, hipStream_t st) {
    using G = Geo<NP, NWV>;
    static PerDeviceOnce attr_once;           // first launch of this instance on any thread
    attr_once.run([&] {
        (void)hipFuncSetAttribute((const void*)conv3x3s2_kernel<NP, NWV, EPI, STATS>, hipFuncAttributeMaxDynamicSharedMemorySize, G::SMEM_B);
    });
    hdy_note_dispatch(NP == 1 ? "conv3x3s2_c32" : "conv3x3s2_c64");
    hipLaunchKernelGGL((conv3x3s2_kernel<NP, NWV, EPI, STATS>), dim3(grid), dim3(G::NTHR), G::SMEM_B, st, a);
}

template <int NP, int NWV>
static void launch_s2_epi(const ConvArgs& a, int grid, int epi, hipStream_t st) {
    if (a.stats) {
        if (epi == 2) launch_s2<NP, NWV, 2, true>(a, grid, st);
        else if (epi == 1) launch_s2<NP, NWV, 1, true>(a, grid, st);
        else launch_s2<NP, NWV, 0, true>(a, grid, st);
    } else {
        if (epi == 2) launch_s2<NP, NWV, 2, false>(a, grid, st);
        else if (epi == 1) launch_s2<NP, NWV, 1, false>(a, grid, st);
        else launch_s2<NP, NWV, 0, false>(a, grid, st);
    }
}

}  // namespace

// planes of the instance that takes the shape: 1 = <1, 4> (32 -> <= 64), 2 = <2, 8> (64 -> 72..128), 0 = not eligible
static int conv3x3s2_planes(int Cin, int K, int R, int S, int stride, int pad, int H, int W, int dtype) {
    const int disabled = hdy_opt(HDY_OPT_NO_CONV3X3S2);     // tests: 1 forces the generic kernel for A/B comparison, 2 only for the two-plane (64 -> 128) form
    if (disabled == 1 || dtype != HDY_BF16 || R != 3 || S != 3 || stride != 2 || pad != 1 || K % 8 || H % 2 || W % 2 || (H / 2) % TH || (W / 2) % TW) return 0;
    if (Cin == 32 && K <= 64) return 1;
    if (Cin == 64 && K > 64 && K <= 128 && disabled != 2) return 2;
    return 0;
}

// one-plane form: three 46 KB, 4-wave workgroups per CU; two-plane form: one 93 KB, 8-wave workgroup
static int conv3x3s2_grid(int planes, int tiles) {
    const int cap = planes == 1 ? 768 : 256;
    return tiles < cap ? tiles : cap;
}

// Number of statistic slabs this kernel writes for the shape (one per workgroup), 0 = not eligible.
int hdy_conv3x3s2_c32_slabs(int N, int H, int W, int Cin, int K, int R, int S, int stride, int pad, int dtype) {
    const int planes = conv3x3s2_planes(Cin, K, R, S, stride, pad, H, W, dtype);
    if (!planes) return 0;
    return conv3x3s2_grid(planes, N * (H / 2 / TH) * (W / 2 / TW));
}

// Returns 1 and launches when the shape qualifies; 0 = not eligible (caller falls back to the generic kernel); <0 / >0 in *rc = error.
int hdy_conv3x3s2_c32_try(const ConvArgs& a, int dtype, int out_f32, hipStream_t st, int* rc) {
    if (dtype != HDY_BF16 || out_f32 || a.act > 1) return 0;
    if (!(a.TH == 3 && a.TW == 3 && a.ih_mul == 2 && a.iw_mul == 2 && a.dh0 == -1 && a.dw0 == -1 && a.dense_out && !a.span_pixels)) return 0;
    const int planes = (a.Hin == 2 * a.Ho && a.Win == 2 * a.Wo) ? conv3x3s2_planes(a.C, a.K, 3, 3, 2, 1, a.Hin, a.Win, dtype) : 0;
    if (!planes) return 0;
    const bool aligned = a.ldx % 8 == 0 && a.ldy % 8 == 0 && ((uintptr_t)a.y & 15) == 0 && ((uintptr_t)a.x & 15) == 0 && ((uintptr_t)a.w & 15) == 0 &&
                         a.Kdp % 8 == 0 && (!a.res || (a.ldr % 8 == 0 && ((uintptr_t)a.res & 15) == 0));
    if (!aligned) {
        if (!a.stats) return 0;
        // the caller sized the slab array with hdy_conv_stat_slabs for THIS kernel: falling back would write a different count
        hdy_set_error("conv3x3s2: statistics requested but x/y/res rows are not 16-byte aligned (ldx=%d ldy=%d)", a.ldx, a.ldy);
        *rc = HDY_EINVAL;
        return 1;
    }
    const int grid = conv3x3s2_grid(planes, a.N * (a.Ho / TH) * (a.Wo / TW));
    HDY_STAT_CAP(a, grid, "conv3x3s2")
    const int epi = a.act == 1 ? 2 : ((a.scale || a.shift) ? 1 : 0);
    if (planes == 1) launch_s2_epi<1, 4>(a, grid, epi, st);
    else launch_s2_epi<2, 8>(a, grid, epi, st);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        hdy_set_error("conv3x3s2: launch failed: %s", hipGetErrorString(e));
        *rc = (int)e;
        return 1;
    }
    *rc = HDY_OK;
    return 1;
}

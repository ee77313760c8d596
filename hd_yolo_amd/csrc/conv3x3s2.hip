// 3x3 / stride 2 / pad 1 convolution for 32 input channels and up to 64 output channels (bf16), patch-resident with the filter in
// registers: yolov5s' second layer (32 -> 64, 320x320 -> 160x160 at 640x640 tiles), which the generic implicit GEMM served at 0.4 of its
// HBM bound — 64-byte input pixels fetched once per tap (nine 64-byte segments per output pixel).
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

constexpr int NTHR = 256;                                            // 4 waves: wave = 16-channel group
constexpr int TH = 4, TW = 16;
constexpr int PROWS = 2 * TH + 1, PPAIR = TW + 1;                    // 9 patch rows of 17 pixel pairs (34 pixels, 33 used)
constexpr int NPAIR = PROWS * PPAIR;                                 // 153 rows of 128 bytes
constexpr int PATCH_B = NPAIR * 128;                                 // 19584
constexpr int STAGE_B = TH * TW * 128;                               // 8192
constexpr int SMEM_B = 2 * PATCH_B + STAGE_B;                        // 47360: three workgroups per CU
constexpr int NPASS = (NPAIR * 8 + NTHR - 1) / NTHR;                 // 5 loader passes (last partial)
constexpr int C = 32;

__device__ __forceinline__ void glds16(const void* g, unsigned char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)g, (void __attribute__((address_space(3)))*)lds_wave_base, 16, 0, 0);
}

// EPI: 0 = raw convolution out (train-mode forward), 1 = scale/shift, 2 = scale/shift + SiLU;  STATS: BatchNorm partial sums (one slab per workgroup)
template <int EPI, bool STATS>
__global__ __launch_bounds__(NTHR, 3) void conv3x3s2_c32_kernel(const ConvArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sP = smem;                    // [2][153][128 B]
    unsigned char* sS = smem + 2 * PATCH_B;      // [64][128 B] staging tile

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int tiles_w = p.Wo / TW, tiles_h = p.Ho / TH, per_img = tiles_w * tiles_h;
    const int tiles = p.N * per_img;
    const int wg = xcd_remap(blockIdx.x, gridDim.x);

    const bf16_t* __restrict__ x = (const bf16_t*)p.x;
    const bf16_t* __restrict__ w = (const bf16_t*)p.w;
    const unsigned char* zero = (const unsigned char*)g_hdy_zero16_s2;

    // ---- filter slice -> registers: row operand of tap t = w[wave*16 + fr][t*32 + fq*8 .. +7]
    V16 bw[9];
    {
        const int k = wave * 16 + fr;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const void* src = (k < p.K) ? (const void*)(w + (size_t)k * p.Kdp + t * C + fq * 8) : (const void*)zero;
            bw[t].i = *(const i32x4*)src;
        }
    }

    const int slot = tid & 7;
    auto issue_patch = [&](int t, int buf) {
        const int n = t / per_img, rem = t - n * per_img;
        const int th = rem / tiles_w, tw = rem - th * tiles_w;
        const int h0 = 2 * th * TH - 1, w0 = 2 * tw * TW - 1;
        const bf16_t* org = x + (((long long)n * p.Hin + h0) * p.Win + w0) * p.ldx;      // patch pixel (0, 0); may lie outside the image
        unsigned char* dst = sP + buf * PATCH_B;
#pragma unroll
        for (int i = 0; i < NPASS; ++i) {
            if ((wave * 64 + NTHR * i) / 8 >= NPAIR) break;                // wave-uniform: whole 1 KB pieces past the patch
            const int pr = (tid + NTHR * i) >> 3;                          // pair row 0..152
            if (pr >= NPAIR) continue;
            const int py = (pr * 3856) >> 16, P = pr - py * PPAIR;          // pr / 17 for pr < 153
            const int L = slot ^ ((P >> 1) & 7);                           // logical 16-byte piece that belongs in this physical slot
            const int px = 2 * P + (L >> 2), h = h0 + py, ww = w0 + px;
            const void* src = ((unsigned)h < (unsigned)p.Hin && (unsigned)ww < (unsigned)p.Win && px < 2 * TW + 1)
                                  ? (const void*)(org + ((long long)py * p.Win + px) * p.ldx + (L & 3) * 8) : (const void*)zero;
            glds16(src, dst + (wave * 64 + NTHR * i) * 16);
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
    const int st_ch = tid & 7, st_rr = tid >> 3;                              // store phase: chunk, first row (rows + 32 j)
    const int st_lds = st_rr * 128 + ((st_ch ^ ((st_rr >> 1) & 7)) << 4);
    const long long st_off = ((long long)(st_rr >> 4) * p.Wo + (st_rr & 15)) * p.ldy + st_ch * 8;
    const long long st_step = (long long)2 * p.Wo * p.ldy;
    const long long rs_off = ((long long)(st_rr >> 4) * p.Wo + (st_rr & 15)) * p.ldr + st_ch * 8;
    const long long rs_step = (long long)2 * p.Wo * p.ldr;
    const int ep_off = fr * 128 + (((wave * 4 + fq) ^ (fr & 15)) << 3);      // key fr & 15: 16 rows -> 16 slots (see conv3x3.hip); odd rows swap a chunk's halves

    auto compute = [&](int t, int cur) {
        f32x4 acc[TH];
#pragma unroll
        for (int a = 0; a < TH; ++a) acc[a] = f32x4{0.f, 0.f, 0.f, 0.f};
        const unsigned char* pb = sP + cur * PATCH_B;
        V16 f[2][3];
#pragma unroll
        for (int s = 0; s < 3; ++s) f[0][s].i = *(const i32x4*)(pb + aoff[s]);
#pragma unroll
        for (int q = 0; q < PROWS; ++q) {
            if (q + 1 < PROWS) {
#pragma unroll
                for (int s = 0; s < 3; ++s) f[(q + 1) & 1][s].i = *(const i32x4*)(pb + (q + 1) * PPAIR * 128 + aoff[s]);
            }
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const bf16x8 v = f[q & 1][s].h;
                if (q & 1) {
                    acc[q >> 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bw[3 + s].h, v, acc[q >> 1], 0, 0, 0);                      // filter row 1
                } else {
                    if ((q >> 1) < TH) acc[q >> 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bw[s].h, v, acc[q >> 1], 0, 0, 0);        // filter row 0
                    if (q >= 2) acc[(q >> 1) - 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bw[6 + s].h, v, acc[(q >> 1) - 1], 0, 0, 0);   // filter row 2
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
            *(bf16x4*)(sS + ep_off + a * 16 * 128) = o;
        }
        __syncthreads();                                   // staging complete; every wave is done with patch `cur`
        if (st_ch * 8 < p.K) {
            const int n = t / per_img, rem = t - n * per_img;
            const int th = rem / tiles_w, tw = rem - th * tiles_w;
            const long long org = (((long long)n * p.Ho + th * TH) * p.Wo + tw * TW);
            bf16_t* yb = (bf16_t*)p.y + org * p.ldy + st_off;
            const bf16_t* rb = p.res ? (const bf16_t*)p.res + org * p.ldr + rs_off : nullptr;
#pragma unroll
            for (int j = 0; j < TH / 2; ++j) {             // 32 pixel rows of 128 bytes per pass
                V16 v;
                v.i = *(const i32x4*)(sS + st_lds + j * 32 * 128);
                if (st_rr & 1) v.i = i32x4{v.i[2], v.i[3], v.i[0], v.i[1]};
                if (p.res || p.accumulate) {
                    float g[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) g[e] = (float)v.h[e];
                    if (p.res) {
                        V16 q;
                        q.i = *(const i32x4*)(rb + j * rs_step);
#pragma unroll
                        for (int e = 0; e < 8; ++e) g[e] += (float)q.h[e];
                    }
                    if (p.accumulate) {
                        V16 q;
                        q.i = *(const i32x4*)(yb + j * st_step);
#pragma unroll
                        for (int e = 0; e < 8; ++e) g[e] += (float)q.h[e];
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) v.h[e] = (bf16_t)g[e];
                }
                *(i32x4*)(yb + j * st_step) = v.i;
            }
        }
        // the next patch's DMA precedes these TH/2 stores in the wave's vm queue: wait for it, not for the stores
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(TH / 2) : "memory");
        __syncthreads();                                   // next patch landed for everyone; staging tile free again
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
#pragma unroll
            for (int m = 1; m < 16; m <<= 1) {
                u += __shfl_xor(u, m);
                q += __shfl_xor(q, m);
            }
            const int c = wave * 16 + fq * 4 + r;
            if (fr == 0 && c < p.K) {
                p.stats[((size_t)wg * 2 + 0) * p.K + c] = u;
                p.stats[((size_t)wg * 2 + 1) * p.K + c] = q;
            }
        }
    }
}

template <int EPI, bool STATS>
static void launch_s2(const ConvArgs& a, int grid, hipStream_t st) {
    static PerDeviceOnce attr_once;           // first launch of this instance on any thread
    attr_once.run([&] {
        (void)hipFuncSetAttribute((const void*)conv3x3s2_c32_kernel<EPI, STATS>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_B);
    });
    hdy_note_dispatch("conv3x3s2_c32");
    hipLaunchKernelGGL((conv3x3s2_c32_kernel<EPI, STATS>), dim3(grid), dim3(NTHR), SMEM_B, st, a);
}

}  // namespace

static bool conv3x3s2_shape_ok(int Cin, int K, int R, int S, int stride, int pad, int H, int W, int dtype) {
    const bool disabled = hdy_opt(HDY_OPT_NO_CONV3X3S2) != 0;     // tests: force the generic kernel for A/B comparison
    return !disabled && dtype == HDY_BF16 && R == 3 && S == 3 && stride == 2 && pad == 1 && Cin == C && K <= 64 && K % 8 == 0 && H % 2 == 0 && W % 2 == 0 &&
           (H / 2) % TH == 0 && (W / 2) % TW == 0;
}

static int conv3x3s2_grid(int tiles) { return tiles < 768 ? tiles : 768; }     // three 46 KB, 4-wave workgroups per CU

// Number of statistic slabs this kernel writes for the shape (one per workgroup), 0 = not eligible.
int hdy_conv3x3s2_c32_slabs(int N, int H, int W, int Cin, int K, int R, int S, int stride, int pad, int dtype) {
    if (!conv3x3s2_shape_ok(Cin, K, R, S, stride, pad, H, W, dtype)) return 0;
    return conv3x3s2_grid(N * (H / 2 / TH) * (W / 2 / TW));
}

// Returns 1 and launches when the shape qualifies; 0 = not eligible (caller falls back to the generic kernel); <0 / >0 in *rc = error.
int hdy_conv3x3s2_c32_try(const ConvArgs& a, int dtype, int out_f32, hipStream_t st, int* rc) {
    if (dtype != HDY_BF16 || out_f32 || a.act > 1) return 0;
    if (!(a.TH == 3 && a.TW == 3 && a.ih_mul == 2 && a.iw_mul == 2 && a.dh0 == -1 && a.dw0 == -1 && a.dense_out && !a.span_pixels)) return 0;
    if (!(a.Hin == 2 * a.Ho && a.Win == 2 * a.Wo && conv3x3s2_shape_ok(a.C, a.K, 3, 3, 2, 1, a.Hin, a.Win, dtype))) return 0;
    const bool aligned = a.ldx % 8 == 0 && a.ldy % 8 == 0 && ((uintptr_t)a.y & 15) == 0 && ((uintptr_t)a.x & 15) == 0 && ((uintptr_t)a.w & 15) == 0 &&
                         a.Kdp % 8 == 0 && (!a.res || (a.ldr % 8 == 0 && ((uintptr_t)a.res & 15) == 0));
    if (!aligned) {
        if (!a.stats) return 0;
        // the caller sized the slab array with hdy_conv_stat_slabs for THIS kernel: falling back would write a different count
        hdy_set_error("conv3x3s2_c32: statistics requested but x/y/res rows are not 16-byte aligned (ldx=%d ldy=%d)", a.ldx, a.ldy);
        *rc = HDY_EINVAL;
        return 1;
    }
    const int grid = conv3x3s2_grid(a.N * (a.Ho / TH) * (a.Wo / TW));
    HDY_STAT_CAP(a, grid, "conv3x3s2_c32")
    const int epi = a.act == 1 ? 2 : ((a.scale || a.shift) ? 1 : 0);
    if (a.stats) {
        if (epi == 2) launch_s2<2, true>(a, grid, st);
        else if (epi == 1) launch_s2<1, true>(a, grid, st);
        else launch_s2<0, true>(a, grid, st);
    } else {
        if (epi == 2) launch_s2<2, false>(a, grid, st);
        else if (epi == 1) launch_s2<1, false>(a, grid, st);
        else launch_s2<0, false>(a, grid, st);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        hdy_set_error("conv3x3s2_c32: launch failed: %s", hipGetErrorString(e));
        *rc = (int)e;
        return 1;
    }
    *rc = HDY_OK;
    return 1;
}

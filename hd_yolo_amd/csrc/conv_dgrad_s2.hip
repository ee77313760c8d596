// Data gradient of the 3x3 / stride 2 / pad 1 convolution 32 -> 64 (yolov5s' second layer, 320x320 -> 160x160 at 640x640 tiles): dx has
// 32 channels, dy 64 (bf16).  Patch-resident, filter in registers — the layer the generic kernel served worst (class walk: 257 us for a
// 100 us HBM bound, 64-byte output pixels written as four interleaved parity classes).
//
//   dx[n, 2i+a, 2j+b, c] = SUM over the taps (r, s) of parity class (a, b), k:  dy[n, i+di, j+dj, k] * w[k][c][r][s]
//   class (0,0): (1,1)@(0,0)        class (0,1): (1,2)@(0,0) (1,0)@(0,1)        class (1,0): (2,1)@(0,0) (0,1)@(1,0)
//   class (1,1): (2,2)@(0,0) (2,0)@(0,1) (0,2)@(1,0) (0,0)@(1,1)                 — nine (class, tap) products over four dy shifts (di, dj)
//
// A 4-wave workgroup takes an 8 x 16 tile of dy positions: the 9 x 17-pixel dy patch (19.6 KB) arrives by LDS-DMA (double buffered, the
// filter-resident 3x3 kernel's source-side XOR swizzle, zero page past the image), and the workgroup produces the 16 x 32 block of dx pixels.
// Wave = (16-channel group wc) x (upper / lower four tile rows wp); it keeps the nine filter slices of its channels as MFMA row operands
// (9 x 2 k-halves x 4 VGPRs = 72) and 4 classes x 4 rows of accumulators (64 VGPRs).  Per patch row and k-half two fragment reads (column
// shift 0 / 1) feed 6 + 3 MFMAs.  The filter is read from the class-walk packing of the generic kernel as it is (hdy_conv_pack_describe,
// kind dgrad: per class rows = c, columns = tap * 64 + k).
// Epilogue: a lane holds 4 consecutive channels of one dx pixel; the block is staged as 256 rows of 128 bytes (two neighbouring dx pixels per
// row, 8-byte slots XORed with (row & 14)) and leaves with 16-byte stores, every dx row of the block as one contiguous run.
//
// Requirements (checked by the launcher, otherwise the class walk runs): bf16, dy 64 channels, dx 32 channels, even H and W, H/2 % 8 == 0,
// W/2 % 16 == 0, 16-byte aligned rows, no producer-side statistics.
#include <stdlib.h>

#include "common.h"
#include "hdyolo_internal.h"

__device__ uint4 g_hdy_zero16_d2[4];   // zero page for patch pixels past the image

namespace {

constexpr int NTHR = 256;
constexpr int TH = 8, TW = 16, PW = TW + 1, PH = TH + 1, PPIX = PH * PW;     // 153 patch pixels
constexpr int CB = 128, CPP = 8;                                             // bytes / 16-byte chunks per dy pixel (64 channels)
constexpr int PATCH_B = PPIX * CB;                                           // 19584
constexpr int STAGE_B = 256 * 128;                                           // 16 dx rows x 16 pixel pairs x 128 B
constexpr int SMEM_B = 2 * PATCH_B + STAGE_B;                                // 71936: two workgroups per CU
constexpr int NPASS = (PPIX * CPP + NTHR - 1) / NTHR;                        // 5 loader passes (last partial)

__device__ __forceinline__ void glds16(const void* g, unsigned char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)g, (void __attribute__((address_space(3)))*)lds_wave_base, 16, 0, 0);
}

__global__ __launch_bounds__(NTHR, 2) void dgrad3x3s2_k64c32_kernel(const ConvArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sP = smem;                    // [2][153][128 B]
    unsigned char* sS = smem + 2 * PATCH_B;      // [256][128 B] staging block

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int wc = wave & 1, wp = wave >> 1;
    const int tiles_w = p.Wo / TW, tiles_h = p.Ho / TH, per_img = tiles_w * tiles_h;
    const int tiles = p.N * per_img;
    const int wg = xcd_remap(blockIdx.x, gridDim.x);

    const bf16_t* __restrict__ x = (const bf16_t*)p.x;          // dy [N][Ho][Wo][ldx]
    const bf16_t* __restrict__ w = (const bf16_t*)p.w;
    const unsigned char* zero = (const unsigned char*)g_hdy_zero16_d2;

    // ---- the nine (class, tap) filter slices of this wave's 16 channels: row operand = w_cls[c = wc*16 + fr][tap*64 + (ks*4 + fq)*8 .. +7]
    V16 bw[9][2];
    {
        constexpr int cls_of[9] = {0, 1, 1, 2, 2, 3, 3, 3, 3}, tap_of[9] = {0, 0, 1, 0, 1, 0, 1, 2, 3};
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const bf16_t* base = w + p.c_w[cls_of[i]] + (size_t)(wc * 16 + fr) * (p.c_nkb[cls_of[i]] * 64) + tap_of[i] * 64;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) bw[i][ks].i = *(const i32x4*)(base + (ks * 4 + fq) * 8);
        }
    }

    const int lc = tid & 7;
    auto issue_patch = [&](int t, int buf) {
        const int n = t / per_img, rem = t - n * per_img;
        const int th = rem / tiles_w, tw = rem - th * tiles_w;
        const int i0 = th * TH, j0 = tw * TW;
        const bf16_t* org = x + (((long long)n * p.Hin + i0) * p.Win + j0) * p.ldx;
        unsigned char* dst = sP + buf * PATCH_B;
#pragma unroll
        for (int i = 0; i < NPASS; ++i) {
            if ((wave * 64 + NTHR * i) / CPP >= PPIX) break;               // wave-uniform: whole 1 KB pieces past the patch
            const int pix = (tid + NTHR * i) / CPP;
            if (pix >= PPIX) continue;
            const int py = (pix * 3856) >> 16, px = pix - py * PW;          // pix / 17 for pix < 153
            const int lcp = lc ^ (((px >> 1) & 3) << 1);                   // 128-byte pixel rows XOR-swizzled by the patch column (conv3x3.hip)
            const void* src = (i0 + py < p.Hin && j0 + px < p.Win) ? (const void*)(org + ((long long)py * p.Win + px) * p.ldx + lcp * 8) : (const void*)zero;
            glds16(src, dst + (wave * 64 + NTHR * i) * 16);
        }
    };

    int aoff[2][2];                               // fragment byte offset inside a patch row: [column shift][k-half]
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int px = fr + s;
            aoff[s][ks] = px * CB + (((ks * 4 + fq) ^ (((px >> 1) & 3) << 1)) << 4);
        }

    const int st_ch = tid & 7, st_rr = tid >> 3;                              // store phase: 16-byte chunk, first staging row (rows + 32 j)
    const int st_lds = st_rr * 128 + ((st_ch ^ ((st_rr >> 1) & 7)) << 4);
    // staging row R = hr * 16 + col: dx row 2*i0 + hr, pixels 2*j0 + 2*col + (st_ch >> 2), channels (st_ch & 3) * 8 ..
    const int st_col = st_rr & 15, st_hr0 = st_rr >> 4;                       // hr = st_hr0 + 2 j
    const long long st_pix = (long long)(2 * st_col + (st_ch >> 2)) * p.ldy + (st_ch & 3) * 8;

    auto compute = [&](int t, int cur) {
        f32x4 acc[4][4];                                                      // [class][tile row of this wave]
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int a = 0; a < 4; ++a) acc[c][a] = f32x4{0.f, 0.f, 0.f, 0.f};
        const unsigned char* pb = sP + cur * PATCH_B + wp * 4 * PW * CB;      // this wave reads patch rows wp*4 .. wp*4 + 4
        V16 f[2][2];                                                          // [buffer][column shift]
        f[0][0].i = *(const i32x4*)(pb + aoff[0][0]);
        f[0][1].i = *(const i32x4*)(pb + aoff[1][0]);
#pragma unroll
        for (int g = 0; g < 10; ++g) {                                        // group g = (patch row q = g / 2, k-half ks = g % 2)
            const int q = g >> 1, ks = g & 1;
            if (g + 1 < 10) {
                f[(g + 1) & 1][0].i = *(const i32x4*)(pb + ((g + 1) >> 1) * PW * CB + aoff[0][(g + 1) & 1]);
                f[(g + 1) & 1][1].i = *(const i32x4*)(pb + ((g + 1) >> 1) * PW * CB + aoff[1][(g + 1) & 1]);
            }
            const bf16x8 f0 = f[g & 1][0].h, f1 = f[g & 1][1].h;
            if (q < 4) {                                                      // row shift 0: tile row a = q
                acc[0][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bw[0][ks].h, f0, acc[0][q], 0, 0, 0);
                acc[1][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bw[1][ks].h, f0, acc[1][q], 0, 0, 0);
                acc[2][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bw[3][ks].h, f0, acc[2][q], 0, 0, 0);
                acc[3][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bw[5][ks].h, f0, acc[3][q], 0, 0, 0);
                acc[1][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bw[2][ks].h, f1, acc[1][q], 0, 0, 0);
                acc[3][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bw[6][ks].h, f1, acc[3][q], 0, 0, 0);
            }
            if (q >= 1) {                                                     // row shift 1: tile row a = q - 1
                acc[2][q - 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bw[4][ks].h, f0, acc[2][q - 1], 0, 0, 0);
                acc[3][q - 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bw[7][ks].h, f0, acc[3][q - 1], 0, 0, 0);
                acc[3][q - 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bw[8][ks].h, f1, acc[3][q - 1], 0, 0, 0);
            }
        }
        // ---- staging: class (ca, cb), tile row wp*4 + a -> staging row (2*(wp*4 + a) + ca) * 16 + fr, 8-byte slot cb*8 + wc*4 + fq
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const int ca = c >> 1, cb = c & 1;
                const int R = (2 * (wp * 4 + a) + ca) * 16 + fr;
                const int slot = cb * 8 + wc * 4 + fq;
                bf16x4 o = {(bf16_t)acc[c][a][0], (bf16_t)acc[c][a][1], (bf16_t)acc[c][a][2], (bf16_t)acc[c][a][3]};
                *(bf16x4*)(sS + R * 128 + ((slot ^ (fr & 15)) << 3)) = o;        // key fr & 15: 16 rows -> 16 slots (see conv3x3.hip); odd rows swap a chunk's halves
            }
        __syncthreads();                                   // staging complete; every wave is done with patch `cur`
        {
            const int n = t / per_img, rem = t - n * per_img;
            const int th = rem / tiles_w, tw = rem - th * tiles_w;
            bf16_t* yb = (bf16_t*)p.y + (((long long)n * p.Hout + 2 * th * TH + st_hr0) * p.Wout + 2 * tw * TW) * p.ldy + st_pix;
            const long long st_step = (long long)2 * p.Wout * p.ldy;          // staging rows + 32 = dx rows + 2
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                V16 v;
                v.i = *(const i32x4*)(sS + st_lds + j * 32 * 128);
                if (st_rr & 1) v.i = i32x4{v.i[2], v.i[3], v.i[0], v.i[1]};
                if (p.accumulate) {
                    V16 u;
                    u.i = *(const i32x4*)(yb + j * st_step);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v.h[e] = (bf16_t)((float)v.h[e] + (float)u.h[e]);
                }
                *(i32x4*)(yb + j * st_step) = v.i;
            }
        }
        // the next patch's DMA precedes these 8 stores in the wave's vm queue: wait for it, not for the stores (accumulate: its loads have returned)
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        __syncthreads();                                   // next patch landed for everyone; staging block free again
    };

    if (wg >= tiles) return;
    issue_patch(wg, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int cur = 0;
    for (int t = wg; t < tiles; t += (int)gridDim.x, cur ^= 1) {
        if (t + (int)gridDim.x < tiles) issue_patch(t + (int)gridDim.x, cur ^ 1);
        compute(t, cur);
    }
}

}  // namespace

// Returns 1 and launches when the class-walk arguments describe this layer; 0 = not eligible (the generic kernel runs).
int hdy_dgrad3x3s2_try(const ConvArgs& a, int dtype, hipStream_t st, int* rc) {
    const bool disabled = hdy_opt(HDY_OPT_NO_DGRAD_S2) != 0;        // tests: force the class walk for A/B comparison
    if (disabled || dtype != HDY_BF16 || a.ncls != 4 || a.nstat != 0 || a.res || a.scale || a.shift || a.act != 0 || a.stats) return 0;
    if (a.C != 64 || a.K != 32 || a.Ho % TH || a.Wo % TW || a.Hin != a.Ho || a.Win != a.Wo || a.Hout != 2 * a.Ho || a.Wout != 2 * a.Wo) return 0;
    static const int dh[4] = {0, 0, 0, 0}, dw[4] = {0, 0, 0, 0}, nth[4] = {1, 1, 2, 2}, ntw[4] = {1, 2, 1, 2}, oh[4] = {0, 0, 1, 1}, ow[4] = {0, 1, 0, 1};
    for (int c = 0; c < 4; ++c)
        if (a.c_dh[c] != dh[c] || a.c_dw[c] != dw[c] || a.c_TH[c] != nth[c] || a.c_TW[c] != ntw[c] || a.c_oh[c] != oh[c] || a.c_ow[c] != ow[c] ||
            a.c_nkb[c] != nth[c] * ntw[c] || a.c_w[c] % 8)
            return 0;
    if (a.ldx % 8 || a.ldy % 8 || ((uintptr_t)a.x & 15) || ((uintptr_t)a.y & 15) || ((uintptr_t)a.w & 15)) return 0;
    const int tiles = a.N * (a.Ho / TH) * (a.Wo / TW);
    static PerDeviceOnce attr_once;           // first launch of this instance on any thread
    attr_once.run([&] {
        (void)hipFuncSetAttribute((const void*)dgrad3x3s2_k64c32_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_B);
    });
    const int grid = tiles < 512 ? tiles : 512;        // two 70 KB, 4-wave workgroups per CU
    hdy_note_dispatch("dgrad3x3s2_k64c32");
    hipLaunchKernelGGL(dgrad3x3s2_k64c32_kernel, dim3(grid), dim3(NTHR), SMEM_B, st, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        hdy_set_error("dgrad3x3s2: launch failed: %s", hipGetErrorString(e));
        *rc = (int)e;
        return 1;
    }
    *rc = HDY_OK;
    return 1;
}

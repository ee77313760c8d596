// The 6x6 / stride 2 / pad 2 RGB stem (metayolo/models/layers.py:25-41 applied to hub yaml row 0: Conv [64, 6, 2, 2]) as a
// patch-resident kernel (bf16).
//
// Through the generic implicit GEMM the stem fetched its A operand once per row-tap: 384 B from L2 per output pixel for
// 64 B written, i.e. 2.5 GB of L2->LDS traffic per 64-tile batch — 378 us against an HBM bound of ~130 us (632 MB).
// Here a workgroup stages the (2*16+4) x (2*32+4) input patch of a 16 x 32 output tile ONCE (19.6 KB of the padded 4-channel
// image hdy_stem_prep wrote: 8 bytes per pixel, zero border, so no bounds tests at all) and every tap reads it from LDS:
//   * k order = the packed stem filter's: k = r*24 + s*4 + c (r = filter row, s = filter column, c = channel incl. the zero 4th),
//     so one 16-byte fragment = filter row r, columns 2*part, 2*part+1 = two adjacent input pixels; 18 real fragments per output
//     pixel, padded to 20 (5 MFMA k-steps of 32); the padding fragments re-read valid image bytes against zero filter columns.
//   * the 16 lanes of a fragment are 16 consecutive output columns = input pixels 32 bytes apart; 16-byte reads at a 16-byte
//     lane pitch... (stride 2 pixels x 8 B) -> contiguous 256 bytes: conflict-free ds_read_b128.
//   * the filter (K x 160 bf16, LDS resident) is the MFMA row operand, so a lane's accumulator is 4
//     consecutive output channels of one pixel: packed 8-byte epilogue writes, BatchNorm sums in registers across all tiles
//     (one statistics slab per workgroup), 16-byte coalesced row stores from an LDS staging tile.
// 8 waves x 64 pixels; two 72 KB workgroups per CU for K = 32.
#include <stdlib.h>

#include "common.h"
#include "hdyolo_internal.h"

namespace {

constexpr int NTHR = 512;
constexpr int TOH = 16, TOW = 32;                      // output tile
constexpr int PH = 2 * TOH + 4, PWX = 2 * TOW + 4;      // 36 x 68 input pixels
constexpr int PROW = PWX * 8;                           // 544 bytes per patch row
constexpr int PATCH_B = PH * PROW;                      // 19584
constexpr int PCHUNKS = PATCH_B / 16;                   // 1224 16-byte pieces
constexpr int KSTEPS = 5;                               // 160 = 5 x 32 >= 144

__device__ __forceinline__ void glds16(const void* g, unsigned char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)g,
                                     (void __attribute__((address_space(3)))*)lds_wave_base, 16, 0, 0);
}

// NT = K / 16 (1..4); EPI: 0 raw, 1 scale/shift, 2 scale/shift + SiLU
template <int NT, int EPI>
__global__ __launch_bounds__(NTHR) void conv_stem_kernel(const ConvArgs p) {
    constexpr int K = NT * 16;
    constexpr int ROWB = K * 2;                          // bytes per staged output pixel
    constexpr int CPR = K / 8;                           // 16-byte chunks per output pixel
    constexpr int WPITCH = 400;                          // filter row pitch in LDS: 100 banks -> the 16 rows of a fragment read never collide
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sP = smem;                            // [2][PATCH_B]
    unsigned char* sS = smem + 2 * PATCH_B;              // [512 pixels][ROWB]
    unsigned char* sW = sS + TOH * TOW * ROWB;           // [K][WPITCH]: filter, 160 bf16 used per row

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int tiles_w = p.Wo / TOW, tiles_h = p.Ho / TOH;
    const int per_img = tiles_w * tiles_h;
    const int tiles_total = p.N * per_img;
    const int tpb = (tiles_total + (int)gridDim.x - 1) / (int)gridDim.x;
    const int wg = blockIdx.x;                           // statistics slab wg; tile ranges are contiguous in memory order
    const int tile_begin = wg * tpb, tile_end = min(tile_begin + tpb, tiles_total);
    if (tile_begin >= tile_end) return;                  // (the launcher sizes the grid so that this never happens)

    const unsigned char* __restrict__ x = (const unsigned char*)p.x;
    const bf16_t* __restrict__ w = (const bf16_t*)p.w;

    // ---- filter -> LDS once (row operand fragments: rows = output channels b*16 + fr, k-chunk ks*4 + fq)
    for (int c = tid; c < K * (KSTEPS * 4); c += NTHR) {
        const int k = c / (KSTEPS * 4), q = c - k * (KSTEPS * 4);
        *(i32x4*)(sW + k * WPITCH + q * 16) = *(const i32x4*)(w + (size_t)k * p.Kdp + q * 8);
    }
    const int boff = fr * WPITCH + fq * 16;

    // ---- patch loader: 1224 pieces / 512 threads
    int prel[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int pos = tid + NTHR * i;
        const int row = pos / (PROW / 16), c = pos - row * (PROW / 16);
        prel[i] = row * p.Win * 8 + c * 16;
    }
    auto issue_patch = [&](int t, int buf) {
        const int n = t / per_img, rem = t - n * per_img;
        const int th = rem / tiles_w, tw = rem - th * tiles_w;
        const unsigned char* org = x + (((long long)n * p.Hin + 2 * th * TOH) * p.Win + 2 * tw * TOW) * 8;
        unsigned char* dst = sP + buf * PATCH_B;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            if (tid + NTHR * i < PCHUNKS) glds16(org + prel[i], dst + (wave * 64 + NTHR * i) * 16);
        }
    };

    // ---- fragment offsets: this lane's k-chunk q = ks*4 + fq -> (filter row, column pair); wave w owns output rows 2w, 2w+1
    int aoff[KSTEPS];
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
        int q = ks * 4 + fq;
        if (q > 17) q = 17;                              // zero filter columns: any valid image bytes will do
        const int r = q / 3, part = q - 3 * r;
        aoff[ks] = (r * PWX + 2 * fr + 2 * part) * 8 + wave * (4 * PROW);
    }

    float sc[NT][4], sh[NT][4], s1[NT][4], s2[NT][4];
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c = b * 16 + fq * 4 + r;
            sc[b][r] = p.scale ? p.scale[c] : 1.0f;
            sh[b][r] = p.shift ? p.shift[c] : 0.0f;
            s1[b][r] = 0.f;
            s2[b][r] = 0.f;
        }
    // staging tile: pixel row of ROWB bytes in 8-byte slots; slot XOR keeps the 16 pixel lanes of a write (ds_write_b64: 16 consecutive lanes over 32 banks)
    // on distinct banks: all four row bits for 128-byte rows, (row >> 1) & 7 for 64-byte rows (see conv3x3.hip; odd keys swap a chunk's halves)
    const int sw_fr = K == 64 ? (fr & 15) : (K == 32 ? ((fr >> 1) & 7) : 0);
    int ep_off[NT];
#pragma unroll
    for (int b = 0; b < NT; ++b) ep_off[b] = fr * ROWB + (((b * 4 + fq) ^ sw_fr) << 3);

    issue_patch(tile_begin, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int cur = 0;
    for (int t = tile_begin; t < tile_end; ++t) {
        if (t + 1 < tile_end) issue_patch(t + 1, cur ^ 1);
        f32x4 acc[4][NT];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < NT; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
        const unsigned char* pb = sP + cur * PATCH_B;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            V16 af[4], bw[NT];
#pragma unroll
            for (int b = 0; b < NT; ++b) bw[b].i = *(const i32x4*)(sW + boff + b * 16 * WPITCH + ks * 64);
#pragma unroll
            for (int a = 0; a < 4; ++a)                  // segment a of the wave: output row 2w + (a >> 1), columns (a & 1)*16 + fr
                af[a].i = *(const i32x4*)(pb + aoff[ks] + (a >> 1) * (2 * PROW) + (a & 1) * (32 * 8));
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < NT; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bw[b].h, af[a].h, acc[a][b], 0, 0, 0);
        }
        if (p.stats) {
#pragma unroll
            for (int b = 0; b < NT; ++b)
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float v = acc[a][b][r];
                        s1[b][r] += v;
                        s2[b][r] = __builtin_fmaf(v, v, s2[b][r]);
                    }
        }
        // the staging tile is separate from the patches: no barrier needed before writing it, one before reading it back
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < NT; ++b) {
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v[r] = acc[a][b][r];
                    if (EPI >= 1) v[r] = v[r] * sc[b][r] + sh[b][r];
                    if (EPI == 2) v[r] = v[r] * __builtin_amdgcn_rcpf(1.0f + __expf(-v[r]));
                }
                bf16x4 o = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
                // tile pixel index = (2w + (a >> 1)) * 32 + (a & 1) * 16 + fr
                *(bf16x4*)(sS + ((wave * 2 + (a >> 1)) * TOW + (a & 1) * 16) * ROWB + ep_off[b]) = o;
            }
        __syncthreads();
        {
            const int n = t / per_img, rem = t - n * per_img;
            const int th = rem / tiles_w, tw = rem - th * tiles_w;
            bf16_t* yb = (bf16_t*)p.y + (((long long)n * p.Ho + th * TOH) * p.Wo + tw * TOW) * p.ldy;
#pragma unroll
            for (int j = 0; j < CPR; ++j) {
                const int c = tid + NTHR * j;
                const int px = c / CPR, ch = c - px * CPR;
                const int oy = px / TOW, ox = px - oy * TOW;
                const int key = K == 64 ? (px & 15) : (K == 32 ? ((px >> 1) & 7) : 0);
                i32x4 v = *(const i32x4*)(sS + px * ROWB + ((ch ^ (key >> 1)) << 4));
                if (key & 1) v = i32x4{v[2], v[3], v[0], v[1]};
                *(i32x4*)(yb + ((long long)oy * p.Wo + ox) * p.ldy + ch * 8) = v;
            }
        }
        // next patch (issued before this tile's CPR row stores) must have landed; the stores may stay in flight
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CPR) : "memory");
        __syncthreads();                                 // also: everyone is done with patch `cur` and with the staging tile
        cur ^= 1;
    }

    if (p.stats) {
        float* red = (float*)sS;                         // [8 waves][K][2]
#pragma unroll
        for (int b = 0; b < NT; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float u = s1[b][r], q = s2[b][r];
                u = row_sum16(u);
                q = row_sum16(q);
                if (fr == 0) {
                    const int c = b * 16 + fq * 4 + r;
                    red[(wave * K + c) * 2 + 0] = u;
                    red[(wave * K + c) * 2 + 1] = q;
                }
            }
        __syncthreads();
        if (tid < 2 * K) {
            const int which = tid / K, c = tid - which * K;
            float v = 0.f;
#pragma unroll
            for (int g = 0; g < 8; ++g) v += red[(g * K + c) * 2 + which];
            p.stats[((size_t)wg * 2 + which) * p.K + c] = v;
        }
    }
}

template <int NT>
int launch_nt(const ConvArgs& a, int grid, hipStream_t st) {
    const size_t smem = 2 * PATCH_B + (size_t)TOH * TOW * NT * 32 + (size_t)NT * 16 * 400;
    const int epi = a.act == 1 ? 2 : ((a.scale || a.shift) ? 1 : 0);
    static PerDeviceOnce attr_once;           // first launch of this instance on any thread
    attr_once.run([&] {
        (void)hipFuncSetAttribute((const void*)conv_stem_kernel<NT, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        (void)hipFuncSetAttribute((const void*)conv_stem_kernel<NT, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        (void)hipFuncSetAttribute((const void*)conv_stem_kernel<NT, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    });
    hdy_note_dispatch("conv_stem");
    if (epi == 2) hipLaunchKernelGGL((conv_stem_kernel<NT, 2>), dim3(grid), dim3(NTHR), smem, st, a);
    else if (epi == 1) hipLaunchKernelGGL((conv_stem_kernel<NT, 1>), dim3(grid), dim3(NTHR), smem, st, a);
    else hipLaunchKernelGGL((conv_stem_kernel<NT, 0>), dim3(grid), dim3(NTHR), smem, st, a);
    return 0;
}

}  // namespace

static bool stem_shape_ok(int K, int Ho, int Wo, int dtype) {
    const bool disabled = hdy_opt(HDY_OPT_NO_STEM_KERNEL) != 0;   // tests: force the generic kernel for A/B comparison
    return !disabled && dtype == HDY_BF16 && K % 16 == 0 && K >= 16 && K <= 64 && Ho % TOH == 0 && Wo % TOW == 0;
}

// grid: every workgroup gets the same number of tiles where possible (two 72 KB workgroups per CU at K = 32)
static int stem_grid(int tiles, int K) {
    const int per_cu = (2 * PATCH_B + TOH * TOW * K * 2 + K * 400) * 2 <= 160 * 1024 ? 2 : 1;
    int grid = 256 * per_cu;
    if (grid > tiles) grid = tiles;
    const int tpb = (tiles + grid - 1) / grid;
    return (tiles + tpb - 1) / tpb;                      // no empty workgroups: each owns a statistics slab
}

// Statistic slabs the stem kernel writes for an H x W image batch (0 = not eligible).
int hdy_conv_stem_slabs(int N, int H, int W, int K, int dtype) {
    const int Ho = H / 2, Wo = W / 2;
    if (H % 2 || W % 2 || !stem_shape_ok(K, Ho, Wo, dtype)) return 0;
    return stem_grid(N * (Ho / TOH) * (Wo / TOW), K);
}

// Returns 1 and launches when the stem shape qualifies; 0 = not eligible (generic kernel runs).
int hdy_conv_stem_try(const ConvArgs& a, int dtype, int out_f32, hipStream_t st, int* rc) {
    if (!a.span_pixels || out_f32 || a.res || a.accumulate || !a.dense_out || a.act > 1) return 0;
    if (!(a.TH == 6 && a.TW == 1 && a.C == 24 && a.ldx == 4 && a.ih_mul == 2 && a.iw_mul == 2 && a.Kdp >= 160)) return 0;
    if (!(a.Hin == 2 * a.Ho + 4 && a.Win == 2 * a.Wo + 4 && stem_shape_ok(a.K, a.Ho, a.Wo, dtype))) return 0;
    const bool aligned = a.ldy % 8 == 0 && ((uintptr_t)a.y & 15) == 0 && ((uintptr_t)a.x & 15) == 0 && ((uintptr_t)a.w & 15) == 0;
    if (!aligned) {
        if (!a.stats) return 0;
        hdy_set_error("conv_stem: statistics requested but y rows are not 16-byte aligned (ldy=%d)", a.ldy);
        *rc = HDY_EINVAL;
        return 1;
    }
    const int grid = stem_grid(a.N * (a.Ho / TOH) * (a.Wo / TOW), a.K);
    HDY_STAT_CAP(a, grid, "conv_stem")
    switch (a.K / 16) {
        case 1: launch_nt<1>(a, grid, st); break;
        case 2: launch_nt<2>(a, grid, st); break;
        case 3: launch_nt<3>(a, grid, st); break;
        default: launch_nt<4>(a, grid, st); break;
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        hdy_set_error("conv_stem: launch failed: %s", hipGetErrorString(e));
        *rc = (int)e;
        return 1;
    }
    *rc = HDY_OK;
    return 1;
}

// 3x3 / stride 1 / pad 1 convolution for 64 input channels (bf16), weights resident in LDS — the kernel behind the
// layer BASELINE.json names ("fused 3x3 conv at batch 64 x 640 x 640": yolov5s' 64->64 @80x80) and its dgrad.
//
// Why a second conv kernel: the generic implicit GEMM (conv_igemm.hip) fetches the A operand once per tap — 9x the
// activation bytes from L2 per output tile, plus the 73 KB filter per tile.  At ~2 us of L2 latency and 64 KB in flight per
// CU that caps it near 9 TB/s of L2->LDS traffic, i.e. ~350 TFLOP/s on this layer whatever the MFMA schedule does.
// Here each workgroup
//   * keeps the whole filter (9 taps x 64 x 64 bf16 = 72 KB) in LDS for its lifetime (persistent over ~12 tiles),
//   * stages one (16+2) x (16+2) input patch (41 KB) per 16x16 output tile by LDS-DMA, double buffered across tiles,
//   * and reads all 9 taps' A fragments out of that patch: L2->LDS traffic per 256 outputs drops from 440 KB to 41 KB;
//   * runs 8 waves (2 per SIMD) so one wave's fragment reads / address VALU overlap the other's MFMAs.
// Fragment addressing: lane row = output pixel (ty, tx) of the tile, tap (r, s) reads patch pixel (ty+r)*18 + tx+s; the
// 16 lanes of a fragment read 16 consecutive patch pixels, and chunk ^ ((patch column>>1)&7) spreads them over all 16
// 16-byte slots of the 256-byte bank row (conflict-free ds_read_b128).  LDS-DMA writes linearly, so that XOR is applied to the
// source address (rule: linear destination + permuted source + permuted read).
// Epilogue, BatchNorm slabs and numerics are those of conv_igemm.hip (same 16x16x32 bf16 MFMA tiling, 2x2 waves).
//
// Requirements (checked by the launcher, otherwise the generic kernel runs): bf16 in/out, C == 64, K <= 64, R = S = 3,
// stride 1, pad 1, H % 16 == 0, W % 16 == 0, 16-byte aligned rows.
#include <stdlib.h>

#include "common.h"
#include "hdyolo_internal.h"

__device__ uint4 g_hdy_zero16_c3[4];   // zero page for out-of-image patch pixels

namespace {

constexpr int NTHR = 512;                                                // 8 waves = 4 (tile-row groups) x 2 (channel halves)
constexpr int TH = 16, TW = 16, PW = TW + 2, PPIX = (TH + 2) * PW;       // 324 patch pixels
constexpr int PATCH_B = PPIX * 128;                                      // 41472
constexpr int W_B = 9 * 64 * 128;                                        // 73728
constexpr int SMEM_B = W_B + 2 * PATCH_B;                                // 156672 of 163840

__device__ __forceinline__ void glds16(const void* g, unsigned char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)g,
                                     (void __attribute__((address_space(3)))*)lds_wave_base, 16, 0, 0);
}

__global__ __launch_bounds__(NTHR) void conv3x3_c64_kernel(const ConvArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sW = smem;                    // [9][64][128 B]
    unsigned char* sP = smem + W_B;              // [2][180][128 B]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 15, fq = lane >> 4;
    const int tiles_w = p.Wo / TW, tiles_h = p.Ho / TH;
    const int tiles_img = tiles_w * tiles_h;
    const int tiles_total = p.N * tiles_img;
    const int tpb = (tiles_total + (int)gridDim.x - 1) / (int)gridDim.x;
    const int tile_begin = xcd_remap(blockIdx.x, gridDim.x) * tpb;
    const int tile_end = min(tile_begin + tpb, tiles_total);
    if (tile_begin >= tile_end) return;

    const bf16_t* __restrict__ x = (const bf16_t*)p.x;
    const bf16_t* __restrict__ w = (const bf16_t*)p.w;
    const unsigned char* zero = (const unsigned char*)g_hdy_zero16_c3;
    const int lc = (tid & 7) ^ ((tid >> 4) & 7);          // logical chunk for physical slot tid & 7 (rows differ by 32*i)

    // ---- filter: 9 taps x 64 rows, once per workgroup.  LDS row (tap*64 + k) <- packed w[k][tap*64 .. +63]
#pragma unroll
    for (int i = 0; i < 9; ++i) {                          // 576 rows * 8 chunks / 512 threads
        const int pos = tid + NTHR * i;
        const int row = pos >> 3;                          // tap*64 + k
        const int tap = row >> 6, k = row & 63;
        const void* src = (k < p.K) ? (const void*)(w + (size_t)k * p.Kdp + tap * 64 + lc * 8) : (const void*)zero;
        glds16(src, sW + (wave * 64 + NTHR * i) * 16);
    }

    auto issue_patch = [&](int t, int buf) {
        const int n = t / tiles_img, rem = t - n * tiles_img;
        const int th_ = rem / tiles_w, tw_ = rem - th_ * tiles_w;
        const int h0 = th_ * TH - 1, w0 = tw_ * TW - 1;
        unsigned char* dst = sP + buf * PATCH_B;
#pragma unroll
        for (int i = 0; i < 6; ++i) {                      // 2592 chunks / 512 threads (last pass partial)
            const int pos = tid + NTHR * i;
            if (pos < PPIX * 8) {
                const int pix = pos >> 3;
                const int py = pix / PW, px = pix - py * PW;
                const int h = h0 + py, ww = w0 + px;
                const int lcp = (tid & 7) ^ ((px >> 1) & 7);      // patch swizzle keys on the COLUMN (see the fragment reads)
                const void* src = zero;
                if ((unsigned)h < (unsigned)p.Hin && (unsigned)ww < (unsigned)p.Win)
                    src = x + (((size_t)n * p.Hin + h) * p.Win + ww) * p.ldx + lcp * 8;
                glds16(src, dst + (wave * 64 + NTHR * i) * 16);
            }
        }
    };

    // Per-lane LDS byte offsets, computed once: with the patch swizzle keyed on the patch column (fr + s) and the filter
    // swizzle on the filter row, every fragment address is lane_offset + compile-time constant, so the 108 reads per tile
    // carry no address VALU at all (they were ~25 % of the issue slots of the MFMA loop).
    int aoff[3][2], boff[2][2];
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int px = fr + s;
            aoff[s][ks] = (wm * 4 * PW + px) * 128 + (((ks * 4 + fq) ^ ((px >> 1) & 7)) << 4);
        }
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int k = wn * 32 + b * 16 + fr;
            boff[b][ks] = k * 128 + (((ks * 4 + fq) ^ ((k >> 1) & 7)) << 4);
        }

    f32x4 acc[4][2];
    const int HoWo = p.Ho * p.Wo;
    issue_patch(tile_begin, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int cur = 0;
    for (int t = tile_begin; t < tile_end; ++t) {
        if (t + 1 < tile_end) issue_patch(t + 1, cur ^ 1);
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
        const unsigned char* pb = sP + cur * PATCH_B;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int r = tap / 3, s = tap - 3 * r;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                V16 af[4], bf[2];
#pragma unroll
                for (int a = 0; a < 4; ++a)      // tile row wm*4 + a, patch row + r: a compile-time offset from the lane base
                    af[a].i = *(const i32x4*)(pb + aoff[s][ks] + (a + r) * PW * 128);
#pragma unroll
                for (int b = 0; b < 2; ++b) bf[b].i = *(const i32x4*)(sW + boff[b][ks] + tap * 64 * 128);
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[a].h, bf[b].h, acc[a][b], 0, 0, 0);
            }
        }
        __syncthreads();                                   // everyone is done with patch `cur`: it becomes scratch
        unsigned char* scratch = sP + cur * PATCH_B;
        const int n = t / tiles_img, rem = t - n * tiles_img;
        const int th_ = rem / tiles_w, tw_ = rem - th_ * tiles_w;
        if (p.stats) {
            float* red = (float*)scratch;                  // [4 (wm)][64][2]
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float v = acc[a][b][r];
                        s1 += v;
                        s2 += v * v;
                    }
                s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16);
                s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
                if (lane < 16) {
                    const int col = wn * 32 + b * 16 + lane;
                    red[(wm * 64 + col) * 2 + 0] = s1;
                    red[(wm * 64 + col) * 2 + 1] = s2;
                }
            }
            __syncthreads();
            if (tid < 128 && (tid & 63) < p.K) {
                // one slab per 128 output pixels (the caller sized the slab array as M / 128): rows 0-7 and 8-15 of the tile
                const int half = tid >> 6, c = tid & 63;
                const size_t slab = (size_t)t * 2 + half;
                p.stats[(slab * 2 + 0) * p.K + c] = red[((2 * half) * 64 + c) * 2] + red[((2 * half + 1) * 64 + c) * 2];
                p.stats[(slab * 2 + 1) * p.K + c] = red[((2 * half) * 64 + c) * 2 + 1] + red[((2 * half + 1) * 64 + c) * 2 + 1];
            }
            __syncthreads();
        }
        // ---- epilogue: scale/shift/act -> bf16 tile in LDS ([128 pixels][64 ch], 32-byte blocks XORed) -> 16-byte row stores
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int col = wn * 32 + b * 16 + fr;
            const bool okk = col < p.K;
            const float sc = (p.scale && okk) ? p.scale[col] : 1.0f;
            const float sh = (p.shift && okk) ? p.shift[col] : 0.0f;
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = (wm * 4 + a) * 16 + fq * 4 + r;         // tile pixel index ty*16 + tx
                    float v = acc[a][b][r] * sc + sh;
                    if (p.act == 1) v = silu_f(v);
                    const int chunk = (col >> 3) ^ (((row >> 2) & 3) << 1);
                    *(bf16_t*)(scratch + row * 128 + chunk * 16 + (col & 7) * 2) = (bf16_t)v;
                }
        }
        __syncthreads();
        {
            const int ch = tid & 7, rr = tid >> 3;            // 8 chunks per row, 64 rows per pass
            const int kc = ch * 8;
            if (kc < p.K) {
#pragma unroll
                for (int row = rr; row < TH * TW; row += NTHR / 8) {
                    const int ty = row >> 4, tx = row & 15;
                    const size_t opix = ((size_t)n * p.Ho + th_ * TH + ty) * p.Wo + tw_ * TW + tx;
                    const int chunk = ch ^ (((row >> 2) & 3) << 1);
                    V16 v;
                    v.i = *(const i32x4*)(scratch + row * 128 + chunk * 16);
                    if (p.res || p.accumulate) {
                        float f[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) f[e] = (float)v.h[e];
                        if (p.res) {
                            V16 q;
                            q.i = *(const i32x4*)((const bf16_t*)p.res + opix * p.ldr + kc);
#pragma unroll
                            for (int e = 0; e < 8; ++e) f[e] += (float)q.h[e];
                        }
                        if (p.accumulate) {
                            V16 q;
                            q.i = *(const i32x4*)((const bf16_t*)p.y + opix * p.ldy + kc);
#pragma unroll
                            for (int e = 0; e < 8; ++e) f[e] += (float)q.h[e];
                        }
#pragma unroll
                        for (int e = 0; e < 8; ++e) v.h[e] = (bf16_t)f[e];
                    }
                    *(i32x4*)((bf16_t*)p.y + opix * p.ldy + kc) = v.i;
                }
            }
        }
        // the next patch's DMA was issued before this tile's 4 row stores, and vmcnt retires in issue order: leaving the 4
        // stores in flight (instead of vmcnt(0)) keeps their ~2 us write latency off the critical path
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        __syncthreads();
        cur ^= 1;
    }
    (void)HoWo;
}

}  // namespace

// Returns 1 and launches when the shape qualifies; 0 = not eligible (caller falls back to the generic kernel); <0 / >0 = error.
int hdy_conv3x3_c64_try(const ConvArgs& a, int dtype, int out_f32, hipStream_t st, int* rc) {
    if (dtype != HDY_BF16 || out_f32) return 0;
    static const bool disabled = getenv("HDY_NO_CONV3X3") != nullptr;      // tests: force the generic kernel for A/B comparison
    if (disabled) return 0;
    if (!(a.TH == 3 && a.TW == 3 && a.ih_mul == 1 && a.iw_mul == 1 && a.dh0 == -1 && a.dw0 == -1 && a.dense_out)) return 0;
    if (!(a.C == 64 && a.K <= 64 && a.K % 8 == 0 && a.Hin == a.Ho && a.Win == a.Wo && a.Ho % TH == 0 && a.Wo % TW == 0)) return 0;
    if (!(a.ldx % 8 == 0 && a.ldy % 8 == 0 && ((uintptr_t)a.y & 15) == 0 && ((uintptr_t)a.x & 15) == 0)) return 0;
    if (a.res && !(a.ldr % 8 == 0 && ((uintptr_t)a.res & 15) == 0)) return 0;
    if (a.span_pixels) return 0;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)conv3x3_c64_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_B);
        attr_set = true;
    }
    const int tiles = a.N * (a.Ho / TH) * (a.Wo / TW);
    int grid = 256;                                        // one 153 KB, 8-wave workgroup per CU
    if (grid > tiles) grid = tiles;
    hipLaunchKernelGGL(conv3x3_c64_kernel, dim3(grid), dim3(NTHR), SMEM_B, st, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        hdy_set_error("conv3x3_c64: launch failed: %s", hipGetErrorString(e));
        *rc = (int)e;
        return 1;
    }
    *rc = HDY_OK;
    return 1;
}

// 3x3 / stride 1 / pad 1 convolution for 64 input channels (bf16), weights resident in LDS — the kernel behind the
// layer BASELINE.json names ("fused 3x3 conv at batch 64 x 640 x 640": yolov5s' 64->64 @80x80) and its dgrad.
//
// Why a second conv kernel: the generic implicit GEMM (conv_igemm.hip) fetches the A operand once per tap — 9x the
// activation bytes from L2 per output tile, plus the 73 KB filter per tile.  At ~2 us of L2 latency and 64 KB in flight per
// CU that caps it near 9 TB/s of L2->LDS traffic, i.e. ~350 TFLOP/s on this layer whatever the MFMA schedule does.
// Here each workgroup
//   * keeps the whole filter (9 taps x 64 x 64 bf16 = 72 KB) in LDS for its lifetime (persistent over ~6 tiles),
//   * stages one (16+2) x (16+2) input patch (41 KB) per 16x16 output tile by LDS-DMA, double buffered across tiles,
//   * and reads all 9 taps' A fragments out of that patch: L2->LDS traffic per 256 outputs drops from 440 KB to 41 KB;
//   * runs 8 waves (2 per SIMD) so one wave's fragment reads / address VALU overlap the other's MFMAs.
// Fragment addressing: lane row = output pixel (ty, tx) of the tile, tap (r, s) reads patch pixel (ty+r)*18 + tx+s; the
// 16 lanes of a fragment read 16 consecutive patch pixels, and chunk ^ (((patch column>>1)&3)<<1) spreads them over the 16
// 16-byte slots of the 256-byte bank row.  ds_read_b128 is served in groups of 16 lanes that MIX two k-chunks (lanes 0-3, 12-15 of
// chunk c with lanes 4-11 of chunk c+1): keying on bits 1-2 only leaves bit 0 to tell the two chunks apart, which is conflict-free
// for every tap shift (the first key, (column>>1)&7, collided at odd shifts: 25 % of the LDS cycles were conflict cycles).  LDS-DMA writes linearly, so that XOR is applied to the
// source address (rule: linear destination + permuted source + permuted read).
// The MFMAs run with the FILTER as the row operand, so a lane's 4 accumulator values are 4 consecutive output channels of ONE
// pixel: the epilogue packs them into one 8-byte LDS write (instead of four 2-byte scatters) and the BatchNorm sums stay in
// registers across all of the workgroup's tiles (one statistics slab per workgroup).
// Work split: every workgroup gets floor(tiles / grid) whole tiles; the remaining tiles are cut into 2 or 4 row bands so that
// they still spread over all CUs (1600 tiles on 256 CUs: 6 tiles + one 4-row band each, instead of 229 CUs x 7 tiles).
// Measured and not adopted: 8-byte stores straight from the accumulators (no staging tile, one barrier per item): 43.3 us vs 39.9 us.
// Measured per-tile phases (shader clocks, 16x16 tile): patch DMA issue ~1.1k, MFMA ~4.7k (at the MFMA rate for 2 waves per
// SIMD), epilogue ~0.5k, row stores ~0.6k, barriers ~0.5k.
//
// Requirements (checked by the launcher, otherwise the generic kernel runs): bf16 in/out, C == 64, K <= 64, R = S = 3,
// stride 1, pad 1, H % 16 == 0, W % 16 == 0, 16-byte aligned rows.
#include <stdlib.h>

#include <type_traits>

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

// The split of `tiles` whole tiles over `grid` workgroups, computed identically on host and device.
struct WorkSplit {
    int per;      // whole tiles per workgroup: workgroup g owns tiles [g*per, (g+1)*per)
    int bands;    // the `rest` tiles after those are cut into this many row bands each (1, 2 or 4)
    int units;    // rest * bands; workgroup g < units also owns band g % bands of tile per*grid + g / bands
};
__host__ __device__ inline WorkSplit work_split(int tiles, int grid) {
    WorkSplit s;
    s.per = tiles / grid;
    const int rest = tiles - s.per * grid;
    s.bands = rest * 4 <= grid ? 4 : (rest * 2 <= grid ? 2 : 1);
    s.units = rest * s.bands;
    return s;
}

struct Item { int n, th, tw, row0, nrows; };   // rows [row0, row0 + nrows) of tile (n, th, tw); nrows in {16, 8, 4}

// EPI: 0 = raw convolution out (train-mode forward, dgrad), 1 = scale/shift, 2 = scale/shift + SiLU
template <int EPI>
__global__ __launch_bounds__(NTHR) void conv3x3_c64_kernel(const ConvArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sW = smem;                    // [9][64][128 B]
    unsigned char* sP = smem + W_B;              // [2][324][128 B]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 15, fq = lane >> 4;
    const int tiles_w = p.Wo / TW, tiles_h = p.Ho / TH;
    const int tiles_total = p.N * tiles_w * tiles_h;
    const int wg = xcd_remap(blockIdx.x, gridDim.x);       // logical workgroup: owns statistics slab wg
    const WorkSplit ws = work_split(tiles_total, (int)gridDim.x);
    const int nitems = ws.per + (wg < ws.units ? 1 : 0);   // >= 1: the launcher never starts more workgroups than tiles

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

    // ---- patch loader: each thread's 6 (pixel, chunk) slots of the 18-wide patch are the same for every item; only the
    // origin, the number of patch rows and the image-border test change: a compare + select + add per DMA.
    // (Issuing the next patch piecewise from inside the MFMA loop was measured slower: 45.1 / 43.5 us against 42.9 us per
    // launch with all DMAs at the top of the item — the loads land later, and an in-order wave cannot issue MFMAs past a DMA
    // that is waiting for a queue slot.)
    int prel[6], pyx[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {                          // 2592 chunks / 512 threads (last pass partial)
        const int pos = tid + NTHR * i;
        const int pix = pos >> 3;
        const int py = pix / PW, px = pix - py * PW;
        const int lcp = (tid & 7) ^ (((px >> 1) & 3) << 1);   // patch swizzle keys on the COLUMN (see the fragment reads)
        prel[i] = (py * p.Win + px) * p.ldx + lcp * 8;
        pyx[i] = py | (px << 8);
    }
    auto item_at = [&](int idx) {
        Item c;
        int t;
        if (idx < ws.per) {
            t = wg * ws.per + idx;
            c.row0 = 0;
            c.nrows = TH;
        } else {
            t = ws.per * (int)gridDim.x + wg / ws.bands;
            c.nrows = TH / ws.bands;
            c.row0 = (wg % ws.bands) * c.nrows;
        }
        const int per_img = tiles_w * tiles_h;
        c.n = t / per_img;
        const int rem = t - c.n * per_img;
        c.th = rem / tiles_w;
        c.tw = rem - c.th * tiles_w;
        return c;
    };
    auto issue_patch = [&](const Item& c, int buf) {
        const int h0 = c.th * TH + c.row0 - 1, w0 = c.tw * TW - 1;
        const int prow = c.nrows + 2;
        const bf16_t* org = x + (((long long)c.n * p.Hin + h0) * p.Win + w0) * p.ldx;
        unsigned char* dst = sP + buf * PATCH_B;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int py = pyx[i] & 255;
            // wave-uniform cut: the patch is filled in pixel order, so whole 1 KB pieces beyond its last row are skipped
            if (((wave * 64 + NTHR * i) >> 3) >= prow * PW) break;
            if (py >= prow) continue;                      // lanes past the patch's last pixel stay masked: no LDS write at all
            const int h = h0 + py, ww = w0 + (pyx[i] >> 8);
            const void* src = ((unsigned)h < (unsigned)p.Hin && (unsigned)ww < (unsigned)p.Win) ? (const void*)(org + prel[i]) : (const void*)zero;
            glds16(src, dst + (wave * 64 + NTHR * i) * 16);
        }
    };

    // Per-lane LDS byte offsets, computed once: with the patch swizzle keyed on the patch column (fr + s) and the filter
    // swizzle on the filter row, every fragment address is lane_offset + compile-time constant, so the fragment reads carry no
    // address VALU at all.  (The wave's first patch row, wm * rows-per-wave, is added to the wave-uniform base pointer.)
    int aoff[3][2], boff[2][2];
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int px = fr + s;
            aoff[s][ks] = px * 128 + (((ks * 4 + fq) ^ (((px >> 1) & 3) << 1)) << 4);
        }
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int k = wn * 32 + b * 16 + fr;
            boff[b][ks] = k * 128 + (((ks * 4 + fq) ^ ((k >> 1) & 7)) << 4);
        }

    float sc[2][4], sh[2][4], s1[2][4], s2[2][4];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c = wn * 32 + b * 16 + fq * 4 + r;
            sc[b][r] = (p.scale && c < p.K) ? p.scale[c] : 1.0f;
            sh[b][r] = (p.shift && c < p.K) ? p.shift[c] : 0.0f;
            s1[b][r] = 0.f;
            s2[b][r] = 0.f;
        }
    // epilogue staging: item pixel row -> 128 bytes; 8-byte slot XOR (row & 14) makes the 16 lanes of a write hit 16 different
    // slots (rows of equal slot differ in parity = in 128-byte half of the bank line); the 16-byte read-back sees whole chunks
    const int st_ch = tid & 7, st_rr = tid >> 3;                              // store phase: chunk, first row (rows + 64 j)
    const int st_lds = st_rr * 128 + ((st_ch ^ ((st_rr >> 1) & 7)) << 4);
    const long long st_off = ((long long)(st_rr >> 4) * p.Wo + (st_rr & 15)) * p.ldy + st_ch * 8;
    const long long st_step = (long long)4 * p.Wo * p.ldy;
    const long long rs_off = ((long long)(st_rr >> 4) * p.Wo + (st_rr & 15)) * p.ldr + st_ch * 8;
    const long long rs_step = (long long)4 * p.Wo * p.ldr;
    int ep_off[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) ep_off[b] = fr * 128 + (((wn * 8 + b * 4 + fq) ^ (fr & 14)) << 3);

    // ---- one item: AR = tile rows per wave (the item has 4*AR rows, wave wm owns rows wm*AR .. wm*AR + AR-1)
    auto compute = [&](auto ar_c, const Item& it, int cur) {
        constexpr int AR = decltype(ar_c)::value;
        f32x4 acc[AR][2];
#pragma unroll
        for (int a = 0; a < AR; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
        const unsigned char* pb = sP + cur * PATCH_B + wm * AR * PW * 128;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int r = tap / 3, s = tap - 3 * r;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                V16 af[AR], bf[2];
#pragma unroll
                for (int a = 0; a < AR; ++a)     // item row wm*AR + a, patch row + r: a compile-time offset from the lane base
                    af[a].i = *(const i32x4*)(pb + aoff[s][ks] + (a + r) * PW * 128);
#pragma unroll
                for (int b = 0; b < 2; ++b) bf[b].i = *(const i32x4*)(sW + boff[b][ks] + tap * 64 * 128);
#pragma unroll
                for (int a = 0; a < AR; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[b].h, af[a].h, acc[a][b], 0, 0, 0);
            }
        }
        __syncthreads();                                   // everyone is done with patch `cur`: it becomes scratch
        unsigned char* scratch = sP + cur * PATCH_B;
        if (p.stats) {
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int a = 0; a < AR; ++a)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float v = acc[a][b][r];
                        s1[b][r] += v;
                        s2[b][r] = __builtin_fmaf(v, v, s2[b][r]);
                    }
        }
#pragma unroll
        for (int a = 0; a < AR; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v[r] = acc[a][b][r];
                    if (EPI >= 1) v[r] = v[r] * sc[b][r] + sh[b][r];
                    if (EPI == 2) v[r] = v[r] * __builtin_amdgcn_rcpf(1.0f + __expf(-v[r]));     // SiLU; 1 ulp, then rounded to bf16
                }
                bf16x4 o = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
                *(bf16x4*)(scratch + ep_off[b] + (wm * AR + a) * 16 * 128) = o;
            }
        __syncthreads();
        if (st_ch * 8 < p.K) {
            const long long org = (((long long)it.n * p.Ho + it.th * TH + it.row0) * p.Wo + it.tw * TW);
            bf16_t* yb = (bf16_t*)p.y + org * p.ldy + st_off;
            const bf16_t* rb = p.res ? (const bf16_t*)p.res + org * p.ldr + rs_off : nullptr;
#pragma unroll
            for (int j = 0; j < AR; ++j) {                 // 64 pixel rows of 128 bytes per pass
                V16 v;
                v.i = *(const i32x4*)(scratch + st_lds + j * 64 * 128);
                if (p.res || p.accumulate) {
                    float f[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] = (float)v.h[e];
                    if (p.res) {
                        V16 q;
                        q.i = *(const i32x4*)(rb + j * rs_step);
#pragma unroll
                        for (int e = 0; e < 8; ++e) f[e] += (float)q.h[e];
                    }
                    if (p.accumulate) {
                        V16 q;
                        q.i = *(const i32x4*)(yb + j * st_step);
#pragma unroll
                        for (int e = 0; e < 8; ++e) f[e] += (float)q.h[e];
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) v.h[e] = (bf16_t)f[e];
                }
                *(i32x4*)(yb + j * st_step) = v.i;
            }
        }
        // the next patch's DMA was issued before this item's AR row stores, and vmcnt retires in issue order: leaving the
        // stores in flight (instead of vmcnt(0)) keeps their ~2 us write latency off the critical path
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AR) : "memory");
        __syncthreads();
    };

    Item cur_it = item_at(0);
    issue_patch(cur_it, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int cur = 0;
    for (int idx = 0; idx < nitems; ++idx) {
        Item nxt_it = cur_it;
        if (idx + 1 < nitems) {
            nxt_it = item_at(idx + 1);
            issue_patch(nxt_it, cur ^ 1);
        }
        if (cur_it.nrows == 16) compute(std::integral_constant<int, 4>{}, cur_it, cur);
        else if (cur_it.nrows == 8) compute(std::integral_constant<int, 2>{}, cur_it, cur);
        else compute(std::integral_constant<int, 1>{}, cur_it, cur);
        cur ^= 1;
        cur_it = nxt_it;
    }

    if (p.stats) {
        // one slab per workgroup: 16 pixel lanes -> 4 tile-row groups (waves) -> global
        float* red = (float*)sP;                           // [4 (wm)][64][2]; every patch buffer is idle after the last barrier
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float u = s1[b][r], q = s2[b][r];
#pragma unroll
                for (int m = 1; m < 16; m <<= 1) {
                    u += __shfl_xor(u, m);
                    q += __shfl_xor(q, m);
                }
                if (fr == 0) {
                    const int c = wn * 32 + b * 16 + fq * 4 + r;
                    red[(wm * 64 + c) * 2 + 0] = u;
                    red[(wm * 64 + c) * 2 + 1] = q;
                }
            }
        __syncthreads();
        if (tid < 128 && (tid & 63) < p.K) {
            const int which = tid >> 6, c = tid & 63;
            float v = 0.f;
#pragma unroll
            for (int g = 0; g < 4; ++g) v += red[(g * 64 + c) * 2 + which];
            p.stats[((size_t)wg * 2 + which) * p.K + c] = v;
        }
    }
}

}  // namespace

// Shape test shared by the launcher and the statistics-slab query (the two must agree on who writes the slabs).
static bool conv3x3_shape_ok(int C, int K, int R, int S, int stride, int pad, int H, int W, int dtype) {
    static const bool disabled = getenv("HDY_NO_CONV3X3") != nullptr;      // tests: force the generic kernel for A/B comparison
    return !disabled && dtype == HDY_BF16 && R == 3 && S == 3 && stride == 1 && pad == 1 && C == 64 && K <= 64 && K % 8 == 0 && H % TH == 0 &&
           W % TW == 0;
}

static int conv3x3_grid(int tiles) {
    static const int g = getenv("HDY_C3_GRID") ? atoi(getenv("HDY_C3_GRID")) : 256;
    return tiles < g ? tiles : g;
}     // one 153 KB, 8-wave workgroup per CU

// Number of statistic slabs the filter-resident kernel writes for this shape (one per workgroup), 0 = not eligible.
int hdy_conv3x3_c64_slabs(int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dtype) {
    if (!conv3x3_shape_ok(C, K, R, S, stride, pad, H, W, dtype)) return 0;
    return conv3x3_grid(N * (H / TH) * (W / TW));
}

// Returns 1 and launches when the shape qualifies; 0 = not eligible (caller falls back to the generic kernel); <0 / >0 = error.
int hdy_conv3x3_c64_try(const ConvArgs& a, int dtype, int out_f32, hipStream_t st, int* rc) {
    if (dtype != HDY_BF16 || out_f32 || a.act > 1) return 0;
    if (!(a.TH == 3 && a.TW == 3 && a.ih_mul == 1 && a.iw_mul == 1 && a.dh0 == -1 && a.dw0 == -1 && a.dense_out && !a.span_pixels)) return 0;
    if (!(a.Hin == a.Ho && a.Win == a.Wo && conv3x3_shape_ok(a.C, a.K, 3, 3, 1, 1, a.Ho, a.Wo, dtype))) return 0;
    const bool aligned = a.ldx % 8 == 0 && a.ldy % 8 == 0 && ((uintptr_t)a.y & 15) == 0 && ((uintptr_t)a.x & 15) == 0 &&
                         (!a.res || (a.ldr % 8 == 0 && ((uintptr_t)a.res & 15) == 0));
    if (!aligned) {
        if (!a.stats) return 0;
        // the caller sized the slab array with hdy_conv_stat_slabs for THIS kernel: falling back would write a different count
        hdy_set_error("conv3x3_c64: statistics requested but x/y/res rows are not 16-byte aligned (ldx=%d ldy=%d)", a.ldx, a.ldy);
        *rc = HDY_EINVAL;
        return 1;
    }
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)conv3x3_c64_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_B);
        (void)hipFuncSetAttribute((const void*)conv3x3_c64_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_B);
        (void)hipFuncSetAttribute((const void*)conv3x3_c64_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_B);
        attr_set = true;
    }
    const int tiles = a.N * (a.Ho / TH) * (a.Wo / TW);
    const int grid = conv3x3_grid(tiles);
    const int epi = a.act == 1 ? 2 : ((a.scale || a.shift) ? 1 : 0);
    if (epi == 2) hipLaunchKernelGGL(conv3x3_c64_kernel<2>, dim3(grid), dim3(NTHR), SMEM_B, st, a);
    else if (epi == 1) hipLaunchKernelGGL(conv3x3_c64_kernel<1>, dim3(grid), dim3(NTHR), SMEM_B, st, a);
    else hipLaunchKernelGGL(conv3x3_c64_kernel<0>, dim3(grid), dim3(NTHR), SMEM_B, st, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        hdy_set_error("conv3x3_c64: launch failed: %s", hipGetErrorString(e));
        *rc = (int)e;
        return 1;
    }
    *rc = HDY_OK;
    return 1;
}

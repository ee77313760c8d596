// Patch-resident weight gradient of the 3x3 / stride 1 / pad 1 convolution for gfx950 (bf16).
//
//   dW[k][r][s][c] = SUM_{n,i,j} dy[n,i,j,k] * x[n, i+r-1, j+s-1, c]
//
// The generic kernel (conv_wgrad.hip) treats the nine taps as nine times as many GEMM columns and re-gathers x from L2 for each of
// them, and dy once per column tile: 480 KB of L2->LDS traffic per 256 output pixels of a 64x64 layer, which is what bounds it
// (0.13-0.20 of the layer's roofline on every 3x3 layer of yolov5s).  Here a workgroup owns a (KB dy-channels x CB x-channels) block
// of the gradient for ALL NINE taps (accumulators: 9 * KB*CB fp32 = 144 VGPRs per lane at 64x64) and walks output tiles of TOH x TOW
// <= 256 pixels: the dy tile [pixels][KB] and the (TOH+2) x (TOW+2) input patch [pixels][CB] land in LDS once (LDS-DMA, 75 KB per
// tile), and the MFMA operands of every tap are transposed reads (ds_read_b64_tr_b16, reduction index = pixel) of the SAME two images
// — the tap only shifts the patch row a lane addresses.  A lane's eight reduction pixels need not be neighbours (each lane passes its
// own address), so a 32-pixel MFMA step simply takes tile pixels 32*step .. +31 in row-major order, whatever the tile shape; the tile
// shape is chosen per layer so that whole image rows / 16x16 blocks fit (20x20, 40x40, 80x80, 160x160 all tile without remainder
// in one direction).
// Output: one fp32 slab [K][9*C] per spatial split (q = tap*C + c, the layout hdy_wgrad_reduce_launch scatters from), summed in fixed
// order (deterministic).  Splits are capped so that the slabs stay within a few times the gradient itself.
#include <stdlib.h>

#include "common.h"
#include "hdyolo_internal.h"
#include "hdyolo.h"

__device__ uint4 g_hdy_zero16_w3[4];

namespace {

struct W3Args {
    const void* x; int ldx;      // [N][H][W][ldx], C channels
    const void* dy; int lddy;    // [N][H][W][lddy], K channels
    float* partial;              // [nsplit][K][9*C]
    int N, H, W, C, K;           // H, W: OUTPUT (dy) size
    int Hin, Win, S;             // input size, stride (1 or 2; pad 1)
    int TOH, TOW, tiles_h, tiles_w;
    int nkb, ncb, nsplit;
};

__device__ __forceinline__ int fsw3(int row) { return (row & 6) ^ (((row >> 3) & 1) * 5); }

constexpr int W3_MAXTP = 256;        // output pixels per tile
constexpr int W3_MAXPP = 352;        // patch pixels per tile

constexpr int W3_PROWS = 384;                           // patch rows an LDS buffer holds (48 DMA instructions: 6 for each of 8 waves)
constexpr int W3_DY_Q = W3_MAXTP / 8 / 4;               // DMA instructions per tile of each dy-loading wave (waves 0-3)
constexpr int W3_P_Q = W3_PROWS / 8 / 8;                // ... of each patch-loading wave (waves 4-11)
constexpr int W3_BUF = (W3_MAXTP + W3_MAXPP) * 128;     // bytes of one (dy tile | patch) buffer: 76 KB.  Two of them leave 8 KB of the CU's 160 KB:
                                                        // with all of it taken (buffers of W3_PROWS rows) even a 5 KB main-stream workgroup — a BatchNorm
                                                        // finalize — waited a whole weight-gradient kernel (~42 us) for LDS.  The four DMA instructions
                                                        // that would fill patch rows 352-383 (waves 8-11, i = 5) are skipped instead
static_assert(W3_MAXPP <= W3_PROWS, "patch buffer");

// One 12-wave workgroup per CU with two LDS buffers: the next tile's dy / patch stream in by LDS-DMA while the current tile's nine
// taps run on the MFMAs.  Wave = (filter row r, 2 x 2 quadrant of the KB x CB block): three waves per SIMD, each with a third of the
// taps — what one wave per SIMD could not hide (its own LDS-read latency before every group of MFMAs, its own DMA issue and address
// arithmetic: 18-20k cycles per tile against 4.6k of MFMA, no faster than the generic kernel) the other two now cover.
// Every loading wave issues a fixed number of DMA instructions per tile (rows beyond the tile fetch the zero page), so a constant
// `s_waitcnt vmcnt(n)` is exactly "the current tile has landed, the next one may still be in flight".
// Everything that does not depend on the tile is computed once: per DMA instruction the element offset of its pixel from the tile
// origin and the (y, x) to bounds-check; per (step, tap) the swizzled LDS offsets of the lane's two patch rows (16-bit pairs).
template <int KB, int CB>
__global__ __launch_bounds__(768) void wgrad3x3_kernel(const W3Args p) {
    constexpr int MTW = KB / 32, NTW = CB / 32;          // 16x16 MFMA tiles per wave (wave = KB/2 x CB/2 of the block, three taps)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tr = wave >> 2, wm = (wave >> 1) & 1, wn = wave & 1;
    const int g = lane >> 4, i16 = lane & 15, q4 = i16 >> 2, p4 = i16 & 3;
    const int blocks = p.nkb * p.ncb;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);      // the (K, C) blocks of one spatial split share its dy tile and x patch: one XCD's L2
    const int split = bid / blocks, blk = bid - split * blocks;
    const int kb = blk / p.ncb, cb = blk - kb * p.ncb;
    const int TP = p.TOH * p.TOW, PW = p.S * p.TOW + 3 - p.S, PP = (p.S * p.TOH + 3 - p.S) * PW;      // patch = S*T + 2 (stride 1), 2*T + 1 (stride 2)
    const int steps = (TP + 31) / 32;
    const float inv_tow = 1.0f / (float)p.TOW, inv_pw = 1.0f / (float)PW;
    const bf16_t* __restrict__ x = (const bf16_t*)p.x;
    const bf16_t* __restrict__ dy = (const bf16_t*)p.dy;
    const unsigned char* zero = (const unsigned char*)g_hdy_zero16_w3;
    const int per_img = p.tiles_h * p.tiles_w;
    const int tiles = p.N * per_img;
    const bool dy_loader = wave < 4;

    constexpr int NQ = W3_DY_Q > W3_P_Q ? W3_DY_Q : W3_P_Q;
    int l_off[NQ], l_yx[NQ];
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        l_off[i] = 0;
        l_yx[i] = 0x7FFF7FFF;                                          // never inside the image
        if (dy_loader && i < W3_DY_Q) {
            const int row = (wave + 4 * i) * 8 + (lane >> 3);
            const int ty = (int)(((float)row + 0.5f) * inv_tow), tx = row - ty * p.TOW;
            const int lc = (lane & 7) ^ fsw3(row);
            l_off[i] = (ty * p.W + tx) * p.lddy + kb * KB + lc * 8;
            if (row < TP && lc * 8 < KB) l_yx[i] = (ty << 16) | tx;
        } else if (!dy_loader && i < W3_P_Q) {
            const int row = (wave - 4 + 8 * i) * 8 + (lane >> 3);
            const int py = (int)(((float)row + 0.5f) * inv_pw), px = row - py * PW;
            const int lc = (lane & 7) ^ fsw3(row);
            l_off[i] = ((py - 1) * p.Win + (px - 1)) * p.ldx + cb * CB + lc * 8;
            if (row < PP && lc * 8 < CB) l_yx[i] = (py << 16) | px;
        }
    }
    auto issue = [&](int t, int buf) {
        unsigned char* sD = smem + buf * W3_BUF;
        unsigned char* sP = sD + W3_MAXTP * 128;
        const int n = t / per_img, rem = t - n * per_img;
        const int th = rem / p.tiles_w, tw = rem - th * p.tiles_w;
        const int oh0 = th * p.TOH, ow0 = tw * p.TOW;
        if (dy_loader) {
            const bf16_t* org = dy + ((size_t)(n * p.H + oh0) * p.W + ow0) * p.lddy;
            const int hy = p.H - oh0, hx = p.W - ow0;                                    // image rows / columns left from the tile origin
#pragma unroll
            for (int i = 0; i < W3_DY_Q; ++i) {
                const bool ok = (l_yx[i] >> 16) < hy && (l_yx[i] & 0xFFFF) < hx;
                const void* src = ok ? (const void*)(org + l_off[i]) : (const void*)zero;
                __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)src,
                                                 (void __attribute__((address_space(3)))*)(sD + (wave + 4 * i) * 1024), 16, 0, 0);
            }
        } else {
            const int ih0 = p.S * oh0, iw0 = p.S * ow0;
            const bf16_t* org = x + ((size_t)(n * p.Hin + ih0) * p.Win + iw0) * p.ldx;   // patch pixel (1, 1); offsets may be negative
#pragma unroll
            for (int i = 0; i < W3_P_Q; ++i) {
                if ((wave - 4 + 8 * i) * 8 >= W3_MAXPP) continue;                        // wave-uniform: rows the buffer does not hold
                const int py = l_yx[i] >> 16, px = l_yx[i] & 0xFFFF;                     // image pixel (ih0 + py - 1, iw0 + px - 1)
                const bool ok = (unsigned)(ih0 + py - 1) < (unsigned)p.Hin && (unsigned)(iw0 + px - 1) < (unsigned)p.Win && py < 0x7FFF;
                const void* src = ok ? (const void*)(org + l_off[i]) : (const void*)zero;
                __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)src,
                                                 (void __attribute__((address_space(3)))*)(sP + (wave - 4 + 8 * i) * 1024), 16, 0, 0);
            }
        }
    };

    constexpr int NSTEP = W3_MAXTP / 32;
    unsigned ptab[NSTEP][3];               // swizzled byte offsets (inside a patch image, < 2^16) of this lane's two rows for (step, tap of
                                           // this wave's filter row), b = 0: low half = pixels 0-3 of its group, high half = pixels 4-7
    int dtab[2];                           // ... of its dy rows at step 0, a = 0 (rows advance by 32 = 4096 bytes per step, same swizzle)
    {
        const int lcb = (wn * (CB / 2)) >> 3, lca = (wm * (KB / 2)) >> 3;
#pragma unroll
        for (int s = 0; s < NSTEP; ++s)
#pragma unroll
            for (int sx = 0; sx < 3; ++sx) ptab[s][sx] = 0;
#pragma unroll
        for (int s = 0; s < NSTEP; ++s)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int ra = s * 32 + 8 * g + q4 + 4 * h;
                const int ty = (int)(((float)ra + 0.5f) * inv_tow), tx = ra - ty * p.TOW;
                const int p0 = min(p.S * (ty * PW + tx), PP - 1 - 2 * PW - 2);      // rows past the tile hold zero dy: any patch row does
#pragma unroll
                for (int sx = 0; sx < 3; ++sx) {
                    const int pr = p0 + tr * PW + sx;
                    ptab[s][sx] |= (unsigned)(pr * 128 + (((lcb + (p4 >> 1)) ^ fsw3(pr)) << 4) + (p4 & 1) * 8) << (16 * h);
                }
                if (s == 0) dtab[h] = ra * 128 + (((lca + (p4 >> 1)) ^ fsw3(ra)) << 4) + (p4 & 1) * 8;
            }
    }

    f32x4 acc[3][MTW][NTW];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int a = 0; a < MTW; ++a)
#pragma unroll
            for (int b = 0; b < NTW; ++b) acc[t][a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    // Main loop, round 5.  One workgroup owns a CU (152 KB of LDS), so nothing else covers an exposed wait — and the first form had one per tile: its counted
    // `s_waitcnt vmcnt(n)` ("tile t has landed, tile t + 1 may still be in flight") was followed by `__syncthreads()`, into which hipcc puts
    // `s_waitcnt vmcnt(0)`: the NEXT tile's 76 KB were waited for before the current tile's MFMAs started (ISA of the round-4 build).  Now: wait for this
    // wave's requests of tile t (nothing newer is in flight at that point), ONE raw barrier (everyone's requests have landed AND everyone has left the
    // other buffer), then the requests of tile t + 1, then the MFMAs — with every LDS read as inline asm, because a compiler-visible LDS read behind a
    // pending LDS-DMA gets the same `vmcnt(0)`.  lgkmcnt is counted by hand: a tap's fragments are requested while the tap before it runs.
    typedef int i32x2 __attribute__((ext_vector_type(2)));
#define W3_TRR(dst, addr) asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(dst) : "v"(addr) : "memory")
#define W3_LGKM(n) do { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(n) : "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
    const unsigned lds0 = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) unsigned char*)smem;
    if (split < tiles) issue(split, 0);
    int cur = 0;
    for (int t = split; t < tiles; t += p.nsplit, cur ^= 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (t + p.nsplit < tiles) issue(t + p.nsplit, cur ^ 1);
        const unsigned sD = lds0 + cur * W3_BUF;
        const unsigned sP = sD + W3_MAXTP * 128;
        // ---- MFMAs: 32 reduction pixels per step, this wave's three taps from the same two LDS images.  16-channel tile a (b) of a wave
        // sits 2 chunks further: chunk ^ f with bit 1 flipped = byte offset ^ 32 per tile.
#pragma unroll
        for (int s = 0; s < NSTEP; ++s) {
            if (s < steps) {
                i32x2 al[MTW], ah[MTW], bl[2][NTW], bh[2][NTW];
#pragma unroll
                for (int a = 0; a < MTW; ++a) {
                    W3_TRR(al[a], sD + s * 4096 + (dtab[0] ^ (a << 5)));
                    W3_TRR(ah[a], sD + s * 4096 + (dtab[1] ^ (a << 5)));
                }
                {
                    unsigned pk = ptab[s][0];
                    asm volatile("" : "+v"(pk));           // keep the unpacking inside the tile loop (hoisted, the unpacked offsets double the registers)
#pragma unroll
                    for (int b = 0; b < NTW; ++b) {
                        W3_TRR(bl[0][b], sP + ((pk & 0xFFFFu) ^ (b << 5)));
                        W3_TRR(bh[0][b], sP + ((pk >> 16) ^ (b << 5)));
                    }
                }
#pragma unroll
                for (int sx = 0; sx < 3; ++sx) {
                    if (sx < 2) {
                        unsigned pk = ptab[s][sx + 1];
                        asm volatile("" : "+v"(pk));
#pragma unroll
                        for (int b = 0; b < NTW; ++b) {
                            W3_TRR(bl[(sx + 1) & 1][b], sP + ((pk & 0xFFFFu) ^ (b << 5)));
                            W3_TRR(bh[(sx + 1) & 1][b], sP + ((pk >> 16) ^ (b << 5)));
                        }
                        W3_LGKM(2 * NTW);                  // everything but the next tap's fragments is back
                    } else {
                        W3_LGKM(0);
                    }
                    V16 af[MTW], bf[NTW];
#pragma unroll
                    for (int a = 0; a < MTW; ++a) af[a].i = i32x4{al[a][0], al[a][1], ah[a][0], ah[a][1]};
#pragma unroll
                    for (int b = 0; b < NTW; ++b) bf[b].i = i32x4{bl[sx & 1][b][0], bl[sx & 1][b][1], bh[sx & 1][b][0], bh[sx & 1][b][1]};
#pragma unroll
                    for (int a = 0; a < MTW; ++a)
#pragma unroll
                        for (int b = 0; b < NTW; ++b)
                            acc[sx][a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[a].h, bf[b].h, acc[sx][a][b], 0, 0, 0);
                }
            }
        }
    }
#undef W3_TRR
#undef W3_LGKM

    const int Q = 9 * p.C;
    float* out = p.partial + (size_t)split * p.K * Q;
#pragma unroll
    for (int sx = 0; sx < 3; ++sx)
#pragma unroll
        for (int a = 0; a < MTW; ++a)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int k = kb * KB + wm * (KB / 2) + a * 16 + g * 4 + rr;
#pragma unroll
                for (int b = 0; b < NTW; ++b)
                    out[(size_t)k * Q + (tr * 3 + sx) * p.C + cb * CB + wn * (CB / 2) + b * 16 + i16] = acc[sx][a][b][rr];
            }
}

struct W3Plan { int KB, CB, TOH, TOW, tiles_h, tiles_w, nkb, ncb, nsplit; };

// tile shape (output pixels): whole image rows when they are short (W <= 40), 16x16 blocks when W divides by 16, else 8x32
bool w3_plan(int N, int Ho, int Wo, int C, int K, int stride, W3Plan* pl) {
    const bool off = hdy_opt(HDY_OPT_NO_WGRAD3X3) != 0;
    // Stride 1 only (57 vs 82, 57 vs 75, 67 vs 95, 84 vs 168 us against the generic kernel on yolov5s B=64).  A stride-2 plan was built and measured in
    // round 2 and is gone from the library: its patch is 4x the output tile, 45 KB of LDS-DMA for two MFMA steps, load bound — 101 vs 100, 86 vs 75,
    // 177 vs 180, 166 vs 172, 166 vs 152, 238 vs 233 us.
    if (off || C % 32 || K % 32 || stride != 1) return false;
    pl->KB = K % 64 == 0 ? 64 : 32;
    pl->CB = C % 64 == 0 ? 64 : 32;
    // several 32-wide blocks per split (yolov5m's 96 x 96 = 3 x 3 of them) do too little MFMA work per staged tile: 83 us against 70 us on the
    // generic kernel at 96x96 @80x80, B=32; a single 32 x 32 block (yolov5s' 32->32 @160x160) is 84 against 168 us
    if ((pl->KB == 32 || pl->CB == 32) && (K / pl->KB) * (C / pl->CB) > 1) return false;
    if (Wo <= 40) { pl->TOW = Wo; pl->TOH = W3_MAXTP / Wo; }
    else if (Wo % 16 == 0) { pl->TOW = 16; pl->TOH = 16; }
    else { pl->TOW = 32; pl->TOH = 8; }
    if (pl->TOH > Ho) pl->TOH = Ho;
    if (pl->TOH < 1) return false;
    // fewest rows per tile that need the same number of tiles: no more padding than necessary
    const int th = cdiv(Ho, pl->TOH);
    pl->TOH = cdiv(Ho, th);
    const int PH = stride * pl->TOH + 3 - stride, PW = stride * pl->TOW + 3 - stride;
    if (PH * PW > W3_MAXPP || pl->TOH * pl->TOW > W3_MAXTP) return false;
    pl->tiles_h = th;
    pl->tiles_w = cdiv(Wo, pl->TOW);
    pl->nkb = K / pl->KB;
    pl->ncb = C / pl->CB;
    const long long tiles = (long long)N * pl->tiles_h * pl->tiles_w;
    const int blocks = pl->nkb * pl->ncb;
    long long ns = 256 / blocks;                                       // one resident workgroup per CU (double-buffered LDS)
    const long long cap = (48LL << 20) / ((long long)K * 9 * C * 4);   // slabs within 48 MB
    if (ns > cap) ns = cap;
    if (ns > tiles) ns = tiles;
    if (ns < 1) ns = 1;
    pl->nsplit = (int)ns;
    return true;
}

}  // namespace

// 0 = shape not handled by the patch-resident kernel
size_t hdy_wgrad3x3_workspace_bytes(int N, int Ho, int Wo, int C, int K, int stride, int dtype) {
    W3Plan pl;
    if (dtype != HDY_BF16 || !w3_plan(N, Ho, Wo, C, K, stride, &pl)) return 0;
    return (size_t)pl.nsplit * K * 9 * C * sizeof(float);
}

// returns 1 when it launched (rc set), 0 when the generic kernel must run; *splits = slabs written
int hdy_wgrad3x3_try(const void* x, int ldx, const void* dy, int lddy, int N, int Hin, int Win, int Ho, int Wo, int C, int K, int stride, float* partial,
                     int dtype, hipStream_t st, int* splits, int* rc) {
    W3Plan pl;
    if (dtype != HDY_BF16 || !w3_plan(N, Ho, Wo, C, K, stride, &pl)) return 0;
    if ((((uintptr_t)x | (uintptr_t)dy) & 15) || ldx % 8 || lddy % 8) return 0;
    W3Args a = {};
    a.x = x; a.ldx = ldx; a.dy = dy; a.lddy = lddy; a.partial = partial;
    a.N = N; a.H = Ho; a.W = Wo; a.C = C; a.K = K; a.Hin = Hin; a.Win = Win; a.S = stride;
    a.TOH = pl.TOH; a.TOW = pl.TOW; a.tiles_h = pl.tiles_h; a.tiles_w = pl.tiles_w;
    a.nkb = pl.nkb; a.ncb = pl.ncb; a.nsplit = pl.nsplit;
    const int grid = pl.nkb * pl.ncb * pl.nsplit;
    constexpr int smem = 2 * W3_BUF;
    static PerDeviceOnce attr_once;           // first launch of this instance on any thread
    attr_once.run([&] {
        (void)hipFuncSetAttribute((const void*)wgrad3x3_kernel<64, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        (void)hipFuncSetAttribute((const void*)wgrad3x3_kernel<64, 32>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        (void)hipFuncSetAttribute((const void*)wgrad3x3_kernel<32, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        (void)hipFuncSetAttribute((const void*)wgrad3x3_kernel<32, 32>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    });
    hdy_note_dispatch("wgrad3x3");
    if (pl.KB == 64 && pl.CB == 64) hipLaunchKernelGGL((wgrad3x3_kernel<64, 64>), dim3(grid), dim3(768), smem, st, a);
    else if (pl.KB == 64) hipLaunchKernelGGL((wgrad3x3_kernel<64, 32>), dim3(grid), dim3(768), smem, st, a);
    else if (pl.CB == 64) hipLaunchKernelGGL((wgrad3x3_kernel<32, 64>), dim3(grid), dim3(768), smem, st, a);
    else hipLaunchKernelGGL((wgrad3x3_kernel<32, 32>), dim3(grid), dim3(768), smem, st, a);
    *splits = pl.nsplit;
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        hdy_set_error("wgrad3x3: launch failed: %s", hipGetErrorString(e));
        *rc = (int)e;
    } else {
        *rc = HDY_OK;
    }
    return 1;
}

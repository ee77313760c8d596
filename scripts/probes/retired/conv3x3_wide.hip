// 3x3 / stride 1 / pad 1 convolution of the wide layers (C % 64 == 0, K >= 128; bf16) with the INPUT PATCH RESIDENT in LDS, on conv_deep.hip's
// pipeline: conv forward and stride-1 data gradient of the 128- to 512-wide bottleneck convolutions.
//
// conv_deep.hip stages, per (tap, 64-channel block) K-tile of a 256 x 128 output tile, 32 KB of input rows and 16 KB of filter for 1 024 MFMA
// cycles: 47 B/clk against the ~36 B/clk the L2 -> LDS path delivers (profiles/r03_deep_stamps.txt) — the 3x3 layers are bound by staged
// bytes, and eight of the nine taps re-stage input pixels that the tile already fetched.  Here the tile's input patch (<= 384 pixels x 64
// channels = 48 KB per channel block, double buffered) is staged ONCE per channel block and all nine taps read their A fragments from it
// at a per-tap pixel shift; only the 16 KB filter tile of each K-tile streams (3-deep ring): (48 + 9 x 16) / 9 = 21 KB per K-tile.
//
//   tile       256 output pixels of ONE image: W <= 63: 256 consecutive pixels in row-major order (the patch is the band of pixels
//              [q0 - W - 1, q0 + 256 + W + 1), pitch W; a tap that leaves the image on the left / right would read the neighbouring row's
//              pixel: such lanes read a zero row instead); wider images: TH x TW = 8 x 32 or 16 x 16 blocks with a (TH + 2) x (TW + 2) patch.
//              Everything that depends on the mode is per-tile setup: 6 DMA offsets per thread, and per lane and 16-row sub-tile the patch
//              index of its pixel, a 9-bit "tap inside the image" mask and the output pixel.
//   pipeline   conv_deep.hip's BN = 128 schedule (8 waves = 4 (M) x 2 (N), two 128-row halves per K-tile = two phases of 16 MFMAs, raw
//              s_barrier, counted vmcnt, the two waves of a SIMD one barrier apart, fragments read one segment ahead into the registers the
//              MFMAs just consumed).  Per K-tile and thread: 2 DMA instructions of the filter tile two K-tiles ahead (phase 1) and, on taps
//              0..5, 1 instruction of the NEXT channel block's (or next tile's) patch (phase 2).  vmcnt is in order: the wait for filter
//              tile kt + 1 in phase 1 of kt also retires every patch instruction issued before it, so the patch needs no wait of its own.
//   epilogue   conv_deep.hip's (per wave, lanes 16 apart trade halves through ds_swizzle, one 16-byte store per lane, BatchNorm sums in
//              registers for the whole launch).
//
// MEASURED AND NOT THE DEFAULT (HDY_WIDE3=1 selects it; parity-tested on every 3x3 row of tests/test_gpu_kernels.py::test_deep_pipelined_conv):
// 2.3x fewer staged bytes bought nothing.  128->128 @40x40 (B = 64): 53-56 us against 47-50 us for conv_deep.hip, yolov5l 3x3 layers 20 % slower,
// yolov5l inference (B = 128, 1024 x 1024) network 52.6 against 48.1 ms.  PMC (scripts/probe_layers_pmc.sh): L2 hits halved (1.35 M vs 2.71 M) as
// designed, but MFMA busy per elapsed cycle is the SAME in both kernels (27.6 vs 27.7 counter units): the deep pipeline is paced by what a phase costs
// besides its 16 MFMAs (barriers, DMA issue, LDS latency), not by the bytes it stages — conv_deep.hip's "47 B/clk against 36" was not the limit — and
// this kernel adds per-lane address arithmetic (VALU instructions 8.6 M vs 4.5 M) and per-image tile quantisation (448 tiles for 400 tiles' worth of
// pixels at 40 x 40).  Timing ablations (-DHDY_W3_DBG=1): all DMA off 76 -> 73 us, MFMAs off 60, fragment reads off 54, everything off 41.
// With the address arithmetic dealt out between the MFMAs (sched_group_barrier: yolov5l 3x3 layers 20 % -> 10 % slower than conv_deep.hip) it still loses.
// What carries over: the rotation swizzle below (fragment rows read at ANY alignment without bank conflicts: 21 % -> 8 % conflict cycles).
//
// Reference semantics replaced: nn.Conv2d(k = 3, s = 1, p = 1) inside metayolo/models/layers.py:92-93 (Bottleneck.cv2), its autograd
// backward-data (train.py:472).
#include "common.h"
#include "hdyolo_internal.h"

// -DHDY_W3_DBG=1: timing ablations through HDY_DEEP_DEBUG (results wrong): 1 no patch DMA, 2 no filter DMA, 4 no MFMAs, 8 no epilogue stores, 16 no fragment reads
#ifndef HDY_W3_DBG
#define HDY_W3_DBG 0
#endif

namespace {

constexpr int NTHR = 512, BN = 128;
constexpr int PATCH_PIX = 384, PATCH_B = PATCH_PIX * 128;        // one channel block of the patch
constexpr int NPI = PATCH_B / 8192;                                // 6 DMA instructions per thread and patch
constexpr int NB = 3, BUNIT = 16384;
constexpr int L_PATCH = 0, L_B = 2 * PATCH_B, L_ZERO = L_B + NB * BUNIT, L_COEF = L_ZERO + 256, L_END = L_COEF + 2 * BN * 4;
static_assert(L_END <= 160 * 1024, "LDS budget");

struct WideGeo {
    int mode;            // 0: flattened band, 1: TH x TW blocks
    int H, W, HW;
    int pitch;           // patch pitch in pixels: W (mode 0) or TW + 2
    int tpi;             // tiles per image
    int tiles_x, th, tw_log;
    int ncb;             // 64-channel blocks
    int mtiles;          // N * tpi
};

__device__ __forceinline__ unsigned fdiv(unsigned n, unsigned mg, int sh) { return __umulhi(n << 1, mg) >> sh; }
__device__ __forceinline__ void lds_dma16(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, unsigned lds_byte) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (void __attribute__((address_space(3)))*)(uintptr_t)lds_byte, 16, (int)voff, (int)soff, 0, 0);
}
#define W3_READ(dst, addr) if (!(HDY_W3_DBG && (dbg & 16))) asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(addr) : "memory")
#define W3_READ_O(dst, addr, off) if (!(HDY_W3_DBG && (dbg & 16))) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off) : "memory")

// EPI: 0 raw output (train forward, data gradient), 1 scale / shift, 2 scale / shift + SiLU, 3 scale / shift + ReLU
template <bool STATS, int EPI>
__global__ __launch_bounds__(NTHR, 2) void conv3x3_wide_kernel(const ConvArgs p, const WideGeo g) {
    constexpr int MT = 2, NTQ = 4, NACC = 2 * MT * NTQ;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned lds0 = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) unsigned char*)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;                                    // stagger group
    const int wr = wave & 3, wc = wave >> 2;
    const int fr = lane & 15, fq = lane >> 4;
    const int dbg = HDY_W3_DBG ? p.dbg : 0;

    const int ntiles = p.ntiles;
    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int nt = wg % ntiles;
    const int mstep = (int)gridDim.x / ntiles;
    const int mt0 = wg / ntiles;
    if (mt0 >= g.mtiles) {
        if (STATS)
            for (int j = tid; j < 2 * BN; j += NTHR) {
                const int which = j / BN, c = j - which * BN;
                if (nt * BN + c < p.K) p.stats[((size_t)mt0 * 2 + which) * p.K + nt * BN + c] = 0.f;
            }
        return;
    }
    const int my_tiles = (g.mtiles - mt0 + mstep - 1) / mstep;
    const int n0 = nt * BN;
    const int pixB = p.ldx * 2;
    constexpr unsigned OOB = 0x80000000u;

    // zero row (out-of-image taps) and epilogue coefficients
    if (tid < 16) ((unsigned*)(smem + L_ZERO))[tid * 4 + 0] = 0u, ((unsigned*)(smem + L_ZERO))[tid * 4 + 1] = 0u,
                  ((unsigned*)(smem + L_ZERO))[tid * 4 + 2] = 0u, ((unsigned*)(smem + L_ZERO))[tid * 4 + 3] = 0u;
    if (EPI >= 1) {
        float* coef = (float*)(smem + L_COEF);
        for (int j = tid; j < BN; j += NTHR) {
            coef[j] = (p.scale && n0 + j < p.K) ? p.scale[n0 + j] : 1.0f;
            coef[BN + j] = (p.shift && n0 + j < p.K) ? p.shift[n0 + j] : 0.0f;
        }
    }

    // ------------------------------------------------------------------ loaders
    const int lc = (tid & 7) ^ ((tid >> 4) & 7);                  // filter tile: logical 16-byte chunk fetched into physical slot (tid & 7)
    const int r0 = tid >> 3;
    // Patch rows are read at ANY pixel alignment (tap shifts), where the XOR key (row >> 1) & 7 of the aligned tiles is 2-way conflicted for half
    // of the alignments (measured: 21 % conflict cycles; ds_read_b128's 16-lane groups are {0-3, 12-15, 20-27}, ...: the middle eight rows of a
    // fragment read chunk c + 1, the outer eight chunk c, and no XOR key keeps every 4-window of rows on aligned slot pairs).  A ROTATION does:
    // pixel p keeps chunk c in slot (c + (p & 6)) & 7 — outer rows land on even, middle rows on odd offsets of c whatever the start pixel.
    const int plc = ((tid & 7) - (r0 & 6)) & 7;                   // patch: logical chunk fetched into physical slot (tid & 7) of pixel 64 i + r0
    // filter: rows n0 + r0 and + 64 of the packed [K][Kdp] block, k offset = (tap * C + cb * 64)
    const unsigned wbytes = (unsigned)((size_t)((p.K + p.bn - 1) / p.bn * p.bn) * p.Kdp * 2);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, wbytes, 0x00020000);
    unsigned woff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) woff[i] = (unsigned)(((n0 + r0 + 64 * i) * p.Kdp) * 2 + lc * 16);
    const int nkt_tile = 9 * g.ncb;
    const long long total_kt = (long long)my_tiles * nkt_tile;
    long long b_kt = 0;                                           // filter loader position (K-tiles issued)
    int b_tap = 0, b_cb = 0, b_buf = 0;
    auto issue_b = [&]() {
        if (b_kt >= total_kt) return;
        if (dbg & 2) { ++b_kt; b_buf = b_buf + 1 == NB ? 0 : b_buf + 1; if (++b_tap == 9) { b_tap = 0; if (++b_cb == g.ncb) b_cb = 0; } return; }
        const unsigned dst = lds0 + (unsigned)(L_B + b_buf * BUNIT + wave * 1024);
        const unsigned wk = (unsigned)((b_tap * p.C + b_cb * 64) * 2);
#pragma unroll
        for (int i = 0; i < 2; ++i) lds_dma16(rw, woff[i], wk, dst + 8192 * i);
        ++b_kt;
        b_buf = b_buf + 1 == NB ? 0 : b_buf + 1;
        if (++b_tap == 9) { b_tap = 0; if (++b_cb == g.ncb) b_cb = 0; }
    };

    // patch: per tile 6 offsets per thread (instruction i covers patch pixels 64 i + (tid >> 3)), descriptor on the tile's image
    __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, OOB, 0x00020000);
    unsigned pv[NPI];
    int p_tile = 0, p_cb = 0, p_buf = 0;                          // (tile, channel block) whose patch is being / will be issued next
    bool p_live = true;
    auto tile_origin = [&](int j, int& n, int& a0, int& a1) {     // mode 0: a0 = first pixel q0; mode 1: a0 = y0, a1 = x0
        const int mt = mt0 + j * mstep;
        n = mt / g.tpi;
        const int ti = mt - n * g.tpi;
        if (g.mode == 0) { a0 = ti * 256; a1 = 0; }
        else { const int ty = ti / g.tiles_x; a0 = ty * g.th; a1 = (ti - ty * g.tiles_x) << g.tw_log; }
    };
    auto patch_set_tile = [&](int j) {
        int n, a0, a1;
        tile_origin(j, n, a0, a1);
        rx = __builtin_amdgcn_make_buffer_rsrc((void*)((const unsigned char*)p.x + (long long)n * g.HW * pixB), 0, OOB, 0x00020000);
#pragma unroll
        for (int i = 0; i < NPI; ++i) {
            const int pix = 64 * i + r0;
            int gq;
            bool ok;
            if (g.mode == 0) {
                gq = a0 - g.W - 1 + pix;
                ok = gq >= 0 && gq < g.HW && pix < 256 + 2 * g.W + 2;
            } else {
                const int pj = pix / g.pitch, pi = pix - pj * g.pitch;
                const int gy = a0 - 1 + pj, gx = a1 - 1 + pi;
                ok = pj < g.th + 2 && gy >= 0 && gy < g.H && gx >= 0 && gx < g.W;
                gq = gy * g.W + gx;
            }
            pv[i] = ok ? (unsigned)(gq * pixB + plc * 16) : OOB;
        }
        asm volatile("" : "+v"(pv[0]), "+v"(pv[1]), "+v"(pv[2]), "+v"(pv[3]), "+v"(pv[4]), "+v"(pv[5]));
    };
    auto issue_patch = [&](int i) {                               // instruction i of patch (p_tile, p_cb) into buffer p_buf
        const unsigned dst = lds0 + (unsigned)(L_PATCH + p_buf * PATCH_B + i * 8192 + wave * 1024);
        if (!(dbg & 1)) lds_dma16(rx, pv[i], (unsigned)(p_cb * 128), dst);
    };
    auto patch_advance = [&]() {                                  // after the last instruction of a patch
        p_buf ^= 1;
        if (++p_cb == g.ncb) {
            p_cb = 0;
            if (++p_tile < my_tiles) patch_set_tile(p_tile);
            else p_live = false;
        }
    };

    // ------------------------------------------------------------------ consumer state of the current tile
    int P[2][MT], msk[2][MT], opx[2][MT];                         // fragment geometry (P, msk) and output pixels of the lane's four 16-row sub-tiles
    int toff[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) toff[t] = (t / 3 - 1) * g.pitch + (t % 3 - 1);
    // pixel (y, x) of image n behind row r of tile j; false: the row is past the image (last tile / block edge)
    auto row_pixel = [&](int n_a0_a1_mode_unused, int a0, int a1, int r, int& y, int& x) -> bool {
        (void)n_a0_a1_mode_unused;
        if (g.mode == 0) {
            const int q = a0 + r, qc = min(q, g.HW - 1);
            y = (int)fdiv((unsigned)qc, p.mg_wo, p.sh_wo);
            x = qc - y * g.W;
            return q < g.HW;
        }
        y = a0 + (r >> g.tw_log);
        x = a1 + (r & ((1 << g.tw_log) - 1));
        return y < g.H && x < g.W;
    };
    auto cons_set_tile = [&](int j) {                             // where the lane's fragments sit in tile j's patch, which taps stay inside the image
        int n, a0, a1;
        tile_origin(j, n, a0, a1);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int r = a * 128 + wr * 32 + m * 16 + fr;
                int y, x;
                const bool ok = row_pixel(0, a0, a1, r, y, x);
                P[a][m] = g.mode == 0 ? r + g.W + 1 : ((r >> g.tw_log) + 1) * g.pitch + (r & ((1 << g.tw_log) - 1)) + 1;
                int mk = 0;
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
                    mk |= (ok && yy >= 0 && yy < g.H && xx >= 0 && xx < g.W) ? (1 << t) : 0;
                }
                msk[a][m] = mk;
            }
    };
    auto out_set_tile = [&](int j) {                              // output pixel of each sub-tile row (-1: none)
        int n, a0, a1;
        tile_origin(j, n, a0, a1);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                int y, x;
                const bool ok = row_pixel(0, a0, a1, a * 128 + wr * 32 + m * 16 + fr, y, x);
                opx[a][m] = ok ? (n * g.HW + y * g.W + x) : -1;
            }
    };
    const unsigned zero_a = lds0 + (unsigned)L_ZERO;
    // byte addresses of the lane's A fragment (both k-halves) for sub-tile (a, m), tap t, patch buffer base pb; a tap outside the image reads zeros
    auto a_addr = [&](int a, int m, int t, unsigned pb, unsigned& ad0, unsigned& ad1) {
        const int idx = P[a][m] + toff[t];
        const unsigned rowb = pb + (unsigned)idx * 128u;
        const unsigned c0 = (unsigned)((fq + (idx & 6)) & 7);
        const bool in = (msk[a][m] >> t) & 1;
        ad0 = in ? rowb + (c0 << 4) : zero_a;
        ad1 = in ? rowb + ((c0 ^ 4u) << 4) : zero_a;
    };

    // ------------------------------------------------------------------ accumulators, fragments
    f32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto AI = [](int a, int m, int n) constexpr { return (a * MT + m) * NTQ + n; };
    V16 afl[2][2], afh[2][2], bfa[4][2];
    const int brow = wc * 64 + fr;
    unsigned fb[2];
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) fb[kh] = lds0 + (unsigned)(L_B + brow * 128 + (((kh * 4 + fq) ^ ((brow >> 1) & 7)) << 4));
    float s1[STATS ? NTQ * 4 : 1], s2[STATS ? NTQ * 4 : 1];
#pragma unroll
    for (int i = 0; i < (STATS ? NTQ * 4 : 1); ++i) s1[i] = s2[i] = 0.f;

    // ------------------------------------------------------------------ prologue
    patch_set_tile(0);
#pragma unroll
    for (int i = 0; i < NPI; ++i) issue_patch(i);
    patch_advance();
    issue_b();
    issue_b();
    issue_b();
    cons_set_tile(0);
    out_set_tile(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    __builtin_amdgcn_s_barrier();
    unsigned cb_off = 0;                                          // consumer's filter ring offset
    unsigned pbase = lds0 + (unsigned)L_PATCH;                    // consumer's patch buffer
#pragma unroll
    for (int kh = 0; kh < 2; ++kh)
#pragma unroll
        for (int n = 0; n < 4; ++n) W3_READ_O(bfa[n][kh].i, fb[kh], n * 2048);
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        unsigned ad0, ad1;
        a_addr(0, m, 0, pbase, ad0, ad1);
        W3_READ(afl[m][0].i, ad0);
        W3_READ(afl[m][1].i, ad1);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if (grp == 1) __builtin_amdgcn_s_barrier();                   // stagger

    bf16_t* __restrict__ y = (bf16_t*)p.y;
    long long kt_done = 0;                                        // K-tiles consumed (for the end-of-stream waits)
    bool pprev = false;                                           // a patch instruction was issued in the previous K-tile's phase 1

// The address arithmetic of the reads that FOLLOW a compute segment (per-lane patch addresses: ~10 VALU instructions per 16-row sub-tile) is written
// in front of the segment's MFMAs and the scheduler is asked to deal it out between them (one MFMA, two VALU instructions, ...): an MFMA holds the
// matrix pipe for 16 cycles and the issue port for 4, the wave's own VALU work fits in the shadow instead of extending the segment.
#define W3_MFMA(A_, AF)                                                                                                                 \
    if (!(dbg & 4)) {                                                                                                                                   \
        __builtin_amdgcn_s_setprio(1);                                                                                                  \
        _Pragma("unroll") for (int kh = 0; kh < 2; ++kh)                                                                                \
            _Pragma("unroll") for (int m = 0; m < 2; ++m)                                                                               \
                _Pragma("unroll") for (int n = 0; n < 4; ++n)                                                                           \
                    acc[AI(A_, m, n)] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfa[n][kh].h, AF[m][kh].h, acc[AI(A_, m, n)], 0, 0, 0); \
        _Pragma("unroll") for (int i = 0; i < 16; ++i) {                                                                                \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                                          \
            __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);                                                                          \
        }                                                                                                                               \
        __builtin_amdgcn_s_setprio(0);                                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                                              \
    }

    for (int j = 0; j < my_tiles; ++j) {
        for (int cb = 0; cb < g.ncb; ++cb) {
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                // ---- phase 1: rows 0..127 x this K-tile's filter; afterwards the rows 128..255 fragments of the same tap.
                // DMA: one instruction of the next patch (taps 0..5).  Then filter tile kt + 1 must have landed (it is read at the end of phase 2):
                // younger than it are the previous K-tile's patch instruction, filter tile kt + 2 (2 instructions) and this phase's patch instruction.
                bool pnow = false;
                if (tap < NPI && p_live) {
                    issue_patch(tap);
                    pnow = true;
                    if (tap == NPI - 1) patch_advance();
                }
                if (b_kt > kt_done + 2) {                         // filter tile kt + 2 has been issued: it may stay in flight
                    const int np = (pprev ? 1 : 0) + (pnow ? 1 : 0);
                    if (np == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                    else if (np == 1) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                } else {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                pprev = pnow;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                unsigned ra0[MT], ra1[MT];
#pragma unroll
                for (int m = 0; m < MT; ++m) a_addr(1, m, tap, pbase, ra0[m], ra1[m]);
                W3_MFMA(0, afl)
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    W3_READ(afh[m][0].i, ra0[m]);
                    W3_READ(afh[m][1].i, ra1[m]);
                }
                __builtin_amdgcn_s_barrier();
                // ---- phase 2: rows 128..255; afterwards the next K-tile's rows 0..127 fragments and filter fragments.
                // DMA: filter tile kt + 3 into the slot of tile kt, whose fragments every wave (the staggered group too) has in registers by now
                issue_b();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                ++kt_done;
                const unsigned cnext = cb_off + (unsigned)BUNIT == (unsigned)(NB * BUNIT) ? 0u : cb_off + (unsigned)BUNIT;
                const bool more = kt_done < total_kt;
                // next K-tile: the same patch at the next tap, or the next channel block's / tile's patch (the other buffer) at tap 0
                const bool newp = tap == 8;
                const int ntap = newp ? 0 : tap + 1;
                unsigned pbn = pbase;
                if (more && newp) {
                    pbn = (pbase - lds0) == (unsigned)L_PATCH ? lds0 + (unsigned)(L_PATCH + PATCH_B) : lds0 + (unsigned)L_PATCH;
                    if (cb + 1 == g.ncb) cons_set_tile(j + 1);            // fragment geometry of the next tile (its output pixels after this tile's epilogue)
                }
                unsigned rb0[MT], rb1[MT];
#pragma unroll
                for (int m = 0; m < MT; ++m) a_addr(0, m, ntap, pbn, rb0[m], rb1[m]);
                W3_MFMA(1, afh)
                if (more) {
#pragma unroll
                    for (int m = 0; m < MT; ++m) {
                        W3_READ(afl[m][0].i, rb0[m]);
                        W3_READ(afl[m][1].i, rb1[m]);
                    }
#pragma unroll
                    for (int kh = 0; kh < 2; ++kh) {
                        const unsigned bb = fb[kh] + cnext;
#pragma unroll
                        for (int n = 0; n < 4; ++n) W3_READ_O(bfa[n][kh].i, bb, n * 2048);
                    }
                    pbase = pbn;
                }
                cb_off = cnext;
                __builtin_amdgcn_s_barrier();
            }
        }
        // ---------------------------------------------------------------- epilogue of the finished tile (per wave, no barrier, no LDS memory)
        {
            const int odd = fq & 1;
            const int cl = odd ? 16 + (fq - 1) * 4 : fq * 4;
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int np = 0; np < 2; ++np) {
                        const int nb = 2 * np;
                        const int cbk = wc * 64 + np * 32;
                        unsigned pk[2][2];
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            f32x4 v = acc[AI(a, m, nb + q)];
                            if (STATS) {
#pragma unroll
                                for (int r = 0; r < 4; ++r) {
                                    s1[(nb + q) * 4 + r] += v[r];
                                    s2[(nb + q) * 4 + r] = __builtin_fmaf(v[r], v[r], s2[(nb + q) * 4 + r]);
                                }
                            }
                            if (EPI >= 1) {
                                f32x4 sc, sh;
                                const unsigned ca = lds0 + (unsigned)(L_COEF + (cbk + q * 16 + fq * 4) * 4);
                                asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:%3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(sc), "=&v"(sh) : "v"(ca), "n"(BN * 4));
#pragma unroll
                                for (int r = 0; r < 4; ++r) {
                                    v[r] = v[r] * sc[r] + sh[r];
                                    if (EPI == 2) v[r] = v[r] * __builtin_amdgcn_rcpf(1.0f + __expf(-v[r]));
                                    else if (EPI == 3) v[r] = fmaxf(v[r], 0.0f);
                                }
                            }
                            union { bf16x4 h; unsigned u[2]; } o;
                            o.h = bf16x4{(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
                            pk[q][0] = o.u[0];
                            pk[q][1] = o.u[1];
                        }
                        const unsigned sa = odd ? pk[0][0] : pk[1][0], sb = odd ? pk[0][1] : pk[1][1];
                        const unsigned ra = (unsigned)__builtin_amdgcn_ds_swizzle((int)sa, 0x401F);       // lane ^ 16
                        const unsigned rb = (unsigned)__builtin_amdgcn_ds_swizzle((int)sb, 0x401F);
                        V16 o;
                        o.i = odd ? i32x4{(int)ra, (int)rb, (int)pk[1][0], (int)pk[1][1]} : i32x4{(int)pk[0][0], (int)pk[0][1], (int)ra, (int)rb};
                        const int kc = n0 + cbk + cl;
                        const int op = opx[a][m];
                        if (op >= 0 && kc < p.K && !(dbg & 8)) {
                            const size_t opix = (size_t)op;
                            if (p.res || p.accumulate) {
                                float f[8];
#pragma unroll
                                for (int e = 0; e < 8; ++e) f[e] = (float)o.h[e];
                                if (p.res) {
                                    V16 q;
                                    q.i = *(const i32x4*)((const bf16_t*)p.res + opix * p.ldr + kc);
#pragma unroll
                                    for (int e = 0; e < 8; ++e) f[e] += (float)q.h[e];
                                }
                                if (p.accumulate) {
                                    V16 q;
                                    q.i = *(const i32x4*)(y + opix * p.ldy + kc);
#pragma unroll
                                    for (int e = 0; e < 8; ++e) f[e] += (float)q.h[e];
                                }
#pragma unroll
                                for (int e = 0; e < 8; ++e) o.h[e] = (bf16_t)f[e];
                            }
                            *(i32x4*)(y + opix * p.ldy + kc) = o.i;
                        }
                    }
        }
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (j + 1 < my_tiles) out_set_tile(j + 1);                // (its fragment geometry is already in place)
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();                   // re-align the two wave groups
#undef W3_MFMA

    if (STATS) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __syncthreads();
        float* red = (float*)smem;                                // [4][BN][2]; the patches are free
#pragma unroll
        for (int n = 0; n < NTQ; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float u = s1[n * 4 + r], q = s2[n * 4 + r];
#pragma unroll
                for (int ms = 1; ms < 16; ms <<= 1) {
                    u += __shfl_xor(u, ms);
                    q += __shfl_xor(q, ms);
                }
                if (fr == 0) {
                    const int col = wc * 64 + n * 16 + fq * 4 + r;
                    red[(wr * BN + col) * 2 + 0] = u;
                    red[(wr * BN + col) * 2 + 1] = q;
                }
            }
        __syncthreads();
        for (int j = tid; j < 2 * BN; j += NTHR) {
            const int which = j / BN, c = j - which * BN;
            if (n0 + c < p.K) {
                float v = 0.f;
#pragma unroll
                for (int q = 0; q < 4; ++q) v += red[(q * BN + c) * 2 + which];
                p.stats[((size_t)mt0 * 2 + which) * p.K + n0 + c] = v;
            }
        }
    }
}

// geometry of a launch; false: the shape is not this kernel's
bool wide_geo(int N, int H, int W, int C, WideGeo* g) {
    WideGeo o = {};
    o.H = H; o.W = W; o.HW = H * W; o.ncb = C / 64;
    if (W <= 63) {
        o.mode = 0; o.pitch = W; o.tpi = cdiv(o.HW, 256);
    } else {
        // 8 x 32 or 16 x 16 blocks: the one with fewer tiles (ties: 8 x 32, whose rows are longer)
        const int t832 = cdiv(H, 8) * cdiv(W, 32), t1616 = cdiv(H, 16) * cdiv(W, 16);
        o.mode = 1;
        if (t832 <= t1616) { o.th = 8; o.tw_log = 5; } else { o.th = 16; o.tw_log = 4; }
        o.pitch = (1 << o.tw_log) + 2;
        o.tiles_x = cdiv(W, 1 << o.tw_log);
        o.tpi = cdiv(H, o.th) * o.tiles_x;
    }
    o.mtiles = N * o.tpi;
    *g = o;
    return true;
}

inline int wide_grid(int mtiles, int ntiles) {
    long long gr = 256 / ntiles * ntiles;
    if (gr > (long long)mtiles * ntiles) gr = (long long)mtiles * ntiles;
    return (int)gr;
}

template <bool STATS, int EPI>
int wide_launch(const ConvArgs& a, const WideGeo& g, int grid, hipStream_t st) {
    static std::once_flag attr_once;
    std::call_once(attr_once, [&] {
        (void)hipFuncSetAttribute((const void*)conv3x3_wide_kernel<STATS, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)L_END);
    });
    hipLaunchKernelGGL((conv3x3_wide_kernel<STATS, EPI>), dim3(grid), dim3(NTHR), L_END, st, a, g);
    HDY_LAUNCH_CHECK("conv3x3_wide");
    return HDY_OK;
}

template <bool STATS>
int wide_launch_epi(const ConvArgs& a, const WideGeo& g, int grid, hipStream_t st) {
    const bool affine = a.scale != nullptr || a.shift != nullptr;
    if (a.act == 1) return wide_launch<STATS, 2>(a, g, grid, st);
    if (a.act == 2) return wide_launch<STATS, 3>(a, g, grid, st);
    if (affine) return wide_launch<STATS, 1>(a, g, grid, st);
    return wide_launch<STATS, 0>(a, g, grid, st);
}

// shapes this kernel takes (3x3 / stride 1 / pad 1 checked by the caller): 64-channel blocks, 128-wide column tiles, enough tiles for the chip
bool wide_shape_ok(int N, int H, int W, int C, int K, WideGeo* g) {
    if (!hdy_opt(HDY_OPT_WIDE3)) return false;                   // opt-in: measured no faster than conv_deep.hip (header)
    if (C % 64 != 0 || C < 128 || K < 128 || K % 8 != 0) return false;
    if (!wide_geo(N, H, W, C, g)) return false;
    return (long long)g->mtiles * cdiv(K, BN) >= hdy_opt(HDY_OPT_DEEP_MIN_TILES);
}

}  // namespace

// statistic slabs of a forward launch on this kernel (one per workgroup position); 0: not this kernel's shape
int hdy_conv3x3_wide_slabs(int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dtype) {
    WideGeo g;
    if (dtype != HDY_BF16 || R != 3 || S != 3 || stride != 1 || pad != 1 || !wide_shape_ok(N, H, W, C, K, &g)) return 0;
    const int ntiles = cdiv(K, BN);
    return wide_grid(g.mtiles, ntiles) / ntiles;
}

// called by hdy_conv_igemm_launch after its validation (forward and stride-1 data gradient of a 3x3 / pad 1 layer look the same here)
int hdy_conv3x3_wide_try(const ConvArgs& a_in, int dtype, int out_f32, hipStream_t st, int* rc) {
    if (dtype != HDY_BF16 || out_f32 || a_in.ncls > 1 || a_in.nstat > 0 || !a_in.dense_out || a_in.span_pixels || !a_in.vec_out || !a_in.utap) return 0;
    if (a_in.TH != 3 || a_in.TW != 3 || a_in.dh0 != -1 || a_in.dw0 != -1 || a_in.ih_mul != 1 || a_in.iw_mul != 1 || a_in.Hin != a_in.Ho || a_in.Win != a_in.Wo) return 0;
    WideGeo g;
    if (!wide_shape_ok(a_in.N, a_in.Hin, a_in.Win, a_in.C, a_in.K, &g)) return 0;
    if (a_in.ldx % 8 != 0 || (long long)a_in.Hin * a_in.Win * a_in.ldx * 2 >= (1LL << 31)) {        // rows not 16-byte aligned / an image beyond a 31-bit buffer offset
        if (!a_in.stats) return 0;
        hdy_set_error("conv: this shape's statistic slabs were sized for the patch-resident 3x3 kernel, which declined the launch (alignment)");
        *rc = HDY_EINVAL;
        return 1;
    }
    ConvArgs a = a_in;
    a.dbg = hdy_opt(HDY_OPT_DEEP_DEBUG);
    a.ntiles = cdiv(a.K, BN);
    const int grid = wide_grid(g.mtiles, a.ntiles);
    hdy_note_dispatch("wide3x3_256x128");
    if (a.stats) *rc = wide_launch_epi<true>(a, g, grid, st);
    else *rc = wide_launch_epi<false>(a, g, grid, st);
    return 1;
}

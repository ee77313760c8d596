// Data gradient of the 3x3 / stride 2 / pad 1 convolution, patch-resident with the filter in registers (bf16), for the two narrow stride-2 layers of
// yolov5s: 32 -> 64 (its second layer, 320x320 -> 160x160 at 640x640 tiles: dx has 32 channels, dy 64) and — round 5 — 64 -> 128 (160x160 -> 80x80:
// dx 64 channels, dy 128).  These are the layers the class-walking generic kernel served worst: 64-byte (128-byte) output pixels written as four
// interleaved parity classes, dy re-gathered per class and tap — 257 us for a 100 us HBM bound on the first, 122 us for 50 us on the second.
//
//   dx[n, 2i+a, 2j+b, c] = SUM over the taps (r, s) of parity class (a, b), k:  dy[n, i+di, j+dj, k] * w[k][c][r][s]
//   class (0,0): (1,1)@(0,0)        class (0,1): (1,2)@(0,0) (1,0)@(0,1)        class (1,0): (2,1)@(0,0) (0,1)@(1,0)
//   class (1,1): (2,2)@(0,0) (2,0)@(0,1) (0,2)@(1,0) (0,0)@(1,1)                 — nine (class, tap) products over four dy shifts (di, dj)
//
// Template <NP, NCG>: dy has 64 * NP channels, kept as NP separate 64-channel PLANES (each a patch of 128-byte pixel rows, so that the swizzle,
// the loader and the fragment reads are the same for every NP); dx has 16 * NCG channels.  <1, 2> is round 2's kernel, <2, 4> the new layer.
// A workgroup of 2 * NCG waves takes an 8 x 16 tile of dy positions: the 9 x 17-pixel dy patch (NP x 19.6 KB) arrives by LDS-DMA (double
// buffered, the filter-resident 3x3 kernel's source-side XOR swizzle, zero page past the image), and the workgroup produces the 16 x 32 block of dx
// pixels.  Wave = (16-channel group wc) x (upper / lower four tile rows wp); it keeps the nine filter slices of its channels as MFMA row operands
// (9 x 2 NP k-halves x 4 VGPRs = 72 / 144) and 4 classes x 4 rows of accumulators (64 VGPRs).  Per patch row and k-half two fragment reads (column
// shift 0 / 1) feed 6 + 3 MFMAs.  The filter is read from the class-walk packing of the generic kernel as it is (hdy_conv_pack_describe, kind dgrad:
// per class rows = c, columns = tap * K + k).
// Epilogue: a lane holds 4 consecutive channels of one dx pixel; the block is staged as rows of 128 bytes (NCG = 2: two neighbouring dx pixels per
// row, NCG = 4: one), 8-byte slots XORed with the tile column, and leaves with 16-byte stores, every dx row of the block as one contiguous run.
//
// Requirements (checked by the launcher, otherwise the class walk runs): bf16, (dy, dx) channels (64, 32) or (128, 64), even H and W, H/2 % 8 == 0,
// W/2 % 16 == 0, 16-byte aligned rows, no producer-side statistics.
#include <stdlib.h>

#include "common.h"
#include "hdyolo_internal.h"

__device__ uint4 g_hdy_zero16_d2[4];   // zero page for patch pixels past the image

namespace {

constexpr int TW = 16, PW = TW + 1;
constexpr int CB = 128, CPP = 8;                                             // bytes / 16-byte chunks per dy pixel of ONE 64-channel plane

// TH = tile rows (dy positions): 8 for <1, 2> (two row halves of four), 4 for <2, 4>: with 144 filter registers per wave a workgroup can have four waves if two
// of them are to share a CU (2 x 4 waves = two per SIMD = 256 registers each) — and two workgroups per CU is what overlaps one's MFMAs with the other's
// stores (the first <2, 4> form, one 8-wave workgroup of 140 KB per CU, ran its three phases back to back: 19 us of patch loads + 49 us of MFMAs + 30 us
// of stores = 92 us, ablations in profiles/r05_dgrad_s2_k128c64.txt)
template <int NP, int NCG, int TH> struct Geo {
    static constexpr int NWP = TH / 4;                                       // row groups of four tile rows
    static constexpr int NTHR = 64 * NCG * NWP;
    static constexpr int PPIX = (TH + 1) * PW;                               // patch pixels (153 / 85)
    static constexpr int PLANE_B = PPIX * CB;
    static constexpr int PATCH_B = NP * PLANE_B;
    static constexpr int PPR = 4 / NCG;                                      // dx pixels per 128-byte staging row (NCG = 2: 64-byte pixels, two per row)
    static constexpr int RPH = 32 / PPR;                                     // staging rows per dx row
    static constexpr int SROWS = 2 * TH * RPH;                               // staging rows: 2 TH dx rows x 32 dx pixels
    static constexpr int STAGE_B = SROWS * 128;
    static constexpr int SMEM_B = 2 * PATCH_B + STAGE_B;                     // <1, 2, 8>: 71 936; <2, 4, 4>: 76 288 — two workgroups per CU either way
    static constexpr int NPASS = (PPIX * CPP + NTHR - 1) / NTHR;             // loader passes per plane (last partial)
    static constexpr int RPP = NTHR / 8;                                     // staging rows per store pass
    static constexpr int SPASS = SROWS / RPP;                                // store passes: 8 in both forms
    static constexpr int HSTEP = RPP / RPH;                                  // dx rows per store pass
    static_assert(SPASS == 8 && RPP % RPH == 0 && PPIX <= 153, "geometry");
};

__device__ __forceinline__ void glds16(const void* g, unsigned char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)g, (void __attribute__((address_space(3)))*)lds_wave_base, 16, 0, 0);
}

// LDS fragment reads as inline asm and raw barriers: hipcc puts `s_waitcnt vmcnt(0)` in front of every compiler-visible LDS read that may alias a pending
// LDS-DMA, and `__syncthreads()` waits for vmcnt(0) too — the next patch's DMA (issued in front of the MFMAs) was waited for at the first fragment
// read, and the tile's output stores at the barrier behind them: patch loads, MFMAs and stores ran back to back (18 + 44 + 30 = 92 us on the 64 <- 128
// layer, profiles/r05_dgrad_s2_k128c64.txt).  The reads the compiler cannot see do not trigger the wait; the waits that ARE needed are counted by hand.
// HAND (= NP == 2) selects that form; the one-plane instance keeps compiler-visible reads and `__syncthreads()`: with its smaller footprint two / three
// workgroups share a CU and cover each other's waits, and the hand-counted form measured 3-8 % slower there (120-128 against 116-118 us).
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

template <int NP, int NCG, int TH, int ABL = 0>      // ABL: timing ablations (results wrong): 1 no patch loads after the first, 4 no MFMAs, 8 no stores
__global__ __launch_bounds__(64 * NCG * (TH / 4), 2) void dgrad3x3s2_kernel(const ConvArgs p) {
    using G = Geo<NP, NCG, TH>;
    constexpr int NTHR = G::NTHR, PATCH_B = G::PATCH_B, PLANE_B = G::PLANE_B, PPIX = G::PPIX, NPASS = G::NPASS, PPR = G::PPR, KD = 64 * NP;
    constexpr bool HAND = NP == 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sP = smem;                    // [2][NP][PPIX][128 B]
    unsigned char* sS = smem + 2 * PATCH_B;      // [SROWS][128 B] staging block

    const unsigned lds0 = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) unsigned char*)smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int wc = wave % NCG, wp = wave / NCG;
    const int tiles_w = p.Wo / TW, tiles_h = p.Ho / TH, per_img = tiles_w * tiles_h;
    const int tiles = p.N * per_img;
    const int wg = xcd_remap(blockIdx.x, gridDim.x);

    const bf16_t* __restrict__ x = (const bf16_t*)p.x;          // dy [N][Ho][Wo][ldx]
    const bf16_t* __restrict__ w = (const bf16_t*)p.w;
    const unsigned char* zero = (const unsigned char*)g_hdy_zero16_d2;

    // ---- the nine (class, tap) filter slices of this wave's 16 channels: row operand = w_cls[c = wc*16 + fr][tap*KD + (kk*4 + fq)*8 .. +7], kk = plane*2 + k-half
    V16 bw[9][2 * NP];
    {
        constexpr int cls_of[9] = {0, 1, 1, 2, 2, 3, 3, 3, 3}, tap_of[9] = {0, 0, 1, 0, 1, 0, 1, 2, 3};
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const bf16_t* base = w + p.c_w[cls_of[i]] + (size_t)(wc * 16 + fr) * (p.c_nkb[cls_of[i]] * 64) + tap_of[i] * KD;
#pragma unroll
            for (int kk = 0; kk < 2 * NP; ++kk) bw[i][kk].i = *(const i32x4*)(base + (kk * 4 + fq) * 8);
        }
    }

    auto issue_patch = [&](int t, int buf) {
        // everything below depends on the thread index only: left visible, hipcc hoists ~20 per-pass values out of the tile loop and — with 144 filter and
        // 64 accumulator registers live — spills them (scratch reloads in the loop, each behind an `s_waitcnt vmcnt(0)` that also waits for the previous
        // tile's output stores).  An opaque copy of the index keeps the arithmetic (a dozen VALU instructions per pass) inside the call
        int tv = tid;
        if constexpr (HAND) asm volatile("" : "+v"(tv));
        const int lc = tv & 7, wv = tv >> 6;
        const int n = t / per_img, rem = t - n * per_img;
        const int th = rem / tiles_w, tw = rem - th * tiles_w;
        const int i0 = th * TH, j0 = tw * TW;
        const bf16_t* org = x + (((long long)n * p.Hin + i0) * p.Win + j0) * p.ldx;
        unsigned char* dst = sP + buf * PATCH_B;
#pragma unroll
        for (int pl = 0; pl < NP; ++pl)
#pragma unroll
            for (int i = 0; i < NPASS; ++i) {
                if ((wave * 64 + NTHR * i) / CPP >= PPIX) break;               // wave-uniform: whole 1 KB pieces past the patch
                const int pix = (tv + NTHR * i) / CPP;
                if (pix >= PPIX) continue;
                const int py = (pix * 3856) >> 16, px = pix - py * PW;          // pix / 17 for pix < 153
                const int lcp = lc ^ (((px >> 1) & 3) << 1);                   // 128-byte pixel rows XOR-swizzled by the patch column (conv3x3.hip)
                const void* src = (i0 + py < p.Hin && j0 + px < p.Win) ? (const void*)(org + ((long long)py * p.Win + px) * p.ldx + pl * 64 + lcp * 8) : (const void*)zero;
                glds16(src, dst + pl * PLANE_B + (wv * 64 + NTHR * i) * 16);
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

    const int st_ch = tid & 7, st_rr = tid >> 3;                              // store phase: 16-byte chunk, first staging row (rows + NTHR / 8 per pass = dx rows + 2)
    // staging row R: PPR = 2: R = hr * 16 + col holds dx pixels 2 col, 2 col + 1 (32 channels each); PPR = 1: R = hr * 32 + px holds dx pixel px (64 channels).
    // Either way the writers' XOR key is the tile column fr = the dx pixel >> 1: 16 rows written by one instruction -> 16 different 8-byte slots
    const int st_key = PPR == 2 ? (st_rr & 15) : ((st_rr >> 1) & 15);
    const int st_lds = st_rr * 128 + ((st_ch ^ (st_key >> 1)) << 4);
    const int st_hr0 = st_rr / G::RPH;                                        // hr = st_hr0 + HSTEP j
    const long long st_pix = PPR == 2 ? (long long)(2 * (st_rr & 15) + (st_ch >> 2)) * p.ldy + (st_ch & 3) * 8 : (long long)(st_rr & 31) * p.ldy + st_ch * 8;

    auto compute = [&](int t, int cur) {
        f32x4 acc[4][4];                                                      // [class][tile row of this wave]
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int a = 0; a < 4; ++a) acc[c][a] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) {
            const unsigned pb = lds0 + cur * PATCH_B + pl * PLANE_B + wp * 4 * PW * CB;      // this wave reads patch rows wp*4 .. wp*4 + 4 of plane pl
            V16 f[2][2];                                                          // [buffer][column shift]
            S2_LDSR(f[0][0].i, pb + aoff[0][0]);
            S2_LDSR(f[0][1].i, pb + aoff[1][0]);
#pragma unroll
            for (int g = 0; g < 10; ++g) {                                        // group g = (patch row q = g / 2, k-half ks = g % 2)
                const int q = g >> 1, ks = pl * 2 + (g & 1);
                if (g + 1 < 10) {
                    S2_LDSR(f[(g + 1) & 1][0].i, pb + ((g + 1) >> 1) * PW * CB + aoff[0][(g + 1) & 1]);
                    S2_LDSR(f[(g + 1) & 1][1].i, pb + ((g + 1) >> 1) * PW * CB + aoff[1][(g + 1) & 1]);
                    S2_LGKM(2);                                                   // this group's two reads are back, the next group's are in flight
                } else {
                    S2_LGKM(0);
                }
                const bf16x8 f0 = f[g & 1][0].h, f1 = f[g & 1][1].h;
                if (ABL & 4) continue;
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
        }
        // ---- staging: class (ca, cb), tile row wp*4 + a -> dx row hr = 2*(wp*4 + a) + ca, dx pixel 2 fr + cb, channels wc*16 + fq*4 ..
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const int ca = c >> 1, cb = c & 1;
                const int hr = 2 * (wp * 4 + a) + ca;
                const int R = PPR == 2 ? hr * 16 + fr : hr * 32 + 2 * fr + cb;
                const int slot = PPR == 2 ? cb * 8 + wc * 4 + fq : wc * 4 + fq;
                bf16x4 o = {(bf16_t)acc[c][a][0], (bf16_t)acc[c][a][1], (bf16_t)acc[c][a][2], (bf16_t)acc[c][a][3]};
                *(bf16x4*)(sS + R * 128 + ((slot ^ (fr & 15)) << 3)) = o;        // key fr & 15: 16 rows -> 16 slots (see conv3x3.hip); odd rows swap a chunk's halves
            }
        S2_BARRIER(true);                                  // staging complete; every wave is done with patch `cur`
        {
            const int n = t / per_img, rem = t - n * per_img;
            const int th = rem / tiles_w, tw = rem - th * tiles_w;
            bf16_t* yb = (bf16_t*)p.y + (((long long)n * p.Hout + 2 * th * TH + st_hr0) * p.Wout + 2 * tw * TW) * p.ldy + st_pix;
            const long long st_step = (long long)G::HSTEP * p.Wout * p.ldy;    // one store pass further = HSTEP dx rows further
            // the block's eight chunks of this thread by asm reads (a compiler-visible LDS read behind a global store costs `s_waitcnt vmcnt(0)` — the
            // store's full latency, eight times per tile), then the stores back to back
            V16 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) S2_LDSR(v[j].i, lds0 + 2 * PATCH_B + st_lds + j * G::RPP * 128);
            S2_LGKM(0);
            if (p.accumulate) {                            // all eight rows requested before the first store: a load behind a store waits for the store too
                V16 u[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) u[j].i = *(const i32x4*)(yb + j * st_step);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if (st_key & 1) v[j].i = i32x4{v[j].i[2], v[j].i[3], v[j].i[0], v[j].i[1]};
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[j].h[e] = (bf16_t)((float)v[j].h[e] + (float)u[j].h[e]);
                }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (st_key & 1) v[j].i = i32x4{v[j].i[2], v[j].i[3], v[j].i[0], v[j].i[1]};
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (ABL & 8) break;
                *(i32x4*)(yb + j * st_step) = v[j].i;
            }
        }
        // the next patch's DMA precedes these 8 stores in the wave's vm queue: wait for it, not for the stores (accumulate: its loads have returned)
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        S2_BARRIER(false);                                 // next patch landed for everyone; staging block free again (its reads returned before the stores left)
    };

    if (wg >= tiles) return;
    if ((ABL & 16) && blockIdx.x >= gridDim.x / 2) __builtin_amdgcn_s_sleep(90);      // probe: the second workgroup of a CU half a tile behind the first
    issue_patch(wg, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int cur = 0;
    for (int t = wg; t < tiles; t += (int)gridDim.x, cur ^= 1) {
        if (t + (int)gridDim.x < tiles && !(ABL & 1)) issue_patch(t + (int)gridDim.x, cur ^ 1);
        compute(t, cur);
    }
}

}  // namespace

template <int NP, int NCG, int TH>
static int dgrad_s2_launch(const ConvArgs& a, hipStream_t st) {
    using G = Geo<NP, NCG, TH>;
    static PerDeviceOnce attr_once;           // first launch of this instance on any thread
    attr_once.run([&] {
        (void)hipFuncSetAttribute((const void*)dgrad3x3s2_kernel<NP, NCG, TH>, hipFuncAttributeMaxDynamicSharedMemorySize, G::SMEM_B);
    });
    const int tiles = a.N * (a.Ho / TH) * (a.Wo / TW);
    const int grid = tiles < 512 ? tiles : 512;                  // two ~70 KB, 4-wave workgroups per CU
#ifdef HDY_PROBE_BUILD      // timing ablations (results wrong) exist only in a -DHDY_PROBE_BUILD library, never in the shipped one
    const int abl = hdy_opt(HDY_OPT_DEEP_DEBUG);
    if constexpr (NP == 2) if (abl) {
#define ABL_CASE(v) if (abl == v) { (void)hipFuncSetAttribute((const void*)dgrad3x3s2_kernel<NP, NCG, TH, v>, hipFuncAttributeMaxDynamicSharedMemorySize, G::SMEM_B); hipLaunchKernelGGL((dgrad3x3s2_kernel<NP, NCG, TH, v>), dim3(grid), dim3(G::NTHR), G::SMEM_B, st, a); return (int)hipGetLastError(); }
        ABL_CASE(1) ABL_CASE(4) ABL_CASE(8) ABL_CASE(12) ABL_CASE(9) ABL_CASE(5) ABL_CASE(16)
#undef ABL_CASE
    }
#endif
    hipLaunchKernelGGL((dgrad3x3s2_kernel<NP, NCG, TH>), dim3(grid), dim3(G::NTHR), G::SMEM_B, st, a);
    return (int)hipGetLastError();
}

// Returns 1 and launches when the class-walk arguments describe one of the two layers; 0 = not eligible (the generic kernel runs).
int hdy_dgrad3x3s2_try(const ConvArgs& a, int dtype, hipStream_t st, int* rc) {
    const int disabled = hdy_opt(HDY_OPT_NO_DGRAD_S2);            // tests: 1 forces the class walk for A/B comparison, 2 only for the 64 <- 128 layer
    if (disabled == 1 || dtype != HDY_BF16 || a.ncls != 4 || a.nstat != 0 || a.res || a.scale || a.shift || a.act != 0 || a.stats) return 0;
    const int np = a.C / 64;                                     // dy planes
    if (!((a.C == 64 && a.K == 32) || (a.C == 128 && a.K == 64 && disabled != 2))) return 0;
    if (a.Ho % (np == 1 ? 8 : 4) || a.Wo % TW || a.Hin != a.Ho || a.Win != a.Wo || a.Hout != 2 * a.Ho || a.Wout != 2 * a.Wo) return 0;
    static const int dh[4] = {0, 0, 0, 0}, dw[4] = {0, 0, 0, 0}, nth[4] = {1, 1, 2, 2}, ntw[4] = {1, 2, 1, 2}, oh[4] = {0, 0, 1, 1}, ow[4] = {0, 1, 0, 1};
    for (int c = 0; c < 4; ++c)
        if (a.c_dh[c] != dh[c] || a.c_dw[c] != dw[c] || a.c_TH[c] != nth[c] || a.c_TW[c] != ntw[c] || a.c_oh[c] != oh[c] || a.c_ow[c] != ow[c] ||
            a.c_nkb[c] != nth[c] * ntw[c] * np || a.c_w[c] % 8)
            return 0;
    if (a.ldx % 8 || a.ldy % 8 || ((uintptr_t)a.x & 15) || ((uintptr_t)a.y & 15) || ((uintptr_t)a.w & 15)) return 0;
    hdy_note_dispatch(np == 1 ? "dgrad3x3s2_k64c32" : "dgrad3x3s2_k128c64");
    const int e = np == 1 ? dgrad_s2_launch<1, 2, 8>(a, st) : dgrad_s2_launch<2, 4, 4>(a, st);
    if (e != (int)hipSuccess) {
        hdy_set_error("dgrad3x3s2: launch failed: %s", hipGetErrorString((hipError_t)e));
        *rc = e;
        return 1;
    }
    *rc = HDY_OK;
    return 1;
}

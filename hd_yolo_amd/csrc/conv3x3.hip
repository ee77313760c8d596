// 3x3 / stride 1 / pad 1 convolution for 64 (or 32) input channels (bf16), filter resident in REGISTERS — the kernel behind the
// layer BASELINE.json names ("fused 3x3 conv at batch 64 x 640 x 640": yolov5s' 64->64 @80x80) and its dgrad.
//
// Why a second conv kernel: the generic implicit GEMM (conv_igemm.hip) fetches the A operand once per tap — 9x the
// activation bytes from L2 per output tile, plus the 73 KB filter per tile.  At ~2 us of L2 latency and 64 KB in flight per
// CU that caps it near 9 TB/s of L2->LDS traffic, i.e. ~350 TFLOP/s on this layer whatever the MFMA schedule does.
// Here a workgroup is 4 waves, persistent over ~6 output tiles of 8 x 16 pixels, and
//   * each wave owns 16 output channels and keeps their 9 x 64 filter slice as 18 MFMA row operands (72 VGPRs) for its
//     lifetime: no filter bytes in LDS, no filter reads in the MFMA loop;
//   * one (8+2) x (16+2) input patch (23 KB) per tile is staged by LDS-DMA, double buffered across tiles; all 9 taps read their
//     A fragments out of it: L2->LDS traffic per 256 outputs is 46 KB instead of 440 KB;
//   * with every tap in registers one patch-row fragment feeds all the (tile row, filter row) pairs that touch it: an 8-row
//     tile reads 10 x 6 fragments, not 8 x 18 (LDS read traffic per 256 outputs: 480 KB; 864 KB with the filter in LDS);
//   * LDS per workgroup is 2 patches + a separate staging tile = 61 KB and the kernel is capped at 256 VGPRs, so TWO workgroups
//     share a CU (2 waves per SIMD).  They are independent — own barriers — and the second one is dispatched ~3 us after the
//     first, so one's patch issue / epilogue / stores overlap the other's MFMAs.
// Fragment addressing: lane column = output pixel (ty, tx) of the tile, tap (r, s) reads patch pixel (ty+r)*18 + tx+s; the
// 16 lanes of a fragment read 16 consecutive patch pixels, and chunk ^ (((patch column>>1)&3)<<1) spreads them over the 16
// 16-byte slots of the 256-byte bank row.  ds_read_b128 is served in groups of 16 lanes that MIX two k-chunks (lanes 0-3, 12-15 of
// chunk c with lanes 4-11 of chunk c+1): keying on bits 1-2 only leaves bit 0 to tell the two chunks apart, which is conflict-free
// for every tap shift (a first key, (column>>1)&7, collided at odd shifts: 25 % of the LDS cycles were conflict cycles).
// LDS-DMA writes linearly, so that XOR is applied to the source address (rule: linear destination + permuted source + permuted read).
// The MFMAs run with the FILTER as the row operand, so a lane's 4 accumulator values are 4 consecutive output channels of ONE
// pixel: the epilogue packs them into one 8-byte LDS write, the BatchNorm sums stay in registers across all of the workgroup's
// tiles (one statistics slab per workgroup) and — a wave's channels being its own — leave without a cross-wave reduction.
// Work split: every workgroup gets floor(tiles / grid) whole tiles; the remaining tiles are cut into 2 or 4 row bands so that
// they still spread over all CUs (3200 tiles on 512 workgroups: 6 tiles + one 2-row band each).
//
// Measured on the BASELINE layer (64 x 80 x 80, 30.2 GFLOP; us per launch, same box):
//   39.6  first generation: one 8-wave workgroup per CU, filter in LDS (72 KB), 16x16 tiles, 6 LDS reads per 8 MFMAs
//   40.0  this design with the compiler's schedule (ds_read -> wait -> 3 MFMAs: a wave is LDS-latency bound, 4.4k cycles per item)
//   36.9  + fragment reads software-pipelined by hand (sched_barrier fences; 3.0k cycles per item alone, 2.3k = MFMA rate)
//   36.0  + those reads as inline asm with hand-counted lgkmcnt(3) waits (the compiler waits lgkmcnt(0) after every second group)
// and NOT adopted, all within +-0.5 us of 36.9 or worse: patch pieces issued from inside the MFMA loop (38.4: an LDS-DMA costs the
// wave ~350 cycles of issue time wherever it sits); the patch through registers (global_load + ds_write, 38.0) and the same with two
// register sets = two items of lead (37.0); a fragment prefetch distance of two groups (36.5 against 35.9); pairing each top-half fragment
// group with a bottom-half one so that no accumulator takes back-to-back MFMAs (35.8 against 35.9: the pipe forwards them); starting the second workgroup of
// a CU on its band item to shift its phase (37.5); first generation with 8-byte stores straight from the accumulators (43.3);
// deterministic anti-phasing — ONE 8-wave workgroup whose two 4-wave groups alternate "MFMAs of item s" and "epilogue of item s-1 +
// patch issue" in barrier-closed half-steps (46.7): a lone MFMA wave per SIMD reaches only ~62 % of the pipe rate (1.8-1.9k cycles
// per half item against 1.15k), which two independently scheduled workgroups hide by overlapping their MFMA phases, and 28 % of
// the cycles went to waiting at the half-step barriers.
// Round 2: waves as (pixel half) x (channel half) — 32 output channels and half the rows per wave, so each patch fragment is read by two
// waves instead of four (LDS reads -40 %, 18 MFMAs per fragment group instead of 9) at 144 filter VGPRs: dgrad 64<-64 33.0 us as before,
// 32<-32 51.1 against 56.5 us, but the forward with statistics needs more than 256 registers (56 B of scratch, 55.7 us).  The LDS reads
// are not what holds the 64-channel kernel back; not adopted.
// 4-row tiles with three workgroups per CU (36 KB of LDS, 168 VGPRs, no spills): forward 64->64 34.6-35.6 against 32.5-33 us, data gradient
// unchanged, 32->32 71 against 60 us — a third wave per SIMD does not pay for the halved MFMA work per item; not adopted.
// Ablation: without the patch fetch 35.1, without the output stores 34.8, without both 32.9 us — the kernel is not memory bound;
// the MFMA pipe is busy ~45 % of the cycles (1.7 GHz under this load), the rest is per-item work that two waves per SIMD do
// not overlap completely (patch issue ~2.0k cycles, statistics + staging ~0.6k, row stores ~0.5k, barriers ~0.4k per 8-row item).
//
// The same kernel is instantiated for C == 32 (yolov5s' 32->32 3x3 at 160x160 and its dgrad): 64-byte patch rows need no swizzle, one
// MFMA k-step per tap, 36 filter VGPRs; for K == 32 two of the four waves carry zero filters.  64 x 160 x 160, 32->32: 75 us against
// 134 us through the generic kernel (in the train step 70 + 58 us for forward + dgrad against 2 x ~103 us).
//
// Requirements (checked by the launcher, otherwise the generic kernel runs): bf16 in/out, C == 64 or 32, K <= 64, R = S = 3,
// stride 1, pad 1, H % 8 == 0, W % 16 == 0, 16-byte aligned rows.
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "hdyolo_internal.h"

__device__ uint4 g_hdy_zero16_c3[4];   // zero page for out-of-image patch pixels

namespace {

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

struct Item { int n, th, tw, row0, nrows; };   // rows [row0, row0 + nrows) of tile (n, th, tw); nrows in {8, 4, 2}

constexpr int NTHR = 256;                                            // 4 waves: wave = 16-channel group
constexpr int TH = 8, TW = 16, PW = TW + 2, PPIX = (TH + 2) * PW;    // 180 patch pixels
constexpr int STAGE_B = TH * TW * 128;                               // 16384
// per input-channel count C (64, or 32: the 3x3 of yolov5s' first C3): pixel row C*2 bytes = C/8 16-byte chunks, C/32 MFMA k-steps per tap
template <int C> struct Geo {
    static constexpr int CB = C * 2, CPP = C / 8, KS = C / 32;
    static constexpr int PATCH_B = PPIX * CB;                        // 23040 (C = 64)
    static constexpr int SMEM_B = 2 * PATCH_B + STAGE_B;             // 62464: two workgroups per CU
    static constexpr int NPASS = (PPIX * CPP + NTHR - 1) / NTHR;     // 6 loader passes (last partial)
};

// EPI: 0 = raw convolution out (train-mode forward, dgrad), 1 = scale/shift, 2 = scale/shift + SiLU
// STATS: accumulate BatchNorm partial sums (one slab per workgroup)
template <int C, int EPI, bool STATS>
__global__ __launch_bounds__(NTHR, C == 32 ? 3 : 2) void conv3x3_c64_kernel(const ConvArgs p) {      // C = 32: 39 KB of LDS, 36 filter VGPRs: three workgroups per CU
    constexpr int CB = Geo<C>::CB, CPP = Geo<C>::CPP, KS = Geo<C>::KS, PATCH_B = Geo<C>::PATCH_B, NPASS = Geo<C>::NPASS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sP = smem;                    // [2][180][C*2 B]
    unsigned char* sS = smem + 2 * PATCH_B;      // [128][128 B] staging tile

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int tiles_w = p.Wo / TW, tiles_h = p.Ho / TH;
    const int tiles_total = p.N * tiles_w * tiles_h;
    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const WorkSplit ws = work_split(tiles_total, (int)gridDim.x);        // here the bands are 8 / 4 / 2 rows
    const int nitems = ws.per + (wg < ws.units ? 1 : 0);

    const bf16_t* __restrict__ x = (const bf16_t*)p.x;
    const bf16_t* __restrict__ w = (const bf16_t*)p.w;
    const unsigned char* zero = (const unsigned char*)g_hdy_zero16_c3;

    // ---- filter slice -> registers: row operand of tap t, k-half ks = w[wave*16 + fr][t*64 + (ks*4 + fq)*8 .. +7]
    V16 bw[9][KS];
    {
        const int k = wave * 16 + fr;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const void* src = (k < p.K) ? (const void*)(w + (size_t)k * p.Kdp + t * C + (ks * 4 + fq) * 8) : (const void*)zero;
                bw[t][ks].i = *(const i32x4*)src;
            }
    }

    // ---- patch loader: each thread's 6 (pixel, chunk) slots of the 18-wide patch are the same for every item; only the origin,
    // the number of patch rows and the image-border test change
    int prel[NPASS], pyx[NPASS];
#pragma unroll
    for (int i = 0; i < NPASS; ++i) {
        const int pos = tid + NTHR * i;
        const int pix = pos / CPP;
        const int py = pix / PW, px = pix - py * PW;
        // 128-byte pixel rows (C = 64) are XOR-swizzled by the patch column; 64-byte rows (C = 32) need none: the 64 lanes of a fragment
        // read (16 consecutive pixels) x (4 chunks) = 1 KB contiguous
        const int lcp = CPP == 8 ? (tid & 7) ^ (((px >> 1) & 3) << 1) : (tid & (CPP - 1));
        prel[i] = (py * p.Win + px) * p.ldx + lcp * 8;
        pyx[i] = py | (px << 8);
    }
    // Items: whole tiles wg*per .. wg*per + per-1 (consecutive, so (n, th, tw) advances without divisions), then the band.
    auto first_item = [&]() {
        Item c;
        int t;
        if (ws.per > 0) {
            t = wg * ws.per;
            c.row0 = 0;
            c.nrows = TH;
        } else {
            t = wg / ws.bands;
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
    auto item_after = [&](Item c, int idx) {               // item idx + 1, given item idx
        if (idx + 1 < ws.per) {
            if (++c.tw == tiles_w) {
                c.tw = 0;
                if (++c.th == tiles_h) {
                    c.th = 0;
                    ++c.n;
                }
            }
            return c;
        }
        const int t = ws.per * (int)gridDim.x + wg / ws.bands;
        c.nrows = TH / ws.bands;
        c.row0 = (wg % ws.bands) * c.nrows;
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
        for (int i = 0; i < NPASS; ++i) {
            const int py = pyx[i] & 255;
            if ((wave * 64 + NTHR * i) / CPP >= prow * PW) break;          // wave-uniform: whole 1 KB pieces past the patch
            if (py >= prow) continue;                                     // lanes past the last pixel: no LDS write
            const int h = h0 + py, ww = w0 + (pyx[i] >> 8);
            const void* src = ((unsigned)h < (unsigned)p.Hin && (unsigned)ww < (unsigned)p.Win) ? (const void*)(org + prel[i]) : (const void*)zero;
            glds16(src, dst + (wave * 64 + NTHR * i) * 16);
        }
    };

    int aoff[3][KS];
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int px = fr + s;
            aoff[s][ks] = px * CB + ((CPP == 8 ? (ks * 4 + fq) ^ (((px >> 1) & 3) << 1) : fq) << 4);
        }

    float sc[4], sh[4], s1[4], s2[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int c = wave * 16 + fq * 4 + r;
        sc[r] = (p.scale && c < p.K) ? p.scale[c] : 1.0f;
        sh[r] = (p.shift && c < p.K) ? p.shift[c] : 0.0f;
        s1[r] = 0.f;
        s2[r] = 0.f;
    }
    const int st_ch = tid & 7, st_rr = tid >> 3;                              // store phase: chunk, first row (rows + 32 j)
    const int st_lds = st_rr * 128 + ((st_ch ^ ((st_rr >> 1) & 7)) << 4);
    const long long st_off = ((long long)(st_rr >> 4) * p.Wo + (st_rr & 15)) * p.ldy + st_ch * 8;
    const long long st_step = (long long)2 * p.Wo * p.ldy;
    const long long rs_off = ((long long)(st_rr >> 4) * p.Wo + (st_rr & 15)) * p.ldr + st_ch * 8;
    const long long rs_step = (long long)2 * p.Wo * p.ldr;
    // staging write: 8-byte slot (wave * 4 + fq) of pixel row fr, keyed by the row.  A ds_write_b64 is served in groups of 16 CONSECUTIVE lanes over 32 banks
    // (MI355X_MICROARCH.md, LDS table) — the 16 rows fr of one slot — so the key must take 16 rows to 16 slots: fr & 15 (rounds 1-3 used fr & 14: rows r and
    // r ^ 1 on one slot, every staging write 2-way conflicted, the 9 % LDS conflict cycles of profiles/r0[1-3]_conv3x3_counters.json).  An odd key swaps the
    // halves of a 16-byte chunk; the reader (whose rows all have the parity of st_rr) swaps them back in registers.
    const int ep_off = fr * 128 + (((wave * 4 + fq) ^ (fr & 15)) << 3);

    // ---- one item of AR rows (every wave covers all rows for its 16 channels)
    auto compute = [&](auto ar_c, const Item& it, int cur) {
        constexpr int AR = decltype(ar_c)::value;
        f32x4 acc[AR];
#pragma unroll
        for (int a = 0; a < AR; ++a) acc[a] = f32x4{0.f, 0.f, 0.f, 0.f};
        const unsigned char* pb = sP + cur * PATCH_B;
        // Software pipeline by hand: the next 3 fragments (half a patch row) are requested before the 9 MFMAs of the current 3 start,
        // and the scheduler is fenced from sinking them next to their uses (left alone it emits read -> wait -> 3 MFMAs, which
        // leaves a lone wave LDS-latency bound: 4.4k instead of 3.0k cycles per 8-row item; 2.3k is the MFMA rate).
        V16 fa[2][3];
        const unsigned pb_lds = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) unsigned char*)pb;
        unsigned abase[3][KS];
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) abase[s][ks] = pb_lds + aoff[s][ks];
        // inline-asm reads with hand-counted waits: the compiler's counter insertion waits lgkmcnt(0) right after every second
        // group's reads were issued and exposes their latency (a lone wave: 3.0k cycles per item instead of 2.3k)
#define C3_LOAD(G, F)                                                                                                              \
    {                                                                                                                              \
        _Pragma("unroll") for (int s_ = 0; s_ < 3; ++s_) {                                                                         \
            i32x4 v_;                                                                                                              \
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v_) : "v"(abase[s_][(G) % KS]), "n"(((G) / KS) * PW * CB));      \
            (F)[s_].i = v_;                                                                                                        \
        }                                                                                                                          \
    }
        C3_LOAD(0, fa[0])
#pragma unroll
        for (int g = 0; g < KS * (AR + 2); ++g) {        // group g = (patch row g / KS, k-step g % KS); patch row q serves tile row a with filter row r = q - a
            const int q = g / KS, ks = g % KS;
            if (g + 1 < KS * (AR + 2)) {
                C3_LOAD(g + 1, fa[(g + 1) & 1])
                asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(fa[g & 1][0].i), "+v"(fa[g & 1][1].i), "+v"(fa[g & 1][2].i));
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[g & 1][0].i), "+v"(fa[g & 1][1].i), "+v"(fa[g & 1][2].i));
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < 3; ++s)
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const int a = q - r;
                    if (a >= 0 && a < AR) acc[a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bw[r * 3 + s][ks].h, fa[g & 1][s].h, acc[a], 0, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0);
        }
#undef C3_LOAD
        if (STATS) {
#pragma unroll
            for (int a = 0; a < AR; ++a)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = acc[a][r];
                    s1[r] += v;
                    s2[r] = __builtin_fmaf(v, v, s2[r]);
                }
        }
        // the staging tile is separate from the patches: no barrier between a wave's last MFMA and its staging writes
#pragma unroll
        for (int a = 0; a < AR; ++a) {
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
            const long long org = (((long long)it.n * p.Ho + it.th * TH + it.row0) * p.Wo + it.tw * TW);
            bf16_t* yb = (bf16_t*)p.y + org * p.ldy + st_off;
            const bf16_t* rb = p.res ? (const bf16_t*)p.res + org * p.ldr + rs_off : nullptr;
#pragma unroll
            for (int j = 0; j < AR / 2; ++j) {             // 32 pixel rows of 128 bytes per pass
                V16 v;
                v.i = *(const i32x4*)(sS + st_lds + j * 32 * 128);
                if (st_rr & 1) v.i = i32x4{v.i[2], v.i[3], v.i[0], v.i[1]};
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
        // the next patch's DMA precedes these AR/2 stores in the wave's vm queue: wait for it, not for the stores
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AR / 2) : "memory");
        __syncthreads();                                   // next patch landed for everyone; staging tile free again
    };

    Item cur_it = first_item();
    issue_patch(cur_it, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int cur = 0;
    for (int idx = 0; idx < nitems; ++idx) {
        Item nxt_it = cur_it;
        if (idx + 1 < nitems) {
            nxt_it = item_after(cur_it, idx);
            issue_patch(nxt_it, cur ^ 1);
        }
        if (cur_it.nrows == 8) compute(std::integral_constant<int, 8>{}, cur_it, cur);
        else if (cur_it.nrows == 4) compute(std::integral_constant<int, 4>{}, cur_it, cur);
        else compute(std::integral_constant<int, 2>{}, cur_it, cur);
        cur ^= 1;
        cur_it = nxt_it;
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

template <int C, int EPI, bool STATS>
static void launch_c3(const ConvArgs& a, int grid, hipStream_t st) {
    static PerDeviceOnce attr_once;           // first launch of this instance on any thread
    attr_once.run([&] {
        (void)hipFuncSetAttribute((const void*)conv3x3_c64_kernel<C, EPI, STATS>, hipFuncAttributeMaxDynamicSharedMemorySize, Geo<C>::SMEM_B);
    });
    hdy_note_dispatch(C == 64 ? "conv3x3_c64" : "conv3x3_c32");
    hipLaunchKernelGGL((conv3x3_c64_kernel<C, EPI, STATS>), dim3(grid), dim3(NTHR), Geo<C>::SMEM_B, st, a);
}

template <int C>
static void launch_c3_c(const ConvArgs& a, int grid, int epi, hipStream_t st) {
    if (a.stats) {
        if (epi == 2) launch_c3<C, 2, true>(a, grid, st);
        else if (epi == 1) launch_c3<C, 1, true>(a, grid, st);
        else launch_c3<C, 0, true>(a, grid, st);
    } else {
        if (epi == 2) launch_c3<C, 2, false>(a, grid, st);
        else if (epi == 1) launch_c3<C, 1, false>(a, grid, st);
        else launch_c3<C, 0, false>(a, grid, st);
    }
}

}  // namespace


// Shape test shared by the launcher and the statistics-slab query (the two must agree on who writes the slabs).
static bool conv3x3_shape_ok(int C, int K, int R, int S, int stride, int pad, int H, int W, int dtype) {
    const bool disabled = hdy_opt(HDY_OPT_NO_CONV3X3) != 0;      // tests: force the generic kernel for A/B comparison
    return !disabled && dtype == HDY_BF16 && R == 3 && S == 3 && stride == 1 && pad == 1 && (C == 64 || C == 32) && K <= 64 && K % 8 == 0 && H % TH == 0 &&
           W % TW == 0;
}

static int conv3x3_grid(int tiles, int C) {
    const int g = hdy_opt(HDY_OPT_C3_GRID);
    const int cap = g ? g : (C == 32 ? 768 : 512);       // two 61 KB (C = 64) or three 39 KB (C = 32) 4-wave workgroups per CU
    return tiles < cap ? tiles : cap;
}

// Number of statistic slabs the filter-resident kernel writes for this shape (one per workgroup), 0 = not eligible.
int hdy_conv3x3_c64_slabs(int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dtype) {
    if (!conv3x3_shape_ok(C, K, R, S, stride, pad, H, W, dtype)) return 0;
    return conv3x3_grid(N * (H / TH) * (W / TW), C);
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
    const int grid = conv3x3_grid(a.N * (a.Ho / TH) * (a.Wo / TW), a.C);
    HDY_STAT_CAP(a, grid, "conv3x3_c64")
    const int epi = a.act == 1 ? 2 : ((a.scale || a.shift) ? 1 : 0);
    if (a.C == 32) launch_c3_c<32>(a, grid, epi, st);
    else launch_c3_c<64>(a, grid, epi, st);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        hdy_set_error("conv3x3_c64: launch failed: %s", hipGetErrorString(e));
        *rc = (int)e;
        return 1;
    }
    *rc = HDY_OK;
    return 1;
}

// Deep-pipelined implicit GEMM for the wide layers (C % 64 == 0, K >= 128; bf16) of gfx950: conv forward and stride-1 data gradient.
//
// Why a second generic kernel.  conv_igemm.hip waits for a WHOLE stage once per k-block: its `__syncthreads()` compiles to
// `s_waitcnt vmcnt(0) lgkmcnt(0); s_barrier` (the fence drains every LDS-DMA in flight, whatever counted wait precedes it), so no load
// ever crossed a barrier and a workgroup's bytes in flight were one stage.  PMC on 128->128 3x3 @40x40 (profiles/r03_layers_pmc.txt):
// MFMA busy 22 %, waves parked on waits 34 %, issue stalls 30 %.  This kernel follows cdna_hip_programming.md §5 "the 256^2 8-phase
// template": one 8-wave workgroup per CU, 256 x BN output tile, K-tiles of 64 channels cut into 16 KB UNITS (128 rows x 128 B) that are
// loaded by LDS-DMA several phases before they are read, counted `vmcnt` (never 0 in the loop), raw `s_barrier`, all LDS traffic as
// inline asm (hipcc inserts `vmcnt(0)` in front of compiler-visible LDS reads that may alias a pending LDS-DMA), the two waves of a SIMD
// staggered by one barrier so that one computes while the other loads.
//
//   BN = 256: waves 2 (M) x 4 (N), wave tile 128 x 64 as 2 x 2 quadrants of 64 x 32; units per K-tile [A-lo, B-lo, B-hi, A-hi] in a
//             2-deep ring of K-tile buffers (128 KB); 4 phases of 16 MFMAs per K-tile, one unit issued per phase two K-tiles ahead,
//             `vmcnt(8)` before a phase's first barrier: the four youngest units (64 KB) stay in flight.
//   BN = 128: waves 4 (M) x 2 (N), wave tile 64 x 64 as two halves of 32 x 64; units [A-lo, B, A-hi] in a 3-deep ring (144 KB);
//             2 phases per K-tile, waits vmcnt(8) / vmcnt(10).
//   Fragments are read ONE SEGMENT AHEAD (second version; stamps in profiles/r03_deep_stamps.txt): a compute segment is
//   [16 MFMAs][ds_reads of the next segment's new fragments into the registers those MFMAs just consumed][barrier]; the reads return
//   while the wave sits in the barriers and its SIMD partner computes.  The first version read them at the top of the phase
//   ([reads][barrier][lgkmcnt(0)][MFMAs]): 350 cycles of LDS wait in front of 256 cycles of MFMAs in every segment.
// Rules the schedule obeys (same guide, "Read a staged buffer one phase AFTER the wait that retires it"): a unit waited for before phase
// p's first barrier is read in phase p + 1 or later; a slot is restaged two phases after its last read (the staggered wave group reads
// one barrier later than the other).  The K-tile stream runs ACROSS output tiles, so the 1x1 layers (2-8 K-tiles per tile) keep their
// loads in flight through the epilogue.  The loader is conv_igemm.hip's descriptor loader (per-lane offsets constant per m-tile, one
// scalar offset per K-tile, out-of-image taps = the descriptor's range check).
//
// Epilogue per wave, no workgroup barrier and no LDS memory: 16 pixels x 32 channels at a time, lanes 16 apart trade 8-byte halves through
// ds_swizzle -> one 16-byte store per lane (64 contiguous bytes per pixel).  Train-mode BatchNorm sums stay in registers over all of a workgroup's tiles (one slab per
// workgroup position; BN = 128 instances only: the 256-wide tile has no registers to spare).
//
// Reference semantics replaced: nn.Conv2d inside metayolo/models/layers.py:31 (Conv), :92-93 (Bottleneck), :124-126 (C3), :179-180 (SPPF)
// and autograd's conv backward-data (train.py:472).
#include <stdio.h>

#include "common.h"
#include "hdyolo_internal.h"

// HDY_DEEP_DEBUG bit 32: shader-clock stamps of wave 0 (and wave 4) of every workgroup: [wg][0] cycles in the K-tile loop, [1] in the
// epilogues, [2] phases run, [3] whole kernel, [4..7] the same for wave 4 (measurement only; read back through hdy_deep_debug_read)
__device__ unsigned long long g_deep_dbg[256 * 8];
__device__ unsigned long long g_deep_seg[256 * 16];
__device__ unsigned long long g_deep_kt[256 * 8];       // HDY_DEEP_DEBUG bit 128, wave 0: cycles in K-tile 0 / 1 / 2 / later of the tiles that FOLLOW an epilogue, then the four counts     // [wg][wave 0 | wave 4][issue, vmcnt wait, lgkmcnt wait, barrier 1, MFMAs, reads + barrier 2, -, -] cycle sums (BN = 128)

// Tile boundary, measured in round 4 (profiles/r04_deep_tile_boundary.txt, scripts/probes/tile_boundary_fit.py): in the stamped build the first K-tile after an
// epilogue takes 7.8 k cycles against 2.3-2.6 k for every other one — the two wave groups run one barrier apart, so the group that leads waits at the next
// tile's second barrier while the other runs its epilogue: two epilogues back to back.  Three remedies were built on this kernel (kept in
// scripts/probes/retired/conv_deep_tile_boundary.hip): the groups brought level for the epilogue and staggered again (stamped build: 7.3 k -> 3.9 k cycles,
// kernel -8 %), the next K-tile's loads retired IN FRONT of the epilogue's stores (vmcnt retires in issue order and counts stores) with that K-tile's
// counted waits skipped, and the rows of an accumulating epilogue all requested before the first store.  In the SHIPPED build (no stamps) none of them moves
// anything: time = tiles * (K-tiles * 1.70 us + 2.7-3.5 us) + 5.8 us (9.2 us with BatchNorm sums) for 256 x 128 tiles of a 1x1 layer under every variant —
// the boundary is the 64 KB output tile leaving at the CU's share of HBM (24 GB/s: 2.7 us), which the other CUs' loads overlap; the stamps' serialisation
// is an artefact of their own `s_waitcnt lgkmcnt(0)`.  The old epilogue stays.
// -DHDY_DEEP_DBG=1 compiles the timing ablations / stamps in (scripts/ab_deep.sh builds that variant); the shipped kernel has none of their branches
#ifndef HDY_DEEP_DBG
#define HDY_DEEP_DBG 0
#endif

namespace {

constexpr int NTHR = 512;
constexpr int UNIT = 16384;

__device__ __forceinline__ unsigned fdiv(unsigned n, unsigned mg, int sh) { return __umulhi(n << 1, mg) >> sh; }

__device__ __forceinline__ void lds_dma16(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, unsigned lds_byte) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (void __attribute__((address_space(3)))*)(uintptr_t)lds_byte, 16, (int)voff, (int)soff, 0, 0);
}

#define DP_READ(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off) : "memory")
#define DP_VMCNT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")

template <int BN> struct Geo;
template <> struct Geo<256> {
    static constexpr int NB = 2, KT_BYTES = 4 * UNIT, RING = NB * KT_BYTES;
    static constexpr int WM = 2, MT = 4, NTQ = 2;                // per quadrant: MT m-tiles x NTQ n-tiles
    static constexpr int A_LO = 0, A_HI = UNIT, B_LO = 2 * UNIT, B_HI = 3 * UNIT;
    static constexpr int NACC = 2 * 2 * MT * NTQ;                 // f32x4 accumulators
};
template <> struct Geo<128> {
    static constexpr int NB = 3, KT_BYTES = 3 * UNIT, RING = NB * KT_BYTES;
    static constexpr int WM = 4, MT = 2, NTQ = 4;
    static constexpr int A_LO = 0, A_HI = UNIT, B_LO = 2 * UNIT, B_HI = 2 * UNIT;
    static constexpr int NACC = 2 * MT * NTQ;
};

// EPI: 0 raw output (train forward, data gradient), 1 scale / shift, 2 scale / shift + SiLU, 3 scale / shift + ReLU
template <int BN, bool STATS, int EPI>
__global__ __launch_bounds__(NTHR, 2) void conv_deep_kernel(const ConvArgs p) {
    using G = Geo<BN>;
    constexpr int MT = G::MT, NTQ = G::NTQ;
    constexpr int NBH = BN == 256 ? 2 : 1;                        // B halves per wave tile
    constexpr int COEF = G::RING;                                 // [2][BN] epilogue coefficients behind the ring
    static_assert(!(STATS && BN == 256), "BatchNorm sums: 128-wide instances only");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned lds0 = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) unsigned char*)smem;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int dbg = HDY_DEEP_DBG ? p.dbg : 0;
    const unsigned long long t_start = (dbg & 32) ? __builtin_readcyclecounter() : 0ull;
    unsigned long long t_loop = 0, t_epi = 0, n_ph = 0;
    unsigned long long seg[6] = {0, 0, 0, 0, 0, 0};
    unsigned long long ktc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define DP_STAMP(i, tprev) if (dbg & 64) { const unsigned long long t_ = __builtin_readcyclecounter(); seg[i] += t_ - tprev; tprev = t_; }
    const int grp = wave >> 2;                                    // stagger group: waves 4-7 are the SIMD partners of waves 0-3
    const int wr = BN == 256 ? (wave >> 2) : (wave & 3);
    const int wc = BN == 256 ? (wave & 3) : (wave >> 2);
    const int fr = lane & 15, fq = lane >> 4;

    const int ntiles = p.ntiles;                                  // column tiles of BN
    const int mtiles = (p.M + 255) >> 8;
    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int nt = wg % ntiles;                                   // this workgroup's column tile (grid % ntiles == 0)
    const int mstep = (int)gridDim.x / ntiles;
    const int mt0 = wg / ntiles;
    if (mt0 >= mtiles) {
        if (STATS) {                                              // its slab row is read by the finalize launch: zeros
            for (int j = tid; j < 2 * BN; j += NTHR) {
                const int which = j / BN, c = j - which * BN;
                if (nt * BN + c < p.K) p.stats[((size_t)mt0 * 2 + which) * p.K + nt * BN + c] = 0.f;
            }
        }
        return;
    }
    // stride-2 data gradient: the four parity classes of a spatial tile back to back (tile j -> m-tile j >> 2, class j & 3), each with its
    // own tap window, K-tile count and packed filter block; the K-tile stream runs across them like across any other tile boundary
    const bool walk = p.ncls > 1;
    const int my_tiles = (mtiles - mt0 + mstep - 1) / mstep * (walk ? 4 : 1);
    const int n0 = nt * BN;

    // ------------------------------------------------------------------ loader (conv_igemm.hip's descriptor loader, 4 A rows per thread)
    constexpr unsigned OOB = 0x80000000u;
    const int r0 = tid >> 3;
    const int lc = (tid & 7) ^ ((tid >> 4) & 7);                  // logical 16-byte chunk fetched into physical slot (tid & 7)
    const bf16_t* __restrict__ x = (const bf16_t*)p.x;
    unsigned roff[4], inv[4];
    bool any_inv = true;
    const int HoWo = p.Ho * p.Wo;
    const int rowB = p.Win * p.ldx * 2, pixB = p.ldx * 2;
    __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, OOB, 0x00020000);
    // filter rows beyond the packed block (last column tile of a K that is not a multiple of BN) fail the range check: zeros
    const unsigned wbytes = (unsigned)((size_t)((p.K + p.bn - 1) / p.bn * p.bn) * p.Kdp * 2);
    __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, wbytes, 0x00020000);
    constexpr int BR = BN / 64;                                   // filter rows per thread
    unsigned woff[BR];
#pragma unroll
    for (int i = 0; i < BR; ++i) woff[i] = (unsigned)(((n0 + r0 + 64 * i) * p.Kdp) * 2 + lc * 16);
    int ld_j = 0, ld_kt = 0, ld_buf = 0;                          // loader position: tile number, K-tile inside it, ring buffer
    int l_th0 = p.dh0 - p.uh0, l_tw0 = p.dw0 - p.uw0, l_TW = l_tw0 + p.TW, l_nkt = p.Kdp >> 6;
    int s_th = l_th0, s_tw = l_tw0, s_cb = 0;
    bool ld_live = true;

    auto loader_set_tile = [&](int j) {
        if (walk) {
            const int cls = j & 3;
            j >>= 2;
            l_th0 = p.c_dh[cls] - p.uh0; l_tw0 = p.c_dw[cls] - p.uw0;
            l_TW = l_tw0 + p.c_TW[cls];
            l_nkt = p.c_nkb[cls];
            const int kdp = l_nkt << 6;
            rw = __builtin_amdgcn_make_buffer_rsrc((void*)((const bf16_t*)p.w + p.c_w[cls]), 0,
                                                   (unsigned)((size_t)((p.K + p.bn - 1) / p.bn * p.bn) * kdp * 2), 0x00020000);
#pragma unroll
            for (int i = 0; i < BR; ++i) woff[i] = (unsigned)(((n0 + r0 + 64 * i) * kdp) * 2 + lc * 16);
            s_th = l_th0; s_tw = l_tw0;
            if (cls != 0) return;                                 // same pixels as the class before: the row offsets and tap bits stand
        }
        const int mb = (mt0 + j * mstep) << 8;
        // lanes 0..31 own the wave's 32 rows: row 8 * wave + (lane & 7) + 64 * ((lane >> 3) & 3)
        const int m = mb + 8 * wave + (lane & 7) + 64 * ((lane >> 3) & 3);
        unsigned my_off, my_inv;
        if (p.pointwise) {
            rx = __builtin_amdgcn_make_buffer_rsrc((void*)((const unsigned char*)x + (long long)mb * pixB), 0, OOB, 0x00020000);
            my_off = (unsigned)((m - mb) * pixB);
            my_inv = m < p.M ? 0x80000000u : 0xFFFFFFFFu;
        } else {
            const int nb = (int)fdiv((unsigned)mb, p.mg_howo, p.sh_howo), remb = mb - nb * HoWo;
            const int oib = (int)fdiv((unsigned)remb, p.mg_wo, p.sh_wo), ojb = remb - oib * p.Wo;
            const int hb = oib * p.ih_mul + p.uh0, wb = ojb * p.iw_mul + p.uw0;
            rx = __builtin_amdgcn_make_buffer_rsrc((void*)((const unsigned char*)x + (((long long)nb * p.Hin + hb) * p.Win + wb) * pixB), 0, OOB, 0x00020000);
            const int mc = min(m, p.M - 1);
            const int n = (int)fdiv((unsigned)mc, p.mg_howo, p.sh_howo), rem = mc - n * HoWo;
            const int oi = (int)fdiv((unsigned)rem, p.mg_wo, p.sh_wo), oj = rem - oi * p.Wo;
            const int h = oi * p.ih_mul + p.uh0, w_ = oj * p.iw_mul + p.uw0;
            my_off = (unsigned)((((n - nb) * p.Hin + (h - hb)) * p.Win + (w_ - wb)) * pixB);
            unsigned wmask = 0;
            for (int tw = 0; tw < p.UW; ++tw) wmask |= ((unsigned)(w_ + tw) >= (unsigned)p.Win ? 1u : 0u) << tw;
            const unsigned full = (1u << p.UW) - 1u;
            my_inv = 0x80000000u;
            for (int th = 0; th < p.UH; ++th) my_inv |= ((unsigned)(h + th) >= (unsigned)p.Hin ? full : wmask) << (th * p.UW);
            if (m >= p.M) my_inv = 0xFFFFFFFFu;
        }
        any_inv = __builtin_amdgcn_ballot_w64((my_inv & 0x7FFFFFFFu) != 0) != 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int src = (i * 8 + (lane >> 3)) << 2;
            roff[i] = (unsigned)__builtin_amdgcn_ds_bpermute(src, (int)my_off) + (unsigned)(lc * 16);
            inv[i] = (unsigned)__builtin_amdgcn_ds_bpermute(src, (int)my_inv);
        }
        // settle the cross-lane results here: left pending, hipcc waits for them at their first use inside the phases with a counted
        // lgkmcnt that also drains the phase's own (inline-asm, invisible to it) fragment reads
        asm volatile("" : "+v"(roff[0]), "+v"(roff[1]), "+v"(roff[2]), "+v"(roff[3]), "+v"(inv[0]), "+v"(inv[1]), "+v"(inv[2]), "+v"(inv[3]));
    };

    // one 16 KB unit = two DMA instructions per thread (rows r0 and r0 + 64 of the unit)
    auto issue_a = [&](int half) {                                // half 0: A-lo (rows 0..127 of the tile), 1: A-hi
        if (!ld_live || (dbg & 1)) return;
        const unsigned dst = lds0 + (unsigned)(ld_buf * G::KT_BYTES + (half ? G::A_HI : G::A_LO) + wave * 1024);
        const unsigned koff = (unsigned)(s_th * rowB + s_tw * pixB + s_cb * 2);
        if (any_inv) {
            const unsigned bit = (unsigned)(s_th * p.UW + s_tw);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int out = __builtin_amdgcn_sbfe(inv[2 * half + i], bit, 1);
                lds_dma16(rx, (out & (int)OOB) | roff[2 * half + i], koff, dst + 8192 * i);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i) lds_dma16(rx, roff[2 * half + i], koff, dst + 8192 * i);
        }
    };
    auto issue_b = [&](int half) {                                // BN = 256: half 0 / 1 = filter rows 0..127 / 128..255 of the column tile
        if (!ld_live || (dbg & 2)) return;
        const unsigned dst = lds0 + (unsigned)(ld_buf * G::KT_BYTES + (half ? G::B_HI : G::B_LO) + wave * 1024);
        const unsigned wk = (unsigned)(ld_kt * 128);
#pragma unroll
        for (int i = 0; i < 2; ++i) lds_dma16(rw, woff[(BN == 256 ? 2 * half : 0) + i], wk, dst + 8192 * i);
    };
    auto loader_advance = [&]() {                                 // after the last unit of the loader's K-tile
        if (!ld_live) return;
        ld_buf = ld_buf + 1 == G::NB ? 0 : ld_buf + 1;
        if (++ld_kt == l_nkt) {
            ld_kt = 0;
            s_th = l_th0; s_tw = l_tw0; s_cb = 0;
            if (++ld_j < my_tiles) loader_set_tile(ld_j);
            else ld_live = false;
        } else {
            s_cb += 64;
            if (s_cb >= p.C) {
                s_cb = 0;
                if (++s_tw == l_TW) { s_tw = l_tw0; ++s_th; }
            }
        }
    };

    // ------------------------------------------------------------------ accumulators, fragments
    f32x4 acc[G::NACC];
#pragma unroll
    for (int i = 0; i < G::NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    // index of accumulator (a: A half, b: B half, m, n)
    auto AI = [](int a, int b, int m, int n) constexpr { return ((a * NBH + b) * MT + m) * NTQ + n; };
    V16 af[MT][2], bf0[NTQ][2], bf1[BN == 256 ? NTQ : 1][2];                 // BN = 256: A quadrant-half, B-lo, B-hi fragments
    V16 afl[2][2], afh[2][2], bfa[4][2];                                     // BN = 128: A-lo / A-hi / B fragments, each refilled right after its last use
    // fragment addresses inside a K-tile buffer: 16 rows further down is +2048 bytes with the same chunk swizzle
    const int arow = (BN == 256 ? wr * 64 : wr * 32) + fr, brow = (BN == 256 ? wc * 32 : wc * 64) + fr;
    unsigned fa[2], fb[2];
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) {
        fa[kh] = lds0 + (unsigned)(arow * 128 + (((kh * 4 + fq) ^ ((arow >> 1) & 7)) << 4));
        fb[kh] = lds0 + (unsigned)(brow * 128 + (((kh * 4 + fq) ^ ((brow >> 1) & 7)) << 4));
    }
    unsigned cbuf = 0;                                            // byte offset of the consumer's K-tile buffer

    float s1[STATS ? NTQ * 4 : 1], s2[STATS ? NTQ * 4 : 1];
#pragma unroll
    for (int i = 0; i < (STATS ? NTQ * 4 : 1); ++i) s1[i] = s2[i] = 0.f;

    // epilogue coefficients of this workgroup's column tile (constant for the launch)
    if (EPI >= 1) {
        float* coef = (float*)(smem + COEF);
        for (int j = tid; j < BN; j += NTHR) {
            coef[j] = (p.scale && n0 + j < p.K) ? p.scale[n0 + j] : 1.0f;
            coef[BN + j] = (p.shift && n0 + j < p.K) ? p.shift[n0 + j] : 0.0f;
        }
    }

    // ------------------------------------------------------------------ prologue: everything the steady state has issued "before phase 1 of K-tile 0"
    loader_set_tile(0);
    if (BN == 256) {
        issue_a(0); issue_b(0); issue_b(1); issue_a(1); loader_advance();     // K-tile 0
        issue_a(0); issue_b(0); issue_b(1);                                   // K-tile 1: A-lo, B-lo, B-hi (its A-hi follows in ph1 of K-tile 0)
    } else {
        issue_a(0); issue_b(0); issue_a(1); loader_advance();                 // K-tile 0
        issue_a(0); issue_b(0); issue_a(1); loader_advance();                 // K-tile 1
        issue_a(0); issue_b(0);                                               // K-tile 2: A-lo, B (its A-hi follows in ph1 of K-tile 0)
    }
    if (ld_live) { DP_VMCNT(BN == 256 ? 8 : 10); } else { DP_VMCNT(0); }     // BN = 256: A-lo, B-lo, B-hi of K-tile 0 landed; 128: all of K-tile 0
    __syncthreads();                                              // also: the coefficients are in LDS (a full fence once, before the loop)
    __builtin_amdgcn_s_barrier();
    if constexpr (BN == 256) {
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
#pragma unroll
            for (int n = 0; n < 2; ++n) DP_READ(bf0[n][kh].i, fb[kh], G::B_LO + n * 2048);
#pragma unroll
            for (int m = 0; m < 4; ++m) DP_READ(af[m][kh].i, fa[kh], G::A_LO + m * 2048);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (BN == 128) {
        // fragments of K-tile 0's first half: from here on every MFMA segment reads the NEXT segment's fragments while it computes
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
#pragma unroll
            for (int n = 0; n < 4; ++n) DP_READ(bfa[n][kh].i, fb[kh], G::B_LO + n * 2048);
#pragma unroll
            for (int m = 0; m < 2; ++m) DP_READ(afl[m][kh].i, fa[kh], G::A_LO + m * 2048);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    }
    if (grp == 1) __builtin_amdgcn_s_barrier();                   // stagger: waves 4-7 run one barrier behind waves 0-3 from here on

    bf16_t* __restrict__ y = (bf16_t*)p.y;

#define DP_WAIT(n)                                             \
    {                                                          \
        if (ld_live) { DP_VMCNT(n); } else { DP_VMCNT(0); }    \
    }
#define DP_MFMA_PHASE(A_, B_, BF)                                                                                                   \
    if (!(dbg & 4)) {                                                                                                             \
        __builtin_amdgcn_s_setprio(1);                                                                                              \
        _Pragma("unroll") for (int kh = 0; kh < 2; ++kh)                                                                            \
            _Pragma("unroll") for (int m = 0; m < MT; ++m)                                                                          \
                _Pragma("unroll") for (int n = 0; n < NTQ; ++n)                                                                     \
                    acc[AI(A_, B_, m, n)] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(BF[n][kh].h, af[m][kh].h, acc[AI(A_, B_, m, n)], 0, 0, 0); \
        __builtin_amdgcn_s_setprio(0);                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                                          \
    }
#define DP_READ_A(UOFF)                                                                                         \
    if (!(dbg & 16)) _Pragma("unroll") for (int kh = 0; kh < 2; ++kh)                                                            \
        _Pragma("unroll") for (int m = 0; m < MT; ++m) DP_READ(af[m][kh].i, fa[kh] + cbuf, (UOFF) + m * 2048);
#define DP_READ_B(BF, UOFF)                                                                                      \
    if (!(dbg & 16)) _Pragma("unroll") for (int kh = 0; kh < 2; ++kh)                                                             \
        _Pragma("unroll") for (int n = 0; n < NTQ; ++n) DP_READ(BF[n][kh].i, fb[kh] + cbuf, (UOFF) + n * 2048);
#define DP_LGKM0_A() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0);

    // the asm reads name their destinations as outputs; the wait below is a barrier for the scheduler (rule 18 of the guide)
    for (int j = 0; j < my_tiles; ++j) {
        const int m0 = (mt0 + (walk ? j >> 2 : j) * mstep) << 8;
        const int nkt = walk ? p.c_nkb[j & 3] : p.Kdp >> 6;          // K-tiles of this output tile
        const unsigned long long t_a = (dbg & 32) ? __builtin_readcyclecounter() : 0ull;
        for (int kt = 0; kt < nkt; ++kt) {
            const unsigned long long t_kt = (dbg & 128) ? __builtin_readcyclecounter() : 0ull;
            if constexpr (BN == 256) {
                // fragments one segment ahead, as in the 128-wide loop below: every MFMA segment ends by requesting the next segment's new
                // fragments into the registers its MFMAs have just consumed
                //   ph1 (a0,b0) af x bf0; then reads B-hi(t) -> bf1      issues A-hi(t+1)   waits (vmcnt 8) for A-hi(t)
                //   ph2 (a0,b1) af x bf1; then reads A-hi(t) -> af       issues A-lo(t+2)
                //   ph3 (a1,b1) af x bf1; no reads                       issues B-lo(t+2)   waits for A-lo, B-lo of t+1
                //   ph4 (a1,b0) af x bf0; then reads A-lo, B-lo of t+1   issues B-hi(t+2)   waits for B-hi(t+1)
                const unsigned cnext = cbuf ^ (unsigned)G::KT_BYTES;
                issue_a(1);
                loader_advance();
                DP_WAIT(8)
                DP_LGKM0_A()
                __builtin_amdgcn_s_barrier();
                DP_MFMA_PHASE(0, 0, bf0)
                DP_READ_B(bf1, G::B_HI)
                __builtin_amdgcn_s_barrier();
                // ---- phase 2
                issue_a(0);
                DP_LGKM0_A()
                __builtin_amdgcn_s_barrier();
                DP_MFMA_PHASE(0, 1, bf1)
                DP_READ_A(G::A_HI)
                __builtin_amdgcn_s_barrier();
                // ---- phase 3
                issue_b(0);
                DP_WAIT(8)
                DP_LGKM0_A()
                __builtin_amdgcn_s_barrier();
                DP_MFMA_PHASE(1, 1, bf1)
                __builtin_amdgcn_s_barrier();
                // ---- phase 4
                issue_b(1);
                DP_WAIT(8)
                __builtin_amdgcn_s_barrier();
                DP_MFMA_PHASE(1, 0, bf0)
                cbuf = cnext;
                DP_READ_B(bf0, G::B_LO)
                DP_READ_A(G::A_LO)
                __builtin_amdgcn_s_barrier();
            } else {
                // BN = 128, fragments read one segment ahead: a compute segment is 16 MFMAs with NO LDS wait in front of them (measured on the
                // first version, which read its fragments at the top of the phase: lgkmcnt wait ~350 + MFMAs 256 cycles per segment, 1230
                // cycles per phase).  Right after its MFMAs are issued a segment requests the NEXT segment's fragments into the registers those
                // MFMAs have just read (an LDS return takes longer than an MFMA holds its A / B operands); they arrive while the wave sits in
                // the barriers and the partner wave computes, and are retired by the lgkmcnt(0) in front of the next phase's first barrier
                // (so a slot restaged two phases after its last read is safe for the staggered group too).
                //   ph1 (a0): MFMA afl x B; then reads A-hi(t) -> afh;              issues A-hi(t+2);     waits for A-lo, B of t+1 (vmcnt 8)
                //   ph2 (a1): MFMA afh x B; then reads A-lo, B of t+1 -> afl, B;    issues A-lo, B (t+3); waits for A-hi(t+1)       (vmcnt 10)
                const unsigned cnext = cbuf + (unsigned)G::KT_BYTES == (unsigned)G::RING ? 0u : cbuf + (unsigned)G::KT_BYTES;
#define DP128_MFMA(A_, AF)                                                                                                                  \
    if (!(dbg & 4)) {                                                                                                                       \
        __builtin_amdgcn_s_setprio(1);                                                                                                      \
        _Pragma("unroll") for (int kh = 0; kh < 2; ++kh)                                                                                    \
            _Pragma("unroll") for (int m = 0; m < 2; ++m)                                                                                   \
                _Pragma("unroll") for (int n = 0; n < 4; ++n)                                                                               \
                    acc[AI(A_, 0, m, n)] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfa[n][kh].h, AF[m][kh].h, acc[AI(A_, 0, m, n)], 0, 0, 0); \
        __builtin_amdgcn_s_setprio(0);                                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                                                  \
    }
                unsigned long long tp = (dbg & 64) ? __builtin_readcyclecounter() : 0ull;
                issue_a(1);
                loader_advance();
                DP_STAMP(0, tp)
                DP_WAIT(8)
                DP_STAMP(1, tp)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                DP_STAMP(2, tp)
                __builtin_amdgcn_s_barrier();
                DP_STAMP(3, tp)
                DP128_MFMA(0, afl)
                DP_STAMP(4, tp)
                if (!(dbg & 16)) {
#pragma unroll
                    for (int kh = 0; kh < 2; ++kh)
#pragma unroll
                        for (int m = 0; m < 2; ++m) DP_READ(afh[m][kh].i, fa[kh] + cbuf, G::A_HI + m * 2048);
                }
                __builtin_amdgcn_s_barrier();
                DP_STAMP(5, tp)
                issue_a(0);
                issue_b(0);
                DP_STAMP(0, tp)
                DP_WAIT(10)
                DP_STAMP(1, tp)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                DP_STAMP(2, tp)
                __builtin_amdgcn_s_barrier();
                DP_STAMP(3, tp)
                DP128_MFMA(1, afh)
                DP_STAMP(4, tp)
                if (!(dbg & 16)) {
#pragma unroll
                    for (int kh = 0; kh < 2; ++kh) {
#pragma unroll
                        for (int m = 0; m < 2; ++m) DP_READ(afl[m][kh].i, fa[kh] + cnext, G::A_LO + m * 2048);
#pragma unroll
                        for (int n = 0; n < 4; ++n) DP_READ(bfa[n][kh].i, fb[kh] + cnext, G::B_LO + n * 2048);
                    }
                }
                __builtin_amdgcn_s_barrier();
                DP_STAMP(5, tp)
#undef DP128_MFMA
                cbuf = cnext;
            }
            if ((dbg & 128) && j > 0) {
                const int slot = kt < 3 ? kt : 3;
                ktc[slot] += __builtin_readcyclecounter() - t_kt;
                ktc[4 + slot] += 1;
            }
        }
        // Tile boundary.  The wave groups run one barrier apart and a group that reaches a barrier waits for the other: left staggered, waves 0-3 run their
        // epilogue while waves 4-7 wait at the barrier behind their last MFMAs, then waves 4-7 run theirs while waves 0-3 wait at the next tile's second barrier
        // — two epilogues back to back.  With an activation in the epilogue (two quarter-rate transcendentals per output: the eval / inference instances) that
        // is worth removing: the groups are brought level here (the leading group waits one barrier), run their epilogues side by side and are staggered again
        // in front of the next tile — C4 shapes with SiLU, one box (profiles/r04_c4_shapes_ab.txt): 512->512 1x1 @64x64 507 -> 461 us, 256->256 1x1 @128x128
        // 718 -> 644, 1024->1024 1x1 @32x32 397 -> 373, 256->256 3x3 @64x64 729 -> 688.  The raw epilogue of the training instances (EPI = 0) measured
        // the same or 2-6 % slower with it (see the file header): it keeps the stagger.
        constexpr bool ALIGN = EPI >= 1;
        if (ALIGN && grp == 0) __builtin_amdgcn_s_barrier();
        const unsigned long long t_b = (dbg & 32) ? __builtin_readcyclecounter() : 0ull;
        t_loop += t_b - t_a;
        n_ph += (unsigned long long)nkt * (BN == 256 ? 4 : 2);
        // ---------------------------------------------------------------- epilogue of the finished tile (per wave, no barrier, no LDS memory)
        // A lane holds 4 consecutive channels of pixel fr for each 16-channel tile; lanes fq and fq ^ 1 (16 lanes apart) trade halves of a
        // PAIR of channel tiles through the LDS crossbar (ds_swizzle, no LDS memory, no staging tile) so that every lane owns 8 consecutive
        // channels = one 16-byte store, 64 contiguous bytes per pixel and pass.  The swizzles are compiler-visible: hipcc counts and
        // pipelines them over the passes (the staged version waited for an LDS round trip per pass: 40 % of the 1x1 layers' time).
        if (!(dbg & 8)) {
            const int odd = fq & 1;
            const int cl = odd ? 16 + (fq - 1) * 4 : fq * 4;       // this lane's first channel inside the 32-channel pass
#pragma unroll
            for (int np = 0; np < NBH * NTQ / 2; ++np) {           // 32 channels of the wave tile at a time; pass = 16 pixels x 32 channels
                const int b = BN == 256 ? np : 0, nb = BN == 256 ? 0 : 2 * np;
                const int cb = BN == 256 ? b * 128 + wc * 32 : wc * 64 + np * 32;      // first channel of the pass inside the column tile
                // the 32 channels' coefficients once, not once per pass (as first written every pass paid an LDS round trip with its own wait:
                // 64 of them per wave and 256 x 256 tile)
                f32x4 csc[2], csh[2];
                if (EPI >= 1) {
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const unsigned ca = lds0 + (unsigned)(COEF + (cb + q * 16 + fq * 4) * 4);
                        asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:%3" : "=&v"(csc[q]), "=&v"(csh[q]) : "v"(ca), "n"(BN * 4));
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(csc[0]), "+v"(csh[0]), "+v"(csc[1]), "+v"(csh[1]));
                }
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int m = 0; m < MT; ++m) {
                        unsigned pk[2][2];
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            f32x4 v = acc[AI(a, b, m, nb + q)];
                            if (STATS && !(dbg & 512)) {
#pragma unroll
                                for (int r = 0; r < 4; ++r) {
                                    s1[(nb + q) * 4 + r] += v[r];
                                    s2[(nb + q) * 4 + r] = __builtin_fmaf(v[r], v[r], s2[(nb + q) * 4 + r]);
                                }
                            }
                            if (EPI >= 1) {
                                const f32x4 sc = csc[q], sh = csh[q];
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
                        // even fq keeps channel tile q = 0 and receives the partner's; odd fq keeps q = 1
                        const unsigned s0 = odd ? pk[0][0] : pk[1][0], s1_ = odd ? pk[0][1] : pk[1][1];
                        const unsigned r0_ = (unsigned)__builtin_amdgcn_ds_swizzle((int)s0, 0x401F);       // lane ^ 16
                        const unsigned r1_ = (unsigned)__builtin_amdgcn_ds_swizzle((int)s1_, 0x401F);
                        V16 o;
                        o.i = odd ? i32x4{(int)r0_, (int)r1_, (int)pk[1][0], (int)pk[1][1]} : i32x4{(int)pk[0][0], (int)pk[0][1], (int)r0_, (int)r1_};
                        const int mg = m0 + a * 128 + (BN == 256 ? wr * 64 : wr * 32) + m * 16 + fr;
                        const int kc = n0 + cb + cl;
                        size_t opix = (size_t)mg;
                        if (walk) {                                // class pixel (n, oi, oj) -> its place in the strided output image
                            const int mc = min(mg, p.M - 1);
                            const int n = (int)fdiv((unsigned)mc, p.mg_howo, p.sh_howo), rem = mc - n * HoWo;
                            const int oi = (int)fdiv((unsigned)rem, p.mg_wo, p.sh_wo), oj = rem - oi * p.Wo;
                            opix = ((size_t)n * p.Hout + (p.c_oh[j & 3] + oi * p.oh_mul)) * p.Wout + (p.c_ow[j & 3] + oj * p.ow_mul);
                        }
                        if (mg < p.M && kc < p.K) {
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
                            if (!(dbg & 256)) *(i32x4*)(y + opix * p.ldy + kc) = o.i;
                            else asm volatile("" ::"v"(o.i));          // timing ablation: the epilogue's arithmetic without its stores
                        }
                    }
            }
        }
#pragma unroll
        for (int i = 0; i < G::NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (dbg & 32) t_epi += __builtin_readcyclecounter() - t_b;
        if (ALIGN && grp == 1 && j + 1 < my_tiles) __builtin_amdgcn_s_barrier();      // stagger again
    }
    if ((dbg & 32) && lane == 0 && (wave & 3) == 0 && blockIdx.x < 256) {
        unsigned long long* o = g_deep_dbg + blockIdx.x * 8 + (wave >> 2) * 4;
        o[0] = t_loop; o[1] = t_epi; o[2] = n_ph; o[3] = __builtin_readcyclecounter() - t_start;
        for (int i = 0; i < 6; ++i) g_deep_seg[blockIdx.x * 16 + (wave >> 2) * 8 + i] = seg[i];
        if (wave == 0) for (int i = 0; i < 8; ++i) g_deep_kt[blockIdx.x * 8 + i] = ktc[i];
    }
    if (EPI == 0 && grp == 0) __builtin_amdgcn_s_barrier();       // re-align the two wave groups (EPI >= 1: level since the last tile's epilogue)
#undef DP_WAIT
#undef DP_MFMA_PHASE
#undef DP_READ_A
#undef DP_READ_B
#undef DP_LGKM0_A

    if (STATS && !(dbg & 1024)) {
        // one slab row per workgroup position (mt0): lane-local sums -> 16 pixel lanes (shuffles) -> the wr waves that share the channels (LDS).
        // No LDS-DMA is in flight here (the loader ran dry and its last units were waited for with vmcnt(0) inside the K loop), so the barriers below wait
        // for LDS traffic only — raw barriers: `__syncthreads()` would also wait (vmcnt(0)) for the last tile's output stores, ~2.5 us of acknowledgement
        // latency in front of a reduction that does not depend on them (scripts/probes/tile_boundary_fit.py: 9.2 us of launch + prologue + tail with the
        // sums against 5.8 us without)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");                            // the barrier builtin is IntrNoMem: keep the C++ `red[]` accesses below on this side of it
        float* red = (float*)smem;                                // [WM][BN][2]; the ring is free
#pragma unroll
        for (int n = 0; n < NTQ; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float u = s1[n * 4 + r], q = s2[n * 4 + r];
                u = row_sum16(u);
                q = row_sum16(q);
                if (fr == 0) {
                    const int col = wc * 64 + n * 16 + fq * 4 + r;
                    red[(wr * BN + col) * 2 + 0] = u;
                    red[(wr * BN + col) * 2 + 1] = q;
                }
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");                            // (as above: the loads of `red[]` stay behind the barrier)
        for (int j = tid; j < 2 * BN; j += NTHR) {
            const int which = j / BN, c = j - which * BN;
            if (n0 + c < p.K) {
                float v = 0.f;
#pragma unroll
                for (int g = 0; g < G::WM; ++g) v += red[(g * BN + c) * 2 + which];
                p.stats[((size_t)mt0 * 2 + which) * p.K + n0 + c] = v;
            }
        }
    }
}

template <int BN> constexpr size_t deep_smem() { return (size_t)Geo<BN>::RING + 2 * BN * sizeof(float); }

// column tile: 256 where K fills it and there are enough 256 x 256 tiles for the chip, else 128 (HDY_DEEP_BN forces one)
inline int deep_bn(long long M, int K) {
    const int forced = hdy_opt(HDY_OPT_DEEP_BN);
    if (forced == 128 || forced == 256) return forced;
    const long long mtiles = (M + 255) / 256;
    // measured: 128 wins while there are fewer than ~2 tiles of 256 x 256 per CU (more, smaller tiles balance better), 256 from K = 512 up
    // K = 256 (a single 256-wide column tile): only with four or more tiles per CU — yolov5l inference at B = 128, 1024 x 1024: network 52.5 -> 51.3 ms
    // (1024 or 512 as the threshold alike, 4096 no change); the 400-tile layers of the train steps stay on 128
    if (K % 256 != 0) return 128;
    const long long tiles = mtiles * (K / 256);
    return (K >= 512 ? tiles >= 192 : tiles >= 1024) ? 256 : 128;
}

// workgroups: one per CU, a multiple of the column tiles, never more than there are tiles
inline int deep_grid(long long M, int ntiles) {
    const long long mtiles = (M + 255) / 256;
    long long g = 256 / ntiles * ntiles;
    if (g > mtiles * ntiles) g = mtiles * ntiles;
    return (int)g;
}

template <int BN, bool STATS, int EPI>
int deep_launch(const ConvArgs& a, int grid, hipStream_t st) {
    constexpr size_t smem = deep_smem<BN>();
    static_assert(smem <= 160 * 1024, "LDS budget");
    static PerDeviceOnce attr_once;
    attr_once.run([&] {
        (void)hipFuncSetAttribute((const void*)conv_deep_kernel<BN, STATS, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    });
    hipLaunchKernelGGL((conv_deep_kernel<BN, STATS, EPI>), dim3(grid), dim3(NTHR), smem, st, a);
    HDY_LAUNCH_CHECK("conv_deep");
    return HDY_OK;
}

template <int BN, bool STATS>
int deep_launch_epi(const ConvArgs& a, int grid, hipStream_t st) {
    const bool affine = a.scale != nullptr || a.shift != nullptr;
    if (a.act == 1) return deep_launch<BN, STATS, 2>(a, grid, st);
    if (a.act == 2) return deep_launch<BN, STATS, 3>(a, grid, st);
    if (affine) return deep_launch<BN, STATS, 1>(a, grid, st);
    return deep_launch<BN, STATS, 0>(a, grid, st);
}

// shapes the deep kernel takes; bn_out: its column tile.  `stats`: the launch writes BatchNorm slabs (128-wide instances only).
// First version (fragments read at the top of a phase): 1x1 layers 5-17 % faster than conv_igemm.hip, 3x3 layers 3-8 % slower.  With the
// fragments read one segment ahead: yolov5s train step 12.37 -> 12.17 ms, yolov5l inference network 56.0 -> 49.6 ms with every eligible
// layer here (HDY_DEEP_ALL = 0 keeps the multi-tap layers on the generic kernel).
bool deep_shape_ok(long long M, int C, int K, int taps, bool pointwise, bool stats, int* bn_out, int classes = 1) {
    if (hdy_opt(HDY_OPT_NO_DEEP)) return false;
    if (!pointwise && !hdy_opt(HDY_OPT_DEEP_ALL)) return false;
    if (C % 64 != 0 || K < 128 || K % 8 != 0) return false;
    const int bn = stats ? 128 : deep_bn(M, K);
    const long long tiles = (M + 255) / 256 * cdiv(K, bn) * classes;
    if (tiles < hdy_opt(HDY_OPT_DEEP_MIN_TILES)) return false;     // too few 256-row tiles for 256 CUs: the 128-row kernel spreads wider
    if (bn_out) *bn_out = bn;
    return true;
}

}  // namespace

// measurement only (not part of the C ABI header): copies the stamp table of the last HDY_DEEP_DEBUG & 32 launch to the host
extern "C" int hdy_deep_debug_read(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_deep_dbg), sizeof(unsigned long long) * 256 * 8);
}
extern "C" int hdy_deep_debug_read_ktiles(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_deep_kt), sizeof(unsigned long long) * 256 * 8);
}
extern "C" int hdy_deep_debug_read_segments(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_deep_seg), sizeof(unsigned long long) * 256 * 16);
}

// statistic slabs a deep-kernel forward writes (one per workgroup position); 0: the shape is not the deep kernel's
int hdy_conv_deep_slabs(long long M, int C, int K, int taps, int pointwise, int dtype) {
    int bn = 0;
    if (dtype != HDY_BF16 || !deep_shape_ok(M, C, K, taps, pointwise != 0, true, &bn)) return 0;
    const int ntiles = cdiv(K, bn);
    return deep_grid(M, ntiles) / ntiles;
}

// called by hdy_conv_igemm_launch after its own validation (a.M, a.Kdp, a.pointwise, union tap window, reciprocals are set)
int hdy_conv_deep_try(const ConvArgs& a_in, int dtype, int out_f32, hipStream_t st, int* rc) {
    const bool walk = a_in.ncls > 1;                               // the four-class stride-2 data gradient (validated by the caller)
    if (dtype != HDY_BF16 || out_f32 || a_in.nstat > 0 || (!a_in.dense_out && !walk) || a_in.span_pixels || !a_in.vec_out || !a_in.utap) return 0;
    // measured at the yolov5s bench shapes (B = 64, scripts/probes/dgrad_walk.py): 105 / 80 / 62 / 48 us here against 94 / 86 / 59 / 51 us for
    // conv_igemm.hip's 128-row walk, and 12.37-12.45 against 12.30 ms in the train step.  Opt-in.
    // HDY_DEEP_WALK: 0 never, 1 always, 2 (default) where it measured faster: 256 or more gradient channels out (256<-512 @40x40 80.5 vs 85.9 us, 256<-256 47.5 vs 51.1)
    if (walk && (hdy_opt(HDY_OPT_DEEP_WALK) == 0 || (hdy_opt(HDY_OPT_DEEP_WALK) == 2 && a_in.K < 256))) return 0;
    int bn = 0;
    if (!deep_shape_ok(a_in.M, a_in.C, a_in.K, a_in.TH * a_in.TW, a_in.pointwise != 0, a_in.stats != nullptr, &bn, walk ? 4 : 1)) return 0;
    if (a_in.ldx % 8 != 0) return 0;
    ConvArgs a = a_in;
    a.dbg = hdy_opt(HDY_OPT_DEEP_DEBUG);
    a.ntiles = cdiv(a.K, bn);
    // a.bn stays the PACKING tile (rows of the packed filter block are padded to it)
    const int grid = deep_grid(a.M, a.ntiles);
    HDY_STAT_CAP(a, grid / a.ntiles, "conv_deep")
    hdy_note_dispatch(bn == 256 ? (walk ? "deep_256x256_walk" : "deep_256x256") : (walk ? "deep_256x128_walk" : "deep_256x128"));
    if (bn == 256) *rc = deep_launch_epi<256, false>(a, grid, st);
    else if (a.stats) *rc = deep_launch_epi<128, true>(a, grid, st);
    else *rc = deep_launch_epi<128, false>(a, grid, st);
    return 1;
}

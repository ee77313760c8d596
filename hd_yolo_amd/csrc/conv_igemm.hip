// Implicit-GEMM "tap convolution" for gfx950: the one kernel behind conv forward, dgrad (stride 1 and the
// four parity classes of stride 2) and the 6x6/s2 stem (as 6 row-taps over a 24-wide pseudo channel).
//
//   y[n, oh_off + i*oh_mul, ow_off + j*ow_mul, k] (+)= act( scale[k] * SUM_{t,c} x[n, i*ih_mul + dh0 + t/TW,
//                                                     j*iw_mul + dw0 + t%TW, c] * w[k][t][c] + shift[k] ) + res
//
// Activations are NHWC with an explicit pixel pitch (ldx / ldy elements), so a tensor may be a channel
// slice of a wider concat buffer; out-of-image taps read as zero.  GEMM view: M = N*Ho*Wo pixels,
// N = K output channels, Kd = T*C.  Weights are pre-packed [Kpad][Kdp] (K-contiguous per output channel,
// zero padded), see conv_wgrad.hip:pack_weight_kernel.
//
// CDNA4 mapping
//  * 256 threads = 4 waves as 2(M) x 2(N); block tile 128 x BN x (128 bytes of K); each wave owns 64 x BN/2 as
//    16x16 MFMA tiles.  One template serves both arithmetic types: LDS rows are 128 B (64 bf16 / 32 f32), a
//    fragment is one 16-byte ds_read_b128 per lane, mma16() is one v_mfma_f32_16x16x32_bf16 or four
//    v_mfma_f32_16x16x4_f32 (exact fp32: the 1e-4 parity mode).
//  * staging is LDS-DMA (buffer_load_dwordx4 ... lds): no staging VGPRs, no ds_write pass (ds_write_b128 moves
//    ~79 B/clk/CU and was the limiter of the register-staged version).  The DMA writes LDS linearly
//    (wave base + lane*16), so the XOR swizzle that makes the fragment reads conflict-free is applied to the
//    SOURCE address: lane l of a row fetches logical chunk (l&7) ^ ((row>>1)&7).  Padding / tail lanes carry an
//    offset beyond the descriptor's range and receive zeros (a masked lane would leave stale LDS bytes).
//  * persistent workgroups: each block walks a contiguous range of output tiles; the loader runs one k-block
//    ahead of the MFMAs ACROSS tile boundaries, so the 1x1 layers (one or two k-blocks per tile) still overlap
//    their loads with compute and the per-launch ramp is paid once.  Tile ranges follow the XCD-aware order,
//    so the n-tiles of one m-tile and neighbouring m-tiles (3x3 halos) share an XCD's L2.
//  * epilogue: accumulators -> (scale, shift, SiLU) -> bf16 -> XOR-swizzled LDS tile -> 16-byte coalesced
//    row stores (+ residual / accumulate on the way out).  fp32 outputs and ragged K use a direct path.
//
// Train-mode BatchNorm: with `stats`, each tile also writes per-channel partial sum / sum-of-squares of its
// fp32 accumulators (rows >= M are zero by construction) to stats[mtile][2][K]; bn_finalize reduces the slabs
// deterministically (no atomics).
//
// Reference semantics replaced: nn.Conv2d inside metayolo/models/layers.py:31 (Conv), :92-93 (Bottleneck),
// :124-126 (C3), :179-180 (SPPF), yolo_head.py:112 (det conv), and autograd's conv backward-data.
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "hdyolo_internal.h"

namespace {

template <typename T> struct Traits;
template <> struct Traits<float> { static constexpr int VE = 4; };
template <> struct Traits<bf16_t> { static constexpr int VE = 8; };

template <typename T> __device__ __forceinline__ f32x4 mma16(const V16& a, const V16& b, f32x4 c);
template <> __device__ __forceinline__ f32x4 mma16<bf16_t>(const V16& a, const V16& b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.h, b.h, c, 0, 0, 0);
}
template <> __device__ __forceinline__ f32x4 mma16<float>(const V16& a, const V16& b, f32x4 c) {
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.f[0], b.f[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.f[1], b.f[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.f[2], b.f[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.f[3], b.f[3], c, 0, 0, 0);
    return c;
}

__device__ __forceinline__ int swz(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

// LDS-DMA through a buffer descriptor: lane l's 16 bytes at base + voff + soff land at lds + l*16; a lane whose offset fails the
// descriptor's range check (>= num_records) gets zeros.
__device__ __forceinline__ void lds_dma16(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, unsigned char* lds_wave_base) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (void __attribute__((address_space(3)))*)lds_wave_base, 16, (int)voff, (int)soff, 0, 0);
}

// n / d for n < 2^31 with the host-made reciprocal of hdy_magic(): mulhi(2n, mg) >> sh
__device__ __forceinline__ unsigned fdiv(unsigned n, unsigned mg, int sh) { return __umulhi(n << 1, mg) >> sh; }

// BM x BN output tile, NTHR = 2*BM threads (BM/64 x 2 waves, each 64 x BN/2), NS-deep LDS ring of (A | B) stages.
//   <128, *, 2>  4 waves, 2 stages: many small workgroups per CU (K <= 64, small problems)
//   <256, *, 3>  8 waves (2 per SIMD), 3 stages with counted vmcnt: half the filter re-fetch per output, two k-blocks of
//                DMA in flight across the barrier, one workgroup per CU (the K >= 128 layers, which were L2-fetch bound)
// STAT: the epilogue also serves p.stat[] (BatchNorm-backward statistics of the tensor whose gradient this launch completes)
template <typename T, typename OT, int BM, int BN, int NS, bool STAT = false>
// second bound = waves per SIMD the LDS footprint allows (2 / 3 / 4 co-resident 4-wave workgroups, 1 x 8 waves): caps the VGPRs there
// (STAT instances: 2 — the statistics operands of the store loop do not fit the 128 / 168 registers of 4 / 3 waves per SIMD)
__global__ __launch_bounds__(2 * BM, (BM == 256 || BN == 128 || STAT) ? 2 : (BN == 64 ? 3 : 4)) void conv_igemm_kernel(const ConvArgs p) {
    constexpr int NTHR = 2 * BM;
    constexpr int VE = Traits<T>::VE;
    constexpr int BKE = 8 * VE;          // elements per 128-byte k-block
    constexpr int MT = 4;                // 16-row tiles per wave (64 rows)
    constexpr int NT = BN / 32;          // 16-col tiles per wave
    constexpr int AR = 4;                // A rows staged per thread (BM*8 chunks / NTHR)
    constexpr int BR = BN * 8 / NTHR;    // B rows staged per thread
    constexpr int RSTEP = NTHR / 8;      // row distance between a thread's staged rows
    constexpr int LPS = AR + BR;         // DMA instructions per wave per stage
    constexpr int ASZ = BM * 128, BSZ = BN * 128, STAGE = ASZ + BSZ;
    constexpr bool VEC_OUT = sizeof(OT) == 2;
    static_assert(BR >= 1, "tile too narrow for this thread count");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // [NS][A | B]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 15, fq = lane >> 4;
    const int ntiles = p.ntiles;
    const int mtiles = (p.M + BM - 1) / BM;
    // class walk (stride-2 dgrad in one launch): tile index = (m-tile, n-tile, parity class), class innermost, so that the four
    // half-line writes of a 2x2 output block leave one CU back to back (they merge in L2) and dy tiles are re-read from L2
    const bool walk = p.ncls > 1;
    const int tiles_total = mtiles * ntiles * (walk ? 4 : 1);
    const int tpb = (tiles_total + (int)gridDim.x - 1) / (int)gridDim.x;
    // Tile order.  One column tile: a workgroup walks a CONTIGUOUS range (neighbouring m-tiles share 3x3 halos, BatchNorm sums stay in
    // registers).  Several column tiles per m-tile: the tiles that read the same A rows are consecutive indices, and walking them one after
    // the other in one workgroup re-read A from the Infinity Cache — by then 64 workgroups per XCD had streamed 8+ MB through its 4 MB L2
    // (PMC: 256->256 1x1 fetched its input twice).  Interleaved (workgroup w takes tiles w, w + grid, ...) those tiles run at the same time
    // on neighbouring workgroups of one XCD and share its L2: yolov5l inference network 59.1 -> 56.5 ms, yolov5s train step 13.46 -> 13.35 ms.
    // The four parity classes of the stride-2 dgrad stay with one workgroup (bit 1 of the switch): interleaved their half-line writes no longer
    // leave one CU back to back and the step lost 0.1 ms (also when limited to the layers whose pixel rows are whole cache lines).  Interleaving
    // the single-column-tile layers too changes nothing (12.95 / 12.97 vs 12.98 / 13.03 ms).
    const bool interleave = (walk && (p.tile_interleave & 2) && p.K * (int)sizeof(OT) >= 128) || (!walk && ntiles > 1 && (p.tile_interleave & 1));
    const int wg_id = xcd_remap(blockIdx.x, gridDim.x);
    const int tile_step = interleave ? (int)gridDim.x : 1;
    const int tile_begin = interleave ? wg_id : wg_id * tpb;
    const int tile_end = interleave ? tiles_total : min(tile_begin + tpb, tiles_total);
    if (tile_begin >= tile_end) return;
    const int nkb = p.Kdp / BKE;

    const T* __restrict__ x = (const T*)p.x;
    const T* __restrict__ w = (const T*)p.w;

    // ------------------------------------------------------------------ loader (runs NS-1 k-blocks ahead)
    // The generic layers are bound by how many instructions a k-block costs, not by bytes (PMC on 32->64 3x3/s2 @320x320: 13 VALU
    // instructions per MFMA, 60 % of them address arithmetic of the LDS-DMA), so the loader keeps everything it can out of the VALU:
    //  * both operands go through buffer descriptors (`buffer_load_dwordx4 ... offen lds`): a 32-bit per-lane offset that is CONSTANT for
    //    a whole m-tile (row origin + the lane's 16-byte chunk) plus a scalar offset for the k-block (tap and channel block), so a k-block
    //    whose 128 bytes lie inside one tap (C % BKE == 0: `utap`) costs no vector instruction at all for interior rows;
    //  * padding is the descriptor's range check: a tap outside the image sets the offset to 2^31 = num_records, and the DMA writes
    //    zeros (probed: scripts/probes/buffer_lds_oob.hip; the range check includes the scalar offset).  Per row one bit per tap of the
    //    union window says "outside"; a wave whose rows are all interior skips the test;
    //  * the descriptor base moves with the m-tile (first row's window origin, 64-bit scalar arithmetic), so offsets stay far below 2^31
    //    whatever the tensor size;
    //  * row -> (image, row, column) is a multiply-high by host-made reciprocals, done by ONE lane per row (lanes 0..31 of a wave own
    //    its 32 rows) and handed to the 8 lanes of the row by ds_bpermute; the LDS destination of every DMA is scalar (m0 by SALU).
    // Measured and NOT adopted earlier (same kernel, flat addressing): the two waves of a SIMD alternating as loader of a stage; taps as
    // the inner k-loop; skipping the filter block once every ring slot holds it; a resident filter with a 7-deep activation ring; a
    // 256x256 tile (numbers in DESIGN.md §4).
    const int wave_s = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r0 = tid >> 3;
    const int lc = (tid & 7) ^ ((tid >> 4) & 7);       // logical chunk fetched into physical slot (tid & 7)
    constexpr unsigned OOB = 0x80000000u;               // = num_records of both descriptors
    constexpr int ES = (int)sizeof(T);
    unsigned roff[AR];                                   // byte offset of the row's window-origin pixel (+ lc * 16) from the tile base
    unsigned inv[AR];                                    // bit (th * UW + tw): that tap of the union window is outside the image; bit 31: always
    unsigned woff[BR];                                   // filter rows: byte offset of (row r0 + RSTEP * i, chunk lc) from the column tile's first row
    bool any_inv = true;                                 // wave-uniform: some row of this wave has an outside tap
    int ld_tile = tile_begin, ld_kb = 0, ld_mtile = -1;
    const int HoWo = p.Ho * p.Wo;
    __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, OOB, 0x00020000);
    __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, OOB, 0x00020000);
    // tap window of the loader's current tile inside the union window (origin uh0 / uw0): l_THc x l_TWc taps from (l_th0, l_tw0)
    int l_th0 = p.dh0 - p.uh0, l_tw0 = p.dw0 - p.uw0, l_TW = l_tw0 + p.TW, l_nkb = nkb;
    unsigned l_mg_tw = p.mg_tw[0];
    int l_sh_tw = p.sh_tw[0], l_TWc = p.TW, l_THc = p.TH;
    int s_th = 0, s_tw = 0, s_cb = 0;                    // utap: scalar tap and channel block of the next k-block
    unsigned w_soff = 0;                                 // scalar byte offset of the column tile's first filter row
    int l_Kdp = p.Kdp;
    const int rowB = p.Win * p.ldx * ES, pixB = p.ldx * ES;

    auto set_woff = [&]() {
#pragma unroll
        for (int i = 0; i < BR; ++i) woff[i] = (unsigned)((r0 + RSTEP * i) * l_Kdp * ES + lc * 16);
    };
    set_woff();

    auto loader_set_tile = [&](int t) {
        if (walk) {
            const int cls = t & 3;
            t >>= 2;
            l_th0 = p.c_dh[cls] - p.uh0; l_tw0 = p.c_dw[cls] - p.uw0;
            l_THc = p.c_TH[cls]; l_TWc = p.c_TW[cls];
            l_TW = l_tw0 + l_TWc;
            l_mg_tw = p.mg_tw[cls]; l_sh_tw = p.sh_tw[cls];
            l_nkb = p.c_nkb[cls];
            l_Kdp = l_nkb * BKE;
            rw = __builtin_amdgcn_make_buffer_rsrc((void*)(w + p.c_w[cls]), 0, OOB, 0x00020000);
            set_woff();
        }
        const int mt = t / ntiles, nt = t - mt * ntiles;
        if (mt != ld_mtile) {
            const int mb = mt * BM;                      // first row of the tile (always < M)
            // this lane's row: lanes 0..31 (and their copies 32..63) own rows 8 * wave + (lane & 7) + RSTEP * ((lane >> 3) & 3)
            const int m = mb + 8 * wave_s + (lane & 7) + RSTEP * ((lane >> 3) & 3);
            unsigned my_off, my_inv;
            if (p.pointwise) {                           // input pixel == output pixel: nothing to recover
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
            for (int i = 0; i < AR; ++i) {
                const int src = (i * 8 + (lane >> 3)) << 2;
                roff[i] = (unsigned)__builtin_amdgcn_ds_bpermute(src, (int)my_off) + (unsigned)(lc * 16);
                inv[i] = (unsigned)__builtin_amdgcn_ds_bpermute(src, (int)my_inv);
            }
            ld_mtile = mt;
        }
        w_soff = (unsigned)(nt * BN * l_Kdp * ES);
        s_th = l_th0; s_tw = l_tw0; s_cb = 0;
    };

    auto loader_issue = [&](int buf) {
        unsigned char* sa = smem + buf * STAGE + wave_s * 1024;
        unsigned char* sb = sa + ASZ;
        if (p.utap) {
            // the whole k-block lies inside tap (s_th, s_tw), channels [s_cb, s_cb + BKE): one scalar offset, one mask bit
            const unsigned koff = (unsigned)(s_th * rowB + s_tw * pixB + s_cb * ES);
            if (any_inv) {
                const unsigned bit = (unsigned)(s_th * p.UW + s_tw);
#pragma unroll
                for (int i = 0; i < AR; ++i) {
                    const int out = __builtin_amdgcn_sbfe(inv[i], bit, 1);       // 0 / -1
                    lds_dma16(rx, (out & (int)OOB) | roff[i], koff, sa + NTHR * 16 * i);
                }
            } else {
#pragma unroll
                for (int i = 0; i < AR; ++i) lds_dma16(rx, roff[i], koff, sa + NTHR * 16 * i);
            }
        } else {
            // k-blocks straddle taps (C < BKE or C % BKE != 0): this lane's chunk has its own tap and channel
            const unsigned ke = (unsigned)(ld_kb * BKE + lc * VE);
            const unsigned tap = fdiv(ke, p.mg_c, p.sh_c);
            const int c = (int)(ke - tap * (unsigned)p.C);
            const int thr = (int)fdiv(tap, l_mg_tw, l_sh_tw), twr = (int)tap - thr * l_TWc;
            const int th = l_th0 + thr, tw = l_tw0 + twr;
            const unsigned koff = (unsigned)(th * rowB + tw * pixB + c * ES - lc * 16);     // roff[] carries lc * 16 (the utap form of the chunk offset)
            const unsigned bit = thr < l_THc ? (unsigned)(th * p.UW + tw) : 31u;    // beyond the last tap: the zero padding of the last k-block
#pragma unroll
            for (int i = 0; i < AR; ++i) {
                const int out = __builtin_amdgcn_sbfe(inv[i], bit, 1);
                lds_dma16(rx, (out & (int)OOB) | (roff[i] + koff), 0, sa + NTHR * 16 * i);
            }
        }
        const unsigned wk = w_soff + (unsigned)(ld_kb * 128);
#pragma unroll
        for (int i = 0; i < BR; ++i) lds_dma16(rw, woff[i], wk, sb + NTHR * 16 * i);
        // advance to the next k-block, possibly of the next tile
        if (++ld_kb == l_nkb) {
            ld_kb = 0;
            if ((ld_tile += tile_step) < tile_end) loader_set_tile(ld_tile);
        } else {
            s_cb += BKE;
            if (s_cb >= p.C) {
                s_cb = 0;
                if (++s_tw == l_TW) { s_tw = l_tw0; ++s_th; }
            }
        }
    };

    f32x4 acc[MT][NT];
    auto zero_acc = [&]() {
#pragma unroll
        for (int a = 0; a < MT; ++a)
#pragma unroll
            for (int b = 0; b < NT; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    };

    // ------------------------------------------------------------------ epilogue of one finished tile
    // The MFMAs run with the FILTER fragment as row operand, so lane (fr, fq) holds, for each (a, b), 4 consecutive output channels
    // (column wn*BN/2 + b*16 + fq*4 + r) of ONE output row (wm*64 + a*16 + fr): 8 packed bytes per (a, b) into the staging tile
    // instead of four 2-byte scatters, and per-channel sums that add up lane-locally.
    // BatchNorm sums: with a single column tile (K <= 128, every large layer) they stay in registers for all of the workgroup's
    // tiles and are written as ONE slab per workgroup at the end; otherwise each tile reduces its 16 pixel lanes and writes the
    // slab rows of its 128-row groups.  (Per tile the single-tap layers spent more cycles here than in their MFMAs.)
    // BatchNorm sums stay in registers across ALL of a workgroup's tiles when every one of them is the same column tile: a single column
    // tile, or the interleaved order with grid % ntiles == 0 (tile index = m-tile * ntiles + n-tile, step = grid).  One slab row per
    // workgroup position then (wg_id / ntiles), each of the ntiles workgroups that share it filling its BN columns.  Otherwise one slab per
    // 128 output rows from every tile's epilogue: 32 values x 4 shuffle steps, two barriers and a slab write per tile — on the 1x1 layers
    // with 4-8 k-blocks per tile that was a third of the kernel (256->256 @40x40: 45.7 -> 30.5 us without it).
    const bool wg_stats = p.stats != nullptr && (ntiles == 1 || (interleave && !walk && (int)gridDim.x % ntiles == 0));
    // epilogue scale / shift of the current column tile: [2][BN] floats behind the ring — except for the 32-wide tile (always a single
    // column tile, and 4 workgroups fill the LDS exactly), whose 4 + 4 values per lane are loaded once into registers
    constexpr bool COEF_LDS = BN > 32;
    float* const coef = (float*)(smem + NS * STAGE);
    int coef_ntile = -1;
    f32x4 sc_reg = {1.f, 1.f, 1.f, 1.f}, sh_reg = {0.f, 0.f, 0.f, 0.f};
    if (!COEF_LDS) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int kc = wn * (BN / 2) + fq * 4 + r;
            if (p.scale && kc < p.K) sc_reg[r] = p.scale[kc];
            if (p.shift && kc < p.K) sh_reg[r] = p.shift[kc];
        }
    }
    float s1[NT][4], s2[NT][4];
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) s1[b][r] = s2[b][r] = 0.f;

    // ---- producer-side BatchNorm-backward statistics (STAT): a thread of the coalesced store loop owns ONE 16-byte channel chunk for the
    // whole launch (single column tile), so its 8 + 8 partial sums stay in registers across all of the workgroup's tiles
    float bs1[8], bs2[8];
    int st_req = -1;
    if constexpr (STAT) {
#pragma unroll
        for (int e = 0; e < 8; ++e) bs1[e] = bs2[e] = 0.f;
        const int kc0 = (tid % (BN * 2 / 16)) * 8;
        for (int r = 0; r < p.nstat; ++r)
            if (kc0 >= p.stat[r].c0 && kc0 < p.stat[r].c1) st_req = r;
    }

    // lane-local sums -> [BM/64 wave rows][BN][2] in `red` (16 pixel lanes by shuffles)
    auto stats_to_lds = [&](float* red, float (&u1)[NT][4], float (&u2)[NT][4]) {
#pragma unroll
        for (int b = 0; b < NT; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float u = u1[b][r], q = u2[b][r];
                u = row_sum16(u);
                q = row_sum16(q);
                if (fr == 0) {
                    const int col = wn * (BN / 2) + b * 16 + fq * 4 + r;
                    red[(wm * BN + col) * 2 + 0] = u;
                    red[(wm * BN + col) * 2 + 1] = q;
                }
            }
    };

    // returns true when the coalesced path issued its fixed number of row stores per wave (full tile)
    auto epilogue = [&](int t, unsigned char* scratch) -> bool {
        int oh_off = p.oh_off, ow_off = p.ow_off;
        if (walk) {
            oh_off = p.c_oh[t & 3]; ow_off = p.c_ow[t & 3];
            t >>= 2;
        }
        const int mtile = t / ntiles, ntile = t - mtile * ntiles;
        const int m0 = mtile * BM, n0 = ntile * BN;
        if (p.stats) {
            if (wg_stats) {
#pragma unroll
                for (int b = 0; b < NT; ++b)
#pragma unroll
                    for (int a = 0; a < MT; ++a)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float v = acc[a][b][r];
                            s1[b][r] += v;
                            s2[b][r] = __builtin_fmaf(v, v, s2[b][r]);
                        }
            } else {
                float t1[NT][4], t2[NT][4];
#pragma unroll
                for (int b = 0; b < NT; ++b)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float u = 0.f, q = 0.f;
#pragma unroll
                        for (int a = 0; a < MT; ++a) {
                            const float v = acc[a][b][r];
                            u += v;
                            q = __builtin_fmaf(v, v, q);
                        }
                        t1[b][r] = u;
                        t2[b][r] = q;
                    }
                float* red = (float*)scratch;       // [BM/64][BN][2]
                stats_to_lds(red, t1, t2);
                __syncthreads();
                // one slab per 128 output rows (what the caller sized): pairs of wave rows
                for (int j = tid; j < (BM / 128) * BN; j += NTHR) {
                    const int half = j / BN, c = j - half * BN;
                    const size_t slab = (size_t)mtile * (BM / 128) + half;
                    if (n0 + c < p.K && (long long)slab * 128 < p.M) {
                        p.stats[(slab * 2 + 0) * p.K + n0 + c] = red[((2 * half) * BN + c) * 2] + red[((2 * half + 1) * BN + c) * 2];
                        p.stats[(slab * 2 + 1) * p.K + n0 + c] = red[((2 * half) * BN + c) * 2 + 1] + red[((2 * half + 1) * BN + c) * 2 + 1];
                    }
                }
                __syncthreads();
            }
        }
        const bool affine = p.scale != nullptr || p.shift != nullptr;
        if (COEF_LDS && affine && ntile != coef_ntile) {       // per-channel epilogue coefficients of this column tile -> LDS (once per workgroup
            __syncthreads();                       // when the layer has a single column tile)
            for (int j = tid; j < BN; j += NTHR) {
                coef[j] = (p.scale && n0 + j < p.K) ? p.scale[n0 + j] : 1.0f;
                coef[BN + j] = (p.shift && n0 + j < p.K) ? p.shift[n0 + j] : 0.0f;
            }
            __syncthreads();
            coef_ntile = ntile;
        }
        OT* __restrict__ y = (OT*)p.y;
        const bool vec4_rows = sizeof(OT) == 4 && (((uintptr_t)p.y | (uintptr_t)p.res) & 15) == 0 && p.ldy % 4 == 0 && (!p.res || p.ldr % 4 == 0);
        if (VEC_OUT && p.vec_out) {
            // stage RPP rows of the bf16 tile at a time in the A part of the finished stage; 8-byte slots XORed with a row key keep the 16 pixel lanes of a
            // write on different slots (a ds_write_b64 is banked over 16 consecutive lanes and 32 banks: the key takes all 16 rows apart — row & 15, or
            // (row >> 1) & 7 for 64-byte rows whose parity already picks the bank half; rounds 1-3 keyed with row & 14: 2-way conflicts, see conv3x3.hip);
            // then full rows leave with 16-byte stores (an odd key swaps the halves of a chunk: swapped back in registers)
            constexpr int ROWB = BN * 2;                  // bytes per staged row
            constexpr int CPR = ROWB / 16;                // 16-byte chunks per row
            constexpr int RPI = NTHR / CPR;               // rows per store instruction of the workgroup
            constexpr int RPP = ASZ / ROWB >= 128 ? 128 : ASZ / ROWB;      // rows per pass
            static_assert(RPP % 64 == 0 && BM % RPP == 0, "epilogue passes must be whole wave row groups");
            auto skey = [](int row) { return BN == 32 ? ((row >> 1) & 7) : (row & 15); };
#pragma unroll
            for (int pass = 0; pass < BM / RPP; ++pass) {
                if ((wm * 64) / RPP == pass) {
                    // one specialised copy of the staging loop per (affine, activation): as runtime selects inside the
                    // value loop they cost ~0.5 ms per train step (raw outputs) for nothing
                    auto stage = [&](auto affine_c, auto act_c) {
                        constexpr bool AFF = decltype(affine_c)::value;
                        constexpr int ACT = decltype(act_c)::value;
#pragma unroll
                        for (int b = 0; b < NT; ++b) {
                            const int col = wn * (BN / 2) + b * 16 + fq * 4;
                            f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
                            if (AFF) {
                                sc = COEF_LDS ? *(const f32x4*)(coef + col) : sc_reg;
                                sh = COEF_LDS ? *(const f32x4*)(coef + BN + col) : sh_reg;
                            }
                            const int slot = wn * (BN / 8) + b * 4 + fq;
#pragma unroll
                            for (int a = 0; a < MT; ++a) {
                                const int row = (wm * 64) % RPP + a * 16 + fr;
                                float v[4];
#pragma unroll
                                for (int r = 0; r < 4; ++r) {
                                    v[r] = acc[a][b][r];
                                    if (AFF) v[r] = v[r] * sc[r] + sh[r];
                                    if (ACT == 1) v[r] = v[r] * __builtin_amdgcn_rcpf(1.0f + __expf(-v[r]));   // SiLU; 1 ulp, then bf16
                                    else if (ACT == 2) v[r] = fmaxf(v[r], 0.0f);
                                }
                                bf16x4 o = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
                                *(bf16x4*)(scratch + row * ROWB + ((slot ^ skey(row)) << 3)) = o;
                            }
                        }
                    };
                    using std::integral_constant;
                    if (!affine && p.act == 0) stage(integral_constant<bool, false>{}, integral_constant<int, 0>{});
                    else if (p.act == 1) stage(integral_constant<bool, true>{}, integral_constant<int, 1>{});
                    else if (p.act == 2) stage(integral_constant<bool, true>{}, integral_constant<int, 2>{});
                    else stage(integral_constant<bool, true>{}, integral_constant<int, 0>{});
                }
                __syncthreads();
                const int ch = tid % CPR, rr = tid / CPR;
                const int kc = n0 + ch * 8;
                // statistics operands of this thread's chunk: the unit's raw output row pointer and its 2 x 8 coefficients (L1-hot)
                const bf16_t* st_y = nullptr;
                int st_ldy = 0, st_act = 0;
                float st_sc[8], st_sh[8];
                if constexpr (STAT) {
                    if (st_req >= 0) {
                        const StatReq& q = p.stat[st_req];
                        const int o = kc - q.c0;
                        st_y = (const bf16_t*)q.y + o;
                        st_ldy = q.ldy;
                        st_act = q.act;
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            *(f32x4*)(st_sc + 4 * h) = *(const f32x4*)(q.scale + o + 4 * h);
                            *(f32x4*)(st_sh + 4 * h) = *(const f32x4*)(q.shift + o + 4 * h);
                        }
                    }
                }
                if (kc < p.K) {
                    auto pixel = [&](int m) -> size_t {
                        if (p.dense_out) return (size_t)m;
                        const int n = (int)fdiv((unsigned)m, p.mg_howo, p.sh_howo), rem = m - n * HoWo;
                        const int oi = (int)fdiv((unsigned)rem, p.mg_wo, p.sh_wo), oj = rem - oi * p.Wo;
                        return ((size_t)n * p.Hout + (oh_off + oi * p.oh_mul)) * p.Wout + (ow_off + oj * p.ow_mul);
                    };
                    // statistics: the raw-output vector of the NEXT row is requested before this row is processed (one load in flight)
                    i32x4 ynext = {0, 0, 0, 0};
                    if constexpr (STAT) {
                        if (st_y && m0 + pass * RPP + rr < p.M) ynext = *(const i32x4*)(st_y + pixel(m0 + pass * RPP + rr) * st_ldy);
                    }
                    for (int row = rr; row < RPP; row += RPI) {
                        const int m = m0 + pass * RPP + row;
                        if (m >= p.M) break;
                        const size_t opix = pixel(m);
                        V16 yv;
                        yv.i = ynext;
                        if constexpr (STAT) {
                            if (st_y && row + RPI < RPP && m + RPI < p.M) ynext = *(const i32x4*)(st_y + pixel(m + RPI) * st_ldy);
                        }
                        const int key = skey(row), chunk = ch ^ (key >> 1);
                        V16 v;
                        v.i = *(const i32x4*)(scratch + row * ROWB + chunk * 16);
                        if (key & 1) v.i = i32x4{v.i[2], v.i[3], v.i[0], v.i[1]};
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
                        if constexpr (STAT) {
                            if (st_y) {                    // v is the FINAL gradient of this pixel (bf16, as the apply pass will read it)
#pragma unroll
                                for (int e = 0; e < 8; ++e) {
                                    const float yy = (float)yv.h[e];
                                    float du = (float)v.h[e];
                                    if (st_act == 1) {
                                        const float u = yy * st_sc[e] + st_sh[e];
                                        const float sg = __builtin_amdgcn_rcpf(1.0f + __expf(-u));
                                        du *= sg * (1.0f + u * (1.0f - sg));
                                    }
                                    bs1[e] += du;
                                    bs2[e] += du * yy;
                                }
                            }
                        }
                    }
                }
                __syncthreads();
            }
            return (mtile + 1) * BM <= p.M;
        }
        // direct path (fp32 outputs, ragged K, unaligned rows): each lane writes its 4 consecutive channels of each of its rows
#pragma unroll
        for (int a = 0; a < MT; ++a) {
            const int m = m0 + wm * 64 + a * 16 + fr;
            if (m >= p.M) continue;
            size_t opix;
            if (p.dense_out) {
                opix = (size_t)m;
            } else {
                const int n = (int)fdiv((unsigned)m, p.mg_howo, p.sh_howo), rem = m - n * HoWo;
                const int oi = (int)fdiv((unsigned)rem, p.mg_wo, p.sh_wo), oj = rem - oi * p.Wo;
                opix = ((size_t)n * p.Hout + (oh_off + oi * p.oh_mul)) * p.Wout + (ow_off + oj * p.ow_mul);
            }
            OT* yrow = y + opix * p.ldy;
            const OT* rrow = p.res ? (const OT*)p.res + opix * p.ldr : nullptr;
#pragma unroll
            for (int b = 0; b < NT; ++b) {
                const int kc0 = n0 + wn * (BN / 2) + b * 16 + fq * 4;
                float v4[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int kc = kc0 + r;
                    float v = acc[a][b][r];
                    if (kc < p.K) {
                        if (affine) v = COEF_LDS ? v * coef[kc - n0] + coef[BN + kc - n0] : v * sc_reg[r] + sh_reg[r];
                        if (p.act == 1) v = silu_f(v);
                        else if (p.act == 2) v = fmaxf(v, 0.0f);
                    }
                    v4[r] = v;
                }
                if constexpr (sizeof(OT) == 4) {
                    // fp32 rows (detection logits, fp32 parity mode): one 16-byte store per lane where its four channels are all real and the row is
                    // 16-byte aligned — the scalar form below issued four 4-byte stores per lane, 64-byte runs per wave instruction (round 5: the
                    // detection convolutions, K = 39 in 40-float rows, ran at 0.35 of their HBM bound)
                    if (vec4_rows && kc0 + 4 <= p.K) {
                        f32x4 o = {v4[0], v4[1], v4[2], v4[3]};
                        if (rrow) o += *(const f32x4*)((const float*)rrow + kc0);
                        if (p.accumulate) o += *(const f32x4*)((const float*)yrow + kc0);
                        *(f32x4*)((float*)yrow + kc0) = o;
                        continue;
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int kc = kc0 + r;
                    if (kc >= p.K) continue;
                    float v = v4[r];
                    if (rrow) v += to_f32<OT>(rrow[kc]);
                    if (p.accumulate) v += to_f32<OT>(yrow[kc]);
                    yrow[kc] = from_f32<OT>(v);
                }
            }
        }
        return false;
    };

    // ------------------------------------------------------------------ main loop over the flattened (tile, k-block) sequence
    // stage s lives in ring slot s % NS.  Iteration `it`: wait until stage it has landed (counted vmcnt: only the stages issued
    // after it may still be in flight; vmcnt retires in issue order), barrier (also: everyone is done with stage it-1), issue
    // stage it+NS-1 into the slot stage it-1 just vacated, then the MFMAs of stage it.
    int total = ((tile_end - tile_begin + tile_step - 1) / tile_step) * nkb;
    int c_nkb = nkb;
    if (walk) {
        total = 0;
        for (int t = tile_begin; t < tile_end; t += tile_step) total += p.c_nkb[t & 3];
        c_nkb = p.c_nkb[tile_begin & 3];
    }
    // fragment addresses inside a stage: 16 rows further down is +2048 bytes with the same chunk swizzle, so the (a, b) tiles are
    // immediate offsets of four per-lane bases
    const unsigned fa[2] = {(unsigned)swz(wm * 64 + fr, fq), (unsigned)swz(wm * 64 + fr, 4 + fq)};
    const unsigned fb[2] = {(unsigned)(ASZ + swz(wn * (BN / 2) + fr, fq)), (unsigned)(ASZ + swz(wn * (BN / 2) + fr, 4 + fq))};
    int issued = 0;
    loader_set_tile(tile_begin);
    for (; issued < NS - 1 && issued < total; ++issued) loader_issue(issued % NS);
    zero_acc();
    int c_tile = tile_begin, c_kb = 0;
    bool stores_pending = false;          // the previous epilogue left its row stores in flight (younger than every issued DMA)
    for (int it = 0; it < total; ++it) {
        const int ahead = issued - it - 1;
        // epilogue stores (if any) are younger than all DMA issued so far: counting them in keeps them off the critical path
        constexpr int NST = VEC_OUT ? 128 / (NTHR / (BN / 8)) * (BM / 128) : 0;
        const bool keep_stores = stores_pending && NST + LPS * (NS - 2) <= 60;
        if (ahead >= 2 && NS >= 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPS) : "memory");
        else if (ahead >= 1 && NS >= 3) {
            if (keep_stores) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPS + NST) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPS) : "memory");
        } else {
            if (keep_stores && NS == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NST) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        stores_pending = false;
        if (issued < total) { loader_issue(issued % NS); ++issued; }
        const unsigned char* st_s = smem + (it % NS) * STAGE;
        if constexpr (BN >= 64 && sizeof(T) == 2) {
            // both k-halves' fragments are requested before the first half's MFMAs start (the wide tiles have the registers for it)
            V16 af[2][MT], bf[2][NT];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                for (int a = 0; a < MT; ++a) af[ks][a].i = *(const i32x4*)(st_s + fa[ks] + a * 2048);
#pragma unroll
                for (int b = 0; b < NT; ++b) bf[ks][b].i = *(const i32x4*)(st_s + fb[ks] + b * 2048);
                if (ks == 0) __builtin_amdgcn_sched_barrier(0);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int a = 0; a < MT; ++a)
#pragma unroll
                    for (int b = 0; b < NT; ++b) acc[a][b] = mma16<T>(bf[ks][b], af[ks][a], acc[a][b]);
        } else {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                V16 af[MT], bf[NT];
#pragma unroll
                for (int a = 0; a < MT; ++a) af[a].i = *(const i32x4*)(st_s + fa[ks] + a * 2048);
#pragma unroll
                for (int b = 0; b < NT; ++b) bf[b].i = *(const i32x4*)(st_s + fb[ks] + b * 2048);
#pragma unroll
                for (int a = 0; a < MT; ++a)
#pragma unroll
                    for (int b = 0; b < NT; ++b) acc[a][b] = mma16<T>(bf[b], af[a], acc[a][b]);
            }
        }
        if (++c_kb == c_nkb) {
            __syncthreads();                       // every wave is done reading stage `it`: its slot is scratch until the next barrier
            stores_pending = epilogue(c_tile, smem + (it % NS) * STAGE);
            zero_acc();
            c_kb = 0;
            c_tile += tile_step;
            if (walk) c_nkb = p.c_nkb[c_tile & 3];
        }
    }
    if constexpr (STAT) {
        // one slab [2][c1 - c0] per workgroup and request: the RPI threads that share a channel chunk are summed through LDS
        constexpr int CPR = BN * 2 / 16, RPI = NTHR / CPR;
        float* red = (float*)smem;                 // [NTHR][16]: every stage has been consumed
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 8; ++e) { red[tid * 16 + e] = bs1[e]; red[tid * 16 + 8 + e] = bs2[e]; }
        __syncthreads();
        for (int j = tid; j < CPR * 16; j += NTHR) {
            const int ch = j / 16, e16 = j - ch * 16;
            float v = 0.f;
            for (int r = 0; r < RPI; ++r) v += red[(r * CPR + ch) * 16 + e16];
            const int kc = ch * 8 + (e16 & 7);
            for (int r = 0; r < p.nstat; ++r)
                if (kc >= p.stat[r].c0 && kc < p.stat[r].c1) {
                    const int w = p.stat[r].c1 - p.stat[r].c0;
                    p.stat[r].slabs[((size_t)blockIdx.x * 2 + (e16 >> 3)) * w + kc - p.stat[r].c0] = v;
                }
        }
    }
    if (wg_stats) {
        // one slab per workgroup (index = its position in the XCD-aware order): 16 pixel lanes -> wave rows -> global
        float* red = (float*)smem;                 // every stage has been consumed
        __syncthreads();
        stats_to_lds(red, s1, s2);
        __syncthreads();
        // one column tile: contiguous ranges, wg_id = tile_begin / tpb; interleaved: this workgroup's column tile is wg_id % ntiles
        const size_t slab = (size_t)(wg_id / ntiles);
        const int n0 = (wg_id % ntiles) * BN;
        for (int j = tid; j < 2 * BN; j += NTHR) {
            const int which = j / BN, c = j - which * BN;
            if (n0 + c < p.K) {
                float v = 0.f;
#pragma unroll
                for (int g = 0; g < BM / 64; ++g) v += red[(g * BN + c) * 2 + which];
                p.stats[(slab * 2 + which) * p.K + n0 + c] = v;
            }
        }
    }
}

inline int igemm_interleave_mode() {      // bit 0: column tiles of one m-tile on neighbouring workgroups, bit 1: parity classes of the stride-2 dgrad too
    return hdy_opt(HDY_OPT_TILE_INTERLEAVE);
}

// Persistent grid of a tile configuration: as many workgroups as stay resident (LDS-limited), never more than tiles.
inline int igemm_grid(long long M, int ntiles, int BM, int BN, int NS) {
    const size_t smem = (size_t)NS * (BM * 128 + BN * 128) + (BN > 32 ? 2 * BN * sizeof(float) : 0);
    const int per_cu = (int)(160 * 1024 / smem) > 4 ? 4 : (int)(160 * 1024 / smem);
    long long grid = 256 * (per_cu < 1 ? 1 : per_cu);
    const long long tiles = (M + BM - 1) / BM * ntiles;
    return (int)(grid > tiles ? tiles : grid);
}

// 256-row tiles pay where the filter is re-fetched many times per output (multi-tap, wide K) AND there are many rounds of them:
// one 8-wave workgroup per CU leaves a long tail.  Measured with both tile shapes on one box: 256->256 3x3 @64x64 B=128 (4096 big
// tiles) 904 vs 934 us, 128->128 3x3 @128x128 (8192) 992 vs 1016 us in favour of the big tile; 512->512 @32x32 (2048) 884 vs 872 us,
// every 3x3 layer of yolov5s at B=64 (400-1600) 4-13 % against it; the yolov5s train step 14.93 -> 14.80 ms and the yolov5l
// inference network 64.2 -> 63.3 ms with the threshold at 4096.  The single-tap layers prefer many small workgroups in flight.
inline bool igemm_big(long long M, int bn, int ntiles, int taps) {
    const bool no_big = hdy_opt(HDY_OPT_NO_BIG_TILES) != 0;
    return bn == 128 && taps > 1 && (M + 255) / 256 * ntiles >= 4096 && !no_big;
}

template <typename T, typename OT, int BM, int BN, int NS, bool STAT = false>
int launch(const ConvArgs& a, hipStream_t st) {
    const int grid = igemm_grid(a.M, a.ntiles * (a.ncls > 1 ? 4 : 1), BM, BN, NS);
    if (a.stats) {
        // the kernel's own rule (wg_stats): one slab per workgroup position when the sums stay in registers across its tiles, else one per 128 rows
        const bool wg = a.ntiles == 1 || ((a.tile_interleave & 1) && a.ncls <= 1 && grid % a.ntiles == 0);
        long long writes = (a.M + 127) / 128;
        if (wg && a.ntiles > 1) writes = grid / a.ntiles;
        else if (wg) {
            const long long tiles = (a.M + BM - 1) / BM, tpb = (tiles + grid - 1) / grid;
            writes = (tiles + tpb - 1) / tpb;
        }
        HDY_ARG(a.stat_cap == writes, "conv: the statistics array holds %d slabs, this launch writes %lld (a kernel-selection option changed between "
                "hdy_conv_stat_slabs and the launch?)", a.stat_cap, writes);
    }
    for (int r = 0; r < a.nstat; ++r)
        HDY_ARG(a.stat[r].nslabs == grid, "conv: statistics request %d holds %d slabs, this launch writes %d", r, a.stat[r].nslabs, grid);
    const size_t smem = (size_t)NS * (BM * 128 + BN * 128) + (BN > 32 ? 2 * BN * sizeof(float) : 0);
    static PerDeviceOnce attr_once;           // first launch of this instance on any thread
    attr_once.run([&] {
        (void)hipFuncSetAttribute((const void*)conv_igemm_kernel<T, OT, BM, BN, NS, STAT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    });
    {
        char what[48];
        snprintf(what, sizeof(what), "igemm_%dx%dx%d%s%s", BM, BN, NS, a.ncls > 1 ? "_walk" : "", STAT ? "_stat" : "");
        hdy_note_dispatch(what);
    }
    hipLaunchKernelGGL((conv_igemm_kernel<T, OT, BM, BN, NS, STAT>), dim3(grid), dim3(2 * BM), smem, st, a);
    HDY_LAUNCH_CHECK("conv_igemm");
    return HDY_OK;
}

template <typename T, typename OT>
int launch_bn(const ConvArgs& a, hipStream_t st) {
    const bool big = a.ncls <= 1 && igemm_big(a.M, a.bn, a.ntiles, a.TH * a.TW);
    if constexpr (std::is_same<T, bf16_t>::value && std::is_same<OT, bf16_t>::value) {
        if (a.nstat > 0) {                         // producer-side BatchNorm-backward statistics (dgrad launches, <= 64 output channels)
            if (a.bn == 32) return launch<T, OT, 128, 32, 2, true>(a, st);
            return launch<T, OT, 128, 64, 2, true>(a, st);
        }
    }
    switch (a.bn) {
        case 32: return launch<T, OT, 128, 32, 2>(a, st);
        case 64: return launch<T, OT, 128, 64, 2>(a, st);
        default: return big ? launch<T, OT, 256, 128, 3>(a, st) : launch<T, OT, 128, 128, 2>(a, st);
    }
}

}  // namespace

int hdy_conv_bn_tile(int K) { return K <= 32 ? 32 : (K <= 64 ? 64 : 128); }

// Statistic slabs the generic kernel writes for M output pixels, K channels, `taps` filter taps: one per workgroup position when the
// sums stay in registers across the workgroup's tiles (a single column tile, or interleaved column tiles with grid % ntiles == 0: the
// kernel's wg_stats), else one per 128 output rows.
int hdy_conv_igemm_slabs(long long M, int K, int taps) {
    const int bn = hdy_conv_bn_tile(K), ntiles = cdiv(K, bn);
    const bool big = igemm_big(M, bn, ntiles, taps);
    const int BM = big ? 256 : 128;
    const int grid = igemm_grid(M, ntiles, BM, bn, big ? 3 : 2);
    if (ntiles > 1) return ((igemm_interleave_mode() & 1) && grid % ntiles == 0) ? grid / ntiles : (int)((M + 127) / 128);
    const long long tiles = (M + BM - 1) / BM;
    const long long tpb = (tiles + grid - 1) / grid;
    return (int)((tiles + tpb - 1) / tpb);
}

// Workgroups (= statistics slabs) a dgrad launch with producer-side statistics uses; 0 when the shape cannot serve them.
int hdy_conv_igemm_stat_grid(long long M, int K, int taps, int ncls) {
    const int bn = hdy_conv_bn_tile(K);
    if (bn > 64 || K % 8) return 0;               // the 128-wide instances have no registers to spare for the statistics operands
    return igemm_grid(M, ncls > 1 ? 4 : 1, 128, bn, 2);
}

// Host-side validation + dispatch shared by the C-ABI entry points (api.hip).
int hdy_conv_igemm_launch(ConvArgs a, int dtype, int out_f32, hipStream_t st) {
    const int VE = dtype == HDY_BF16 ? 8 : 4;
    HDY_ARG(a.x && a.w && a.y, "conv: null pointer");
    HDY_ARG(a.N > 0 && a.Hin > 0 && a.Win > 0 && a.Ho > 0 && a.Wo > 0 && a.K > 0 && a.C > 0, "conv: non-positive dim");
    HDY_ARG(a.C % VE == 0, "conv: C=%d must be a multiple of %d for this dtype", a.C, VE);
    HDY_ARG(a.ldx % (a.span_pixels ? 4 : VE) == 0 && (a.span_pixels || a.ldx >= a.C), "conv: ldx=%d must be >= C and a multiple of %d", a.ldx, VE);
    HDY_ARG(a.ldy >= a.K, "conv: ldy=%d < K=%d", a.ldy, a.K);
    HDY_ARG(((uintptr_t)a.x & 15) == 0 && ((uintptr_t)a.w & 15) == 0, "conv: x/w must be 16-byte aligned");
    HDY_ARG(a.TH > 0 && a.TW > 0, "conv: empty tap window");
    HDY_ARG(a.Hin < 24000 && a.Win < 24000 && a.TH < 64 && a.TW < 64 && a.dh0 > -4000 && a.dw0 > -4000, "conv: image side beyond the loader's 16-bit coordinates");
    HDY_ARG((long long)a.N * a.Hin * a.Win < (1LL << 31) && (long long)a.N * a.Ho * a.Wo < (1LL << 31), "conv: too many pixels");
    a.Kd = a.TH * a.TW * a.C;
    const int BKE = 8 * VE;
    a.bn = hdy_conv_bn_tile(a.K);
    HDY_ARG(a.Kdp == round_up(a.Kd, BKE), "conv: packed weight pitch %d != %d", a.Kdp, round_up(a.Kd, BKE));
    a.M = a.N * a.Ho * a.Wo;
    a.mtiles = cdiv(a.M, 128);
    a.ntiles = cdiv(a.K, a.bn);
    if (a.dense_out) HDY_ARG(a.oh_mul == 1 && a.ow_mul == 1 && a.oh_off == 0 && a.ow_off == 0 && a.Hout == a.Ho && a.Wout == a.Wo, "conv: dense_out geometry mismatch");
    // 1x1 / stride 1 / no padding: input pixel == output pixel, no coordinate arithmetic in the loader
    a.pointwise = (a.TH == 1 && a.TW == 1 && a.ih_mul == 1 && a.iw_mul == 1 && a.dh0 == 0 && a.dw0 == 0 && a.Hin == a.Ho && a.Win == a.Wo &&
                   !a.span_pixels && a.ncls <= 1) ? 1 : 0;
    if (a.ncls > 1) {
        HDY_ARG(a.ncls == 4 && !a.dense_out && !a.stats, "conv: class walk is the four-class stride-2 dgrad");
        for (int c = 0; c < 4; ++c) HDY_ARG(a.c_nkb[c] == round_up(a.c_TH[c] * a.c_TW[c] * a.C, BKE) / BKE, "conv: class %d k-blocks", c);
    }
    // coalesced 16-byte epilogue needs bf16 output, whole vectors and aligned rows
    const bool bf16_out = dtype == HDY_BF16 && !out_f32;
    a.vec_out = (bf16_out && a.K % 8 == 0 && a.ldy % 8 == 0 && ((uintptr_t)a.y & 15) == 0 &&
                 (!a.res || (a.ldr % 8 == 0 && ((uintptr_t)a.res & 15) == 0))) ? 1 : 0;
    if (a.nstat > 0) {
        HDY_ARG(a.nstat <= 2 && a.vec_out && a.ntiles == 1 && a.bn <= 64 && !a.stats && !a.res, "conv: producer-side statistics need the bf16 vector epilogue and at most 64 output channels");
        for (int r = 0; r < a.nstat; ++r) {
            const StatReq& q = a.stat[r];
            HDY_ARG(q.y && q.scale && q.shift && q.slabs && q.c0 >= 0 && q.c0 < q.c1 && q.c1 <= a.K && q.c0 % 8 == 0 && q.c1 % 8 == 0 &&
                    q.ldy % 8 == 0 && (((uintptr_t)q.y | (uintptr_t)q.scale | (uintptr_t)q.shift) & 15) == 0,
                    "conv: bad statistics request %d", r);
        }
    }
    // loader geometry: union tap window over the classes, reciprocals for the row / chunk decompositions
    a.uh0 = a.dh0; a.uw0 = a.dw0;
    int uh1 = a.dh0 + a.TH, uw1 = a.dw0 + a.TW;
    for (int c = 0; c < (a.ncls > 1 ? 4 : 0); ++c) {
        a.uh0 = a.c_dh[c] < a.uh0 ? a.c_dh[c] : a.uh0; a.uw0 = a.c_dw[c] < a.uw0 ? a.c_dw[c] : a.uw0;
        uh1 = a.c_dh[c] + a.c_TH[c] > uh1 ? a.c_dh[c] + a.c_TH[c] : uh1; uw1 = a.c_dw[c] + a.c_TW[c] > uw1 ? a.c_dw[c] + a.c_TW[c] : uw1;
    }
    a.UH = uh1 - a.uh0; a.UW = uw1 - a.uw0;
    HDY_ARG(a.UH * a.UW <= 31, "conv: %d x %d tap window beyond the loader's 31 tap bits", a.UH, a.UW);
    HDY_ARG(((long long)(a.UH + 1) * a.Win + a.UW) * a.ldx * (dtype == HDY_BF16 ? 2 : 4) < (1LL << 28), "conv: tap window spans too many bytes");
    a.utap = a.C % BKE == 0 ? 1 : 0;
    a.tile_interleave = igemm_interleave_mode();
    hdy_magic((unsigned)(a.Ho * a.Wo), &a.mg_howo, &a.sh_howo);
    hdy_magic((unsigned)a.Wo, &a.mg_wo, &a.sh_wo);
    hdy_magic((unsigned)a.C, &a.mg_c, &a.sh_c);
    for (int c = 0; c < 4; ++c) hdy_magic((unsigned)(a.ncls > 1 ? a.c_TW[c] : a.TW), &a.mg_tw[c], &a.sh_tw[c]);
    int rc = 0;
    if (a.ncls <= 1 && a.nstat == 0) {
        if (hdy_conv_stem_try(a, dtype, out_f32, st, &rc)) return rc;         // patch-resident 6x6/s2 stem
        if (hdy_conv3x3_c64_try(a, dtype, out_f32, st, &rc)) return rc;      // filter-resident 3x3 kernel when the shape qualifies
        if (hdy_conv3x3_c128_try(a, dtype, out_f32, st, &rc)) return rc;     // ... its 128-input-channel form (round 6)
        if (hdy_conv3x3s2_c32_try(a, dtype, out_f32, st, &rc)) return rc;    // patch-resident 3x3 / stride 2 kernel (32 input channels)
        if (hdy_conv_deep_try(a, dtype, out_f32, st, &rc)) return rc;        // deep-pipelined 256-row kernel (C % 64 == 0, K >= 128)
        // the slab count the caller sized its statistics buffer with must be the generic kernel's from here on
        HDY_ARG(!a.stats || a.span_pixels || hdy_conv_deep_slabs(a.M, a.C, a.K, a.TH * a.TW, a.pointwise, dtype) == 0,
                "conv: this shape's statistic slabs were sized for the deep-pipelined kernel, which declined the launch (alignment)");
    }
    if (a.ncls > 1 && a.nstat == 0 && hdy_conv_deep_try(a, dtype, out_f32, st, &rc)) return rc;      // stride-2 data gradient on the deep pipeline
    if (dtype == HDY_BF16) return out_f32 ? launch_bn<bf16_t, float>(a, st) : launch_bn<bf16_t, bf16_t>(a, st);
    return launch_bn<float, float>(a, st);
}

// Fused backward of a 1x1 Conv + BatchNorm(train) + SiLU unit for gfx950 (bf16): BatchNorm/SiLU backward "apply", weight gradient
// and data gradient in ONE pass over the pixels.
//
//   du = dz * silu'(y*scale + shift)                       (y = raw conv output kept by the forward pass)
//   dy = scale * (du - c1 - xhat*c2), xhat = (y-mean)*invstd      (c1, c2 from the reduce + finalize launches, bn_act.hip)
//   dx[m][c] (+)= SUM_k dy[m][k] * W[k][c]                  (data gradient)
//   dW[k][c]   = SUM_m dy[m][k] * x[m][c]                   (weight gradient, one fp32 slab per workgroup)
//
// Why: as three launches (bn_act_bwd_apply -> dy; wgrad(dy, x); dgrad(dy)) the layer moves dz, y, x in and dy out, dy in twice,
// dx out: 7 activation-sized HBM passes.  The 1x1 layers of yolov5s at 160x160 / 80x80 (C, K <= 128) are bandwidth bound in all
// three; here dy never leaves the CU: 3 passes in (dz, y, x), 1 out (dx).
//
// Per 128-pixel tile (256 threads = 4 waves, persistent workgroups):
//   1. the tile's dz / y vectors (prefetched into registers during the previous tile's MFMAs) -> dy in fp32 -> bf16 -> LDS tile D
//      [128 pixels][K], 128-byte rows per 64-channel sub-tile, 16-byte chunks XOR-swizzled by f(row) (below);
//   2. x tile [128][C] arrives by LDS-DMA (double buffered, issued one tile ahead, swizzle applied to the source address);
//   3. dgrad: D (row operand fragments, ds_read_b128) x Wd [C][K] (resident in LDS) -> dx accumulators, filter as MFMA row operand
//      so a lane owns 4 consecutive channels of one pixel;
//   4. wgrad: D^T x X with BOTH operands as transposed LDS reads (ds_read_b64_tr_b16: the reduction index is the tile row), fp32
//      accumulators live across all of the workgroup's tiles;
//   5. dx -> bf16 -> LDS staging (the D tile's storage) -> 16-byte coalesced row stores (+ accumulate).
// One LDS image of dy serves both the row-major fragment reads of (3) and the transposed reads of (4): the chunk swizzle
//   f(row) = (row & 6) ^ (bit3(row) * 5)
// was found by exhaustive search over the 8! bijections of (row >> 1) & 7 for one that is conflict-free under ds_read_b128's lane
// groups ({0-3,12-15,20-27}, ...: MI355X_MICROARCH.md, LDS) AND maps the 4 even (odd) rows of a ds_read_b64_tr_b16 half
// (rows {0-3, 8-11} / {4-7, 12-15} of a 32-pixel step) to 4 different 32-byte column pairs.
//
// Reference semantics replaced: autograd's backward of nn.Conv2d(k=1) + nn.BatchNorm2d + nn.SiLU inside Conv.forward
// (metayolo/models/layers.py:31-38), reached from train.py:472.
#include "common.h"
#include "hdyolo_internal.h"
#include "hdyolo.h"

__device__ uint4 g_hdy_zero16_f[4];   // zero page for masked 16-byte fetches

namespace {

struct FusedArgs {
    const void* dz_a; int lddz_a;     // gradient of the unit's output, channels [0, Ka)
    const void* dz_b; int lddz_b;     // ... channels [Ka, K) (merged cv1 | cv2 unit; unused when Ka == K)
    int Ka;
    const void* y; int ldy;           // raw conv output [M][K]
    const float *scale, *shift, *mean, *invstd, *c1, *c2;
    const void* x; int ldx;           // input activation [M][C]
    const void* wd; int Kdp;          // packed dgrad weights [Cpad][Kdp] (K contiguous per input channel)
    void* dx; int lddx; int accumulate;
    float* partial;                   // [gridDim.x][K][C]
    int M, K, C;
    int do_dgrad, do_wgrad;
    int nstat;                        // producer-side BatchNorm-backward statistics of the tensor(s) whose gradient dx completes
    StatReq stat[2];
};

__device__ __forceinline__ int fsw3(int row) { return (row & 6) ^ (((row >> 3) & 1) * 5); }
// byte offset of 16-byte chunk `chunk` (0..7) of row `row` inside a [rows][128 B] sub-tile
__device__ __forceinline__ int toff(int row, int chunk) { return row * 128 + ((chunk ^ fsw3(row)) << 4); }

__device__ __forceinline__ float fast_sigmoid(float u) { return __builtin_amdgcn_rcpf(1.0f + __expf(-u)); }
__device__ __forceinline__ float dsilu_f(float u) {
    const float s = fast_sigmoid(u);
    return s * (1.0f + u * (1.0f - s));
}

// The element-wise part of this kernel was its bound (PMC, 64<>64 @160x160: 6.2e7 VALU wave-instructions for 1.05e8 elements = 38 lane
// instructions per element, ~180 us of VALU issue in a 258 us launch whose bytes take 167 us).  It now runs on PAIRS of channels with packed
// fp32 instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: two lanes' worth per issue) and explicit FMAs (the file is built with
// -ffp-contract=off): 17 packed / conversion instructions + 4 quarter-rate transcendentals per pair instead of ~27 + 4 per ELEMENT.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 fma2(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
// d SiLU(u) / du = s (1 + u (1 - s)), s = sigmoid(u) by v_exp_f32 + v_rcp_f32 (the same exponent argument as __expf(-u))
__device__ __forceinline__ f32x2 dsilu2(f32x2 u) {
    const f32x2 one = {1.0f, 1.0f};
    const f32x2 e = u * f32x2{-1.4426950408889634f, -1.4426950408889634f};
    const f32x2 t = f32x2{__builtin_amdgcn_exp2f(e.x), __builtin_amdgcn_exp2f(e.y)} + one;
    const f32x2 s = {__builtin_amdgcn_rcpf(t.x), __builtin_amdgcn_rcpf(t.y)};
    return s * fma2(u, one - s, one);
}
// the two bf16 values of one dword as fp32
__device__ __forceinline__ f32x2 bf16_pair(int w) { return f32x2{__uint_as_float((unsigned)w << 16), __uint_as_float((unsigned)w & 0xFFFF0000u)}; }

// BM pixels per tile, NXB x-tile buffers (2: the next tile's x is fetched during this tile's MFMAs; 1: after them, for the instance
// whose LDS budget must leave room for a second workgroup on the CU — the other workgroup's phases cover the wait)
template <int KT, int CT, int BM, int NXB>
__global__ __launch_bounds__(256, 2) void conv1x1_bwd_kernel(const FusedArgs p) {
    constexpr int NSK = (KT + 63) / 64, NSC = (CT + 63) / 64;     // 64-channel sub-tiles per row of D / X
    constexpr int SUB = BM * 128;                                 // bytes of a [128 rows][128 B] sub-tile
    constexpr int D_SZ = NSK * SUB, X_SZ = NSC * SUB, WSUB = CT * 128;
    constexpr int NCH = KT / 8;                                   // 16-byte chunks per dy row
    constexpr bool POW2 = (NCH & (NCH - 1)) == 0;                 // round 6: KT = CT = 96 (yolov5m) — 12 chunks per row: 192 of the 256 threads own a (row, chunk)
    constexpr int RPP = POW2 ? 256 / NCH : 16, NPASS = BM / RPP;  // elementwise phase: rows per pass, passes
    constexpr int NACT = NCH * RPP;                               // threads that take part in the elementwise and the store phases
    constexpr int MT = BM / 32;                                   // dgrad: 16-pixel tiles per wave (wave = BM/2 pixels x CT/2 channels)
    constexpr int NT = CT / 32;                                   //        16-channel tiles per wave
    constexpr int MTW = KT / 32, NTW = CT / 32;                   // wgrad: 16x16 tiles per wave (wave = KT/2 x CT/2)
    constexpr int ROWB_L = POW2 ? CT * 2 : 256;                   // staging row pitch (a 192-byte row is padded: the row key XORs 8-byte slots up to 31)
    static_assert(BM * ROWB_L <= D_SZ, "dx staging must fit the D tile");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const sD = smem;
    unsigned char* const sX = smem + D_SZ;                        // NXB buffers
    unsigned char* const sW = smem + D_SZ + NXB * X_SZ;           // [NSK][CT rows][128 B]
    float* const sCo = (float*)(sW + NSK * WSUB);                 // [6][KT] BatchNorm-backward coefficients

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 15, fq = lane >> 4;
    const int tiles = (p.M + BM - 1) / BM;
    static_assert(BM % 32 == 0 && BM % RPP == 0, "tile rows");
    const int tpb = (tiles + (int)gridDim.x - 1) / (int)gridDim.x;
    const int tile_begin = blockIdx.x * tpb;
    const int tile_end = min(tile_begin + tpb, tiles);

    const bf16_t* __restrict__ y = (const bf16_t*)p.y;
    const bf16_t* __restrict__ x = (const bf16_t*)p.x;
    const unsigned char* zero = (const unsigned char*)g_hdy_zero16_f;

    // ---- resident dgrad filter: Wd[c][k] -> LDS [k sub-tile][c][128 B], swizzled like every other tile
    if (p.do_dgrad) {
        const bf16_t* wd = (const bf16_t*)p.wd;
        for (int i = tid; i < CT * NCH; i += 256) {
            const int c = i / NCH, kc = i - c * NCH;
            const i32x4 v = *(const i32x4*)(wd + (size_t)c * p.Kdp + kc * 8);
            *(i32x4*)(sW + (kc >> 3) * WSUB + toff(c, kc & 7)) = v;
        }
    }

    // ---- elementwise phase mapping: a thread owns ONE 8-channel vector (its coefficients stay in registers) and NPASS rows
    const bool ew = tid < NACT;                                   // (all 256 when NCH is a power of two)
    const int ch = POW2 ? (tid & (NCH - 1)) : (ew ? tid % NCH : 0), r0 = POW2 ? tid / NCH : (ew ? tid / NCH : 0);
    const int c0 = ch * 8;
    // the 48 per-channel coefficients of a thread's vector are re-read from LDS at the start of every tile instead of living in
    // registers across the MFMA phase (where the 128-wide instance would spill)
    for (int i = tid; i < KT; i += 256) {
        sCo[0 * KT + i] = p.scale[i]; sCo[1 * KT + i] = p.shift[i]; sCo[2 * KT + i] = p.mean[i];
        sCo[3 * KT + i] = p.invstd[i]; sCo[4 * KT + i] = p.c1[i]; sCo[5 * KT + i] = p.c2[i];
    }
    const bf16_t* const dzc = c0 < p.Ka ? (const bf16_t*)p.dz_a + c0 : (const bf16_t*)p.dz_b + (c0 - p.Ka);
    const int lddzc = c0 < p.Ka ? p.lddz_a : p.lddz_b;

    i32x4 gq[NPASS], vq[NPASS];
    auto load_tile = [&](int t) {
#pragma unroll
        for (int j = 0; j < NPASS; ++j) {
            const int m = t * BM + r0 + j * RPP;
            if (m < p.M && ew) {
                gq[j] = *(const i32x4*)(dzc + (size_t)m * lddzc);
                vq[j] = *(const i32x4*)(y + (size_t)m * p.ldy + c0);
            } else {
                gq[j] = i32x4{0, 0, 0, 0};
                vq[j] = i32x4{0, 0, 0, 0};
            }
        }
    };
    auto issue_x = [&](int t, int buf) {
        unsigned char* dst = sX + buf * X_SZ;
#pragma unroll
        for (int s = 0; s < NSC; ++s)
#pragma unroll
            for (int i = 0; i < BM / 32; ++i) {
                const int q = wave * (BM / 32) + i;                 // 8-row group of the sub-tile
                const int row = q * 8 + (lane >> 3);
                const int lc = (lane & 7) ^ fsw3(row);              // logical chunk that belongs in this lane's physical slot
                const int chan = s * 64 + lc * 8;
                const int m = t * BM + row;
                const void* src = (m < p.M && chan < CT) ? (const void*)(x + (size_t)m * p.ldx + chan) : (const void*)zero;
                __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)src,
                                                 (void __attribute__((address_space(3)))*)(dst + s * SUB + q * 1024), 16, 0, 0);
            }
    };

    // statistics served by the dx store loop (instances up to 64 channels: the 128-wide one has no registers to spare): a thread owns
    // one 16-byte channel chunk of dx for the whole launch
    constexpr bool STATS_OK = CT <= 64;
    f32x2 bs1[4], bs2[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) bs1[e] = bs2[e] = f32x2{0.f, 0.f};
    int st_req = -1;
    {
        const int kc0 = (tid % (CT * 2 / 16)) * 8;
        for (int r = 0; r < p.nstat; ++r)
            if (kc0 >= p.stat[r].c0 && kc0 < p.stat[r].c1) st_req = r;
    }

    f32x4 accw[MTW][NTW];
#pragma unroll
    for (int a = 0; a < MTW; ++a)
#pragma unroll
        for (int b = 0; b < NTW; ++b) accw[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (tile_begin < tile_end) {
        load_tile(tile_begin);
        if (p.do_wgrad) issue_x(tile_begin, 0);
    }
    __syncthreads();                                              // coefficients (and the filter) are in LDS
    const int g = fq, i16 = fr, q4 = i16 >> 2, p4 = i16 & 3;

    for (int t = tile_begin; t < tile_end; ++t) {
        const int cur = NXB == 2 ? (t - tile_begin) & 1 : 0;
        // ---- 1. dy tile -> LDS
        float sc[8], sh[8], mu[8], is[8], k1[8], k2[8];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            *(f32x4*)(sc + 4 * h) = *(const f32x4*)(sCo + 0 * KT + c0 + 4 * h); *(f32x4*)(sh + 4 * h) = *(const f32x4*)(sCo + 1 * KT + c0 + 4 * h);
            *(f32x4*)(mu + 4 * h) = *(const f32x4*)(sCo + 2 * KT + c0 + 4 * h); *(f32x4*)(is + 4 * h) = *(const f32x4*)(sCo + 3 * KT + c0 + 4 * h);
            *(f32x4*)(k1 + 4 * h) = *(const f32x4*)(sCo + 4 * KT + c0 + 4 * h); *(f32x4*)(k2 + 4 * h) = *(const f32x4*)(sCo + 5 * KT + c0 + 4 * h);
        }
        f32x2 sc2[4], sh2[4], is2[4], nm2[4], k12[4], nk22[4];       // channel pairs; xhat = y * invstd + (-mean * invstd)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            sc2[q] = f32x2{sc[2 * q], sc[2 * q + 1]}; sh2[q] = f32x2{sh[2 * q], sh[2 * q + 1]};
            is2[q] = f32x2{is[2 * q], is[2 * q + 1]}; nm2[q] = f32x2{-mu[2 * q] * is[2 * q], -mu[2 * q + 1] * is[2 * q + 1]};
            k12[q] = f32x2{k1[2 * q], k1[2 * q + 1]}; nk22[q] = f32x2{-k2[2 * q], -k2[2 * q + 1]};
        }
#pragma unroll
        for (int j = 0; j < NPASS; ++j) {
            const int row = r0 + j * RPP;
            V16 o;
            const i32x4 gw = gq[j], yw = vq[j];
            const bool live = t * BM + row < p.M;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x2 v = bf16_pair(yw[q]), g = bf16_pair(gw[q]);
                const f32x2 du = g * dsilu2(fma2(v, sc2[q], sh2[q]));
                const f32x2 xh = fma2(v, is2[q], nm2[q]);
                const f32x2 d = sc2[q] * fma2(xh, nk22[q], du - k12[q]);        // scale * (du - c1 - xhat * c2)
                o.h[2 * q] = (bf16_t)(live ? d.x : 0.0f);
                o.h[2 * q + 1] = (bf16_t)(live ? d.y : 0.0f);
            }
            if (ew) *(i32x4*)(sD + (ch >> 3) * SUB + toff(row, ch & 7)) = o.i;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // this wave's share of the x tile has landed
        __syncthreads();
        // ---- 2. next tile's operands: dz / y into registers, x by LDS-DMA into the other buffer
        if (t + 1 < tile_end) {
            load_tile(t + 1);
            if (NXB == 2 && p.do_wgrad) issue_x(t + 1, cur ^ 1);
        }
        // ---- 3. dgrad MFMAs
        f32x4 acc[MT][NT];
        if (p.do_dgrad) {
#pragma unroll
            for (int a = 0; a < MT; ++a)
#pragma unroll
                for (int b = 0; b < NT; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KT / 32; ++ks) {
                const int sub = ks >> 1, cbase = (ks & 1) * 4;
                V16 pf[MT], wf[NT];
#pragma unroll
                for (int a = 0; a < MT; ++a) pf[a].i = *(const i32x4*)(sD + sub * SUB + toff(wm * (BM / 2) + a * 16 + fr, cbase + fq));
#pragma unroll
                for (int b = 0; b < NT; ++b) wf[b].i = *(const i32x4*)(sW + sub * WSUB + toff(wn * (CT / 2) + b * 16 + fr, cbase + fq));
#pragma unroll
                for (int a = 0; a < MT; ++a)
#pragma unroll
                    for (int b = 0; b < NT; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[b].h, pf[a].h, acc[a][b], 0, 0, 0);
            }
        }
        // ---- 4. wgrad MFMAs: reduction over the tile's 128 pixels, 32 per step
        if (p.do_wgrad) {
            const unsigned char* xs = sX + cur * X_SZ;
#pragma unroll
            for (int ks = 0; ks < BM / 32; ++ks) {
                const int ra = ks * 32 + 8 * g + q4, rb = ra + 4;
                const int oa = ra * 128 + (p4 & 1) * 8, ob = rb * 128 + (p4 & 1) * 8;
                const int fa = fsw3(ra), fb = fsw3(rb);
                V16 af[MTW], bf[NTW];
#pragma unroll
                for (int a = 0; a < MTW; ++a) {
                    const int cm = wm * (KT / 2) + a * 16;
                    const unsigned char* base = sD + (cm >> 6) * SUB;
                    const int lc = ((cm & 63) >> 3) + (p4 >> 1);
                    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(base + oa + ((lc ^ fa) << 4)));
                    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(base + ob + ((lc ^ fb) << 4)));
                    af[a].h = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                }
#pragma unroll
                for (int b = 0; b < NTW; ++b) {
                    const int cn = wn * (CT / 2) + b * 16;
                    const unsigned char* base = xs + (cn >> 6) * SUB;
                    const int lc = ((cn & 63) >> 3) + (p4 >> 1);
                    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(base + oa + ((lc ^ fa) << 4)));
                    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(base + ob + ((lc ^ fb) << 4)));
                    bf[b].h = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                }
#pragma unroll
                for (int a = 0; a < MTW; ++a)
#pragma unroll
                    for (int b = 0; b < NTW; ++b) accw[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[a].h, bf[b].h, accw[a][b], 0, 0, 0);
            }
        }
        __syncthreads();                                            // every wave is done reading D and X[cur]
        if (NXB == 1 && p.do_wgrad && t + 1 < tile_end) issue_x(t + 1, 0);
        // ---- 5. dx tile: accumulators -> bf16 -> staging (D's storage, 8-byte slots XORed with a row key) -> 16-byte row stores
        // A ds_write_b64 is served in groups of 16 CONSECUTIVE lanes over 32 banks (MI355X_MICROARCH.md, LDS table): here the 16 pixel rows fr of one
        // 8-byte slot.  Round 2's key (row & 14) put rows r and r ^ 1 on the same slot: every staging write 2-way conflicted — the 13-21 % of LDS-active
        // cycles that profiles/r03_layers_pmc_table.txt shows for this kernel (8 such writes per wave and tile among ~230 LDS cycles).  The key now takes
        // all 16 rows to 16 different slots: row & 15 for rows of 128 bytes or more (the row's own bank offset is 0 mod 32), (row >> 1) & 7 for the 64-byte
        // rows of the 32-channel instance (two rows per 128 bytes: bit 0 of the row already selects the bank half).  An odd key swaps the two 8-byte
        // halves of a 16-byte chunk, which the reader undoes in registers.
        if (p.do_dgrad) {
            constexpr int ROWB = ROWB_L, CPR = CT * 2 / 16, RPI = POW2 ? 256 / CPR : 16;      // (CT = 96: 12 chunks x 16 rows = 192 storing threads)
#ifdef HDY_F1X1_OLDKEY                                                  // A/B build (scripts/build_variant.sh ... -DHDY_F1X1_OLDKEY): round 2's key
            auto skey = [](int row) { return row & ((2 * CPR - 1) & 14); };
#else
            auto skey = [](int row) { return CT >= 64 ? (row & 15) : ((row >> 1) & 7); };
#endif
#pragma unroll
            for (int b = 0; b < NT; ++b) {
                const int slot = wn * (CT / 8) + b * 4 + fq;
#pragma unroll
                for (int a = 0; a < MT; ++a) {
                    const int row = wm * (BM / 2) + a * 16 + fr;
                    const bf16x4 o = {(bf16_t)acc[a][b][0], (bf16_t)acc[a][b][1], (bf16_t)acc[a][b][2], (bf16_t)acc[a][b][3]};
                    *(bf16x4*)(sD + row * ROWB + ((slot ^ skey(row)) << 3)) = o;
                }
            }
            __syncthreads();
            const bool stq = POW2 || tid < CPR * RPI;
            const int cc = stq ? tid % CPR : 0, rr = stq ? tid / CPR : 0;
            bf16_t* const dx = (bf16_t*)p.dx;
            const bf16_t* st_y = nullptr;
            int st_ldy = 0, st_act = 0;
            float st_sc[8], st_sh[8];
            if constexpr (STATS_OK) {
                if (st_req >= 0) {
                    const StatReq& q = p.stat[st_req];
                    const int o = cc * 8 - q.c0;
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
            // statistics: the raw outputs of ALL of this thread's rows of the tile are requested up front (one row ahead — the first version —
            // made the store loop a chain of NIT global-load latencies per tile)
            constexpr int NIT = BM / RPI;
            i32x4 yrows[NIT];
            if constexpr (STATS_OK) {
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    const int m = t * BM + rr + it * RPI;
                    yrows[it] = (st_y && m < p.M) ? *(const i32x4*)(st_y + (size_t)m * st_ldy) : i32x4{0, 0, 0, 0};
                }
            }
            // accumulate: the gradient already in dx, all rows in flight too (not in the 128-wide instance: it sits at 255 registers and
            // got 30 % slower with the extra live values; it loads row by row as before)
            constexpr bool PRE = CT <= 64;
            i32x4 arows[PRE ? NIT : 1];
            if (PRE && p.accumulate) {
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    const int m = t * BM + rr + it * RPI;
                    arows[it] = (m < p.M && stq) ? *(const i32x4*)(dx + (size_t)m * p.lddx + cc * 8) : i32x4{0, 0, 0, 0};
                }
            }
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int row = rr + it * RPI;
                const int m = t * BM + row;
                if (m >= p.M || !stq) break;
                V16 v, yv;
                yv.i = STATS_OK ? yrows[it] : i32x4{0, 0, 0, 0};
                {
                    const int key = skey(row);
                    const i32x4 w = *(const i32x4*)(sD + row * ROWB + ((cc ^ (key >> 1)) << 4));
                    v.i = (key & 1) ? i32x4{w[2], w[3], w[0], w[1]} : w;
                }
                bf16_t* dst = dx + (size_t)m * p.lddx + cc * 8;
                if (p.accumulate) {
                    V16 q;
                    q.i = PRE ? arows[PRE ? it : 0] : *(const i32x4*)dst;
#pragma unroll
                    for (int e = 0; e < 8; ++e) v.h[e] = (bf16_t)((float)v.h[e] + (float)q.h[e]);
                }
                *(i32x4*)dst = v.i;
                if constexpr (STATS_OK) {
                    if (st_y) {                                     // v = the final gradient of this pixel, as the consumer will read it
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const f32x2 yy = bf16_pair(yv.i[q]);
                            f32x2 du = bf16_pair(v.i[q]);
                            if (st_act == 1) du = du * dsilu2(fma2(yy, f32x2{st_sc[2 * q], st_sc[2 * q + 1]}, f32x2{st_sh[2 * q], st_sh[2 * q + 1]}));
                            bs1[q] = bs1[q] + du;
                            bs2[q] = fma2(du, yy, bs2[q]);
                        }
                    }
                }
            }
            __syncthreads();                                        // staging is free: the next tile's dy may be written
        }
    }

    if (STATS_OK && p.nstat > 0) {
        constexpr int CPR = CT * 2 / 16, RPI = 256 / CPR;
        float* red = (float*)sD;                                     // [256][16]; D / staging are free
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 8; ++e) { red[tid * 16 + e] = bs1[e >> 1][e & 1]; red[tid * 16 + 8 + e] = bs2[e >> 1][e & 1]; }
        __syncthreads();
        for (int j = tid; j < CPR * 16; j += 256) {
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
    if (p.do_wgrad) {
        float* out = p.partial + (size_t)blockIdx.x * p.K * p.C;
#pragma unroll
        for (int a = 0; a < MTW; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k = wm * (KT / 2) + a * 16 + g * 4 + r;
#pragma unroll
                for (int b = 0; b < NTW; ++b) out[(size_t)k * p.C + wn * (CT / 2) + b * 16 + i16] = accw[a][b][r];
            }
    }
}

template <int KT, int CT, int BM, int NXB>
constexpr size_t fused_smem() {
    return (size_t)((KT + 63) / 64) * BM * 128 + NXB * (size_t)((CT + 63) / 64) * BM * 128 + (size_t)((KT + 63) / 64) * CT * 128 + 6 * KT * sizeof(float);
}

template <int KT, int CT, int BM, int NXB>
int fused_launch(const FusedArgs& a, int grid, hipStream_t st) {
    constexpr size_t smem = fused_smem<KT, CT, BM, NXB>();
    static_assert(2 * smem <= 160 * 1024, "two workgroups per CU");
    static PerDeviceOnce attr_once;           // first launch of this instance on any thread
    attr_once.run([&] {
        (void)hipFuncSetAttribute((const void*)conv1x1_bwd_kernel<KT, CT, BM, NXB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    });
    hdy_note_dispatch(KT == 32 ? "conv1x1_bwd_32" : (KT == 64 ? "conv1x1_bwd_64" : (KT == 96 ? "conv1x1_bwd_96" : "conv1x1_bwd_128")));
    hipLaunchKernelGGL((conv1x1_bwd_kernel<KT, CT, BM, NXB>), dim3(grid), dim3(256), smem, st, a);
    HDY_LAUNCH_CHECK("conv1x1_bwd");
    return HDY_OK;
}

// 128-wide: 64-pixel tiles, so that two workgroups fit a CU's LDS.  (64-pixel tiles / four workgroups per CU for the 32- and 64-wide instances were
// measured in round 3 and are gone: 64<>64 @160x160 265 -> 288 us, 32<>32 207 -> 228 us)
inline int tile_rows(int K) { return K >= 96 ? 64 : 128; }

}  // namespace

int hdy_wgrad_reduce_launch(const float* partial, int splits, size_t slab_stride, int K, int Q, int mode, int C, int R, int S, float* grad,
                            int accumulate, hipStream_t st);

extern "C" {

// 1 when hdy_conv1x1_bwd_fused has a kernel for this (C, K, dtype)
int hdy_conv1x1_bwd_fused_ok(int C, int K, int dtype) {
    return dtype == HDY_BF16 && C == K && (K == 32 || K == 64 || K == 128 || (K == 96 && !hdy_opt(HDY_OPT_NO_F1X1_96))) ? 1 : 0;
}

// workgroups (= fp32 weight-gradient slabs) the fused kernel uses for M pixels
int hdy_conv1x1_bwd_fused_grid(long long M, int K) {
    const int bm = tile_rows(K);
    const long long tiles = (M + bm - 1) / bm;
    const long long cap = 512;                              // two resident workgroups per CU (768 for the 32-wide instance measured slower: 179 vs 160 us)
    return (int)(tiles < cap ? tiles : cap);
}

// slabs (= workgroups) a statistics-serving fused launch writes; 0: this instance does not serve statistics
int hdy_conv1x1_bwd_fused_stat_slabs(long long M, int C, int K, int dtype) {
    return hdy_conv1x1_bwd_fused_ok(C, K, dtype) && C <= 64 ? hdy_conv1x1_bwd_fused_grid(M, K) : 0;
}

size_t hdy_conv1x1_bwd_fused_workspace_bytes(long long M, int C, int K) {
    return (size_t)hdy_conv1x1_bwd_fused_grid(M, K) * K * C * sizeof(float);
}

static int fused_impl(const void* dz_a, int lddz_a, const void* dz_b, int lddz_b, int Ka, const void* y, int ldy, const float* scale,
                      const float* shift, const float* mean, const float* invstd, const float* c1, const float* c2, const void* x, int ldx,
                      const void* w_packed_dgrad, void* dx, int lddx, int accumulate_dx, float* grad_a, int K_a, float* grad_b, int K_b,
                      int accumulate_w, long long M, int C, int K, void* workspace, size_t ws_bytes, int dtype, const hdy_stat_req* stats, int nstat,
                      void* stream);

int hdy_conv1x1_bwd_fused(const void* dz_a, int lddz_a, const void* dz_b, int lddz_b, int Ka, const void* y, int ldy, const float* scale,
                          const float* shift, const float* mean, const float* invstd, const float* c1, const float* c2, const void* x, int ldx,
                          const void* w_packed_dgrad, void* dx, int lddx, int accumulate_dx, float* grad_a, int K_a, float* grad_b, int K_b,
                          int accumulate_w, long long M, int C, int K, void* workspace, size_t ws_bytes, int dtype, void* stream) {
    return fused_impl(dz_a, lddz_a, dz_b, lddz_b, Ka, y, ldy, scale, shift, mean, invstd, c1, c2, x, ldx, w_packed_dgrad, dx, lddx, accumulate_dx, grad_a, K_a,
                      grad_b, K_b, accumulate_w, M, C, K, workspace, ws_bytes, dtype, nullptr, 0, stream);
}

int hdy_conv1x1_bwd_fused_stats(const void* dz_a, int lddz_a, const void* dz_b, int lddz_b, int Ka, const void* y, int ldy, const float* scale,
                                const float* shift, const float* mean, const float* invstd, const float* c1, const float* c2, const void* x, int ldx,
                                const void* w_packed_dgrad, void* dx, int lddx, int accumulate_dx, float* grad_a, int K_a, float* grad_b, int K_b,
                                int accumulate_w, long long M, int C, int K, void* workspace, size_t ws_bytes, int dtype, const hdy_stat_req* stats,
                                int nstat, void* stream) {
    return fused_impl(dz_a, lddz_a, dz_b, lddz_b, Ka, y, ldy, scale, shift, mean, invstd, c1, c2, x, ldx, w_packed_dgrad, dx, lddx, accumulate_dx, grad_a, K_a,
                      grad_b, K_b, accumulate_w, M, C, K, workspace, ws_bytes, dtype, stats, nstat, stream);
}

static int fused_impl(const void* dz_a, int lddz_a, const void* dz_b, int lddz_b, int Ka, const void* y, int ldy, const float* scale,
                          const float* shift, const float* mean, const float* invstd, const float* c1, const float* c2, const void* x, int ldx,
                          const void* w_packed_dgrad, void* dx, int lddx, int accumulate_dx, float* grad_a, int K_a, float* grad_b, int K_b,
                      int accumulate_w, long long M, int C, int K, void* workspace, size_t ws_bytes, int dtype, const hdy_stat_req* stats, int nstat,
                      void* stream) {
    HDY_ARG(hdy_conv1x1_bwd_fused_ok(C, K, dtype), "conv1x1_bwd_fused: no kernel for C=%d K=%d dtype=%d", C, K, dtype);
    HDY_ARG(dz_a && y && scale && shift && mean && invstd && c1 && c2 && x && M > 0 && M < (1LL << 31), "conv1x1_bwd_fused: bad args");
    HDY_ARG(Ka == K || (Ka > 0 && Ka < K && Ka % 8 == 0 && dz_b), "conv1x1_bwd_fused: bad gradient split Ka=%d", Ka);
    HDY_ARG(lddz_a % 8 == 0 && (Ka == K || lddz_b % 8 == 0) && ldy % 8 == 0 && ldx % 8 == 0 && ldy >= K && ldx >= C, "conv1x1_bwd_fused: pitches must be multiples of 8 elements");
    HDY_ARG((((uintptr_t)dz_a | (uintptr_t)dz_b | (uintptr_t)y | (uintptr_t)x | (uintptr_t)dx | (uintptr_t)w_packed_dgrad) & 15) == 0, "conv1x1_bwd_fused: operands must be 16-byte aligned");
    HDY_ARG(!dx || (w_packed_dgrad && lddx % 8 == 0 && lddx >= C), "conv1x1_bwd_fused: dx needs packed dgrad weights and an aligned pitch");
    HDY_ARG((grad_a == nullptr && K_a == 0 && grad_b == nullptr) || (grad_a && K_a > 0 && K_a + K_b <= K && (K_b == 0) == (grad_b == nullptr)), "conv1x1_bwd_fused: bad weight gradient split");
    HDY_ARG(!grad_a || (workspace && ws_bytes >= hdy_conv1x1_bwd_fused_workspace_bytes(M, C, K)), "conv1x1_bwd_fused: workspace too small");
    HDY_ARG(dx || grad_a, "conv1x1_bwd_fused: nothing to compute");
    FusedArgs a = {};
    a.dz_a = dz_a; a.lddz_a = lddz_a; a.dz_b = dz_b; a.lddz_b = lddz_b; a.Ka = Ka;
    a.y = y; a.ldy = ldy; a.scale = scale; a.shift = shift; a.mean = mean; a.invstd = invstd; a.c1 = c1; a.c2 = c2;
    a.x = x; a.ldx = ldx; a.wd = w_packed_dgrad; a.Kdp = round_up(K, 64);
    a.dx = dx; a.lddx = lddx; a.accumulate = accumulate_dx;
    a.partial = (float*)workspace;
    a.M = (int)M; a.K = K; a.C = C;
    a.do_dgrad = dx != nullptr; a.do_wgrad = grad_a != nullptr;
    HDY_ARG(nstat >= 0 && nstat <= 2 && (nstat == 0 || (stats && dx)), "conv1x1_bwd_fused: statistics requests need dx");
    a.nstat = nstat;
    const int grid = hdy_conv1x1_bwd_fused_grid(M, K);      // = slabs every statistics request receives
    for (int r = 0; r < nstat; ++r) {
        const hdy_stat_req& q = stats[r];
        HDY_ARG(q.nslabs == grid, "conv1x1_bwd_fused: statistics request %d holds %d slabs, this launch writes %d", r, q.nslabs, grid);
        HDY_ARG(q.y && q.scale && q.shift && q.slabs && q.c0 >= 0 && q.c0 < q.c1 && q.c1 <= C && q.c0 % 8 == 0 && q.c1 % 8 == 0 &&
                q.ldy % 8 == 0 && (((uintptr_t)q.y | (uintptr_t)q.scale | (uintptr_t)q.shift) & 15) == 0 && C <= 64,
                "conv1x1_bwd_fused: bad statistics request %d (served up to 64 channels)", r);
        a.stat[r] = StatReq{q.y, q.ldy, q.scale, q.shift, q.slabs, q.c0, q.c1, q.act, q.nslabs};
    }
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if (K == 32) rc = fused_launch<32, 32, 128, 2>(a, grid, st);
    else if (K == 64) rc = fused_launch<64, 64, 128, 2>(a, grid, st);
    else if (K == 96) rc = fused_launch<96, 96, 64, 2>(a, grid, st);          // round 6: yolov5m's 96-wide 1x1 units (two 64-channel sub-tiles per row, the second half full)
    else rc = fused_launch<128, 128, 64, 1>(a, grid, st);
    if (rc || !grad_a) return rc;
    rc = hdy_wgrad_reduce_launch(a.partial, grid, (size_t)K * C, K_a, C, 0, C, 1, 1, grad_a, accumulate_w, st);
    if (rc) return rc;
    if (K_b) rc = hdy_wgrad_reduce_launch(a.partial + (size_t)K_a * C, grid, (size_t)K * C, K_b, C, 0, C, 1, 1, grad_b, accumulate_w, st);
    return rc;
}

}  // extern "C"

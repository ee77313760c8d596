// Weight gradient of the tap convolution for gfx950, plus weight packing / gradient unpacking.
//
//   dWp[k][t][c] = SUM_{n,i,j} dy[n,i,j,k] * x[n, i*ih_mul + dh0 + t/TW, j*iw_mul + dw0 + t%TW, c]
//
// GEMM view: rows = K (dy channels), cols = Q = T*C, reduction = P = N*Ho*Wo pixels (up to 1.6 M for a
// 64-tile batch at 160x160: tall-skinny, so the pixel axis is split over workgroups).  Both operands are
// NHWC, i.e. the reduction index is the *row* of each LDS tile, which is exactly the case
// ds_read_b64_tr_b16 exists for: tiles are stored [pixel][channel] as they arrive (coalesced 16-byte
// loads), and the MFMA fragments (8 consecutive reduction elements per lane) come out of two transposed
// LDS reads per operand — no transpose pass, no scalar LDS traffic.  32-byte column blocks are
// XOR-swizzled with row bits 1 and 3 so the 8 rows touched by a 32-lane half hit 8 distinct bank octets.
// fp32 (parity mode) uses plain ds_read_b32 + v_mfma_f32_16x16x4_f32 on the same image.
//
// Each workgroup writes its fp32 tile to partial[split][K][Q]; hdy_wgrad_reduce sums the slabs in a fixed
// order and scatters into the framework-layout gradient [K][C][R][S] (bitwise reproducible, no atomics).
//
// Replaces autograd's conv backward-weight for nn.Conv2d in metayolo/models/layers.py:31 and
// yolo_head.py:112 (reached from train.py:472).
#include <stdlib.h>

#include "common.h"
#include "hdyolo_internal.h"
#include "hdyolo.h"

__device__ uint4 g_hdy_zero16_w[4];   // zero page for masked 16-byte fetches

namespace {

__device__ __forceinline__ int fsw(int row) { return ((row >> 1) & 1) | (((row >> 3) & 1) << 1); }

template <typename T> struct WT;
template <> struct WT<bf16_t> { static constexpr int VE = 8, TK = 64, WTL = 2; };   // channels per 128-byte sub-tile row, MFMA tiles per 32 ch
template <> struct WT<float> { static constexpr int VE = 4, TK = 32, WTL = 1; };

constexpr int PB = 64;   // pixels per LDS stage

// Output tile of a workgroup = (SD x TK dy-channels) x (SX x TK tap-channels); SD, SX in {1, 2}.  Each operand tile is SD
// (SX) sub-tiles of [64 pixels][128 B]; with SD = SX = 2 every staged byte feeds twice as many MFMAs and each dy / x element
// is fetched by half as many workgroups as with 64x64 tiles (the wgrad of the 128+-channel layers was L2-fetch bound).
template <typename T, int SD, int SX, bool XCD_WGRAD = true>
__global__ __launch_bounds__(256) void wgrad_kernel(const WgradArgs p) {
    constexpr int VE = WT<T>::VE, TK = WT<T>::TK, WTL = WT<T>::WTL;
    constexpr int MTW = WTL * SD, NTW = WTL * SX;            // 16x16 MFMA tiles per wave along k / q
    constexpr int SUB = PB * 128;                              // bytes of one sub-tile
    constexpr int STAGE = (SD + SX) * SUB;
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles = p.ktiles * p.qtiles;
    // the tiles of one pixel split read the same dy / x rows: consecutive LOGICAL ids, i.e. one XCD and its L2 (blockIdx.x itself goes round
    // the eight XCDs, so every XCD used to fetch every split's operands for itself)
    const int bid = XCD_WGRAD ? xcd_remap(blockIdx.x, gridDim.x) : (int)blockIdx.x;
    const int split = bid / tiles;
    const int tile = bid - split * tiles;
    const int kt = tile / p.qtiles, qt = tile - kt * p.qtiles;
    const int k0 = kt * TK * SD, q0 = qt * TK * SX;
    const int pbeg = split * p.pix_per_split;
    const int pend = min(pbeg + p.pix_per_split, p.P);

    const T* __restrict__ x = (const T*)p.x;
    const T* __restrict__ dy = (const T*)p.dy;
    const unsigned char* zero = (const unsigned char*)g_hdy_zero16_w;

    const int r0 = tid >> 3;
    // LDS-DMA writes linearly (wave base + lane*16): this thread fills physical slot (row, tid&7) of every sub-tile, so it
    // fetches the LOGICAL 16-byte chunk that the 32-byte-block swizzle places there (row bits 1,3 are the same for its 2 rows)
    const int c8 = ((((tid & 7) >> 1) ^ fsw(r0)) << 1) | (tid & 1);
    int kch[SD], xth[SX], xtw[SX], xc[SX];
    bool k_ok[SD], q_ok[SX];
#pragma unroll
    for (int u = 0; u < SD; ++u) {
        kch[u] = k0 + u * TK + c8 * VE;
        k_ok[u] = kch[u] < p.K;
    }
#pragma unroll
    for (int u = 0; u < SX; ++u) {
        const int q = q0 + u * TK + c8 * VE;
        q_ok[u] = q < p.Q;
        xth[u] = xtw[u] = xc[u] = 0;
        if (q_ok[u]) {
            const int tap = q / p.C;
            xc[u] = q - tap * p.C;
            xth[u] = tap / p.TW;
            xtw[u] = tap - xth[u] * p.TW;
        }
    }
    const int HoWo = p.Ho * p.Wo;
    int pn[2], pi[2], pj[2], pp[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        pp[i] = pbeg + r0 + 32 * i;
        const int pc = min(pp[i], p.P - 1);
        pn[i] = pc / HoWo;
        const int rem = pc - pn[i] * HoWo;
        pi[i] = rem / p.Wo;
        pj[i] = rem - pi[i] * p.Wo;
    }

    auto stage = [&](int buf) {
        unsigned char* sD = smem + buf * STAGE;
        unsigned char* sX = sD + SD * SUB;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const bool live = pp[i] < pend;
            const int woff = (wave * 64 + 256 * i) * 16;
#pragma unroll
            for (int u = 0; u < SD; ++u) {
                const void* src = zero;
                if (live && k_ok[u]) src = dy + (size_t)pp[i] * p.lddy + kch[u];
                __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)src,
                                                 (void __attribute__((address_space(3)))*)(sD + u * SUB + woff), 16, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < SX; ++u) {
                const void* src = zero;
                if (live && q_ok[u]) {
                    const int hi = pi[i] * p.ih_mul + p.dh0 + xth[u], wi = pj[i] * p.iw_mul + p.dw0 + xtw[u];
                    if ((unsigned)hi < (unsigned)p.Hin && (unsigned)wi < (unsigned)p.Win)
                        src = x + ((size_t)(pn[i] * p.Hin + hi) * p.Win + wi) * p.ldx + xc[u];
                }
                __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)src,
                                                 (void __attribute__((address_space(3)))*)(sX + u * SUB + woff), 16, 0, 0);
            }
            // advance this row by one stage
            pp[i] += PB;
            pj[i] += PB;
            while (pj[i] >= p.Wo) {
                pj[i] -= p.Wo;
                if (++pi[i] == p.Ho) { pi[i] = 0; ++pn[i]; }
            }
        }
    };

    f32x4 acc[MTW][NTW];
#pragma unroll
    for (int a = 0; a < MTW; ++a)
#pragma unroll
        for (int b = 0; b < NTW; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nst = (pend - pbeg + PB - 1) / PB;
    if (nst > 0) stage(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int g = lane >> 4, i16 = lane & 15;
    // channel offset of this wave's tile a (b) inside the workgroup tile -> (sub-tile, column)
    constexpr int WCH_M = TK * SD / 2, WCH_N = TK * SX / 2;
    for (int s = 0; s < nst; ++s) {
        const int cur = s & 1;
        if (s + 1 < nst) stage(cur ^ 1);
        const unsigned char* d_s = smem + cur * STAGE;
        const unsigned char* x_s = d_s + SD * SUB;
        if constexpr (sizeof(T) == 2) {
            const int q4 = i16 >> 2, p4 = i16 & 3;
#pragma unroll
            for (int ks = 0; ks < PB / 32; ++ks) {
                V16 af[MTW], bf[NTW];
                const int ra = ks * 32 + 8 * g + q4, rb = ra + 4;
#pragma unroll
                for (int a = 0; a < MTW; ++a) {
                    const int cm = wm * WCH_M + a * 16;
                    const unsigned char* base = d_s + (cm / TK) * SUB;
                    const int cb = (cm % TK) >> 4;
                    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                        (bf16x4 __attribute__((address_space(3)))*)(base + ra * 128 + ((cb ^ fsw(ra)) << 5) + p4 * 8));
                    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                        (bf16x4 __attribute__((address_space(3)))*)(base + rb * 128 + ((cb ^ fsw(rb)) << 5) + p4 * 8));
                    af[a].h = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                }
#pragma unroll
                for (int b = 0; b < NTW; ++b) {
                    const int cn = wn * WCH_N + b * 16;
                    const unsigned char* base = x_s + (cn / TK) * SUB;
                    const int cb = (cn % TK) >> 4;
                    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                        (bf16x4 __attribute__((address_space(3)))*)(base + ra * 128 + ((cb ^ fsw(ra)) << 5) + p4 * 8));
                    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                        (bf16x4 __attribute__((address_space(3)))*)(base + rb * 128 + ((cb ^ fsw(rb)) << 5) + p4 * 8));
                    bf[b].h = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                }
#pragma unroll
                for (int a = 0; a < MTW; ++a)
#pragma unroll
                    for (int b = 0; b < NTW; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[a].h, bf[b].h, acc[a][b], 0, 0, 0);
            }
        } else {
            // fp32: A[k-channel = lane&15][pixel = lane>>4] straight from the [pixel][channel] image
#pragma unroll 2
            for (int ks = 0; ks < PB / 4; ++ks) {
                const int row = ks * 4 + g;
                float av[MTW], bv[NTW];
#pragma unroll
                for (int a = 0; a < MTW; ++a) {
                    const int cm = wm * WCH_M + a * 16 + i16;
                    av[a] = *(const float*)(d_s + (cm / TK) * SUB + row * 128 + ((((cm % TK) >> 3) ^ fsw(row)) << 5) + (cm & 7) * 4);
                }
#pragma unroll
                for (int b = 0; b < NTW; ++b) {
                    const int cn = wn * WCH_N + b * 16 + i16;
                    bv[b] = *(const float*)(x_s + (cn / TK) * SUB + row * 128 + ((((cn % TK) >> 3) ^ fsw(row)) << 5) + (cn & 7) * 4);
                }
#pragma unroll
                for (int a = 0; a < MTW; ++a)
#pragma unroll
                    for (int b = 0; b < NTW; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[a], bv[b], acc[a][b], 0, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    float* out = p.partial + (size_t)split * p.K * p.Q;
#pragma unroll
    for (int a = 0; a < MTW; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k = k0 + wm * WCH_M + a * 16 + g * 4 + r;
            if (k >= p.K) continue;
#pragma unroll
            for (int b = 0; b < NTW; ++b) {
                const int qq = q0 + wn * WCH_N + b * 16 + i16;
                if (qq < p.Q) out[(size_t)k * p.Q + qq] = acc[a][b][r];
            }
        }
}

// ------------------------------------------------------------------------------------------------------------------------
// Weight gradient of the 6x6 / stride 2 / pad 2 RGB stem as a patch-resident kernel (bf16).
// Through wgrad_kernel the stem gathers, per output pixel, six 48-byte window rows of the 4-channel padded image: 288 B of x per
// 64 B of dy, 2.3 GB of L2->LDS traffic per 64-tile batch -> 484 us against an HBM bound of ~130 us — and it is the LAST weight
// gradient of the backward pass, with nothing left on the main stream to hide behind.  Here a workgroup stages, per 16 x 32 output
// tile, the (2*16+4) x (2*32+4) input patch (19.6 KB, read once) and the dy tile (512 pixels x K), and both MFMA operands are
// transposed LDS reads (ds_read_b64_tr_b16, reduction = the 32 pixels of one tile row):
//   * A (dy): rows = pixels of the tile row, pitch K*2 bytes;
//   * B (x):  rows = the same pixels' windows, i.e. patch row 2*ty + r at a row pitch of 16 bytes (two input pixels per output pixel):
//     overlapping rows, which a transposed read does not mind; columns = (s, c) = 24 contiguous bf16, padded to 32 (the 8 extra
//     columns read the neighbouring pixels and are dropped at the end).
// Workgroups are persistent over ~17 tiles and write ONE fp32 slab [K][144] each, summed by wgrad_reduce_kernel as before.
namespace stemw {

constexpr int TOH = 16, TOW = 32;
constexpr int PH = 2 * TOH + 4, PWX = 2 * TOW + 4;      // 36 x 68 input pixels (padded image coordinates)
constexpr int PROW = PWX * 8, PATCH_B = PH * PROW;      // 544, 19584
constexpr int PCH = PATCH_B / 16;                       // 1224 16-byte pieces
constexpr int Q = 144;                                  // 6 x 6 x 4

__device__ __forceinline__ float stem_dsilu(float u) {
    const float s = __builtin_amdgcn_rcpf(1.0f + __expf(-u));
    return s * (1.0f + u * (1.0f - s));
}

// FUSED: the stem has no data gradient, so its BatchNorm-backward output dy has exactly one reader — this kernel.  Instead of an apply
// pass that writes dy (read dz, read y, write dy: 1.26 GB at 64 x 320 x 320 x 32) and a DMA that reads it back, the tile is staged through
// registers from dz and y with the apply pass's arithmetic (same expression, same bf16 rounding point: the results are bit-identical).
template <int MT, bool FUSED>   // K = 16 * MT
__global__ __launch_bounds__(256) void wgrad_stem_kernel(const WgradArgs p) {
    constexpr int K = MT * 16, DROW = K * 2;             // bytes per dy pixel
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sP = smem;                            // [36][68][8 B]
    unsigned char* sD = smem + PATCH_B + 64;             // [512 pixels][DROW]  (+64: the padded B columns of the last patch row read past it)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, i16 = lane & 15, q4 = i16 >> 2, p4 = i16 & 3;
    const int tiles_w = p.Wo / TOW, tiles_h = p.Ho / TOH;
    const int per_img = tiles_w * tiles_h;
    const int tiles_total = p.N * per_img;
    const unsigned char* __restrict__ x = (const unsigned char*)p.x;
    const bf16_t* __restrict__ dy = (const bf16_t*)p.dy;

    f32x4 acc[MT][3];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    // this wave's three 16-column tiles of the 6 x 32 column space: nt = 3*wave + b -> (filter row r, column half j)
    int boff[3];
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        const int nt = wave * 3 + b, r = nt >> 1, j = nt & 1;
        boff[b] = r * PROW + j * 32 + p4 * 8 + (8 * g + q4) * 16;
    }
    const int aoff = (8 * g + q4) * DROW + p4 * 8;

    // FUSED: this thread's 16-byte chunk of a dy pixel is always the same 8 channels (256 % (K/8) == 0): their coefficients stay in registers
    float sc[8], sh[8], mu[8], is[8], k1[8], k2[8];
    const bf16_t* __restrict__ yraw = (const bf16_t*)p.y;
    if constexpr (FUSED) {
        const int c0 = (tid % (K / 8)) * 8;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            sc[i] = p.bn_scale[c0 + i]; sh[i] = p.bn_shift[c0 + i]; mu[i] = p.bn_mean[c0 + i];
            is[i] = p.bn_invstd[c0 + i]; k1[i] = p.bn_c1[c0 + i]; k2[i] = p.bn_c2[c0 + i];
        }
    }

    for (int t = blockIdx.x; t < tiles_total; t += gridDim.x) {
        const int n = t / per_img, rem = t - n * per_img;
        const int th = rem / tiles_w, tw = rem - th * tiles_w;
        const unsigned char* org = x + (((long long)n * p.Hin + 2 * th * TOH) * p.Win + 2 * tw * TOW) * 8;
        for (int pos = tid; pos < PCH; pos += 256) {
            const int row = pos / (PROW / 16), c = pos - row * (PROW / 16);
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(org + (long long)row * p.Win * 8 + c * 16),
                                             (void __attribute__((address_space(3)))*)(sP + (pos - lane) * 16), 16, 0, 0);
        }
        const bf16_t* dorg = dy + (((long long)n * p.Ho + th * TOH) * p.Wo + tw * TOW) * p.lddy;
        if constexpr (FUSED) {
            constexpr int NV = TOH * TOW * (K / 8) / 256;
            const bf16_t* yorg = yraw + (((long long)n * p.Ho + th * TOH) * p.Wo + tw * TOW) * p.ldy;
            i32x4 gq[NV], vq[NV];
#pragma unroll
            for (int i = 0; i < NV; ++i) {                 // all of the tile's dz / y vectors of this thread in flight at once
                const int pos = tid + 256 * i;
                const int pix = pos / (K / 8), part = pos - pix * (K / 8);
                const int py = pix / TOW, px = pix - py * TOW;
                gq[i] = *(const i32x4*)(dorg + ((long long)py * p.Wo + px) * p.lddy + part * 8);
                vq[i] = *(const i32x4*)(yorg + ((long long)py * p.Wo + px) * p.ldy + part * 8);
            }
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                V16 gv, yv, o;
                gv.i = gq[i];
                yv.i = vq[i];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float v = (float)yv.h[e];
                    const float du = (float)gv.h[e] * stem_dsilu(v * sc[e] + sh[e]);
                    const float xh = (v - mu[e]) * is[e];
                    o.h[e] = (bf16_t)(sc[e] * (du - k1[e] - xh * k2[e]));
                }
                *(i32x4*)(sD + (tid + 256 * i) * 16) = o.i;
            }
        } else {
#pragma unroll
            for (int i = 0; i < TOH * TOW * (K / 8) / 256; ++i) {
                const int pos = tid + 256 * i;
                const int pix = pos / (K / 8), part = pos - pix * (K / 8);
                const int py = pix / TOW, px = pix - py * TOW;
                __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(dorg + ((long long)py * p.Wo + px) * p.lddy + part * 8),
                                                 (void __attribute__((address_space(3)))*)(sD + (pos - lane) * 16), 16, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#pragma unroll 4
        for (int ty = 0; ty < TOH; ++ty) {                 // one MFMA k-step = the 32 pixels of tile row ty
            V16 af[MT], bf[3];
            const unsigned char* da = sD + ty * TOW * DROW + aoff;
#pragma unroll
            for (int a = 0; a < MT; ++a) {
                const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(da + a * 32));
                const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(da + a * 32 + 4 * DROW));
                af[a].h = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
            const unsigned char* xb = sP + 2 * ty * PROW;
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(xb + boff[b]));
                const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(xb + boff[b] + 4 * 16));
                bf[b].h = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
#pragma unroll
            for (int a = 0; a < MT; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[a].h, bf[b].h, acc[a][b], 0, 0, 0);
        }
        __syncthreads();                                   // everyone is done with this tile's LDS image
    }

    float* out = p.partial + (size_t)blockIdx.x * K * Q;
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        const int nt = wave * 3 + b, r = nt >> 1, col = (nt & 1) * 16 + i16;
        if (col >= 24) continue;                           // padding columns
#pragma unroll
        for (int a = 0; a < MT; ++a)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) out[(size_t)(a * 16 + g * 4 + rr) * Q + r * 24 + col] = acc[a][b][rr];
    }
}

}  // namespace stemw

// grad[k][c][r][s] (+)= sum_split partial[split][k][q]
// mode 0: q = (r*S + s)*C + c          mode 1 (stem): q = r*(S*4) + s*4 + c, c < 3
// block = 64 four-output columns x 16 split-lanes; a lane keeps four 16-byte loads of different slabs in flight (the first version read
// one dword per lane in a dependent loop: 1.5 TB/s on 32 MB of slabs, 1.25 ms of the train step).  Fixed summation order.
__device__ __forceinline__ void wgrad_scatter(float v, int idx, int K, int Q, int mode, int C, int R, int S, float* __restrict__ grad, int accumulate) {
    const int k = idx / Q, q = idx - k * Q;
    int c, r, s;
    if (mode == 0) {
        const int t = q / C;
        c = q - t * C;
        r = t / S;
        s = t - r * S;
    } else {
        r = q / (S * 4);
        const int rem = q - r * (S * 4);
        s = rem >> 2;
        c = rem & 3;
        if (c >= C) return;
    }
    float* g = grad + (((size_t)k * C + c) * R + r) * S + s;
    *g = accumulate ? *g + v : v;
}

__global__ __launch_bounds__(1024) void wgrad_reduce_kernel(const float* __restrict__ partial, int splits, size_t slab_stride, int K, int Q,
                                                            int mode, int C, int R, int S, float* __restrict__ grad, int accumulate) {
    __shared__ f32x4 red[16][64];
    const int ql = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int idx = (blockIdx.x * 64 + ql) * 4;
    const bool ok = idx < K * Q;
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
    if (ok) {
        const float* src = partial + idx;
        int sp = sl;
        for (; sp + 48 < splits; sp += 64) {
            const f32x4 a = *(const f32x4*)(src + (size_t)sp * slab_stride);
            const f32x4 b = *(const f32x4*)(src + (size_t)(sp + 16) * slab_stride);
            const f32x4 c = *(const f32x4*)(src + (size_t)(sp + 32) * slab_stride);
            const f32x4 d = *(const f32x4*)(src + (size_t)(sp + 48) * slab_stride);
            s0 += a; s1 += b; s2 += c; s3 += d;
        }
        for (; sp < splits; sp += 16) s0 += *(const f32x4*)(src + (size_t)sp * slab_stride);
    }
    red[sl][ql] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (sl != 0 || !ok) return;
    f32x4 sum = red[0][ql];
#pragma unroll
    for (int i = 1; i < 16; ++i) sum += red[i][ql];
    if (mode == 0 && R == 1 && S == 1) {                 // 1x1: q == c, the four outputs are contiguous
        f32x4* g = (f32x4*)(grad + idx);
        *g = accumulate ? *g + sum : sum;
        return;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) wgrad_scatter(sum[e], idx + e, K, Q, mode, C, R, S, grad, accumulate);
}


// the same for slabs that are not whole 16-byte columns (K*Q, the slab pitch or the base not a multiple of four floats)
__global__ __launch_bounds__(1024) void wgrad_reduce_scalar_kernel(const float* __restrict__ partial, int splits, size_t slab_stride, int K, int Q,
                                                                   int mode, int C, int R, int S, float* __restrict__ grad, int accumulate) {
    __shared__ float red[16][65];
    const int ql = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int idx = blockIdx.x * 64 + ql;
    const bool ok = idx < K * Q;
    float sum = 0.f;
    if (ok)
        for (int sp = sl; sp < splits; sp += 16) sum += partial[(size_t)sp * slab_stride + idx];
    red[sl][ql] = sum;
    __syncthreads();
    if (sl != 0 || !ok) return;
#pragma unroll
    for (int i = 1; i < 16; ++i) sum += red[i][ql];
    wgrad_scatter(sum, idx, K, Q, mode, C, R, S, grad, accumulate);
}

// Packing: framework weight w[K][C][R][S] (fp32) -> out[row][t][cdim] of type T, pitch Kdp, zero padded.
//   transpose == 0: row = k, cdim = c (forward / wgrad geometry)
//   transpose == 1: row = c, cdim = k (dgrad geometry)
//   tap t = (tr, ts) in a TH x TW window reads source (r, s) = (rbase + rstep*tr, sbase + sstep*ts)
//   stem == 1: forward only, cdim = s*4 + c over a [R][S*4] window (TH = R, TW = 1), c == 3 is zero
__device__ __forceinline__ float pack_value(const hdy_pack_desc& d, unsigned idx) {      // idx < rows_total * Kdp < 2^31 (checked where the job is made)
    // logical weight W[k][c][r][s], k < Kl: rows of w_a, then rows of w_b, then zeros (channel padding)
    const int row = (int)(idx / (unsigned)d.Kdp), col = (int)(idx - (unsigned)row * (unsigned)d.Kdp);
    int k = -1, c = 0, r = 0, s = 0;
    if (d.stem) {
        r = col / (d.S * 4);
        const int rem = col - r * (d.S * 4);
        s = rem >> 2;
        c = rem & 3;
        if (r < d.R && c < d.C) k = row;
    } else {
        const int cd = d.transpose ? d.Kl : d.C;
        const int t = col / cd, ci = col - t * cd;
        if (t < d.TH * d.TW) {
            const int tr = t / d.TW, ts = t - tr * d.TW;
            r = d.rbase + d.rstep * tr;
            s = d.sbase + d.sstep * ts;
            c = d.transpose ? row : ci;
            if (r >= 0 && r < d.R && s >= 0 && s < d.S && c < d.C) k = d.transpose ? ci : row;
        }
    }
    float v = 0.f;
    if (k >= 0 && k < d.K_a) v = d.w_a[(((size_t)k * d.C + c) * d.R + r) * d.S + s];
    else if (k >= d.K_a && k < d.K_a + d.K_b) v = d.w_b[(((size_t)(k - d.K_a) * d.C + c) * d.R + r) * d.S + s];
    return v;
}

// Eight consecutive outputs (one 16-byte bf16 store) per thread: inside one tap the source row / column and the bounds are shared, only the
// channel moves (stride R*S or C*R*S floats in the source).  Kdp is a multiple of 8, so a vector never crosses a packed row.
__device__ __forceinline__ void pack8(const hdy_pack_desc& d, unsigned idx0) {
    const int row = (int)(idx0 / (unsigned)d.Kdp), col0 = (int)(idx0 - (unsigned)row * (unsigned)d.Kdp);
    float v[8];
    const int cd = d.transpose ? d.Kl : d.C;
    const int t = d.stem ? 0 : col0 / cd, ci0 = col0 - t * cd;
    if (!d.stem && ci0 + 8 <= cd) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = 0.f;
        if (t < d.TH * d.TW) {
            const int tr = t / d.TW, ts = t - tr * d.TW;
            const int r = d.rbase + d.rstep * tr, sx = d.sbase + d.sstep * ts;
            if (r >= 0 && r < d.R && sx >= 0 && sx < d.S) {
                const size_t rs = (size_t)r * d.S + sx, RS = (size_t)d.R * d.S;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int c = d.transpose ? row : ci0 + j, k = d.transpose ? ci0 + j : row;
                    if (c < d.C) {
                        if (k < d.K_a) v[j] = d.w_a[((size_t)k * d.C + c) * RS + rs];
                        else if (k < d.K_a + d.K_b) v[j] = d.w_b[((size_t)(k - d.K_a) * d.C + c) * RS + rs];
                    }
                }
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = pack_value(d, idx0 + j);
    }
    if (d.dtype == HDY_BF16) {
        V16 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o.h[j] = (bf16_t)v[j];
        *(i32x4*)((bf16_t*)d.out + idx0) = o.i;
    } else {
        *(f32x4*)((float*)d.out + idx0) = f32x4{v[0], v[1], v[2], v[3]};
        *(f32x4*)((float*)d.out + idx0 + 4) = f32x4{v[4], v[5], v[6], v[7]};
    }
}

// a workgroup packs 256 x 8 = 2048 outputs: hdy_pack_desc.nblocks (api.hip make_desc) counts those

// one job per launch (descriptor by value)
__global__ __launch_bounds__(256) void pack_weight_kernel(const hdy_pack_desc d) {
    const unsigned idx = (blockIdx.x * blockDim.x + threadIdx.x) * 8;
    if (idx >= (unsigned)d.rows_total * (unsigned)d.Kdp) return;
    pack8(d, idx);
}

// every job of a plan in one launch: block -> descriptor by binary search over first_block
__global__ __launch_bounds__(256) void pack_batch_kernel(const hdy_pack_desc* __restrict__ table, int n) {
    int lo = 0, hi = n - 1;
    const int b = blockIdx.x;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (table[mid].first_block <= b) lo = mid; else hi = mid - 1;
    }
    const hdy_pack_desc d = table[lo];
    const unsigned idx = ((unsigned)(b - d.first_block) * blockDim.x + threadIdx.x) * 8;
    if (idx >= (unsigned)d.rows_total * (unsigned)d.Kdp) return;
    pack8(d, idx);
}

}  // namespace

static inline void wgrad_tile(int K, int Q, long long P, int dtype, int* sd, int* sx) {
    const int TK = dtype == HDY_BF16 ? 64 : 32;
    // measurement switch (profiles/r05_wgrad_splits_ab.txt): 64 = 64 x 64 blocks everywhere, N > 64 = only for layers with at most N pixels
    const int opt = hdy_opt(HDY_OPT_WGRAD_TILE);
    const bool small = opt == 64 || (opt > 64 && P <= opt);
    *sd = (K > TK && !small) ? 2 : 1;
    *sx = (Q > TK && !small) ? 2 : 1;
}

int hdy_wgrad_plan(int K, int Q, long long P, int dtype, int* splits, int* pix_per_split) {
    const int TK = dtype == HDY_BF16 ? 64 : 32;
    int sd, sx;
    wgrad_tile(K, Q, P, dtype, &sd, &sx);
    const int tiles = cdiv(K, TK * sd) * cdiv(Q, TK * sx);
    const int target = hdy_opt(HDY_OPT_WGRAD_BLOCKS);     // = resident workgroups (2 per CU): one wave of blocks, half the slab traffic of 1024
    int s = cdiv(target, tiles);
    if (s > 512) s = 512;
    const int maxs = cdiv(P, 256);
    if (s > maxs) s = maxs;
    if (s < 1) s = 1;
    int pps = round_up(cdiv(P, s), PB);
    s = cdiv(P, pps);
    *splits = s;
    *pix_per_split = pps;
    return HDY_OK;
}

template <typename T>
static void wgrad_dispatch(const WgradArgs& a, int sd, int sx, int grid, hipStream_t st) {
    hdy_note_dispatch("wgrad_generic");
    if (sd == 2 && sx == 2) hipLaunchKernelGGL((wgrad_kernel<T, 2, 2>), dim3(grid), dim3(256), 0, st, a);
    else if (sd == 2) hipLaunchKernelGGL((wgrad_kernel<T, 2, 1>), dim3(grid), dim3(256), 0, st, a);
    else if (sx == 2) hipLaunchKernelGGL((wgrad_kernel<T, 1, 2>), dim3(grid), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((wgrad_kernel<T, 1, 1>), dim3(grid), dim3(256), 0, st, a);
}

int hdy_wgrad_launch(WgradArgs a, int dtype, hipStream_t st) {
    const int VE = dtype == HDY_BF16 ? 8 : 4, TK = dtype == HDY_BF16 ? 64 : 32;
    HDY_ARG(a.x && a.dy && a.partial, "wgrad: null pointer");
    HDY_ARG(a.C % VE == 0 && a.K % VE == 0, "wgrad: C=%d and K=%d must be multiples of %d", a.C, a.K, VE);
    HDY_ARG(a.ldx % (a.span_pixels ? 4 : VE) == 0 && a.lddy % VE == 0 && (a.span_pixels || a.ldx >= a.C) && a.lddy >= a.K, "wgrad: bad pitches ldx=%d lddy=%d", a.ldx, a.lddy);
    HDY_ARG(((uintptr_t)a.x & 15) == 0 && ((uintptr_t)a.dy & 15) == 0, "wgrad: x/dy must be 16-byte aligned");
    HDY_ARG((long long)a.N * a.Hin * a.Win < (1LL << 31) && (long long)a.N * a.Ho * a.Wo < (1LL << 31), "wgrad: too many pixels");
    a.Q = a.TH * a.TW * a.C;
    a.P = a.N * a.Ho * a.Wo;
    int sd, sx;
    wgrad_tile(a.K, a.Q, a.P, dtype, &sd, &sx);
    a.ktiles = cdiv(a.K, TK * sd);
    a.qtiles = cdiv(a.Q, TK * sx);
    HDY_ARG(a.splits >= 1 && a.pix_per_split % PB == 0 && (long long)a.splits * a.pix_per_split >= a.P, "wgrad: bad split plan");
    const int grid = a.splits * a.ktiles * a.qtiles;
    if (dtype == HDY_BF16) wgrad_dispatch<bf16_t>(a, sd, sx, grid, st);
    else wgrad_dispatch<float>(a, sd, sx, grid, st);
    HDY_LAUNCH_CHECK("wgrad");
    return HDY_OK;
}

// Workgroups (= fp32 slabs) of the patch-resident stem weight gradient, 0 = shape not eligible (generic kernel).
int hdy_wgrad_stem_grid(int N, int Ho, int Wo, int K, int dtype) {
    const bool disabled = hdy_opt(HDY_OPT_NO_STEM_WGRAD) != 0;
    if (disabled || dtype != HDY_BF16 || K % 16 != 0 || K > 64 || Ho % stemw::TOH != 0 || Wo % stemw::TOW != 0) return 0;
    const long long tiles = (long long)N * (Ho / stemw::TOH) * (Wo / stemw::TOW);
    const int cap = K <= 32 ? 768 : 256;                 // 52 KB of LDS per workgroup at K = 32: three per CU
    return (int)(tiles < cap ? tiles : cap);
}

int hdy_wgrad_stem_launch(const WgradArgs& a, int grid, hipStream_t st) {
    HDY_ARG(((uintptr_t)a.x & 15) == 0 && ((uintptr_t)a.dy & 15) == 0 && a.lddy % 8 == 0 && a.Win % 2 == 0, "wgrad(stem): x/dy alignment");
    const size_t smem = stemw::PATCH_B + 64 + (size_t)stemw::TOH * stemw::TOW * a.K * 2;
    const int mt = a.K / 16;
    hdy_note_dispatch(a.y ? "wgrad_stem_fused" : "wgrad_stem");
#define STEMW_LAUNCH(MT, FU)                                                                                                         \
    {                                                                                                                              \
        (void)hipFuncSetAttribute((const void*)stemw::wgrad_stem_kernel<MT, FU>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); \
        hipLaunchKernelGGL((stemw::wgrad_stem_kernel<MT, FU>), dim3(grid), dim3(256), smem, st, a);                                \
    }
    if (a.y) {
        HDY_ARG((mt == 1 || mt == 2 || mt == 4) && a.ldy % 8 == 0 && ((uintptr_t)a.y & 15) == 0 && a.bn_scale && a.bn_shift && a.bn_mean && a.bn_invstd &&
                a.bn_c1 && a.bn_c2, "wgrad(stem, fused BatchNorm backward): K must be 16, 32 or 64 and all coefficient vectors given");
        if (mt == 1) STEMW_LAUNCH(1, true) else if (mt == 2) STEMW_LAUNCH(2, true) else STEMW_LAUNCH(4, true)
    } else if (mt == 1) STEMW_LAUNCH(1, false) else if (mt == 2) STEMW_LAUNCH(2, false) else if (mt == 3) STEMW_LAUNCH(3, false) else STEMW_LAUNCH(4, false)
#undef STEMW_LAUNCH
    HDY_LAUNCH_CHECK("wgrad(stem)");
    return HDY_OK;
}

// partial points at the first row to reduce; slabs are slab_stride floats apart; K rows of Q are reduced.
int hdy_wgrad_reduce_launch(const float* partial, int splits, size_t slab_stride, int K, int Q, int mode, int C, int R, int S, float* grad,
                            int accumulate, hipStream_t st) {
    const int n = K * Q;
    const bool vec = n % 4 == 0 && slab_stride % 4 == 0 && ((uintptr_t)partial & 15) == 0 && (!(mode == 0 && R == 1 && S == 1) || ((uintptr_t)grad & 15) == 0);
    if (vec)
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(cdiv(n, 256)), dim3(1024), 0, st, partial, splits, slab_stride, K, Q, mode, C, R, S, grad, accumulate);
    else
        hipLaunchKernelGGL(wgrad_reduce_scalar_kernel, dim3(cdiv(n, 64)), dim3(1024), 0, st, partial, splits, slab_stride, K, Q, mode, C, R, S, grad,
                           accumulate);
    HDY_LAUNCH_CHECK("wgrad_reduce");
    return HDY_OK;
}

int hdy_pack_weight_launch(const hdy_pack_desc& d, hipStream_t st) {
    hipLaunchKernelGGL(pack_weight_kernel, dim3(d.nblocks), dim3(256), 0, st, d);
    HDY_LAUNCH_CHECK("pack_weight");
    return HDY_OK;
}

int hdy_pack_batch_launch(const hdy_pack_desc* table, int n, int total_blocks, hipStream_t st) {
    hipLaunchKernelGGL(pack_batch_kernel, dim3(total_blocks), dim3(256), 0, st, table, n);
    HDY_LAUNCH_CHECK("pack_batch");
    return HDY_OK;
}

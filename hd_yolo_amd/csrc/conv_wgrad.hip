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
#include "common.h"
#include "hdyolo_internal.h"

__device__ uint4 g_hdy_zero16_w[4];   // zero page for masked 16-byte fetches

namespace {

__device__ __forceinline__ int fsw(int row) { return ((row >> 1) & 1) | (((row >> 3) & 1) << 1); }

template <typename T> struct WT;
template <> struct WT<bf16_t> { static constexpr int VE = 8, TK = 64, WTL = 2; };   // tiles per wave side
template <> struct WT<float> { static constexpr int VE = 4, TK = 32, WTL = 1; };

constexpr int PB = 64;   // pixels per LDS stage

template <typename T>
__global__ __launch_bounds__(256) void wgrad_kernel(const WgradArgs p) {
    constexpr int VE = WT<T>::VE, TK = WT<T>::TK, WTL = WT<T>::WTL;
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * 2 * PB * 128];
    unsigned char* sD = smem;                   // dy tiles  [2][PB][128]
    unsigned char* sX = smem + 2 * PB * 128;    // x tiles   [2][PB][128]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles = p.ktiles * p.qtiles;
    const int split = blockIdx.x / tiles;
    const int tile = blockIdx.x - split * tiles;
    const int kt = tile / p.qtiles, qt = tile - kt * p.qtiles;
    const int k0 = kt * TK, q0 = qt * TK;
    const int pbeg = split * p.pix_per_split;
    const int pend = min(pbeg + p.pix_per_split, p.P);

    const T* __restrict__ x = (const T*)p.x;
    const T* __restrict__ dy = (const T*)p.dy;

    const int r0 = tid >> 3;
    // LDS-DMA writes linearly (wave base + lane*16): this thread fills physical slot (row, tid&7) of both tiles, so it
    // fetches the LOGICAL 16-byte chunk that the 32-byte-block swizzle places there (row bits 1,3 are the same for its 2 rows)
    const int c8 = ((((tid & 7) >> 1) ^ fsw(r0)) << 1) | (tid & 1);
    // fixed per thread: dy channel of its chunk, x (tap, channel) of its chunk
    const int kch = k0 + c8 * VE;
    const bool k_ok = kch < p.K;
    const int q = q0 + c8 * VE;
    const bool q_ok = q < p.Q;
    int th = 0, tw = 0, cch = 0;
    if (q_ok) {
        const int tap = q / p.C;
        cch = q - tap * p.C;
        th = tap / p.TW;
        tw = tap - th * p.TW;
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

    const unsigned char* zero = (const unsigned char*)g_hdy_zero16_w;
    auto stage = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const bool live = pp[i] < pend;
            const void* sd = zero;
            const void* sx = zero;
            if (live && k_ok) sd = dy + (size_t)pp[i] * p.lddy + kch;
            if (live && q_ok) {
                const int hi = pi[i] * p.ih_mul + p.dh0 + th, wi = pj[i] * p.iw_mul + p.dw0 + tw;
                if ((unsigned)hi < (unsigned)p.Hin && (unsigned)wi < (unsigned)p.Win)
                    sx = x + ((size_t)(pn[i] * p.Hin + hi) * p.Win + wi) * p.ldx + cch;
            }
            const int woff = (wave * 64 + 256 * i) * 16;
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)sd,
                                             (void __attribute__((address_space(3)))*)(sD + buf * PB * 128 + woff), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)sx,
                                             (void __attribute__((address_space(3)))*)(sX + buf * PB * 128 + woff), 16, 0, 0);
            // advance this row by one stage
            pp[i] += PB;
            pj[i] += PB;
            while (pj[i] >= p.Wo) {
                pj[i] -= p.Wo;
                if (++pi[i] == p.Ho) { pi[i] = 0; ++pn[i]; }
            }
        }
    };

    f32x4 acc[WTL][WTL];
#pragma unroll
    for (int a = 0; a < WTL; ++a)
#pragma unroll
        for (int b = 0; b < WTL; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nst = (pend - pbeg + PB - 1) / PB;
    if (nst > 0) stage(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int g = lane >> 4, i16 = lane & 15;
    for (int s = 0; s < nst; ++s) {
        const int cur = s & 1;
        if (s + 1 < nst) stage(cur ^ 1);
        const unsigned char* d_s = sD + cur * PB * 128;
        const unsigned char* x_s = sX + cur * PB * 128;
        if constexpr (sizeof(T) == 2) {
            const int q4 = i16 >> 2, p4 = i16 & 3;
#pragma unroll
            for (int ks = 0; ks < PB / 32; ++ks) {
                V16 af[WTL], bf[WTL];
                const int ra = ks * 32 + 8 * g + q4, rb = ra + 4;
#pragma unroll
                for (int a = 0; a < WTL; ++a) {
                    const int cb = (wm * 32 + a * 16) >> 4;
                    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                        (bf16x4 __attribute__((address_space(3)))*)(d_s + ra * 128 + ((cb ^ fsw(ra)) << 5) + p4 * 8));
                    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                        (bf16x4 __attribute__((address_space(3)))*)(d_s + rb * 128 + ((cb ^ fsw(rb)) << 5) + p4 * 8));
                    af[a].h = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                }
#pragma unroll
                for (int b = 0; b < WTL; ++b) {
                    const int cb = (wn * 32 + b * 16) >> 4;
                    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                        (bf16x4 __attribute__((address_space(3)))*)(x_s + ra * 128 + ((cb ^ fsw(ra)) << 5) + p4 * 8));
                    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                        (bf16x4 __attribute__((address_space(3)))*)(x_s + rb * 128 + ((cb ^ fsw(rb)) << 5) + p4 * 8));
                    bf[b].h = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                }
#pragma unroll
                for (int a = 0; a < WTL; ++a)
#pragma unroll
                    for (int b = 0; b < WTL; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[a].h, bf[b].h, acc[a][b], 0, 0, 0);
            }
        } else {
            // fp32: A[k-channel = lane&15][pixel = lane>>4] straight from the [pixel][channel] image
            const int ca = wm * 16 + i16, cb_ = wn * 16 + i16;
#pragma unroll 4
            for (int ks = 0; ks < PB / 4; ++ks) {
                const int row = ks * 4 + g;
                const float av = *(const float*)(d_s + row * 128 + (((ca >> 3) ^ fsw(row)) << 5) + (ca & 7) * 4);
                const float bv = *(const float*)(x_s + row * 128 + (((cb_ >> 3) ^ fsw(row)) << 5) + (cb_ & 7) * 4);
                acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[0][0], 0, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    float* out = p.partial + (size_t)split * p.K * p.Q;
#pragma unroll
    for (int a = 0; a < WTL; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k = k0 + wm * (TK / 2) + a * 16 + g * 4 + r;
            if (k >= p.K) continue;
#pragma unroll
            for (int b = 0; b < WTL; ++b) {
                const int qq = q0 + wn * (TK / 2) + b * 16 + i16;
                if (qq < p.Q) out[(size_t)k * p.Q + qq] = acc[a][b][r];
            }
        }
}

// grad[k][c][r][s] (+)= sum_split partial[split][k][q]
// mode 0: q = (r*S + s)*C + c          mode 1 (stem): q = r*(S*4) + s*4 + c, c < 3
// block = 64 outputs x 16 split-lanes: slabs are read coalesced along q and 16 splits are in flight per output
__global__ __launch_bounds__(1024) void wgrad_reduce_kernel(const float* __restrict__ partial, int splits, size_t slab_stride, int K, int Q,
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
    *g = accumulate ? *g + sum : sum;
}

// Packing: framework weight w[K][C][R][S] (fp32) -> out[row][t][cdim] of type T, pitch Kdp, zero padded.
//   transpose == 0: row = k, cdim = c (forward / wgrad geometry)
//   transpose == 1: row = c, cdim = k (dgrad geometry)
//   tap t = (tr, ts) in a TH x TW window reads source (r, s) = (rbase + rstep*tr, sbase + sstep*ts)
//   stem == 1: forward only, cdim = s*4 + c over a [R][S*4] window (TH = R, TW = 1), c == 3 is zero
template <typename T>
__global__ void pack_weight_kernel(const float* __restrict__ w_a, int K_a, const float* __restrict__ w_b, int K_b, T* __restrict__ out,
                                   int Kl, int C, int R, int S, int transpose, int TH, int TW, int rbase, int rstep, int sbase,
                                   int sstep, int stem, int rows_total, int Kdp) {
    // logical weight W[k][c][r][s], k < Kl: rows of w_a, then rows of w_b, then zeros (channel padding)
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)rows_total * Kdp) return;
    const int row = (int)(idx / Kdp), col = (int)(idx - (long long)row * Kdp);
    int k = -1, c = 0, r = 0, s = 0;
    if (stem) {
        r = col / (S * 4);
        const int rem = col - r * (S * 4);
        s = rem >> 2;
        c = rem & 3;
        if (r < R && c < C) k = row;
    } else {
        const int cd = transpose ? Kl : C;
        const int t = col / cd, ci = col - t * cd;
        if (t < TH * TW) {
            const int tr = t / TW, ts = t - tr * TW;
            r = rbase + rstep * tr;
            s = sbase + sstep * ts;
            c = transpose ? row : ci;
            if (r >= 0 && r < R && s >= 0 && s < S && c < C) k = transpose ? ci : row;
        }
    }
    float v = 0.f;
    if (k >= 0 && k < K_a) v = w_a[(((size_t)k * C + c) * R + r) * S + s];
    else if (k >= K_a && k < K_a + K_b) v = w_b[(((size_t)(k - K_a) * C + c) * R + r) * S + s];
    out[idx] = from_f32<T>(v);
}

}  // namespace

int hdy_wgrad_plan(int K, int Q, long long P, int dtype, int* splits, int* pix_per_split) {
    const int TK = dtype == HDY_BF16 ? 64 : 32;
    const int tiles = cdiv(K, TK) * cdiv(Q, TK);
    int s = cdiv(1024, tiles);
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

int hdy_wgrad_launch(WgradArgs a, int dtype, hipStream_t st) {
    const int VE = dtype == HDY_BF16 ? 8 : 4, TK = dtype == HDY_BF16 ? 64 : 32;
    HDY_ARG(a.x && a.dy && a.partial, "wgrad: null pointer");
    HDY_ARG(a.C % VE == 0 && a.K % VE == 0, "wgrad: C=%d and K=%d must be multiples of %d", a.C, a.K, VE);
    HDY_ARG(a.ldx % (a.span_pixels ? 4 : VE) == 0 && a.lddy % VE == 0 && (a.span_pixels || a.ldx >= a.C) && a.lddy >= a.K, "wgrad: bad pitches ldx=%d lddy=%d", a.ldx, a.lddy);
    HDY_ARG(((uintptr_t)a.x & 15) == 0 && ((uintptr_t)a.dy & 15) == 0, "wgrad: x/dy must be 16-byte aligned");
    HDY_ARG((long long)a.N * a.Hin * a.Win < (1LL << 31) && (long long)a.N * a.Ho * a.Wo < (1LL << 31), "wgrad: too many pixels");
    a.Q = a.TH * a.TW * a.C;
    a.P = a.N * a.Ho * a.Wo;
    a.ktiles = cdiv(a.K, TK);
    a.qtiles = cdiv(a.Q, TK);
    HDY_ARG(a.splits >= 1 && a.pix_per_split % PB == 0 && (long long)a.splits * a.pix_per_split >= a.P, "wgrad: bad split plan");
    const int grid = a.splits * a.ktiles * a.qtiles;
    if (dtype == HDY_BF16)
        hipLaunchKernelGGL(wgrad_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, a);
    else
        hipLaunchKernelGGL(wgrad_kernel<float>, dim3(grid), dim3(256), 0, st, a);
    HDY_LAUNCH_CHECK("wgrad");
    return HDY_OK;
}

// partial points at the first row to reduce; slabs are slab_stride floats apart; K rows of Q are reduced.
int hdy_wgrad_reduce_launch(const float* partial, int splits, size_t slab_stride, int K, int Q, int mode, int C, int R, int S, float* grad,
                            int accumulate, hipStream_t st) {
    const int n = K * Q;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(cdiv(n, 64)), dim3(1024), 0, st, partial, splits, slab_stride, K, Q, mode, C, R, S, grad,
                       accumulate);
    HDY_LAUNCH_CHECK("wgrad_reduce");
    return HDY_OK;
}

int hdy_pack_weight_launch(const float* w_a, int K_a, const float* w_b, int K_b, void* out, int Kl, int C, int R, int S, int transpose,
                           int TH, int TW, int rbase, int rstep, int sbase, int sstep, int stem, int rows_total, int Kdp, int dtype,
                           hipStream_t st) {
    const long long n = (long long)rows_total * Kdp;
    const int grid = cdiv(n, 256);
    if (dtype == HDY_BF16)
        hipLaunchKernelGGL(pack_weight_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, w_a, K_a, w_b, K_b, (bf16_t*)out, Kl, C, R, S,
                           transpose, TH, TW, rbase, rstep, sbase, sstep, stem, rows_total, Kdp);
    else
        hipLaunchKernelGGL(pack_weight_kernel<float>, dim3(grid), dim3(256), 0, st, w_a, K_a, w_b, K_b, (float*)out, Kl, C, R, S, transpose,
                           TH, TW, rbase, rstep, sbase, sstep, stem, rows_total, Kdp);
    HDY_LAUNCH_CHECK("pack_weight");
    return HDY_OK;
}

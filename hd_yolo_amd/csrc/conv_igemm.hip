// Implicit-GEMM "tap convolution" for gfx950: the one kernel behind conv forward, dgrad (stride 1 and the
// four parity classes of stride 2) and the 6x6/s2 stem (as 6 row-taps over a 24-wide pseudo channel).
//
//   y[n, oh_off + i*oh_mul, ow_off + j*ow_mul, k] (+)= act( scale[k] * SUM_{t,c} x[n, i*ih_mul + dh0 + t/TW,
//                                                     j*iw_mul + dw0 + t%TW, c] * w[k][t][c] + shift[k] )
//
// Activations are NHWC with an explicit pixel pitch (ldx / ldy elements), so a tensor may be a channel
// slice of a wider concat buffer; out-of-image taps read as zero.  GEMM view: M = N*Ho*Wo pixels,
// N = K output channels, Kd = T*C.  Weights are pre-packed [Kpad][Kdp] (K-contiguous per output channel,
// zero padded), see pack.hip.
//
// Tiling (CDNA4): 256 threads = 4 waves as 2(M) x 2(N); block tile 128 x BN x 128 bytes of K; each wave owns
// (64 x BN/2) as 16x16 MFMA tiles.  One template serves both arithmetic types: LDS rows are 128 B
// (64 bf16 / 32 f32), a fragment is one 16-byte ds_read_b128 per lane, and mma16() is one
// v_mfma_f32_16x16x32_bf16 or four v_mfma_f32_16x16x4_f32 (exact fp32, used for the 1e-4 parity mode).
// LDS chunk index is XOR-swizzled with (row>>1)&7 so the 16 rows of a fragment read hit 16 distinct
// 16-byte slots (conflict-free ds_read_b128).  Register-staged double buffering: global loads for
// k-block kb+1 are issued before the MFMAs of kb and written to the other LDS buffer after them.
//
// Train-mode BatchNorm support: when `stats` is given, each block also writes per-channel partial
// sum / sum-of-squares of its fp32 accumulators (rows >= M are zero by construction) to
// stats[mtile][2][K]; bn_finalize reduces the slabs deterministically (no atomics).
//
// Reference semantics replaced: nn.Conv2d inside metayolo/models/layers.py:31 (Conv), :92-93 (Bottleneck),
// :124-126 (C3), :179-180 (SPPF), yolo_head.py:112 (det conv), and autograd's conv backward-data.
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

constexpr int BM = 128;

template <typename T, typename OT, int BN>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvArgs p) {
    constexpr int VE = Traits<T>::VE;
    constexpr int BKE = 8 * VE;          // elements per 128-byte k-block
    constexpr int MT = BM / 32;          // 16-row tiles per wave
    constexpr int NT = BN / 32;          // 16-col tiles per wave
    constexpr int AR = BM / 32;          // A rows staged per thread
    constexpr int BR = BN / 32;          // B rows staged per thread
    constexpr int ASZ = BM * 128, BSZ = BN * 128;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sA = smem;                 // [2][BM][128]
    unsigned char* sB = smem + 2 * ASZ;       // [2][BN][128]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int ntiles = p.ntiles;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int mtile = bid / ntiles, ntile = bid - mtile * ntiles;
    const int m0 = mtile * BM, n0 = ntile * BN;

    const T* __restrict__ x = (const T*)p.x;
    const T* __restrict__ w = (const T*)p.w;

    // ---- per-thread staging coordinates (fixed for the whole k loop)
    const int c8 = tid & 7, r0 = tid >> 3;
    int pixbase[AR], hb[AR], wb[AR];
    const int HoWo = p.Ho * p.Wo;
#pragma unroll
    for (int i = 0; i < AR; ++i) {
        const int m = m0 + r0 + 32 * i;
        if (m < p.M) {
            const int n = m / HoWo, rem = m - n * HoWo;
            const int oi = rem / p.Wo, oj = rem - oi * p.Wo;
            pixbase[i] = n * p.Hin * p.Win;
            hb[i] = oi * p.ih_mul + p.dh0;
            wb[i] = oj * p.iw_mul + p.dw0;
        } else {
            pixbase[i] = 0;
            hb[i] = -(1 << 28);          // fails every bounds test -> zero rows
            wb[i] = 0;
        }
    }
    int cc = c8 * VE, th = 0, tw = 0;   // (tap row, tap col, channel) of this thread's chunk in k-block 0
    while (cc >= p.C) {
        cc -= p.C;
        if (++tw == p.TW) { tw = 0; ++th; }
    }
    const T* wrow[BR];
#pragma unroll
    for (int i = 0; i < BR; ++i) wrow[i] = w + (size_t)(n0 + r0 + 32 * i) * p.Kdp + c8 * VE;

    i32x4 ra[AR], rb[BR];
    auto load_tiles = [&](int kb) {
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            const int hi = hb[i] + th, wi = wb[i] + tw;
            const bool ok = (th < p.TH) && (unsigned)hi < (unsigned)p.Hin && (unsigned)wi < (unsigned)p.Win;
            i32x4 v = {0, 0, 0, 0};
            if (ok) v = *(const i32x4*)(x + ((size_t)(pixbase[i] + hi * p.Win + wi) * p.ldx + cc));
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < BR; ++i) rb[i] = *(const i32x4*)(wrow[i] + (size_t)kb * BKE);
        // advance this thread's chunk to the next k-block
        cc += BKE;
        while (cc >= p.C) {
            cc -= p.C;
            if (++tw == p.TW) { tw = 0; ++th; }
        }
    };
    auto store_tiles = [&](int buf) {
#pragma unroll
        for (int i = 0; i < AR; ++i) *(i32x4*)(sA + buf * ASZ + swz(r0 + 32 * i, c8)) = ra[i];
#pragma unroll
        for (int i = 0; i < BR; ++i) *(i32x4*)(sB + buf * BSZ + swz(r0 + 32 * i, c8)) = rb[i];
    };

    f32x4 acc[MT][NT];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nkb = p.Kdp / BKE;
    load_tiles(0);
    store_tiles(0);
    __syncthreads();
    const int fr = lane & 15, fq = lane >> 4;
    for (int kb = 0; kb < nkb; ++kb) {
        const int cur = kb & 1;
        if (kb + 1 < nkb) load_tiles(kb + 1);
        const unsigned char* a_s = sA + cur * ASZ;
        const unsigned char* b_s = sB + cur * BSZ;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            V16 af[MT], bf[NT];
#pragma unroll
            for (int a = 0; a < MT; ++a) af[a].i = *(const i32x4*)(a_s + swz(wm * (BM / 2) + a * 16 + fr, ks * 4 + fq));
#pragma unroll
            for (int b = 0; b < NT; ++b) bf[b].i = *(const i32x4*)(b_s + swz(wn * (BN / 2) + b * 16 + fr, ks * 4 + fq));
#pragma unroll
            for (int a = 0; a < MT; ++a)
#pragma unroll
                for (int b = 0; b < NT; ++b) acc[a][b] = mma16<T>(af[a], bf[b], acc[a][b]);
        }
        if (kb + 1 < nkb) store_tiles(cur ^ 1);
        __syncthreads();
    }

    // ---- BatchNorm partial statistics of the raw accumulators
    if (p.stats) {
        float* red = (float*)smem;       // [2 (wm)][BN][2]; the k loop ended with a barrier
#pragma unroll
        for (int b = 0; b < NT; ++b) {
            float s = 0.f, ss = 0.f;
#pragma unroll
            for (int a = 0; a < MT; ++a)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = acc[a][b][r];
                    s += v;
                    ss += v * v;
                }
            s += __shfl_xor(s, 16);
            ss += __shfl_xor(ss, 16);
            s += __shfl_xor(s, 32);
            ss += __shfl_xor(ss, 32);
            if (lane < 16) {
                const int col = wn * (BN / 2) + b * 16 + lane;
                red[(wm * BN + col) * 2 + 0] = s;
                red[(wm * BN + col) * 2 + 1] = ss;
            }
        }
        __syncthreads();
        if (tid < BN && n0 + tid < p.K) {
            const float s = red[tid * 2] + red[(BN + tid) * 2];
            const float ss = red[tid * 2 + 1] + red[(BN + tid) * 2 + 1];
            p.stats[((size_t)mtile * 2 + 0) * p.K + n0 + tid] = s;
            p.stats[((size_t)mtile * 2 + 1) * p.K + n0 + tid] = ss;
        }
    }

    // ---- epilogue: scale/shift, activation, optional accumulate, store
    OT* __restrict__ y = (OT*)p.y;
    float sc[NT], sh[NT];
    int kcol[NT];
#pragma unroll
    for (int b = 0; b < NT; ++b) {
        kcol[b] = n0 + wn * (BN / 2) + b * 16 + fr;
        const bool okk = kcol[b] < p.K;
        sc[b] = (p.scale && okk) ? p.scale[kcol[b]] : 1.0f;
        sh[b] = (p.shift && okk) ? p.shift[kcol[b]] : 0.0f;
    }
#pragma unroll
    for (int a = 0; a < MT; ++a) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + wm * (BM / 2) + a * 16 + fq * 4 + r;
            if (m >= p.M) continue;
            size_t opix;
            if (p.dense_out) {
                opix = (size_t)m;
            } else {
                const int n = m / HoWo, rem = m - n * HoWo;
                const int oi = rem / p.Wo, oj = rem - oi * p.Wo;
                opix = ((size_t)n * p.Hout + (p.oh_off + oi * p.oh_mul)) * p.Wout + (p.ow_off + oj * p.ow_mul);
            }
            OT* yrow = y + opix * p.ldy;
            const OT* rrow = p.res ? (const OT*)p.res + opix * p.ldr : nullptr;
#pragma unroll
            for (int b = 0; b < NT; ++b) {
                if (kcol[b] >= p.K) continue;
                float v = acc[a][b][r] * sc[b] + sh[b];
                if (p.act == 1) v = silu_f(v);
                if (rrow) v += to_f32<OT>(rrow[kcol[b]]);
                if (p.accumulate) v += to_f32<OT>(yrow[kcol[b]]);
                yrow[kcol[b]] = from_f32<OT>(v);
            }
        }
    }
}

template <typename T, typename OT, int BN>
int launch(const ConvArgs& a, hipStream_t st) {
    const size_t smem = 2 * (BM * 128 + BN * 128);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)conv_igemm_kernel<T, OT, BN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        attr_set = true;
    }
    const int grid = a.mtiles * a.ntiles;
    hipLaunchKernelGGL((conv_igemm_kernel<T, OT, BN>), dim3(grid), dim3(256), smem, st, a);
    HDY_LAUNCH_CHECK("conv_igemm");
    return HDY_OK;
}

template <typename T, typename OT>
int launch_bn(const ConvArgs& a, hipStream_t st) {
    switch (a.bn) {
        case 32: return launch<T, OT, 32>(a, st);
        case 64: return launch<T, OT, 64>(a, st);
        default: return launch<T, OT, 128>(a, st);
    }
}

}  // namespace

int hdy_conv_bn_tile(int K) { return K <= 32 ? 32 : (K <= 64 ? 64 : 128); }

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
    HDY_ARG((long long)a.N * a.Hin * a.Win < (1LL << 31) && (long long)a.N * a.Ho * a.Wo < (1LL << 31), "conv: too many pixels");
    a.Kd = a.TH * a.TW * a.C;
    const int BKE = 8 * VE;
    a.bn = hdy_conv_bn_tile(a.K);
    HDY_ARG(a.Kdp == round_up(a.Kd, BKE), "conv: packed weight pitch %d != %d", a.Kdp, round_up(a.Kd, BKE));
    a.M = a.N * a.Ho * a.Wo;
    a.mtiles = cdiv(a.M, BM);
    a.ntiles = cdiv(a.K, a.bn);
    if (a.dense_out) HDY_ARG(a.oh_mul == 1 && a.ow_mul == 1 && a.oh_off == 0 && a.ow_off == 0 && a.Hout == a.Ho && a.Wout == a.Wo, "conv: dense_out geometry mismatch");
    if (dtype == HDY_BF16) return out_f32 ? launch_bn<bf16_t, float>(a, st) : launch_bn<bf16_t, bf16_t>(a, st);
    return launch_bn<float, float>(a, st);
}

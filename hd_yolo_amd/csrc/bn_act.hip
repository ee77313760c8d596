// BatchNorm (train / eval) + SiLU (+ residual) forward and backward around the conv kernels, gfx950.
//
// All of these are HBM-bound streaming kernels over NHWC tensors with explicit pixel pitches: one 16-byte
// vector (8 bf16 / 4 f32 channels) per lane per access, fp32 math.  Thread mapping: a lane owns ONE channel
// vector for its whole life (vc = tid % vectors_per_row, computed once in 32-bit), keeps that vector's
// per-channel coefficients in registers and walks rows with a grid stride — no per-element integer division,
// no per-element coefficient loads; consecutive lanes read consecutive 16-byte vectors of a row (coalesced).
//
//   forward (train):  conv writes raw y and per-tile (sum, sumsq) slabs  ->  bn_finalize (deterministic slab
//                     reduction in fp64, running-stat update, scale/shift)  ->  bn_act_fwd: z = silu(y*scale+shift) [+ res]
//   backward:         bn_act_bwd_reduce: per-channel partial sums of du = dz*silu'(u) and du*xhat
//                     ->  bn_bwd_finalize: dgamma, dbeta, c1 = dbeta/M, c2 = dgamma/M
//                     ->  bn_act_bwd_apply: dy = scale*(du - c1 - xhat*c2)        (recomputes du, xhat from dz, y)
//
// Measured and NOT adopted (round 1): a single cooperative launch for the backward of the <= 27 M-element layers that keeps the
// (dz, y) vectors in registers + LDS between the reduce and the apply phase (2 reads + 1 write of HBM instead of 4 + 1).  It was
// correct but slower: 73 us minimum per launch (two cooperative-groups grid syncs over 256 workgroups) against 30-63 us for the
// three launches, and a grid that must own every CU serialises against the weight-gradient stream (step 17.6 -> 20.9 ms).
//
// Reference semantics replaced: nn.BatchNorm2d (eps 1e-3, momentum 0.03: metayolo/models/utils_torch.py:47-49)
// and nn.SiLU inside Conv.forward (metayolo/models/layers.py:37-38), the Bottleneck residual add (:97),
// and their autograd backward.
#include <type_traits>

#include "common.h"
#include "hdyolo.h"

namespace {

template <typename T> struct VT;
template <> struct VT<float> { static constexpr int VE = 4; };
template <> struct VT<bf16_t> { static constexpr int VE = 8; };

template <typename T> __device__ __forceinline__ void unpack(const i32x4& v, float* f);
template <> __device__ __forceinline__ void unpack<float>(const i32x4& v, float* f) {
    V16 u; u.i = v;
#pragma unroll
    for (int i = 0; i < 4; ++i) f[i] = u.f[i];
}
template <> __device__ __forceinline__ void unpack<bf16_t>(const i32x4& v, float* f) {
    V16 u; u.i = v;
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = (float)u.h[i];
}
template <typename T> __device__ __forceinline__ i32x4 pack(const float* f);
template <> __device__ __forceinline__ i32x4 pack<float>(const float* f) {
    V16 u;
#pragma unroll
    for (int i = 0; i < 4; ++i) u.f[i] = f[i];
    return u.i;
}
template <> __device__ __forceinline__ i32x4 pack<bf16_t>(const float* f) {
    V16 u;
#pragma unroll
    for (int i = 0; i < 8; ++i) u.h[i] = (bf16_t)f[i];
    return u.i;
}

// v_exp_f32 + v_rcp_f32 (1 ulp): plenty for activations and far cheaper than the IEEE division sequence
__device__ __forceinline__ float fast_sigmoid(float u) { return __builtin_amdgcn_rcpf(1.0f + __expf(-u)); }
__device__ __forceinline__ float fast_silu(float u) { return u * fast_sigmoid(u); }
__device__ __forceinline__ float dsilu_f(float u) {
    const float s = fast_sigmoid(u);
    return s * (1.0f + u * (1.0f - s));
}

// lane -> (channel vector, row lane) for a 256-thread block over rows of VCt vectors
struct Lane {
    int vc, rl, RL;
    bool live;
};
__device__ __forceinline__ Lane lane_map(int VCt, int chunk) {
    Lane L;
    const int VC = min(256, VCt - chunk * 256);
    L.RL = 256 / VC;
    L.vc = chunk * 256 + (int)(threadIdx.x % (unsigned)VC);
    L.rl = (int)(threadIdx.x / (unsigned)VC);
    L.live = L.rl < L.RL;
    return L;
}

// ---------------------------------------------------------------- finalize (forward)
// Stage A (only for many tiles): grid (ceil(K/32), G): partial fp64 sums of a range of tiles -> part[g][2][K]
__global__ __launch_bounds__(1024) void bn_partial_kernel(const float* __restrict__ stats, int stats_ld, int mtiles, int K, int tiles_per_group,
                                                          double* __restrict__ part, double* __restrict__ count_out = nullptr, double count = 0.0) {
    if (count_out && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *count_out = count;      // SyncBatchNorm: the element count travels with the sums
    __shared__ double red[2][32][33];
    const int cl = threadIdx.x & 31, tl = threadIdx.x >> 5;
    const int k = blockIdx.x * 32 + cl;
    const int t0 = blockIdx.y * tiles_per_group, t1 = min(t0 + tiles_per_group, mtiles);
    double s = 0.0, ss = 0.0;
    if (k < K) {
        for (int t = t0 + tl; t < t1; t += 32) {
            s += (double)stats[((size_t)t * 2 + 0) * stats_ld + k];
            ss += (double)stats[((size_t)t * 2 + 1) * stats_ld + k];
        }
    }
    red[0][tl][cl] = s;
    red[1][tl][cl] = ss;
    __syncthreads();
    if (tl == 0 && k < K) {
        s = 0.0; ss = 0.0;
        for (int t = 0; t < 32; ++t) { s += red[0][t][cl]; ss += red[1][t][cl]; }
        part[((size_t)blockIdx.y * 2 + 0) * K + k] = s;
        part[((size_t)blockIdx.y * 2 + 1) * K + k] = ss;
    }
}

// Learnable parameters and running statistics of one BatchNorm, or of the two BatchNorms of a C3's merged cv1 | cv2 convolution:
// channels [0, Ka) belong to the first module, [Ka, K) to the second (Ka == K: a single module, the *_b pointers are unused).
// BatchNorm works per channel, so the pair runs through every pass as ONE K-wide layer; only the places that touch module-owned
// tensors (parameters, running statistics, parameter gradients, the two destinations / gradient sources) split at Ka.
struct BnParams {
    const float* gamma; const float* beta; float* rmean; float* rvar;
    const float* gamma_b; const float* beta_b; float* rmean_b; float* rvar_b;
    int Ka;
};

// Final stage.  Input is either the fp32 slabs (ST = float) or stage A's fp64 partials (ST = double).
// grid = ceil(K/8), block = 8 channels x 32 tile-lanes.  (32 x 32 = 1024-thread workgroups were the first shape: a workgroup that needs 16 wave
// slots and 17 KB of LDS on ONE CU waits whenever the weight-gradient stream's kernels fill the CUs — 35-47 us for a 5 us kernel.)
constexpr int FCL = 8, FTL = 32;
template <typename ST>
__global__ __launch_bounds__(256) void bn_finalize_kernel(const ST* __restrict__ stats, int stats_ld, int mtiles, int K, double count,
                                                          BnParams bn, float eps, float momentum,
                                                          float* __restrict__ scale, float* __restrict__ shift,
                                                          float* __restrict__ save_mean, float* __restrict__ save_invstd,
                                                          const double* __restrict__ count_dev = nullptr) {
    __shared__ double red[2][FTL][FCL + 1];
    const int cl = threadIdx.x % FCL, tl = threadIdx.x / FCL;
    const int k = blockIdx.x * FCL + cl;
    if (count_dev) count = *count_dev;                   // SyncBatchNorm: the all-reduced element count
    double s = 0.0, ss = 0.0;
    // the whole kernel is one chain of load latencies (each a round trip to another XCD's slabs): the parameters the last stage needs are
    // requested first, and the slabs in batches of 8 per lane — 16 loads in flight, out-of-range slots clamped and weighted 0 so that the
    // batch has no branch (a tail loop of single loads cost a round trip per slab: 3-4 of them for the usual 100-250 slabs)
    const bool fin = tl == 0 && k < K;
    const bool second = k >= bn.Ka;                      // channels of the second BatchNorm of a pair
    const int kk = second ? k - bn.Ka : k;
    float* const rmean = second ? bn.rmean_b : bn.rmean;
    float* const rvar = second ? bn.rvar_b : bn.rvar;
    float p_gamma = 0.f, p_beta = 0.f, p_rmean = 0.f, p_rvar = 0.f;
    if (fin) {
        p_gamma = (second ? bn.gamma_b : bn.gamma)[kk];
        p_beta = (second ? bn.beta_b : bn.beta)[kk];
        if (rmean) { p_rmean = rmean[kk]; p_rvar = rvar[kk]; }
    }
    if (k < K) {
        for (int t = tl; t < mtiles; t += 8 * FTL) {
            ST a[8], b[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int tt = min(t + FTL * u, mtiles - 1);
                a[u] = stats[((size_t)tt * 2 + 0) * stats_ld + k];
                b[u] = stats[((size_t)tt * 2 + 1) * stats_ld + k];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const bool live = t + FTL * u < mtiles;
                s += live ? (double)a[u] : 0.0;
                ss += live ? (double)b[u] : 0.0;
            }
        }
    }
    red[0][tl][cl] = s;
    red[1][tl][cl] = ss;
    __syncthreads();
    if (fin) {
        s = 0.0; ss = 0.0;
        for (int t = 0; t < FTL; ++t) { s += red[0][t][cl]; ss += red[1][t][cl]; }
        const double mean = s / count;
        double var = ss / count - mean * mean;
        if (var < 0.0) var = 0.0;
        const float invstd = (float)(1.0 / sqrt(var + (double)eps));
        const float sc = p_gamma * invstd;
        scale[k] = sc;
        shift[k] = p_beta - (float)mean * sc;
        save_mean[k] = (float)mean;
        save_invstd[k] = invstd;
        if (rmean) {
            const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
            rmean[kk] = (1.0f - momentum) * p_rmean + momentum * (float)mean;
            rvar[kk] = (1.0f - momentum) * p_rvar + momentum * (float)unbiased;
        }
    }
}

__global__ void bn_eval_coeffs_kernel(const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ rmean,
                                      const float* __restrict__ rvar, float eps, int K, float* __restrict__ scale, float* __restrict__ shift) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    const float sc = gamma[k] / sqrtf(rvar[k] + eps);
    scale[k] = sc;
    shift[k] = beta[k] - rmean[k] * sc;
}

__global__ __launch_bounds__(256) void bn_eval_coeffs_batch_kernel(const hdy_bn_eval_desc* __restrict__ table) {
    const hdy_bn_eval_desc d = table[blockIdx.x];           // one workgroup per BatchNorm
    for (int k = threadIdx.x; k < d.K; k += 256) {
        const float sc = d.gamma[k] / sqrtf(d.running_var[k] + d.eps);
        d.scale[k] = sc;
        d.shift[k] = d.beta[k] - d.running_mean[k] * sc;
    }
}

// ---------------------------------------------------------------- forward apply
// ACT (0 none, 1 SiLU, 2 ReLU) and RES are template flags: as runtime selects inside the element loop they cost ~0.6 ms per step
template <typename T, int ACT, bool RES>
__global__ __launch_bounds__(256) void bn_act_fwd_kernel(const T* __restrict__ y, int ldy, const float* __restrict__ scale,
                                                         const float* __restrict__ shift, const T* __restrict__ res, int ldr,
                                                         T* __restrict__ z, int ldz, T* __restrict__ z_b, int ldz_b, int Ka, int M, int K,
                                                         int act) {
    constexpr int VE = VT<T>::VE;
    const int VCt = K / VE;
    for (int chunk = 0; chunk * 256 < VCt; ++chunk) {
        const Lane L = lane_map(VCt, chunk);
        if (!L.live) continue;
        const int c = L.vc * VE;
        float sc[VE], sh[VE];
#pragma unroll
        for (int i = 0; i < VE; ++i) { sc[i] = scale[c + i]; sh[i] = shift[c + i]; }
        T* const zc = c < Ka ? z + c : z_b + (c - Ka);    // a lane's channels are fixed: the pair's second destination is a pointer choice
        const int ldzc = c < Ka ? ldz : ldz_b;
        for (int m = blockIdx.x * L.RL + L.rl; m < M; m += gridDim.x * L.RL) {
            float v[VE], r[VE];
            unpack<T>(*(const i32x4*)(y + (size_t)m * ldy + c), v);
            if (RES) unpack<T>(*(const i32x4*)(res + (size_t)m * ldr + c), r);
#pragma unroll
            for (int i = 0; i < VE; ++i) {
                float u = v[i] * sc[i] + sh[i];
                if (ACT == 1) u = fast_silu(u);
                else if (ACT == 2) u = fmaxf(u, 0.0f);
                if (RES) u += r[i];
                v[i] = u;
            }
            *(i32x4*)(zc + (size_t)m * ldzc) = pack<T>(v);
        }
    }
}

// ---------------------------------------------------------------- backward
// partial[block][2][K]: sum(du), sum(du*xhat) over this block's rows.  grid.x = row blocks, grid.y = column chunks.
// MODE 0: plain column sums of dz (colsum: y and the coefficients are null); 1: BatchNorm backward without activation; 2: with SiLU
template <typename T, int MODE>
__global__ __launch_bounds__(256) void bn_act_bwd_reduce_kernel(const T* __restrict__ dz, int lddz, const T* __restrict__ dz_b, int lddz_b, int Ka,
                                                                const T* __restrict__ y, int ldy,
                                                                const float* __restrict__ scale, const float* __restrict__ shift,
                                                                const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                int M, int K, int act, int rows_per_block, float* __restrict__ partial) {
    constexpr int VE = VT<T>::VE;
    __shared__ float red[256 * VE];                      // one sum at a time (8 KB, not 16: the weight-gradient stream's kernels leave 32 KB or less per CU)
    const int VCt = K / VE;
    const Lane L = lane_map(VCt, blockIdx.y);
    const int VC = min(256, VCt - blockIdx.y * 256);
    const int c = L.vc * VE;
    float a1[VE], a2[VE], sc[VE], sh[VE], mu[VE], is[VE];
#pragma unroll
    for (int i = 0; i < VE; ++i) {
        a1[i] = 0.f; a2[i] = 0.f;
        sc[i] = (MODE != 0 && L.live) ? scale[c + i] : 1.f;
        sh[i] = (MODE != 0 && L.live) ? shift[c + i] : 0.f;
        mu[i] = (MODE != 0 && L.live) ? mean[c + i] : 0.f;
        is[i] = (MODE != 0 && L.live) ? invstd[c + i] : 0.f;
    }
    const int mbeg = blockIdx.x * rows_per_block;
    const int mend = min(mbeg + rows_per_block, M);
    const T* const dzc = c < Ka ? dz + c : dz_b + (c - Ka);        // gradient source of this lane's channels (second module of a pair)
    const int lddzc = c < Ka ? lddz : lddz_b;
    if (L.live) {
        auto accumulate = [&](const i32x4& gq, const i32x4& vq) {
            float g[VE], v[VE] = {};
            unpack<T>(gq, g);
            if (MODE != 0) unpack<T>(vq, v);
#pragma unroll
            for (int i = 0; i < VE; ++i) {
                float du = g[i];
                if (MODE == 2) du *= dsilu_f(v[i] * sc[i] + sh[i]);
                a1[i] += du;
                if (MODE != 0) a2[i] += du * ((v[i] - mu[i]) * is[i]);
            }
        };
        // 4 rows per trip: 8 independent 16-byte loads in flight per lane.  With one pair per trip the pass was latency bound (the y
        // operand was written a whole forward pass ago and comes from HBM): -11 % on this kernel, -0.27 ms per bench step (same box).
        // Cutting wide-K / small-M tensors into column chunks for more workgroups in flight was measured too: no change.
        constexpr int U = 4;
        int m = mbeg + L.rl;
        for (; m + (U - 1) * L.RL < mend; m += U * L.RL) {
            i32x4 gq[U], vq[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                gq[u] = *(const i32x4*)(dzc + (size_t)(m + u * L.RL) * lddzc);
                vq[u] = MODE != 0 ? *(const i32x4*)(y + (size_t)(m + u * L.RL) * ldy + c) : i32x4{0, 0, 0, 0};
            }
#pragma unroll
            for (int u = 0; u < U; ++u) accumulate(gq[u], vq[u]);
        }
        for (; m < mend; m += L.RL) {
            const i32x4 gq = *(const i32x4*)(dzc + (size_t)m * lddzc);
            const i32x4 vq = MODE != 0 ? *(const i32x4*)(y + (size_t)m * ldy + c) : i32x4{0, 0, 0, 0};
            accumulate(gq, vq);
        }
    }
#pragma unroll
    for (int which = 0; which < 2; ++which) {
        if (L.live) {
#pragma unroll
            for (int i = 0; i < VE; ++i) red[(L.rl * VC + (L.vc - blockIdx.y * 256)) * VE + i] = which ? a2[i] : a1[i];
        }
        __syncthreads();
        for (int j = threadIdx.x; j < VC * VE; j += 256) {
            float s = 0.f;
            for (int r = 0; r < L.RL; ++r) s += red[r * VC * VE + j];
            partial[((size_t)blockIdx.x * 2 + which) * K + blockIdx.y * 256 * VE + j] = s;
        }
        __syncthreads();
    }
}

// The same pass with FOUR channels per lane (bf16: 8-byte loads) — round 4.  The kernel above needs 120 VGPRs (8 channels x 4 coefficients, 16 sums, 8 rows
// in flight) and shares the chip with the weight-gradient stream: beside wgrad_kernel<2, 2> (2 waves x 152 registers per SIMD) ONE such wave fits per SIMD,
// beside wgrad3x3_kernel (3 x 144) and wgrad_deep_kernel (2 x 209) none — in the step the pass runs at 1.9 TB/s where it reaches 3.2-5 alone
// (profiles/r04_step_kernel_stats.txt: 2.57 ms for 4.95 GB).  Half the channels per lane is half the coefficients and sums: ~60 registers, three waves per SIMD
// beside the generic weight gradient, one beside the 3x3 one.  Same slab layout, same finalize.
template <int MODE, int VE, int U>
__global__ __launch_bounds__(256) void bn_act_bwd_reduce4_kernel(const bf16_t* __restrict__ dz, int lddz, const bf16_t* __restrict__ dz_b, int lddz_b, int Ka,
                                                                 const bf16_t* __restrict__ y, int ldy,
                                                                 const float* __restrict__ scale, const float* __restrict__ shift,
                                                                 const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                 int M, int K, int rows_per_block, float* __restrict__ partial) {
    static_assert(VE == 4 || VE == 2, "channels per lane");
    typedef int ivec __attribute__((ext_vector_type(VE / 2)));       // VE bf16 values
    __shared__ float red[256 * VE];
    const int VCt = K / VE;
    const Lane L = lane_map(VCt, blockIdx.y);
    const int VC = min(256, VCt - blockIdx.y * 256);
    const int c = L.vc * VE;
    float a1[VE], a2[VE], sc[VE], sh[VE], is[VE], nm[VE];          // xhat = y * invstd + (-mean * invstd)
#pragma unroll
    for (int i = 0; i < VE; ++i) {
        a1[i] = 0.f; a2[i] = 0.f;
        sc[i] = L.live ? scale[c + i] : 1.f;
        sh[i] = L.live ? shift[c + i] : 0.f;
        is[i] = L.live ? invstd[c + i] : 0.f;
        nm[i] = L.live ? -mean[c + i] * is[i] : 0.f;
    }
    const int mbeg = blockIdx.x * rows_per_block;
    const int mend = min(mbeg + rows_per_block, M);
    const bf16_t* const dzc = c < Ka ? dz + c : dz_b + (c - Ka);
    const int lddzc = c < Ka ? lddz : lddz_b;
    if (L.live) {
        auto accumulate = [&](const ivec& gq, const ivec& vq) {
#pragma unroll
            for (int h = 0; h < VE / 2; ++h) {
                int gw, vw;
                gw = gq[h]; vw = vq[h];
                const float g[2] = {__uint_as_float((unsigned)gw << 16), __uint_as_float((unsigned)gw & 0xFFFF0000u)};
                const float v[2] = {__uint_as_float((unsigned)vw << 16), __uint_as_float((unsigned)vw & 0xFFFF0000u)};
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int i = 2 * h + e;
                    float du = g[e];
                    if (MODE == 2) du *= dsilu_f(v[e] * sc[i] + sh[i]);
                    a1[i] += du;
                    a2[i] += du * (v[e] * is[i] + nm[i]);
                }
            }
        };
        int m = mbeg + L.rl;
        for (; m + (U - 1) * L.RL < mend; m += U * L.RL) {
            ivec gq[U], vq[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                gq[u] = *(const ivec*)(dzc + (size_t)(m + u * L.RL) * lddzc);
                vq[u] = *(const ivec*)(y + (size_t)(m + u * L.RL) * ldy + c);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) accumulate(gq[u], vq[u]);
        }
        for (; m < mend; m += L.RL) {
            const ivec gq = *(const ivec*)(dzc + (size_t)m * lddzc);
            const ivec vq = *(const ivec*)(y + (size_t)m * ldy + c);
            accumulate(gq, vq);
        }
    }
#pragma unroll
    for (int which = 0; which < 2; ++which) {
        if (L.live) {
#pragma unroll
            for (int i = 0; i < VE; ++i) red[(L.rl * VC + (L.vc - blockIdx.y * 256)) * VE + i] = which ? a2[i] : a1[i];
        }
        __syncthreads();
        for (int j = threadIdx.x; j < VC * VE; j += 256) {
            float s = 0.f;
            for (int r = 0; r < L.RL; ++r) s += red[r * VC * VE + j];
            partial[((size_t)blockIdx.x * 2 + which) * K + blockIdx.y * 256 * VE + j] = s;
        }
        __syncthreads();
    }
}

// dbeta/dgamma (+)=, c1 = dbeta/M, c2 = dgamma/M
// (Measured and not adopted: dropping this launch by letting the reduce pass add its sums into fp64 accumulators with
// global_atomic_add_f64 and deriving c1 / c2 in the apply pass — up to 1024 device-scope atomics per address cost ~100 us per
// layer: 17.1 -> 22.9 ms per yolov5s bench step.)
// CL channels x TL slab lanes per workgroup: 8 x 32 for the usual few hundred slabs (small workgroups, see bn_finalize_kernel), 8 x 128 for
// the slab arrays of the fused 1x1 backward kernel (one slab per workgroup of a 3 000 - 6 000 workgroup launch).
template <int CL, int TL>
__global__ __launch_bounds__(CL * TL) void bn_bwd_finalize_kernel(const float* __restrict__ partial, int nblocks, int K, double count,
                                                               float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                               float* __restrict__ dgamma_b, float* __restrict__ dbeta_b, int Ka, int accumulate,
                                                               float* __restrict__ c1, float* __restrict__ c2,
                                                               const float* __restrict__ mean = nullptr, const float* __restrict__ invstd = nullptr) {
    __shared__ double red[2][TL][CL + 1];
    const int cl = threadIdx.x % CL, tl = threadIdx.x / CL;
    const int k = blockIdx.x * CL + cl;
    double s1 = 0.0, s2 = 0.0;
    const bool fin = tl == 0 && k < K;
    float p_mean = 0.f, p_invstd = 0.f;
    if (fin && mean) { p_mean = mean[k]; p_invstd = invstd[k]; }
    if (k < K) {
        for (int t = tl; t < nblocks; t += 8 * TL) {          // 16 independent loads in flight per lane, no tail loop (see bn_finalize_kernel)
            float a[8], b[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int tt = min(t + TL * u, nblocks - 1);
                a[u] = partial[((size_t)tt * 2 + 0) * K + k];
                b[u] = partial[((size_t)tt * 2 + 1) * K + k];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const bool live = t + TL * u < nblocks;
                s1 += live ? (double)a[u] : 0.0;
                s2 += live ? (double)b[u] : 0.0;
            }
        }
    }
    red[0][tl][cl] = s1;
    red[1][tl][cl] = s2;
    __syncthreads();
    if (fin) {
        s1 = 0.0; s2 = 0.0;
        for (int t = 0; t < TL; ++t) { s1 += red[0][t][cl]; s2 += red[1][t][cl]; }
        if (mean) s2 = (double)p_invstd * (s2 - (double)p_mean * s1);      // slabs of (SUM du, SUM du*y) from a producer's epilogue
        float* const db = k < Ka ? dbeta : dbeta_b;      // parameter gradients of the pair's second module
        float* const dg = k < Ka ? dgamma : dgamma_b;
        const int kk = k < Ka ? k : k - Ka;
        if (db) db[kk] = accumulate ? db[kk] + (float)s1 : (float)s1;
        if (dg) dg[kk] = accumulate ? dg[kk] + (float)s2 : (float)s2;
        if (c1) c1[k] = (float)(s1 / count);
        if (c2) c2[k] = (float)(s2 / count);
    }
}

static void bn_bwd_finalize_launch(hipStream_t st, const float* partial, int nblocks, int K, double count, float* dgamma, float* dbeta, float* dgamma_b,
                                   float* dbeta_b, int Ka, int accumulate, float* c1, float* c2, const float* mean = nullptr,
                                   const float* invstd = nullptr) {
    if (nblocks >= 1024)
        hipLaunchKernelGGL((bn_bwd_finalize_kernel<8, 128>), dim3(cdiv(K, 8)), dim3(1024), 0, st, partial, nblocks, K, count, dgamma, dbeta, dgamma_b, dbeta_b, Ka,
                           accumulate, c1, c2, mean, invstd);
    else
        hipLaunchKernelGGL((bn_bwd_finalize_kernel<8, 32>), dim3(cdiv(K, 8)), dim3(256), 0, st, partial, nblocks, K, count, dgamma, dbeta, dgamma_b, dbeta_b, Ka,
                           accumulate, c1, c2, mean, invstd);
}

// FROZEN: constant scale / shift (FrozenBatchNorm2d): dy = scale * du, no statistics terms (mean / invstd / c1 / c2 unused)
template <typename T, bool FROZEN, int ACT>
__global__ __launch_bounds__(256) void bn_act_bwd_apply_kernel(const T* __restrict__ dz, int lddz, const T* __restrict__ dz_b, int lddz_b, int Ka,
                                                               const T* __restrict__ y, int ldy,
                                                               const float* __restrict__ scale, const float* __restrict__ shift,
                                                               const float* __restrict__ mean, const float* __restrict__ invstd,
                                                               const float* __restrict__ c1, const float* __restrict__ c2,
                                                               T* __restrict__ dy, int lddy, int M, int K, int act) {
    constexpr int VE = VT<T>::VE;
    const int VCt = K / VE;
    for (int chunk = 0; chunk * 256 < VCt; ++chunk) {
        const Lane L = lane_map(VCt, chunk);
        if (!L.live) continue;
        const int c = L.vc * VE;
        float sc[VE], sh[VE], mu[VE], is[VE], k1[VE], k2[VE];
#pragma unroll
        for (int i = 0; i < VE; ++i) {
            sc[i] = scale[c + i]; sh[i] = shift[c + i];
            mu[i] = FROZEN ? 0.f : mean[c + i]; is[i] = FROZEN ? 0.f : invstd[c + i];
            k1[i] = FROZEN ? 0.f : c1[c + i]; k2[i] = FROZEN ? 0.f : c2[c + i];
        }
        const T* const dzc = c < Ka ? dz + c : dz_b + (c - Ka);
        const int lddzc = c < Ka ? lddz : lddz_b;
        for (int m = blockIdx.x * L.RL + L.rl; m < M; m += gridDim.x * L.RL) {
            float g[VE], v[VE];
            unpack<T>(*(const i32x4*)(dzc + (size_t)m * lddzc), g);
            unpack<T>(*(const i32x4*)(y + (size_t)m * ldy + c), v);
#pragma unroll
            for (int i = 0; i < VE; ++i) {
                float du = g[i];
                if (ACT == 1) du *= dsilu_f(v[i] * sc[i] + sh[i]);
                const float xh = (v[i] - mu[i]) * is[i];
                g[i] = sc[i] * (du - k1[i] - xh * k2[i]);
            }
            *(i32x4*)(dy + (size_t)m * lddy + c) = pack<T>(g);
        }
    }
}

// out[m][c] (+)= a[m][c]  -- gradient accumulation between pitched NHWC views
template <typename T>
__global__ __launch_bounds__(256) void add_inplace_kernel(T* __restrict__ out, int ldo, const T* __restrict__ a, int lda, int M, int K) {
    constexpr int VE = VT<T>::VE;
    const int VCt = K / VE;
    for (int chunk = 0; chunk * 256 < VCt; ++chunk) {
        const Lane L = lane_map(VCt, chunk);
        if (!L.live) continue;
        const int c = L.vc * VE;
        for (int m = blockIdx.x * L.RL + L.rl; m < M; m += gridDim.x * L.RL) {
            float o[VE], v[VE];
            unpack<T>(*(const i32x4*)(out + (size_t)m * ldo + c), o);
            unpack<T>(*(const i32x4*)(a + (size_t)m * lda + c), v);
#pragma unroll
            for (int i = 0; i < VE; ++i) o[i] += v[i];
            *(i32x4*)(out + (size_t)m * ldo + c) = pack<T>(o);
        }
    }
}

// rows each block-iteration covers = 256 / min(256, VC); enough blocks for ~8 per CU, grid-stride the rest
inline int stream_grid(long long M, int VC) {
    const int rl = 256 / (VC < 256 ? VC : 256);
    long long g = (M + rl - 1) / rl;
    if (g > 256 * 8) g = 256 * 8;
    if (g < 1) g = 1;
    return (int)g;
}

// a tensor that may come in two channel ranges: [0, Ka) from (a, lda), [Ka, K) from (b, ldb)  (Ka == K: b unused)
struct Split {
    const void* a; int lda;
    const void* b; int ldb;
    int Ka;
};

template <typename T>
void bn_bwd_reduce_launch(dim3 grid, hipStream_t st, const Split& dz, const void* y, int ldy, const float* scale, const float* shift,
                          const float* mean, const float* invstd, int M, int K, int act, int rows, float* partial) {
    if constexpr (std::is_same<T, bf16_t>::value) {
        // four channels per lane (see bn_act_bwd_reduce4_kernel): every BatchNorm of the path (K % 8 == 0, 8-element pitches); HDY_NO_BN_REDUCE4 keeps the
        // eight-channel form (A/B)
        if (y && !hdy_opt(HDY_OPT_NO_BN_REDUCE4) && dz.Ka % 4 == 0) {
            const dim3 g4(grid.x, cdiv(K / 4, 256));
            // rows in flight per lane: 4 (72 VGPRs) — 3 (60) and 2 (54) fit one more wave per SIMD beside the generic weight gradient and measured the same /
            // 0.1 ms slower in the step (11.94-11.96 | 11.92-11.99 | 12.06-12.08 ms; 8 channels per lane: 12.19-12.24).  A buffer-load form (one descriptor per
            // block, 32-bit lane offsets, scalar row offsets: no 64-bit address per load) with 4 / 6 / 8 rows in flight at 60 / 72 / 90 VGPRs measured
            // 12.07-12.09 / 12.08-12.11 / 12.24-12.28 against 12.02-12.05 ms: more requests in flight per wave do not buy the pass a larger share of the
            // HBM it shares with the weight gradient; removed
            if (act == 1)
                hipLaunchKernelGGL((bn_act_bwd_reduce4_kernel<2, 4, 4>), g4, dim3(256), 0, st, (const bf16_t*)dz.a, dz.lda, (const bf16_t*)dz.b, dz.ldb, dz.Ka,
                                   (const bf16_t*)y, ldy, scale, shift, mean, invstd, M, K, rows, partial);
            else
                hipLaunchKernelGGL((bn_act_bwd_reduce4_kernel<1, 4, 4>), g4, dim3(256), 0, st, (const bf16_t*)dz.a, dz.lda, (const bf16_t*)dz.b, dz.ldb, dz.Ka,
                                   (const bf16_t*)y, ldy, scale, shift, mean, invstd, M, K, rows, partial);
            return;
        }
    }
    if (act == 1)
        hipLaunchKernelGGL((bn_act_bwd_reduce_kernel<T, 2>), grid, dim3(256), 0, st, (const T*)dz.a, dz.lda, (const T*)dz.b, dz.ldb, dz.Ka, (const T*)y, ldy,
                           scale, shift, mean, invstd, M, K, act, rows, partial);
    else if (y)
        hipLaunchKernelGGL((bn_act_bwd_reduce_kernel<T, 1>), grid, dim3(256), 0, st, (const T*)dz.a, dz.lda, (const T*)dz.b, dz.ldb, dz.Ka, (const T*)y, ldy,
                           scale, shift, mean, invstd, M, K, act, rows, partial);
    else
        hipLaunchKernelGGL((bn_act_bwd_reduce_kernel<T, 0>), grid, dim3(256), 0, st, (const T*)dz.a, dz.lda, (const T*)dz.b, dz.ldb, dz.Ka, (const T*)y, ldy,
                           scale, shift, mean, invstd, M, K, act, rows, partial);
}

template <typename T, bool FROZEN>
void bn_bwd_apply_launch(int grid, hipStream_t st, const Split& dz, const void* y, int ldy, const float* scale, const float* shift,
                         const float* mean, const float* invstd, const float* c1, const float* c2, void* dy, int lddy, int M, int K, int act) {
    if (act == 1)
        hipLaunchKernelGGL((bn_act_bwd_apply_kernel<T, FROZEN, 1>), dim3(grid), dim3(256), 0, st, (const T*)dz.a, dz.lda, (const T*)dz.b, dz.ldb, dz.Ka,
                           (const T*)y, ldy, scale, shift, mean, invstd, c1, c2, (T*)dy, lddy, M, K, act);
    else
        hipLaunchKernelGGL((bn_act_bwd_apply_kernel<T, FROZEN, 0>), dim3(grid), dim3(256), 0, st, (const T*)dz.a, dz.lda, (const T*)dz.b, dz.ldb, dz.Ka,
                           (const T*)y, ldy, scale, shift, mean, invstd, c1, c2, (T*)dy, lddy, M, K, act);
}

template <typename T, int ACT>
void bn_act_fwd_launch(int grid, hipStream_t st, const void* y, int ldy, const float* scale, const float* shift, const void* res, int ldr,
                       const Split& z, int M, int K, int act) {
    if (res)
        hipLaunchKernelGGL((bn_act_fwd_kernel<T, ACT, true>), dim3(grid), dim3(256), 0, st, (const T*)y, ldy, scale, shift, (const T*)res, ldr, (T*)z.a,
                           z.lda, (T*)z.b, z.ldb, z.Ka, M, K, act);
    else
        hipLaunchKernelGGL((bn_act_fwd_kernel<T, ACT, false>), dim3(grid), dim3(256), 0, st, (const T*)y, ldy, scale, shift, (const T*)res, ldr, (T*)z.a,
                           z.lda, (T*)z.b, z.ldb, z.Ka, M, K, act);
}

template <typename T>
void bn_act_fwd_dispatch(int grid, hipStream_t st, const void* y, int ldy, const float* scale, const float* shift, const void* res, int ldr,
                         const Split& z, int M, int K, int act) {
    if (act == 1) bn_act_fwd_launch<T, 1>(grid, st, y, ldy, scale, shift, res, ldr, z, M, K, act);
    else if (act == 2) bn_act_fwd_launch<T, 2>(grid, st, y, ldy, scale, shift, res, ldr, z, M, K, act);
    else bn_act_fwd_launch<T, 0>(grid, st, y, ldy, scale, shift, res, ldr, z, M, K, act);
}

}  // namespace

#define VEC_OK(ptr, ld, VE) ((((uintptr_t)(ptr)) & 15) == 0 && (ld) % (VE) == 0)
#define M_OK(M) ((M) > 0 && (M) < (1LL << 31))

extern "C" {

size_t hdy_bn_finalize_workspace_bytes(int mtiles, int K) { return mtiles > 1024 ? (size_t)32 * 2 * K * sizeof(double) : 0; }

static int bn_finalize_impl(const float* stats, int stats_ld, int mtiles, int K, long long count, const BnParams& bn, float eps, float momentum,
                            float* scale, float* shift, float* save_mean, float* save_invstd, void* workspace, size_t ws_bytes, void* stream) {
    HDY_ARG(stats && bn.gamma && bn.beta && scale && shift && save_mean && save_invstd, "bn_finalize: null pointer");
    HDY_ARG(!workspace || ws_bytes >= hdy_bn_finalize_workspace_bytes(mtiles, K), "bn_finalize: workspace of %zu bytes, %zu needed", ws_bytes,
            hdy_bn_finalize_workspace_bytes(mtiles, K));
    HDY_ARG(mtiles > 0 && K > 0 && count > 0 && stats_ld >= K, "bn_finalize: bad sizes");
    HDY_ARG((bn.rmean == nullptr) == (bn.rvar == nullptr), "bn_finalize: running_mean/var must both be given or both null");
    HDY_ARG(bn.Ka == K || (bn.Ka > 0 && bn.Ka < K && bn.gamma_b && bn.beta_b && (bn.rmean_b == nullptr) == (bn.rvar_b == nullptr)),
            "bn_finalize_pair: second module's parameters missing or split point outside (0, K)");
    hipStream_t st = (hipStream_t)stream;
    if (mtiles > 1024 && workspace) {
        // two stages: 32 groups of tiles reduced in parallel, then the usual final stage over 32 fp64 partials
        const int G = 32, tpg = cdiv(mtiles, G);
        double* part = (double*)workspace;
        hipLaunchKernelGGL(bn_partial_kernel, dim3(cdiv(K, 32), G), dim3(1024), 0, st, stats, stats_ld, mtiles, K, tpg, part);
        HDY_LAUNCH_CHECK("bn_partial");
        hipLaunchKernelGGL(bn_finalize_kernel<double>, dim3(cdiv(K, FCL)), dim3(256), 0, st, (const double*)part, K, cdiv(mtiles, tpg), K,
                           (double)count, bn, eps, momentum, scale, shift, save_mean, save_invstd);
    } else {
        hipLaunchKernelGGL(bn_finalize_kernel<float>, dim3(cdiv(K, FCL)), dim3(256), 0, st, stats, stats_ld, mtiles, K, (double)count, bn, eps,
                           momentum, scale, shift, save_mean, save_invstd);
    }
    HDY_LAUNCH_CHECK("bn_finalize");
    return HDY_OK;
}

int hdy_bn_finalize(const float* stats, int stats_ld, int mtiles, int K, long long count, const float* gamma, const float* beta, float* running_mean,
                    float* running_var, float eps, float momentum, float* scale, float* shift, float* save_mean, float* save_invstd,
                    void* workspace, size_t ws_bytes, void* stream) {
    const BnParams bn = {gamma, beta, running_mean, running_var, nullptr, nullptr, nullptr, nullptr, K};
    return bn_finalize_impl(stats, stats_ld, mtiles, K, count, bn, eps, momentum, scale, shift, save_mean, save_invstd, workspace, ws_bytes, stream);
}

int hdy_bn_finalize_pair(const float* stats, int stats_ld, int mtiles, int K, int Ka, long long count, const float* gamma_a, const float* beta_a,
                         float* running_mean_a, float* running_var_a, const float* gamma_b, const float* beta_b, float* running_mean_b,
                         float* running_var_b, float eps, float momentum, float* scale, float* shift, float* save_mean, float* save_invstd,
                         void* workspace, size_t ws_bytes, void* stream) {
    const BnParams bn = {gamma_a, beta_a, running_mean_a, running_var_a, gamma_b, beta_b, running_mean_b, running_var_b, Ka};
    return bn_finalize_impl(stats, stats_ld, mtiles, K, count, bn, eps, momentum, scale, shift, save_mean, save_invstd, workspace, ws_bytes, stream);
}

// ---- SyncBatchNorm (reference: train.py:281-283 converts the model with torch.nn.SyncBatchNorm when --sync-bn): the per-rank slabs are
// summed to fp64 [2][K] + the element count, the caller all-reduces those 2K + 1 doubles, and the finalize reads sums and count from
// device memory.  Backward: the local statistics pass stays as it is (dgamma / dbeta are local, the gradient all-reduce sums them), its
// partial slabs go through hdy_bn_slab_sums + all-reduce and c1 / c2 come from the global sums.
__global__ void bn_bwd_coeffs_sums_kernel(const double* __restrict__ sums, int K, float* __restrict__ c1, float* __restrict__ c2) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    const double count = sums[2 * K];                    // [SUM du | SUM du*xhat | count] over all ranks
    c1[k] = (float)(sums[k] / count);
    c2[k] = (float)(sums[K + k] / count);
}

int hdy_bn_slab_sums(const float* slabs, int slab_ld, int nslabs, int K, long long count, double* sums, void* stream) {
    HDY_ARG(slabs && sums && nslabs > 0 && K > 0 && slab_ld >= K && count > 0, "bn_slab_sums: bad args");
    hipLaunchKernelGGL(bn_partial_kernel, dim3(cdiv(K, 32), 1), dim3(1024), 0, (hipStream_t)stream, slabs, slab_ld, nslabs, K, nslabs, sums, sums + 2 * K,
                       (double)count);
    HDY_LAUNCH_CHECK("bn_slab_sums");
    return HDY_OK;
}

int hdy_bn_finalize_sums(const double* sums, int sums_ld, const double* count, int K, int Ka, const float* gamma_a, const float* beta_a, float* running_mean_a, float* running_var_a,
                         const float* gamma_b, const float* beta_b, float* running_mean_b, float* running_var_b, float eps, float momentum,
                         float* scale, float* shift, float* save_mean, float* save_invstd, void* stream) {
    const BnParams bn = {gamma_a, beta_a, running_mean_a, running_var_a, gamma_b, beta_b, running_mean_b, running_var_b, Ka};
    HDY_ARG(sums && count && gamma_a && beta_a && scale && shift && save_mean && save_invstd && K > 0 && sums_ld >= K, "bn_finalize_sums: bad args");
    HDY_ARG(Ka == K || (Ka > 0 && Ka < K && gamma_b && beta_b), "bn_finalize_sums: second module's parameters missing or split point outside (0, K)");
    hipLaunchKernelGGL(bn_finalize_kernel<double>, dim3(cdiv(K, FCL)), dim3(256), 0, (hipStream_t)stream, sums, sums_ld, 1, K, 1.0, bn, eps, momentum, scale,
                       shift, save_mean, save_invstd, count);
    HDY_LAUNCH_CHECK("bn_finalize_sums");
    return HDY_OK;
}

int hdy_bn_bwd_coeffs_sums(const double* sums, int K, float* c1, float* c2, void* stream) {
    HDY_ARG(sums && c1 && c2 && K > 0, "bn_bwd_coeffs_sums: bad args");
    hipLaunchKernelGGL(bn_bwd_coeffs_sums_kernel, dim3(cdiv(K, 256)), dim3(256), 0, (hipStream_t)stream, sums, K, c1, c2);
    HDY_LAUNCH_CHECK("bn_bwd_coeffs_sums");
    return HDY_OK;
}

int hdy_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean, const float* running_var, float eps, int K,
                       float* scale, float* shift, void* stream) {
    HDY_ARG(gamma && beta && running_mean && running_var && scale && shift && K > 0, "bn_eval_coeffs: bad args");
    hipLaunchKernelGGL(bn_eval_coeffs_kernel, dim3(cdiv(K, 256)), dim3(256), 0, (hipStream_t)stream, gamma, beta, running_mean,
                       running_var, eps, K, scale, shift);
    HDY_LAUNCH_CHECK("bn_eval_coeffs");
    return HDY_OK;
}

int hdy_bn_eval_coeffs_batch(const hdy_bn_eval_desc* descs_device, int ndesc, void* stream) {
    HDY_ARG(descs_device && ndesc > 0, "bn_eval_coeffs_batch: bad args");
    hipLaunchKernelGGL(bn_eval_coeffs_batch_kernel, dim3(ndesc), dim3(256), 0, (hipStream_t)stream, descs_device);
    HDY_LAUNCH_CHECK("bn_eval_coeffs_batch");
    return HDY_OK;
}

static int bn_act_fwd_impl(const void* y, int ldy, const float* scale, const float* shift, const void* res, int ldr, const Split& z, long long M, int K,
                           int act, int dtype, void* stream) {
    const int VE = dtype == HDY_BF16 ? 8 : 4;
    HDY_ARG(y && z.a && scale && shift && M_OK(M) && K > 0, "bn_act_fwd: bad args");
    HDY_ARG(K % VE == 0 && VEC_OK(y, ldy, VE) && VEC_OK(z.a, z.lda, VE) && (!res || VEC_OK(res, ldr, VE)), "bn_act_fwd: K/pitch/alignment must be multiples of one 16-byte vector");
    HDY_ARG(z.Ka == K || (z.Ka > 0 && z.Ka < K && z.Ka % VE == 0 && z.b && VEC_OK(z.b, z.ldb, VE)), "bn_act_fwd_pair: second destination missing / unaligned / bad split");
    HDY_ARG(act >= 0 && act <= 2, "bn_act_fwd: unknown activation %d", act);
    const int grid = stream_grid(M, K / VE);
    if (dtype == HDY_BF16) bn_act_fwd_dispatch<bf16_t>(grid, (hipStream_t)stream, y, ldy, scale, shift, res, ldr, z, (int)M, K, act);
    else bn_act_fwd_dispatch<float>(grid, (hipStream_t)stream, y, ldy, scale, shift, res, ldr, z, (int)M, K, act);
    HDY_LAUNCH_CHECK("bn_act_fwd");
    return HDY_OK;
}

int hdy_bn_act_fwd(const void* y, int ldy, const float* scale, const float* shift, const void* res, int ldr, void* z, int ldz,
                   long long M, int K, int act, int dtype, void* stream) {
    return bn_act_fwd_impl(y, ldy, scale, shift, res, ldr, Split{z, ldz, nullptr, 0, K}, M, K, act, dtype, stream);
}

int hdy_bn_act_fwd_pair(const void* y, int ldy, const float* scale, const float* shift, void* z_a, int ldz_a, void* z_b, int ldz_b, int Ka,
                        long long M, int K, int act, int dtype, void* stream) {
    return bn_act_fwd_impl(y, ldy, scale, shift, nullptr, 0, Split{z_a, ldz_a, z_b, ldz_b, Ka}, M, K, act, dtype, stream);
}

// number of row blocks hdy_bn_act_bwd uses (size of the partial slab = blocks*2*K floats)
int hdy_bn_bwd_blocks(long long M) {
    // large tensors: 256+ rows per block (amortises the per-block coefficient loads and LDS reduction), at most 1024 blocks;
    // small ones: down to 64 rows per block so that a few hundred blocks are in flight (measured on MI355X, both ends)
    long long big = (M + 255) / 256, small = (M + 63) / 64;
    if (big > 1024) big = 1024;
    if (small > 512) small = 512;
    long long b = big > small ? big : small;
    if (b < 1) b = 1;
    return (int)b;
}

size_t hdy_bn_bwd_workspace_bytes(long long M, int K) { return ((size_t)hdy_bn_bwd_blocks(M) * 2 * K + 2 * (size_t)K) * sizeof(float); }
size_t hdy_colsum_workspace_bytes(long long M, int K) { return (size_t)hdy_bn_bwd_blocks(M) * 2 * K * sizeof(float); }

// Full backward of z = act(BN_train(y)) [+ res]:  dy, and dgamma/dbeta (+)=.
// workspace: hdy_bn_bwd_workspace_bytes(M, K) = (hdy_bn_bwd_blocks(M)*2*K + 2*K) floats.
static int bn_act_bwd_impl(const Split& dz, const void* y, int ldy, const float* scale, const float* shift, const float* mean, const float* invstd,
                           void* dy, int lddy, float* dgamma, float* dbeta, float* dgamma_b, float* dbeta_b, int accumulate, long long M, int K,
                           int act, int dtype, float* workspace, size_t ws_bytes, void* stream) {
    const int VE = dtype == HDY_BF16 ? 8 : 4;
    HDY_ARG(dz.a && y && scale && shift && workspace && M_OK(M) && K > 0 && (!mean == !invstd), "bn_act_bwd: bad args");
    HDY_ARG(!mean || ws_bytes >= hdy_bn_bwd_workspace_bytes(M, K), "bn_act_bwd: workspace of %zu bytes, %zu needed", ws_bytes, hdy_bn_bwd_workspace_bytes(M, K));
    HDY_ARG(dy || mean, "bn_act_bwd: dy == NULL (statistics only) needs live BatchNorm statistics");
    HDY_ARG(K % VE == 0 && VEC_OK(dz.a, dz.lda, VE) && VEC_OK(y, ldy, VE) && (!dy || VEC_OK(dy, lddy, VE)), "bn_act_bwd: K/pitch/alignment must be multiples of one 16-byte vector");
    HDY_ARG(dz.Ka == K || (dz.Ka > 0 && dz.Ka < K && dz.Ka % VE == 0 && dz.b && VEC_OK(dz.b, dz.ldb, VE)), "bn_act_bwd_pair: second gradient source missing / unaligned / bad split");
    const int nb = hdy_bn_bwd_blocks(M);
    const int rows = (int)((M + nb - 1) / nb);
    float* partial = workspace;
    float* c1 = workspace + (size_t)nb * 2 * K;
    float* c2 = c1 + K;
    hipStream_t st = (hipStream_t)stream;
    HDY_ARG(act == 0 || act == 1, "bn_act_bwd: activation %d has no BatchNorm backward here", act);
    const int g2 = stream_grid(M, K / VE);
    if (!mean) {
        // frozen BatchNorm (torchvision FrozenBatchNorm2d after Model.freeze): z = act(y*scale + shift) with constant scale/shift,
        // so dy = scale * dz * act'(u) and there is no statistics gradient: the apply pass alone
        if (dtype == HDY_BF16) bn_bwd_apply_launch<bf16_t, true>(g2, st, dz, y, ldy, scale, shift, nullptr, nullptr, nullptr, nullptr, dy, lddy, (int)M, K, act);
        else bn_bwd_apply_launch<float, true>(g2, st, dz, y, ldy, scale, shift, nullptr, nullptr, nullptr, nullptr, dy, lddy, (int)M, K, act);
        HDY_LAUNCH_CHECK("bn_act_bwd_apply(frozen)");
        return HDY_OK;
    }
    dim3 grid(nb, cdiv(K / VE, 256));
    if (dtype == HDY_BF16) bn_bwd_reduce_launch<bf16_t>(grid, st, dz, y, ldy, scale, shift, mean, invstd, (int)M, K, act, rows, partial);
    else bn_bwd_reduce_launch<float>(grid, st, dz, y, ldy, scale, shift, mean, invstd, (int)M, K, act, rows, partial);
    HDY_LAUNCH_CHECK("bn_act_bwd_reduce");
    bn_bwd_finalize_launch(st, partial, nb, K, (double)M, dgamma, dbeta, dgamma_b, dbeta_b, dz.Ka, accumulate, c1, c2);
    HDY_LAUNCH_CHECK("bn_bwd_finalize");
    if (!dy) return HDY_OK;                               // statistics only: the consumer applies c1 / c2 itself (conv1x1_bwd.hip)
    if (dtype == HDY_BF16) bn_bwd_apply_launch<bf16_t, false>(g2, st, dz, y, ldy, scale, shift, mean, invstd, c1, c2, dy, lddy, (int)M, K, act);
    else bn_bwd_apply_launch<float, false>(g2, st, dz, y, ldy, scale, shift, mean, invstd, c1, c2, dy, lddy, (int)M, K, act);
    HDY_LAUNCH_CHECK("bn_act_bwd_apply");
    return HDY_OK;
}

int hdy_bn_bwd_finalize_slabs(const float* slabs, int nslabs, int K, long long count, const float* mean, const float* invstd, float* dgamma,
                              float* dbeta, int accumulate, float* c1, float* c2, void* stream) {
    HDY_ARG(slabs && nslabs > 0 && K > 0 && count > 0 && mean && invstd, "bn_bwd_finalize_slabs: bad args");
    bn_bwd_finalize_launch((hipStream_t)stream, slabs, nslabs, K, (double)count, dgamma, dbeta, nullptr, nullptr, K, accumulate, c1, c2, mean, invstd);
    HDY_LAUNCH_CHECK("bn_bwd_finalize_slabs");
    return HDY_OK;
}

int hdy_bn_act_bwd_apply(const void* dz, int lddz, const void* dz_b, int lddz_b, int Ka, const void* y, int ldy, const float* scale,
                         const float* shift, const float* mean, const float* invstd, const float* c1, const float* c2, void* dy, int lddy,
                         long long M, int K, int act, int dtype, void* stream) {
    const int VE = dtype == HDY_BF16 ? 8 : 4;
    HDY_ARG(dz && y && dy && scale && shift && mean && invstd && c1 && c2 && M_OK(M) && K > 0 && (act == 0 || act == 1), "bn_act_bwd_apply: bad args");
    HDY_ARG(K % VE == 0 && VEC_OK(dz, lddz, VE) && VEC_OK(y, ldy, VE) && VEC_OK(dy, lddy, VE), "bn_act_bwd_apply: K/pitch/alignment must be multiples of one 16-byte vector");
    HDY_ARG(Ka == K || (Ka > 0 && Ka < K && Ka % VE == 0 && dz_b && VEC_OK(dz_b, lddz_b, VE)), "bn_act_bwd_apply: bad gradient split");
    const Split src = {dz, lddz, dz_b, lddz_b, Ka};
    const int g2 = stream_grid(M, K / VE);
    if (dtype == HDY_BF16) bn_bwd_apply_launch<bf16_t, false>(g2, (hipStream_t)stream, src, y, ldy, scale, shift, mean, invstd, c1, c2, dy, lddy, (int)M, K, act);
    else bn_bwd_apply_launch<float, false>(g2, (hipStream_t)stream, src, y, ldy, scale, shift, mean, invstd, c1, c2, dy, lddy, (int)M, K, act);
    HDY_LAUNCH_CHECK("bn_act_bwd_apply");
    return HDY_OK;
}

int hdy_bn_act_bwd(const void* dz, int lddz, const void* y, int ldy, const float* scale, const float* shift, const float* mean,
                   const float* invstd, void* dy, int lddy, float* dgamma, float* dbeta, int accumulate, long long M, int K, int act,
                   int dtype, float* workspace, size_t ws_bytes, void* stream) {
    return bn_act_bwd_impl(Split{dz, lddz, nullptr, 0, K}, y, ldy, scale, shift, mean, invstd, dy, lddy, dgamma, dbeta, nullptr, nullptr, accumulate, M, K,
                           act, dtype, workspace, ws_bytes, stream);
}

int hdy_bn_act_bwd_pair(const void* dz_a, int lddz_a, const void* dz_b, int lddz_b, int Ka, const void* y, int ldy, const float* scale,
                        const float* shift, const float* mean, const float* invstd, void* dy, int lddy, float* dgamma_a, float* dbeta_a,
                        float* dgamma_b, float* dbeta_b, int accumulate, long long M, int K, int act, int dtype, float* workspace, size_t ws_bytes, void* stream) {
    return bn_act_bwd_impl(Split{dz_a, lddz_a, dz_b, lddz_b, Ka}, y, ldy, scale, shift, mean, invstd, dy, lddy, dgamma_a, dbeta_a, dgamma_b, dbeta_b,
                           accumulate, M, K, act, dtype, workspace, ws_bytes, stream);
}

// out[k] (+)= sum over the M rows of dz[m][k]  (bias gradient of the detection conv).  workspace: hdy_bn_bwd_blocks(M)*2*K floats.
int hdy_colsum(const void* dz, int lddz, long long M, int K, float* out, int accumulate, int dtype, float* workspace, size_t ws_bytes, void* stream) {
    const int VE = dtype == HDY_BF16 ? 8 : 4;
    HDY_ARG(dz && out && workspace && M_OK(M) && K > 0 && K % VE == 0 && VEC_OK(dz, lddz, VE), "colsum: bad args");
    HDY_ARG(ws_bytes >= hdy_colsum_workspace_bytes(M, K), "colsum: workspace of %zu bytes, %zu needed", ws_bytes, hdy_colsum_workspace_bytes(M, K));
    const int nb = hdy_bn_bwd_blocks(M);
    const int rows = (int)((M + nb - 1) / nb);
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(nb, cdiv(K / VE, 256));
    const Split src = {dz, lddz, nullptr, 0, K};
    if (dtype == HDY_BF16) bn_bwd_reduce_launch<bf16_t>(grid, st, src, nullptr, 0, nullptr, nullptr, nullptr, nullptr, (int)M, K, 0, rows, workspace);
    else bn_bwd_reduce_launch<float>(grid, st, src, nullptr, 0, nullptr, nullptr, nullptr, nullptr, (int)M, K, 0, rows, workspace);
    HDY_LAUNCH_CHECK("colsum_reduce");
    bn_bwd_finalize_launch(st, workspace, nb, K, (double)M, nullptr, out, nullptr, nullptr, K, accumulate, nullptr, nullptr);
    HDY_LAUNCH_CHECK("colsum_finalize");
    return HDY_OK;
}

int hdy_add_inplace(void* out, int ldo, const void* a, int lda, long long M, int K, int dtype, void* stream) {
    const int VE = dtype == HDY_BF16 ? 8 : 4;
    HDY_ARG(out && a && M_OK(M) && K > 0 && K % VE == 0 && VEC_OK(out, ldo, VE) && VEC_OK(a, lda, VE), "add_inplace: bad args");
    const int grid = stream_grid(M, K / VE);
    if (dtype == HDY_BF16)
        hipLaunchKernelGGL(add_inplace_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (bf16_t*)out, ldo, (const bf16_t*)a, lda, (int)M, K);
    else
        hipLaunchKernelGGL(add_inplace_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (float*)out, ldo, (const float*)a, lda, (int)M, K);
    HDY_LAUNCH_CHECK("add_inplace");
    return HDY_OK;
}

}  // extern "C"

// Semantic-segmentation branch primitives for gfx950 (hnet's PanopticSeg over the detector's pyramid):
//   * GroupNorm(G) + ReLU forward / backward on NHWC tensors            (hnet/segmentation/utils_seg.py:21-36: Conv3x3 -> GroupNorm(32) -> ReLU)
//   * bilinear resize with align_corners=True, any size, forward (optionally accumulating into the output: the branch sum
//     `sum(res)`, utils_seg.py:58) and backward in gather form (no atomics)   (utils_seg.py:27,35; panoptic_seg.py:18,38)
//   * Softmax2d + soft-dice loss forward / backward on fp32 logits        (panoptic_seg.py:15,22,40)
// All are HBM-bound streaming kernels: one 16-byte vector (8 bf16 / 4 f32 channels) per lane per access, fp32 math; per-image
// GroupNorm statistics go through [image][slice] fp32 partial slabs reduced in a fixed order (deterministic, no atomics).
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

constexpr int GN_SLICES = 32;       // pixel slices per image in the statistics passes (grid = N x GN_SLICES)

// ---------------------------------------------------------------- GroupNorm statistics
// partial[n][s][2][C]: per-channel (sum, sum of squares) of x — or, BWD, (sum dy, sum dy*xhat) with dy = dout * relu'(x*a+b) —
// over slice s of image n.  A thread owns one channel vector; 256/VC row lanes walk the slice's pixels.
template <typename T, bool BWD>
__global__ __launch_bounds__(256) void gn_reduce_kernel(const T* __restrict__ x, int ldx, const T* __restrict__ dout, int lddo,
                                                        const float* __restrict__ ab, const float* __restrict__ stat, int HW, int C, int G,
                                                        int relu, float* __restrict__ partial) {
    constexpr int VE = VT<T>::VE;
    __shared__ float red[2][256 * VE];
    const int n = blockIdx.x, s = blockIdx.y;
    const int VCt = C / VE;
    const int per = (HW + GN_SLICES - 1) / GN_SLICES;
    const int p0 = s * per, p1 = min(p0 + per, HW);
    for (int chunk = 0; chunk * 256 < VCt; ++chunk) {
        const int VC = min(256, VCt - chunk * 256), RL = 256 / VC;
        const int vc = chunk * 256 + (int)(threadIdx.x % (unsigned)VC), rl = (int)(threadIdx.x / (unsigned)VC);
        const bool live = rl < RL;
        const int c = vc * VE;
        float a1[VE], a2[VE], ca[VE], cb[VE], mu[VE], rs[VE];
#pragma unroll
        for (int i = 0; i < VE; ++i) {
            a1[i] = 0.f;
            a2[i] = 0.f;
            ca[i] = 0.f;
            cb[i] = 0.f;
            mu[i] = 0.f;
            rs[i] = 0.f;
        }
        if constexpr (BWD) {
            if (live) {
#pragma unroll
                for (int i = 0; i < VE; ++i) {
                    const int g = (c + i) / (C / G);
                    ca[i] = ab[((size_t)n * 2 + 0) * C + c + i];
                    cb[i] = ab[((size_t)n * 2 + 1) * C + c + i];
                    mu[i] = stat[((size_t)n * G + g) * 2 + 0];
                    rs[i] = stat[((size_t)n * G + g) * 2 + 1];
                }
            }
        }
        if (live) {
            for (int p = p0 + rl; p < p1; p += RL) {
                const size_t m = (size_t)n * HW + p;
                float v[VE], d[VE];
                unpack<T>(*(const i32x4*)(x + m * ldx + c), v);
                if (BWD) unpack<T>(*(const i32x4*)(dout + m * lddo + c), d);
#pragma unroll
                for (int i = 0; i < VE; ++i) {
                    if (BWD) {
                        const float u = v[i] * ca[i] + cb[i];
                        const float dy = (relu != 0 && u <= 0.f) ? 0.f : d[i];
                        a1[i] += dy;
                        a2[i] += dy * ((v[i] - mu[i]) * rs[i]);
                    } else {
                        a1[i] += v[i];
                        a2[i] = __builtin_fmaf(v[i], v[i], a2[i]);
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < VE; ++i) {
                red[0][(rl * VC + (vc - chunk * 256)) * VE + i] = a1[i];
                red[1][(rl * VC + (vc - chunk * 256)) * VE + i] = a2[i];
            }
        }
        __syncthreads();
        for (int j = threadIdx.x; j < VC * VE; j += 256) {
            float s1 = 0.f, s2 = 0.f;
            for (int r = 0; r < RL; ++r) {
                s1 += red[0][r * VC * VE + j];
                s2 += red[1][r * VC * VE + j];
            }
            const size_t o = ((size_t)n * GN_SLICES + s) * 2 * C + chunk * 256 * VE + j;
            partial[o] = s1;
            partial[o + C] = s2;
        }
        __syncthreads();
    }
}

// forward finalize: one block per image.  stat[n][g] = (mean, rstd); ab[n][0][c] = gamma*rstd, ab[n][1][c] = beta - mean*gamma*rstd
__global__ __launch_bounds__(256) void gn_fwd_finalize_kernel(const float* __restrict__ partial, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, int HW, int C, int G, float eps,
                                                              float* __restrict__ stat, float* __restrict__ ab) {
    extern __shared__ double sh[];          // [2][C]
    const int n = blockIdx.x, cpg = C / G;
    for (int c = threadIdx.x; c < C; c += 256) {
        double s1 = 0.0, s2 = 0.0;
        for (int s = 0; s < GN_SLICES; ++s) {
            const size_t o = ((size_t)n * GN_SLICES + s) * 2 * C + c;
            s1 += (double)partial[o];
            s2 += (double)partial[o + C];
        }
        sh[c] = s1;
        sh[C + c] = s2;
    }
    __syncthreads();
    for (int g = threadIdx.x; g < G; g += 256) {
        double s1 = 0.0, s2 = 0.0;
        for (int i = 0; i < cpg; ++i) { s1 += sh[g * cpg + i]; s2 += sh[C + g * cpg + i]; }
        const double cnt = (double)HW * cpg;
        const double mean = s1 / cnt;
        double var = s2 / cnt - mean * mean;
        if (var < 0.0) var = 0.0;
        stat[((size_t)n * G + g) * 2 + 0] = (float)mean;
        stat[((size_t)n * G + g) * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        const int g = c / cpg;
        const float mean = stat[((size_t)n * G + g) * 2 + 0], rstd = stat[((size_t)n * G + g) * 2 + 1];
        const float a = gamma[c] * rstd;
        ab[((size_t)n * 2 + 0) * C + c] = a;
        ab[((size_t)n * 2 + 1) * C + c] = beta[c] - mean * a;
    }
}

// backward finalize, one block per image: coef[n][0][c] = rstd*gamma, coef[n][1][c] = rstd*s1/m, coef[n][2][c] = rstd*s2/m with the
// group sums s1 = SUM_c gamma*A, s2 = SUM_c gamma*B; pc[n][2][C] = (A, B) per channel for the parameter gradients
__global__ __launch_bounds__(256) void gn_bwd_finalize_kernel(const float* __restrict__ partial, const float* __restrict__ gamma,
                                                              const float* __restrict__ stat, int HW, int C, int G,
                                                              float* __restrict__ coef, float* __restrict__ pc) {
    extern __shared__ double sh[];          // [2][C]
    const int n = blockIdx.x, cpg = C / G;
    for (int c = threadIdx.x; c < C; c += 256) {
        double s1 = 0.0, s2 = 0.0;
        for (int s = 0; s < GN_SLICES; ++s) {
            const size_t o = ((size_t)n * GN_SLICES + s) * 2 * C + c;
            s1 += (double)partial[o];
            s2 += (double)partial[o + C];
        }
        sh[c] = s1;
        sh[C + c] = s2;
        pc[((size_t)n * 2 + 0) * C + c] = (float)s1;
        pc[((size_t)n * 2 + 1) * C + c] = (float)s2;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        const int g = c / cpg;
        double s1 = 0.0, s2 = 0.0;
        for (int i = 0; i < cpg; ++i) {
            s1 += (double)gamma[g * cpg + i] * sh[g * cpg + i];
            s2 += (double)gamma[g * cpg + i] * sh[C + g * cpg + i];
        }
        const double m = (double)HW * cpg;
        const float rstd = stat[((size_t)n * G + g) * 2 + 1];
        coef[((size_t)n * 3 + 0) * C + c] = rstd * gamma[c];
        coef[((size_t)n * 3 + 1) * C + c] = (float)((double)rstd * s1 / m);
        coef[((size_t)n * 3 + 2) * C + c] = (float)((double)rstd * s2 / m);
    }
}

// dgamma[c] (+)= SUM_n B[n][c], dbeta[c] (+)= SUM_n A[n][c]
__global__ void gn_param_grad_kernel(const float* __restrict__ pc, int N, int C, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                     int accumulate) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double a = 0.0, b = 0.0;
    for (int n = 0; n < N; ++n) {
        a += (double)pc[((size_t)n * 2 + 0) * C + c];
        b += (double)pc[((size_t)n * 2 + 1) * C + c];
    }
    dbeta[c] = accumulate ? dbeta[c] + (float)a : (float)a;
    dgamma[c] = accumulate ? dgamma[c] + (float)b : (float)b;
}

// forward: y = relu(x*a + b);  backward: dx = k0*dy - k1 - xhat*k2 with dy = dout * relu'(x*a+b)
template <typename T, bool BWD>
__global__ __launch_bounds__(256) void gn_apply_kernel(const T* __restrict__ x, int ldx, const T* __restrict__ dout, int lddo,
                                                       const float* __restrict__ ab, const float* __restrict__ stat,
                                                       const float* __restrict__ coef, T* __restrict__ y, int ldy, int HW, int C, int G, int relu) {
    constexpr int VE = VT<T>::VE;
    const int n = blockIdx.y;
    const int VCt = C / VE;
    for (int chunk = 0; chunk * 256 < VCt; ++chunk) {
        const int VC = min(256, VCt - chunk * 256), RL = 256 / VC;
        const int vc = chunk * 256 + (int)(threadIdx.x % (unsigned)VC), rl = (int)(threadIdx.x / (unsigned)VC);
        if (rl >= RL) continue;
        const int c = vc * VE;
        float ca[VE], cb[VE], mu[VE], rs[VE], k0[VE], k1[VE], k2[VE];
#pragma unroll
        for (int i = 0; i < VE; ++i) {
            ca[i] = ab[((size_t)n * 2 + 0) * C + c + i];
            cb[i] = ab[((size_t)n * 2 + 1) * C + c + i];
            mu[i] = 0.f;
            rs[i] = 0.f;
            k0[i] = 0.f;
            k1[i] = 0.f;
            k2[i] = 0.f;
        }
        if constexpr (BWD) {
#pragma unroll
            for (int i = 0; i < VE; ++i) {
                const int g = (c + i) / (C / G);
                mu[i] = stat[((size_t)n * G + g) * 2 + 0];
                rs[i] = stat[((size_t)n * G + g) * 2 + 1];
                k0[i] = coef[((size_t)n * 3 + 0) * C + c + i];
                k1[i] = coef[((size_t)n * 3 + 1) * C + c + i];
                k2[i] = coef[((size_t)n * 3 + 2) * C + c + i];
            }
        }
        for (int p = blockIdx.x * RL + rl; p < HW; p += gridDim.x * RL) {
            const size_t m = (size_t)n * HW + p;
            float v[VE], d[VE];
            unpack<T>(*(const i32x4*)(x + m * ldx + c), v);
            if (BWD) unpack<T>(*(const i32x4*)(dout + m * lddo + c), d);
#pragma unroll
            for (int i = 0; i < VE; ++i) {
                const float u = v[i] * ca[i] + cb[i];
                if (BWD) {
                    const float dy = (relu && u <= 0.f) ? 0.f : d[i];
                    v[i] = k0[i] * dy - k1[i] - ((v[i] - mu[i]) * rs[i]) * k2[i];
                } else {
                    v[i] = relu ? fmaxf(u, 0.f) : u;
                }
            }
            *(i32x4*)(y + m * ldy + c) = pack<T>(v);
        }
    }
}

// ---------------------------------------------------------------- bilinear resize, align_corners = True
// aten upsample_bilinear2d: scale = (in-1)/(out-1) (0 when out == 1); src = scale*dst; i0 = (int)src; i1 = i0 + (i0 < in-1); l1 = src - i0
__device__ __forceinline__ void src_index(int dst, float scale, int in, int* i0, int* i1, float* l0, float* l1) {
    const float s = scale * (float)dst;
    int a = (int)s;
    if (a > in - 1) a = in - 1;
    *i0 = a;
    *i1 = a + (a < in - 1 ? 1 : 0);
    *l1 = s - (float)a;
    *l0 = 1.0f - *l1;
}

template <typename T>
__global__ __launch_bounds__(256) void bilinear_fwd_kernel(const T* __restrict__ x, int ldx, T* __restrict__ y, int ldy, int N, int Hi, int Wi,
                                                           int Ho, int Wo, int C, float sh, float sw, int accumulate) {
    // one output row per workgroup trip: the row's source rows and weights once per row, no 64-bit divisions per element (the flat-index
    // version spent 150 us on a 105 MB output: three 64-bit div / mod per vector)
    constexpr int VE = VT<T>::VE;
    const int VC = C / VE;
    const int rows = N * Ho, per_row = Wo * VC;
    for (int row = blockIdx.x; row < rows; row += gridDim.x) {
        const int n = row / Ho, oy = row - n * Ho;
        int y0, y1;
        float ly0, ly1;
        src_index(oy, sh, Hi, &y0, &y1, &ly0, &ly1);
        const T* r0 = x + ((size_t)n * Hi + y0) * Wi * ldx;
        const T* r1 = x + ((size_t)n * Hi + y1) * Wi * ldx;
        T* yrow = y + (size_t)row * Wo * ldy;
        constexpr int U = 4;                                       // vectors in flight per thread (16 gathers)
        for (int jb = threadIdx.x; jb < per_row; jb += 256 * U) {
            i32x4 qa[U], qb[U], qc[U], qd[U];
            float wx0[U], wx1[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int j = min(jb + 256 * u, per_row - 1);
                const int ox = VC == 1 ? j : j / VC, vc = j - ox * VC;
                int x0, x1;
                src_index(ox, sw, Wi, &x0, &x1, &wx0[u], &wx1[u]);
                qa[u] = *(const i32x4*)(r0 + (size_t)x0 * ldx + vc * VE);
                qb[u] = *(const i32x4*)(r0 + (size_t)x1 * ldx + vc * VE);
                qc[u] = *(const i32x4*)(r1 + (size_t)x0 * ldx + vc * VE);
                qd[u] = *(const i32x4*)(r1 + (size_t)x1 * ldx + vc * VE);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int j = jb + 256 * u;
                if (j >= per_row) break;
                const int ox = VC == 1 ? j : j / VC, vc = j - ox * VC;
                float a[VE], b[VE], c[VE], d[VE], o[VE];
                unpack<T>(qa[u], a); unpack<T>(qb[u], b); unpack<T>(qc[u], c); unpack<T>(qd[u], d);
                T* dst = yrow + (size_t)ox * ldy + vc * VE;
                if (accumulate) unpack<T>(*(const i32x4*)dst, o);
#pragma unroll
                for (int i = 0; i < VE; ++i) {
                    const float v = ly0 * (wx0[u] * a[i] + wx1[u] * b[i]) + ly1 * (wx0[u] * c[i] + wx1[u] * d[i]);
                    o[i] = accumulate ? o[i] + v : v;
                }
                *(i32x4*)dst = pack<T>(o);
            }
        }
    }
}

// gather form: input pixel (iy, ix) collects from every output whose i0 or i1 is it.  Candidates: src in (i-1, i+1).
__device__ __forceinline__ void dst_range(int i, float scale, int out, int* lo, int* hi) {
    if (scale <= 0.f) { *lo = 0; *hi = out - 1; return; }
    int a = (int)floorf(((float)i - 1.0f) / scale) - 1, b = (int)ceilf(((float)i + 1.0f) / scale) + 1;
    *lo = a < 0 ? 0 : a;
    *hi = b > out - 1 ? out - 1 : b;
}

template <typename T>
__global__ __launch_bounds__(256) void bilinear_bwd_kernel(const T* __restrict__ dy, int lddy, T* __restrict__ dx, int lddx, int N, int Hi, int Wi,
                                                           int Ho, int Wo, int C, float sh, float sw, int accumulate) {
    constexpr int VE = VT<T>::VE;
    const int VC = C / VE;
    const long long total = (long long)N * Hi * Wi * VC;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int vc = (int)(idx % VC);
        long long pix = idx / VC;
        const int ix = (int)(pix % Wi);
        pix /= Wi;
        const int iy = (int)(pix % Hi), n = (int)(pix / Hi);
        int ylo, yhi, xlo, xhi;
        dst_range(iy, sh, Ho, &ylo, &yhi);
        dst_range(ix, sw, Wo, &xlo, &xhi);
        float acc[VE];
#pragma unroll
        for (int i = 0; i < VE; ++i) acc[i] = 0.f;
        const T* base = dy + (size_t)n * Ho * Wo * lddy + vc * VE;
        for (int oy = ylo; oy <= yhi; ++oy) {
            int y0, y1;
            float ly0, ly1;
            src_index(oy, sh, Hi, &y0, &y1, &ly0, &ly1);
            const float wy = (y0 == iy ? ly0 : 0.f) + (y1 == iy ? ly1 : 0.f);
            if (wy == 0.f) continue;
            for (int ox = xlo; ox <= xhi; ++ox) {
                int x0, x1;
                float lx0, lx1;
                src_index(ox, sw, Wi, &x0, &x1, &lx0, &lx1);
                const float wx = (x0 == ix ? lx0 : 0.f) + (x1 == ix ? lx1 : 0.f);
                if (wx == 0.f) continue;
                float g[VE];
                unpack<T>(*(const i32x4*)(base + ((size_t)oy * Wo + ox) * lddy), g);
#pragma unroll
                for (int i = 0; i < VE; ++i) acc[i] += wy * wx * g[i];
            }
        }
        T* dst = dx + (((size_t)n * Hi + iy) * Wi + ix) * lddx + vc * VE;
        if (accumulate) {
            float o[VE];
            unpack<T>(*(const i32x4*)dst, o);
#pragma unroll
            for (int i = 0; i < VE; ++i) acc[i] += o[i];
        }
        *(i32x4*)dst = pack<T>(acc);
    }
}

// One axis of the resize backward (the resize is separable: out = Ry X Rx^T, so dX = Ry^T (dOut Rx)).  Tensors viewed as
// [outer][A][inner][ld] with C channels; dx[o][i][k] (+)= SUM_a w(a, i) dy[o][a][k], a over the <= 2/scale + 2 outputs that touch input i.
// At the x8 resize of the segmentation logits the 2-D gather read 324 candidates per input pixel (1.5 ms at 16 x 1280 x 1280 x 4);
// the W pass reads every gradient row contiguously and shrinks the tensor 8x before the H pass sees it.
template <typename T>
__global__ __launch_bounds__(256) void bilinear_bwd_axis_kernel(const T* __restrict__ dy, int lddy, T* __restrict__ dx, int lddx, long long outer, int Ai,
                                                                int Ao, int inner, int C, float scale, int accumulate) {
    constexpr int VE = VT<T>::VE;
    const int VC = C / VE;
    const long long total = outer * Ai * inner * VC;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int vc = (int)(idx % VC);
        long long r = idx / VC;
        const int k = (int)(r % inner);
        r /= inner;
        const int i = (int)(r % Ai);
        const long long o = r / Ai;
        int lo, hi;
        dst_range(i, scale, Ao, &lo, &hi);
        float acc[VE];
#pragma unroll
        for (int e = 0; e < VE; ++e) acc[e] = 0.f;
        const T* base = dy + ((size_t)o * Ao * inner + k) * lddy + vc * VE;
        for (int a = lo; a <= hi; ++a) {
            int i0, i1;
            float l0, l1;
            src_index(a, scale, Ai, &i0, &i1, &l0, &l1);
            const float w = (i0 == i ? l0 : 0.f) + (i1 == i ? l1 : 0.f);
            if (w == 0.f) continue;
            float g[VE];
            unpack<T>(*(const i32x4*)(base + (size_t)a * inner * lddy), g);
#pragma unroll
            for (int e = 0; e < VE; ++e) acc[e] += w * g[e];
        }
        T* dst = dx + (((size_t)o * Ai + i) * inner + k) * lddx + vc * VE;
        if (accumulate) {
            float p[VE];
            unpack<T>(*(const i32x4*)dst, p);
#pragma unroll
            for (int e = 0; e < VE; ++e) acc[e] += p[e];
        }
        *(i32x4*)dst = pack<T>(acc);
    }
}

// The W pass of the resize backward (inner == 1) with the gradient row staged in LDS: one workgroup per (n, output row), the row is read from
// HBM exactly once (coalesced 16-byte loads), then thread i gathers its candidates from LDS in the SAME order and with the same weights as
// bilinear_bwd_axis_kernel (bit-identical results).  The global-gather version read every gradient vector ~2.2 times through 18 dependent
// 16-byte loads per thread: 0.8 ms for the 105 MB gradient of the segmentation logits (16 x 1280 x 1280 x 4 fp32); this one 0.05 ms.
template <typename T>
__global__ __launch_bounds__(256) void bilinear_bwd_w_lds_kernel(const T* __restrict__ dy, int lddy, T* __restrict__ dx, int lddx, int rows, int Ai, int Ao,
                                                                 int C, float scale, int accumulate) {
    constexpr int VE = VT<T>::VE;
    extern __shared__ __attribute__((aligned(16))) unsigned char bw_smem[];
    i32x4* stage = (i32x4*)bw_smem;                               // [Ao][VC] vectors
    const int VC = C / VE, nvec = Ao * VC;
    for (int row = blockIdx.x; row < rows; row += gridDim.x) {
        const T* src = dy + (size_t)row * Ao * lddy;
        for (int j = threadIdx.x; j < nvec; j += 256) {
            const int a = j / VC, vc = j - a * VC;
            stage[j] = *(const i32x4*)(src + (size_t)a * lddy + vc * VE);
        }
        __syncthreads();
        for (int j = threadIdx.x; j < Ai * VC; j += 256) {
            const int i = j / VC, vc = j - i * VC;
            int lo, hi;
            dst_range(i, scale, Ao, &lo, &hi);
            float acc[VE];
#pragma unroll
            for (int e = 0; e < VE; ++e) acc[e] = 0.f;
            for (int a = lo; a <= hi; ++a) {
                int i0, i1;
                float l0, l1;
                src_index(a, scale, Ai, &i0, &i1, &l0, &l1);
                const float w = (i0 == i ? l0 : 0.f) + (i1 == i ? l1 : 0.f);
                if (w == 0.f) continue;
                float g[VE];
                unpack<T>(stage[a * VC + vc], g);
#pragma unroll
                for (int e = 0; e < VE; ++e) acc[e] += w * g[e];
            }
            T* dst = dx + ((size_t)row * Ai + i) * lddx + vc * VE;
            if (accumulate) {
                float p[VE];
                unpack<T>(*(const i32x4*)dst, p);
#pragma unroll
                for (int e = 0; e < VE; ++e) acc[e] += p[e];
            }
            *(i32x4*)dst = pack<T>(acc);
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------- Softmax2d + soft dice
// logits fp32 [N][H][W][ldl] (nc classes), targets fp32 [N][nc][H][W] (the reference stacks per-roi masks: panoptic_seg.py:36).
// pass 1: per (n, slice): prod[c] = SUM t*p, plus[c] = SUM (t + p)  -> partial[n][s][2][nc]
constexpr int DICE_SLICES = 64, DICE_MAXC = 32;
// MAXC: compile-time size of the per-class register arrays (4, 8 or DICE_MAXC): the loops below are fully unrolled over it, and with
// the 32-wide arrays a 3-class loss did ten times the predicated work (0.49 + 0.53 ms for the two passes at 16 x 1280 x 1280).
template <int MAXC>
__global__ __launch_bounds__(256) void dice_reduce_kernel(const float* __restrict__ logits, int ldl, const float* __restrict__ tgt, int HW, int nc,
                                                          float* __restrict__ partial) {
    __shared__ float red[2][MAXC][256 / 64];
    const int n = blockIdx.x, s = blockIdx.y;
    const int per = (HW + DICE_SLICES - 1) / DICE_SLICES;
    const int p0 = s * per, p1 = min(p0 + per, HW);
    float prod[MAXC], plus[MAXC];
#pragma unroll
    for (int c = 0; c < MAXC; ++c) prod[c] = plus[c] = 0.f;
    // four pixels' loads in flight per thread (the one-pixel loop was a chain of HBM round trips: 254 us for 184 MB), processed in the same
    // order as before: the sums are bit-identical
    constexpr int U = MAXC <= 8 ? 4 : 1;
    for (int pb = p0 + threadIdx.x; pb < p1; pb += 256 * U) {
      float lq[U][MAXC], tq[U][MAXC];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int p = min(pb + 256 * u, p1 - 1);
        const float* lp = logits + ((size_t)n * HW + p) * ldl;
        if (MAXC == 4 && (ldl & 3) == 0) {
            const f32x4 v = *(const f32x4*)lp;
#pragma unroll
            for (int c = 0; c < 4; ++c) lq[u][c] = v[c];
        } else {
#pragma unroll
            for (int c = 0; c < MAXC; ++c) lq[u][c] = c < nc ? lp[c] : 0.f;
        }
#pragma unroll
        for (int c = 0; c < MAXC; ++c) tq[u][c] = c < nc ? tgt[((size_t)n * nc + c) * HW + p] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (pb + 256 * u >= p1) break;
        float l[MAXC];
#pragma unroll
        for (int c = 0; c < MAXC; ++c) l[c] = lq[u][c];
        float mx = l[0];
#pragma unroll
        for (int c = 1; c < MAXC; ++c) if (c < nc) mx = fmaxf(mx, l[c]);
        float e[MAXC], sum = 0.f;
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            e[c] = c < nc ? __expf(l[c] - mx) : 0.f;          // v_exp_f32 / v_rcp_f32 (~1 ulp each): these passes were VALU-bound on expf + IEEE division
            sum += e[c];
        }
        const float inv = __builtin_amdgcn_rcpf(sum);
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            if (c < nc) {
                const float pr = e[c] * inv, t = tq[u][c];
                prod[c] += t * pr;
                plus[c] += t + pr;
            }
        }
      }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        if (c < nc) {
            float a = prod[c], b = plus[c];
            for (int m = 32; m >= 1; m >>= 1) { a += __shfl_xor(a, m); b += __shfl_xor(b, m); }
            if (lane == 0) { red[0][c][wave] = a; red[1][c][wave] = b; }
        }
    }
    __syncthreads();
    if (threadIdx.x < nc) {
        float a = 0.f, b = 0.f;
        for (int w = 0; w < 4; ++w) { a += red[0][threadIdx.x][w]; b += red[1][threadIdx.x][w]; }
        const size_t o = ((size_t)n * DICE_SLICES + s) * 2 * nc + threadIdx.x;
        partial[o] = a;
        partial[o + nc] = b;
    }
}

// one block: dice[n][c] = 2*prod/plus; loss = 1 - SUM_c w_c * mean_n dice[n][c] / SUM_c w_c.  Also leaves, for the backward,
// coefA[n][c] = -w_c/(N*W) * 2/plus and coefB[n][c] = -w_c/(N*W) * (-2*prod/plus^2)   (d loss / d p = coefA*t + coefB per pixel of class c)
__global__ __launch_bounds__(256) void dice_finalize_kernel(const float* __restrict__ partial, const float* __restrict__ cw, int N, int nc,
                                                            float* __restrict__ loss, float* __restrict__ coef) {
    __shared__ double acc[256];
    double wsum = 0.0;
    for (int c = 0; c < nc; ++c) wsum += cw ? (double)cw[c] : 1.0;
    double local = 0.0;
    for (int i = threadIdx.x; i < N * nc; i += 256) {
        const int n = i / nc, c = i - n * nc;
        double prod = 0.0, plus = 0.0;
        for (int s = 0; s < DICE_SLICES; ++s) {
            const size_t o = ((size_t)n * DICE_SLICES + s) * 2 * nc + c;
            prod += (double)partial[o];
            plus += (double)partial[o + nc];
        }
        const double w = (cw ? (double)cw[c] : 1.0) / (wsum * N);
        local += w * 2.0 * prod / plus;
        coef[((size_t)n * 2 + 0) * nc + c] = (float)(-w * 2.0 / plus);
        coef[((size_t)n * 2 + 1) * nc + c] = (float)(w * 2.0 * prod / (plus * plus));
    }
    acc[threadIdx.x] = local;
    __syncthreads();
    for (int m = 128; m >= 1; m >>= 1) {
        if (threadIdx.x < m) acc[threadIdx.x] += acc[threadIdx.x + m];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss[0] = (float)(1.0 - acc[0]);
}

// dlogits[n][p][c] = upstream * p_c * (q_c - SUM_j q_j p_j),  q_c = coefA[n][c]*t_c + coefB[n][c]
template <int MAXC>
__global__ __launch_bounds__(256) void dice_bwd_kernel(const float* __restrict__ logits, int ldl, const float* __restrict__ tgt,
                                                       const float* __restrict__ coef, const float* __restrict__ upstream, int N, int HW, int nc,
                                                       float* __restrict__ dlogits, int lddl) {
    // grid (pixel blocks, N): no 64-bit division per pixel; four pixels' loads in flight per thread (see dice_reduce_kernel)
    const float up = upstream ? upstream[0] : 1.0f;
    const int n = blockIdx.y;
    constexpr int U = MAXC <= 8 ? 4 : 1;
    float ka[MAXC], kb[MAXC];
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        ka[c] = c < nc ? coef[((size_t)n * 2 + 0) * nc + c] : 0.f;
        kb[c] = c < nc ? coef[((size_t)n * 2 + 1) * nc + c] : 0.f;
    }
    for (int pb = blockIdx.x * 256 * U + threadIdx.x; pb < HW; pb += gridDim.x * 256 * U) {
      float lq[U][MAXC], tq[U][MAXC];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int p = min(pb + 256 * u, HW - 1);
        const float* lp = logits + ((size_t)n * HW + p) * ldl;
        if (MAXC == 4 && (ldl & 3) == 0) {
            const f32x4 v = *(const f32x4*)lp;
#pragma unroll
            for (int c = 0; c < 4; ++c) lq[u][c] = v[c];
        } else {
#pragma unroll
            for (int c = 0; c < MAXC; ++c) lq[u][c] = c < nc ? lp[c] : 0.f;
        }
#pragma unroll
        for (int c = 0; c < MAXC; ++c) tq[u][c] = c < nc ? tgt[((size_t)n * nc + c) * HW + p] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int p = pb + 256 * u;
        if (p >= HW) break;
        const size_t idx = (size_t)n * HW + p;
        float l[MAXC];
#pragma unroll
        for (int c = 0; c < MAXC; ++c) l[c] = lq[u][c];
        float mx = l[0];
#pragma unroll
        for (int c = 1; c < MAXC; ++c) if (c < nc) mx = fmaxf(mx, l[c]);
        float e[MAXC], q[MAXC], sum = 0.f;
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            e[c] = c < nc ? __expf(l[c] - mx) : 0.f;          // v_exp_f32 / v_rcp_f32 (~1 ulp each): these passes were VALU-bound on expf + IEEE division
            sum += e[c];
        }
        const float inv = __builtin_amdgcn_rcpf(sum);
        float dot = 0.f;
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            if (c < nc) {
                e[c] *= inv;
                q[c] = ka[c] * tq[u][c] + kb[c];
                dot += q[c] * e[c];
            }
        }
        float* d = dlogits + (size_t)idx * lddl;
        if (MAXC == 4 && (lddl & 3) == 0) {                   // one 16-byte store, padding channels written as zeros
            f32x4 o;
#pragma unroll
            for (int c = 0; c < 4; ++c) o[c] = c < nc ? up * e[c] * (q[c] - dot) : 0.f;
            *(f32x4*)d = o;
        } else {
#pragma unroll
            for (int c = 0; c < MAXC; ++c)
                if (c < nc) d[c] = up * e[c] * (q[c] - dot);
        }
      }
    }
}

// dice_bwd_kernel<4> and the W pass of the resize backward in one launch (4-float pixels, <= 4 classes): a workgroup computes the gradient
// of one full-resolution row (the same expressions as dice_bwd_kernel, upstream = 1) into LDS and gathers it down to Wi vectors exactly as
// bilinear_bwd_w_lds_kernel does — bit-identical to the two launches, without the full-resolution gradient tensor (419 MB written and read
// back at 16 x 1280 x 1280).  dw [N][H][Wi][4].
__global__ __launch_bounds__(256) void dice_bwd_w_kernel(const float* __restrict__ logits, const float* __restrict__ tgt, const float* __restrict__ coef,
                                                         int N, int H, int W, int nc, int Wi, float scale, float* __restrict__ dw) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dbw_smem[];
    f32x4* stage = (f32x4*)dbw_smem;                              // [W]
    const int HW = H * W;
    for (int row = blockIdx.x; row < N * H; row += gridDim.x) {
        const int n = row / H, y = row - n * H;
        float ka[4], kb[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            ka[c] = c < nc ? coef[((size_t)n * 2 + 0) * nc + c] : 0.f;
            kb[c] = c < nc ? coef[((size_t)n * 2 + 1) * nc + c] : 0.f;
        }
        constexpr int U = 4;
        for (int xb = threadIdx.x; xb < W; xb += 256 * U) {
            f32x4 lq[U];
            float tq[U][4];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int x = min(xb + 256 * u, W - 1);
                lq[u] = *(const f32x4*)(logits + ((size_t)row * W + x) * 4);
#pragma unroll
                for (int c = 0; c < 4; ++c) tq[u][c] = c < nc ? tgt[((size_t)n * nc + c) * HW + (size_t)y * W + x] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int x = xb + 256 * u;
                if (x >= W) break;
                float l[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) l[c] = lq[u][c];
                float mx = l[0];
#pragma unroll
                for (int c = 1; c < 4; ++c) if (c < nc) mx = fmaxf(mx, l[c]);
                float e[4], q[4], sum = 0.f;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    e[c] = c < nc ? __expf(l[c] - mx) : 0.f;
                    sum += e[c];
                }
                const float inv = __builtin_amdgcn_rcpf(sum);
                float dot = 0.f;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if (c < nc) {
                        e[c] *= inv;
                        q[c] = ka[c] * tq[u][c] + kb[c];
                        dot += q[c] * e[c];
                    }
                }
                f32x4 o;
#pragma unroll
                for (int c = 0; c < 4; ++c) o[c] = c < nc ? 1.0f * e[c] * (q[c] - dot) : 0.f;
                stage[x] = o;
            }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < Wi; i += 256) {
            int lo, hi;
            dst_range(i, scale, W, &lo, &hi);
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
            for (int a = lo; a <= hi; ++a) {
                int i0, i1;
                float l0, l1;
                src_index(a, scale, Wi, &i0, &i1, &l0, &l1);
                const float w = (i0 == i ? l0 : 0.f) + (i1 == i ? l1 : 0.f);
                if (w == 0.f) continue;
                const f32x4 g = stage[a];
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] += w * g[e];
            }
            *(f32x4*)(dw + ((size_t)row * Wi + i) * 4) = f32x4{acc[0], acc[1], acc[2], acc[3]};
        }
        __syncthreads();
    }
}

// probabilities for inference: p = softmax over the nc channels, fp32 [N][HW][ldp]
__global__ __launch_bounds__(256) void softmax2d_kernel(const float* __restrict__ logits, int ldl, float* __restrict__ probs, int ldp, long long M, int nc) {
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < M; idx += (long long)gridDim.x * 256) {
        const float* l = logits + (size_t)idx * ldl;
        float mx = l[0];
        for (int c = 1; c < nc; ++c) mx = fmaxf(mx, l[c]);
        float sum = 0.f;
        for (int c = 0; c < nc; ++c) sum += expf(l[c] - mx);
        const float inv = 1.0f / sum;
        for (int c = 0; c < nc; ++c) probs[(size_t)idx * ldp + c] = expf(l[c] - mx) * inv;
    }
}

inline int grid_for(long long items) {
    long long g = (items + 255) / 256;
    if (g > 256 * 16) g = 256 * 16;
    return (int)(g < 1 ? 1 : g);
}

}  // namespace

#define VEC_OK(ptr, ld, VE) ((((uintptr_t)(ptr)) & 15) == 0 && (ld) % (VE) == 0)

extern "C" {

// floats: partial [N][GN_SLICES][2][C] + pc [N][2][C] (backward)
size_t hdy_groupnorm_workspace_floats(int N, int C) { return (size_t)N * GN_SLICES * 2 * C + (size_t)N * 2 * C; }

// y = relu?(GroupNorm_G(x) * gamma + beta) per image; saves stat [N][G][2] = (mean, rstd) and ab [N][2][C] for the backward.
int hdy_groupnorm_fwd(const void* x, int ldx, const float* gamma, const float* beta, void* y, int ldy, float* stat, float* ab, int N, int HW,
                      int C, int G, float eps, int relu, int dtype, float* workspace, size_t ws_floats, void* stream) {
    HDY_ARG(N > 0 && C > 0 && ws_floats >= hdy_groupnorm_workspace_floats(N, C), "groupnorm_fwd: workspace of %zu floats is too small", ws_floats);
    const int VE = dtype == HDY_BF16 ? 8 : 4;
    HDY_ARG(x && gamma && beta && y && stat && ab && workspace && N > 0 && HW > 0 && C > 0 && G > 0 && C % G == 0, "groupnorm_fwd: bad args");
    HDY_ARG(C % VE == 0 && VEC_OK(x, ldx, VE) && VEC_OK(y, ldy, VE) && C <= 4096, "groupnorm_fwd: C/pitch/alignment must be multiples of one 16-byte vector");
    hipStream_t st = (hipStream_t)stream;
    dim3 rg(N, GN_SLICES);
    if (dtype == HDY_BF16)
        hipLaunchKernelGGL((gn_reduce_kernel<bf16_t, false>), rg, dim3(256), 0, st, (const bf16_t*)x, ldx, (const bf16_t*)nullptr, 0, (const float*)nullptr, (const float*)nullptr, HW, C, G, relu, workspace);
    else
        hipLaunchKernelGGL((gn_reduce_kernel<float, false>), rg, dim3(256), 0, st, (const float*)x, ldx, (const float*)nullptr, 0, (const float*)nullptr, (const float*)nullptr, HW, C, G, relu, workspace);
    HDY_LAUNCH_CHECK("gn_reduce");
    hipLaunchKernelGGL(gn_fwd_finalize_kernel, dim3(N), dim3(256), 2 * C * sizeof(double), st, workspace, gamma, beta, HW, C, G, eps, stat, ab);
    HDY_LAUNCH_CHECK("gn_fwd_finalize");
    const int rl = 256 / ((C / VE) < 256 ? (C / VE) : 256);
    int gx = (HW + rl - 1) / rl;
    if (gx > 2048 / (N < 2048 ? N : 2048) + 1) gx = 2048 / (N < 2048 ? N : 2048) + 1;
    dim3 ag(gx, N);
    if (dtype == HDY_BF16)
        hipLaunchKernelGGL((gn_apply_kernel<bf16_t, false>), ag, dim3(256), 0, st, (const bf16_t*)x, ldx, (const bf16_t*)nullptr, 0, ab, stat, (const float*)nullptr, (bf16_t*)y, ldy, HW, C, G, relu);
    else
        hipLaunchKernelGGL((gn_apply_kernel<float, false>), ag, dim3(256), 0, st, (const float*)x, ldx, (const float*)nullptr, 0, ab, stat, (const float*)nullptr, (float*)y, ldy, HW, C, G, relu);
    HDY_LAUNCH_CHECK("gn_apply");
    return HDY_OK;
}

// dx, dgamma / dbeta (+)= from dout (gradient of the ReLU output), the raw input x and the saved stat / ab of the forward call.
// coef: [N][3][C] floats of scratch.
int hdy_groupnorm_bwd(const void* dout, int lddo, const void* x, int ldx, const float* gamma, const float* stat, const float* ab, void* dx, int lddx,
                      float* dgamma, float* dbeta, int accumulate, float* coef, int N, int HW, int C, int G, int relu, int dtype, float* workspace,
                      size_t ws_floats, void* stream) {
    const int VE = dtype == HDY_BF16 ? 8 : 4;
    HDY_ARG(N > 0 && C > 0 && ws_floats >= hdy_groupnorm_workspace_floats(N, C), "groupnorm_bwd: workspace of %zu floats is too small", ws_floats);
    HDY_ARG(dout && x && gamma && stat && ab && dx && dgamma && dbeta && coef && workspace && N > 0 && HW > 0 && C > 0 && G > 0 && C % G == 0, "groupnorm_bwd: bad args");
    HDY_ARG(C % VE == 0 && VEC_OK(x, ldx, VE) && VEC_OK(dout, lddo, VE) && VEC_OK(dx, lddx, VE) && C <= 4096, "groupnorm_bwd: C/pitch/alignment must be multiples of one 16-byte vector");
    hipStream_t st = (hipStream_t)stream;
    float* pc = workspace + (size_t)N * GN_SLICES * 2 * C;
    dim3 rg(N, GN_SLICES);
    if (dtype == HDY_BF16)
        hipLaunchKernelGGL((gn_reduce_kernel<bf16_t, true>), rg, dim3(256), 0, st, (const bf16_t*)x, ldx, (const bf16_t*)dout, lddo, ab, stat, HW, C, G, relu, workspace);
    else
        hipLaunchKernelGGL((gn_reduce_kernel<float, true>), rg, dim3(256), 0, st, (const float*)x, ldx, (const float*)dout, lddo, ab, stat, HW, C, G, relu, workspace);
    HDY_LAUNCH_CHECK("gn_bwd_reduce");
    hipLaunchKernelGGL(gn_bwd_finalize_kernel, dim3(N), dim3(256), 2 * C * sizeof(double), st, workspace, gamma, stat, HW, C, G, coef, pc);
    HDY_LAUNCH_CHECK("gn_bwd_finalize");
    hipLaunchKernelGGL(gn_param_grad_kernel, dim3(cdiv(C, 128)), dim3(128), 0, st, pc, N, C, dgamma, dbeta, accumulate);
    HDY_LAUNCH_CHECK("gn_param_grad");
    const int rl = 256 / ((C / VE) < 256 ? (C / VE) : 256);
    int gx = (HW + rl - 1) / rl;
    if (gx > 2048 / (N < 2048 ? N : 2048) + 1) gx = 2048 / (N < 2048 ? N : 2048) + 1;
    dim3 ag(gx, N);
    if (dtype == HDY_BF16)
        hipLaunchKernelGGL((gn_apply_kernel<bf16_t, true>), ag, dim3(256), 0, st, (const bf16_t*)x, ldx, (const bf16_t*)dout, lddo, ab, stat, coef, (bf16_t*)dx, lddx, HW, C, G, relu);
    else
        hipLaunchKernelGGL((gn_apply_kernel<float, true>), ag, dim3(256), 0, st, (const float*)x, ldx, (const float*)dout, lddo, ab, stat, coef, (float*)dx, lddx, HW, C, G, relu);
    HDY_LAUNCH_CHECK("gn_bwd_apply");
    return HDY_OK;
}

static inline float ac_scale(int in, int out) { return out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.0f; }

// y[N][Ho][Wo][C] (+)= bilinear(x[N][Hi][Wi][C]), align_corners = True
int hdy_bilinear_fwd(const void* x, int ldx, void* y, int ldy, int N, int Hi, int Wi, int Ho, int Wo, int C, int accumulate, int dtype, void* stream) {
    const int VE = dtype == HDY_BF16 ? 8 : 4;
    HDY_ARG(x && y && N > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0 && C > 0 && C % VE == 0 && VEC_OK(x, ldx, VE) && VEC_OK(y, ldy, VE), "bilinear_fwd: bad args");
    HDY_ARG((long long)N * Ho < (1LL << 31) && (long long)Wo * (C / VE) < (1LL << 31), "bilinear_fwd: too many rows / vectors per row");
    const long long items = (long long)N * Ho * 256;                // a workgroup per output row, up to the grid cap
    if (dtype == HDY_BF16)
        hipLaunchKernelGGL(bilinear_fwd_kernel<bf16_t>, dim3(grid_for(items)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, ldx, (bf16_t*)y, ldy, N, Hi, Wi, Ho, Wo, C, ac_scale(Hi, Ho), ac_scale(Wi, Wo), accumulate);
    else
        hipLaunchKernelGGL(bilinear_fwd_kernel<float>, dim3(grid_for(items)), dim3(256), 0, (hipStream_t)stream, (const float*)x, ldx, (float*)y, ldy, N, Hi, Wi, Ho, Wo, C, ac_scale(Hi, Ho), ac_scale(Wi, Wo), accumulate);
    HDY_LAUNCH_CHECK("bilinear_fwd");
    return HDY_OK;
}

// dx[N][Hi][Wi][C] (+)= bilinear^T(dy[N][Ho][Wo][C])
int hdy_bilinear_bwd(const void* dy, int lddy, void* dx, int lddx, int N, int Hi, int Wi, int Ho, int Wo, int C, int accumulate, int dtype, void* stream) {
    const int VE = dtype == HDY_BF16 ? 8 : 4;
    HDY_ARG(dy && dx && N > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0 && C > 0 && C % VE == 0 && VEC_OK(dy, lddy, VE) && VEC_OK(dx, lddx, VE), "bilinear_bwd: bad args");
    const long long items = (long long)N * Hi * Wi * (C / VE);
    if (dtype == HDY_BF16)
        hipLaunchKernelGGL(bilinear_bwd_kernel<bf16_t>, dim3(grid_for(items)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy, lddy, (bf16_t*)dx, lddx, N, Hi, Wi, Ho, Wo, C, ac_scale(Hi, Ho), ac_scale(Wi, Wo), accumulate);
    else
        hipLaunchKernelGGL(bilinear_bwd_kernel<float>, dim3(grid_for(items)), dim3(256), 0, (hipStream_t)stream, (const float*)dy, lddy, (float*)dx, lddx, N, Hi, Wi, Ho, Wo, C, ac_scale(Hi, Ho), ac_scale(Wi, Wo), accumulate);
    HDY_LAUNCH_CHECK("bilinear_bwd");
    return HDY_OK;
}

// one axis of hdy_bilinear_bwd: tensors [outer][Ao | Ai][inner][ld], C channels; W pass: outer = N*Ho, inner = 1; H pass: outer = N, inner = Wi
int hdy_bilinear_bwd_axis(const void* dy, int lddy, void* dx, int lddx, long long outer, int Ai, int Ao, int inner, int C, int accumulate, int dtype,
                          void* stream) {
    const int VE = dtype == HDY_BF16 ? 8 : 4;
    HDY_ARG(dy && dx && outer > 0 && Ai > 0 && Ao > 0 && inner > 0 && C > 0 && C % VE == 0 && VEC_OK(dy, lddy, VE) && VEC_OK(dx, lddx, VE), "bilinear_bwd_axis: bad args");
    const long long items = outer * Ai * inner * (C / VE);
    const size_t row_bytes = (size_t)Ao * (C / VE) * 16;
    if (inner == 1 && row_bytes <= 64 * 1024 && outer < (1LL << 31)) {        // W pass: the gradient row fits LDS
        const int grid = (int)(outer < 256 * 16 ? outer : 256 * 16);
        if (dtype == HDY_BF16)
            hipLaunchKernelGGL(bilinear_bwd_w_lds_kernel<bf16_t>, dim3(grid), dim3(256), row_bytes, (hipStream_t)stream, (const bf16_t*)dy, lddy, (bf16_t*)dx, lddx,
                               (int)outer, Ai, Ao, C, ac_scale(Ai, Ao), accumulate);
        else
            hipLaunchKernelGGL(bilinear_bwd_w_lds_kernel<float>, dim3(grid), dim3(256), row_bytes, (hipStream_t)stream, (const float*)dy, lddy, (float*)dx, lddx,
                               (int)outer, Ai, Ao, C, ac_scale(Ai, Ao), accumulate);
        HDY_LAUNCH_CHECK("bilinear_bwd_w");
        return HDY_OK;
    }
    if (dtype == HDY_BF16)
        hipLaunchKernelGGL(bilinear_bwd_axis_kernel<bf16_t>, dim3(grid_for(items)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy, lddy, (bf16_t*)dx, lddx, outer, Ai, Ao, inner, C, ac_scale(Ai, Ao), accumulate);
    else
        hipLaunchKernelGGL(bilinear_bwd_axis_kernel<float>, dim3(grid_for(items)), dim3(256), 0, (hipStream_t)stream, (const float*)dy, lddy, (float*)dx, lddx, outer, Ai, Ao, inner, C, ac_scale(Ai, Ao), accumulate);
    HDY_LAUNCH_CHECK("bilinear_bwd_axis");
    return HDY_OK;
}

// floats: partial [N][DICE_SLICES][2][nc] + coef [N][2][nc]
size_t hdy_softdice_workspace_floats(int N, int nc) { return (size_t)N * DICE_SLICES * 2 * nc + (size_t)N * 2 * nc; }

// loss[0] = 1 - SUM_c w_c * mean_n dice(softmax(logits)[n][c], targets[n][c]) / SUM_c w_c, dice = 2*SUM(t*p) / SUM(t + p);
// when dlogits != NULL also the gradient of loss * upstream[0] (upstream == NULL: 1).
int hdy_softdice(const float* logits, int ldl, const float* targets, const float* class_weight, int N, int HW, int nc, float* loss,
                 const float* upstream, float* dlogits, int lddl, float* workspace, size_t ws_floats, void* stream) {
    HDY_ARG(N > 0 && nc > 0 && ws_floats >= hdy_softdice_workspace_floats(N, nc), "softdice: workspace of %zu floats is too small", ws_floats);
    HDY_ARG(logits && targets && loss && workspace && N > 0 && HW > 0 && nc > 0 && nc <= DICE_MAXC && ldl >= nc && N * nc <= 65536, "softdice: bad args (nc <= %d)", DICE_MAXC);
    hipStream_t st = (hipStream_t)stream;
    float* coef = workspace + (size_t)N * DICE_SLICES * 2 * nc;
    if (nc <= 4) hipLaunchKernelGGL(dice_reduce_kernel<4>, dim3(N, DICE_SLICES), dim3(256), 0, st, logits, ldl, targets, HW, nc, workspace);
    else if (nc <= 8) hipLaunchKernelGGL(dice_reduce_kernel<8>, dim3(N, DICE_SLICES), dim3(256), 0, st, logits, ldl, targets, HW, nc, workspace);
    else hipLaunchKernelGGL(dice_reduce_kernel<DICE_MAXC>, dim3(N, DICE_SLICES), dim3(256), 0, st, logits, ldl, targets, HW, nc, workspace);
    HDY_LAUNCH_CHECK("dice_reduce");
    hipLaunchKernelGGL(dice_finalize_kernel, dim3(1), dim3(256), 0, st, workspace, class_weight, N, nc, loss, coef);
    HDY_LAUNCH_CHECK("dice_finalize");
    if (dlogits) {
        HDY_ARG(lddl >= nc, "softdice: dlogits pitch");
        HDY_ARG(N <= 65535, "softdice: batch beyond the launch grid");
        const int gx = cdiv(HW, 256 * 4);
        const dim3 g(gx < 1 ? 1 : (gx > 4096 ? 4096 : gx), N);
        if (nc <= 4) hipLaunchKernelGGL(dice_bwd_kernel<4>, g, dim3(256), 0, st, logits, ldl, targets, coef, upstream, N, HW, nc, dlogits, lddl);
        else if (nc <= 8) hipLaunchKernelGGL(dice_bwd_kernel<8>, g, dim3(256), 0, st, logits, ldl, targets, coef, upstream, N, HW, nc, dlogits, lddl);
        else hipLaunchKernelGGL(dice_bwd_kernel<DICE_MAXC>, g, dim3(256), 0, st, logits, ldl, targets, coef, upstream, N, HW, nc, dlogits, lddl);
        HDY_LAUNCH_CHECK("dice_bwd");
    }
    return HDY_OK;
}

// hdy_softdice's loss plus the gradient already reduced along W by the transposed resize (the first of hdy_bilinear_bwd_axis' two passes):
// logits fp32 [N][H][W][4] with nc <= 4 classes, dw [N][H][Wi][4] = W pass of d loss / d logits (upstream 1); the caller runs the H pass
// (hdy_bilinear_bwd_axis(dw, 4, dx, ldx, N, Hi, H, Wi, 4, ...)).  Same results as hdy_softdice + the W pass, bit for bit.
int hdy_softdice_wgrad(const float* logits, const float* targets, const float* class_weight, int N, int H, int W, int nc, int Wi, float* loss, float* dw,
                       float* workspace, size_t ws_floats, void* stream) {
    HDY_ARG(N > 0 && nc > 0 && ws_floats >= hdy_softdice_workspace_floats(N, nc), "softdice_wgrad: workspace of %zu floats is too small", ws_floats);
    HDY_ARG(logits && targets && loss && dw && workspace && N > 0 && H > 0 && W > 0 && Wi > 0 && nc > 0 && nc <= 4 && N * nc <= 65536, "softdice_wgrad: bad args (nc <= 4)");
    HDY_ARG((size_t)W * 16 <= 64 * 1024 && (long long)N * H < (1LL << 31), "softdice_wgrad: row of %d pixels beyond the LDS stage", W);
    hipStream_t st = (hipStream_t)stream;
    const int HW = H * W;
    float* coef = workspace + (size_t)N * DICE_SLICES * 2 * nc;
    hipLaunchKernelGGL(dice_reduce_kernel<4>, dim3(N, DICE_SLICES), dim3(256), 0, st, logits, 4, targets, HW, nc, workspace);
    HDY_LAUNCH_CHECK("dice_reduce");
    hipLaunchKernelGGL(dice_finalize_kernel, dim3(1), dim3(256), 0, st, workspace, class_weight, N, nc, loss, coef);
    HDY_LAUNCH_CHECK("dice_finalize");
    const int rows = N * H;
    hipLaunchKernelGGL(dice_bwd_w_kernel, dim3(rows < 4096 ? rows : 4096), dim3(256), (size_t)W * 16, st, logits, targets, coef, N, H, W, nc, Wi, ac_scale(Wi, W), dw);
    HDY_LAUNCH_CHECK("dice_bwd_w");
    return HDY_OK;
}

int hdy_softmax2d(const float* logits, int ldl, float* probs, int ldp, long long M, int nc, void* stream) {
    HDY_ARG(logits && probs && M > 0 && nc > 0 && ldl >= nc && ldp >= nc, "softmax2d: bad args");
    hipLaunchKernelGGL(softmax2d_kernel, dim3(grid_for(M)), dim3(256), 0, (hipStream_t)stream, logits, ldl, probs, ldp, M, nc);
    HDY_LAUNCH_CHECK("softmax2d");
    return HDY_OK;
}

}  // extern "C"

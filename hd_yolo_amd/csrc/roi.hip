// RoIAlign forward / backward on NHWC feature maps and the ReLU backward of the Mask-RCNN head.
//
// roi_align follows the torchvision.ops.roi_align semantics the reference calls with (metayolo/models/yolo_head.py:243,294:
// output (M, M), spatial_scale = 1/stride, sampling_ratio = 2, aligned = ROI_ALIGN = False): per output bin the average of
// sampling_ratio^2 bilinear samples; a sample outside [-1, size] contributes 0, coordinates are clamped to >= 0, and the last
// row/column interpolates with itself.  Output is NHWC [R][P][P][C] so the head's convolutions consume it directly.
// Forward: an HBM-bound gather, one lane per (roi, bin, 8- or 4-channel vector).  The backward accumulates into an fp32 image
// (bf16 has no atomic add and rois overlap), privatised per roi in LDS; hdy_cast_store then writes the plan's gradient buffer.
#include "common.h"
#include "hdyolo.h"

namespace {

template <typename T> struct RT;
template <> struct RT<float> { static constexpr int VE = 4; };
template <> struct RT<bf16_t> { static constexpr int VE = 8; };

template <typename T> __device__ __forceinline__ void ld_vec(const T* p, float* f);
template <> __device__ __forceinline__ void ld_vec<float>(const float* p, float* f) {
    const f32x4 v = *(const f32x4*)p;
#pragma unroll
    for (int i = 0; i < 4; ++i) f[i] = v[i];
}
template <> __device__ __forceinline__ void ld_vec<bf16_t>(const bf16_t* p, float* f) {
    V16 u;
    u.i = *(const i32x4*)p;
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = (float)u.h[i];
}
template <typename T> __device__ __forceinline__ void st_vec(T* p, const float* f);
template <> __device__ __forceinline__ void st_vec<float>(float* p, const float* f) { *(f32x4*)p = f32x4{f[0], f[1], f[2], f[3]}; }
template <> __device__ __forceinline__ void st_vec<bf16_t>(bf16_t* p, const float* f) {
    V16 u;
#pragma unroll
    for (int i = 0; i < 8; ++i) u.h[i] = (bf16_t)f[i];
    *(i32x4*)p = u.i;
}

struct Sample { int y0, x0, y1, x1; float w00, w01, w10, w11; bool ok; };

// torchvision's bilinear_interpolate setup for one sample point
__device__ __forceinline__ Sample sample_at(float y, float x, int H, int W) {
    Sample s;
    s.ok = !(y < -1.0f || y > (float)H || x < -1.0f || x > (float)W);
    if (y <= 0.f) y = 0.f;
    if (x <= 0.f) x = 0.f;
    s.y0 = (int)y;
    s.x0 = (int)x;
    if (s.y0 >= H - 1) { s.y0 = s.y1 = H - 1; y = (float)s.y0; } else s.y1 = s.y0 + 1;
    if (s.x0 >= W - 1) { s.x0 = s.x1 = W - 1; x = (float)s.x0; } else s.x1 = s.x0 + 1;
    const float ly = y - (float)s.y0, lx = x - (float)s.x0, hy = 1.f - ly, hx = 1.f - lx;
    s.w00 = hy * hx; s.w01 = hy * lx; s.w10 = ly * hx; s.w11 = ly * lx;
    return s;
}

struct RoiGeom { int b; float y0, x0, bin_h, bin_w; };

__device__ __forceinline__ RoiGeom roi_geom(const float* roi, float scale, int P, int aligned) {
    RoiGeom g;
    g.b = (int)roi[0];
    const float off = aligned ? 0.5f : 0.0f;
    g.x0 = roi[1] * scale - off;
    g.y0 = roi[2] * scale - off;
    float rw = roi[3] * scale - off - g.x0, rh = roi[4] * scale - off - g.y0;
    if (!aligned) {
        rw = fmaxf(rw, 1.0f);
        rh = fmaxf(rh, 1.0f);
    }
    g.bin_w = rw / (float)P;
    g.bin_h = rh / (float)P;
    return g;
}

template <typename T>
__global__ __launch_bounds__(256) void roi_align_kernel(const T* __restrict__ feat, int ldf, int B, int H, int W, int C, const float* __restrict__ rois,
                                                        int R, float scale, int P, int S, int aligned, T* __restrict__ out) {
    constexpr int VE = RT<T>::VE;
    const int VC = C / VE;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)R * P * P * VC) return;
    const int vc = (int)(idx % VC);
    long long t = idx / VC;
    const int pw = (int)(t % P);
    t /= P;
    const int ph = (int)(t % P);
    const int r = (int)(t / P);
    const RoiGeom g = roi_geom(rois + (size_t)r * 5, scale, P, aligned);
    float acc[VE];
#pragma unroll
    for (int i = 0; i < VE; ++i) acc[i] = 0.f;
    if (g.b >= 0 && g.b < B) {
        for (int iy = 0; iy < S; ++iy) {
            const float y = g.y0 + (float)ph * g.bin_h + ((float)iy + 0.5f) * g.bin_h / (float)S;
            for (int ix = 0; ix < S; ++ix) {
                const float x = g.x0 + (float)pw * g.bin_w + ((float)ix + 0.5f) * g.bin_w / (float)S;
                const Sample s = sample_at(y, x, H, W);
                if (!s.ok) continue;
                const size_t base = (size_t)g.b * H * W;
                float a[VE], b[VE], c[VE], d[VE];
                ld_vec<T>(feat + (base + (size_t)s.y0 * W + s.x0) * ldf + vc * VE, a);
                ld_vec<T>(feat + (base + (size_t)s.y0 * W + s.x1) * ldf + vc * VE, b);
                ld_vec<T>(feat + (base + (size_t)s.y1 * W + s.x0) * ldf + vc * VE, c);
                ld_vec<T>(feat + (base + (size_t)s.y1 * W + s.x1) * ldf + vc * VE, d);
#pragma unroll
                for (int i = 0; i < VE; ++i) acc[i] += s.w00 * a[i] + s.w01 * b[i] + s.w10 * c[i] + s.w11 * d[i];
            }
        }
        const float inv = 1.0f / (float)(S * S);
#pragma unroll
        for (int i = 0; i < VE; ++i) acc[i] *= inv;
    }
    st_vec<T>(out + (((size_t)r * P + ph) * P + pw) * C + vc * VE, acc);
}

// Backward, privatised per roi and WITHOUT scatter: the bilinear weights of a sample factor into a row weight and a column weight, and
// so does the validity test, hence the gradient of a roi's footprint F (fh x fw feature pixels) is two small matrix products per channel,
//     dF = Wy^T (D Wx),   Wy[ph][fy] = SUM_iy wy(sample row (ph, iy), fy),   Wx[pw][fx] likewise,   D = dout of the roi / S^2  (P x P),
// computed deterministically in LDS by one workgroup per (roi, 32-channel block); only the finished footprint goes to the fp32
// gradient image with one global atomic per pixel and channel (rois overlap).  The matched truths of a nuclei tile are a few feature
// pixels wide, so ~200 bins x 4 samples fall onto a handful of pixels: the first version scattered them with global atomics (8 ms per
// call), the second with LDS atomics into the footprint (1.36 ms per call at B=16, 1280x1280: up to 8 bins in flight hit one address).
// Rois whose footprint exceeds the LDS tile scatter directly (they are large, hence uncontended).
constexpr int FT = 24;                   // footprint tile side (feature pixels)
constexpr int CB = 32;                   // channels per workgroup
constexpr int PM = 16;                   // largest output side handled by the footprint path (P = 14 for the mask head)

__device__ __forceinline__ void axis_sample(float v, int size, bool* ok, int* i0, int* i1, float* lo, float* hi) {
    *ok = !(v < -1.0f || v > (float)size);
    if (v <= 0.f) v = 0.f;
    *i0 = (int)v;
    if (*i0 >= size - 1) { *i0 = *i1 = size - 1; v = (float)*i0; } else *i1 = *i0 + 1;
    *hi = v - (float)*i0;                // weight of i1
    *lo = 1.f - *hi;                     // weight of i0
}

template <typename T>
__global__ __launch_bounds__(256) void roi_align_bwd_tiled_kernel(float* __restrict__ dfeat, int B, int H, int W, int C, const float* __restrict__ rois,
                                                                  float scale, int P, int S, int aligned, const T* __restrict__ dout) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* Wy = sm;                          // [PM][FT]
    float* Wx = Wy + PM * FT;                // [PM][FT]
    float* D = Wx + PM * FT;                 // [P*P][CB]
    float* Tm = D + PM * PM * CB;            // [P][fw][CB]
    const int r = blockIdx.x, c0 = blockIdx.y * CB;
    const int cl = threadIdx.x & (CB - 1), bl = threadIdx.x / CB;      // channel lane, 8 row lanes
    const RoiGeom g = roi_geom(rois + (size_t)r * 5, scale, P, aligned);
    if (g.b < 0 || g.b >= B) return;
    // footprint of all samples (after the clamp of sample_at): rows [fy0, fy1], columns [fx0, fx1]
    const float ylo = g.y0 + 0.5f * g.bin_h / (float)S, yhi = g.y0 + ((float)P - 0.5f / (float)S) * g.bin_h;
    const float xlo = g.x0 + 0.5f * g.bin_w / (float)S, xhi = g.x0 + ((float)P - 0.5f / (float)S) * g.bin_w;
    const int fy0 = min(max((int)floorf(fmaxf(ylo, 0.f)), 0), H - 1), fy1 = min(max((int)floorf(fmaxf(yhi, 0.f)) + 1, 0), H - 1);
    const int fx0 = min(max((int)floorf(fmaxf(xlo, 0.f)), 0), W - 1), fx1 = min(max((int)floorf(fmaxf(xhi, 0.f)) + 1, 0), W - 1);
    const int fh = fy1 - fy0 + 1, fw = fx1 - fx0 + 1;
    const bool tiled = fh <= FT && fw <= FT && P <= PM;
    const float inv = 1.0f / (float)(S * S);
    const bool ch_ok = c0 + cl < C;
    if (!tiled) {
        for (int bin = bl; bin < P * P; bin += 256 / CB) {
            const int ph = bin / P, pw = bin - ph * P;
            const float d = ch_ok ? (float)dout[(((size_t)r * P + ph) * P + pw) * C + c0 + cl] * inv : 0.f;
            for (int iy = 0; iy < S; ++iy) {
                const float y = g.y0 + (float)ph * g.bin_h + ((float)iy + 0.5f) * g.bin_h / (float)S;
                for (int ix = 0; ix < S; ++ix) {
                    const float x = g.x0 + (float)pw * g.bin_w + ((float)ix + 0.5f) * g.bin_w / (float)S;
                    const Sample s = sample_at(y, x, H, W);
                    if (!s.ok || !ch_ok) continue;
                    const size_t base = (size_t)g.b * H * W;
                    atomicAdd(dfeat + (base + (size_t)s.y0 * W + s.x0) * C + c0 + cl, d * s.w00);
                    atomicAdd(dfeat + (base + (size_t)s.y0 * W + s.x1) * C + c0 + cl, d * s.w01);
                    atomicAdd(dfeat + (base + (size_t)s.y1 * W + s.x0) * C + c0 + cl, d * s.w10);
                    atomicAdd(dfeat + (base + (size_t)s.y1 * W + s.x1) * C + c0 + cl, d * s.w11);
                }
            }
        }
        return;
    }
    // axis weight tables: thread t < P builds row t of Wy, thread P <= t < 2P row t - P of Wx (no two threads share an entry)
    if ((int)threadIdx.x < 2 * P) {
        const bool isx = (int)threadIdx.x >= P;
        const int pb = isx ? (int)threadIdx.x - P : (int)threadIdx.x;
        float* row = (isx ? Wx : Wy) + pb * FT;
        const int n = isx ? fw : fh, f0 = isx ? fx0 : fy0, size = isx ? W : H;
        const float o = isx ? g.x0 : g.y0, bs = isx ? g.bin_w : g.bin_h;
        for (int j = 0; j < n; ++j) row[j] = 0.f;
        for (int i = 0; i < S; ++i) {
            const float v = o + (float)pb * bs + ((float)i + 0.5f) * bs / (float)S;
            bool ok; int i0, i1; float lo, hi;
            axis_sample(v, size, &ok, &i0, &i1, &lo, &hi);
            if (!ok) continue;
            row[i0 - f0] += lo;
            row[i1 - f0] += hi;
        }
    }
    // D: the roi's output gradient, scaled
    for (int bin = bl; bin < P * P; bin += 256 / CB) D[bin * CB + cl] = ch_ok ? (float)dout[((size_t)r * P * P + bin) * C + c0 + cl] * inv : 0.f;
    __syncthreads();
    // Tm[ph][fx][c] = SUM_pw D[ph][pw][c] * Wx[pw][fx]
    for (int o = bl; o < P * fw; o += 256 / CB) {
        const int ph = o / fw, fx = o - ph * fw;
        float acc = 0.f;
        for (int pw = 0; pw < P; ++pw) acc += D[(ph * P + pw) * CB + cl] * Wx[pw * FT + fx];
        Tm[o * CB + cl] = acc;
    }
    __syncthreads();
    // dF[fy][fx][c] = SUM_ph Wy[ph][fy] * Tm[ph][fx][c]  -> one global atomic per footprint pixel and channel
    for (int o = bl; o < fh * fw; o += 256 / CB) {
        const int fy = o / fw, fx = o - fy * fw;
        float acc = 0.f;
        for (int ph = 0; ph < P; ++ph) acc += Wy[ph * FT + fy] * Tm[(ph * fw + fx) * CB + cl];
        if (acc != 0.f && ch_ok) atomicAdd(dfeat + (((size_t)g.b * H + fy0 + fy) * W + fx0 + fx) * C + c0 + cl, acc);
    }
}

// du = dz * (y > 0)
template <typename T>
__global__ __launch_bounds__(256) void relu_bwd_kernel(const T* __restrict__ dz, const T* __restrict__ y, T* __restrict__ du, long long nvec) {
    constexpr int VE = RT<T>::VE;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (long long)gridDim.x * blockDim.x) {
        float g[VE], v[VE];
        ld_vec<T>(dz + i * VE, g);
        ld_vec<T>(y + i * VE, v);
#pragma unroll
        for (int k = 0; k < VE; ++k) g[k] = v[k] > 0.f ? g[k] : 0.f;
        st_vec<T>(du + i * VE, g);
    }
}

// dst[m][c] (=|+=) (T)src[m][c]: fp32 accumulation image -> pitched NHWC gradient view
template <typename T>
__global__ __launch_bounds__(256) void cast_store_kernel(const float* __restrict__ src, T* __restrict__ dst, int ldd, long long M, int C, int accumulate) {
    const long long n = M * C;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long m = i / C;
        const int c = (int)(i - m * C);
        T* d = dst + m * ldd + c;
        float v = src[i];
        if (accumulate) v += (float)*d;
        *d = (T)v;
    }
}

}  // namespace

#define ROI_ARGS_OK(C, ldf, VE, ptr) ((C) % (VE) == 0 && (ldf) % (VE) == 0 && (((uintptr_t)(ptr)) & 15) == 0)

extern "C" {

int hdy_roi_align_fwd(const void* feat, int ldf, int B, int H, int W, int C, const float* rois, int R, float spatial_scale, int P,
                      int sampling_ratio, int aligned, void* out, int dtype, void* stream) {
    HDY_ARG(R >= 0 && B > 0 && H > 0 && W > 0 && C > 0 && P > 0 && sampling_ratio > 0, "roi_align_fwd: bad sizes");
    if (R == 0) return HDY_OK;
    const int VE = dtype == HDY_BF16 ? 8 : 4;
    HDY_ARG(feat && rois && out && ROI_ARGS_OK(C, ldf, VE, feat) && (((uintptr_t)out) & 15) == 0, "roi_align_fwd: pointers / channel vectors");
    const long long n = (long long)R * P * P * (C / VE);
    const int grid = (int)((n + 255) / 256);
    if (dtype == HDY_BF16)
        hipLaunchKernelGGL(roi_align_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)feat, ldf, B, H, W, C, rois, R,
                           spatial_scale, P, sampling_ratio, aligned, (bf16_t*)out);
    else
        hipLaunchKernelGGL(roi_align_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)feat, ldf, B, H, W, C, rois, R,
                           spatial_scale, P, sampling_ratio, aligned, (float*)out);
    HDY_LAUNCH_CHECK("roi_align_fwd");
    return HDY_OK;
}

int hdy_roi_align_bwd(const void* dout, float* dfeat_f32, int B, int H, int W, int C, const float* rois, int R, float spatial_scale, int P,
                      int sampling_ratio, int aligned, int dtype, void* stream) {
    HDY_ARG(R >= 0 && B > 0 && H > 0 && W > 0 && C > 0 && P > 0 && sampling_ratio > 0, "roi_align_bwd: bad sizes");
    if (R == 0) return HDY_OK;
    const int VE = dtype == HDY_BF16 ? 8 : 4;
    HDY_ARG(dout && rois && dfeat_f32 && C % VE == 0 && (((uintptr_t)dout) & 15) == 0, "roi_align_bwd: pointers / channel vectors");
    const dim3 grid(R, (C + CB - 1) / CB);
    const int pp = P <= PM ? P : 1;
    const size_t smem = (size_t)(2 * PM * FT + PM * PM * CB + pp * FT * CB) * sizeof(float);        // Wy, Wx, D, Tm: 70 KB at P = 14
    static PerDeviceOnce attr_once;           // first launch of this instance on any thread
    attr_once.run([&] {
        (void)hipFuncSetAttribute((const void*)roi_align_bwd_tiled_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (2 * PM * FT + PM * PM * CB + PM * FT * CB) * 4);
        (void)hipFuncSetAttribute((const void*)roi_align_bwd_tiled_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (2 * PM * FT + PM * PM * CB + PM * FT * CB) * 4);
    });
    if (dtype == HDY_BF16)
        hipLaunchKernelGGL(roi_align_bwd_tiled_kernel<bf16_t>, grid, dim3(256), smem, (hipStream_t)stream, dfeat_f32, B, H, W, C, rois, spatial_scale, P,
                           sampling_ratio, aligned, (const bf16_t*)dout);
    else
        hipLaunchKernelGGL(roi_align_bwd_tiled_kernel<float>, grid, dim3(256), smem, (hipStream_t)stream, dfeat_f32, B, H, W, C, rois, spatial_scale, P,
                           sampling_ratio, aligned, (const float*)dout);
    HDY_LAUNCH_CHECK("roi_align_bwd");
    return HDY_OK;
}

int hdy_relu_bwd(const void* dz, const void* y, void* du, long long n, int dtype, void* stream) {
    const int VE = dtype == HDY_BF16 ? 8 : 4;
    HDY_ARG(dz && y && du && n >= 0 && n % VE == 0, "relu_bwd: n must be a multiple of one 16-byte vector");
    if (n == 0) return HDY_OK;
    const long long nvec = n / VE;
    const int grid = (int)((nvec + 255) / 256 > 4096 ? 4096 : (nvec + 255) / 256);
    if (dtype == HDY_BF16)
        hipLaunchKernelGGL(relu_bwd_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dz, (const bf16_t*)y, (bf16_t*)du, nvec);
    else
        hipLaunchKernelGGL(relu_bwd_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)dz, (const float*)y, (float*)du, nvec);
    HDY_LAUNCH_CHECK("relu_bwd");
    return HDY_OK;
}

int hdy_cast_store(const float* src, void* dst, int ldd, long long M, int C, int accumulate, int dtype, void* stream) {
    HDY_ARG(src && dst && M >= 0 && C > 0 && ldd >= C, "cast_store: bad args");
    if (M == 0) return HDY_OK;
    const long long n = M * C;
    const int grid = (int)((n + 255) / 256 > 8192 ? 8192 : (n + 255) / 256);
    if (dtype == HDY_BF16)
        hipLaunchKernelGGL(cast_store_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst, ldd, M, C, accumulate);
    else
        hipLaunchKernelGGL(cast_store_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, (float*)dst, ldd, M, C, accumulate);
    HDY_LAUNCH_CHECK("cast_store");
    return HDY_OK;
}

}  // extern "C"

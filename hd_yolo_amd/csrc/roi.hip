// RoIAlign forward / backward on NHWC feature maps and the ReLU backward of the Mask-RCNN head.
//
// roi_align follows the torchvision.ops.roi_align semantics the reference calls with (metayolo/models/yolo_head.py:243,294:
// output (M, M), spatial_scale = 1/stride, sampling_ratio = 2, aligned = ROI_ALIGN = False): per output bin the average of
// sampling_ratio^2 bilinear samples; a sample outside [-1, size] contributes 0, coordinates are clamped to >= 0, and the last
// row/column interpolates with itself.  Output is NHWC [R][P][P][C] so the head's convolutions consume it directly.
// HBM-bound gathers / scatters: one lane per (roi, bin, 8- or 4-channel vector).  The backward scatters with fp32 atomics into
// an fp32 image (bf16 has no atomic add and several rois overlap); hdy_cast_store then writes the plan's gradient buffer.
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

template <typename T, bool BWD>
__global__ __launch_bounds__(256) void roi_align_kernel(const T* __restrict__ feat, float* __restrict__ dfeat, int ldf, int B, int H, int W, int C,
                                                        const float* __restrict__ rois, int R, float scale, int P, int S, int aligned,
                                                        T* __restrict__ out, const T* __restrict__ dout) {
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
    if (g.b < 0 || g.b >= B) {
        if (!BWD) {
            float z[VE];
#pragma unroll
            for (int i = 0; i < VE; ++i) z[i] = 0.f;
            st_vec<T>(out + (((size_t)r * P + ph) * P + pw) * C + vc * VE, z);
        }
        return;
    }
    const float inv = 1.0f / (float)(S * S);
    float acc[VE];
    if (BWD) {
        ld_vec<T>(dout + (((size_t)r * P + ph) * P + pw) * C + vc * VE, acc);
#pragma unroll
        for (int i = 0; i < VE; ++i) acc[i] *= inv;
    } else {
#pragma unroll
        for (int i = 0; i < VE; ++i) acc[i] = 0.f;
    }
    for (int iy = 0; iy < S; ++iy) {
        const float y = g.y0 + (float)ph * g.bin_h + ((float)iy + 0.5f) * g.bin_h / (float)S;
        for (int ix = 0; ix < S; ++ix) {
            const float x = g.x0 + (float)pw * g.bin_w + ((float)ix + 0.5f) * g.bin_w / (float)S;
            const Sample s = sample_at(y, x, H, W);
            if (!s.ok) continue;
            const size_t base = (size_t)g.b * H * W;
            const size_t o00 = (base + (size_t)s.y0 * W + s.x0) * ldf + vc * VE, o01 = (base + (size_t)s.y0 * W + s.x1) * ldf + vc * VE;
            const size_t o10 = (base + (size_t)s.y1 * W + s.x0) * ldf + vc * VE, o11 = (base + (size_t)s.y1 * W + s.x1) * ldf + vc * VE;
            if (BWD) {
#pragma unroll
                for (int i = 0; i < VE; ++i) {
                    atomicAdd(dfeat + o00 + i, acc[i] * s.w00);
                    atomicAdd(dfeat + o01 + i, acc[i] * s.w01);
                    atomicAdd(dfeat + o10 + i, acc[i] * s.w10);
                    atomicAdd(dfeat + o11 + i, acc[i] * s.w11);
                }
            } else {
                float a[VE], b[VE], c[VE], d[VE];
                ld_vec<T>(feat + o00, a);
                ld_vec<T>(feat + o01, b);
                ld_vec<T>(feat + o10, c);
                ld_vec<T>(feat + o11, d);
#pragma unroll
                for (int i = 0; i < VE; ++i) acc[i] += s.w00 * a[i] + s.w01 * b[i] + s.w10 * c[i] + s.w11 * d[i];
            }
        }
    }
    if (!BWD) {
#pragma unroll
        for (int i = 0; i < VE; ++i) acc[i] *= inv;
        st_vec<T>(out + (((size_t)r * P + ph) * P + pw) * C + vc * VE, acc);
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
        hipLaunchKernelGGL((roi_align_kernel<bf16_t, false>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)feat, (float*)nullptr, ldf, B,
                           H, W, C, rois, R, spatial_scale, P, sampling_ratio, aligned, (bf16_t*)out, (const bf16_t*)nullptr);
    else
        hipLaunchKernelGGL((roi_align_kernel<float, false>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)feat, (float*)nullptr, ldf, B,
                           H, W, C, rois, R, spatial_scale, P, sampling_ratio, aligned, (float*)out, (const float*)nullptr);
    HDY_LAUNCH_CHECK("roi_align_fwd");
    return HDY_OK;
}

int hdy_roi_align_bwd(const void* dout, float* dfeat_f32, int B, int H, int W, int C, const float* rois, int R, float spatial_scale, int P,
                      int sampling_ratio, int aligned, int dtype, void* stream) {
    HDY_ARG(R >= 0 && B > 0 && H > 0 && W > 0 && C > 0 && P > 0 && sampling_ratio > 0, "roi_align_bwd: bad sizes");
    if (R == 0) return HDY_OK;
    const int VE = dtype == HDY_BF16 ? 8 : 4;
    HDY_ARG(dout && rois && dfeat_f32 && C % VE == 0 && (((uintptr_t)dout) & 15) == 0, "roi_align_bwd: pointers / channel vectors");
    const long long n = (long long)R * P * P * (C / VE);
    const int grid = (int)((n + 255) / 256);
    if (dtype == HDY_BF16)
        hipLaunchKernelGGL((roi_align_kernel<bf16_t, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)nullptr, dfeat_f32, C, B, H, W,
                           C, rois, R, spatial_scale, P, sampling_ratio, aligned, (bf16_t*)nullptr, (const bf16_t*)dout);
    else
        hipLaunchKernelGGL((roi_align_kernel<float, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)nullptr, dfeat_f32, C, B, H, W, C,
                           rois, R, spatial_scale, P, sampling_ratio, aligned, (float*)nullptr, (const float*)dout);
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

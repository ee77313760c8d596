// Fused detection loss for gfx950: target assignment + CIoU box loss + objectness / class BCE, forward AND the
// gradient with respect to the logits, in a handful of launches per step.
//
// The reference does this with ~300 eager tensor ops per step plus their autograd backward
// (metayolo/models/yolo_head.py:358-417 matcher, metayolo/models/loss.py:190-244 DetLoss.forward,
// metayolo/models/utils_general.py:193-231 bbox_iou(CIoU)); on MI355X that was ~12 ms of host time per step
// around ~0.3 ms of device work.  Here, per pyramid level:
//
//   match_kernel   one lane per (offset variant j in 0..4, anchor a, target g): anchor-ratio test (< anchor_t), the
//                  centre cell and up to two neighbour cells (the 0.5-offset rule), then for each match
//                    - CIoU(pred box, target box) in forward-mode dual numbers -> d(1-CIoU)/d(4 box logits)
//                    - class BCE (pos_weight, class weights, label smoothing) and its gradient
//                    - atomic accumulation of the (unscaled) gradients into an fp32 scratch image,
//                    - the objectness target: atomicMax of (candidate order << 32 | iou bits), so that among
//                      candidates that hit the same cell the LAST one in the reference's enumeration order wins,
//                      which is what the reference's sequential CPU scatter does (deterministic, unlike index_put on a GPU)
//                    - per-level sums / counts (fp64 atomics)
//   dense_kernel   one lane per cell x anchor: objectness BCE against the scattered target (+ its gradient), scales the
//                  accumulated box / class gradients by gain/count, writes the logits gradient straight into the NHWC
//                  buffer the backward plan consumes (bf16 or fp32, zero padded channels)
//   final_kernel   loss = (box*sum_l mean_l + obj*sum_l balance_l*mean_l + cls*sum_l mean_l) * batch, and the three items
//
// Gradient accumulation uses fp32 atomics (a few thousand adds spread over the image): run-to-run differences are
// in the last bits only; every mean is taken over fp64 sums.
#include "common.h"

namespace {

constexpr int MAXL = 5, MAXA = 8, MAXC = 128;

template <typename T> struct Quad;                        // 4 consecutive channels in memory
template <> struct Quad<float> { typedef f32x4 type; };
template <> struct Quad<bf16_t> { typedef bf16x4 type; };

struct D4 {            // value + 4 partial derivatives (w.r.t. px, py, pw, ph)
    float v, d[4];
};
__device__ __forceinline__ D4 cst(float v) { return D4{v, {0.f, 0.f, 0.f, 0.f}}; }
__device__ __forceinline__ D4 var(float v, int i) { D4 r = cst(v); r.d[i] = 1.f; return r; }
__device__ __forceinline__ D4 operator+(const D4& a, const D4& b) { D4 r; r.v = a.v + b.v; for (int i = 0; i < 4; ++i) r.d[i] = a.d[i] + b.d[i]; return r; }
__device__ __forceinline__ D4 operator-(const D4& a, const D4& b) { D4 r; r.v = a.v - b.v; for (int i = 0; i < 4; ++i) r.d[i] = a.d[i] - b.d[i]; return r; }
__device__ __forceinline__ D4 operator*(const D4& a, const D4& b) { D4 r; r.v = a.v * b.v; for (int i = 0; i < 4; ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i]; return r; }
__device__ __forceinline__ D4 operator/(const D4& a, const D4& b) {
    D4 r; const float inv = 1.0f / b.v; r.v = a.v * inv;
    for (int i = 0; i < 4; ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) * inv;
    return r;
}
__device__ __forceinline__ D4 operator*(const D4& a, float s) { D4 r; r.v = a.v * s; for (int i = 0; i < 4; ++i) r.d[i] = a.d[i] * s; return r; }
__device__ __forceinline__ D4 dmin(const D4& a, const D4& b) { return a.v <= b.v ? a : b; }
__device__ __forceinline__ D4 dmax(const D4& a, const D4& b) { return a.v >= b.v ? a : b; }
__device__ __forceinline__ D4 clamp0(const D4& a) { return a.v > 0.f ? a : cst(0.f); }
__device__ __forceinline__ D4 datan(const D4& a) { D4 r; r.v = atanf(a.v); const float g = 1.0f / (1.0f + a.v * a.v); for (int i = 0; i < 4; ++i) r.d[i] = a.d[i] * g; return r; }

struct LossArgs {
    const float* logits[MAXL];   // [B][ny][nx][ldl], channel = a*no + o
    void* gdet[MAXL];            // [B][ny][nx][ldg]
    float* scratch[MAXL];        // [B][ny][nx][ldl] fp32 gradient accumulators (zeroed here)
    unsigned long long* tobj[MAXL];   // [B][na][ny][nx] packed (order+1)<<32 | iou bits (zeroed here)
    int ny[MAXL], nx[MAXL];
    float anc[MAXL][MAXA][2];    // anchors in grid units
    float balance[MAXL];
    int nl, B, na, nc, no, ldl, ldg, nt;
    const float* gts;            // [nt][5] img, cx, cy, w, h (normalised)
    const float* tcls;           // [nt][nc] class targets (one- or multi-hot)
    float cw[MAXC];
    float cls_pw, obj_pw, anchor_t, smooth;
    float h_box, h_obj, h_cls;
    double* acc;                 // [nl][6]: sum(1-ciou), n, sum cls bce, n_cls rows, sum obj bce, unused
    float* out;                  // loss, lbox, lobj, lcls
};

__device__ __forceinline__ float sigm(float x) { return 1.0f / (1.0f + expf(-x)); }
// BCE with logits and pos_weight: loss and d loss / d logit
__device__ __forceinline__ float bce(float x, float t, float pw, float* grad) {
    const float s = sigm(x);
    // log(sigmoid) and log(1 - sigmoid), stable
    const float ls = fminf(x, 0.f) - log1pf(expf(-fabsf(x)));
    const float l1s = ls - x;
    *grad = s * (1.f - t + pw * t) - pw * t;
    return -(pw * t * ls + (1.f - t) * l1s);
}

__global__ __launch_bounds__(256) void zero_kernel(uint4* p, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) p[i] = uint4{0, 0, 0, 0};
}

__global__ __launch_bounds__(256) void match_kernel(const LossArgs p, int l) {
    const int na = p.na, nt = p.nt, nc = p.nc, no = p.no;
    const int total = 5 * na * nt;
    const int ny = p.ny[l], nx = p.nx[l];
    double s_box = 0.0, s_cls = 0.0;
    int n_box = 0, n_cls = 0;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int j = idx / (na * nt);
        const int rem = idx - j * (na * nt);
        const int a = rem / nt, g = rem - a * nt;
        const float* gt = p.gts + (size_t)g * 5;
        const float gx = gt[1] * nx, gy = gt[2] * ny, gw = gt[3] * nx, gh = gt[4] * ny;
        const float aw = p.anc[l][a][0], ah = p.anc[l][a][1];
        const float rw = gw / aw, rh = gh / ah;
        const float worst = fmaxf(fmaxf(rw, 1.f / rw), fmaxf(rh, 1.f / rh));
        if (!(worst < p.anchor_t)) continue;
        float ox = 0.f, oy = 0.f;
        if (j == 1) { if (!(fmodf(gx, 1.f) < 0.5f && gx > 1.f)) continue; ox = 0.5f; }
        else if (j == 2) { if (!(fmodf(gy, 1.f) < 0.5f && gy > 1.f)) continue; oy = 0.5f; }
        else if (j == 3) { const float ix = nx - gx; if (!(fmodf(ix, 1.f) < 0.5f && ix > 1.f)) continue; ox = -0.5f; }
        else if (j == 4) { const float iy = ny - gy; if (!(fmodf(iy, 1.f) < 0.5f && iy > 1.f)) continue; oy = -0.5f; }
        int gi = (int)(gx - ox), gj = (int)(gy - oy);            // truncation toward zero, as .long()
        gi = min(max(gi, 0), nx - 1);
        gj = min(max(gj, 0), ny - 1);
        const int b = (int)gt[0];
        const size_t pix = ((size_t)b * ny + gj) * nx + gi;
        const float* lg = p.logits[l] + pix * p.ldl + a * no;
        float* gr = p.scratch[l] + pix * p.ldl + a * no;
        // ---- box: CIoU between the decoded prediction and the target, both relative to the cell
        const float s0 = sigm(lg[0]), s1 = sigm(lg[1]), s2 = sigm(lg[2]), s3 = sigm(lg[3]);
        const D4 x1 = var(s0 * 2.f - 0.5f, 0), y1 = var(s1 * 2.f - 0.5f, 1);
        // a saturated sigmoid (logit < -88) gives w or h == 0 and the CIoU aspect term 0 * inf = NaN in the gradient (the
        // reference's autograd has the same hole); a 1e-12 floor keeps everything finite and changes nothing measurable
        const D4 w1 = var(fmaxf(4.f * s2 * s2 * aw, 1e-12f), 2), h1 = var(fmaxf(4.f * s3 * s3 * ah, 1e-12f), 3);
        const D4 x2 = cst(gx - gi), y2 = cst(gy - gj), w2 = cst(gw), h2 = cst(gh);
        const float eps = 1e-7f;
        const D4 a_x1 = x1 - w1 * 0.5f, a_x2 = x1 + w1 * 0.5f, a_y1 = y1 - h1 * 0.5f, a_y2 = y1 + h1 * 0.5f;
        const D4 b_x1 = x2 - w2 * 0.5f, b_x2 = x2 + w2 * 0.5f, b_y1 = y2 - h2 * 0.5f, b_y2 = y2 + h2 * 0.5f;
        const D4 inter = clamp0(dmin(a_x2, b_x2) - dmax(a_x1, b_x1)) * clamp0(dmin(a_y2, b_y2) - dmax(a_y1, b_y1));
        const D4 uni = w1 * h1 + w2 * h2 - inter + cst(eps);
        const D4 iou = inter / uni;
        const D4 cw_ = dmax(a_x2, b_x2) - dmin(a_x1, b_x1), ch_ = dmax(a_y2, b_y2) - dmin(a_y1, b_y1);
        const D4 c2 = cw_ * cw_ + ch_ * ch_ + cst(eps);
        const D4 dx = b_x1 + b_x2 - a_x1 - a_x2, dy = b_y1 + b_y2 - a_y1 - a_y2;
        const D4 rho2 = (dx * dx + dy * dy) * 0.25f;
        const D4 da = datan(w2 / h2) - datan(w1 / h1);
        const D4 v = da * da * (4.0f / (3.14159265358979323846f * 3.14159265358979323846f));
        const float alpha = v.v / (v.v - iou.v + (1.f + eps));           // no gradient through alpha
        const D4 ciou = iou - (rho2 / c2 + v * alpha);
        s_box += (double)(1.f - ciou.v);
        ++n_box;
        // d(1 - ciou)/d logit = -dciou/dp * dp/dlogit
        const float dpl[4] = {2.f * s0 * (1.f - s0), 2.f * s1 * (1.f - s1), 8.f * s2 * aw * s2 * (1.f - s2), 8.f * s3 * ah * s3 * (1.f - s3)};
#pragma unroll
        for (int i = 0; i < 4; ++i) atomicAdd(gr + i, -ciou.d[i] * dpl[i]);
        // ---- objectness target: last candidate in reference order wins
        const float t_iou = fmaxf(ciou.v, 0.f);
        const unsigned long long packed = ((unsigned long long)(idx + 1) << 32) | __float_as_uint(t_iou);
        atomicMax(p.tobj[l] + (((size_t)b * na + a) * ny + gj) * nx + gi, packed);
        // ---- classes
        if (nc > 1) {
            const float* tc = p.tcls + (size_t)g * nc;
            float any = 0.f;
            for (int c = 0; c < nc; ++c) any += tc[c];
            if (any > 0.f) {
                ++n_cls;
                for (int c = 0; c < nc; ++c) {
                    const float t = tc[c] - (tc[c] - 0.5f) * p.smooth;
                    float gd;
                    const float ls = bce(lg[5 + c], t, p.cls_pw, &gd);
                    s_cls += (double)(ls * p.cw[c]);
                    atomicAdd(gr + 5 + c, gd * p.cw[c]);
                }
            }
        }
    }
    // block reduction of the four statistics, one atomic each
    __shared__ double sh[4][256];
    sh[0][threadIdx.x] = s_box; sh[1][threadIdx.x] = (double)n_box; sh[2][threadIdx.x] = s_cls; sh[3][threadIdx.x] = (double)n_cls;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st)
            for (int q = 0; q < 4; ++q) sh[q][threadIdx.x] += sh[q][threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x < 4 && sh[threadIdx.x][0] != 0.0) atomicAdd(p.acc + l * 6 + threadIdx.x, sh[threadIdx.x][0]);
}

// One lane per (cell, 4-channel group): the accumulated gradients are read as 16-byte vectors and the logits gradient leaves as 8-
// (bf16) or 16-byte (fp32) vectors, lanes of a wave covering consecutive addresses.  (One lane per cell walking its 40 channels with
// scalar accesses 160 bytes apart took 87 us per level on average.)  Requires ldl % 4 == 0 and ldg % 4 == 0 (checked by the launcher).
template <typename T>
__global__ __launch_bounds__(256) void dense_kernel(const LossArgs p, int l) {
    const int na = p.na, no = p.no, ny = p.ny[l], nx = p.nx[l];
    const long long cells = (long long)p.B * ny * nx;
    const double n_box = p.acc[l * 6 + 1], n_cls = p.acc[l * 6 + 3];
    const float bs = (float)p.B;
    const float k_box = n_box > 0 ? bs * p.h_box / (float)n_box : 0.f;
    const float k_cls = n_cls > 0 ? bs * p.h_cls / (float)(n_cls * p.nc) : 0.f;
    const float k_obj = bs * p.h_obj * p.balance[l] / (float)(cells * na);
    T* gd = (T*)p.gdet[l];
    const int groups = p.ldg / 4;                          // 4-channel groups per cell (padding channels included)
    const long long items = cells * groups;
    double s_obj = 0.0;
    for (long long it = (long long)blockIdx.x * blockDim.x + threadIdx.x; it < items; it += (long long)gridDim.x * blockDim.x) {
        const long long pix = it / groups;
        const int c0 = (int)(it - pix * groups) * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (c0 < p.ldl) {
            const f32x4 gr = *(const f32x4*)(p.scratch[l] + pix * p.ldl + c0);
            int a = c0 / no, o = c0 - a * no;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (a < na) {
                    if (o == 4) {
                        const int x = (int)(pix % nx);
                        const long long t = pix / nx;
                        const int y = (int)(t % ny), b = (int)(t / ny);
                        const unsigned long long pk = p.tobj[l][(((size_t)b * na + a) * ny + y) * nx + x];
                        const float tgt = pk ? __uint_as_float((unsigned)(pk & 0xFFFFFFFFull)) : 0.f;
                        float go;
                        s_obj += (double)bce(p.logits[l][pix * p.ldl + c0 + i], tgt, p.obj_pw, &go);
                        v[i] = go * k_obj;
                    } else {
                        v[i] = gr[i] * (o < 4 ? k_box : k_cls);
                    }
                }
                if (++o == no) {
                    o = 0;
                    ++a;
                }
            }
        }
        typename Quad<T>::type out = {(T)v[0], (T)v[1], (T)v[2], (T)v[3]};
        *(typename Quad<T>::type*)(gd + pix * p.ldg + c0) = out;
    }
    __shared__ double sh[256];
    sh[threadIdx.x] = s_obj;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) sh[threadIdx.x] += sh[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) atomicAdd(p.acc + l * 6 + 4, sh[0]);
}

__global__ void final_kernel(const LossArgs p) {
    double lbox = 0.0, lobj = 0.0, lcls = 0.0;
    for (int l = 0; l < p.nl; ++l) {
        const double* a = p.acc + l * 6;
        if (a[1] > 0) lbox += a[0] / a[1];
        if (a[3] > 0) lcls += a[2] / (a[3] * p.nc);
        lobj += a[4] / ((double)p.B * p.na * p.ny[l] * p.nx[l]) * p.balance[l];
    }
    lbox *= p.h_box; lobj *= p.h_obj; lcls *= p.h_cls;
    p.out[0] = (float)((lbox + lobj + lcls) * p.B);
    p.out[1] = (float)lbox;
    p.out[2] = (float)lobj;
    p.out[3] = (float)lcls;
}

template <typename T>
__global__ __launch_bounds__(256) void scale_kernel(T* p, size_t n, const float* scale) {
    const float s = *scale;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        p[i] = from_f32<T>(to_f32<T>(p[i]) * s);
}

inline size_t a16(size_t v) { return (v + 15) / 16 * 16; }

}  // namespace

extern "C" {

size_t hdy_det_loss_workspace_bytes(int nl, const int* ny, const int* nx, int B, int na, int ldl) {
    size_t n = a16((size_t)MAXL * 6 * sizeof(double));
    for (int l = 0; l < nl; ++l) {
        const size_t cells = (size_t)B * ny[l] * nx[l];
        n += a16(cells * ldl * sizeof(float)) + a16(cells * na * sizeof(unsigned long long));
    }
    return n;
}

int hdy_det_loss(const float* const* logits, int ldl, void* const* gdet, int ldg, int dtype, const int* ny, const int* nx, int nl, int B,
                 int na, int nc, const float* anchors_grid, const float* balance, const float* gts, const float* tcls, int nt,
                 const float* cls_cw, float cls_pw, float obj_pw, float anchor_t, float label_smoothing, float h_box, float h_obj, float h_cls,
                 float* out, void* workspace, size_t ws_bytes, void* stream) {
    HDY_ARG(logits && gdet && ny && nx && anchors_grid && balance && cls_cw && out && workspace, "det_loss: null pointer");
    HDY_ARG(nl >= 1 && nl <= MAXL && na >= 1 && na <= MAXA && nc >= 1 && nc <= MAXC && B >= 1 && nt >= 0, "det_loss: bad sizes");
    HDY_ARG(nt == 0 || (gts && tcls), "det_loss: targets missing");
    const int no = nc + 5;
    HDY_ARG(ldl >= na * no && ldg >= na * no && ldl % 4 == 0 && ldg % 4 == 0, "det_loss: pitches too small or not multiples of 4");
    HDY_ARG(ws_bytes >= hdy_det_loss_workspace_bytes(nl, ny, nx, B, na, ldl) && ((uintptr_t)workspace & 15) == 0, "det_loss: workspace too small / unaligned");
    HDY_ARG(dtype == HDY_BF16 || dtype == HDY_F32, "det_loss: unknown dtype");
    HDY_ARG((long long)5 * na * nt < (1LL << 31), "det_loss: too many targets");
    LossArgs a = {};
    char* w = (char*)workspace;
    a.acc = (double*)w;
    w += a16((size_t)MAXL * 6 * sizeof(double));
    for (int l = 0; l < nl; ++l) {
        HDY_ARG(logits[l] && gdet[l] && ny[l] > 0 && nx[l] > 0, "det_loss: level %d missing", l);
        const size_t cells = (size_t)B * ny[l] * nx[l];
        a.logits[l] = logits[l]; a.gdet[l] = gdet[l]; a.ny[l] = ny[l]; a.nx[l] = nx[l]; a.balance[l] = balance[l];
        a.scratch[l] = (float*)w;
        w += a16(cells * ldl * sizeof(float));
        a.tobj[l] = (unsigned long long*)w;
        w += a16(cells * na * sizeof(unsigned long long));
        for (int i = 0; i < na; ++i) { a.anc[l][i][0] = anchors_grid[(l * na + i) * 2]; a.anc[l][i][1] = anchors_grid[(l * na + i) * 2 + 1]; }
    }
    a.nl = nl; a.B = B; a.na = na; a.nc = nc; a.no = no; a.ldl = ldl; a.ldg = ldg; a.nt = nt; a.gts = gts; a.tcls = tcls;
    for (int c = 0; c < nc; ++c) a.cw[c] = cls_cw[c];
    a.cls_pw = cls_pw; a.obj_pw = obj_pw; a.anchor_t = anchor_t; a.smooth = label_smoothing;
    a.h_box = h_box; a.h_obj = h_obj; a.h_cls = h_cls; a.out = out;
    hipStream_t st = (hipStream_t)stream;
    const size_t used = (size_t)(w - (char*)workspace);
    hipLaunchKernelGGL(zero_kernel, dim3(2048), dim3(256), 0, st, (uint4*)workspace, used / 16);
    HDY_LAUNCH_CHECK("det_loss zero");
    for (int l = 0; l < nl && nt > 0; ++l) {
        const int total = 5 * na * nt;
        hipLaunchKernelGGL(match_kernel, dim3(cdiv(total, 256) < 1024 ? cdiv(total, 256) : 1024), dim3(256), 0, st, a, l);
        HDY_LAUNCH_CHECK("det_loss match");
    }
    for (int l = 0; l < nl; ++l) {
        const long long items = (long long)B * ny[l] * nx[l] * (ldg / 4);
        const int grid = (int)((items + 255) / 256 < 4096 ? (items + 255) / 256 : 4096);
        if (dtype == HDY_BF16) hipLaunchKernelGGL(dense_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, a, l);
        else hipLaunchKernelGGL(dense_kernel<float>, dim3(grid), dim3(256), 0, st, a, l);
        HDY_LAUNCH_CHECK("det_loss dense");
    }
    hipLaunchKernelGGL(final_kernel, dim3(1), dim3(1), 0, st, a);
    HDY_LAUNCH_CHECK("det_loss final");
    return HDY_OK;
}

// p[i] *= *scale (device scalar): applies the upstream gradient of the loss to the stored logits gradient
int hdy_scale_inplace(void* p, long long n, const float* scale, int dtype, void* stream) {
    HDY_ARG(p && scale && n > 0, "scale_inplace: bad args");
    const int grid = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    if (dtype == HDY_BF16) hipLaunchKernelGGL(scale_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (bf16_t*)p, (size_t)n, scale);
    else hipLaunchKernelGGL(scale_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (float*)p, (size_t)n, scale);
    HDY_LAUNCH_CHECK("scale_inplace");
    return HDY_OK;
}

}  // extern "C"

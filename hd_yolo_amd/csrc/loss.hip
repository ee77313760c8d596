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
//                    - one RECORD per match (4 box gradients, iou, candidate order, class gradients), appended to a dense array and
//                      chained to its (cell, anchor) with one atomicExch on that cell's list head.  (First version: 12 fp32
//                      atomicAdds into an image-sized scratch + a 64-bit atomicMax per match — 2.3 M device-scope atomics per step were
//                      100 of the kernel's 166 us — and the scratch image had to be zeroed and re-read in full.)
//                    - per-level sums / counts (fp64 atomics, one per workgroup)
//   dense_kernel   one lane per (cell, 4 channels): walks the cell's record lists (mostly empty), sums the box / class gradients,
//                  takes the objectness target from the record with the HIGHEST candidate order — among candidates that hit the same
//                  cell the LAST one in the reference's enumeration order wins, which is what the reference's sequential CPU scatter
//                  does (deterministic, unlike index_put on a GPU) — objectness BCE (+ its gradient), scales by gain/count, writes the
//                  logits gradient straight into the NHWC buffer the backward plan consumes (bf16 or fp32, zero padded channels)
//   final_kernel   loss = (box*sum_l mean_l + obj*sum_l balance_l*mean_l + cls*sum_l mean_l) * batch, and the three items
//
// Records of one cell are summed in list order, which depends on the order the matches arrived in: run-to-run differences are in the
// last bits only (and only where three or more matches share a cell and anchor); every mean is taken over fp64 sums.
#include <stdlib.h>

#include "common.h"

namespace {

constexpr int MAXL = 5, MAXA = 8, MAXC = 128;
constexpr int OSL = 64;      // objectness-sum slots per level: a workgroup adds to slot (its index % OSL); atomics on one address serialise

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
    int* head[MAXL];             // [B][ny][nx][na] list heads: record index + 1, 0 = empty (zeroed here)
    float* recs;                 // [capacity][rs] match records: 0-3 box gradients, 4 iou, 5 candidate order, 6 next (index + 1), 8.. class gradients
    int* nrec;                   // records in use (zeroed here)
    int rs, capacity;
    int ny[MAXL], nx[MAXL];
    float anc[MAXL][MAXA][2];    // anchors in grid units
    float balance[MAXL];
    int nl, B, na, nc, no, ldl, ldg, nt;
    const float* gts;            // [nt][5] img, cx, cy, w, h (normalised)
    const float* tcls;           // [nt][nc] class targets (one- or multi-hot)
    float cw[MAXC];
    float cls_pw, obj_pw, anchor_t, smooth;
    float h_box, h_obj, h_cls;
    double* acc;                 // [MAXL][6]: sum(1-ciou), n, sum cls bce, n_cls rows, unused x2; then [MAXL][OSL] partial sums of the obj bce
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

// grid.y = pyramid level: the levels are independent and each is a chain of dependent global accesses per lane (target row -> logits ->
// atomics), so they run side by side in one launch (three launches back to back: 178 us per yolov5s step)
__global__ __launch_bounds__(256) void match_kernel(const LossArgs p) {
    const int l = blockIdx.y;
    const int na = p.na, nt = p.nt, nc = p.nc, no = p.no;
    const int total = 5 * na * nt;
    const int ny = p.ny[l], nx = p.nx[l];
    double s_box = 0.0, s_cls = 0.0;
    int n_box = 0, n_cls = 0;
    // Only ~30 % of the (offset, anchor, target) candidates pass the anchor-ratio and the half-cell tests, in no pattern: evaluated in
    // place, every wave ran the whole CIoU / BCE body for its few live lanes.  So a workgroup first runs the cheap tests for 1024
    // candidates, packs the survivors into an LDS list, and then works through the list with full waves.
    constexpr int CPT = 4;                                  // candidates per thread and trip
    __shared__ int list[256 * CPT];
    __shared__ int wcount[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    auto cell_of = [&](int idx, int& gi, int& gj, int& a, int& g) -> bool {
        const int j = idx / (na * nt);
        const int rem = idx - j * (na * nt);
        a = rem / nt;
        g = rem - a * nt;
        const float* gt = p.gts + (size_t)g * 5;
        const float gx = gt[1] * nx, gy = gt[2] * ny, gw = gt[3] * nx, gh = gt[4] * ny;
        const float aw = p.anc[l][a][0], ah = p.anc[l][a][1];
        const float rw = gw / aw, rh = gh / ah;
        const float worst = fmaxf(fmaxf(rw, 1.f / rw), fmaxf(rh, 1.f / rh));
        if (!(worst < p.anchor_t)) return false;
        float ox = 0.f, oy = 0.f;
        if (j == 1) { if (!(fmodf(gx, 1.f) < 0.5f && gx > 1.f)) return false; ox = 0.5f; }
        else if (j == 2) { if (!(fmodf(gy, 1.f) < 0.5f && gy > 1.f)) return false; oy = 0.5f; }
        else if (j == 3) { const float ix = nx - gx; if (!(fmodf(ix, 1.f) < 0.5f && ix > 1.f)) return false; ox = -0.5f; }
        else if (j == 4) { const float iy = ny - gy; if (!(fmodf(iy, 1.f) < 0.5f && iy > 1.f)) return false; oy = -0.5f; }
        gi = (int)(gx - ox);                                     // truncation toward zero, as .long()
        gj = (int)(gy - oy);
        gi = min(max(gi, 0), nx - 1);
        gj = min(max(gj, 0), ny - 1);
        return true;
    };
    for (int base = blockIdx.x * 256 * CPT; base < total; base += gridDim.x * 256 * CPT) {
        bool ok[CPT];
        int nmine = 0;
#pragma unroll
        for (int k = 0; k < CPT; ++k) {
            const int idx = base + k * 256 + threadIdx.x;
            int gi, gj, a, g;
            ok[k] = idx < total && cell_of(idx, gi, gj, a, g);
            nmine += ok[k];
        }
        // exclusive prefix of nmine over the workgroup: lanes below in the wave, then the waves below
        int below = 0;
#pragma unroll
        for (int k = 1; k <= CPT; ++k) below += k * __popcll(__ballot(nmine == k) & ((1ull << lane) - 1ull));
        int wtot = 0;
#pragma unroll
        for (int k = 1; k <= CPT; ++k) wtot += k * __popcll(__ballot(nmine == k));
        if (lane == 0) wcount[wave] = wtot;
        __syncthreads();
        int nlist = 0;
        for (int w = 0; w < 4; ++w) {
            if (w < wave) below += wcount[w];
            nlist += wcount[w];
        }
#pragma unroll
        for (int k = 0; k < CPT; ++k)
            if (ok[k]) list[below++] = base + k * 256 + threadIdx.x;
        __syncthreads();
        if (threadIdx.x == 0) wcount[0] = nlist ? atomicAdd(p.nrec, nlist) : 0;      // this trip's records: [first, first + nlist)
        __syncthreads();
        const int first = wcount[0];
        for (int t = threadIdx.x; t < nlist && first + t < p.capacity; t += 256) {
        const int idx = list[t];
        int gi, gj, a, g;
        cell_of(idx, gi, gj, a, g);
        const float* gt = p.gts + (size_t)g * 5;
        const float gx = gt[1] * nx, gy = gt[2] * ny, gw = gt[3] * nx, gh = gt[4] * ny;
        const float aw = p.anc[l][a][0], ah = p.anc[l][a][1];
        const int b = (int)gt[0];
        const size_t pix = ((size_t)b * ny + gj) * nx + gi;
        const float* lg = p.logits[l] + pix * p.ldl + a * no;
        float* rec = p.recs + (size_t)(first + t) * p.rs;
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
        // ---- the record: box gradients, objectness target (iou) and the candidate order that decides between targets of one cell
        *(f32x4*)rec = f32x4{-ciou.d[0] * dpl[0], -ciou.d[1] * dpl[1], -ciou.d[2] * dpl[2], -ciou.d[3] * dpl[3]};
        const int prev = atomicExch(p.head[l] + pix * na + a, first + t + 1);
        *(f32x4*)(rec + 4) = f32x4{fmaxf(ciou.v, 0.f), __int_as_float(idx), __int_as_float(prev), 0.f};
        // ---- classes
        if (nc > 1) {
            const float* tc = p.tcls + (size_t)g * nc;
            float any = 0.f;
            for (int c = 0; c < nc; ++c) any += tc[c];
            if (any > 0.f) ++n_cls;
            for (int c = 0; c < nc; ++c) {
                float gd = 0.f;
                if (any > 0.f) {
                    const float tt = tc[c] - (tc[c] - 0.5f) * p.smooth;
                    const float ls = bce(lg[5 + c], tt, p.cls_pw, &gd);
                    s_cls += (double)(ls * p.cw[c]);
                    gd *= p.cw[c];
                }
                rec[8 + c] = gd;
            }
        }
        }
        __syncthreads();                                        // the list is rewritten by the next trip
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

// One lane per (cell, 4-channel group); the logits gradient leaves as 8- (bf16) or 16-byte (fp32) vectors, lanes of a wave covering
// consecutive addresses.  A group's four channels belong to one or two anchors; for each the lane walks that (cell, anchor)'s record
// list — empty for ~96 % of them — adding the box / class gradients of its channels and, for an objectness channel, keeping the iou of
// the record with the highest candidate order.  Requires ldl % 4 == 0 and ldg % 4 == 0 (checked by the launcher).
// grid.y = pyramid level (one launch for all levels); index arithmetic in 32 bits (the launcher checks cells * groups < 2^31).
template <typename T>
__global__ __launch_bounds__(256) void dense_kernel(const LossArgs p) {
    const int l = blockIdx.y;
    const int na = p.na, no = p.no, ny = p.ny[l], nx = p.nx[l];
    const int cells = p.B * ny * nx;
    const double n_box = p.acc[l * 6 + 1], n_cls = p.acc[l * 6 + 3];
    const float bs = (float)p.B;
    const float k_box = n_box > 0 ? bs * p.h_box / (float)n_box : 0.f;
    const float k_cls = (n_cls > 0 && p.nc > 1) ? bs * p.h_cls / (float)(n_cls * p.nc) : 0.f;
    const float k_obj = bs * p.h_obj * p.balance[l] / (float)((long long)cells * na);
    T* gd = (T*)p.gdet[l];
    const unsigned groups = p.ldg / 4;                     // 4-channel groups per cell (padding channels included)
    const unsigned items = (unsigned)cells * groups;
    const float* __restrict__ logits = p.logits[l];
    const int* __restrict__ head = p.head[l];
    double s_obj = 0.0;
    for (unsigned it = blockIdx.x * blockDim.x + threadIdx.x; it < items; it += gridDim.x * blockDim.x) {
        const unsigned pix = it / groups;
        const int c0 = (int)(it - pix * groups) * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        const int a0 = (unsigned)c0 / (unsigned)no, o0 = c0 - a0 * no;
        if (a0 < na) {
            // channel i of the group is output o0 + i of anchor a0, or output o0 + i - no of anchor a0 + 1 (no >= 6: at most two anchors,
            // at most one objectness logit: channel i4 of anchor a4 — its BCE is evaluated once, after the walk)
            const bool two = o0 + 3 >= no && a0 + 1 < na;
            int i4 = 4 - o0;
            if (i4 < 0) i4 += no;
            const int a4 = a0 + (o0 + i4 >= no ? 1 : 0);
            const bool has_obj = i4 < 4 && a4 < na;
            float tgt = 0.f;
            int best = -1;
            int r0 = head[(size_t)pix * na + a0];
            int r1 = two ? head[(size_t)pix * na + a0 + 1] : 0;
            // the objectness logit travels with the list heads (its address does not depend on them): requested after the walk it was a second memory
            // round trip per item for the third of the groups that hold an objectness channel
            const float lg_obj = logits[(size_t)pix * p.ldl + c0 + (has_obj ? i4 : 0)];
            if (r0 | r1) {
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const int ob = half ? o0 - no : o0;                    // output index of channel 0 relative to this anchor
                    for (int r = half ? r1 : r0; r != 0;) {
                        const float* rec = p.recs + (size_t)(r - 1) * p.rs;
                        const f32x4 h1 = *(const f32x4*)(rec + 4);
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int o = ob + i;
                            if (o < 0 || o >= no) continue;
                            if (o < 4) v[i] += rec[o];
                            else if (o == 4) {
                                const int order = __float_as_int(h1[1]);
                                if (order > best) { best = order; tgt = h1[0]; }
                            } else if (p.nc > 1) v[i] += rec[8 + o - 5];
                        }
                        r = __float_as_int(h1[2]);
                    }
                }
            }
            float g_obj = 0.f;
            if (has_obj) {
                float go;
                s_obj += (double)bce(lg_obj, tgt, p.obj_pw, &go);
                g_obj = go * k_obj;
            }
            int a = a0, o = o0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                v[i] = a < na ? (o == 4 ? g_obj : v[i] * (o < 4 ? k_box : k_cls)) : 0.f;
                if (++o == no) {
                    o = 0;
                    ++a;
                }
            }
        }
        typename Quad<T>::type out = {(T)v[0], (T)v[1], (T)v[2], (T)v[3]};
        *(typename Quad<T>::type*)(gd + (size_t)pix * p.ldg + c0) = out;
    }
    __shared__ double sh[256];
    sh[threadIdx.x] = s_obj;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) sh[threadIdx.x] += sh[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0 && sh[0] != 0.0) atomicAdd(p.acc + MAXL * 6 + l * OSL + (blockIdx.x % OSL), sh[0]);
}

__global__ void final_kernel(const LossArgs p) {
    double lbox = 0.0, lobj = 0.0, lcls = 0.0;
    for (int l = 0; l < p.nl; ++l) {
        const double* a = p.acc + l * 6;
        if (a[1] > 0) lbox += a[0] / a[1];
        if (a[3] > 0) lcls += a[2] / (a[3] * p.nc);
        double so = 0.0;
        for (int i = 0; i < OSL; ++i) so += p.acc[MAXL * 6 + l * OSL + i];
        lobj += so / ((double)p.B * p.na * p.ny[l] * p.nx[l]) * p.balance[l];
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
    if (s == 1.0f) return;          // the usual upstream gradient of `loss.backward()`: x * 1.0f == x for every bf16 / fp32 value, so nothing to do (20 us for the 43 MB of logits gradients)
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        p[i] = from_f32<T>(to_f32<T>(p[i]) * s);
}

// Target rows of hdy_det_loss from the batch's concatenated annotation tensors: gts[t] = (img, cx, cy, w, h) of the corner box,
// tcls[t][c] = 1 if label == c + 1 (labels outside 1..nc select no class: column 0 of the reference's one-hot, dropped by its [:, 1:]).
__global__ __launch_bounds__(256) void det_targets_kernel(const float* __restrict__ boxes, const float* __restrict__ img,
                                                          const long long* __restrict__ labels, int nt, int nc, float* __restrict__ gts,
                                                          float* __restrict__ tcls) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= nt) return;
    const f32x4 b = *(const f32x4*)(boxes + (size_t)t * 4);
    float* g = gts + (size_t)t * 5;
    g[0] = img[t];
    g[1] = (b[0] + b[2]) / 2;
    g[2] = (b[1] + b[3]) / 2;
    g[3] = b[2] - b[0];
    g[4] = b[3] - b[1];
    const long long lab = labels[t];
    for (int c = 0; c < nc; ++c) tcls[(size_t)t * nc + c] = lab == c + 1 ? 1.f : 0.f;
}

inline size_t a16(size_t v) { return (v + 15) / 16 * 16; }

// ---------------------------------------------------------------- which matched cell feeds the mask branch
// Reference (metayolo/models/yolo_head.py:231-262): of all cells matched to a target only the one whose decoded box has the best IoU with the
// truth goes through the mask head, and only when that IoU >= 0.8 (torch_scatter.scatter_max: the FIRST row attaining the maximum, rows
// ordered level by level and inside a level by (offset variant, anchor, target), which is the candidate index of match_kernel).  The
// tensor-expression version re-ran the matcher and decoded every level on the host side of ~20 device-to-host syncs per step.
// Here: mask_iou_kernel enumerates the same candidates, decodes the matched cell with decode_kernel's arithmetic (bit-identical boxes),
// takes the IoU in input pixels as paired_box_iou does, and keeps per target max(iou bits << 32 | ~order) with one 64-bit atomicMax;
// mask_pick_kernel (one workgroup) compacts the targets that pass, in target order, into per-level roi lists.
struct MaskSelArgs {
    const float* logits[MAXL];   // [B][ny][nx][ldl], channel = a*no + o
    int ny[MAXL], nx[MAXL];
    float anc[MAXL][MAXA][2];    // anchors in grid units (the matcher's ratio test)
    float anc_px[MAXL][MAXA][2]; // anchors in input pixels (decode)
    float stride[MAXL];
    int nl, B, na, no, ldl, nt;
    const float* gts;            // [nt][5] img, cx, cy, w, h (normalised)
    float anchor_t, min_iou;
    unsigned long long* key;     // [nt], zeroed by the caller
};

__global__ __launch_bounds__(256) void mask_iou_kernel(const MaskSelArgs p) {
    const int l = blockIdx.y;
    const int na = p.na, nt = p.nt;
    const int total = 5 * na * nt;
    const int ny = p.ny[l], nx = p.nx[l];
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
        const int j = idx / (na * nt);
        const int rem = idx - j * (na * nt);
        const int a = rem / nt, g = rem - a * nt;
        const float* gt = p.gts + (size_t)g * 5;
        const float gx = gt[1] * nx, gy = gt[2] * ny, gw = gt[3] * nx, gh = gt[4] * ny;
        const float rw = gw / p.anc[l][a][0], rh = gh / p.anc[l][a][1];
        const float worst = fmaxf(fmaxf(rw, 1.f / rw), fmaxf(rh, 1.f / rh));
        if (!(worst < p.anchor_t)) continue;
        float ox = 0.f, oy = 0.f;
        if (j == 1) { if (!(fmodf(gx, 1.f) < 0.5f && gx > 1.f)) continue; ox = 0.5f; }
        else if (j == 2) { if (!(fmodf(gy, 1.f) < 0.5f && gy > 1.f)) continue; oy = 0.5f; }
        else if (j == 3) { const float ix = nx - gx; if (!(fmodf(ix, 1.f) < 0.5f && ix > 1.f)) continue; ox = -0.5f; }
        else if (j == 4) { const float iy = ny - gy; if (!(fmodf(iy, 1.f) < 0.5f && iy > 1.f)) continue; oy = -0.5f; }
        int gi = (int)(gx - ox), gj = (int)(gy - oy);
        gi = min(max(gi, 0), nx - 1);
        gj = min(max(gj, 0), ny - 1);
        const int b = (int)gt[0];
        const float* lp = p.logits[l] + (((size_t)b * ny + gj) * nx + gi) * p.ldl + a * p.no;
        float sg[4];
#pragma unroll
        for (int o = 0; o < 4; ++o) sg[o] = __builtin_amdgcn_rcpf(1.0f + __expf(-lp[o]));          // decode_kernel's sigmoid
        const float st = p.stride[l];
        const float cx = (sg[0] * 2.0f - 0.5f + (float)gi) * st, cy = (sg[1] * 2.0f - 0.5f + (float)gj) * st;
        const float qw = sg[2] * 2.0f, qh = sg[3] * 2.0f;
        const float w = qw * qw * p.anc_px[l][a][0], h = qh * qh * p.anc_px[l][a][1];
        // xywh2xyxy of both boxes, the truth scaled to input pixels afterwards (yolo_head.py:243-244)
        const float px1 = cx - w / 2, py1 = cy - h / 2, px2 = cx + w / 2, py2 = cy + h / 2;
        const float tx1 = (gx - gw / 2) * st, ty1 = (gy - gh / 2) * st, tx2 = (gx + gw / 2) * st, ty2 = (gy + gh / 2) * st;
        const float iw = fmaxf(fminf(px2, tx2) - fmaxf(px1, tx1), 0.f), ih = fmaxf(fminf(py2, ty2) - fmaxf(py1, ty1), 0.f);
        const float inter = iw * ih;
        const float a1 = (px2 - px1) * (py2 - py1), a2 = (tx2 - tx1) * (ty2 - ty1);
        const float iou = inter / (a1 + a2 - inter);
        if (!(iou >= 0.f)) continue;                              // NaN (two empty boxes) never wins
        const unsigned order = (unsigned)(l * total + idx);
        const unsigned long long k = ((unsigned long long)__float_as_uint(iou) << 32) | (unsigned long long)(0xFFFFFFFFu - order);
        atomicMax(p.key + g, k);
    }
}

// one workgroup: the targets whose best cell passes, in target order.  counts[0] = kept, counts[1 + l] = kept at level l;
// keep_t[k] = target of kept row k; rois[l][i] = (image, x1, y1, x2, y2 in input pixels) of the i-th kept row of level l;
// order[k] = position of kept row k in the level-by-level concatenation of those lists.
__global__ __launch_bounds__(256) void mask_pick_kernel(const MaskSelArgs p, int* __restrict__ counts, long long* __restrict__ keep_t,
                                                        float* __restrict__ rois, long long* __restrict__ order, int* __restrict__ tmp) {
    __shared__ int wsum[MAXL + 1][4];
    __shared__ int base[MAXL + 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nt = p.nt, nl = p.nl, per_level = 5 * p.na * nt;
    if (tid <= MAXL) base[tid] = 0;
    __syncthreads();
    for (int g0 = 0; g0 < nt; g0 += 256) {
        const int g = g0 + tid;
        int lev = -1;
        if (g < nt) {
            const unsigned long long k = p.key[g];
            const float iou = __uint_as_float((unsigned)(k >> 32));
            if (k != 0ull && iou >= p.min_iou) lev = (int)((0xFFFFFFFFu - (unsigned)k) / (unsigned)per_level);
        }
        int pos[MAXL + 1];                                        // [nl]: position among all kept; [l]: among level l's
#pragma unroll
        for (int q = 0; q <= MAXL; ++q) {
            if (q < nl || q == MAXL) {
                const bool m = q == MAXL ? lev >= 0 : lev == q;
                const unsigned long long bal = __ballot(m);
                pos[q] = __popcll(bal & ((1ull << lane) - 1ull));
                if (lane == 0) wsum[q][wave] = __popcll(bal);
            }
        }
        __syncthreads();
        if (lev >= 0) {
            int pk = base[MAXL] + pos[MAXL], pl = base[lev];
#pragma unroll
            for (int q = 0; q < MAXL; ++q) if (q == lev) pl += pos[q];
            for (int w = 0; w < wave; ++w) { pk += wsum[MAXL][w]; pl += wsum[lev][w]; }
            keep_t[pk] = g;
            tmp[2 * pk] = lev;
            tmp[2 * pk + 1] = pl;
            const float* gt = p.gts + (size_t)g * 5;
            const float st = p.stride[lev];
            const float gx = gt[1] * p.nx[lev], gy = gt[2] * p.ny[lev], gw = gt[3] * p.nx[lev], gh = gt[4] * p.ny[lev];
            float* r = rois + ((size_t)lev * nt + pl) * 5;
            r[0] = gt[0];
            r[1] = (gx - gw / 2) * st; r[2] = (gy - gh / 2) * st; r[3] = (gx + gw / 2) * st; r[4] = (gy + gh / 2) * st;
        }
        __syncthreads();
        if (tid <= MAXL && (tid < nl || tid == MAXL)) base[tid] += wsum[tid][0] + wsum[tid][1] + wsum[tid][2] + wsum[tid][3];
        __syncthreads();
    }
    const int nk = base[MAXL];
    for (int k = tid; k < nk; k += 256) {
        int off = 0;
        for (int q = 0; q < tmp[2 * k]; ++q) off += base[q];
        order[k] = off + tmp[2 * k + 1];
    }
    if (tid == 0) counts[0] = nk;
    if (tid < nl) counts[1 + tid] = base[tid];
}

}  // namespace

extern "C" {

static int det_loss_record_floats(int nc) { return (8 + nc + 3) / 4 * 4; }

// nt = the largest number of targets (rows of gts) a call will be given with this workspace
size_t hdy_det_loss_workspace_bytes(int nl, const int* ny, const int* nx, int B, int na, int nc, int nt) {
    size_t n = a16((size_t)MAXL * (6 + OSL) * sizeof(double) + sizeof(int));
    for (int l = 0; l < nl; ++l) n += a16((size_t)B * ny[l] * nx[l] * na * sizeof(int));
    return n + a16((size_t)nl * 5 * na * (size_t)nt * det_loss_record_floats(nc) * sizeof(float));
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
    HDY_ARG(ws_bytes >= hdy_det_loss_workspace_bytes(nl, ny, nx, B, na, nc, nt) && ((uintptr_t)workspace & 15) == 0,
            "det_loss: workspace too small for %d targets / unaligned", nt);
    HDY_ARG(dtype == HDY_BF16 || dtype == HDY_F32, "det_loss: unknown dtype");
    HDY_ARG((long long)nl * 5 * na * nt < (1LL << 31), "det_loss: too many targets");
    for (int l = 0; l < nl; ++l) HDY_ARG((long long)B * ny[l] * nx[l] * (ldg / 4) < (1LL << 31), "det_loss: level %d has too many cells", l);
    LossArgs a = {};
    char* w = (char*)workspace;
    a.acc = (double*)w;
    a.nrec = (int*)(w + (size_t)MAXL * (6 + OSL) * sizeof(double));
    w += a16((size_t)MAXL * (6 + OSL) * sizeof(double) + sizeof(int));
    for (int l = 0; l < nl; ++l) {
        HDY_ARG(logits[l] && gdet[l] && ny[l] > 0 && nx[l] > 0, "det_loss: level %d missing", l);
        const size_t cells = (size_t)B * ny[l] * nx[l];
        a.logits[l] = logits[l]; a.gdet[l] = gdet[l]; a.ny[l] = ny[l]; a.nx[l] = nx[l]; a.balance[l] = balance[l];
        a.head[l] = (int*)w;
        w += a16(cells * na * sizeof(int));
        for (int i = 0; i < na; ++i) { a.anc[l][i][0] = anchors_grid[(l * na + i) * 2]; a.anc[l][i][1] = anchors_grid[(l * na + i) * 2 + 1]; }
    }
    a.nl = nl; a.B = B; a.na = na; a.nc = nc; a.no = no; a.ldl = ldl; a.ldg = ldg; a.nt = nt; a.gts = gts; a.tcls = tcls;
    for (int c = 0; c < nc; ++c) a.cw[c] = cls_cw[c];
    a.cls_pw = cls_pw; a.obj_pw = obj_pw; a.anchor_t = anchor_t; a.smooth = label_smoothing;
    a.h_box = h_box; a.h_obj = h_obj; a.h_cls = h_cls; a.out = out;
    hipStream_t st = (hipStream_t)stream;
    const size_t used = (size_t)(w - (char*)workspace);          // sums, record counter, list heads: zeroed; the records behind them are not
    a.recs = (float*)w;
    a.rs = det_loss_record_floats(nc);
    a.capacity = nl * 5 * na * nt;
    hipLaunchKernelGGL(zero_kernel, dim3(used / 16 / 256 < 2048 ? (unsigned)(used / 16 / 256 + 1) : 2048u), dim3(256), 0, st, (uint4*)workspace, used / 16);
    HDY_LAUNCH_CHECK("det_loss zero");
    if (nt > 0) {
        const int total = 5 * na * nt;
        hipLaunchKernelGGL(match_kernel, dim3(cdiv(total, 1024) < 1024 ? cdiv(total, 1024) : 1024, nl), dim3(256), 0, st, a);
        HDY_LAUNCH_CHECK("det_loss match");
    }
    long long most = 0;
    for (int l = 0; l < nl; ++l) {
        const long long items = (long long)B * ny[l] * nx[l] * (ldg / 4);
        if (items > most) most = items;
    }
    // the largest level sets grid.x, the others stride less.  Every workgroup ends with one fp64 atomic and atomics on one address
    // serialise (~50-100 ns each): with one sum per level, 4096 workgroups made that tail longer than the pass itself (140 us; 71 us
    // with 1024 workgroups) — hence the OSL slots per level
    const int dense_grid = hdy_opt(HDY_OPT_LOSS_GRID);
    const int grid = (int)((most + 255) / 256 < dense_grid ? (most + 255) / 256 : dense_grid);
    if (dtype == HDY_BF16) hipLaunchKernelGGL(dense_kernel<bf16_t>, dim3(grid, nl), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(dense_kernel<float>, dim3(grid, nl), dim3(256), 0, st, a);
    HDY_LAUNCH_CHECK("det_loss dense");
    hipLaunchKernelGGL(final_kernel, dim3(1), dim3(1), 0, st, a);
    HDY_LAUNCH_CHECK("det_loss final");
    return HDY_OK;
}

int hdy_det_targets(const float* boxes, const float* img, const long long* labels, int nt, int nc, float* gts, float* tcls, void* stream) {
    HDY_ARG(nt >= 0 && nc >= 1 && nc <= MAXC, "det_targets: bad sizes");
    if (nt == 0) return HDY_OK;
    HDY_ARG(boxes && img && labels && gts && tcls && ((uintptr_t)boxes & 15) == 0, "det_targets: null / unaligned pointer");
    hipLaunchKernelGGL(det_targets_kernel, dim3(cdiv(nt, 256)), dim3(256), 0, (hipStream_t)stream, boxes, img, labels, nt, nc, gts, tcls);
    HDY_LAUNCH_CHECK("det_targets");
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

// Mask-branch selection for one batch (see mask_iou_kernel): logits / ny / nx / anchors_grid as for hdy_det_loss, anchors_px [nl][na][2] and
// strides [nl] as for hdy_decode.  Device outputs: counts [1 + nl] ints, keep_t [nt] int64, rois [nl][nt][5] floats, order [nt] int64.
// workspace: nt * 16 bytes (8-byte aligned).
int hdy_mask_select(const float* const* logits, int ldl, const int* ny, const int* nx, int nl, int B, int na, int no, const float* anchors_grid,
                    const float* anchors_px, const float* strides, const float* gts, int nt, float anchor_t, float min_iou, int* counts,
                    long long* keep_t, float* rois, long long* order, void* workspace, size_t ws_bytes, void* stream) {
    HDY_ARG(logits && ny && nx && anchors_grid && anchors_px && strides && counts, "mask_select: null pointer");
    HDY_ARG(nl >= 1 && nl <= MAXL && na >= 1 && na <= MAXA && B >= 1 && nt >= 0 && no >= 5 && ldl >= na * no, "mask_select: bad sizes");
    HDY_ARG((long long)nl * 5 * na * nt < (1LL << 31), "mask_select: too many targets");
    hipStream_t st = (hipStream_t)stream;
    if (nt == 0) {
        HDY_ARG(hipMemsetAsync(counts, 0, (size_t)(1 + nl) * sizeof(int), st) == hipSuccess, "mask_select: memset failed");
        return HDY_OK;
    }
    HDY_ARG(gts && keep_t && rois && order && workspace && ws_bytes >= (size_t)nt * 16 && ((uintptr_t)workspace & 7) == 0, "mask_select: outputs / workspace missing or too small");
    MaskSelArgs a = {};
    for (int l = 0; l < nl; ++l) {
        HDY_ARG(logits[l] && ny[l] > 0 && nx[l] > 0, "mask_select: level %d missing", l);
        a.logits[l] = logits[l]; a.ny[l] = ny[l]; a.nx[l] = nx[l]; a.stride[l] = strides[l];
        for (int i = 0; i < na; ++i)
            for (int c = 0; c < 2; ++c) {
                a.anc[l][i][c] = anchors_grid[(l * na + i) * 2 + c];
                a.anc_px[l][i][c] = anchors_px[(l * na + i) * 2 + c];
            }
    }
    a.nl = nl; a.B = B; a.na = na; a.no = no; a.ldl = ldl; a.nt = nt; a.gts = gts; a.anchor_t = anchor_t; a.min_iou = min_iou;
    a.key = (unsigned long long*)workspace;
    HDY_ARG(hipMemsetAsync(a.key, 0, (size_t)nt * 8, st) == hipSuccess, "mask_select: memset failed");
    const int total = 5 * na * nt;
    hipLaunchKernelGGL(mask_iou_kernel, dim3(cdiv(total, 256) < 1024 ? cdiv(total, 256) : 1024, nl), dim3(256), 0, st, a);
    HDY_LAUNCH_CHECK("mask_select iou");
    hipLaunchKernelGGL(mask_pick_kernel, dim3(1), dim3(256), 0, st, a, counts, keep_t, rois, order, (int*)((char*)workspace + (size_t)nt * 8));
    HDY_LAUNCH_CHECK("mask_select pick");
    return HDY_OK;
}

}  // extern "C"

// SGD with momentum / Nesterov / weight decay over MANY parameter tensors in ONE launch (reference: train.py:208-233 builds
// torch.optim.SGD(nesterov=True) with three parameter groups and steps it at train.py:478; torch's multi-tensor path is ~25 launches of
// 5-25 us for yolov5s' 180 tensors).
//
//   g' = g + wd * p;   buf = first ? g' : momentum * buf + (1 - dampening) * g';   p -= lr * (nesterov ? g' + momentum * buf : buf)
//
// A device table of descriptors (pointer triple, length, group, first block) is built once; a block owns 4096 consecutive elements of one
// tensor and finds its descriptor by binary search over the first-block column.  The per-group hyper-parameters travel by value (the
// warm-up changes them every step), so the table is replayed unchanged.  Pure HBM streaming: 5 passes of 4 bytes per parameter.
#include "common.h"
#include "hdyolo.h"

namespace {

constexpr int SGD_BLOCK_ELEMS = 4096;

struct SgdHyper { float lr[HDY_SGD_MAX_GROUPS], momentum[HDY_SGD_MAX_GROUPS], dampening[HDY_SGD_MAX_GROUPS], wd[HDY_SGD_MAX_GROUPS]; };

__global__ __launch_bounds__(256) void sgd_step_kernel(const hdy_sgd_desc* __restrict__ table, int ndesc, SgdHyper h, int nesterov) {
    int lo = 0, hi = ndesc - 1;                           // last descriptor whose first_block <= blockIdx.x
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (table[mid].first_block <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const hdy_sgd_desc d = table[lo];
    const long long base = (long long)((int)blockIdx.x - d.first_block) * SGD_BLOCK_ELEMS;
    const float lr = h.lr[d.group], mom = h.momentum[d.group], damp = 1.0f - h.dampening[d.group], wd = h.wd[d.group];
    float* __restrict__ p = d.p;
    const float* __restrict__ g = d.g;
    float* __restrict__ b = d.buf;
    auto upd = [&](float pv, float gv, float bv, float& po, float& bo) {
        if (wd != 0.0f) gv = gv + wd * pv;
        if (b) {
            bo = d.first ? gv : mom * bv + damp * gv;
            gv = nesterov ? gv + mom * bo : bo;
        }
        po = pv - lr * gv;
    };
    const bool vec = (d.n & 3) == 0 && ((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)b)) & 15) == 0;
    if (vec) {
#pragma unroll
        for (int r = 0; r < SGD_BLOCK_ELEMS / 1024; ++r) {
            const long long i = base + (long long)(r * 256 + (int)threadIdx.x) * 4;
            if (i >= d.n) break;
            const f32x4 pv = *(const f32x4*)(p + i), gv = *(const f32x4*)(g + i);
            const f32x4 bv = (b && !d.first) ? *(const f32x4*)(b + i) : f32x4{0.f, 0.f, 0.f, 0.f};
            f32x4 po, bo = bv;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float pe, be = bv[e];
                upd(pv[e], gv[e], bv[e], pe, be);
                po[e] = pe;
                bo[e] = be;
            }
            *(f32x4*)(p + i) = po;
            if (b) *(f32x4*)(b + i) = bo;
        }
    } else {
        for (int r = 0; r < SGD_BLOCK_ELEMS / 256; ++r) {
            const long long i = base + r * 256 + (int)threadIdx.x;
            if (i >= d.n) break;
            float po, bo = 0.f;
            upd(p[i], g[i], (b && !d.first) ? b[i] : 0.f, po, bo);
            p[i] = po;
            if (b) b[i] = bo;
        }
    }
}

}  // namespace

extern "C" int hdy_sgd_blocks(long long n) { return (int)((n + SGD_BLOCK_ELEMS - 1) / SGD_BLOCK_ELEMS); }

extern "C" int hdy_sgd_step(const hdy_sgd_desc* table_device, int ndesc, int total_blocks, const float* lr, const float* momentum,
                            const float* dampening, const float* weight_decay, int ngroups, int nesterov, void* stream) {
    HDY_ARG(table_device && ndesc > 0 && total_blocks > 0, "sgd_step: empty table");
    HDY_ARG(ngroups > 0 && ngroups <= HDY_SGD_MAX_GROUPS && lr && momentum && dampening && weight_decay, "sgd_step: 1..%d parameter groups", HDY_SGD_MAX_GROUPS);
    SgdHyper h = {};
    for (int i = 0; i < ngroups; ++i) { h.lr[i] = lr[i]; h.momentum[i] = momentum[i]; h.dampening[i] = dampening[i]; h.wd[i] = weight_decay[i]; }
    hipLaunchKernelGGL(sgd_step_kernel, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, table_device, ndesc, h, nesterov);
    HDY_LAUNCH_CHECK("sgd_step");
    return HDY_OK;
}

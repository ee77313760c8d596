// SPPF triple max-pool (forward with argmax, backward), 2x nearest upsample (forward/backward) and the
// layout kernels at the framework boundary (NCHW fp32 images -> NHWC tiles), gfx950.  All HBM/LDS-bound; no MFMA.
//
// Reference semantics replaced: nn.MaxPool2d(5,1,2) x3 + torch.cat in SPPF.forward
// (metayolo/models/layers.py:181-189), nn.Upsample(None, 2, 'nearest') (hub yaml `fpn` rows), and the
// host-side `torch.stack(imgs).to(device)` hand-off (train.py:432).
#include <mutex>

#include "common.h"

namespace {

constexpr int CG = 8;   // channels per workgroup in the SPPF kernels

// One workgroup owns the whole H x W plane of CG channels of one image in LDS and produces the three chained
// 5x5/s1/p2 max-pools (= 5x5, 9x9, 13x13 windows) in one launch.  Ties resolve to the first maximum in
// row-major window order, as ATen's max_pool2d does.
// The plane is held with a 2-pixel border ((H+4) x (W+4); -inf forward, "no source" backward), so the taps are straight-line
// code at compile-time offsets from the element's own address: no bounds tests, no per-tap index arithmetic (the first version,
// bounds-tested loops over an unpadded plane, took 158 us forward / 294 us backward on 64 x 20 x 20 x 256).
// Each 5x5 maximum is a ROW maximum followed by a COLUMN maximum (10 taps for 25), with the position tracked through both stages:
// the first maximum in row-major order is the first row (smallest dy) whose row maximum equals the window maximum, and in that row
// the first column (smallest dx) — exactly what the two stages pick when each keeps its first maximum (and the LAST NaN, as the
// 25-tap scan `v > best || v != v` does).  The byte stored per element is  dy | dx' << 4  where dy (0..4) is the column stage's
// choice for THIS output and dx' (0..4) the row stage's choice for the row maximum at THIS position — the two tables the backward
// needs: the gradient of an output goes to row maximum (h + dy - 2, w), the gradient of a row maximum to source (h, w + dx' - 2).
// Backward is the same two stages transposed, each a 5-tap gather (deterministic, no atomics).  The 25-tap forms took 117 / 165 us.
constexpr int PB = 2;                                    // border pixels

// thread -> (pixel, 4-channel group) walk over the interior of the padded plane without divisions: 256 / (CG/4) pixels per trip.
// Four channels per lane: one 16-byte LDS read serves a tap for all four (and one dword the four stored window positions).
// CGK channels per workgroup, NT threads: CGK / 4 four-channel groups per pixel, NT / (CGK / 4) pixels per trip.
template <int CGK, int NT>
struct PlaneWalkT {
    static constexpr int QPP = CGK / 4;
    int pix, h, w;
    __device__ PlaneWalkT(int W) : pix(threadIdx.x / QPP), h(0), w(threadIdx.x / QPP) { wrap(W); }
    __device__ void wrap(int W) {
        while (w >= W) {
            w -= W;
            ++h;
        }
    }
    __device__ void next(int W) {
        pix += NT / QPP;
        w += NT / QPP;
        wrap(W);
    }
};

template <typename T> struct Quad;                        // 4 consecutive channels in global memory
template <> struct Quad<float> { typedef f32x4 type; };
template <> struct Quad<bf16_t> { typedef bf16x4 type; };
template <typename T>
__device__ __forceinline__ f32x4 load4(const T* p) {
    const typename Quad<T>::type v = *(const typename Quad<T>::type*)p;
    return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}
template <typename T>
__device__ __forceinline__ void store4(T* p, const f32x4& v) {
    typename Quad<T>::type o = {(T)v[0], (T)v[1], (T)v[2], (T)v[3]};
    *(typename Quad<T>::type*)p = o;
}

// plane elements in LDS: fp32, or bf16 for the forward pass of bf16 tensors (a maximum of bf16 values is one of them: same outputs, half the LDS,
// so that a workgroup can own 32 channels = 64 contiguous bytes per pixel instead of 8 = 16 bytes)
__device__ __forceinline__ f32x4 ldp(const float* p) { return *(const f32x4*)p; }
__device__ __forceinline__ f32x4 ldp(const bf16_t* p) {
    const bf16x4 v = *(const bf16x4*)p;
    return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}
__device__ __forceinline__ void stp(float* p, const f32x4& v) { *(f32x4*)p = v; }
__device__ __forceinline__ void stp(bf16_t* p, const f32x4& v) { *(bf16x4*)p = bf16x4{(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]}; }

// Access granularity is what the first form of these kernels (8 channels per workgroup, fp32 planes, 256 threads) paid for: 16 contiguous bytes per
// pixel in every load and store — 85 / 102 us forward / backward alone for a 13 MB tensor (64 x 20 x 20 x 256), 7-8 x their HBM time.
template <typename T, typename PT, int CGK, int NT>
__global__ __launch_bounds__(NT) void sppf_pool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y1, T* __restrict__ y2, T* __restrict__ y3,
                                                            int ld, unsigned char* __restrict__ i1, unsigned char* __restrict__ i2,
                                                            unsigned char* __restrict__ i3, int H, int W, int C) {
    extern __shared__ __attribute__((aligned(16))) unsigned char pl_raw[];
    PT* const pl = (PT*)pl_raw;                                       // [2][(H+4)*(W+4)][CGK]
    constexpr int CG = CGK, QPP = CGK / 4;
    typedef PlaneWalkT<CGK, NT> PlaneWalk;
    const int HW = H * W, WP = W + 2 * PB, PP = (H + 2 * PB) * WP;
    PT* a = pl;
    PT* b = pl + PP * CG;
    const int cgs = C / CG;
    const int n = blockIdx.x / cgs, c0 = (blockIdx.x - n * cgs) * CG + (threadIdx.x % QPP) * 4;
    const int cl = (threadIdx.x % QPP) * 4;
    const size_t base = (size_t)n * HW;
    for (int e = threadIdx.x; e < 2 * PP * CG; e += NT) pl[e] = (PT)(-INFINITY);   // both planes, borders included
    __syncthreads();
    for (PlaneWalk q(W); q.pix < HW; q.next(W)) stp(a + ((q.h + PB) * WP + q.w + PB) * CG + cl, load4<T>(x + (base + q.pix) * ld + c0));
    __syncthreads();
    T* outs[3] = {y1, y2, y3};
    unsigned char* idxs[3] = {i1, i2, i3};
    if (!i1) {
        // inference: no window positions wanted, so each 5x5 maximum is a row maximum followed by a column maximum (10 taps for 25;
        // same values, NaN included: a NaN anywhere in the window wins both stages)
        auto vmax = [](const f32x4& m, const f32x4& v) {
            f32x4 r;
#pragma unroll
            for (int i = 0; i < 4; ++i) r[i] = (v[i] > m[i] || v[i] != v[i]) ? v[i] : m[i];
            return r;
        };
#pragma unroll
        for (int pass = 0; pass < 3; ++pass) {
            for (PlaneWalk q(W); q.pix < HW; q.next(W)) {          // rows: a -> b
                const PT* ctr = a + ((q.h + PB) * WP + q.w + PB) * CG + cl;
                f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
                for (int dx = -2; dx <= 2; ++dx) m = vmax(m, ldp(ctr + dx * CG));
                stp(b + ((q.h + PB) * WP + q.w + PB) * CG + cl, m);
            }
            __syncthreads();
            for (PlaneWalk q(W); q.pix < HW; q.next(W)) {          // columns: b -> a (the next pool's source)
                const PT* ctr = b + ((q.h + PB) * WP + q.w + PB) * CG + cl;
                f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
                for (int dy = -2; dy <= 2; ++dy) m = vmax(m, ldp(ctr + dy * WP * CG));
                stp(a + ((q.h + PB) * WP + q.w + PB) * CG + cl, m);
                store4<T>(outs[pass] + (base + q.pix) * ld + c0, m);
            }
            __syncthreads();
        }
        return;
    }
    unsigned char* rx = (unsigned char*)(pl + 2 * PP * CG);          // [(H+4)*(W+4)][CG] row-stage positions
#pragma unroll
    for (int pass = 0; pass < 3; ++pass) {
        for (PlaneWalk q(W); q.pix < HW; q.next(W)) {              // rows: a -> b
            const int o = ((q.h + PB) * WP + q.w + PB) * CG + cl;
            f32x4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
            unsigned bi[4] = {0, 0, 0, 0};
#pragma unroll
            for (int dx = 0; dx < 5; ++dx) {
                const f32x4 v = ldp(a + o + (dx - 2) * CG);
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (v[i] > best[i] || v[i] != v[i]) { best[i] = v[i]; bi[i] = dx; }               // border taps are -inf: never selected
            }
            stp(b + o, best);
            *(unsigned*)(rx + o) = bi[0] | bi[1] << 8 | bi[2] << 16 | bi[3] << 24;
        }
        __syncthreads();
        for (PlaneWalk q(W); q.pix < HW; q.next(W)) {              // columns: b -> a (the next pool's source)
            const int o = ((q.h + PB) * WP + q.w + PB) * CG + cl;
            f32x4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
            unsigned bj[4] = {0, 0, 0, 0};
#pragma unroll
            for (int dy = 0; dy < 5; ++dy) {
                const f32x4 v = ldp(b + o + (dy - 2) * WP * CG);
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (v[i] > best[i] || v[i] != v[i]) { best[i] = v[i]; bj[i] = dy; }
            }
            stp(a + o, best);
            store4<T>(outs[pass] + (base + q.pix) * ld + c0, best);
            *(unsigned*)(idxs[pass] + (base + q.pix) * C + c0) = (bj[0] | bj[1] << 8 | bj[2] << 16 | bj[3] << 24) | (*(const unsigned*)(rx + o) << 4);
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------- bf16 forward on ORDER-PRESERVING KEYS (round 5)
// One workgroup of the kernel above takes ~26 us whatever the batch (profiles/r05_small_probes_ab.txt): four serial rounds of global loads, and six 5-tap
// stages whose per-channel-and-tap work is a compare, a NaN test and two selects (value, position) — ~1400 VALU instructions per thread at four
// waves per SIMD.  Here the planes hold 16-bit keys  k(v) = v < 0 ? ~bits : bits | 0x8000  (NaN -> the largest key, -0 -> +0: ATen's comparison treats
// the zeros as equal), a maximum of values is a maximum of keys, and
//   * with positions: key << 16 | (7 - tap) as ONE unsigned maximum per channel and tap — the largest value wins, among equals the first tap;
//   * without (inference): v_pk_max_u16 on the packed pairs, one instruction per two channels and tap;
// Same outputs and positions as the kernel above (the FIRST of several NaNs in a window where that one keeps the last — a NaN window is NaN either
// way).  Requesting a thread's plane loads before the border fill was tried on top (forward: +1 us; backward, with all four gradients and three
// position tables: 81 -> 106 us): sixteen waves per workgroup already overlap those loads.
__device__ __forceinline__ unsigned sppf_key2(unsigned d) {           // two packed bf16 -> two packed keys
    unsigned r = 0;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        unsigned v = (d >> (16 * h)) & 0xFFFFu;
        v = (v & 0x7FFFu) > 0x7F80u ? 0x7FC0u : v;                  // any NaN -> +qNaN
        v = v == 0x8000u ? 0u : v;                                   // -0 -> +0
        v = (v & 0x8000u) ? (~v & 0xFFFFu) : (v | 0x8000u);
        r |= v << (16 * h);
    }
    return r;
}
__device__ __forceinline__ unsigned sppf_unkey2(unsigned d) {
    unsigned r = 0;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const unsigned k = (d >> (16 * h)) & 0xFFFFu;
        r |= ((k & 0x8000u) ? (k & 0x7FFFu) : (~k & 0xFFFFu)) << (16 * h);
    }
    return r;
}
typedef unsigned short sppf_us2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned sppf_pkmax(unsigned a, unsigned b) {
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(sppf_us2, a), __builtin_bit_cast(sppf_us2, b)));
}

template <int CGK, int NT>
__global__ __launch_bounds__(NT) void sppf_pool_fwd_keys_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y1, bf16_t* __restrict__ y2,
                                                                 bf16_t* __restrict__ y3, int ld, unsigned char* __restrict__ i1,
                                                                 unsigned char* __restrict__ i2, unsigned char* __restrict__ i3, int H, int W, int C) {
    extern __shared__ __attribute__((aligned(16))) unsigned char pl_raw[];
    typedef PlaneWalkT<CGK, NT> PlaneWalk;
    constexpr int CG = CGK, QPP = CGK / 4;
    const int HW = H * W, WP = W + 2 * PB, PP = (H + 2 * PB) * WP;
    unsigned short* a = (unsigned short*)pl_raw;                     // [2][(H+4)*(W+4)][CG] keys; 0 = below every key of a value: the border
    unsigned short* b = a + PP * CG;
    unsigned char* rx = (unsigned char*)(a + 2 * PP * CG);           // row-stage positions (training form)
    const int cgs = C / CG;
    const int cl = (threadIdx.x % QPP) * 4;
    const int n = blockIdx.x / cgs, c0 = (blockIdx.x - n * cgs) * CG + cl;
    const size_t base = (size_t)n * HW;
    for (int e = threadIdx.x; e < 2 * PP * CG / 8; e += NT) ((uint4*)a)[e] = uint4{0u, 0u, 0u, 0u};
    for (int e = 2 * PP * CG / 8 * 8 + threadIdx.x; e < 2 * PP * CG; e += NT) a[e] = 0;
    __syncthreads();
    for (PlaneWalk q(W); q.pix < HW; q.next(W)) {
        const uint2 d = *(const uint2*)(x + (base + q.pix) * ld + c0);
        *(uint2*)(a + ((q.h + PB) * WP + q.w + PB) * CG + cl) = uint2{sppf_key2(d.x), sppf_key2(d.y)};
    }
    __syncthreads();
    bf16_t* outs[3] = {y1, y2, y3};
    unsigned char* idxs[3] = {i1, i2, i3};
    if (!i1) {
#pragma unroll
        for (int pass = 0; pass < 3; ++pass) {
            for (PlaneWalk q(W); q.pix < HW; q.next(W)) {          // rows: a -> b
                const int o = ((q.h + PB) * WP + q.w + PB) * CG + cl;
                uint2 m = *(const uint2*)(a + o - 2 * CG);
#pragma unroll
                for (int dx = -1; dx <= 2; ++dx) {
                    const uint2 v = *(const uint2*)(a + o + dx * CG);
                    m.x = sppf_pkmax(m.x, v.x);
                    m.y = sppf_pkmax(m.y, v.y);
                }
                *(uint2*)(b + o) = m;
            }
            __syncthreads();
            for (PlaneWalk q(W); q.pix < HW; q.next(W)) {          // columns: b -> a
                const int o = ((q.h + PB) * WP + q.w + PB) * CG + cl;
                uint2 m = *(const uint2*)(b + o - 2 * WP * CG);
#pragma unroll
                for (int dy = -1; dy <= 2; ++dy) {
                    const uint2 v = *(const uint2*)(b + o + dy * WP * CG);
                    m.x = sppf_pkmax(m.x, v.x);
                    m.y = sppf_pkmax(m.y, v.y);
                }
                *(uint2*)(a + o) = m;
                *(uint2*)(outs[pass] + (base + q.pix) * ld + c0) = uint2{sppf_unkey2(m.x), sppf_unkey2(m.y)};
            }
            __syncthreads();
        }
        return;
    }
#pragma unroll
    for (int pass = 0; pass < 3; ++pass) {
        for (PlaneWalk q(W); q.pix < HW; q.next(W)) {              // rows: a -> b, position = 7 - low bits of the winning key
            const int o = ((q.h + PB) * WP + q.w + PB) * CG + cl;
            unsigned k0 = 0, k1 = 0, k2 = 0, k3 = 0;
#pragma unroll
            for (int dx = 0; dx < 5; ++dx) {
                const uint2 v = *(const uint2*)(a + o + (dx - 2) * CG);
                const unsigned t = 7u - dx;
                k0 = max(k0, (v.x << 16) | t);
                k1 = max(k1, (v.x & 0xFFFF0000u) | t);
                k2 = max(k2, (v.y << 16) | t);
                k3 = max(k3, (v.y & 0xFFFF0000u) | t);
            }
            *(uint2*)(b + o) = uint2{(k0 >> 16) | (k1 & 0xFFFF0000u), (k2 >> 16) | (k3 & 0xFFFF0000u)};
            *(unsigned*)(rx + o) = 0x07070707u - ((k0 & 7u) | (k1 & 7u) << 8 | (k2 & 7u) << 16 | (k3 & 7u) << 24);
        }
        __syncthreads();
        for (PlaneWalk q(W); q.pix < HW; q.next(W)) {              // columns: b -> a (the next pool's source)
            const int o = ((q.h + PB) * WP + q.w + PB) * CG + cl;
            unsigned k0 = 0, k1 = 0, k2 = 0, k3 = 0;
#pragma unroll
            for (int dy = 0; dy < 5; ++dy) {
                const uint2 v = *(const uint2*)(b + o + (dy - 2) * WP * CG);
                const unsigned t = 7u - dy;
                k0 = max(k0, (v.x << 16) | t);
                k1 = max(k1, (v.x & 0xFFFF0000u) | t);
                k2 = max(k2, (v.y << 16) | t);
                k3 = max(k3, (v.y & 0xFFFF0000u) | t);
            }
            const uint2 m = uint2{(k0 >> 16) | (k1 & 0xFFFF0000u), (k2 >> 16) | (k3 & 0xFFFF0000u)};
            *(uint2*)(a + o) = m;
            *(uint2*)(outs[pass] + (base + q.pix) * ld + c0) = uint2{sppf_unkey2(m.x), sppf_unkey2(m.y)};
            const unsigned bj = 0x07070707u - ((k0 & 7u) | (k1 & 7u) << 8 | (k2 & 7u) << 16 | (k3 & 7u) << 24);
            *(unsigned*)(idxs[pass] + (base + q.pix) * C + c0) = bj | (*(const unsigned*)(rx + o) << 4);
        }
        __syncthreads();
    }
}

// dx = g0 + P1^T( g1 + P2^T( g2 + P3^T g3 ) ), P^T = scatter-to-argmax written as two 5-tap gathers (deterministic):
// column stage  r[h'][w] = SUM_dy [dy(h' - dy + 2, w) == dy] t[h' - dy + 2][w],  row stage  s[h][w'] = SUM_dx [dx'(h, w' - dx + 2) == dx] r[h][w' - dx + 2].
template <typename T, int CGK, int NT>
__global__ __launch_bounds__(NT) void sppf_pool_bwd_kernel(const T* __restrict__ g0, const T* __restrict__ g1, const T* __restrict__ g2,
                                                            const T* __restrict__ g3, int ldg, const unsigned char* __restrict__ i1,
                                                            const unsigned char* __restrict__ i2, const unsigned char* __restrict__ i3,
                                                            T* __restrict__ dx, int lddx, int H, int W, int C, int two_ix) {
    // two_ix: a second position plane, so that the next level's positions are fetched during the row stage instead of between two extra
    // barriers (planes up to 38 x 38; a 40 x 40 plane — 1280 x 1280 tiles at stride 32 — only fits with one)
    extern __shared__ __attribute__((aligned(16))) float pl[];          // [2][(H+4)*(W+4)][CGK] floats + [1 or 2][(H+4)*(W+4)][CGK] bytes
    constexpr int CG = CGK, QPP = CGK / 4;
    typedef PlaneWalkT<CGK, NT> PlaneWalk;
    const int HW = H * W, WP = W + 2 * PB, PP = (H + 2 * PB) * WP;
    float* a = pl;
    float* b = pl + PP * CG;
    unsigned char* ixb = (unsigned char*)(pl + 2 * PP * CG);
    const int cgs = C / CG;
    const int cl = (threadIdx.x % QPP) * 4;
    const int n = blockIdx.x / cgs, c0 = (blockIdx.x - n * cgs) * CG + cl;
    const size_t base = (size_t)n * HW;
    const T* gs[3] = {g2, g1, g0};
    const unsigned char* idxs[3] = {i3, i2, i1};
    for (int e = threadIdx.x; e < (two_ix ? 2 : 1) * PP * CG / 4; e += NT) ((unsigned*)ixb)[e] = 0xffffffffu;   // border: positions that never match
    for (int e = threadIdx.x; e < 2 * PP * CG; e += NT) pl[e] = 0.f;              // border sources are read (and discarded) below
    __syncthreads();
    for (PlaneWalk q(W); q.pix < HW; q.next(W)) {
        const int o = ((q.h + PB) * WP + q.w + PB) * CG + cl;
        *(f32x4*)(a + o) = load4<T>(g3 + (base + q.pix) * ldg + c0);
        *(unsigned*)(ixb + o) = *(const unsigned*)(idxs[0] + (base + q.pix) * C + c0);
    }
    __syncthreads();
#pragma unroll
    for (int pass = 0; pass < 3; ++pass) {
        const unsigned char* ix = ixb + (two_ix ? (pass & 1) * PP * CG : 0);
        unsigned char* ixn = ixb + (two_ix ? ((pass + 1) & 1) * PP * CG : 0);
        for (PlaneWalk q(W); q.pix < HW; q.next(W)) {              // columns: a -> b.  All candidate reads are unconditional and independent
            const int o = ((q.h + PB) * WP + q.w + PB) * CG + cl;
            f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int dy = 0; dy < 5; ++dy) {
                const int src = o - (dy - 2) * WP * CG;
                const unsigned pos = *(const unsigned*)(ix + src);
                const f32x4 v = *(const f32x4*)(a + src);
#pragma unroll
                for (int i = 0; i < 4; ++i) s[i] += ((pos >> (8 * i)) & 15u) == (unsigned)dy ? v[i] : 0.f;
            }
            *(f32x4*)(b + o) = s;
        }
        __syncthreads();
        for (PlaneWalk q(W); q.pix < HW; q.next(W)) {              // rows: b -> a, plus this level's own gradient
            const int o = ((q.h + PB) * WP + q.w + PB) * CG + cl;
            f32x4 s = load4<T>(gs[pass] + (base + q.pix) * ldg + c0);
#pragma unroll
            for (int ex = 0; ex < 5; ++ex) {
                const int src = o - (ex - 2) * CG;
                const unsigned pos = *(const unsigned*)(ix + src);
                const f32x4 v = *(const f32x4*)(b + src);
#pragma unroll
                for (int i = 0; i < 4; ++i) s[i] += ((pos >> (8 * i + 4)) & 15u) == (unsigned)ex ? v[i] : 0.f;
            }
            if (pass == 2) {
                store4<T>(dx + (base + q.pix) * lddx + c0, s);
            } else {
                *(f32x4*)(a + o) = s;
                if (two_ix) *(unsigned*)(ixn + o) = *(const unsigned*)(idxs[pass + 1] + (base + q.pix) * C + c0);
            }
        }
        if (pass < 2) {
            __syncthreads();
            if (!two_ix) {                                         // the one position plane is free only now
                for (PlaneWalk q(W); q.pix < HW; q.next(W))
                    *(unsigned*)(ixn + ((q.h + PB) * WP + q.w + PB) * CG + cl) = *(const unsigned*)(idxs[pass + 1] + (base + q.pix) * C + c0);
                __syncthreads();
            }
        }
    }
}

// IT: index type of the element walk — unsigned when the element count is below 2^31 (the launcher decides), long long otherwise.  The
// decomposition of the flat index costs five divisions per 16-byte vector; as 64-bit divisions they, not the memory system, set the pace
// (upsample backward 40x40 <- 80x80, 128 channels, B = 64: 60 us against an HBM bound of 21 us).
template <typename T, typename IT>
__global__ __launch_bounds__(256) void upsample2x_fwd_kernel(const T* __restrict__ x, int ldx, T* __restrict__ y, int ldy, int N, int H, int W,
                                                             int C) {
    constexpr int VE = 16 / sizeof(T);
    const IT VC = C / VE;
    const IT total = (IT)N * 4 * H * W * VC;
    for (IT idx = (IT)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (IT)gridDim.x * blockDim.x) {
        const IT op = idx / VC;
        const int c = (int)(idx - op * VC) * VE;
        const IT t = op / (IT)(2 * W);
        const int ow = (int)(op - t * (2 * W));
        const int n = (int)(t / (IT)(2 * H));
        const int oh = (int)(t - (IT)n * (2 * H));
        const size_t ip = ((size_t)n * H + (oh >> 1)) * W + (ow >> 1);
        *(i32x4*)(y + (size_t)op * ldy + c) = *(const i32x4*)(x + ip * ldx + c);
    }
}

template <typename T, typename IT>
__global__ __launch_bounds__(256) void upsample2x_bwd_kernel(const T* __restrict__ dy, int lddy, T* __restrict__ dx, int lddx, int N, int H,
                                                             int W, int C, int accumulate) {
    constexpr int VE = 16 / sizeof(T);
    const IT VC = C / VE;
    const IT total = (IT)N * H * W * VC;
    for (IT idx = (IT)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (IT)gridDim.x * blockDim.x) {
        const IT ip = idx / VC;
        const int c = (int)(idx - ip * VC) * VE;
        const IT t = ip / (IT)W;
        const int w = (int)(ip - t * W);
        const int n = (int)(t / (IT)H);
        const int h = (int)(t - (IT)n * H);
        float s[VE];
#pragma unroll
        for (int i = 0; i < VE; ++i) s[i] = 0.f;
        if (accumulate) {
            V16 u; u.i = *(const i32x4*)(dx + (size_t)ip * lddx + c);
#pragma unroll
            for (int i = 0; i < VE; ++i) s[i] = sizeof(T) == 2 ? (float)u.h[i] : u.f[i & 3];
        }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const size_t op = ((size_t)n * 2 * H + 2 * h + a) * 2 * W + 2 * w + b;
                V16 u; u.i = *(const i32x4*)(dy + op * lddy + c);
#pragma unroll
                for (int i = 0; i < VE; ++i) s[i] += sizeof(T) == 2 ? (float)u.h[i] : u.f[i & 3];
            }
        V16 o;
#pragma unroll
        for (int i = 0; i < VE; ++i) {
            if (sizeof(T) == 2) o.h[i] = (bf16_t)s[i];
            else o.f[i & 3] = s[i];
        }
        *(i32x4*)(dx + (size_t)ip * lddx + c) = o.i;
    }
}

// images [B][3][H][W] fp32 (NCHW) -> [B][H+2*pad][W+2*pad][4] of T, zero border and zero 4th channel: the
// layout the stem's 6-row-tap conv reads (24 contiguous pseudo-channels = 6 pixels x 4).
template <typename T, typename IT>
__global__ __launch_bounds__(256) void stem_prep_kernel(const float* __restrict__ img, T* __restrict__ out, int B, int H, int W, int pad) {
    const int Hp = H + 2 * pad, Wp = W + 2 * pad;
    const IT total = (IT)B * Hp * Wp;
    for (IT idx = (IT)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (IT)gridDim.x * blockDim.x) {
        const IT t = idx / (IT)Wp;
        const int wp = (int)(idx - t * Wp);
        const int b = (int)(t / (IT)Hp);
        const int hp = (int)(t - (IT)b * Hp);
        const int h = hp - pad, w = wp - pad;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (h >= 0 && h < H && w >= 0 && w < W) {
#pragma unroll
            for (int c = 0; c < 3; ++c) v[c] = img[(((size_t)b * 3 + c) * H + h) * W + w];
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) out[(size_t)idx * 4 + c] = from_f32<T>(v[c]);
    }
}

// generic NCHW fp32 -> NHWC T (pitched); tiled through LDS so both sides are coalesced
template <typename T>
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ src, T* __restrict__ dst, int ldd, int C, int HW) {
    __shared__ float tile[32][33];
    const int n = blockIdx.z, c0 = blockIdx.y * 32, p0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, p = p0 + tx;
        tile[i][tx] = (c < C && p < HW) ? src[((size_t)n * C + c) * HW + p] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int p = p0 + i, c = c0 + tx;
        if (p < HW && c < C) dst[((size_t)n * HW + p) * ldd + c] = from_f32<T>(tile[tx][i]);
    }
}

inline int sgrid(long long total) {
    long long g = (total + 255) / 256;
    if (g > 4096) g = 4096;
    if (g < 1) g = 1;
    return (int)g;
}

// configuration of an SPPF launch: the widest channel group whose planes fit the LDS (64 / 32 contiguous bytes per pixel instead of 16), more threads
// for the larger groups; CG = 8 with 256 threads is the form that fits every plane up to 40 x 40
template <typename T, typename PT, int CGK, int NT>
int sppf_fwd_launch(const void* x, void* y1, void* y2, void* y3, int ld, unsigned char* i1, unsigned char* i2, unsigned char* i3, int N, int H, int W,
                           int C, size_t smem, hipStream_t st) {
    static PerDeviceOnce once;
    once.run([] { (void)hipFuncSetAttribute((const void*)sppf_pool_fwd_kernel<T, PT, CGK, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024); });
    hipLaunchKernelGGL((sppf_pool_fwd_kernel<T, PT, CGK, NT>), dim3(N * (C / CGK)), dim3(NT), smem, st, (const T*)x, (T*)y1, (T*)y2, (T*)y3, ld, i1, i2, i3, H, W, C);
    HDY_LAUNCH_CHECK("sppf_pool_fwd");
    return HDY_OK;
}

template <int CGK, int NT>
int sppf_fwd_keys_launch(const void* x, void* y1, void* y2, void* y3, int ld, unsigned char* i1, unsigned char* i2, unsigned char* i3, int N, int H, int W, int C,
                         size_t smem, hipStream_t st) {
    static PerDeviceOnce once;
    once.run([] { (void)hipFuncSetAttribute((const void*)sppf_pool_fwd_keys_kernel<CGK, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024); });
    hipLaunchKernelGGL((sppf_pool_fwd_keys_kernel<CGK, NT>), dim3(N * (C / CGK)), dim3(NT), smem, st, (const bf16_t*)x, (bf16_t*)y1, (bf16_t*)y2, (bf16_t*)y3, ld, i1,
                       i2, i3, H, W, C);
    HDY_LAUNCH_CHECK("sppf_pool_fwd(keys)");
    return HDY_OK;
}

template <typename T, int CGK, int NT>
int sppf_bwd_launch(const void* g0, const void* g1, const void* g2, const void* g3, int ldg, const unsigned char* i1, const unsigned char* i2,
                           const unsigned char* i3, void* dx, int lddx, int N, int H, int W, int C, int two_ix, size_t smem, hipStream_t st) {
    static PerDeviceOnce once;
    once.run([] { (void)hipFuncSetAttribute((const void*)sppf_pool_bwd_kernel<T, CGK, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024); });
    hipLaunchKernelGGL((sppf_pool_bwd_kernel<T, CGK, NT>), dim3(N * (C / CGK)), dim3(NT), smem, st, (const T*)g0, (const T*)g1, (const T*)g2, (const T*)g3, ldg, i1, i2,
                       i3, (T*)dx, lddx, H, W, C, two_ix);
    HDY_LAUNCH_CHECK("sppf_pool_bwd");
    return HDY_OK;
}

}  // namespace

extern "C" {

int hdy_sppf_pool_fwd(const void* x, void* y1, void* y2, void* y3, int ld, unsigned char* idx1, unsigned char* idx2, unsigned char* idx3,
                      int N, int H, int W, int C, int dtype, void* stream) {
    HDY_ARG(x && y1 && y2 && y3 && N > 0 && H > 0 && W > 0 && C > 0, "sppf_pool_fwd: bad args");
    HDY_ARG(C % CG == 0 && ld >= C && ld % 4 == 0, "sppf_pool_fwd: C=%d must be a multiple of %d, ld >= C and a multiple of 4", C, CG);
    HDY_ARG((idx1 == nullptr) == (idx2 == nullptr) && (idx1 == nullptr) == (idx3 == nullptr), "sppf_pool_fwd: idx buffers all or none");
    const size_t pix = (size_t)(H + 2 * PB) * (W + 2 * PB);
    const size_t ix = idx1 ? 1 : 0;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == HDY_BF16) {
        // bf16 planes, 32 channels per workgroup when they fit (20 x 20: 92 KB with the position plane)
        // order-preserving keys (sppf_pool_fwd_keys_kernel): the same two shapes of workgroup; HDY_SPPF_NO_KEYS keeps the float-compare kernels (A/B)
        const bool keys = !hdy_opt(HDY_OPT_SPPF_NO_KEYS) && ld % 4 == 0 && ((uintptr_t)x & 7) == 0 && ((uintptr_t)y1 & 7) == 0 && ((uintptr_t)y2 & 7) == 0 &&
                          ((uintptr_t)y3 & 7) == 0;
        if (keys && C % 32 == 0 && pix * 32 * (2 * sizeof(bf16_t) + ix) <= 150 * 1024)
            return sppf_fwd_keys_launch<32, 1024>(x, y1, y2, y3, ld, idx1, idx2, idx3, N, H, W, C, pix * 32 * (2 * sizeof(bf16_t) + ix), st);
        if (keys && C % 16 == 0 && pix * 16 * (2 * sizeof(bf16_t) + ix) <= 150 * 1024)
            return sppf_fwd_keys_launch<16, 1024>(x, y1, y2, y3, ld, idx1, idx2, idx3, N, H, W, C, pix * 16 * (2 * sizeof(bf16_t) + ix), st);
        if (C % 32 == 0 && pix * 32 * (2 * sizeof(bf16_t) + ix) <= 150 * 1024)
            return sppf_fwd_launch<bf16_t, bf16_t, 32, 1024>(x, y1, y2, y3, ld, idx1, idx2, idx3, N, H, W, C, pix * 32 * (2 * sizeof(bf16_t) + ix), st);
        // larger planes (32 x 32 of yolov5l at 1024 x 1024, 40 x 40 of the 1280 x 1280 tiles without positions): 16 channels per workgroup, still bf16
        // planes and 32-byte pieces per pixel instead of the fp32 / 8-channel / 16-byte form below (945 us per C4 forward for a 112 us byte count)
        if (C % 16 == 0 && pix * 16 * (2 * sizeof(bf16_t) + ix) <= 150 * 1024)
            return sppf_fwd_launch<bf16_t, bf16_t, 16, 1024>(x, y1, y2, y3, ld, idx1, idx2, idx3, N, H, W, C, pix * 16 * (2 * sizeof(bf16_t) + ix), st);
        const size_t smem = pix * CG * (2 * sizeof(float) + ix);
        HDY_ARG(smem <= 150 * 1024, "sppf_pool_fwd: plane %dx%d does not fit LDS", H, W);
        return sppf_fwd_launch<bf16_t, float, CG, 256>(x, y1, y2, y3, ld, idx1, idx2, idx3, N, H, W, C, smem, st);
    }
    const size_t smem = pix * CG * (2 * sizeof(float) + ix);
    HDY_ARG(smem <= 150 * 1024, "sppf_pool_fwd: plane %dx%d does not fit LDS", H, W);
    return sppf_fwd_launch<float, float, CG, 256>(x, y1, y2, y3, ld, idx1, idx2, idx3, N, H, W, C, smem, st);
}

int hdy_sppf_pool_bwd(const void* g0, const void* g1, const void* g2, const void* g3, int ldg, const unsigned char* idx1,
                      const unsigned char* idx2, const unsigned char* idx3, void* dx, int lddx, int N, int H, int W, int C, int dtype,
                      void* stream) {
    HDY_ARG(g0 && g1 && g2 && g3 && idx1 && idx2 && idx3 && dx && N > 0 && H > 0 && W > 0, "sppf_pool_bwd: bad args");
    HDY_ARG(C % CG == 0 && ldg >= C && lddx >= C && ldg % 4 == 0 && lddx % 4 == 0, "sppf_pool_bwd: bad channel count / pitch");
    const size_t pix = (size_t)(H + 2 * PB) * (W + 2 * PB);
    hipStream_t st = (hipStream_t)stream;
    // fp32 planes (the sums of the three levels are rounded once, at the end): 16 channels per workgroup when two position planes fit beside them
    if (dtype == HDY_BF16 && C % 16 == 0 && pix * 16 * (2 * sizeof(float) + 2) <= 150 * 1024)
        return sppf_bwd_launch<bf16_t, 16, 1024>(g0, g1, g2, g3, ldg, idx1, idx2, idx3, dx, lddx, N, H, W, C, 1, pix * 16 * (2 * sizeof(float) + 2), st);
    const size_t plane = pix * CG;
    const int two_ix = plane * (2 * sizeof(float) + 2) <= 150 * 1024;
    const size_t smem = plane * (2 * sizeof(float) + (two_ix ? 2 : 1));
    HDY_ARG(smem <= 150 * 1024, "sppf_pool_bwd: plane %dx%d does not fit LDS", H, W);
    if (dtype == HDY_BF16) return sppf_bwd_launch<bf16_t, CG, 256>(g0, g1, g2, g3, ldg, idx1, idx2, idx3, dx, lddx, N, H, W, C, two_ix, smem, st);
    return sppf_bwd_launch<float, CG, 256>(g0, g1, g2, g3, ldg, idx1, idx2, idx3, dx, lddx, N, H, W, C, two_ix, smem, st);
}

int hdy_upsample2x_fwd(const void* x, int ldx, void* y, int ldy, int N, int H, int W, int C, int dtype, void* stream) {
    const int VE = dtype == HDY_BF16 ? 8 : 4;
    HDY_ARG(x && y && N > 0 && H > 0 && W > 0 && C > 0 && C % VE == 0 && ldx % VE == 0 && ldy % VE == 0 && ldx >= C && ldy >= C,
            "upsample2x_fwd: bad args");
    HDY_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0, "upsample2x_fwd: 16-byte alignment");
    const long long total = (long long)N * 4 * H * W * (C / VE);
    const int grid = sgrid(total);
    const bool small = total < (1LL << 31);
    if (dtype == HDY_BF16 && small)
        hipLaunchKernelGGL((upsample2x_fwd_kernel<bf16_t, unsigned>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, ldx, (bf16_t*)y, ldy, N, H, W, C);
    else if (dtype == HDY_BF16)
        hipLaunchKernelGGL((upsample2x_fwd_kernel<bf16_t, long long>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, ldx, (bf16_t*)y, ldy, N, H, W, C);
    else if (small)
        hipLaunchKernelGGL((upsample2x_fwd_kernel<float, unsigned>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)x, ldx, (float*)y, ldy, N, H, W, C);
    else
        hipLaunchKernelGGL((upsample2x_fwd_kernel<float, long long>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)x, ldx, (float*)y, ldy, N, H, W, C);
    HDY_LAUNCH_CHECK("upsample2x_fwd");
    return HDY_OK;
}

int hdy_upsample2x_bwd(const void* dy, int lddy, void* dx, int lddx, int N, int H, int W, int C, int accumulate, int dtype, void* stream) {
    const int VE = dtype == HDY_BF16 ? 8 : 4;
    HDY_ARG(dy && dx && N > 0 && H > 0 && W > 0 && C > 0 && C % VE == 0 && lddx % VE == 0 && lddy % VE == 0 && lddx >= C && lddy >= C,
            "upsample2x_bwd: bad args");
    HDY_ARG(((uintptr_t)dx & 15) == 0 && ((uintptr_t)dy & 15) == 0, "upsample2x_bwd: 16-byte alignment");
    const long long total = (long long)N * H * W * (C / VE);
    const int grid = sgrid(total);
    const bool small = total < (1LL << 31);
    if (dtype == HDY_BF16 && small)
        hipLaunchKernelGGL((upsample2x_bwd_kernel<bf16_t, unsigned>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy, lddy, (bf16_t*)dx, lddx, N, H, W, C, accumulate);
    else if (dtype == HDY_BF16)
        hipLaunchKernelGGL((upsample2x_bwd_kernel<bf16_t, long long>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy, lddy, (bf16_t*)dx, lddx, N, H, W, C, accumulate);
    else if (small)
        hipLaunchKernelGGL((upsample2x_bwd_kernel<float, unsigned>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)dy, lddy, (float*)dx, lddx, N, H, W, C, accumulate);
    else
        hipLaunchKernelGGL((upsample2x_bwd_kernel<float, long long>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)dy, lddy, (float*)dx, lddx, N, H, W, C, accumulate);
    HDY_LAUNCH_CHECK("upsample2x_bwd");
    return HDY_OK;
}

int hdy_stem_prep(const float* img, void* out, int B, int H, int W, int pad, int dtype, void* stream) {
    HDY_ARG(img && out && B > 0 && H > 0 && W > 0 && pad >= 0, "stem_prep: bad args");
    const long long total = (long long)B * (H + 2 * pad) * (W + 2 * pad);
    const int grid = sgrid(total);
    const bool small = total < (1LL << 31);
    if (dtype == HDY_BF16 && small)
        hipLaunchKernelGGL((stem_prep_kernel<bf16_t, unsigned>), dim3(grid), dim3(256), 0, (hipStream_t)stream, img, (bf16_t*)out, B, H, W, pad);
    else if (dtype == HDY_BF16)
        hipLaunchKernelGGL((stem_prep_kernel<bf16_t, long long>), dim3(grid), dim3(256), 0, (hipStream_t)stream, img, (bf16_t*)out, B, H, W, pad);
    else if (small)
        hipLaunchKernelGGL((stem_prep_kernel<float, unsigned>), dim3(grid), dim3(256), 0, (hipStream_t)stream, img, (float*)out, B, H, W, pad);
    else
        hipLaunchKernelGGL((stem_prep_kernel<float, long long>), dim3(grid), dim3(256), 0, (hipStream_t)stream, img, (float*)out, B, H, W, pad);
    HDY_LAUNCH_CHECK("stem_prep");
    return HDY_OK;
}

int hdy_nchw_to_nhwc(const float* src, void* dst, int ldd, int N, int C, int H, int W, int dtype, void* stream) {
    HDY_ARG(src && dst && N > 0 && C > 0 && H > 0 && W > 0 && ldd >= C, "nchw_to_nhwc: bad args");
    const int HW = H * W;
    dim3 grid(cdiv(HW, 32), cdiv(C, 32), N);
    HDY_ARG(grid.y <= 65535 && grid.z <= 65535, "nchw_to_nhwc: grid too large");
    if (dtype == HDY_BF16)
        hipLaunchKernelGGL(nchw_to_nhwc_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst, ldd, C, HW);
    else
        hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, src, (float*)dst, ldd, C, HW);
    HDY_LAUNCH_CHECK("nchw_to_nhwc");
    return HDY_OK;
}

}  // extern "C"

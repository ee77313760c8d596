// Detection head post-processing for gfx950: anchor decode and batched per-tile NMS.  No MFMA: this is
// byte/compare work bounded by HBM (decode, filter) and by the greedy dependency chain (NMS).
//
// hdy_decode      logits (B,na,ny,nx,no) -> [cx,cy,w,h px, sigmoid(obj), sigmoid(cls)..., level id]
//                 Replaces Detect.compute_proposals + the pad/cat in compute_outputs
//                 (metayolo/models/yolo_head.py:185-213, :311-312, :419-429).
// hdy_nms_batched one 1024-thread workgroup per tile:
//     1. filter (w,h >= min_wh; score > conf, strict) with an order-preserving ballot/prefix-sum compaction
//        into 64-bit keys  (~score_bits << 32 | row)            [utils_general.py:327-338]
//     2. bitonic sort of the keys in LDS (global workspace above 8192 survivors): ascending key ==
//        descending score, ties by ascending row == stable descending, the order the oracle pins
//     3. greedy suppression in score order, 1024 candidates per round: every lane tests its candidate against
//        the kept list (LDS, broadcast reads); then wave by wave a 64-step in-register resolve
//        (ballot + cross-lane broadcast of the pivot box) appends survivors, and later waves test against them.
//        Stops at max_det.  IoU arithmetic is IEEE fp32 with explicit round-to-nearest ops (no FMA
//        contraction), identical to the oracle, so kept indices are bit-exact.   [torchvision.ops.nms semantics]
//     4. gathers boxes / scores / extra of the kept rows.
//     class_aware = 1 gives non_max_suppression()'s behaviour (utils_general.py:423-523): score = obj * max cls,
//     boxes shifted by class * 7680 for the overlap test, no small-box filter, 30000-candidate pre-cut.
#include "common.h"

namespace {

struct DecodeArgs {
    const float* det;
    long long sb, sa, sy, sx;   // element strides of (b, a, y, x); o is contiguous
    float anchor_w[8], anchor_h[8];
    float stride;
    float* out;                 // [B][rows_per_image][no+1]
    int row_offset, rows_per_image;
    float level_id;
    int B, na, ny, nx, no;
};

__global__ __launch_bounds__(256) void decode_kernel(const DecodeArgs p) {
    const long long total = (long long)p.B * p.na * p.ny * p.nx;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(idx % p.nx);
        long long t = idx / p.nx;
        const int y = (int)(t % p.ny);
        t /= p.ny;
        const int a = (int)(t % p.na);
        const int b = (int)(t / p.na);
        const float* src = p.det + b * p.sb + a * p.sa + y * p.sy + x * p.sx;
        float* dst = p.out + ((size_t)b * p.rows_per_image + p.row_offset + ((size_t)a * p.ny + y) * p.nx + x) * (p.no + 1);
        for (int o = 0; o < p.no; ++o) {
            const float s = __builtin_amdgcn_rcpf(1.0f + __expf(-src[o]));      // v_exp_f32 + v_rcp_f32, the same expression as the tiled kernel (bit-identical rows)
            float v = s;
            if (o == 0) v = (s * 2.0f - 0.5f + (float)x) * p.stride;
            else if (o == 1) v = (s * 2.0f - 0.5f + (float)y) * p.stride;
            else if (o == 2) { const float q = s * 2.0f; v = q * q * p.anchor_w[a]; }
            else if (o == 3) { const float q = s * 2.0f; v = q * q * p.anchor_h[a]; }
            dst[o] = v;
        }
        dst[p.no] = p.level_id;
    }
}

// The same decode for logits stored pixel-major, [B][ny][nx][ld] fp32 with channel = a*no + o (how the plan's detection convs write
// them): a workgroup stages TP pixels (TP*ld floats, 16-byte coalesced loads) in LDS and writes, anchor by anchor, TP consecutive
// output rows = TP*(no+1) consecutive floats with one dword per lane (256 contiguous bytes per wave instruction).  The per-candidate
// kernel above reads 13 scalars and writes 14 at a 52 / 56-byte lane stride: 0.6 TB/s on 128 x 64 512 candidates; this one streams.
constexpr int DEC_TP = 128;
__global__ __launch_bounds__(256) void decode_tile_kernel(const DecodeArgs p, int ld) {
    extern __shared__ __attribute__((aligned(16))) float tile[];      // [DEC_TP][ld]
    const int hw = p.ny * p.nx;
    const long long npix = (long long)p.B * hw;
    const int row = p.no + 1;
    for (long long base = (long long)blockIdx.x * DEC_TP; base < npix; base += (long long)gridDim.x * DEC_TP) {
        const int np = (int)min((long long)DEC_TP, npix - base);
        const float4* src = (const float4*)(p.det + base * ld);
        for (int i = threadIdx.x; i < np * (ld / 4); i += 256) ((float4*)tile)[i] = src[i];
        __syncthreads();
        // thread -> (pixel = t >> 4, output column = t & 15): no division in the value loop (the row length no + 1 = 14 is not a power of two: the
        // j / row, gp / hw, rem % nx of the first version were most of its instructions), pixel coordinates once per pixel, sigmoid as
        // v_exp_f32 + v_rcp_f32 (1-2 ulp: the box / score tensors are pinned to 1e-4 relative)
        if (row <= 16) {
            const int o = threadIdx.x & 15;
            for (int px = threadIdx.x >> 4; px < np; px += 16) {
                const long long gp = base + px;
                const int b = (int)(gp / hw), rem = (int)(gp - (long long)b * hw);
                const int gy = rem / p.nx, gx = rem - gy * p.nx;
                if (o < row) {
                    float* dst = p.out + ((size_t)b * p.rows_per_image + p.row_offset + rem) * row + o;
                    for (int a = 0; a < p.na; ++a) {
                        float v = p.level_id;
                        if (o < p.no) {
                            const float s = __builtin_amdgcn_rcpf(1.0f + __expf(-tile[px * ld + a * p.no + o]));
                            v = s;
                            if (o == 0) v = (s * 2.0f - 0.5f + (float)gx) * p.stride;
                            else if (o == 1) v = (s * 2.0f - 0.5f + (float)gy) * p.stride;
                            else if (o == 2) { const float q = s * 2.0f; v = q * q * p.anchor_w[a]; }
                            else if (o == 3) { const float q = s * 2.0f; v = q * q * p.anchor_h[a]; }
                        }
                        dst[(size_t)a * hw * row] = v;
                    }
                }
            }
        } else
        for (int a = 0; a < p.na; ++a) {
            for (int j = threadIdx.x; j < np * row; j += 256) {
                const int px = j / row, o = j - px * row;
                const long long gp = base + px;
                const int b = (int)(gp / hw), rem = (int)(gp - (long long)b * hw);
                float v = p.level_id;
                if (o < p.no) {
                    const float s = __builtin_amdgcn_rcpf(1.0f + __expf(-tile[px * ld + a * p.no + o]));
                    v = s;
                    if (o == 0) v = (s * 2.0f - 0.5f + (float)(rem % p.nx)) * p.stride;
                    else if (o == 1) v = (s * 2.0f - 0.5f + (float)(rem / p.nx)) * p.stride;
                    else if (o == 2) { const float q = s * 2.0f; v = q * q * p.anchor_w[a]; }
                    else if (o == 3) { const float q = s * 2.0f; v = q * q * p.anchor_h[a]; }
                }
                p.out[((size_t)b * p.rows_per_image + p.row_offset + (size_t)a * hw + rem) * row + o] = v;
            }
        }
        __syncthreads();
    }
}

// gradient of the logits as autograd hands it over, (b,a,y,x,o) fp32 with arbitrary element strides ->
// NHWC [B][ny][nx][ldo] of T with channel = a*no + o and zero padding up to ldo (what dgrad/wgrad consume)
template <typename T>
__global__ __launch_bounds__(256) void det_grad_pack_kernel(const float* __restrict__ g, long long sb, long long sa, long long sy, long long sx,
                                                            long long so, T* __restrict__ out, int ldo, int B, int na, int ny, int nx, int no) {
    const long long total = (long long)B * ny * nx * ldo;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int ch = (int)(idx % ldo);
        long long t = idx / ldo;
        const int x = (int)(t % nx);
        t /= nx;
        const int y = (int)(t % ny);
        const int b = (int)(t / ny);
        float v = 0.f;
        if (ch < na * no) {
            const int a = ch / no, o = ch - a * no;
            v = g[b * sb + a * sa + y * sy + x * sx + o * so];
        }
        out[idx] = from_f32<T>(v);
    }
}

// ------------------------------------------------------------------------------------------- NMS
constexpr int NT = 1024;          // threads per tile
constexpr int LDS_KEYS = 8192;    // sort capacity in LDS
constexpr int MAX_KEEP = 4096;    // kept-list capacity in LDS

struct NmsArgs {
    const float* preds;           // [B][N][row]
    int B, N, row, nc;
    float conf, iou;
    int max_det;
    float min_wh;
    int class_aware;
    int xyxy;                     // rows are (x1, y1, x2, y2, score): torchvision.ops.nms on explicit boxes
    unsigned long long* ws;       // [B][P] sort workspace (global)
    int P;
    long long* keep;              // [B][max_det]
    int* n_keep;                  // [B]
    float* out_boxes;             // [B][max_det][4]
    float* out_scores;            // [B][max_det][1+nc]
    float* out_extra;             // [B][max_det][row-5-nc] or null
    float* out_conf;              // [B][max_det] ranking score, or null
    int* out_cls;                 // [B][max_det] best class (class_aware) or null
    float4* kept_g;               // max_det > MAX_KEEP: the kept list lives here, [B][max_det] boxes then [B][max_det] areas (behind the sort workspace)
    float* kept_area_g;
};

__device__ __forceinline__ bool iou_gt(float ax1, float ay1, float ax2, float ay2, float aarea, float bx1, float by1, float bx2, float by2,
                                       float barea, float thr) {
    const float xx1 = ax1 > bx1 ? ax1 : bx1;
    const float yy1 = ay1 > by1 ? ay1 : by1;
    const float xx2 = ax2 < bx2 ? ax2 : bx2;
    const float yy2 = ay2 < by2 ? ay2 : by2;
    float w = __fsub_rn(xx2, xx1);
    float h = __fsub_rn(yy2, yy1);
    w = w < 0.f ? 0.f : w;
    h = h < 0.f ? 0.f : h;
    const float inter = __fmul_rn(w, h);
    const float ovr = __fdiv_rn(inter, __fsub_rn(__fadd_rn(aarea, barea), inter));
    return ovr > thr;
}

struct Cand { float x1, y1, x2, y2, area, score; int cls; };

// 32-bit key whose ASCENDING unsigned order is DESCENDING score order for every float (negative scores included: explicit-box calls
// carry caller scores of any sign; -0.0 ranks as +0.0).  For the positive scores of the thresholded paths this is ~bits, as before.
__device__ __forceinline__ unsigned desc_key(float s) {
    unsigned u = s == 0.0f ? 0u : __float_as_uint(s);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);      // ascending order-preserving map of the float line
    return ~u;
}

__device__ __forceinline__ Cand make_cand(const float* p, int nc, int class_aware, int xyxy = 0) {
    Cand c;
    c.cls = 0;
    if (xyxy) {
        c.x1 = p[0]; c.y1 = p[1]; c.x2 = p[2]; c.y2 = p[3];
        c.score = p[4];
        return c;
    }
    const float hw = p[2] / 2.0f, hh = p[3] / 2.0f;
    c.x1 = __fsub_rn(p[0], hw);
    c.y1 = __fsub_rn(p[1], hh);
    c.x2 = __fadd_rn(p[0], hw);
    c.y2 = __fadd_rn(p[1], hh);
    c.score = p[4];
    if (class_aware) {
        float best = __fmul_rn(p[5], p[4]);
        for (int k = 1; k < nc; ++k) {
            const float v = __fmul_rn(p[5 + k], p[4]);
            if (v > best) { best = v; c.cls = k; }
        }
        c.score = best;
    }
    return c;
}

// GK: the kept list in global memory (max_det beyond the 4096 entries LDS holds: nms_per_image slices [:max_det] for ANY value,
// utils_general.py:342).  Same greedy pass, same order; the list is written by one wave and read by the others after a workgroup barrier, which
// orders global accesses of one workgroup as it orders LDS ones.  The LDS instance keeps its ds_read addressing (a generic pointer would turn them
// into flat loads).
template <bool GK>
__global__ __launch_bounds__(NT) void nms_kernel(const NmsArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned long long* lkeys = (unsigned long long*)smem;                 // [LDS_KEYS]
    float4* kept_l = (float4*)(smem + (size_t)LDS_KEYS * 8);               // [MAX_KEEP]
    float* kept_area_l = (float*)(kept_l + MAX_KEEP);                      // [MAX_KEEP]
    float4* const kept_gb = GK ? p.kept_g + (size_t)blockIdx.x * p.max_det : nullptr;
    float* const kept_area_gb = GK ? p.kept_area_g + (size_t)blockIdx.x * p.max_det : nullptr;
#define kept (GK ? kept_gb : kept_l)
#define kept_area (GK ? kept_area_gb : kept_area_l)
    __shared__ int wave_cnt[NT / 64];
    __shared__ int base_sh;
    __shared__ int nk_hist[NT / 64 + 1];

    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* rows = p.preds + (size_t)b * p.N * p.row;
    unsigned long long* gkeys = p.ws + (size_t)b * p.P;

    // ---- 1. filter + ordered compaction (keys to the global workspace; moved to LDS below when they fit)
    if (tid == 0) base_sh = 0;
    __syncthreads();
    for (int t0 = 0; t0 < p.N; t0 += NT) {
        const int i = t0 + tid;
        bool ok = false;
        unsigned long long key = 0;
        if (i < p.N) {
            const float* r = rows + (size_t)i * p.row;
            const Cand c = make_cand(r, p.nc, p.class_aware, p.xyxy);
            if (p.xyxy) ok = true;
            else if (p.class_aware) ok = (r[4] > p.conf) && (c.score > p.conf);
            else ok = (__fsub_rn(c.x2, c.x1) >= p.min_wh) && (__fsub_rn(c.y2, c.y1) >= p.min_wh) && (c.score > p.conf);
            key = ((unsigned long long)desc_key(c.score) << 32) | (unsigned)i;
        }
        const unsigned long long m = __ballot(ok);
        const int pos = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wave_cnt[wave] = __popcll(m);
        __syncthreads();
        int off = base_sh;
        for (int w = 0; w < wave; ++w) off += wave_cnt[w];
        if (ok) gkeys[off + pos] = key;
        __syncthreads();
        if (tid == 0) {
            int tot = 0;
            for (int w = 0; w < NT / 64; ++w) tot += wave_cnt[w];
            base_sh += tot;
        }
        __syncthreads();
    }
    int M = base_sh;
    int P2 = 1;
    while (P2 < M) P2 <<= 1;
    // ---- 2. bitonic sort, ascending
    unsigned long long* keys = gkeys;
    if (P2 <= LDS_KEYS) {
        keys = lkeys;
        for (int i = tid; i < P2; i += NT) lkeys[i] = i < M ? gkeys[i] : ~0ull;
    } else {
        for (int i = M + tid; i < P2; i += NT) gkeys[i] = ~0ull;
    }
    __syncthreads();
    for (int k = 2; k <= P2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < P2; i += NT) {
                const int l = i ^ j;
                if (l > i) {
                    const unsigned long long a = keys[i], c = keys[l];
                    const bool up = (i & k) == 0;
                    if ((a > c) == up) { keys[i] = c; keys[l] = a; }
                }
            }
            __syncthreads();
        }
    }
    if (p.class_aware && M > 30000) M = 30000;

    // ---- 3. greedy suppression
    long long* keep = p.keep + (size_t)b * p.max_det;
    if (tid == 0) nk_hist[0] = 0;
    __syncthreads();
    int nk = 0;
    bool done = false;
    for (int s0 = 0; s0 < M && !done; s0 += NT) {
        const int s = s0 + tid;
        bool alive = s < M;
        Cand c = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0};
        int src = 0;
        if (alive) {
            src = (int)(keys[s] & 0xFFFFFFFFull);
            c = make_cand(rows + (size_t)src * p.row, p.nc, p.class_aware, p.xyxy);
            if (p.class_aware) {
                const float off = (float)c.cls * 7680.0f;
                c.x1 = __fadd_rn(c.x1, off); c.y1 = __fadd_rn(c.y1, off); c.x2 = __fadd_rn(c.x2, off); c.y2 = __fadd_rn(c.y2, off);
            }
            c.area = __fmul_rn(__fsub_rn(c.x2, c.x1), __fsub_rn(c.y2, c.y1));
        }
        // 3a. against everything kept in earlier rounds
        for (int k = 0; k < nk; ++k) {
            const float4 kb = kept[k];
            if (alive && iou_gt(kb.x, kb.y, kb.z, kb.w, kept_area[k], c.x1, c.y1, c.x2, c.y2, c.area, p.iou)) alive = false;
        }
        // 3b. inside the round, wave by wave
        const int nw = min(NT / 64, (M - s0 + 63) / 64);
        int nk_old = nk;
        for (int w = 0; w < nw; ++w) {
            if (wave == w) {
                for (int i = 0; i < 64; ++i) {
                    const unsigned long long m = __ballot(alive);
                    if (!((m >> i) & 1ull)) continue;
                    const float px1 = __shfl(c.x1, i), py1 = __shfl(c.y1, i), px2 = __shfl(c.x2, i), py2 = __shfl(c.y2, i);
                    const float pa = __shfl(c.area, i);
                    if (lane > i && alive && iou_gt(px1, py1, px2, py2, pa, c.x1, c.y1, c.x2, c.y2, c.area, p.iou)) alive = false;
                }
                const unsigned long long m = __ballot(alive);
                const int pos = nk_old + __popcll(m & ((1ull << lane) - 1ull));
                if (alive && pos < p.max_det) {
                    kept[pos] = make_float4(c.x1, c.y1, c.x2, c.y2);
                    kept_area[pos] = c.area;
                    keep[pos] = (long long)src;
                }
                if (lane == 0) nk_hist[w + 1] = nk_old + __popcll(m);
            }
            __syncthreads();
            const int nk_new = min(nk_hist[w + 1], p.max_det);
            if (wave > w) {
                for (int k = nk_old; k < nk_new; ++k) {
                    const float4 kb = kept[k];
                    if (alive && iou_gt(kb.x, kb.y, kb.z, kb.w, kept_area[k], c.x1, c.y1, c.x2, c.y2, c.area, p.iou)) alive = false;
                }
            }
            nk_old = nk_new;
            if (nk_new >= p.max_det) { done = true; break; }
        }
        nk = nk_old;
        __syncthreads();          // nk_hist is rewritten by the next round
    }
    if (tid == 0) p.n_keep[b] = nk;
    __syncthreads();

    // ---- 4. gather outputs of the kept rows
    const int nsc = 1 + p.nc, nex = p.row - 5 - p.nc;
    for (int k = tid; k < p.max_det; k += NT) {
        if (k >= nk) { keep[k] = -1; continue; }
        if (!p.out_boxes) continue;
        const float* r = rows + (size_t)keep[k] * p.row;
        const Cand c = make_cand(r, p.nc, p.class_aware);
        float* ob = p.out_boxes + ((size_t)b * p.max_det + k) * 4;
        ob[0] = c.x1; ob[1] = c.y1; ob[2] = c.x2; ob[3] = c.y2;
        float* os = p.out_scores + ((size_t)b * p.max_det + k) * nsc;
        for (int j = 0; j < nsc; ++j) os[j] = r[4 + j];
        if (p.out_extra)
            for (int j = 0; j < nex; ++j) p.out_extra[((size_t)b * p.max_det + k) * nex + j] = r[5 + p.nc + j];
        if (p.out_conf) p.out_conf[(size_t)b * p.max_det + k] = c.score;
        if (p.out_cls) p.out_cls[(size_t)b * p.max_det + k] = c.cls;
    }
}

#undef kept
#undef kept_area

inline int next_pow2(int v) {
    int p = 1;
    while (p < v) p <<= 1;
    return p;
}

}  // namespace

// Tail of Detect.compute_outputs (metayolo/models/yolo_head.py:335-345) for the whole batch in one launch: hierarchical scores (every
// (child, parent) pair of the class tree in the reference's order: child *= parent, in place on the padded NMS rows), then per kept box
// score = best class score if it beats conf, else objectness; label = best class + 1, else -100 — written COMPACTED (image b's rows start
// at the sum of the earlier images' counts, which every workgroup adds up for itself) so that the host splits three tensors instead of
// slicing 3 x B padded ones.  multi_label: all 1 + nc scores per box and (score > conf) flags.
__global__ __launch_bounds__(256) void det_outputs_kernel(float* __restrict__ scores, const float* __restrict__ boxes, const int* __restrict__ n_keep,
                                                          int B, int max_det, int nc, const int* __restrict__ pairs, int npairs, float conf,
                                                          int multi_label, float* __restrict__ out_boxes, float* __restrict__ out_scores,
                                                          void* __restrict__ out_labels, int* __restrict__ offsets) {
    __shared__ int red[256];
    const int b = blockIdx.x, tid = threadIdx.x;
    int part = 0;
    for (int i = tid; i < b; i += 256) part += n_keep[i];
    red[tid] = part;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) red[tid] += red[tid + s];
        __syncthreads();
    }
    const int off = red[0], n = n_keep[b], C = 1 + nc;
    if (offsets && tid == 0) {
        offsets[b] = off;
        if (b == B - 1) offsets[B] = off + n;
    }
    for (int j = tid; j < n; j += 256) {
        float* row = scores + ((size_t)b * max_det + j) * C;
        for (int q = 0; q < npairs; ++q) row[pairs[2 * q]] *= row[pairs[2 * q + 1]];
        const float* bx = boxes + ((size_t)b * max_det + j) * 4;
        float* ob = out_boxes + (size_t)(off + j) * 4;
        ob[0] = bx[0]; ob[1] = bx[1]; ob[2] = bx[2]; ob[3] = bx[3];
        if (multi_label) {
            for (int c = 0; c < C; ++c) {
                out_scores[(size_t)(off + j) * C + c] = row[c];
                ((unsigned char*)out_labels)[(size_t)(off + j) * C + c] = row[c] > conf ? 1 : 0;
            }
        } else {
            float best = row[1];
            int arg = 0;
            for (int c = 1; c < nc; ++c)
                if (row[1 + c] > best) { best = row[1 + c]; arg = c; }       // first maximum, as torch.max
            const bool hit = best > conf;
            out_scores[off + j] = hit ? best : row[0];
            ((long long*)out_labels)[off + j] = hit ? arg + 1 : -100;
        }
    }
}

extern "C" {

int hdy_decode(const float* det, long long sb, long long sa, long long sy, long long sx, const float* anchor_px, float stride, float* out,
               int row_offset, int rows_per_image, int level_id, int B, int na, int ny, int nx, int no, void* stream) {
    HDY_ARG(det && anchor_px && out, "decode: null pointer");
    HDY_ARG(B > 0 && na > 0 && na <= 8 && ny > 0 && nx > 0 && no >= 5, "decode: bad shape B=%d na=%d ny=%d nx=%d no=%d", B, na, ny, nx, no);
    HDY_ARG(row_offset >= 0 && row_offset + na * ny * nx <= rows_per_image, "decode: level does not fit rows_per_image");
    DecodeArgs a;
    a.det = det; a.sb = sb; a.sa = sa; a.sy = sy; a.sx = sx;
    for (int i = 0; i < na; ++i) { a.anchor_w[i] = anchor_px[2 * i]; a.anchor_h[i] = anchor_px[2 * i + 1]; }
    a.stride = stride; a.out = out; a.row_offset = row_offset; a.rows_per_image = rows_per_image; a.level_id = (float)level_id;
    a.B = B; a.na = na; a.ny = ny; a.nx = nx; a.no = no;
    // pixel-major logits (the plan's layout): the tiled kernel
    const bool tiled = sa == no && sy == (long long)nx * sx && sb == (long long)ny * sy && sx % 4 == 0 && sx >= (long long)na * no && sx <= 64 &&
                       (((uintptr_t)det) & 15) == 0;
    if (tiled) {
        long long g = ((long long)B * ny * nx + DEC_TP - 1) / DEC_TP;
        if (g > 2048) g = 2048;
        hipLaunchKernelGGL(decode_tile_kernel, dim3((int)g), dim3(256), DEC_TP * (size_t)sx * sizeof(float), (hipStream_t)stream, a, (int)sx);
        HDY_LAUNCH_CHECK("decode(tiled)");
        return HDY_OK;
    }
    long long g = ((long long)B * na * ny * nx + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(decode_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream, a);
    HDY_LAUNCH_CHECK("decode");
    return HDY_OK;
}

int hdy_det_grad_pack(const float* g, long long sb, long long sa, long long sy, long long sx, long long so, void* out, int ldo, int B, int na,
                      int ny, int nx, int no, int dtype, void* stream) {
    HDY_ARG(g && out && B > 0 && na > 0 && ny > 0 && nx > 0 && no > 0 && ldo >= na * no, "det_grad_pack: bad args");
    long long n = ((long long)B * ny * nx * ldo + 255) / 256;
    if (n > 8192) n = 8192;
    if (dtype == HDY_BF16)
        hipLaunchKernelGGL(det_grad_pack_kernel<bf16_t>, dim3((int)n), dim3(256), 0, (hipStream_t)stream, g, sb, sa, sy, sx, so, (bf16_t*)out, ldo,
                           B, na, ny, nx, no);
    else
        hipLaunchKernelGGL(det_grad_pack_kernel<float>, dim3((int)n), dim3(256), 0, (hipStream_t)stream, g, sb, sa, sy, sx, so, (float*)out, ldo, B,
                           na, ny, nx, no);
    HDY_LAUNCH_CHECK("det_grad_pack");
    return HDY_OK;
}

size_t hdy_nms_workspace_bytes(int B, int N) { return (size_t)(B > 0 ? B : 0) * next_pow2(N > 1 ? N : 1) * sizeof(unsigned long long); }

// workspace of a call with this max_det: the sort keys, and beyond 4096 kept boxes per tile the kept list itself (20 bytes per entry)
size_t hdy_nms_workspace_bytes_for(int B, int N, int max_det) {
    size_t n = (hdy_nms_workspace_bytes(B, N) + 15) & ~(size_t)15;
    if (max_det > MAX_KEEP) n += (size_t)(B > 0 ? B : 0) * (size_t)max_det * 20;
    return n;
}

static int nms_launch(NmsArgs& a, void* workspace, hipStream_t st, const char* who) {
    const size_t sort_bytes = (hdy_nms_workspace_bytes(a.B, a.N) + 15) & ~(size_t)15;
    if (a.max_det > MAX_KEEP) {
        a.kept_g = (float4*)((char*)workspace + sort_bytes);                        
        a.kept_area_g = (float*)(a.kept_g + (size_t)a.B * a.max_det);
        HDY_ARG(((uintptr_t)a.kept_g & 15) == 0, "%s: workspace must be 16-byte aligned", who);
        const size_t smem = (size_t)LDS_KEYS * 8;
        (void)hipFuncSetAttribute((const void*)nms_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipLaunchKernelGGL(nms_kernel<true>, dim3(a.B), dim3(NT), smem, st, a);
    } else {
        a.kept_g = nullptr; a.kept_area_g = nullptr;
        const size_t smem = (size_t)LDS_KEYS * 8 + (size_t)MAX_KEEP * 16 + (size_t)MAX_KEEP * 4;
        (void)hipFuncSetAttribute((const void*)nms_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipLaunchKernelGGL(nms_kernel<false>, dim3(a.B), dim3(NT), smem, st, a);
    }
    HDY_LAUNCH_CHECK(who);
    return HDY_OK;
}

int hdy_nms_batched(const float* preds, int B, int N, int row, int nc, float conf, float iou, int max_det, float min_wh, int class_aware,
                    long long* keep, int* n_keep, float* out_boxes, float* out_scores, float* out_extra, float* out_conf, int* out_cls,
                    void* workspace, size_t ws_bytes, void* stream) {
    HDY_ARG(B >= 0 && N >= 0, "nms: negative size");
    if (B == 0) return HDY_OK;
    HDY_ARG(keep && n_keep && out_boxes && out_scores, "nms: null output pointer");
    HDY_ARG(N == 0 || preds, "nms: null preds");
    HDY_ARG(nc >= 1 && row >= 5 + nc, "nms: row=%d too short for nc=%d", row, nc);
    HDY_ARG(conf >= 0.f && conf <= 1.f && iou >= 0.f && iou <= 1.f, "nms: thresholds must be in [0,1]");
    HDY_ARG(max_det >= 1, "nms: max_det=%d must be positive", max_det);
    HDY_ARG(workspace && ws_bytes >= hdy_nms_workspace_bytes_for(B, N, max_det), "nms: workspace too small");
    HDY_ARG(row == 5 + nc || out_extra, "nms: out_extra required when rows carry extra columns");
    NmsArgs a;
    a.preds = preds; a.B = B; a.N = N; a.row = row; a.nc = nc; a.conf = conf; a.iou = iou; a.max_det = max_det; a.min_wh = min_wh;
    a.class_aware = class_aware; a.xyxy = 0; a.ws = (unsigned long long*)workspace; a.P = next_pow2(N > 1 ? N : 1);
    a.keep = keep; a.n_keep = n_keep; a.out_boxes = out_boxes; a.out_scores = out_scores; a.out_extra = out_extra; a.out_conf = out_conf;
    a.out_cls = out_cls;
    return nms_launch(a, workspace, (hipStream_t)stream, "nms");
}

int hdy_det_outputs(float* scores, const float* boxes, const int* n_keep, int B, int max_det, int nc, const int* pairs, int npairs, float conf,
                    int multi_label, float* out_boxes, float* out_scores, void* out_labels, int* offsets, void* stream) {
    HDY_ARG(B >= 0 && max_det >= 1 && nc >= 1 && npairs >= 0, "det_outputs: bad sizes");
    if (B == 0) return HDY_OK;
    HDY_ARG(scores && boxes && n_keep && out_boxes && out_scores && out_labels && (npairs == 0 || pairs), "det_outputs: null pointer");
    hipLaunchKernelGGL(det_outputs_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, scores, boxes, n_keep, B, max_det, nc, pairs, npairs, conf, multi_label,
                       out_boxes, out_scores, out_labels, offsets);
    HDY_LAUNCH_CHECK("det_outputs");
    return HDY_OK;
}

int hdy_nms_boxes(const float* boxes_scores, int B, int N, float iou, int max_det, long long* keep, int* n_keep, void* workspace,
                  size_t ws_bytes, void* stream) {
    HDY_ARG(B >= 0 && N >= 0, "nms_boxes: negative size");
    if (B == 0) return HDY_OK;
    HDY_ARG(keep && n_keep, "nms_boxes: null output pointer");
    HDY_ARG(N == 0 || boxes_scores, "nms_boxes: null input");
    HDY_ARG(iou >= 0.f && iou <= 1.f, "nms_boxes: iou threshold must be in [0,1]");
    HDY_ARG(max_det >= 1, "nms_boxes: max_det=%d must be positive", max_det);
    HDY_ARG(workspace && ws_bytes >= hdy_nms_workspace_bytes_for(B, N, max_det), "nms_boxes: workspace too small");
    NmsArgs a;
    a.preds = boxes_scores; a.B = B; a.N = N; a.row = 5; a.nc = 0; a.conf = 0.f; a.iou = iou; a.max_det = max_det; a.min_wh = 0.f;
    a.class_aware = 0; a.xyxy = 1; a.ws = (unsigned long long*)workspace; a.P = next_pow2(N > 1 ? N : 1);
    a.keep = keep; a.n_keep = n_keep; a.out_boxes = nullptr; a.out_scores = nullptr; a.out_extra = nullptr; a.out_conf = nullptr;
    a.out_cls = nullptr;
    return nms_launch(a, workspace, (hipStream_t)stream, "nms_boxes");
}

}  // extern "C"

// C-ABI entry points for the convolution family (see include/hdyolo.h).  Everything here is host code that
// validates shapes, derives the tap-window geometry and launches the kernels in conv_igemm.hip / conv_wgrad.hip
// on the caller's stream.  No allocation, no synchronisation; process state = the option table below (atomics, initialised once from the
// environment), the per-kernel "LDS size attribute set" once-flags, and the thread-local error text / dispatch log.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>

#include "common.h"
#include "hdyolo_internal.h"
#include "hdyolo.h"

int hdy_wgrad_reduce_launch(const float* partial, int splits, size_t slab_stride, int K, int Q, int mode, int C, int R, int S, float* grad,
                            int accumulate, hipStream_t st);
int hdy_pack_weight_launch(const hdy_pack_desc& d, hipStream_t st);
int hdy_pack_batch_launch(const hdy_pack_desc* table, int n, int total_blocks, hipStream_t st);

static thread_local char g_err[512] = "";

void hdy_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

namespace {
struct OptDef { const char* name; int def; bool flag; };       // flag: the variable's presence means 1 (its value is not parsed)
const OptDef g_opt_def[HDY_OPT_COUNT] = {
    {"HDY_NO_CLASS_WALK", 0, true}, {"HDY_NO_CONV3X3", 0, true}, {"HDY_C3_GRID", 0, false}, {"HDY_NO_CONV3X3S2", 0, false},
    {"HDY_NO_DGRAD_S2", 0, false}, {"HDY_TILE_INTERLEAVE", 1, false}, {"HDY_NO_BIG_TILES", 0, true}, {"HDY_NO_STEM_KERNEL", 0, true},
    {"HDY_WGRAD_BLOCKS", 512, false}, {"HDY_NO_STEM_WGRAD", 0, true}, {"HDY_NO_WGRAD3X3", 0, true}, 
    {"HDY_LOSS_GRID", 2048, false}, {"HDY_NO_DEEP", 0, true}, {"HDY_NO_WGRAD_S2", 0, true}, {"HDY_NO_WGRAD_DEEP", 0, true}, {"HDY_DEEP_BN", 0, false}, {"HDY_DEEP_DEBUG", 0, false}, {"HDY_DEEP_ALL", 1, false}, {"HDY_DEEP_MIN_TILES", 160, false}, {"HDY_DEEP_WALK", 2, false}, {"HDY_NO_BN_REDUCE4", 0, true},
    {"HDY_WGRAD_TILE", 0, false}, {"HDY_SPPF_NO_KEYS", 0, true}, {"HDY_WGRAD_DEEP_KMIN", 192, false}, {"HDY_NO_CONV3X3_C128", 0, true}, {"HDY_NO_F1X1_96", 0, true},
};
std::atomic<int> g_opt[HDY_OPT_COUNT];
std::once_flag g_opt_once;
void opt_init() {
    std::call_once(g_opt_once, [] {
        for (int i = 0; i < HDY_OPT_COUNT; ++i) {
            const char* v = getenv(g_opt_def[i].name);
            g_opt[i].store(v ? (g_opt_def[i].flag ? 1 : atoi(v)) : g_opt_def[i].def, std::memory_order_relaxed);
        }
    });
}
int opt_index(const char* name) {
    for (int i = 0; name && i < HDY_OPT_COUNT; ++i)
        if (!strcmp(name, g_opt_def[i].name)) return i;
    return -1;
}
thread_local char g_disp_last[64] = "";
// the log is process-wide (autograd runs the backward launch list on its own thread) and ordered; readers get a thread-local copy
std::mutex g_disp_mu;
char g_disp_log[32768] = "";
size_t g_disp_len = 0;
thread_local char g_disp_copy[32768] = "";
}  // namespace

int hdy_opt(int id) {
    opt_init();
    return g_opt[id].load(std::memory_order_relaxed);
}

void hdy_note_dispatch(const char* what) {
    snprintf(g_disp_last, sizeof(g_disp_last), "%s", what);
    const size_t n = strlen(g_disp_last);
    std::lock_guard<std::mutex> lock(g_disp_mu);
    if (g_disp_len + n + 2 < sizeof(g_disp_log)) {
        memcpy(g_disp_log + g_disp_len, g_disp_last, n);
        g_disp_log[g_disp_len + n] = ';';
        g_disp_log[g_disp_len + n + 1] = 0;
        g_disp_len += n + 1;
    }
}

namespace {

enum { KIND_FWD = 0, KIND_DGRAD = 1, KIND_STEM = 2 };

inline int velems(int dtype) { return dtype == HDY_BF16 ? 8 : 4; }
inline int bke(int dtype) { return 8 * velems(dtype); }
inline size_t esize(int dtype) { return dtype == HDY_BF16 ? 2 : 4; }

// One spatial axis of a stride-2 dgrad parity class: output positions h = 2*i + a take the kernel taps
// r = rmax, rmax-2, ... (same parity as a + pad), reading dy row i + d0 + t for the t-th of them.
struct Axis { int taps, d0, rmax; };
inline Axis class_axis(int R, int pad, int a) {
    Axis ax = {0, 0, -1};
    for (int r = R - 1; r >= 0; --r)
        if (((a + pad - r) & 1) == 0) {
            if (ax.rmax < 0) { ax.rmax = r; ax.d0 = (a + pad - r) / 2; }
            ++ax.taps;
        }
    return ax;
}

// rows (padded to the N-tile) x pitch of one packed block
inline size_t block_elems(int rows, int kd, int dtype) {
    const int bn = hdy_conv_bn_tile(rows);
    return (size_t)round_up(rows, bn) * round_up(kd, bke(dtype));
}

}  // namespace

extern "C" {

const char* hdy_last_error(void) { return g_err; }

// name of the kernel family the last launcher call on this thread picked ("" before the first)
const char* hdy_last_dispatch(void) { return g_disp_last; }

// every pick of every thread since the last hdy_dispatch_log_reset(), in launch order, ';'-separated (first 32 KB)
const char* hdy_dispatch_log(void) {
    std::lock_guard<std::mutex> lock(g_disp_mu);
    memcpy(g_disp_copy, g_disp_log, g_disp_len + 1);
    return g_disp_copy;
}

void hdy_dispatch_log_reset(void) {
    std::lock_guard<std::mutex> lock(g_disp_mu);
    g_disp_log[0] = 0;
    g_disp_len = 0;
    g_disp_last[0] = 0;
}

// process-wide switch by the name of its environment variable (common.h HdyOption); returns the previous value, HDY_EINVAL for an unknown name
int hdy_set_option(const char* name, int value) {
    const int i = opt_index(name);
    HDY_ARG(i >= 0, "set_option: unknown option %s", name ? name : "(null)");
    opt_init();
    return g_opt[i].exchange(value, std::memory_order_relaxed);
}

int hdy_get_option(const char* name) {
    const int i = opt_index(name);
    HDY_ARG(i >= 0, "get_option: unknown option %s", name ? name : "(null)");
    return hdy_opt(i);
}

int hdy_version(void) { return HDY_ABI_VERSION; }

// the reciprocal the conv loader divides by (host-only; exported so that the identity can be tested without a GPU)
int hdy_fastdiv_magic(unsigned d, unsigned* magic, int* shift) {
    HDY_ARG(d > 0 && magic && shift, "fastdiv_magic: d must be positive");
    hdy_magic(d, magic, shift);
    return HDY_OK;
}

int hdy_conv_out_dim(int in, int k, int stride, int pad) { return (in + 2 * pad - k) / stride + 1; }

int hdy_conv_mtiles(long long M) { return (int)((M + 127) / 128); }

int hdy_conv_stat_slabs(int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dtype) {
    int own = hdy_conv3x3_c64_slabs(N, H, W, C, K, R, S, stride, pad, dtype);
    if (own > 0) return own;
    own = hdy_conv3x3_c128_slabs(N, H, W, C, K, R, S, stride, pad, dtype);
    if (own > 0) return own;
    own = hdy_conv3x3s2_c32_slabs(N, H, W, C, K, R, S, stride, pad, dtype);
    if (own > 0) return own;
    const bool stem = C == 3 && R == 6 && S == 6 && stride == 2 && pad == 2;
    if (stem) {
        own = hdy_conv_stem_slabs(N, H, W, K, dtype);
        if (own > 0) return own;
    }
    const long long M = (long long)N * hdy_conv_out_dim(H, R, stride, pad) * hdy_conv_out_dim(W, S, stride, pad);
    if (!stem) {
        own = hdy_conv_deep_slabs(M, C, K, R * S, R == 1 && S == 1 && stride == 1 && pad == 0, dtype);
        if (own > 0) return own;
    }
    return hdy_conv_igemm_slabs(M, K, stem ? 6 : R * S);
}

size_t hdy_conv_pack_elems(int K, int C, int R, int S, int stride, int pad, int kind, int dtype) {
    if (kind == KIND_FWD) return block_elems(K, R * S * C, dtype);
    if (kind == KIND_STEM) return block_elems(K, R * S * 4, dtype);
    if (stride == 1) return block_elems(C, R * S * K, dtype);
    size_t n = 0;
    for (int a = 0; a < 2; ++a)
        for (int b = 0; b < 2; ++b) {
            const Axis ah = class_axis(R, pad, a), aw = class_axis(S, pad, b);
            if (ah.taps && aw.taps) n += block_elems(C, ah.taps * aw.taps * K, dtype);
        }
    return n;
}

static hdy_pack_desc make_desc(const float* w_a, int K_a, const float* w_b, int K_b, void* out, int Kl, int C, int R, int S, int transpose,
                               int TH, int TW, int rbase, int rstep, int sbase, int sstep, int stem, int rows_total, int Kdp, int dtype,
                               int first_block) {
    hdy_pack_desc d = {};
    d.w_a = w_a; d.w_b = w_b; d.out = out; d.K_a = K_a; d.K_b = K_b; d.Kl = Kl; d.C = C; d.R = R; d.S = S; d.transpose = transpose;
    d.TH = TH; d.TW = TW; d.rbase = rbase; d.rstep = rstep; d.sbase = sbase; d.sstep = sstep; d.stem = stem; d.rows_total = rows_total;
    d.Kdp = Kdp; d.dtype = dtype; d.first_block = first_block;
    d.nblocks = cdiv((long long)rows_total * Kdp, 2048);      // 256 threads x 8 outputs (conv_wgrad.hip PACK_PER_BLOCK)
    return d;
}

// Logical weight W[K][C][R][S]: rows 0..K_a-1 from w_a, the next K_b rows from w_b (two convs fused along K), remaining rows up to
// K are zero (channel padding, e.g. 39 -> 40 detection outputs).  Describes the packing job(s) for `kind`; returns their number.
int hdy_conv_pack_describe(const float* w_a, int K_a, const float* w_b, int K_b, int K, int C, int R, int S, int stride, int pad, int kind,
                           int dtype, void* out, hdy_pack_desc* descs, int first_block) {
    HDY_ARG(w_a && out && descs && K_a > 0 && K_b >= 0 && K >= K_a + K_b && C > 0 && R > 0 && S > 0, "conv_pack: bad args");
    HDY_ARG((K_b == 0) == (w_b == nullptr), "conv_pack: w_b / K_b mismatch");
    HDY_ARG(stride == 1 || stride == 2, "conv_pack: stride %d unsupported", stride);
    HDY_ARG(dtype == HDY_BF16 || dtype == HDY_F32, "conv_pack: unknown dtype %d", dtype);
    HDY_ARG(((long long)K + 256) * ((long long)R * S * ((long long)C + 256) + 64) < (1LL << 31), "conv_pack: weight too large for 32-bit packing indices");
    if (kind == KIND_FWD || kind == KIND_STEM) {
        const int stem = kind == KIND_STEM;
        HDY_ARG(!stem || (C == 3 && K_b == 0), "conv_pack: stem expects C == 3 and a single weight");
        const int kd = stem ? R * S * 4 : R * S * C;
        const int Kdp = round_up(kd, bke(dtype));
        const int rows_total = round_up(K, hdy_conv_bn_tile(K));
        descs[0] = make_desc(w_a, K_a, w_b, K_b, out, K, C, R, S, 0, R, stem ? 1 : S, 0, 1, 0, 1, stem, rows_total, Kdp, dtype, first_block);
        return 1;
    }
    HDY_ARG(kind == KIND_DGRAD, "conv_pack: unknown kind %d", kind);
    const int rows_total = round_up(C, hdy_conv_bn_tile(C));
    if (stride == 1) {
        const int Kdp = round_up(R * S * K, bke(dtype));
        descs[0] = make_desc(w_a, K_a, w_b, K_b, out, K, C, R, S, 1, R, S, R - 1, -1, S - 1, -1, 0, rows_total, Kdp, dtype, first_block);
        return 1;
    }
    size_t off = 0;
    int n = 0;
    for (int a = 0; a < 2; ++a)
        for (int b = 0; b < 2; ++b) {
            const Axis ah = class_axis(R, pad, a), aw = class_axis(S, pad, b);
            if (!ah.taps || !aw.taps) continue;
            const int Kdp = round_up(ah.taps * aw.taps * K, bke(dtype));
            descs[n] = make_desc(w_a, K_a, w_b, K_b, (char*)out + off * esize(dtype), K, C, R, S, 1, ah.taps, aw.taps, ah.rmax, -2, aw.rmax,
                                 -2, 0, rows_total, Kdp, dtype, first_block);
            first_block += descs[n].nblocks;
            off += (size_t)rows_total * Kdp;
            ++n;
        }
    return n;
}

int hdy_conv_pack(const float* w_a, int K_a, const float* w_b, int K_b, int K, int C, int R, int S, int stride, int pad, int kind,
                  int dtype, void* out, void* stream) {
    hdy_pack_desc d[4];
    const int n = hdy_conv_pack_describe(w_a, K_a, w_b, K_b, K, C, R, S, stride, pad, kind, dtype, out, d, 0);
    if (n < 0) return n;
    for (int i = 0; i < n; ++i) {
        const int rc = hdy_pack_weight_launch(d[i], (hipStream_t)stream);
        if (rc) return rc;
    }
    return HDY_OK;
}

int hdy_conv_pack_run(const hdy_pack_desc* descs_device, int ndesc, int total_blocks, void* stream) {
    HDY_ARG(descs_device && ndesc > 0 && total_blocks > 0, "conv_pack_run: bad args");
    return hdy_pack_batch_launch(descs_device, ndesc, total_blocks, (hipStream_t)stream);
}

// y = act(scale * conv(x, w) + shift) [+= y]; NHWC with pixel pitches; optional BatchNorm slabs in `stats`.
// stem != 0: x is the hdy_stem_prep() buffer [N][H+2*pad][W+2*pad][4] and (C,R,S,stride,pad) must be (3,6,6,2,2).
int hdy_conv_fwd(const void* x, int ldx, const void* w_packed, const float* scale, const float* shift, const void* res, int ldr, void* y,
                 int ldy, float* stats, int stat_slabs, int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int act, int accumulate,
                 int dtype, int out_f32, int stem, void* stream) {
    HDY_ARG(!stats || stat_slabs > 0, "conv_fwd: stats given with stat_slabs = %d", stat_slabs);
    HDY_ARG(stride >= 1 && R >= 1 && S >= 1 && pad >= 0, "conv_fwd: bad window");
    HDY_ARG(dtype == HDY_BF16 || dtype == HDY_F32, "conv_fwd: unknown dtype %d", dtype);
    ConvArgs a = {};
    a.x = x; a.w = w_packed; a.y = y; a.scale = scale; a.shift = shift; a.stats = stats; a.stat_cap = stat_slabs; a.res = res; a.ldr = ldr;
    HDY_ARG(!res || ldr >= K, "conv_fwd: residual pitch %d < K", ldr);
    a.N = N;
    a.Ho = hdy_conv_out_dim(H, R, stride, pad);
    a.Wo = hdy_conv_out_dim(W, S, stride, pad);
    HDY_ARG(a.Ho > 0 && a.Wo > 0, "conv_fwd: empty output");
    a.K = K; a.ldy = ldy;
    a.Hout = a.Ho; a.Wout = a.Wo; a.oh_mul = a.ow_mul = 1; a.oh_off = a.ow_off = 0; a.dense_out = 1;
    a.act = act; a.accumulate = accumulate;
    if (stem) {
        HDY_ARG(C == 3 && R == 6 && S == 6 && stride == 2 && pad == 2 && ldx == 4, "conv_fwd: stem expects C=3 k=6 s=2 p=2 on a 4-channel padded image");
        a.Hin = H + 2 * pad; a.Win = W + 2 * pad; a.C = 24; a.ldx = 4; a.span_pixels = 1;
        a.ih_mul = 2; a.iw_mul = 2; a.dh0 = 0; a.dw0 = 0; a.TH = 6; a.TW = 1;
        a.Kdp = round_up(6 * 24, bke(dtype));
        return hdy_conv_igemm_launch(a, dtype, out_f32, (hipStream_t)stream);
    }
    a.Hin = H; a.Win = W; a.C = C; a.ldx = ldx;
    a.ih_mul = stride; a.iw_mul = stride; a.dh0 = -pad; a.dw0 = -pad; a.TH = R; a.TW = S;
    a.Kdp = round_up(R * S * C, bke(dtype));
    return hdy_conv_igemm_launch(a, dtype, out_f32, (hipStream_t)stream);
}

// dx (+)= conv_transpose(dy, w): dx is [N][H][W][lddx] (C channels), dy is [N][Ho][Wo][lddy] (K channels).
static int dgrad_impl(const void* dy, int lddy, const void* w_packed_dgrad, void* dx, int lddx, int N, int H, int W, int C, int K, int R,
                      int S, int stride, int pad, int accumulate, int dtype, const hdy_stat_req* stats, int nstat, void* stream);

int hdy_conv_dgrad(const void* dy, int lddy, const void* w_packed_dgrad, void* dx, int lddx, int N, int H, int W, int C, int K, int R,
                   int S, int stride, int pad, int accumulate, int dtype, void* stream) {
    return dgrad_impl(dy, lddy, w_packed_dgrad, dx, lddx, N, H, W, C, K, R, S, stride, pad, accumulate, dtype, nullptr, 0, stream);
}

int hdy_conv_dgrad_stats(const void* dy, int lddy, const void* w_packed_dgrad, void* dx, int lddx, int N, int H, int W, int C, int K, int R,
                         int S, int stride, int pad, int accumulate, int dtype, const hdy_stat_req* stats, int nstat, void* stream) {
    HDY_ARG(nstat >= 0 && nstat <= 2 && (nstat == 0 || stats), "conv_dgrad_stats: bad request count");
    HDY_ARG(nstat == 0 || hdy_conv_dgrad_stat_slabs(N, H, W, C, K, R, S, stride, pad, dtype) > 0, "conv_dgrad_stats: this shape cannot serve statistics");
    return dgrad_impl(dy, lddy, w_packed_dgrad, dx, lddx, N, H, W, C, K, R, S, stride, pad, accumulate, dtype, stats, nstat, stream);
}

// workgroups of the data-gradient launch that would serve statistics: stride 1, or stride 2 as ONE class-walking launch (even H, W)
int hdy_conv_dgrad_stat_slabs(int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dtype) {
    if (dtype != HDY_BF16 || C % 8 || (stride != 1 && stride != 2)) return 0;
    if (stride == 1) return hdy_conv_igemm_stat_grid((long long)N * H * W, C, R * S, 1);
    if (H % 2 || W % 2 || hdy_opt(HDY_OPT_NO_CLASS_WALK)) return 0;
    for (int a = 0; a < 2; ++a)
        if (!class_axis(R, pad, a).taps || !class_axis(S, pad, a).taps) return 0;
    return hdy_conv_igemm_stat_grid((long long)N * (H / 2) * (W / 2), C, 1, 4);
}

static int dgrad_impl(const void* dy, int lddy, const void* w_packed_dgrad, void* dx, int lddx, int N, int H, int W, int C, int K, int R,
                      int S, int stride, int pad, int accumulate, int dtype, const hdy_stat_req* stats, int nstat, void* stream) {
    HDY_ARG(stride == 1 || stride == 2, "conv_dgrad: stride %d unsupported", stride);
    HDY_ARG(dtype == HDY_BF16 || dtype == HDY_F32, "conv_dgrad: unknown dtype %d", dtype);
    const int Ho = hdy_conv_out_dim(H, R, stride, pad), Wo = hdy_conv_out_dim(W, S, stride, pad);
    HDY_ARG(Ho > 0 && Wo > 0, "conv_dgrad: empty dy");
    ConvArgs a = {};
    a.x = dy; a.y = dx; a.N = N; a.Hin = Ho; a.Win = Wo; a.C = K; a.ldx = lddy;
    a.K = C; a.ldy = lddx; a.Hout = H; a.Wout = W;
    a.ih_mul = a.iw_mul = 1; a.accumulate = accumulate;
    a.nstat = nstat;
    for (int r = 0; r < nstat; ++r)
        a.stat[r] = StatReq{stats[r].y, stats[r].ldy, stats[r].scale, stats[r].shift, stats[r].slabs, stats[r].c0, stats[r].c1, stats[r].act, stats[r].nslabs};
    if (stride == 1) {
        a.w = w_packed_dgrad;
        a.Ho = H; a.Wo = W; a.oh_mul = a.ow_mul = 1; a.dense_out = 1;
        a.dh0 = pad - (R - 1); a.dw0 = pad - (S - 1); a.TH = R; a.TW = S;
        a.Kdp = round_up(R * S * K, bke(dtype));
        return hdy_conv_igemm_launch(a, dtype, 0, (hipStream_t)stream);
    }
    const int rows_total = round_up(C, hdy_conv_bn_tile(C));
    size_t off = 0;
    // One launch walking the four parity classes per spatial tile (conv_igemm.hip, `walk`): every class has the same Ho x Wo when H and W
    // are even.  As four launches each class wrote every other pixel of every other row (half cache lines, each line written by two
    // launches) and read dy from HBM again: 32<-64 @320x320 B=64 took 353 us against a 100 us bound.
    const bool no_walk = hdy_opt(HDY_OPT_NO_CLASS_WALK) != 0;
    if (H % 2 == 0 && W % 2 == 0 && !no_walk) {
        ConvArgs c = a;
        c.ncls = 4;
        c.Ho = H / 2; c.Wo = W / 2;
        c.oh_mul = c.ow_mul = 2; c.dense_out = 0;
        bool ok = true;
        for (int ca = 0; ca < 2 && ok; ++ca)
            for (int cb = 0; cb < 2; ++cb) {
                const Axis ah = class_axis(R, pad, ca), aw = class_axis(S, pad, cb);
                if (!ah.taps || !aw.taps) { ok = false; break; }
                const int i = ca * 2 + cb;
                const int Kdp = round_up(ah.taps * aw.taps * K, bke(dtype));
                c.c_dh[i] = ah.d0; c.c_dw[i] = aw.d0; c.c_TH[i] = ah.taps; c.c_TW[i] = aw.taps;
                c.c_nkb[i] = Kdp / bke(dtype); c.c_oh[i] = ca; c.c_ow[i] = cb; c.c_w[i] = (long long)off;
                off += (size_t)rows_total * Kdp;
            }
        if (ok) {
            c.w = w_packed_dgrad;
            c.dh0 = c.c_dh[0]; c.dw0 = c.c_dw[0]; c.TH = c.c_TH[0]; c.TW = c.c_TW[0]; c.oh_off = c.ow_off = 0;
            c.Kdp = c.c_nkb[0] * bke(dtype);
            int rc2 = HDY_OK;
            if (hdy_dgrad3x3s2_try(c, dtype, (hipStream_t)stream, &rc2)) return rc2;     // patch-resident kernel for the 32<-64 layer
            return hdy_conv_igemm_launch(c, dtype, 0, (hipStream_t)stream);
        }
        off = 0;
    }
    for (int ca = 0; ca < 2; ++ca)
        for (int cb = 0; cb < 2; ++cb) {
            const Axis ah = class_axis(R, pad, ca), aw = class_axis(S, pad, cb);
            ConvArgs c = a;
            c.Ho = (H - ca + 1) / 2; c.Wo = (W - cb + 1) / 2;
            c.oh_mul = c.ow_mul = 2; c.oh_off = ca; c.ow_off = cb; c.dense_out = 0;
            if (!ah.taps || !aw.taps) {
                HDY_ARG(false, "conv_dgrad: kernel %dx%d pad %d leaves a parity class without taps (unsupported)", R, S, pad);
            }
            if (c.Ho <= 0 || c.Wo <= 0) continue;
            c.dh0 = ah.d0; c.dw0 = aw.d0; c.TH = ah.taps; c.TW = aw.taps;
            c.Kdp = round_up(ah.taps * aw.taps * K, bke(dtype));
            c.w = (const char*)w_packed_dgrad + off * esize(dtype);
            off += (size_t)rows_total * c.Kdp;
            const int rc = hdy_conv_igemm_launch(c, dtype, 0, (hipStream_t)stream);
            if (rc) return rc;
        }
    return HDY_OK;
}

size_t hdy_conv_wgrad_workspace_bytes(int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dtype, int stem) {
    const int Ho = hdy_conv_out_dim(H, R, stride, pad), Wo = hdy_conv_out_dim(W, S, stride, pad);
    const int Q = stem ? R * S * 4 : R * S * C;
    int splits = 1, pps = 64;
    hdy_wgrad_plan(K, Q, (long long)N * Ho * Wo, dtype, &splits, &pps);
    if (stem) {
        const int g = hdy_wgrad_stem_grid(N, Ho, Wo, K, dtype);        // patch-resident stem kernel: one slab per workgroup
        if (g > splits) splits = g;
    }
    size_t bytes = (size_t)splits * K * Q * sizeof(float);
    if (!stem && R == 3 && S == 3 && pad == 1) {                       // patch-resident 3x3 kernel: its own split count
        const size_t b3 = hdy_wgrad3x3_workspace_bytes(N, Ho, Wo, C, K, stride, dtype);
        if (b3 > bytes) bytes = b3;
    }
    if (!stem) {                                                       // deep-pipelined 256 x 256 kernel: one slab per pixel split of its own plan
        const size_t bd = hdy_wgrad_deep_workspace_bytes(N, H, W, Ho, Wo, C, K, R, S, stride, dtype);
        if (bd > bytes) bytes = bd;
    }
    return bytes;
}

// grad_a [K_a][C][R][S] (and optionally grad_b [K_b][C][R][S], the lower rows of a stacked weight) (+)= dW.
int hdy_conv_wgrad(const void* x, int ldx, const void* dy, int lddy, int N, int H, int W, int C, int K, int R, int S, int stride, int pad,
                   float* grad_a, int K_a, float* grad_b, int K_b, int accumulate, void* workspace, size_t ws_bytes, int dtype, int stem,
                   void* stream) {
    HDY_ARG(grad_a && K_a > 0 && K_a + K_b <= K && K_b >= 0 && (K_b == 0) == (grad_b == nullptr), "conv_wgrad: bad gradient split");
    HDY_ARG(dtype == HDY_BF16 || dtype == HDY_F32, "conv_wgrad: unknown dtype %d", dtype);
    HDY_ARG(workspace && ws_bytes >= hdy_conv_wgrad_workspace_bytes(N, H, W, C, K, R, S, stride, pad, dtype, stem), "conv_wgrad: workspace too small");
    WgradArgs a = {};
    a.x = x; a.dy = dy; a.partial = (float*)workspace;
    a.N = N; a.K = K; a.lddy = lddy;
    a.Ho = hdy_conv_out_dim(H, R, stride, pad);
    a.Wo = hdy_conv_out_dim(W, S, stride, pad);
    HDY_ARG(a.Ho > 0 && a.Wo > 0, "conv_wgrad: empty dy");
    if (stem) {
        HDY_ARG(C == 3 && R == 6 && S == 6 && stride == 2 && pad == 2 && ldx == 4, "conv_wgrad: stem expects C=3 k=6 s=2 p=2 on a 4-channel padded image");
        a.Hin = H + 2 * pad; a.Win = W + 2 * pad; a.C = 24; a.ldx = 4; a.span_pixels = 1;
        a.ih_mul = a.iw_mul = 2; a.dh0 = a.dw0 = 0; a.TH = 6; a.TW = 1;
    } else {
        a.Hin = H; a.Win = W; a.C = C; a.ldx = ldx;
        a.ih_mul = a.iw_mul = stride; a.dh0 = a.dw0 = -pad; a.TH = R; a.TW = S;
    }
    const int Q = a.TH * a.TW * a.C;
    const int stem_grid = stem ? hdy_wgrad_stem_grid(N, a.Ho, a.Wo, K, dtype) : 0;
    int rc;
    if (!stem && R == 3 && S == 3 && pad == 1 &&
        hdy_wgrad3x3_try(x, ldx, dy, lddy, N, H, W, a.Ho, a.Wo, C, K, stride, a.partial, dtype, (hipStream_t)stream, &a.splits, &rc)) {
        // patch-resident kernel launched (conv_wgrad3x3.hip)
    } else if (!stem && hdy_wgrad_deep_try(x, ldx, dy, lddy, N, H, W, a.Ho, a.Wo, C, K, R, S, stride, pad, a.partial, dtype, (hipStream_t)stream, &a.splits, &rc)) {
        // deep-pipelined 256 x 256 kernel launched (conv_wgrad_deep.hip)
    } else if (stem_grid > 0) {
        a.splits = stem_grid;
        rc = hdy_wgrad_stem_launch(a, stem_grid, (hipStream_t)stream);
    } else {
        hdy_wgrad_plan(K, Q, (long long)N * a.Ho * a.Wo, dtype, &a.splits, &a.pix_per_split);
        rc = hdy_wgrad_launch(a, dtype, (hipStream_t)stream);
    }
    if (rc) return rc;
    const int mode = stem ? 1 : 0;
    rc = hdy_wgrad_reduce_launch(a.partial, a.splits, (size_t)K * Q, K_a, Q, mode, C, R, S, grad_a, accumulate, (hipStream_t)stream);
    if (rc) return rc;
    if (K_b) {
        // rows K_a.. of every slab belong to the second tensor
        rc = hdy_wgrad_reduce_launch(a.partial + (size_t)K_a * Q, a.splits, (size_t)K * Q, K_b, Q, mode, C, R, S, grad_b, accumulate,
                                     (hipStream_t)stream);
    }
    return rc;
}

// The stem's weight gradient with the BatchNorm / SiLU backward of its unit applied while the tile is staged (conv_wgrad.hip,
// wgrad_stem_kernel<.., true>): dz = gradient of the unit's output, y = its raw conv output, c1 / c2 from the statistics pass
// (hdy_bn_act_bwd with dy == NULL).  The stem has no data gradient, so dy is never materialised.  bf16, K in {16, 32, 64}.
int hdy_conv_wgrad_stem_fused_ok(int N, int H, int W, int K) {
    if (K != 16 && K != 32 && K != 64) return 0;
    const int Ho = hdy_conv_out_dim(H, 6, 2, 2), Wo = hdy_conv_out_dim(W, 6, 2, 2);
    return hdy_wgrad_stem_grid(N, Ho, Wo, K, HDY_BF16) > 0 ? 1 : 0;
}

int hdy_conv_wgrad_stem_fused(const void* x, const void* dz, int lddz, const void* y, int ldy, const float* scale, const float* shift, const float* mean,
                              const float* invstd, const float* c1, const float* c2, int N, int H, int W, int K, float* grad_a, int K_a, float* grad_b,
                              int K_b, int accumulate, void* workspace, size_t ws_bytes, void* stream) {
    HDY_ARG(x && dz && y && grad_a && K_a > 0 && K_a + K_b <= K && K_b >= 0 && (K_b == 0) == (grad_b == nullptr), "conv_wgrad_stem_fused: bad arguments");
    HDY_ARG(hdy_conv_wgrad_stem_fused_ok(N, H, W, K), "conv_wgrad_stem_fused: shape not served (K in {16, 32, 64}, output a multiple of 16 x 32)");
    HDY_ARG(workspace && ws_bytes >= hdy_conv_wgrad_workspace_bytes(N, H, W, 3, K, 6, 6, 2, 2, HDY_BF16, 1), "conv_wgrad_stem_fused: workspace too small");
    WgradArgs a = {};
    a.x = x; a.dy = dz; a.lddy = lddz; a.partial = (float*)workspace;
    a.y = y; a.ldy = ldy; a.bn_scale = scale; a.bn_shift = shift; a.bn_mean = mean; a.bn_invstd = invstd; a.bn_c1 = c1; a.bn_c2 = c2;
    a.N = N; a.K = K;
    a.Ho = hdy_conv_out_dim(H, 6, 2, 2);
    a.Wo = hdy_conv_out_dim(W, 6, 2, 2);
    a.Hin = H + 4; a.Win = W + 4; a.C = 24; a.ldx = 4; a.span_pixels = 1;
    a.ih_mul = a.iw_mul = 2; a.dh0 = a.dw0 = 0; a.TH = 6; a.TW = 1;
    const int Q = a.TH * a.TW * a.C;
    a.splits = hdy_wgrad_stem_grid(N, a.Ho, a.Wo, K, HDY_BF16);
    int rc = hdy_wgrad_stem_launch(a, a.splits, (hipStream_t)stream);
    if (rc) return rc;
    rc = hdy_wgrad_reduce_launch(a.partial, a.splits, (size_t)K * Q, K_a, Q, 1, 3, 6, 6, grad_a, accumulate, (hipStream_t)stream);
    if (rc) return rc;
    if (K_b) rc = hdy_wgrad_reduce_launch(a.partial + (size_t)K_a * Q, a.splits, (size_t)K * Q, K_b, Q, 1, 3, 6, 6, grad_b, accumulate, (hipStream_t)stream);
    return rc;
}

}  // extern "C"

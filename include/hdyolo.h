/* hdyolo.h — C ABI of libhdyolo_hip.so: the MI355X (gfx950) kernels behind hd_yolo's metayolo detection hot path.
 *
 * The reference (impromptuRong/hd_yolo) is pure Python on PyTorch: it has no FFI / plugin registry.  The native
 * boundary of its hot path is wherever ATen / torchvision are entered, so each entry point below names the
 * reference call site whose native work it replaces (paths relative to the reference repo root).
 * INTEGRATION.md shows the ctypes binding a maintainer of the reference would add.
 *
 * Conventions
 *  - Plain pointers and sizes only; every buffer (inputs, outputs, workspaces) is owned by the caller.
 *    The library allocates nothing, keeps no pointers after a call returns and never synchronises.
 *  - All work is enqueued on `stream` (a hipStream_t passed as void*; NULL = the null stream).
 *  - Return value: 0 = ok; < 0 = invalid argument / unsupported shape; > 0 = a hipError_t from the launch.
 *    hdy_last_error() returns a thread-local description of the last failure.  Re-entrant.  Process state is limited to: the
 *    option table (atomics, initialised once from the environment under std::call_once, changed by hdy_set_option), per-kernel
 *    once-flags for the "dynamic LDS size" function attribute, and the thread-local error text / dispatch log.
 *  - Activations are NHWC ("channels last") with an explicit pixel pitch `ld*` in ELEMENTS, so a tensor may be
 *    a channel slice of a wider buffer (this is how torch.cat along C costs nothing).  Framework weights stay
 *    in the reference layout [K][C][R][S] fp32 and are re-packed by hdy_conv_pack.
 *  - dtype: HDY_F32 (exact fp32 MFMA, parity mode) or HDY_BF16 (bf16 operands, fp32 accumulate).  Per-channel
 *    vectors (scale/shift/statistics/gradients of them) are always fp32.
 *  - Alignment: activation and packed-weight pointers 16 bytes; channel counts and pitches multiples of one
 *    16-byte vector (8 bf16 / 4 f32) unless stated otherwise.
 */
#ifndef HDYOLO_H
#define HDYOLO_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HDY_F32 0
#define HDY_BF16 1

#define HDY_PACK_FWD 0   /* operand of hdy_conv_fwd / layout of hdy_conv_wgrad results */
#define HDY_PACK_DGRAD 1 /* operand of hdy_conv_dgrad */
#define HDY_PACK_STEM 2  /* operand of hdy_conv_fwd(stem=1) */

#define HDY_ACT_NONE 0
#define HDY_ACT_SILU 1
#define HDY_ACT_RELU 2 /* the Mask-RCNN head's convolutions (row f2) */

const char* hdy_last_error(void);
/* ABI revision of THIS header: bumped whenever an entry point's parameter list, a structure or an option changes meaning.  hdy_version() returns the
 * value the library was built with; a binding written against another revision must refuse the library (hd_yolo_amd/_lib.py:load does) — with
 * plain pointers and sizes a mismatched parameter list would otherwise shift arguments silently. */
#define HDY_ABI_VERSION 6
int hdy_version(void);
/* Which kernel ran: every launcher names the kernel family it picked ("igemm_128x128x2", "conv3x3_c64", "deep_256x128", "wgrad3x3", ...).
 * hdy_last_dispatch: the last pick on this thread; hdy_dispatch_log: every pick of every thread since hdy_dispatch_log_reset(), in launch
 * order, ';'-separated (process-wide under a mutex — autograd runs the backward list on its own thread — first 32 KB).  The reference has no counterpart (ATen picks its kernels silently); the tests use it to assert that the shapes meant to hit
 * a specialised kernel do. */
const char* hdy_last_dispatch(void);
const char* hdy_dispatch_log(void);
void hdy_dispatch_log_reset(void);
/* Process-wide kernel-selection switches by the name of their environment variable (HDY_NO_CONV3X3, HDY_NO_DEEP, HDY_WGRAD_BLOCKS, ...:
 * csrc/common.h HdyOption).  hdy_set_option returns the previous value (< 0: unknown name).  They also steer the sizing queries, so set
 * them before sizing buffers / building plans. */
int hdy_set_option(const char* name, int value);
int hdy_get_option(const char* name);
/* host-only: the reciprocal conv_igemm.hip divides row indices by (n / d == mulhi(2n, *magic) >> *shift for n < 2^31) */
int hdy_fastdiv_magic(unsigned d, unsigned* magic, int* shift);

/* ---- convolution family --------------------------------------------------------------------------------------
 * Replaces nn.Conv2d forward/backward as used by metayolo/models/layers.py:31,37-41 (Conv), :92-97 (Bottleneck),
 * :124-131 (C3), :179-189 (SPPF) and metayolo/models/yolo_head.py:112,142 (Detect's 1x1 conv with bias);
 * backward is what train.py:472 `scaler.scale(loss).backward()` reaches through autograd. */
int hdy_conv_out_dim(int in, int k, int stride, int pad);
int hdy_conv_mtiles(long long M); /* upper bound of the statistic slab count: one per 128 output pixels */
/* Number of [2][K] BatchNorm statistic slabs hdy_conv_fwd writes for this layer (kernel-dependent: one per 128 output pixels in
 * the generic kernel, one per workgroup in the filter-resident 3x3 kernel).  hdy_bn_finalize takes the same number. */
int hdy_conv_stat_slabs(int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dtype);

size_t hdy_conv_pack_elems(int K, int C, int R, int S, int stride, int pad, int kind, int dtype);
/* Logical weight [K][C][R][S] = rows of w_a, then rows of w_b (two convs fused along K; may be NULL/0), then zero rows up to K. */
int hdy_conv_pack(const float* w_a, int K_a, const float* w_b, int K_b, int K, int C, int R, int S, int stride, int pad, int kind,
                  int dtype, void* out, void* stream);

/* Batched packing: one launch re-packs every layer's weights after an optimizer step.  hdy_conv_pack_describe fills 1 (or 4:
 * stride-2 dgrad parity classes) descriptors on the HOST for the same arguments as hdy_conv_pack and returns how many it wrote
 * (first_block = running block count of the batch; each descriptor reports its nblocks); the caller copies all descriptors into
 * device memory once and replays hdy_conv_pack_run(table, n, total_blocks) every step. */
typedef struct hdy_pack_desc {
    const float* w_a;
    const float* w_b;
    void* out;
    int K_a, K_b, Kl, C, R, S, transpose, TH, TW, rbase, rstep, sbase, sstep, stem, rows_total, Kdp, dtype;
    int first_block, nblocks;
    int pad_;
} hdy_pack_desc;
int hdy_conv_pack_describe(const float* w_a, int K_a, const float* w_b, int K_b, int K, int C, int R, int S, int stride, int pad, int kind,
                           int dtype, void* out, hdy_pack_desc* descs_host, int first_block);
int hdy_conv_pack_run(const hdy_pack_desc* descs_device, int ndesc, int total_blocks, void* stream);

/* y = act(scale[k] * conv(x, w)[.., k] + shift[k]) + res (+= y when accumulate).  scale/shift/res may be NULL (1 / 0 / none).
 * stats (optional, train-mode BN): [stat_slabs][2][K] floats, partial sums and sums of squares of the raw convolution (before
 * scale/shift/act); stat_slabs = what hdy_conv_stat_slabs(...) returned when the caller sized the array.  The launcher compares it with the
 * number of slabs the kernel it is about to start writes and returns HDY_EINVAL on a mismatch (a kernel-selection switch flipped
 * between the sizing query and the launch would otherwise write past the array or leave stale slabs for hdy_bn_finalize).  out_f32: write fp32 even when dtype is bf16 (detection logits).
 * stem: x is the hdy_stem_prep buffer; requires C=3, R=S=6, stride=2, pad=2, ldx=4. */
int hdy_conv_fwd(const void* x, int ldx, const void* w_packed, const float* scale, const float* shift, const void* res, int ldr, void* y,
                 int ldy, float* stats, int stat_slabs, int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int act, int accumulate,
                 int dtype, int out_f32, int stem, void* stream);

/* dx[N][H][W][C] (+)= conv_transpose(dy[N][Ho][Wo][K], w); stride 1 or 2. */
int hdy_conv_dgrad(const void* dy, int lddy, const void* w_packed_dgrad, void* dx, int lddx, int N, int H, int W, int C, int K, int R,
                   int S, int stride, int pad, int accumulate, int dtype, void* stream);

/* grad_a[K_a][C][R][S] (and grad_b[K_b][..] = the following K_b output channels, for two convs fused along K)
 * (+)= dW, fp32, deterministic (slab reduction, no atomics). */
size_t hdy_conv_wgrad_workspace_bytes(int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dtype, int stem);
int hdy_conv_wgrad(const void* x, int ldx, const void* dy, int lddy, int N, int H, int W, int C, int K, int R, int S, int stride, int pad,
                   float* grad_a, int K_a, float* grad_b, int K_b, int accumulate, void* workspace, size_t ws_bytes, int dtype, int stem,
                   void* stream);

/* Fused backward of a 1x1 Conv + BatchNorm(train) + SiLU unit (metayolo/models/layers.py:31-38 under autograd, train.py:472): from
 * the output gradient dz ([0,Ka) from dz_a, [Ka,K) from dz_b), the raw conv output y, the BatchNorm coefficients and c1 / c2 of
 * hdy_bn_act_bwd(dy = NULL), computes dy = scale*(dz*silu'(u) - c1 - xhat*c2) on the fly and from it BOTH dx (+)= dy * W (NULL: skipped)
 * and the weight gradient grad_a / grad_b (+)= dy^T * x (NULL: skipped) in one pass: dy never goes to HBM.  bf16, C == K in {32, 64, 128}
 * (hdy_conv1x1_bwd_fused_ok); w_packed_dgrad = hdy_conv_pack(kind = HDY_PACK_DGRAD). */
/* Producer-side BatchNorm-backward statistics.  The kernel that writes the LAST contribution of a gradient tensor (a data-gradient
 * launch) can also serve the reduce pass of the Conv+BN+act unit(s) whose output gradient that tensor is: for its output channels
 * [c0, c1) it reads the unit's raw conv output y (same pixels; channel c0 <-> element 0 of y / scale / shift), forms
 * du = dz * act'(y*scale + shift) and writes per workgroup one fp32 slab [2][c1 - c0] = (SUM du, SUM du*y) to slabs[wg][2][c1-c0].
 * hdy_bn_bwd_finalize_slabs (SUM du*xhat = invstd * (SUM du*y - mean * SUM du)) then gives dgamma / dbeta / c1 / c2 without the
 * unit's own pass over dz and y.  Served by launches whose gradient is at most 64 channels wide (the wider instances have no
 * registers to spare): hdy_conv_dgrad_stat_slabs / hdy_conv1x1_bwd_fused_stat_slabs return 0 otherwise.
 * (reference: what autograd's BatchNorm backward reduces, metayolo/models/layers.py:37 under train.py:472) */
typedef struct {
    const void* y; int ldy;
    const float *scale, *shift;
    float* slabs;
    int c0, c1, act;
    int nslabs;          /* slabs the caller's array holds (the hdy_*_stat_slabs query it was sized with): HDY_EINVAL unless the launch writes exactly that many */
} hdy_stat_req;
int hdy_conv1x1_bwd_fused_stat_slabs(long long M, int C, int K, int dtype);
/* slabs a stats-serving launch writes (= its workgroups); 0: this shape cannot serve statistics */
int hdy_conv_dgrad_stat_slabs(int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dtype);
int hdy_conv_dgrad_stats(const void* dy, int lddy, const void* w_packed_dgrad, void* dx, int lddx, int N, int H, int W, int C, int K, int R,
                         int S, int stride, int pad, int accumulate, int dtype, const hdy_stat_req* stats, int nstat, void* stream);
int hdy_conv1x1_bwd_fused_stats(const void* dz_a, int lddz_a, const void* dz_b, int lddz_b, int Ka, const void* y, int ldy, const float* scale,
                                const float* shift, const float* mean, const float* invstd, const float* c1, const float* c2, const void* x, int ldx,
                                const void* w_packed_dgrad, void* dx, int lddx, int accumulate_dx, float* grad_a, int K_a, float* grad_b, int K_b,
                                int accumulate_w, long long M, int C, int K, void* workspace, size_t ws_bytes, int dtype, const hdy_stat_req* stats,
                                int nstat, void* stream);
/* from `nslabs` slabs [2][K] of (SUM du, SUM du*y): dbeta (+)= SUM du, dgamma (+)= invstd * (SUM du*y - mean * SUM du);
 * c1 = dbeta / count, c2 = dgamma / count (either may be NULL) */
int hdy_bn_bwd_finalize_slabs(const float* slabs, int nslabs, int K, long long count, const float* mean, const float* invstd, float* dgamma,
                              float* dbeta, int accumulate, float* c1, float* c2, void* stream);
/* the apply pass alone: dy = scale * (dz*act'(u) - c1 - xhat*c2) with c1 / c2 given */
int hdy_bn_act_bwd_apply(const void* dz, int lddz, const void* dz_b, int lddz_b, int Ka, const void* y, int ldy, const float* scale,
                         const float* shift, const float* mean, const float* invstd, const float* c1, const float* c2, void* dy, int lddy,
                         long long M, int K, int act, int dtype, void* stream);
int hdy_conv1x1_bwd_fused_ok(int C, int K, int dtype);
int hdy_conv1x1_bwd_fused_grid(long long M, int K);
size_t hdy_conv1x1_bwd_fused_workspace_bytes(long long M, int C, int K);
int hdy_conv1x1_bwd_fused(const void* dz_a, int lddz_a, const void* dz_b, int lddz_b, int Ka, const void* y, int ldy, const float* scale,
                          const float* shift, const float* mean, const float* invstd, const float* c1, const float* c2, const void* x, int ldx,
                          const void* w_packed_dgrad, void* dx, int lddx, int accumulate_dx, float* grad_a, int K_a, float* grad_b, int K_b,
                          int accumulate_w, long long M, int C, int K, void* workspace, size_t ws_bytes, int dtype, void* stream);

/* ---- BatchNorm + SiLU (+ residual) ---------------------------------------------------------------------------
 * Replaces nn.BatchNorm2d (eps 1e-3, momentum 0.03: metayolo/models/utils_torch.py:47-49) and nn.SiLU in
 * Conv.forward (metayolo/models/layers.py:37-38), the shortcut add of Bottleneck.forward (:97), their backward,
 * and the eval-time folding of fuse_conv_and_bn (metayolo/models/utils_torch.py:79-99). */
/* stats: [mtiles][2][stats_ld] slabs from hdy_conv_fwd (channel slice of K).  workspace (optional, enables the parallel
 * two-stage reduction for mtiles > 1024): ws_bytes >= hdy_bn_finalize_workspace_bytes(mtiles, K), 8-byte aligned (HDY_EINVAL when a
 * non-NULL workspace is smaller). */
size_t hdy_bn_finalize_workspace_bytes(int mtiles, int K);
int hdy_bn_finalize(const float* stats, int stats_ld, int mtiles, int K, long long count, const float* gamma, const float* beta, float* running_mean,
                    float* running_var, float eps, float momentum, float* scale, float* shift, float* save_mean, float* save_invstd,
                    void* workspace, size_t ws_bytes, void* stream);
int hdy_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean, const float* running_var, float eps, int K,
                       float* scale, float* shift, void* stream);
/* The same for every BatchNorm of an inference plan in ONE launch: the caller fills the descriptors on the host, copies them into
 * device memory once and replays hdy_bn_eval_coeffs_batch before every forward (57 launches of 5 us otherwise for yolov5s). */
typedef struct hdy_bn_eval_desc {
    const float* gamma;
    const float* beta;
    const float* running_mean;
    const float* running_var;
    float* scale;
    float* shift;
    int K;
    float eps;
} hdy_bn_eval_desc;
int hdy_bn_eval_coeffs_batch(const hdy_bn_eval_desc* descs_device, int ndesc, void* stream);
int hdy_bn_act_fwd(const void* y, int ldy, const float* scale, const float* shift, const void* res, int ldr, void* z, int ldz,
                   long long M, int K, int act, int dtype, void* stream);
int hdy_bn_bwd_blocks(long long M);
/* workspace: ws_bytes >= hdy_bn_bwd_workspace_bytes(M, K) = (hdy_bn_bwd_blocks(M) * 2 * K + 2 * K) floats (HDY_EINVAL otherwise) */
size_t hdy_bn_bwd_workspace_bytes(long long M, int K);
/* mean == invstd == NULL: frozen statistics (FrozenBatchNorm2d, metayolo/models/utils_torch.py:180-203): dy = scale * dz * act'(u),
 * dgamma / dbeta untouched. */
int hdy_bn_act_bwd(const void* dz, int lddz, const void* y, int ldy, const float* scale, const float* shift, const float* mean,
                   const float* invstd, void* dy, int lddy, float* dgamma, float* dbeta, int accumulate, long long M, int K, int act,
                   int dtype, float* workspace, size_t ws_bytes, void* stream);
/* dy == NULL: statistics only (reduce + finalize: dgamma, dbeta, and c1 = dbeta/M, c2 = dgamma/M left in
 * workspace[hdy_bn_bwd_blocks(M)*2*K .. +2K)) for a consumer that applies them itself (hdy_conv1x1_bwd_fused). */
/* The same three passes for the PAIR of BatchNorms behind a C3's cv1 | cv2 (metayolo/models/layers.py:126-131: both read the same
 * input, so their convolutions run as one K = Ka + Kb wide launch): BatchNorm is per channel, so the pair is one K-wide layer; only
 * what the two modules own separately splits at Ka — parameters, running statistics and parameter gradients (finalize), the two
 * outputs (z_a: cv1's activation, z_b: cv2's slice of the concat buffer) and the two gradient sources (dz_a, dz_b).  One pass over
 * the 2c-wide raw tensor instead of two passes over half-width slices (K = 32 halves read 64 of every 128 bytes). */
int hdy_bn_finalize_pair(const float* stats, int stats_ld, int mtiles, int K, int Ka, long long count, const float* gamma_a, const float* beta_a,
                         float* running_mean_a, float* running_var_a, const float* gamma_b, const float* beta_b, float* running_mean_b,
                         float* running_var_b, float eps, float momentum, float* scale, float* shift, float* save_mean, float* save_invstd,
                         void* workspace, size_t ws_bytes, void* stream);
int hdy_bn_act_fwd_pair(const void* y, int ldy, const float* scale, const float* shift, void* z_a, int ldz_a, void* z_b, int ldz_b, int Ka,
                        long long M, int K, int act, int dtype, void* stream);
int hdy_bn_act_bwd_pair(const void* dz_a, int lddz_a, const void* dz_b, int lddz_b, int Ka, const void* y, int ldy, const float* scale,
                        const float* shift, const float* mean, const float* invstd, void* dy, int lddy, float* dgamma_a, float* dbeta_a,
                        float* dgamma_b, float* dbeta_b, int accumulate, long long M, int K, int act, int dtype, float* workspace, size_t ws_bytes, void* stream);
int hdy_add_inplace(void* out, int ldo, const void* a, int lda, long long M, int K, int dtype, void* stream);
/* out[k] (+)= sum_m dz[m][k]: bias gradient of Detect's conv (yolo_head.py:112).  workspace: ws_bytes >= hdy_colsum_workspace_bytes(M, K) = hdy_bn_bwd_blocks(M)*2*K floats */
size_t hdy_colsum_workspace_bytes(long long M, int K);
int hdy_colsum(const void* dz, int lddz, long long M, int K, float* out, int accumulate, int dtype, float* workspace, size_t ws_bytes, void* stream);

/* ---- SPPF pooling, upsample, layout --------------------------------------------------------------------------
 * Replaces nn.MaxPool2d(5,1,2) x3 of SPPF.forward (metayolo/models/layers.py:181-189), nn.Upsample(None,2,'nearest')
 * (hub yaml fpn rows) and the NCHW image hand-off (train.py:432, val_nuclei.py:135). */
int hdy_sppf_pool_fwd(const void* x, void* y1, void* y2, void* y3, int ld, unsigned char* idx1, unsigned char* idx2, unsigned char* idx3,
                      int N, int H, int W, int C, int dtype, void* stream);
int hdy_sppf_pool_bwd(const void* g0, const void* g1, const void* g2, const void* g3, int ldg, const unsigned char* idx1,
                      const unsigned char* idx2, const unsigned char* idx3, void* dx, int lddx, int N, int H, int W, int C, int dtype,
                      void* stream);
int hdy_upsample2x_fwd(const void* x, int ldx, void* y, int ldy, int N, int H, int W, int C, int dtype, void* stream);
int hdy_upsample2x_bwd(const void* dy, int lddy, void* dx, int lddx, int N, int H, int W, int C, int accumulate, int dtype, void* stream);
int hdy_stem_prep(const float* img_nchw, void* out, int B, int H, int W, int pad, int dtype, void* stream);
int hdy_nchw_to_nhwc(const float* src, void* dst, int ldd, int N, int C, int H, int W, int dtype, void* stream);

/* ---- detection head ------------------------------------------------------------------------------------------
 * hdy_decode replaces Detect.compute_proposals and the level-id pad + cat of compute_outputs
 * (metayolo/models/yolo_head.py:185-213, :311-312, :419-429): logits (b,a,y,x,o) addressed through element strides
 * (sb,sa,sy,sx; o contiguous) -> out[b][row_offset + (a*ny + y)*nx + x][0..no] = cx,cy,w,h (pixels), sigmoid(obj),
 * sigmoid(cls..), level id.  anchor_px: na*2 HOST floats (anchor w,h in pixels).
 *
 * hdy_nms_batched replaces nms_per_image (metayolo/models/utils_general.py:299-356; class_aware=0, the path's
 * default: class-agnostic, ranked by objectness, boxes with w or h < min_wh dropped, obj > conf strict) and
 * non_max_suppression (:423-523; class_aware=1) including torchvision.ops.nms / remove_small_boxes.
 * preds [B][N][row] fp32 with row = 5 + nc + extra.  Outputs per tile: keep[max_det] original row indices in
 * descending-score order (stable: ties by lower row), -1 padded; n_keep; and the gathered rows.
 * max_det: any positive value, as the reference's `[:max_det]` slice (utils_general.py:342); up to 4096 the kept list lives in LDS, beyond
 * that in the workspace (round 6).  workspace (16-byte aligned): hdy_nms_workspace_bytes_for(B, N, max_det) (= hdy_nms_workspace_bytes(B, N)
 * for max_det <= 4096). */
int hdy_decode(const float* det, long long sb, long long sa, long long sy, long long sx, const float* anchor_px, float stride, float* out,
               int row_offset, int rows_per_image, int level_id, int B, int na, int ny, int nx, int no, void* stream);
/* autograd's logits gradient (b,a,y,x,o; element strides) -> NHWC [B][ny][nx][ldo] of dtype, channel a*no+o, zero padded */
int hdy_det_grad_pack(const float* g, long long sb, long long sa, long long sy, long long sx, long long so, void* out, int ldo, int B, int na,
                      int ny, int nx, int no, int dtype, void* stream);
size_t hdy_nms_workspace_bytes(int B, int N);
size_t hdy_nms_workspace_bytes_for(int B, int N, int max_det);
int hdy_nms_batched(const float* preds, int B, int N, int row, int nc, float conf, float iou, int max_det, float min_wh, int class_aware,
                    long long* keep, int* n_keep, float* out_boxes, float* out_scores, float* out_extra, float* out_conf, int* out_cls,
                    void* workspace, size_t ws_bytes, void* stream);

/* Tail of Detect.compute_outputs (metayolo/models/yolo_head.py:335-345) on hdy_nms_batched's padded rows, whole batch, one launch:
 * hierarchical scores in place on scores [B][max_det][1 + nc] (pairs: npairs x (child, parent) column indices in the order the reference
 * applies them: child *= parent), then per kept box score / label (best class if > conf, else objectness / -100; multi_label: all 1 + nc
 * scores and score > conf flags).  Outputs are COMPACTED: image b's rows start at n_keep[0] + .. + n_keep[b - 1].  out_boxes [T][4],
 * out_scores [T] (multi_label: [T][1 + nc]), out_labels int64 [T] (multi_label: uint8 [T][1 + nc]) with T >= the total; offsets [B + 1]
 * (optional) receives the row offsets. */
int hdy_det_outputs(float* scores, const float* boxes, const int* n_keep, int B, int max_det, int nc, const int* pairs, int npairs, float conf,
                    int multi_label, float* out_boxes, float* out_scores, void* out_labels, int* offsets, void* stream);

/* torchvision.ops.nms on explicit boxes, as the reference calls it outside nms_per_image (Ensemble.merge, metayolo/models/yolo.py:189-199):
 * boxes_scores [B][N][5] = (x1, y1, x2, y2, score >= 0) fp32; every row is a candidate; keep[B][max_det] row indices in descending
 * score order (stable), -1 padded; n_keep[B].  Same kernel and workspace rule as hdy_nms_batched (any max_det). */
int hdy_nms_boxes(const float* boxes_scores, int B, int N, float iou, int max_det, long long* keep, int* n_keep, void* workspace,
                  size_t ws_bytes, void* stream);

/* ---- mask branch primitives ------------------------------------------------------------
 * hdy_roi_align_fwd/bwd replace torchvision.ops.roi_align as the reference calls it (metayolo/models/yolo_head.py:243 on ground
 * truth boxes in training, :294 multiscale_roi_align on detections): feat NHWC [B][H][W][ldf] (C channels), rois [R][5] fp32
 * (image index, x1, y1, x2, y2 in input pixels), out NHWC [R][P][P][C]; sampling_ratio^2 bilinear samples per bin.
 * The backward scatters dout into dfeat_f32 [B][H][W][C] (fp32, zeroed by the caller) with atomic adds; hdy_cast_store moves
 * that image into a pitched gradient view (dst[m][c] (+)= src[m][c]).  hdy_relu_bwd: du = dz * (y > 0) for the head's
 * conv + bias + ReLU layers (torchvision MaskRCNNHeads / MaskRCNNPredictor, yolo_head.py:125-128). */
int hdy_roi_align_fwd(const void* feat, int ldf, int B, int H, int W, int C, const float* rois, int R, float spatial_scale, int P,
                      int sampling_ratio, int aligned, void* out, int dtype, void* stream);
int hdy_roi_align_bwd(const void* dout, float* dfeat_f32, int B, int H, int W, int C, const float* rois, int R, float spatial_scale, int P,
                      int sampling_ratio, int aligned, int dtype, void* stream);
int hdy_relu_bwd(const void* dz, const void* y, void* du, long long n, int dtype, void* stream);
int hdy_cast_store(const float* src, void* dst, int ldd, long long M, int C, int accumulate, int dtype, void* stream);

/* ---- fused detection loss --------------------------------------------------------------
 * Replaces Detect.matcher (metayolo/models/yolo_head.py:358-417), DetLoss.forward (metayolo/models/loss.py:190-244) with
 * bbox_iou(CIoU) (metayolo/models/utils_general.py:193-231) and their autograd backward: target assignment, CIoU box loss,
 * objectness / class BCE, and the gradient w.r.t. the logits, written into the NHWC buffers the backward plan consumes.
 * logits[l]: fp32 [B][ny][nx][ldl], channel a*no+o;  gdet[l]: dtype [B][ny][nx][ldg] (ldl, ldg multiples of 4, 16-byte aligned
 * bases; channels >= na*no of gdet are written as zero);  anchors_grid: nl*na*2 HOST floats in grid
 * units; balance: nl HOST floats; gts: device [nt][5] (img, cx, cy, w, h normalised); tcls: device [nt][nc] class targets;
 * cls_cw: nc HOST floats.  out: device [4] = loss (x batch), box, obj, cls items.  Supported: fl_gamma = 0, no autobalance.
 * workspace: hdy_det_loss_workspace_bytes(..., nt) bytes for calls with up to nt targets (sums, one list head per cell and anchor,
 * one record per possible match: nl * 5 * na * nt of them). */
size_t hdy_det_loss_workspace_bytes(int nl, const int* ny, const int* nx, int B, int na, int nc, int nt);
int hdy_det_loss(const float* const* logits, int ldl, void* const* gdet, int ldg, int dtype, const int* ny, const int* nx, int nl, int B,
                 int na, int nc, const float* anchors_grid, const float* balance, const float* gts, const float* tcls, int nt,
                 const float* cls_cw, float cls_pw, float obj_pw, float anchor_t, float label_smoothing, float h_box, float h_obj, float h_cls,
                 float* out, void* workspace, size_t ws_bytes, void* stream);
/* Which matched cell of each target feeds the mask branch (metayolo/models/yolo_head.py:231-262: per target the matched cell whose decoded
 * box has the best IoU with the truth — first in the reference's row order on ties — kept when that IoU >= min_iou = 0.8).  Same logits /
 * geometry / anchors_grid / gts / anchor_t as hdy_det_loss, anchors_px [nl][na][2] and strides [nl] as hdy_decode.  Device outputs:
 * counts [1 + nl] (kept, kept per level), keep_t [nt] int64 target of kept row k (target order), rois [nl][nt][5] (image, x1, y1, x2, y2 of
 * the TRUTH in input pixels, compact per level), order [nt] int64 = position of kept row k in the level-by-level concatenation.
 * workspace: nt * 16 bytes, 8-byte aligned. */
int hdy_mask_select(const float* const* logits, int ldl, const int* ny, const int* nx, int nl, int B, int na, int no, const float* anchors_grid,
                    const float* anchors_px, const float* strides, const float* gts, int nt, float anchor_t, float min_iou, int* counts,
                    long long* keep_t, float* rois, long long* order, void* workspace, size_t ws_bytes, void* stream);
/* gts / tcls of hdy_det_loss from the batch's annotations (replaces the xyxy -> xywh conversion and the one-hot encoding of
 * Detect.forward's target preparation, metayolo/models/yolo_head.py:217-222: ~15 eager tensor ops per step): boxes device [nt][4] corner boxes
 * (normalised, already clamped to [0, 1]), img device [nt] image index of each row (fp32), labels device [nt] int64 class labels
 * 1..nc (anything else: no class) -> gts [nt][5], tcls [nt][nc]. */
int hdy_det_targets(const float* boxes, const float* img, const long long* labels, int nt, int nc, float* gts, float* tcls, void* stream);
int hdy_scale_inplace(void* p, long long n, const float* scale_dev, int dtype, void* stream);

/* ---- Semantic-segmentation branch (hnet's PanopticSeg) ---------------------------------------------
 * Replaces, under autograd, torch.nn.GroupNorm(32, C) + ReLU and nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True)
 * of PanopticFeatureConnector (hnet/segmentation/utils_seg.py:21-36), the branch sum (:58), nn.Upsample / F.interpolate(...,
 * align_corners=True) and torch.nn.Softmax2d of PanopticSeg (hnet/segmentation/panoptic_seg.py:13-19,38-39) and its soft-dice
 * criterion (:22,40; `SoftDiceLoss` is not defined anywhere upstream: restated from the repository's own dice, mask_iou(factor=0),
 * metayolo/models/utils_general.py:268-280).  NHWC, pixel pitches, caller-owned workspaces as everywhere else. */
size_t hdy_groupnorm_workspace_floats(int N, int C);
int hdy_groupnorm_fwd(const void* x, int ldx, const float* gamma, const float* beta, void* y, int ldy, float* stat, float* ab, int N, int HW,
                      int C, int G, float eps, int relu, int dtype, float* workspace, size_t ws_floats, void* stream);
int hdy_groupnorm_bwd(const void* dout, int lddo, const void* x, int ldx, const float* gamma, const float* stat, const float* ab, void* dx, int lddx,
                      float* dgamma, float* dbeta, int accumulate, float* coef, int N, int HW, int C, int G, int relu, int dtype, float* workspace,
                      size_t ws_floats, void* stream);
int hdy_bilinear_fwd(const void* x, int ldx, void* y, int ldy, int N, int Hi, int Wi, int Ho, int Wo, int C, int accumulate, int dtype, void* stream);
int hdy_bilinear_bwd(const void* dy, int lddy, void* dx, int lddx, int N, int Hi, int Wi, int Ho, int Wo, int C, int accumulate, int dtype, void* stream);
/* one axis of the same (the resize is separable): tensors [outer][Ao -> Ai][inner][ld]; W pass outer = N*Ho, inner = 1, then H pass outer = N,
 * inner = Wi over the W pass's result — 8x fewer candidate reads than the 2-D gather at the x8 resize of the segmentation logits */
int hdy_bilinear_bwd_axis(const void* dy, int lddy, void* dx, int lddx, long long outer, int Ai, int Ao, int inner, int C, int accumulate, int dtype,
                          void* stream);
size_t hdy_softdice_workspace_floats(int N, int nc);
int hdy_softdice(const float* logits, int ldl, const float* targets, const float* class_weight, int N, int HW, int nc, float* loss,
                 const float* upstream, float* dlogits, int lddl, float* workspace, size_t ws_floats, void* stream);
/* hdy_softdice's loss, and its gradient already reduced along W by the transposed resize (the W pass of hdy_bilinear_bwd_axis) without the
 * full-resolution gradient tensor: logits fp32 [N][H][W][4] with nc <= 4 classes (the segmentation header's resized class logits,
 * hnet/segmentation/panoptic_seg.py:37-40), dw [N][H][Wi][4]; the caller finishes with the H pass.  Bit-identical to the two-call path.
 * workspace: ws_floats >= hdy_softdice_workspace_floats(N, nc) (every workspace of this section: HDY_EINVAL when smaller). */
int hdy_softdice_wgrad(const float* logits, const float* targets, const float* class_weight, int N, int H, int W, int nc, int Wi, float* loss, float* dw,
                       float* workspace, size_t ws_floats, void* stream);
int hdy_softmax2d(const float* logits, int ldl, float* probs, int ldp, long long M, int nc, void* stream);

/* The stem's weight gradient (layers.py:31 Conv(3, c, 6, 2, 2), backward of train.py:472) with the BatchNorm / SiLU backward of its unit applied
 * while the gradient tile is staged: dz = gradient of the unit's activation output, y = its raw conv output, c1 / c2 from
 * hdy_bn_act_bwd(dy = NULL).  The stem has no data gradient, so its dy is never written.  x = the hdy_stem_prep buffer; bf16; K in
 * {16, 32, 64}; workspace as hdy_conv_wgrad_workspace_bytes(..., stem = 1). */
int hdy_conv_wgrad_stem_fused_ok(int N, int H, int W, int K);
int hdy_conv_wgrad_stem_fused(const void* x, const void* dz, int lddz, const void* y, int ldy, const float* scale, const float* shift, const float* mean,
                              const float* invstd, const float* c1, const float* c2, int N, int H, int W, int K, float* grad_a, int K_a, float* grad_b,
                              int K_b, int accumulate, void* workspace, size_t ws_bytes, void* stream);

/* ---- SyncBatchNorm (reference: train.py:281-283, torch.nn.SyncBatchNorm.convert_sync_batchnorm when --sync-bn).  sums = 2*K + 1 doubles:
 * [SUM x | SUM x^2 | element count].  Forward: hdy_bn_slab_sums over the conv's statistic slabs -> all-reduce(sums) by the caller ->
 * hdy_bn_finalize_sums (same outputs and running-statistic update as hdy_bn_finalize[_pair]; sums may point at a channel slice of a
 * wider [2][sums_ld] block, count at its 2*sums_ld-th double; Ka == K: one module).  Backward: hdy_bn_act_bwd[_pair] with dy == NULL (local statistics, local dgamma / dbeta) -> hdy_bn_slab_sums over its
 * workspace partials (hdy_bn_bwd_blocks(M) slabs of [2][K]) -> all-reduce -> hdy_bn_bwd_coeffs_sums -> hdy_bn_act_bwd_apply. */
int hdy_bn_slab_sums(const float* slabs, int slab_ld, int nslabs, int K, long long count, double* sums, void* stream);
int hdy_bn_finalize_sums(const double* sums, int sums_ld, const double* count, int K, int Ka, const float* gamma_a, const float* beta_a, float* running_mean_a, float* running_var_a,
                         const float* gamma_b, const float* beta_b, float* running_mean_b, float* running_var_b, float eps, float momentum,
                         float* scale, float* shift, float* save_mean, float* save_invstd, void* stream);
int hdy_bn_bwd_coeffs_sums(const double* sums, int K, float* c1, float* c2, void* stream);

/* ---- optimizer step of the training loop (reference: train.py:208-233 torch.optim.SGD(momentum, nesterov=True) in three parameter
 * groups, stepped at train.py:478).  One launch for all tensors: a device table of descriptors (fp32 parameter, gradient, momentum
 * buffer or NULL, element count, parameter group, first = the buffer is uninitialised: buf = g'), first_block = running sum of
 * hdy_sgd_blocks(n).  g' = g + wd*p; buf = momentum*buf + (1-dampening)*g'; p -= lr * (nesterov ? g' + momentum*buf : buf).
 * lr / momentum / dampening / weight_decay: HOST arrays of ngroups (<= HDY_SGD_MAX_GROUPS) values, passed by value to the kernel. */
#define HDY_SGD_MAX_GROUPS 8
typedef struct hdy_sgd_desc {
    float* p;
    const float* g;
    float* buf;
    long long n;
    int group, first, first_block, pad_;
} hdy_sgd_desc;
int hdy_sgd_blocks(long long n);
int hdy_sgd_step(const hdy_sgd_desc* table_device, int ndesc, int total_blocks, const float* lr, const float* momentum, const float* dampening,
                 const float* weight_decay, int ngroups, int nesterov, void* stream);

/* ---- launch-list executor (csrc/exec.hip): one call issues a whole precomputed list of launches on two streams — a plan's forward or backward
 * list, whose pointers, shapes and order are fixed (no reference counterpart: the order is the one PyTorch's autograd engine gives the same work,
 * train.py:472).  program = 64-bit words, per item [op][nargs][arg 0]..[arg nargs-1]:
 *   op = hdy_exec_op("hdy_...") (>= 0; -1: that entry point cannot be listed): the entry point's parameters in order WITHOUT the trailing
 *        stream, each widened to 64 bits (pointers / integers by value, float / double by bit pattern);
 *   op = HDY_EXEC_FORK, args {token, words}: the next `words` words run on side_stream once everything issued so far on main_stream is done;
 *   op = HDY_EXEC_JOIN, args {token}: main_stream waits for the launches of that fork (a token never forked: no wait).
 * Tokens are < 65536; their events live per device for the life of the process.  Returns the first non-zero status of a listed entry point
 * (hdy_last_error names it), HDY_EINVAL for a malformed program.  hdy_exec_join: the join alone, for a caller that runs the rest of a
 * list itself.  hdy_copy_f32: dst[0..n) = src[0..n) on the stream (a list item in place of a host-side tensor copy). */
#define HDY_EXEC_FORK 0xF0F0F0F0ull
#define HDY_EXEC_JOIN 0xF0F0F0F1ull
int hdy_exec_op(const char* name);
int hdy_exec_run(const unsigned long long* program, size_t nwords, void* main_stream, void* side_stream);
int hdy_exec_join(unsigned long long token, void* main_stream);
int hdy_copy_f32(const float* src, float* dst, long long n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* HDYOLO_H */

// Internal (not part of the C ABI): argument blocks shared between api.hip and the kernel files.
#pragma once
#include "common.h"

// BatchNorm-backward statistics requested from the kernel that produces a gradient tensor (its LAST contribution): for the output
// channels [c0, c1) of the producer, which are the gradient dz of a Conv+BN+act unit's output, the epilogue also reads that unit's raw
// conv output y (same pixels; channel c0 of the producer = channel 0 of y/scale/...) and accumulates SUM du and SUM du*xhat per channel
// (du = dz * act'(y*scale + shift)) and SUM du*y into one fp32 slab [2][c1 - c0] per workgroup: slabs[wg][2][c1 - c0]; the finalize
// launch turns SUM du*y into SUM du*xhat = invstd * (SUM du*y - mean * SUM du), so the epilogue needs two coefficient vectors, not four.
struct StatReq {
    const void* y; int ldy;
    const float *scale, *shift;
    float* slabs;
    int c0, c1, act;
    int nslabs;           // slabs the caller's array holds (hdy_stat_req.nslabs)
};

struct ConvArgs {
    const void* x;        // [N][Hin][Win][ldx]
    const void* w;        // packed [Kpad][Kdp]
    void* y;              // [N][Hout][Wout][ldy]
    const float* scale;   // per-channel epilogue scale (or null = 1)
    const float* shift;   // per-channel epilogue shift / bias (or null = 0)
    float* stats;         // [stat_cap][2][K] BatchNorm partial sums (or null)
    int stat_cap;         // slabs the caller's `stats` array holds (hdy_conv_fwd's stat_slabs): every launcher that writes statistics checks it
                          // against the slab count of the kernel and grid it is about to start (HDY_STAT_CAP) — the sizing query and the launch
                          // both read the process-wide option table, which another thread may change in between
    const void* res;      // residual added after the activation, same pixel grid as y, pitch ldr (or null)
    int ldr;
    int N, Hin, Win, C, ldx;
    int Ho, Wo, K, ldy;
    int Hout, Wout, oh_mul, oh_off, ow_mul, ow_off;
    int ih_mul, iw_mul, dh0, dw0, TH, TW;
    int Kd, Kdp, M;
    int act, accumulate, dense_out;
    int mtiles, ntiles, bn;
    int span_pixels;      // 1: the C-wide read deliberately spans several ldx-pitched pixels (stem)
    int pointwise;        // derived: 1x1 stride-1 unpadded (input pixel == output pixel)
    int vec_out;          // derived: bf16 output rows can be written with 16-byte stores
    // Stride-2 dgrad as ONE launch: ncls = 4 parity classes of output pixels walked back to back per spatial tile (class = tile & 3).
    // Per class: tap window (c_TH x c_TW taps starting at dy offset c_dh / c_dw), k-blocks, output offsets, packed-weight offset (elements).
    // The scalar fields above (dh0, dw0, TH, TW, Kdp, oh_off, ow_off, w) hold class 0.  ncls <= 1: an ordinary launch.
    int nstat;            // 0..2 statistics requests served by the epilogue (dgrad, bf16 vector epilogue, single column tile)
    StatReq stat[2];
    int ncls;
    int c_dh[4], c_dw[4], c_TH[4], c_TW[4], c_nkb[4], c_oh[4], c_ow[4];
    long long c_w[4];
    // derived by hdy_conv_igemm_launch for the loader: union tap window over the classes (origin uh0 / uw0, UH x UW taps <= 31),
    // utap = every 128-byte k-block lies inside one tap (C % BKE == 0), reciprocals (hdy_magic) of Ho*Wo, Wo, C and the tap-window width
    int uh0, uw0, UH, UW, utap;
    int tile_interleave;  // 1: tiles that share A rows (column tiles of one m-tile, parity classes) go to neighbouring workgroups instead of one
    unsigned mg_howo, mg_wo, mg_c, mg_tw[4];
    int sh_howo, sh_wo, sh_c, sh_tw[4];
    int dbg;              // conv_deep.hip timing ablations (HDY_DEEP_DEBUG; results are wrong when set): 1 no A loads, 2 no B loads, 4 no MFMAs, 8 no epilogue, 16 no fragment reads
};

// `writes` = slabs the launch about to start writes; inside a *_try function (error through *rc, return 1 = handled)
#define HDY_STAT_CAP(a, writes, who)                                                                                                      \
    if ((a).stats && (a).stat_cap != (writes)) {                                                                                          \
        hdy_set_error("%s: the statistics array holds %d slabs, this launch writes %d (a kernel-selection option changed between "        \
                      "hdy_conv_stat_slabs and the launch?)", who, (a).stat_cap, (int)(writes));                                          \
        *rc = HDY_EINVAL;                                                                                                                 \
        return 1;                                                                                                                         \
    }

// reciprocal for n / d, n < 2^31: q = mulhi(2n, *mg) >> *sh (conv_igemm.hip fdiv)
inline void hdy_magic(unsigned d, unsigned* mg, int* sh) {
    int s = 0;
    while ((1ull << s) < d) ++s;
    *mg = (unsigned)((((unsigned long long)1 << (31 + s)) + d - 1) / d);
    *sh = s;
}

struct WgradArgs {
    const void* x;        // [N][Hin][Win][ldx]
    const void* dy;       // [N][Ho][Wo][lddy]
    float* partial;       // [splits][K][Q]   Q = TH*TW*C
    int N, Hin, Win, C, ldx;
    int Ho, Wo, K, lddy;
    int ih_mul, iw_mul, dh0, dw0, TH, TW;
    int Q, P;             // Q = taps*C, P = N*Ho*Wo pixels
    int splits, pix_per_split;
    int ktiles, qtiles;
    int span_pixels;
    // stem kernel only: y != NULL = `dy` holds dz (gradient of the unit's activation output) and the kernel applies the BatchNorm / SiLU
    // backward itself while staging the tile: dy = scale * (dz * silu'(y*scale + shift) - c1 - (y - mean)*invstd * c2)
    const void* y; int ldy;
    const float *bn_scale, *bn_shift, *bn_mean, *bn_invstd, *bn_c1, *bn_c2;
};

int hdy_conv_bn_tile(int K);
int hdy_wgrad_stem_grid(int N, int Ho, int Wo, int K, int dtype);
int hdy_wgrad_stem_launch(const WgradArgs& a, int grid, hipStream_t st);
int hdy_conv_igemm_slabs(long long M, int K, int taps);
int hdy_conv3x3_c64_try(const ConvArgs& a, int dtype, int out_f32, hipStream_t st, int* rc);
int hdy_conv3x3_c64_slabs(int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dtype);
int hdy_conv3x3_c128_try(const ConvArgs& a, int dtype, int out_f32, hipStream_t st, int* rc);
int hdy_conv3x3_c128_slabs(int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dtype);
int hdy_conv_stem_try(const ConvArgs& a, int dtype, int out_f32, hipStream_t st, int* rc);
int hdy_conv_stem_slabs(int N, int H, int W, int K, int dtype);
int hdy_conv_igemm_launch(ConvArgs a, int dtype, int out_f32, hipStream_t st);
int hdy_wgrad_launch(WgradArgs a, int dtype, hipStream_t st);
int hdy_wgrad_plan(int K, int Q, long long P, int dtype, int* splits, int* pix_per_split);
size_t hdy_wgrad3x3_workspace_bytes(int N, int Ho, int Wo, int C, int K, int stride, int dtype);
int hdy_wgrad3x3_try(const void* x, int ldx, const void* dy, int lddy, int N, int Hin, int Win, int Ho, int Wo, int C, int K, int stride, float* partial,
                     int dtype, hipStream_t st, int* splits, int* rc);
size_t hdy_wgrad_deep_workspace_bytes(int N, int Hin, int Win, int Ho, int Wo, int C, int K, int R, int S, int stride, int dtype);
int hdy_wgrad_deep_try(const void* x, int ldx, const void* dy, int lddy, int N, int Hin, int Win, int Ho, int Wo, int C, int K, int R, int S, int stride,
                       int pad, float* partial, int dtype, hipStream_t st, int* splits, int* rc);
int hdy_conv_igemm_stat_grid(long long M, int K, int taps, int ncls);
int hdy_dgrad3x3s2_try(const ConvArgs& a, int dtype, hipStream_t st, int* rc);
int hdy_conv3x3s2_c32_try(const ConvArgs& a, int dtype, int out_f32, hipStream_t st, int* rc);
int hdy_conv_deep_try(const ConvArgs& a, int dtype, int out_f32, hipStream_t st, int* rc);
int hdy_conv_deep_slabs(long long M, int C, int K, int taps, int pointwise, int dtype);
int hdy_conv3x3s2_c32_slabs(int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dtype);

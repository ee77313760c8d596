// hd_yolo_amd — shared device/host helpers for the gfx950 (MI355X, CDNA4) kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include <mutex>

#define HDY_OK 0
#define HDY_EINVAL (-1)
#define HDY_EUNSUPPORTED (-2)

enum { HDY_F32 = 0, HDY_BF16 = 1 };

// thread-local error text, returned by hdy_last_error()
void hdy_set_error(const char* fmt, ...);

// Process-wide switches (tests, A/B measurements): kernel selection and launch geometry knobs.  Each has an environment variable that
// sets its initial value (read once, under std::call_once) and can be changed at run time through hdy_set_option(); values are atomics.
// Switches also steer the sizing queries (statistic slabs, workspaces), so change them BEFORE building a plan, never under a live one.
enum HdyOption {
    HDY_OPT_NO_CLASS_WALK,     // HDY_NO_CLASS_WALK: stride-2 dgrad as four launches instead of one class-walking launch
    HDY_OPT_NO_CONV3X3,        // HDY_NO_CONV3X3: filter-resident 3x3 kernel off (generic implicit GEMM instead)
    HDY_OPT_C3_GRID,           // HDY_C3_GRID: workgroups of the filter-resident 3x3 kernel (0 = default)
    HDY_OPT_NO_CONV3X3S2,      // HDY_NO_CONV3X3S2: patch-resident 3x3 / stride-2 forward off (1: both forms, 2: the 64 -> 128 form only)
    HDY_OPT_NO_DGRAD_S2,       // HDY_NO_DGRAD_S2: patch-resident stride-2 data gradient off (1: both forms, 2: the 64 <- 128 form only)
    HDY_OPT_TILE_INTERLEAVE,   // HDY_TILE_INTERLEAVE: tile order of the generic kernel (bit 0: column tiles, bit 1: parity classes)
    HDY_OPT_NO_BIG_TILES,      // HDY_NO_BIG_TILES: never the 256-row tile of the generic kernel
    HDY_OPT_NO_STEM_KERNEL,    // HDY_NO_STEM_KERNEL: patch-resident stem forward off
    HDY_OPT_WGRAD_BLOCKS,      // HDY_WGRAD_BLOCKS: workgroups of the generic weight-gradient kernel (default 512)
    HDY_OPT_NO_STEM_WGRAD,     // HDY_NO_STEM_WGRAD: patch-resident stem weight gradient off
    HDY_OPT_NO_WGRAD3X3,       // HDY_NO_WGRAD3X3: patch-resident 3x3 weight gradient off
    HDY_OPT_LOSS_GRID,         // HDY_LOSS_GRID: workgroups of the detection loss' dense pass (default 2048)
    HDY_OPT_NO_DEEP,           // HDY_NO_DEEP: deep-pipelined 256-row implicit GEMM off (generic kernel instead)
    HDY_OPT_NO_WGRAD_S2,       // HDY_NO_WGRAD_S2: tap-walking stride-2 3x3 weight gradient off
    HDY_OPT_NO_WGRAD_DEEP,     // HDY_NO_WGRAD_DEEP: deep-pipelined 256 x 256 weight gradient off (generic weight gradient instead)
    HDY_OPT_DEEP_BN,           // HDY_DEEP_BN: column tile of the deep-pipelined kernel (0 = by shape, 128, 256)
    HDY_OPT_DEEP_DEBUG,        // HDY_DEEP_DEBUG: timing ablations of the deep-pipelined kernel (bit mask, results wrong; measurement only)
    HDY_OPT_DEEP_ALL,          // HDY_DEEP_ALL: 1 (default) the deep-pipelined kernel takes multi-tap (3x3) layers too, 0 the 1x1 layers only
    HDY_OPT_DEEP_MIN_TILES,    // HDY_DEEP_MIN_TILES: fewest 256-row tiles the deep-pipelined kernel takes a layer with (default 160)
    HDY_OPT_DEEP_WALK,         // HDY_DEEP_WALK: stride-2 data gradients (four-class walk) on the deep pipeline: 0 never, 1 always, 2 (default) with >= 256 output channels
    HDY_OPT_NO_BN_REDUCE4,     // HDY_NO_BN_REDUCE4: BatchNorm-backward statistics pass with 8 channels per lane (the generic form, 120 VGPRs) instead of 4 (72)
    HDY_OPT_WGRAD_TILE,        // HDY_WGRAD_TILE: dW tile of the generic weight-gradient kernel per workgroup: 0 (default) 128 x 128 where K and T*C allow, 64 = 64 x 64 (four times the pixel range per workgroup at the same grid: a quarter of the split slabs, twice the operand fetches)
    HDY_OPT_SPPF_NO_KEYS,      // HDY_SPPF_NO_KEYS: SPPF forward with the float-compare kernels instead of the order-preserving 16-bit keys (A/B)
    HDY_OPT_WGRAD_DEEP_KMIN,   // HDY_WGRAD_DEEP_KMIN: fewest output channels (a multiple of 64) the deep-pipelined multi-tap weight gradient takes (default 192; 256 = rounds 3-5)
    HDY_OPT_NO_CONV3X3_C128,   // HDY_NO_CONV3X3_C128: filter-resident 3x3 kernel for 128 input channels off (deep-pipelined / generic kernel instead)
    HDY_OPT_NO_F1X1_96,        // HDY_NO_F1X1_96: fused 1x1 backward instance for 96 channels off (three launches instead; A/B)
    HDY_OPT_COUNT
};
int hdy_opt(int id);

// thread-local dispatch log: every host-side launcher names the kernel family it picked (hdy_last_dispatch / hdy_dispatch_log)
void hdy_note_dispatch(const char* what);

#define HDY_ARG(cond, ...)                 \
    do {                                   \
        if (!(cond)) {                     \
            hdy_set_error(__VA_ARGS__);    \
            return HDY_EINVAL;             \
        }                                  \
    } while (0)

#define HDY_LAUNCH_CHECK(what)                                                       \
    do {                                                                             \
        hipError_t e__ = hipGetLastError();                                          \
        if (e__ != hipSuccess) {                                                     \
            hdy_set_error("%s: launch failed: %s", what, hipGetErrorString(e__));    \
            return (int)e__;                                                         \
        }                                                                            \
    } while (0)

typedef __bf16 bf16_t;
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

// One-time per-kernel setup that is bound to a DEVICE (hipFuncSetAttribute: dynamic LDS above 64 KB): once per process and device, not once per
// process — a second device in the same process (one plan per device; the executor keeps per-device events for that case) would otherwise launch
// without the attribute.
// Sum over the 16 lanes of a DPP row (lanes 16k .. 16k + 15), delivered to every lane of the row: four v_add_f32 with row_ror operands, no LDS
// crossbar.  The BatchNorm sums of the conv epilogues end in this reduction (pixel lanes of an MFMA accumulator tile); as `u += __shfl_xor(u, m)`
// hipcc emitted ds_bpermute_b32 + s_waitcnt lgkmcnt(0) per step and value, one after the other: 64 LDS round trips = 3.2 us for the 32 sums at the
// end of every conv_deep forward launch (profiles/r05_small_probes_ab.txt).
#if defined(__HIPCC__)
__device__ __forceinline__ float row_sum16(float v) {
#ifdef HDY_ROW_SUM_SHFL      // A/B build: the former shuffle chain
    for (int m = 1; m < 16; m <<= 1) v += __shfl_xor(v, m);
    return v;
#endif
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xF, 0xF, true));   // row_ror:8
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xF, 0xF, true));   // row_ror:4
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xF, 0xF, true));   // row_ror:2
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xF, 0xF, true));   // row_ror:1
    return v;
}
#endif

struct PerDeviceOnce {
    std::once_flag flag[16];
    template <typename F> void run(F&& fn) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        std::call_once(flag[dev & 15], fn);
    }
};

static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }
static inline int round_up(int a, int b) { return (a + b - 1) / b * b; }

#ifdef __HIPCC__
// 16-byte vector <-> 8 bf16 / 4 f32 views
union V16 {
    i32x4 i;
    f32x4 f;
    bf16x8 h;
};

__device__ __forceinline__ float silu_f(float u) { return u / (1.0f + __expf(-u)); }
__device__ __forceinline__ float sigmoid_f(float u) { return 1.0f / (1.0f + __expf(-u)); }

template <typename T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f32<bf16_t>(bf16_t v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return (bf16_t)v; }

// Bijective XCD-aware remap of a 1-D grid (cdna_hip_programming.md §5.5 T1): blocks that the
// dispatcher deals to one XCD (same id % 8) get a contiguous range of logical tile ids, so
// neighbouring tiles share that XCD's L2.  Speed only; any mapping is correct.
__device__ __forceinline__ int xcd_remap(int id, int n) {
    const int q = n >> 3, r = n & 7, x = id & 7, s = id >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + s;
}
#endif

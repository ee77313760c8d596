// Launch-list executor: one call runs a whole precomputed list of launches (a plan's forward or backward list) on two HIP streams.
//
// Why: a plan's lists are static — fixed pointers, shapes and order, ~150 launches forward and ~300 backward for yolov5s — and the host side of
// a step was a Python loop of ctypes calls (5-8 us each, more for a fork: two event objects, a record, a wait).  Where the GPU's kernels are
// short (the detection heads at the start of the backward list, the 20x20 layers) the GPU waited for the host: ~130 us of idle gaps per step in
// the kernel trace, and a box with a slow or noisy host lost more.  hipGraph replay of the same lists measured slower than eager launches on
// ROCm 7.2, so the list stays a list of ordinary launches — issued from C.
//
// Program = 64-bit words: [op][nargs][arg 0] ... [arg nargs-1] per item.
//   op < HDY_EXEC_FORK   index into the table below (hdy_exec_op(name)); the args are the entry point's parameters in order, WITHOUT the trailing
//                        stream, each widened to 64 bits (pointers and integers by value, float / double by bit pattern);
//   HDY_EXEC_FORK        args = {token, words}: the next `words` words run on the side stream once everything issued so far on the main stream is
//                        done (event token: recorded on main, awaited by side), and leave event `done[token]` behind them;
//   HDY_EXEC_JOIN        args = {token}: the main stream waits for done[token].
// The semantics are those of hd_yolo_amd/ops.py's SideStream.fork / join, whose Python form remains for lists with host callbacks between
// launches.  Reference: the order PyTorch's autograd engine gives the same work on its streams (train.py:472); nothing in the reference
// corresponds to the list itself.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <tuple>
#include <type_traits>
#include <utility>
#include <vector>

#include "common.h"
#include "hdyolo.h"

namespace {

typedef unsigned long long u64;

#define HDY_HIP(call)                                                                  \
    do {                                                                               \
        hipError_t e__ = (call);                                                       \
        if (e__ != hipSuccess) {                                                       \
            hdy_set_error("exec: %s failed: %s", #call, hipGetErrorString(e__));       \
            return (int)e__;                                                           \
        }                                                                              \
    } while (0)

template <typename T>
T arg_from(u64 v) {
    if constexpr (std::is_pointer_v<T>) {
        return reinterpret_cast<T>((uintptr_t)v);
    } else if constexpr (std::is_same_v<T, float>) {
        const unsigned u = (unsigned)v;
        float f;
        __builtin_memcpy(&f, &u, 4);
        return f;
    } else if constexpr (std::is_same_v<T, double>) {
        double d;
        __builtin_memcpy(&d, &v, 8);
        return d;
    } else {
        static_assert(std::is_integral_v<T>, "entry point parameter that is neither pointer, integer nor floating point");
        return static_cast<T>(v);
    }
}

template <typename... P, size_t... I>
int call_impl(int (*f)(P...), const u64* a, void* stream, std::index_sequence<I...>) {
    using Tup = std::tuple<P...>;
    return f(arg_from<std::tuple_element_t<I, Tup>>(a[I])..., stream);
}

// every listed entry point is `int f(..., void* stream)`
template <typename... P>
int invoke(int (*f)(P...), const u64* a, int nargs, void* stream) {
    static_assert(sizeof...(P) >= 1, "no stream parameter");
    static_assert(std::is_same_v<std::tuple_element_t<sizeof...(P) - 1, std::tuple<P...>>, void*>, "the last parameter must be the stream");
    if (nargs != (int)sizeof...(P) - 1) return -1000 - (int)sizeof...(P);
    return call_impl(f, a, stream, std::make_index_sequence<sizeof...(P) - 1>{});
}

struct Entry {
    const char* name;
    int (*run)(const u64*, int, void*);
};
#define E(fn) {#fn, [](const u64* a, int n, void* s) { return invoke(&fn, a, n, s); }}
const Entry TABLE[] = {
    E(hdy_conv_pack), E(hdy_conv_pack_run), E(hdy_conv_fwd), E(hdy_conv_dgrad), E(hdy_conv_wgrad), E(hdy_conv_dgrad_stats),
    E(hdy_conv1x1_bwd_fused_stats), E(hdy_bn_bwd_finalize_slabs), E(hdy_bn_act_bwd_apply), E(hdy_conv1x1_bwd_fused), E(hdy_bn_finalize),
    E(hdy_bn_eval_coeffs), E(hdy_bn_eval_coeffs_batch), E(hdy_bn_act_fwd), E(hdy_bn_act_bwd), E(hdy_bn_finalize_pair), E(hdy_bn_act_fwd_pair),
    E(hdy_bn_act_bwd_pair), E(hdy_add_inplace), E(hdy_colsum), E(hdy_sppf_pool_fwd), E(hdy_sppf_pool_bwd), E(hdy_upsample2x_fwd),
    E(hdy_upsample2x_bwd), E(hdy_stem_prep), E(hdy_nchw_to_nhwc), E(hdy_decode), E(hdy_det_grad_pack), E(hdy_nms_batched), E(hdy_det_outputs),
    E(hdy_nms_boxes), E(hdy_roi_align_fwd), E(hdy_roi_align_bwd), E(hdy_relu_bwd), E(hdy_cast_store), E(hdy_det_loss), E(hdy_mask_select),
    E(hdy_det_targets), E(hdy_scale_inplace), E(hdy_groupnorm_fwd), E(hdy_groupnorm_bwd), E(hdy_bilinear_fwd), E(hdy_bilinear_bwd),
    E(hdy_bilinear_bwd_axis), E(hdy_softdice), E(hdy_softdice_wgrad), E(hdy_softmax2d), E(hdy_conv_wgrad_stem_fused), E(hdy_bn_slab_sums),
    E(hdy_bn_finalize_sums), E(hdy_bn_bwd_coeffs_sums), E(hdy_sgd_step), E(hdy_copy_f32),
};
#undef E
constexpr int NTABLE = (int)(sizeof(TABLE) / sizeof(TABLE[0]));

// events of the fork tokens: [token] -> {recorded on main at the fork, recorded on side behind the fork's launches}; per device, created on demand
struct TokenEvents {
    hipEvent_t start = nullptr, done = nullptr;
};
std::mutex g_ev_mutex;
std::vector<std::vector<TokenEvents>> g_events;       // [device][token]

int events_for(u64 token, TokenEvents* out) {
    if (token >= 65536) return HDY_EINVAL;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return HDY_EINVAL;
    std::lock_guard<std::mutex> lock(g_ev_mutex);
    if ((int)g_events.size() <= dev) g_events.resize(dev + 1);
    auto& v = g_events[dev];
    if (v.size() <= token) v.resize(token + 1);
    TokenEvents& t = v[token];
    if (!t.start) {
        // both streams are queues of ONE device: a device-scope release is all a record has to do (the default releases to the system — the
        // marker packet then costs the main queue ~8 us per fork, 34 forks per yolov5s step: profiles/r04_fork_markers.txt)
        static const unsigned flags = hipEventDisableTiming | (getenv("HDY_EVENT_SYSTEM_SCOPE") ? 0u : (unsigned)hipEventReleaseToDevice);
        if (hipEventCreateWithFlags(&t.start, flags) != hipSuccess || hipEventCreateWithFlags(&t.done, flags) != hipSuccess) return HDY_EINVAL;
    }
    *out = t;
    return HDY_OK;
}

int run_words(const u64* w, size_t n, hipStream_t main, hipStream_t side, hipStream_t on, bool in_fork) {
    size_t i = 0;
    while (i < n) {
        HDY_ARG(i + 2 <= n, "exec: truncated item at word %zu", i);
        const u64 op = w[i], nargs = w[i + 1];
        HDY_ARG(nargs <= 64 && i + 2 + nargs <= n, "exec: item at word %zu has %llu arguments, %zu words left", i, nargs, n - i - 2);
        const u64* a = w + i + 2;
        const size_t item = i;                                   // the item's first word, for messages
        i += 2 + nargs;
        if (op == HDY_EXEC_FORK) {
            // a[1] <= n - i, not i + a[1] <= n: the sum wraps for a huge length word (i <= n holds here)
            HDY_ARG(!in_fork && nargs == 2 && side && a[1] <= (u64)(n - i), "exec: bad fork at word %zu (nested, no side stream, or longer than the program)", item);
            TokenEvents ev;
            HDY_ARG(events_for(a[0], &ev) == HDY_OK, "exec: no events for fork token %llu", a[0]);
            HDY_HIP(hipEventRecord(ev.start, main));
            HDY_HIP(hipStreamWaitEvent(side, ev.start, 0));
            const int rc = run_words(w + i, (size_t)a[1], main, side, side, true);
            if (rc) return rc;
            HDY_HIP(hipEventRecord(ev.done, side));
            i += a[1];
        } else if (op == HDY_EXEC_JOIN) {
            HDY_ARG(!in_fork && nargs == 1, "exec: bad join at word %zu", item);
            TokenEvents ev;
            HDY_ARG(events_for(a[0], &ev) == HDY_OK, "exec: no events for join token %llu", a[0]);
            HDY_HIP(hipStreamWaitEvent(main, ev.done, 0));        // (an event never recorded: no wait)
        } else {
            HDY_ARG(op < (u64)NTABLE, "exec: unknown op %llu at word %zu", op, i);
            const int rc = TABLE[op].run(a, (int)nargs, (void*)on);
            HDY_ARG(rc > -1000, "exec: %s takes %d arguments before the stream, the item has %llu", TABLE[op].name, -rc - 1001, nargs);
            if (rc) return rc;                                    // the entry point has set hdy_last_error
        }
    }
    return HDY_OK;
}

__global__ void copy_f32_kernel(const float* __restrict__ src, float* __restrict__ dst, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

}  // namespace

extern "C" {

int hdy_exec_op(const char* name) {
    if (!name) return -1;
    for (int i = 0; i < NTABLE; ++i)
        if (!strcmp(TABLE[i].name, name)) return i;
    return -1;
}

int hdy_exec_run(const unsigned long long* program, size_t nwords, void* main_stream, void* side_stream) {
    HDY_ARG(program || nwords == 0, "exec: null program");
    return run_words(program, nwords, (hipStream_t)main_stream, (hipStream_t)side_stream, (hipStream_t)main_stream, false);
}

int hdy_exec_join(unsigned long long token, void* main_stream) {
    TokenEvents ev;
    HDY_ARG(events_for(token, &ev) == HDY_OK, "exec: no events for join token %llu", token);
    HDY_HIP(hipStreamWaitEvent((hipStream_t)main_stream, ev.done, 0));
    return HDY_OK;
}

int hdy_copy_f32(const float* src, float* dst, long long n, void* stream) {
    HDY_ARG(src && dst && n > 0, "copy_f32: bad args");
    hipLaunchKernelGGL(copy_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, dst, n);
    HDY_LAUNCH_CHECK("copy_f32");
    return HDY_OK;
}

}  // extern "C"

// MEASURED AND NOT ADOPTED (round 3) — kept as the record of the experiment; not part of the build.
// In-launch finalize of per-workgroup partial-sum slabs ("the last workgroup to arrive reduces"), gfx950.
//
// Result: wired into bn_act_bwd_reduce_kernel (53 launches per yolov5s step lose their finalize launch), all GPU tests green, train step
// 12.33 ms with the separate finalize launches against 12.36 - 12.53 ms with this tail.  The critical path of the last workgroup is
// write-through + s_waitcnt (~1.5 us) -> agent-scope atomic (~1.5 us) -> one round trip of sc1 loads (~2 us), twice for two levels: the
// same ~9 us a kernel boundary plus the 5 us finalize kernel cost, because both are made of the same cross-XCD coherence round trips.  And
// every workgroup of the pass pays the drain + atomic at its end.  The first version used __threadfence(): 18.0 ms per step (see below).
//
// Every statistics pass here leaves one fp32 slab [2][K] per workgroup and used to be followed by a few-microsecond finalize launch that
// sums the slabs in fp64 and derives per-channel coefficients.  102 such launches per yolov5s train step cost 0.95 ms of a 12.4 ms step
// (measured by skipping them): a 5 us kernel on 16-64 workgroups drains the chip before it and refills it after it.  With this helper the
// producing launch finishes the job itself:
//
//   level 1  the slabs are cut into groups of G consecutive slabs; a workgroup that has written its slab drains its stores and counts
//            itself into its group; whoever completes the group sums the group's slabs in fp64, IN SLAB ORDER, into gpart[group][2][K]
//   level 2  ... then counts the group into the launch; whoever completes the launch sums the group partials in group order and calls
//            fin(k, sum0, sum1) for every channel
//
// The summation order is fixed by slab and group index, not by arrival order: results are deterministic (bit-identical from run to run).
// One level costs a fence + atomic + one round trip of independent loads (~1-1.5 us); a single level over 256-1024 slabs of 1 KB would
// pull 0.25-1 MB through one CU (3-10 us).
//
// Coherence without fences: an agent-scope fence (__threadfence) on gfx950 is buffer_wbl2 + buffer_inv of the XCD's whole 4 MB L2 — issued by
// every workgroup of a 400-1024 workgroup pass it cost ~100 us per launch (measured: train step 12.5 -> 18.0 ms).  Instead the slabs and
// group partials are written and read with agent-scope relaxed atomic accesses (global_store / global_load with sc1: write-through to, and
// read from, the point where the eight XCDs' L2s agree), ordered against the counter updates by s_waitcnt vmcnt(0) + the workgroup barrier.
// Producers therefore store their slabs through tail_store().
//
// Counters: ctr[0] counts finished groups, ctr[1 + g] the arrivals of group g; all are zero between launches (the workgroup that completes
// a count resets it).  They come from a library-owned pool (hdy_tail_counters: round robin over 1024 sets, zeroed once) because a caller
// workspace has no defined contents; gpart lives in the caller's workspace.
#pragma once
#include "common.h"

constexpr int TAIL_MAX_GROUPS = 64;

struct TailGeo {
    int nslabs, G, ngroups, per_slab;       // per_slab: workgroups that contribute to one slab row (column tiles of a conv launch)
};

// G: the power of two with G * G >= nslabs (at least 8, so that a group's sum is worth a level), at most 64 groups
static inline TailGeo tail_geo(int nslabs, int per_slab) {
    TailGeo g;
    g.nslabs = nslabs; g.per_slab = per_slab;
    g.G = 8;
    while (g.G * g.G < nslabs || cdiv(nslabs, g.G) > TAIL_MAX_GROUPS) g.G *= 2;
    g.ngroups = cdiv(nslabs, g.G);
    return g;
}
// doubles of group partials the tail of a launch with `nslabs` slabs of K channels needs (0: one group, no second level)
static inline size_t tail_gpart_doubles(int nslabs, int K) {
    const TailGeo g = tail_geo(nslabs, 1);
    return g.ngroups > 1 ? (size_t)g.ngroups * 2 * K : 0;
}

// host: counters for one launch (device memory, 1 + TAIL_MAX_GROUPS zeroed uints); nullptr when the pool could not be allocated
unsigned* hdy_tail_counters();

#ifdef __HIPCC__
__device__ __forceinline__ void tail_store(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float tail_load(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void tail_store(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double tail_load(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// all of this thread's stores have reached their scope (stores count in vmcnt on gfx9)
__device__ __forceinline__ void tail_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// Called by ALL threads of every workgroup of the launch (1-D blocks of NTHR threads) after the workgroup's slab stores (tail_store).
// slabs[(s * 2 + which) * ld + k].  fin(k, sum0, sum1) runs once per channel k < K, in the single workgroup that completed the launch.
template <int NTHR, class Fin>
__device__ __forceinline__ void tail_finalize(const float* slabs, int ld, int K, const TailGeo geo, int my_slab, unsigned* ctr, double* gpart, Fin fin) {
    __shared__ int s_last;
    const int tid = threadIdx.x;
    const int g = my_slab / geo.G;
    const int t0 = g * geo.G, t1 = min(t0 + geo.G, geo.nslabs);
    tail_drain();                                                // this thread's slab stores are out
    __syncthreads();
    if (tid == 0) {
        const unsigned old = __hip_atomic_fetch_add(&ctr[1 + g], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = old + 1 == (unsigned)((t1 - t0) * geo.per_slab);
        if (s_last) __hip_atomic_store(&ctr[1 + g], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (!s_last) return;
    const bool one = geo.ngroups == 1;
    for (int k = tid; k < K; k += NTHR) {
        double s = 0.0, ss = 0.0;
        int t = t0;
        for (; t + 8 <= t1; t += 8) {                            // 16 independent loads in flight per lane
            float a[8], b[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                a[u] = tail_load(slabs + ((size_t)(t + u) * 2 + 0) * ld + k);
                b[u] = tail_load(slabs + ((size_t)(t + u) * 2 + 1) * ld + k);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) { s += (double)a[u]; ss += (double)b[u]; }
        }
        for (; t < t1; ++t) {
            s += (double)tail_load(slabs + ((size_t)t * 2 + 0) * ld + k);
            ss += (double)tail_load(slabs + ((size_t)t * 2 + 1) * ld + k);
        }
        if (one) fin(k, s, ss);
        else {
            tail_store(gpart + ((size_t)g * 2 + 0) * K + k, s);
            tail_store(gpart + ((size_t)g * 2 + 1) * K + k, ss);
        }
    }
    if (one) return;
    tail_drain();
    __syncthreads();
    if (tid == 0) {
        const unsigned old = __hip_atomic_fetch_add(&ctr[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = old + 1 == (unsigned)geo.ngroups;
        if (s_last) __hip_atomic_store(&ctr[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (!s_last) return;
    for (int k = tid; k < K; k += NTHR) {
        double s = 0.0, ss = 0.0;
        int q = 0;
        for (; q + 8 <= geo.ngroups; q += 8) {
            double a[8], b[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                a[u] = tail_load(gpart + ((size_t)(q + u) * 2 + 0) * K + k);
                b[u] = tail_load(gpart + ((size_t)(q + u) * 2 + 1) * K + k);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) { s += a[u]; ss += b[u]; }
        }
        for (; q < geo.ngroups; ++q) {
            s += tail_load(gpart + ((size_t)q * 2 + 0) * K + k);
            ss += tail_load(gpart + ((size_t)q * 2 + 1) * K + k);
        }
        fin(k, s, ss);
    }
}
#endif

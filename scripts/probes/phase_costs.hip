// Probe: what the building blocks of conv_deep.hip's phase cost on one CU-filling launch (256 workgroups x 512 threads, one per CU):
//   (a) s_barrier in an 8-wave workgroup, all waves arriving together / two wave groups staggered by one barrier;
//   (b) one LDS-DMA unit (2 x buffer_load_dwordx4 ... lds per thread = 16 KB per workgroup) from an L2-resident source: issue time per
//       instruction with 0, 4, 8 units already in flight;
//   (c) 12 ds_read_b128 per lane + s_waitcnt lgkmcnt(0) with 8 waves reading at once;
//   (d) 16 v_mfma_f32_16x16x32_bf16 back to back.
// Cycles by s_memtime (lane 0 of wave 0), averaged over the loop.
// Build + run:  hipcc --offload-arch=gfx950 -O3 -o /tmp/phase_costs scripts/probes/phase_costs.hip && /tmp/phase_costs
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
union V16 { i32x4 i; bf16x8 h; };

__device__ __forceinline__ unsigned long long now() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}

__global__ __launch_bounds__(512, 2) void probe(const unsigned char* src, unsigned long long* out, int iters, int mode) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned lds0 = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) unsigned char*)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)(src + (size_t)(blockIdx.x & 31) * 65536), 0, 0x80000000u, 0x00020000);
    unsigned long long t0 = 0, t1 = 0, acc_t = 0;
    V16 a[12];
    f32x4 c[4] = {};
    for (int i = 0; i < 12; ++i) a[i].i = i32x4{tid, i, 1, 2};
    __syncthreads();
    if (mode == 0) {                      // barriers, all together
        t0 = now();
        for (int it = 0; it < iters; ++it) __builtin_amdgcn_s_barrier();
        t1 = now();
    } else if (mode == 1) {               // staggered groups: waves 4-7 one barrier behind; a little work in each segment
        if (wave >= 4) __builtin_amdgcn_s_barrier();
        t0 = now();
        for (int it = 0; it < iters; ++it) {
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_s_barrier();
        }
        t1 = now();
        if (wave < 4) __builtin_amdgcn_s_barrier();
    } else if (mode >= 2 && mode <= 4) {  // DMA issue cost with (mode-2)*4 units kept in flight
        const int keep = (mode - 2) * 4;
        for (int it = 0; it < iters; ++it) {
            const unsigned dst = lds0 + (unsigned)((it & 7) * 16384 + wave * 1024);
            const unsigned long long s = now();
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (void __attribute__((address_space(3)))*)(uintptr_t)dst, 16, tid * 16, (it & 3) * 16384, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (void __attribute__((address_space(3)))*)(uintptr_t)(dst + 8192), 16, tid * 16 + 8192, (it & 3) * 16384, 0, 0);
            const unsigned long long e = now();
            acc_t += e - s;
            if (keep == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (keep == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        t0 = 0; t1 = acc_t;
    } else if (mode == 5) {               // 12 fragment reads + wait, all 8 waves
        const unsigned base = lds0 + (unsigned)(((tid >> 2) & 127) * 128 + (((tid & 3) ^ ((tid >> 3) & 7)) << 4));
        t0 = now();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 12; ++j) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(a[j].i) : "v"(base), "n"(j * 2048) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
        t1 = now();
    } else if (mode == 6) {               // 16 MFMAs per iteration, 8 waves (2 per SIMD)
        t0 = now();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 16; ++j) c[j & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[j % 12].h, a[(j + 5) % 12].h, c[j & 3], 0, 0, 0);
        }
        t1 = now();
    } else if (mode == 7) {               // the full phase skeleton: reads + 1 unit DMA + vmcnt(8) + barrier + lgkm + 16 MFMA + barrier, staggered
        const unsigned base = lds0 + (unsigned)(((tid >> 2) & 127) * 128 + (((tid & 3) ^ ((tid >> 3) & 7)) << 4));
        if (wave >= 4) __builtin_amdgcn_s_barrier();
        t0 = now();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 12; ++j) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(a[j].i) : "v"(base), "n"(j * 2048) : "memory");
            const unsigned dst = lds0 + (unsigned)(32768 + (it & 3) * 16384 + wave * 1024);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (void __attribute__((address_space(3)))*)(uintptr_t)dst, 16, tid * 16, (it & 3) * 16384, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (void __attribute__((address_space(3)))*)(uintptr_t)(dst + 8192), 16, tid * 16 + 8192, (it & 3) * 16384, 0, 0);
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int j = 0; j < 16; ++j) c[j & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[j % 12].h, a[(j + 5) % 12].h, c[j & 3], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        t1 = now();
        if (wave < 4) __builtin_amdgcn_s_barrier();
    }
    float sink = c[0][0] + c[1][1] + c[2][2] + c[3][3];
    for (int i = 0; i < 12; ++i) sink += (float)a[i].i[0];
    if (tid == 0) { out[blockIdx.x * 2] = t1 - t0; out[blockIdx.x * 2 + 1] = (unsigned long long)(sink != 12345.f); }
}

int main() {
    unsigned char* src; unsigned long long* out;
    hipMalloc(&src, 32 * 65536 + 65536); hipMemset(src, 1, 32 * 65536 + 65536);
    hipMalloc(&out, 256 * 16);
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const int iters = 2000;
    const char* names[] = {"s_barrier, 8 waves together", "2 x s_barrier per iteration, groups staggered", "DMA unit (2 instr/thread) issue, drained each time",
                           "DMA unit issue, 4 units kept in flight", "DMA unit issue, 8 units kept in flight", "12 x ds_read_b128 + lgkmcnt(0), 8 waves",
                           "16 MFMA 16x16x32, 2 waves per SIMD", "full phase skeleton (staggered)"};
    for (int mode = 0; mode < 8; ++mode) {
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(probe, dim3(256), dim3(512), 144 * 1024, 0, src, out, iters, mode);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(512);
        hipMemcpy(h.data(), out, 512 * 8, hipMemcpyDeviceToHost);
        double s = 0; for (int b = 0; b < 256; ++b) s += (double)h[b * 2];
        printf("mode %d  %-52s %8.1f cycles per iteration (avg over 256 workgroups)\n", mode, names[mode], s / 256 / iters);
    }
    return 0;
}

// Probe: what does `buffer_load_dwordx4 ... offen lds` leave in LDS for a lane whose offset fails the descriptor's range check?
// (conv_igemm.hip relies on: zeros.)  Also: is the SGPR offset part of the range check?
// Build + run:  hipcc --offload-arch=gfx950 -O2 -o /tmp/buffer_lds_oob scripts/probes/buffer_lds_oob.hip && /tmp/buffer_lds_oob
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const int* x, int* out, unsigned nbytes, int soff) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int* s = (int*)smem;
    for (int i = threadIdx.x; i < 1024; i += 64) s[i] = 0x7777;            // junk
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, nbytes, 0x00020000);
    // lanes 0..31 in range, 32..47 offset 0x80000000 (beyond num_records), 48..63 offset nbytes - 8 (straddles the end)
    unsigned voff = threadIdx.x * 16;
    if (threadIdx.x >= 32) voff = 0x80000000u;
    if (threadIdx.x >= 48) voff = nbytes - 8;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)smem, 16, voff, soff, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += 64) out[i] = s[i];
}
int main() {
    const int n = 4096;
    std::vector<int> h(n);
    for (int i = 0; i < n; ++i) h[i] = i + 1;
    int *dx, *dout;
    hipMalloc(&dx, n * 4); hipMalloc(&dout, 256 * 4 * 2);
    hipMemcpy(dx, h.data(), n * 4, hipMemcpyHostToDevice);
    for (int pass = 0; pass < 2; ++pass) {
        const int soff = pass ? 64 : 0;
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 4096, 0, dx, dout, 2048u, soff);
        std::vector<int> o(256);
        hipMemcpy(o.data(), dout, 256 * 4, hipMemcpyDeviceToHost);
        printf("soffset=%d\n", soff);
        for (int l : {0, 1, 31, 32, 40, 47, 48, 63}) printf("  lane %2d: %d %d %d %d\n", l, o[l * 4], o[l * 4 + 1], o[l * 4 + 2], o[l * 4 + 3]);
    }
    // range check with the SGPR offset: num_records 2048, voffset 2040-ish in range only without soffset
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 4096, 0, dx, dout, 600u, 512);   // lanes 0..31: voff up to 496+16 <= 600 in range by voffset alone; +512 beyond
    std::vector<int> o(256);
    hipMemcpy(o.data(), dout, 256 * 4, hipMemcpyDeviceToHost);
    printf("num_records=600 soffset=512 (is soffset range-checked?)\n");
    for (int l : {0, 5, 6, 20, 31}) printf("  lane %2d: %d %d %d %d\n", l, o[l * 4], o[l * 4 + 1], o[l * 4 + 2], o[l * 4 + 3]);
    return 0;
}

// Per-CU operand-delivery ceilings on gfx950 from an L2-resident source: global_load_dwordx4 -> VGPR versus
// global_load_lds_dwordx4 (LDS-DMA), 8 waves per CU (one 512-thread block per CU), 256 CUs.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/l2_to_cu_bw.hip -o tools/ubench/l2_to_cu_bw
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int KIND, int INFLIGHT> __global__ __launch_bounds__(512) void k(const char* src, float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const char* base = src + (size_t)(blockIdx.x & 63) * 131072;          // 8 MiB total window: L2 / Infinity Cache resident
    u32x4 acc = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        const int o = ((it * INFLIGHT) & 15) * 8192;                        // walk 128 KiB per block
#pragma unroll
        for (int j = 0; j < INFLIGHT; ++j) {
            const char* p = base + ((o + j * 8192 + wave * 1024) & 131071) + lane * 16;
            if (KIND == 0) {
                const u32x4 v = *(const u32x4*)p;
                acc ^= v;
            } else {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                                 (__attribute__((address_space(3))) void*)(lds + ((j & 7) * 8 + wave) * 1024), 16, 0, 0);
            }
        }
        if (KIND == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    out[blockIdx.x * 512 + threadIdx.x] = (float)(acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) + lds[threadIdx.x];
}
template <int KIND, int INFLIGHT> void run(const char* name) {
    char* src; float* d;
    hipMalloc(&src, (size_t)64 * 131072); hipMemset(src, 1, (size_t)64 * 131072); hipMalloc(&d, 256 * 512 * 4);
    hipFuncSetAttribute((const void*)k<KIND, INFLIGHT>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    const int iters = 4096, blocks = 256;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<KIND, INFLIGHT><<<blocks, 512, 65536>>>(src, d, 16);
    hipEventRecord(a);
    k<KIND, INFLIGHT><<<blocks, 512, 65536>>>(src, d, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double bytes_per_cu = (double)iters * INFLIGHT * 8 * 1024;
    printf("%-28s %d x 1 KiB per wave in flight: %7.3f ms  %6.1f GB/s per CU = %5.1f B/clk @1.9GHz, %5.1f TB/s chip\n", name, INFLIGHT, ms,
           bytes_per_cu / ms / 1e6, bytes_per_cu / ms / 1e6 / 1.9, bytes_per_cu * 256 / ms / 1e9);
    hipFree(src); hipFree(d);
}
int main() {
    run<0, 4>("global_load_dwordx4 -> VGPR");
    run<0, 8>("global_load_dwordx4 -> VGPR");
    run<1, 4>("global_load_lds_dwordx4");
    run<1, 8>("global_load_lds_dwordx4");
    return 0;
}

// Per-block scheduling cost on gfx950 for GEMM-like launch shapes: 512 threads, 128 KiB dynamic LDS (one block per CU),
// kernels that (a) exit at once, (b) touch their LDS once, (c) additionally wait for one 64-KiB DMA from global memory.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/block_launch_cost.hip -o tools/ubench/block_launch_cost
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int KIND> __global__ __launch_bounds__(512) void k(const char* src, float* out, int window_mask) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    if (KIND == 0) return;
    if (KIND >= 2) {
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        const char* p = src + (size_t)(blockIdx.x & window_mask) * 65536 + wave * 8192 + lane * 16;
        for (int j = 0; j < 8; ++j)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p + j * 1024),
                                             (__attribute__((address_space(3))) void*)(lds + wave * 8192 + j * 1024), 16, 0, 0);
        __syncthreads();
    }
    lds[threadIdx.x * 4] = 1;
    __syncthreads();
    if (threadIdx.x == 0 && lds[4] == 77) out[blockIdx.x] = 1.f;
}
template <int KIND> void run(const char* name, int lds_bytes, int window_mask = 4095) {
    char* src; float* d;
    hipMalloc(&src, (size_t)4096 * 65536); hipMalloc(&d, 65536 * 4);
    hipFuncSetAttribute((const void*)k<KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    const int blocks = 28672;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<KIND><<<blocks, 512, lds_bytes>>>(src, d, window_mask);
    hipEventRecord(a);
    for (int i = 0; i < 5; ++i) k<KIND><<<blocks, 512, lds_bytes>>>(src, d, window_mask);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
    printf("%-52s LDS %3d KiB: %7.3f ms for %d blocks = %5.2f us per block per CU\n", name, lds_bytes / 1024, ms, blocks, ms * 1e3 / (blocks / 256.0));
    hipFree(src); hipFree(d);
}
int main() {
    run<0>("exit at once", 128 * 1024);
    run<0>("exit at once", 16 * 1024);
    run<1>("touch LDS + barrier", 128 * 1024);
    run<2>("one 64 KiB DMA + barrier, 256 MiB window (HBM)", 128 * 1024, 4095);
    run<2>("one 64 KiB DMA + barrier, 16 MiB window (MALL)", 128 * 1024, 255);
    run<2>("one 64 KiB DMA + barrier, 2 MiB window (L2)", 128 * 1024, 31);
    return 0;
}

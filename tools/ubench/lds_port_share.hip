// How many LDS port cycles does one KiB of LDS-DMA (global_load_lds_dwordx4) cost?  4 waves of a block stream ds_read_b128
// flat out (LDS-read bound), the other 4 waves issue LDS-DMA at a controlled rate; the drop in read throughput per DMA byte
// gives the port cost of the DMA writes.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/lds_port_share.hip -o tools/ubench/lds_port_share
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int DMA_PER_ITER> __global__ __launch_bounds__(512) void k(const char* src, float* out, int iters, unsigned long long* clk) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 32768; i += 512) ((unsigned*)lds)[i] = i;
    __syncthreads();
    u32x4 acc = {0, 0, 0, 0};
    const long long t0 = clock64();
    if (wave < 4) {                                                          // readers: 16 x ds_read_b128 per iteration
        const char* p = lds + wave * 16384 + lane * 16;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 16; ++j) acc ^= *(const u32x4*)(p + j * 1024);
        }
    } else if (DMA_PER_ITER > 0) {                                           // DMA writers into the upper 64 KiB
        const char* base = src + (size_t)(blockIdx.x & 63) * 131072 + (wave - 4) * 32768 + lane * 16;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < DMA_PER_ITER; ++j)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + ((it * DMA_PER_ITER + j) & 31) * 1024),
                                                 (__attribute__((address_space(3))) void*)(lds + 65536 + ((wave - 4) * 8 + (j & 7)) * 1024), 16, 0, 0);
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const long long t1 = clock64();
    if (lane == 0 && blockIdx.x == 0) clk[wave] = (unsigned long long)(t1 - t0);
    out[blockIdx.x * 512 + threadIdx.x] = (float)(acc[0] ^ acc[1] ^ acc[2] ^ acc[3]);
}
template <int DMA_PER_ITER> void run() {
    char* src; float* d; unsigned long long* clk;
    hipMalloc(&src, (size_t)64 * 131072); hipMemset(src, 1, (size_t)64 * 131072); hipMalloc(&d, 256 * 512 * 4); hipMalloc(&clk, 64);
    hipFuncSetAttribute((const void*)k<DMA_PER_ITER>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    const int iters = 4096;
    k<DMA_PER_ITER><<<256, 512, 131072>>>(src, d, 64, clk);
    k<DMA_PER_ITER><<<256, 512, 131072>>>(src, d, iters, clk);
    hipDeviceSynchronize();
    unsigned long long h[8]; hipMemcpy(h, clk, 64, hipMemcpyDeviceToHost);
    const double rd_clk = (h[0] + h[1] + h[2] + h[3]) / 4.0, wr_clk = (h[4] + h[5] + h[6] + h[7]) / 4.0;
    const double rd_bytes = 4.0 * iters * 16 * 1024, dma_bytes = 4.0 * iters * DMA_PER_ITER * 1024;
    printf("DMA %d KiB per writer-iteration: readers %.1f B/clk (s_memtime clocks), DMA %.1f B/clk over its own runtime\n", DMA_PER_ITER,
           rd_bytes / rd_clk, DMA_PER_ITER ? dma_bytes / wr_clk : 0.0);
    hipFree(src); hipFree(d); hipFree(clk);
}
int main() { run<0>(); run<1>(); run<2>(); run<4>(); run<8>(); return 0; }

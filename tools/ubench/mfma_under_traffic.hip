// Does independent memory traffic slow the matrix pipe (clock throttling / shared issue resources)?  Per block: waves 0-3 run
// back-to-back MFMAs (one per SIMD, no memory at all), waves 4-7 (their SIMD siblings) stream LDS-DMA from an HBM-sized or an
// L2-sized window, or idle.  Reports the MFMA rate of waves 0-3.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_under_traffic.hip -o tools/ubench/mfma_under_traffic
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int TRAFFIC> __global__ __launch_bounds__(512) void k(const char* src, float* out, int iters, int window_mask, int dma_per_iter, unsigned long long* ticks) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    float s = 0.f;
    if (wave < 4) {
        f16x8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.01f * (lane + i)); b[i] = (_Float16)(0.02f * i); }
        f32x16 c[4];
        for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) c[j][i] = 0.f;
        const unsigned long long t0 = wall_clock64();          // constant-rate counter (100 MHz): the MFMA waves are timed on their own,
        for (int it = 0; it < iters; ++it) {                    // the DMA waves may run on after them
#pragma unroll
            for (int j = 0; j < 8; ++j) c[j & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c[j & 3], 0, 0, 0);
        }
        for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) s += c[j][i];
        if (lane == 0 && wave == 0) ticks[blockIdx.x] = wall_clock64() - t0;
    } else if (TRAFFIC) {
        const char* base = src + (size_t)(blockIdx.x & window_mask) * 1048576 + (wave - 4) * 262144 + lane * 16;
        for (int it = 0; it < iters; ++it) {                 // per iteration (8 MFMAs = 256 clk on the sibling): dma_per_iter KiB per wave
            for (int j = 0; j < dma_per_iter; ++j)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + (((it * dma_per_iter + j) & 255) * 1024)),
                                                 (__attribute__((address_space(3))) void*)(lds + ((wave - 4) * 8 + (j & 7)) * 1024), 16, 0, 0);
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    out[blockIdx.x * 512 + threadIdx.x] = s + lds[threadIdx.x];
}
template <int TRAFFIC> void run(const char* name, int window_mask, int dma_per_iter) {
    char* src; float* d; unsigned long long* ticks;
    hipMalloc(&src, (size_t)4096 * 1048576 / 4); hipMalloc(&d, 256 * 512 * 4); hipMalloc(&ticks, 256 * 8);    // 1 GiB
    hipFuncSetAttribute((const void*)k<TRAFFIC>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    const int iters = 20000;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<TRAFFIC><<<256, 512, 65536>>>(src, d, 100, window_mask, dma_per_iter, ticks);
    hipEventRecord(a);
    k<TRAFFIC><<<256, 512, 65536>>>(src, d, iters, window_mask, dma_per_iter, ticks);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    unsigned long long h[256]; hipMemcpy(h, ticks, sizeof(h), hipMemcpyDeviceToHost);
    double mean = 0; for (int i = 0; i < 256; ++i) mean += h[i]; mean /= 256;
    const double mfma_ms = mean / 100e6 * 1e3;                 // 100 MHz counter
    const double flops = 256.0 * 4 * iters * 8 * 32768;
    const double bytes = TRAFFIC ? 256.0 * 4 * iters * dma_per_iter * 1024 : 0;
    printf("%-46s MFMA waves: %6.2f ms = %7.1f TFLOP/s   (whole kernel %.2f ms, DMA %.2f TB/s over it)\n", name, mfma_ms, flops / mfma_ms / 1e9, ms,
           bytes / ms / 1e9);
    hipFree(ticks);
    hipFree(src); hipFree(d);
}
int main() {
    run<0>("siblings idle", 0, 0);
    run<1>("siblings: 1 KiB/iter each, 4 MiB window (L2)", 3, 1);
    run<1>("siblings: 2 KiB/iter each, 4 MiB window (L2)", 3, 2);
    run<1>("siblings: 1 KiB/iter each, 1 GiB window (HBM)", 1023, 1);
    run<1>("siblings: 2 KiB/iter each, 1 GiB window (HBM)", 1023, 2);
    return 0;
}

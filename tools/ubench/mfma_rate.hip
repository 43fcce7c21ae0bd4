// MFMA issue-rate microbenchmark for gfx950 (no memory traffic): achievable dense TFLOP/s per instruction shape.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_rate.hip -o tools/ubench/mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int OP, int CHAINS> __global__ __launch_bounds__(256) void k(float* out, int iters) {
    f16x8 a, b;
    bf16x8 ab, bb;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.01f * (threadIdx.x + i)); b[i] = (_Float16)(0.02f * i); ab[i] = (__bf16)(float)a[i]; bb[i] = (__bf16)(float)b[i]; }
    f32x16 c[CHAINS];
    f32x4 d[CHAINS];
    for (int j = 0; j < CHAINS; ++j) { for (int i = 0; i < 16; ++i) c[j][i] = 0.f; for (int i = 0; i < 4; ++i) d[j][i] = 0.f; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < CHAINS; ++j) {
            if (OP == 0) c[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c[j], 0, 0, 0);
            else if (OP == 1) d[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, d[j], 0, 0, 0);
            else if (OP == 2) c[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, c[j], 0, 0, 0);
        }
    }
    float s = 0;
    for (int j = 0; j < CHAINS; ++j) { for (int i = 0; i < 16; ++i) s += c[j][i]; for (int i = 0; i < 4; ++i) s += d[j][i]; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP, int CHAINS> void run(const char* name, double flops_per_inst, int blocks_per_cu) {
    float* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    const int iters = 8192, blocks = 256 * blocks_per_cu;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<OP, CHAINS><<<blocks, 256>>>(d, 16);
    hipEventRecord(a);
    k<OP, CHAINS><<<blocks, 256>>>(d, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double winst = (double)blocks * 4 * iters * CHAINS;
    printf("%-22s chains=%d waves/SIMD=%d  %8.3f ms  %7.1f TFLOP/s   %.1f clk/inst/SIMD @2.4GHz\n", name, CHAINS, blocks_per_cu, ms,
           winst * flops_per_inst / ms / 1e9, (256 * 4 * 2.4e9) / (winst / ms * 1e3));
    hipFree(d);
}
int main() {
    run<0, 4>("mfma_32x32x16_f16", 32768, 1);
    run<0, 4>("mfma_32x32x16_f16", 32768, 2);
    run<0, 1>("mfma_32x32x16_f16", 32768, 4);
    run<0, 2>("mfma_32x32x16_f16", 32768, 4);
    run<2, 4>("mfma_32x32x16_bf16", 32768, 2);
    run<1, 4>("mfma_16x16x32_f16", 16384, 2);
    run<1, 8>("mfma_16x16x32_f16", 16384, 2);
    return 0;
}

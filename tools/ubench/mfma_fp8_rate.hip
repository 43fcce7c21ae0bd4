// Issue rate of v_mfma_scale_f32_32x32x64_f8f6f4 (fp8) from ONE wave per SIMD vs two: the phased GEMM schedule lets one wave per SIMD
// own the matrix pipe during its MFMA interval, so what a single wave can issue back to back bounds gemm_fp8_kernel.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_fp8_rate.hip -o tools/ubench/mfma_fp8_rate && tools/ubench/mfma_fp8_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <int NACC, bool FP8, int THREADS>
__global__ __launch_bounds__(THREADS) void k(float* out, int iters, int seed) {
    i32x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = seed * (threadIdx.x + 1) * 0x01010101 & 0x3f3f3f3f; b[i] = seed * (threadIdx.x + 7) * 0x01010101 & 0x3f3f3f3f; }
    f32x16 c[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) c[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            if constexpr (FP8) c[i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c[i], 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
            else {
                f16x8 ha = __builtin_bit_cast(f16x8, __builtin_shufflevector(a, a, 0, 1, 2, 3)), hb = __builtin_bit_cast(f16x8, __builtin_shufflevector(b, b, 0, 1, 2, 3));
                c[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, c[i], 0, 0, 0);
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += c[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC, bool FP8, int threads> void run(const char* name, int seed) {
    float* out; hipMalloc(&out, 256 * 512 * 4);
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<NACC, FP8, threads>), dim3(256), dim3(threads), 0, 0, out, iters, seed);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double fl = 256.0 * (threads / 64) * iters * NACC * (FP8 ? 131072.0 : 32768.0);
        if (rep) printf("%-44s waves/SIMD %d acc %2d operands %s: %7.1f TF/s\n", name, threads / 256, NACC, seed ? "random" : "zero", fl / ms / 1e9);
    }
    hipFree(out);
}
int main() {
    for (int seed : {0, 12345}) {
        run<8, true, 256>("fp8 scaled 32x32x64", seed); run<8, true, 512>("fp8 scaled 32x32x64", seed); run<4, true, 1024>("fp8 scaled 32x32x64", seed);
        run<2, true, 256>("fp8 scaled 32x32x64", seed); run<1, true, 256>("fp8 scaled 32x32x64 (dependent chain)", seed);
        run<8, false, 256>("f16 32x32x16", seed); run<8, false, 512>("f16 32x32x16", seed); run<1, false, 256>("f16 32x32x16 (dependent chain)", seed);
    }
    return 0;
}

// Average shader clock WHILE the library GEMM runs: a one-wave probe kernel on a second stream shares a CU with the GEMM (it needs
// no LDS and a handful of registers) and reads s_memtime (shader-clock cycles) against s_memrealtime (constant 100 MHz).
// Run with random and with all-zero operands: the difference is the data-dependent (power-managed) clock.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/gemm_clock.hip -o tools/ubench/gemm_clock -ldl
// run:   tools/ubench/gemm_clock <path to libvideoitg_hip.so> [M N K]
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef int (*gemm_fn)(const void*, int, const void*, int, int, int, int, int, int, const float*, int, void*, int, const float*, int, int, void*);

__global__ void probe(long long* out, volatile int* stop) {
    const long long c0 = clock64(), w0 = wall_clock64();
    while (!*stop) __builtin_amdgcn_s_sleep(64);
    out[0] = clock64() - c0; out[1] = wall_clock64() - w0;
}
__global__ void fill(unsigned short* p, size_t n, int zero, unsigned seed) {       // bf16 in [-1, 1) or zeros
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u + seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        const float v = zero ? 0.f : (float)(h & 0xffffff) / 8388608.f - 1.f;
        p[i] = (unsigned short)(__float_as_uint(v) >> 16);
    }
}

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: gemm_clock libvideoitg_hip.so [M N K]\n"); return 2; }
    void* lib = dlopen(argv[1], RTLD_NOW);
    if (!lib) { fprintf(stderr, "%s\n", dlerror()); return 1; }
    gemm_fn gemm = (gemm_fn)dlsym(lib, "vitg_gemm");
    const int M = argc > 4 ? atoi(argv[2]) : 8192, N = argc > 4 ? atoi(argv[3]) : 8192, K = argc > 4 ? atoi(argv[4]) : 8192;
    unsigned short *A, *W, *O; float* bias; long long* clk; int* stop;
    hipMalloc(&A, (size_t)M * K * 2); hipMalloc(&W, (size_t)N * K * 2); hipMalloc(&O, (size_t)M * N * 2); hipMalloc(&bias, N * 4);
    hipMemset(bias, 0, N * 4);
    hipHostMalloc(&clk, 16); hipHostMalloc(&stop, 4);
    hipStream_t s1, s2; hipStreamCreate(&s1); hipStreamCreate(&s2);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int zero = 0; zero < 2; ++zero) {
        fill<<<1024, 256, 0, s1>>>(A, (size_t)M * K, zero, 1u);
        fill<<<1024, 256, 0, s1>>>(W, (size_t)N * K, zero, 2u);
        for (int i = 0; i < 3; ++i) gemm(A, K, W, K, M, N, K, /*bf16*/ 0, /*STORE16*/ 0, bias, 0, O, N, nullptr, 0, 0, s1);
        hipStreamSynchronize(s1);
        const int reps = (int)(4e14 / (2.0 * M * N * K)) + 1;          // ~0.3 s of GEMM
        *stop = 0;
        probe<<<1, 64, 0, s2>>>(clk, stop);
        hipEventRecord(e0, s1);
        for (int i = 0; i < reps; ++i) gemm(A, K, W, K, M, N, K, 0, 0, bias, 0, O, N, nullptr, 0, 0, s1);
        hipEventRecord(e1, s1);
        hipStreamSynchronize(s1);
        *stop = 1;
        hipStreamSynchronize(s2);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%dx%dx%d bf16 %-6s operands: %7.1f TFLOP/s, shader clock %4.0f MHz over %.0f ms\n", M, N, K, zero ? "zero" : "random",
               2.0 * M * N * K * reps / ms / 1e9, (double)clk[0] / (double)clk[1] * 100.0, (double)clk[1] / 1e5);
    }
    return 0;
}

// VALU / transcendental issue-rate microbenchmark for gfx950: how many wave-instructions per cycle per SIMD?
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_rate.hip -o tools/ubench/valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));

template <int OP> __global__ void k(float* out, int iters) {
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = 0.001f * (threadIdx.x + i);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) x[i] = __builtin_amdgcn_exp2f(x[i]);                         // v_exp_f32
            else if (OP == 1) x[i] = fmaf(x[i], 1.0001f, 0.5f);                       // v_fma_f32
            else if (OP == 2) { _Float16 h = (_Float16)x[i]; asm volatile("v_exp_f16 %0, %0" : "+v"(h)); x[i] = (float)h; }
            else if (OP == 3) x[i] = __builtin_amdgcn_rcpf(x[i]);                     // v_rcp_f32
        }
        if (OP == 4) {                                                                // v_pk_fma_f32: 2 values per instruction
#pragma unroll
            for (int i = 0; i < 8; i += 2) {
                f32x2 v = {x[i], x[i + 1]};
                v = __builtin_elementwise_fma(v, f32x2{1.0001f, 1.0001f}, f32x2{0.5f, 0.5f});
                x[i] = v[0]; x[i + 1] = v[1];
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP> void run(const char* name, int per_iter) {
    float* d; hipMalloc(&d, 256 * 8 * 256 * 4 * 4);
    const int iters = 4096, blocks = 256 * 8;      // 8 blocks x 256 threads per CU -> 8 waves per SIMD
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<OP><<<blocks, 256>>>(d, 16);
    hipEventRecord(a);
    k<OP><<<blocks, 256>>>(d, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double winst = (double)blocks * 4 * iters * per_iter;          // wave-instructions
    printf("%-14s %8.3f ms  %.3e wave-inst/s  = %.3f wave-inst/clk/SIMD @2.4GHz  (%.1f clk per wave-inst)\n", name, ms, winst / ms * 1e3,
           winst / ms * 1e3 / (256 * 4 * 2.4e9), (256 * 4 * 2.4e9) / (winst / ms * 1e3));
    hipFree(d);
}
int main() {
    run<1>("v_fma_f32", 8);
    run<4>("v_pk_fma_f32", 4);
    run<0>("v_exp_f32", 8);
    run<2>("v_exp_f16(+cvt)", 8);
    run<3>("v_rcp_f32", 8);
    return 0;
}

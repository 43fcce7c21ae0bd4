// Can one wave overlap MFMA with independent VALU / transcendental work on gfx950?  Time per (1 MFMA + NV VALU ops) slice.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_valu_overlap.hip -o tools/ubench/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// KIND 0: v_exp_f32, 1: v_fma_f32, 2: v_cvt_pk_f16_f32-like (v_add), 3: ds_read_b128 from LDS
template <int NMFMA, int NV, int KIND> __global__ __launch_bounds__(256) void k(float* out, int iters) {
    __shared__ float lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = i;
    __syncthreads();
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.01f * (threadIdx.x + i)); b[i] = (_Float16)(0.02f * i); }
    f32x16 c[4];
    for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) c[j][i] = 0.f;
    float x[16];
    for (int i = 0; i < 16; ++i) x[i] = 0.001f * (threadIdx.x + i);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (NMFMA) c[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c[j], 0, 0, 0);
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                float& r = x[(j * NV + v) & 15];
                if (KIND == 0) r = __builtin_amdgcn_exp2f(r);
                else if (KIND == 1) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(r));
                else if (KIND == 3) { float4 t = *(const float4*)&lds[((threadIdx.x * 4) + (int)(j * 64 + v * 16)) & 4092]; r += t.x; }
                else if (KIND == 4) {
                    typedef short s16x4 __attribute__((ext_vector_type(4)));
                    s16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)&lds[((threadIdx.x * 2) + (int)(j * 64 + v * 16)) & 4094]);
                    r += (float)t[0];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0;
    for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) s += c[j][i];
    for (int i = 0; i < 16; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NMFMA, int NV, int KIND> void run(const char* name, int blocks_per_cu) {
    float* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    const int iters = 4096, blocks = 256 * blocks_per_cu;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<NMFMA, NV, KIND><<<blocks, 256>>>(d, 16);
    hipEventRecord(a);
    k<NMFMA, NV, KIND><<<blocks, 256>>>(d, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double slices = (double)blocks_per_cu * iters * 4;             // slices per SIMD
    printf("%-10s mfma=%d valu/slice=%d waves/SIMD=%d : %7.3f ms  %6.1f ns/slice/SIMD = %5.1f clk @1.9GHz\n", name, NMFMA, NV, blocks_per_cu, ms,
           ms * 1e6 / slices, ms * 1e6 / slices * 1.9);
    hipFree(d);
}
int main() {
    run<1, 0, 0>("mfma only", 1);
    run<0, 2, 0>("exp", 1);  run<1, 2, 0>("exp", 1);  run<1, 2, 0>("exp", 2);
    run<0, 4, 0>("exp", 1);  run<1, 4, 0>("exp", 1);  run<1, 4, 0>("exp", 2);
    run<0, 4, 1>("fma", 1);  run<1, 4, 1>("fma", 1);  run<1, 4, 1>("fma", 2);
    run<0, 8, 1>("fma", 1);  run<1, 8, 1>("fma", 1);  run<1, 8, 1>("fma", 2);
    run<0, 2, 3>("ds_b128", 1);  run<1, 2, 3>("ds_b128", 1);  run<1, 2, 3>("ds_b128", 2);
    run<0, 1, 3>("ds_b128", 1);  run<0, 1, 3>("ds_b128", 2);  run<1, 1, 3>("ds_b128", 2);
    run<0, 2, 4>("ds_tr_b64", 1);  run<0, 2, 4>("ds_tr_b64", 2);  run<1, 2, 4>("ds_tr_b64", 2);
    run<0, 1, 4>("ds_tr_b64", 2);
    return 0;
}

// What does one wave pay (in MFMA issue time) for putting a 1-KiB global->LDS copy in flight?  One wave per SIMD, 4 MFMAs +
// one copy instruction per slice, compared with the MFMA-only loop.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/dma_issue_cost.hip -o tools/ubench/dma_issue_cost
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// KIND 0: none, 1: global_load_lds_dwordx4 (64-bit vaddr), 2: global_load_lds_dwordx4 saddr + 32-bit voffset,
//      3: global_load_dwordx4 -> VGPR (+ ds_write_b128 one slice later), 4: raw buffer load ... lds
template <int KIND, int NM> __global__ __launch_bounds__(256) void k(const char* src, float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.01f * (threadIdx.x + i)); b[i] = (_Float16)(0.02f * i); }
    f32x16 c[4];
    for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) c[j][i] = 0.f;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const char* base = src + (size_t)blockIdx.x * 65536;            // 64 KiB window per block (L2 resident)
    const char* p = base + wave * 16384 + lane * 16;
    const unsigned voff = wave * 16384 + lane * 16;
    u32x4 held = {0, 0, 0, 0};
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 65536, 0x00020000);
    for (int it = 0; it < iters; ++it) {
        const int o = (it & 15) * 1024;
#pragma unroll
        for (int j = 0; j < NM; ++j) {
            c[j & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c[j & 3], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (j == 0) {
                char* dst = lds + wave * 16384 + o;
                if (KIND == 1) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p + o), (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
                else if (KIND == 2) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + o + voff), (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
                else if (KIND == 3) {
                    *(u32x4*)(dst + lane * 16) = held;
                    held = *(const u32x4*)(p + o);
                } else if (KIND == 4) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)dst, 16, voff, o, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if ((it & 7) == 7) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    }
    float s = held[0];
    for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) s += c[j][i];
    s += lds[threadIdx.x * 4];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND, int NM> void run(const char* name) {
    char* src; float* d;
    hipMalloc(&src, 256 * 65536); hipMemset(src, 0, 256 * 65536); hipMalloc(&d, 256 * 256 * 4);
    const int iters = 8192, blocks = 256;
    hipFuncSetAttribute((const void*)k<KIND, NM>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<KIND, NM><<<blocks, 256, 65536>>>(src, d, 16);
    hipEventRecord(a);
    k<KIND, NM><<<blocks, 256, 65536>>>(src, d, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%-34s %d MFMA + 1 copy per slice: %7.3f ms  %6.1f ns/slice  (MFMA-only would be %5.1f ns)\n", name, NM, ms, ms * 1e6 / iters, NM * 16.3);
    hipFree(src); hipFree(d);
}
int main() {
    run<0, 4>("no copy");
    run<1, 4>("global_load_lds vaddr64");
    run<2, 4>("global_load_lds saddr+voff");
    run<4, 4>("buffer_load_lds");
    run<3, 4>("global_load -> vgpr -> ds_write");
    run<1, 2>("global_load_lds vaddr64");
    run<4, 2>("buffer_load_lds");
    run<3, 2>("global_load -> vgpr -> ds_write");
    return 0;
}

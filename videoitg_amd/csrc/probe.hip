// Profiling aid, not on the product path: an MFMA-only kernel with the register shape of the GEMM main loop (8 waves per CU,
// 8 accumulators of 32x32 per wave, no memory traffic in the loop).  On MI355X the shader clock under sustained MFMA load is
// power managed and depends on the operand DATA: the same instruction stream runs at ~2.39 GHz on all-zero operands and at
// ~1.78 GHz (bf16) / ~1.64 GHz (fp16) on random ones (tools/ubench/mfma_power.hip, profiles/r01_gemm_shapes.md section 13), so
// the nominal 2.5 PFLOP/s dense peak is not reachable on real data by any kernel.  bench.py times this probe on the bench box and
// reports both rates next to the roofline so that `frac` can be read against what the silicon sustains.
#include "common.h"
#include "../../include/videoitg_hip.h"

namespace vitg {

template <typename T>
__global__ __launch_bounds__(512, 2) void mfma_probe_kernel(const u32x4* __restrict__ pool, float* __restrict__ out, int iters) {
    typedef typename T16<T>::vec8 vec8;
    u32x4 fr[8];                                         // 4 "A" and 4 "B" fragments, 16 bytes per lane each
#pragma unroll
    for (int i = 0; i < 8; ++i) fr[i] = pool[i * 64 + (threadIdx.x & 63)];
    f32x16 c[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) c[i][j][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) c[i][j] = T16<T>::mfma32(*(vec8*)&fr[u * 2 + i], *(vec8*)&fr[4 + j], c[i][j]);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) s += c[i][j][r];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

// the same for the fp8 weight path's instruction: v_mfma_scale_f32_32x32x64_f8f6f4 on e4m3 bytes (32 bytes per lane and operand:
// two 16-byte pool fragments each), block scales 2^0
__global__ __launch_bounds__(512, 2) void mfma_probe_fp8_kernel(const u32x4* __restrict__ pool, float* __restrict__ out, int iters) {
    typedef int i32x8 __attribute__((ext_vector_type(8)));
    i32x8 fr[4];                                         // 2 "A" and 2 "B" operands
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const u32x4 lo = pool[(2 * i) * 64 + (threadIdx.x & 63)], hi = pool[(2 * i + 1) * 64 + (threadIdx.x & 63)];
        fr[i] = i32x8{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
    }
    f32x16 c[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) c[i][j][r] = 0.f;
    const int one = 0x7F7F7F7F;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    c[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fr[(u + i) & 1], fr[2 + ((u + j) & 1)], c[i][j], 0, 0, 0, one, 0, one);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) s += c[i][j][r];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

}  // namespace vitg

using namespace vitg;

extern "C" int vitg_mfma_probe(const void* pool, int dtype, int blocks, int iters, float* out, double* flops, void* stream) {
    if (!pool || !out || blocks <= 0 || iters <= 0) return VITG_ERR_ARG;
    if (dtype == 3) {                                    // e4m3 bytes through the block-scaled fp8 MFMA (pool = any 8 x 64 x 16 bytes)
        hipLaunchKernelGGL(mfma_probe_fp8_kernel, dim3(blocks), dim3(512), 0, (hipStream_t)stream, (const u32x4*)pool, out, iters);
        if (flops) *flops = (double)blocks * 8.0 * iters * 16.0 * 131072.0;      // 32x32x64 MFMA = 131072 flop per wave instruction
        return hipGetLastError() == hipSuccess ? 0 : VITG_ERR_LAUNCH;
    }
    if (dtype != DT_BF16 && dtype != DT_F16) return VITG_ERR_DTYPE;
    if (dtype == DT_BF16) hipLaunchKernelGGL(mfma_probe_kernel<bf16_t>, dim3(blocks), dim3(512), 0, (hipStream_t)stream, (const u32x4*)pool, out, iters);
    else hipLaunchKernelGGL(mfma_probe_kernel<f16_t>, dim3(blocks), dim3(512), 0, (hipStream_t)stream, (const u32x4*)pool, out, iters);
    if (flops) *flops = (double)blocks * 8.0 * iters * 16.0 * 32768.0;           // 32x32x16 MFMA = 32768 flop per wave instruction
    return hipGetLastError() == hipSuccess ? 0 : VITG_ERR_LAUNCH;
}

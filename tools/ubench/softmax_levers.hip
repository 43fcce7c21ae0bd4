// Two hardware questions behind VERDICT r3 item 3 (SigLIP attention, head dim 72), answered on the chip:
//  (1) does a HALF-K MFMA cost half?  QK^T over d = 72 runs 5 k-steps of 16 today (80: one step is half padding).  If
//      v_mfma_f32_32x32x8_f16 (the CDNA3-era K = 8 form) took 16 instead of 32 pipe cycles, "4 x K16 + 1 x K8" would drop the padding.
//      Same for v_mfma_f32_16x16x16_f16 against v_mfma_f32_16x16x32_f16.
//  (2) is exp2 cheaper as a polynomial on the plain VALU than on the transcendental unit?  v_exp_f32 against the FA-4 style emulation
//      (round-to-nearest split by the 1.5 * 2^23 magic number, degree-3 minimax on [-0.5, 0.5], exponent added into the float's bits),
//      scalar and packed, each alone and beside an MFMA stream (1 MFMA + N exponentials per slice, two waves per SIMD).
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/softmax_levers.hip -o tools/ubench/softmax_levers
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// ---- (1) MFMA cost per shape ------------------------------------------------------------------------------------------------------
template <int OP> __global__ __launch_bounds__(256) void k_mfma(float* out, int iters) {
    f16x8 a, b;
    f16x4 a4, b4;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.01f * (threadIdx.x + i)); b[i] = (_Float16)(0.02f * i); }
    for (int i = 0; i < 4; ++i) { a4[i] = a[i]; b4[i] = b[i]; }
    f32x16 c[4];
    f32x4 d[8];
    for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) c[j][i] = 0.f;
    for (int j = 0; j < 8; ++j) for (int i = 0; i < 4; ++i) d[j][i] = 0.f;
    for (int it = 0; it < iters; ++it) {
        if (OP == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) c[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c[j], 0, 0, 0);
        } else if (OP == 1) {
#pragma unroll
            for (int j = 0; j < 4; ++j) c[j] = __builtin_amdgcn_mfma_f32_32x32x8f16(a4, b4, c[j], 0, 0, 0);
        } else if (OP == 2) {
#pragma unroll
            for (int j = 0; j < 8; ++j) d[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, d[j], 0, 0, 0);
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) d[j] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, d[j], 0, 0, 0);
        }
    }
    float s = 0;
    for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) s += c[j][i];
    for (int j = 0; j < 8; ++j) for (int i = 0; i < 4; ++i) s += d[j][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// ---- (2) exp2: transcendental unit vs polynomial on the plain VALU -------------------------------------------------------------------
__device__ __forceinline__ float exp2_poly(float x) {
    // x <= 0 in the softmax; clamp keeps the exponent field in range.  n = round(x), f = x - n in [-0.5, 0.5];
    // 2^f ~ 1 + f (c1 + f (c2 + f c3)) (minimax, max rel err 1.1e-4 - more than P's fp16 rounding needs); result bits += n << 23.
    x = fmaxf(x, -126.0f);
    const float magic = 12582912.0f;                       // 1.5 * 2^23
    const float t = x + magic;
    const float n = t - magic;
    const float f = x - n;
    float p = __builtin_fmaf(f, 0.0555054f, 0.2402265f);
    p = __builtin_fmaf(f, p, 0.6931472f);
    p = __builtin_fmaf(f, p, 1.0f);
    return __builtin_bit_cast(float, __builtin_bit_cast(int, p) + (__builtin_bit_cast(int, t) << 23));
}
__device__ __forceinline__ f32x2 exp2_poly2(f32x2 x) {     // the same on the packed fp32 forms (v_pk_add / v_pk_fma)
    const f32x2 lo = {-126.0f, -126.0f}, magic = {12582912.0f, 12582912.0f};
    x = f32x2{fmaxf(x[0], lo[0]), fmaxf(x[1], lo[1])};
    const f32x2 t = x + magic, n = t - magic, f = x - n;
    f32x2 p = __builtin_elementwise_fma(f, f32x2{0.0555054f, 0.0555054f}, f32x2{0.2402265f, 0.2402265f});
    p = __builtin_elementwise_fma(f, p, f32x2{0.6931472f, 0.6931472f});
    p = __builtin_elementwise_fma(f, p, f32x2{1.0f, 1.0f});
    return f32x2{__builtin_bit_cast(float, __builtin_bit_cast(int, p[0]) + (__builtin_bit_cast(int, t[0]) << 23)),
                 __builtin_bit_cast(float, __builtin_bit_cast(int, p[1]) + (__builtin_bit_cast(int, t[1]) << 23))};
}

// KIND 0 v_exp_f32, 1 scalar polynomial, 2 packed polynomial, 3 half of the values on each path (KIND 0 + KIND 1 mixed)
template <int NMFMA, int NV, int KIND> __global__ __launch_bounds__(256) void k_exp(float* out, int iters) {
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.01f * (threadIdx.x + i)); b[i] = (_Float16)(0.02f * i); }
    f32x16 c[4];
    for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) c[j][i] = 0.f;
    float x[16];
    for (int i = 0; i < 16; ++i) x[i] = -0.001f * (threadIdx.x + i);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (NMFMA) c[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c[j], 0, 0, 0);
            if (KIND == 2) {
#pragma unroll
                for (int v = 0; v < NV; v += 2) {
                    float &r0 = x[(j * NV + v) & 15], &r1 = x[(j * NV + v + 1) & 15];
                    const f32x2 e = exp2_poly2(f32x2{r0, r1});
                    r0 = e[0] - 1.5f; r1 = e[1] - 1.5f;
                }
            } else {
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    float& r = x[(j * NV + v) & 15];
                    const bool poly = KIND == 1 || (KIND == 3 && (v & 1));
                    r = (poly ? exp2_poly(r) : __builtin_amdgcn_exp2f(r)) - 1.5f;     // (the subtract keeps the argument negative and bounded)
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

static float time_it(void (*launch)(float*, int, int), float* d, int blocks, int iters) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    launch(d, blocks, 16);
    hipEventRecord(a);
    launch(d, blocks, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms;
}
template <int OP> void launch_mfma(float* d, int blocks, int iters) { k_mfma<OP><<<blocks, 256>>>(d, iters); }
template <int NM, int NV, int KIND> void launch_exp(float* d, int blocks, int iters) { k_exp<NM, NV, KIND><<<blocks, 256>>>(d, iters); }

template <int OP> void run_mfma(const char* name, int per_iter, double flops) {
    float* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    const int iters = 8192, bpc = 2, blocks = 256 * bpc;
    const float ms = time_it(launch_mfma<OP>, d, blocks, iters);
    const double winst = (double)blocks * 4 * iters * per_iter;
    printf("%-28s %8.3f ms  %7.1f TFLOP/s  %5.1f clk per instruction per SIMD @2.4 GHz nominal\n", name, ms, winst * flops / ms / 1e9,
           (256 * 4 * 2.4e9) / (winst / ms * 1e3));
    hipFree(d);
}
template <int NM, int NV, int KIND> void run_exp(const char* name, int bpc) {
    float* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    const int iters = 4096, blocks = 256 * bpc;
    const float ms = time_it(launch_exp<NM, NV, KIND>, d, blocks, iters);
    const double slices = (double)bpc * iters * 4;                  // slices per SIMD
    printf("%-34s mfma=%d exp/slice=%d waves/SIMD=%d : %7.3f ms  %6.1f clk per slice per SIMD @2.4 GHz nominal  (%5.1f per exponential)\n", name, NM, NV, bpc, ms,
           ms * 1e-3 * 2.4e9 / slices, ms * 1e-3 * 2.4e9 / slices / NV);
    hipFree(d);
}

__global__ void k_acc(float* out) {      // accuracy of the polynomial over the softmax's argument range
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const float x = -24.0f * (float)i / (float)(gridDim.x * blockDim.x);
    const float e = exp2_poly(x), r = __builtin_amdgcn_exp2f(x);
    out[i] = fabsf(e - r) / r;
}

int main() {
    printf("## (1) MFMA pipe cost per shape (2 waves / SIMD, independent accumulators)\n");
    run_mfma<0>("v_mfma_f32_32x32x16_f16", 4, 32768);
    run_mfma<1>("v_mfma_f32_32x32x8_f16", 4, 16384);
    run_mfma<2>("v_mfma_f32_16x16x32_f16", 8, 16384);
    run_mfma<3>("v_mfma_f32_16x16x16_f16", 8, 8192);
    printf("## (2) exp2: v_exp_f32 vs polynomial on the plain VALU (8 per slice = what one 32x32x16 MFMA of QK^T produces per lane pair of tiles)\n");
    run_exp<0, 8, 0>("v_exp_f32 alone", 2);
    run_exp<0, 8, 1>("polynomial alone", 2);
    run_exp<0, 8, 2>("packed polynomial alone", 2);
    run_exp<0, 8, 3>("half and half alone", 2);
    run_exp<1, 0, 0>("MFMA alone", 2);
    run_exp<1, 8, 0>("MFMA + v_exp_f32", 2);
    run_exp<1, 8, 1>("MFMA + polynomial", 2);
    run_exp<1, 8, 2>("MFMA + packed polynomial", 2);
    run_exp<1, 8, 3>("MFMA + half and half", 2);
    run_exp<1, 8, 0>("MFMA + v_exp_f32", 4);
    run_exp<1, 8, 3>("MFMA + half and half", 4);
    float* d; hipMalloc(&d, 65536 * 4);
    k_acc<<<256, 256>>>(d);
    static float h[65536];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    float mx = 0; for (int i = 0; i < 65536; ++i) mx = h[i] > mx ? h[i] : mx;
    printf("polynomial max relative error vs v_exp_f32 over [-24, 0]: %.3e\n", mx);
    return 0;
}

// Shared device helpers for the VideoITG gfx950 (CDNA4) kernels.  wave = 64 lanes everywhere.
#pragma once
#ifndef VITG_PROFILING_BUILD
#define VITG_PROFILING_BUILD 0     // 1 only for libvideoitg_hip_prof.so (build.sh --profiling): research variants + ablation switches
#endif
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>

namespace vitg {

typedef __bf16 bf16_t;
typedef _Float16 f16_t;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

enum : int { DT_BF16 = 0, DT_F16 = 1, DT_F32 = 2 };

// ---- 16-bit element traits -------------------------------------------------------------------
template <typename T> struct T16;
template <> struct T16<bf16_t> {
    typedef bf16x8 vec8;
    typedef bf16x4 vec4;
    static __device__ __forceinline__ float to_f32(bf16_t v) { return (float)v; }
    static __device__ __forceinline__ bf16_t from_f32(float f) { return (bf16_t)f; }   // RNE
    static __device__ __forceinline__ f32x16 mfma32(vec8 a, vec8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x4 mfma16(vec8 a, vec8 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
};
template <> struct T16<f16_t> {
    typedef f16x8 vec8;
    typedef f16x4 vec4;
    static __device__ __forceinline__ float to_f32(f16_t v) { return (float)v; }
    static __device__ __forceinline__ f16_t from_f32(float f) { return (f16_t)f; }
    static __device__ __forceinline__ f32x16 mfma32(vec8 a, vec8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x4 mfma16(vec8 a, vec8 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
};

template <typename V, typename U> __device__ __forceinline__ V bitcast(const U& u) {
    static_assert(sizeof(V) == sizeof(U), "bitcast size");
    return __builtin_bit_cast(V, u);
}

// ---- wave / block reductions -------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// ---- activations -------------------------------------------------------------------------------
__device__ __forceinline__ float gelu_tanh_f(float x) {
    // 0.5 x (1 + tanh(sqrt(2/pi) (x + 0.044715 x^3)))  ==  x * sigmoid(2u)
    // exponent argument -2 log2(e) sqrt(2/pi) (x + 0.044715 x^3) with the constants folded: x * (B + A x^2) = mul, fma, mul (it was
    // five dependent multiplies); v_exp_f32 + v_rcp_f32 (1 ulp) instead of an IEEE division (~10 VALU).  This runs on 128-192 values per
    // lane in the fc1 epilogue, where it costs 14 % of that GEMM (tools/gemm_cfg_sweep.py).
    const float a = x * __builtin_fmaf(x * x, -0.10294324302f, -2.30220819814f);
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(a));
}
// two values at a time on the packed fp32 VALU forms (v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32) - the same operations in the same order as the
// scalar forms, so the results are bit-identical.  (Half the instructions, NOT half the time: tools/ubench/valu_rate.hip measures 5.4 clocks per
// v_pk_fma_f32 against 2.7 per v_fma_f32 on gfx950 - the same rate per value.  The gain of the epilogue rewrite came from taking the tests out of
// the quad loop, profiles/r03_gemm_epilogue.md.)
__device__ __forceinline__ f32x2 gelu_tanh_f2(f32x2 x) {
    const f32x2 ca = {-0.10294324302f, -0.10294324302f}, cb = {-2.30220819814f, -2.30220819814f}, one = {1.0f, 1.0f};
    const f32x2 a = x * __builtin_elementwise_fma(x * x, ca, cb);
    const f32x2 d = f32x2{__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1])} + one;
    return x * f32x2{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
}
__device__ __forceinline__ f32x2 silu_mul_f2(f32x2 gate, f32x2 up) {          // silu(gate) * up
    const f32x2 c = {-1.4426950408889634f, -1.4426950408889634f}, one = {1.0f, 1.0f};
    const f32x2 a = c * gate;
    const f32x2 d = one + f32x2{__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1])};
    return (gate * f32x2{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])}) * up;
}
__device__ __forceinline__ float gelu_erf_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.7071067811865476f)); }
__device__ __forceinline__ float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x)); }

__host__ __device__ __forceinline__ int ceil_div(int a, int b) { return (a + b - 1) / b; }

// Per-device host-side caches: one bit / one slot per device instead of a single "last device" variable, so host threads driving
// different GPUs neither thrash nor race each other's state.  Set-up work guarded by a bit is idempotent (hipFuncSetAttribute), and the bit is
// set AFTER it ran, so a second thread either repeats it or sees it finished.
inline int current_device() { int d = 0; (void)hipGetDevice(&d); return d; }
inline bool device_done(const std::atomic<unsigned>& mask, int dev) { return (mask.load(std::memory_order_acquire) >> (dev & 31)) & 1u; }
inline void mark_device_done(std::atomic<unsigned>& mask, int dev) { mask.fetch_or(1u << (dev & 31), std::memory_order_release); }
inline int device_cu_count(int dev) {
    static std::atomic<int> cus[32];
    int c = cus[dev & 31].load(std::memory_order_relaxed);
    if (c <= 0) {
        if (hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || c <= 0) c = 256;
        cus[dev & 31].store(c, std::memory_order_relaxed);
    }
    return c;
}

// Host-side tuning knobs (vitg_set_option / environment VITG_*): kernel selection for A/B measurements and tests.
struct Tuning {
    int gemm_tile;     // 0 auto | 128 | 256
    int gemm_cfg;      // 0 auto | 1 256x256x32 ring-4 | 2 256x128x32 ring-3 | 3 256x256x64 ring-2 | 4 persistent 256x256x64
    int gemm_flags;    // bit0 non-temporal stores; bits 1-3 profiling ablations (wrong results)
    int gemm_stagger;  // fp32-residual GEMMs in the tile loop: start the tile groups a third of a tile period apart (0 off, 1 = estimated period, > 1 = ns);
                       // +6 % on the out-proj shape alone, nothing measurable in the step (profiles/r02_gemm_timeline.md): default off
    int gemm_group;    // gemm256: M-tiles per tile-order group (0 = default 4)
    int gemm_pref;     // A/B: forced tile shape per epilogue mode as three decimal digits <16-bit out><fp32 residual><SwiGLU>, digit 0 auto | 1 256x256 | 2 256x384 | 3 384x256
    int gemm_mi;       // MFMA shape of the 256-tile GEMMs: 0 auto (= 16) | 16 v_mfma_f32_16x16x32 (gemm_mi16.hip) | 32 v_mfma_f32_32x32x16 (gemm256.hip)
    int gemm_mi_modes; // A/B: epilogue modes that take the 16x16x32 kernels when gemm_mi selects them, bit0 16-bit output, bit1 fp32 residual, bit2 SwiGLU (0 = all)
    int attn_waves;    // 0 auto | 4 | 8
    int attn_pipe;     // 0 off (default) | 1 cross-tile pipelined hd-72 variant
    int attn_dma;      // 0 auto (DMA for hd 72, registers for hd 128) | 1 LDS-DMA K/V staging | 2 register staging
    int attn_ab;       // hd 72: 0 auto | 1 half-tile-out-of-phase kernel (attention_ab.hip) | 2 off | 3 the same kernel with 16 waves / 512 queries per block | 4 its PV-on-32x32x16 form | 5 PV on 16x16x32 with VALU row sums | 6 pre-scaled Q without the shift column (A/B)
    int attn_sp;       // 0 auto | 1 software-pipelined kernel (attention_sp.hip) | 2 per-tile kernel (attention.hip) | 3 hd 128: software-pipelined with PV on 16x16x32 (opt-in, slower)
    int gemm_wm4;      // 384x256 tile of gemm_mi16.hip as a 4 x 2 wave grid (96 x 128 wave tiles): 0 library default | -1 never | bit mask of epilogue modes (1 16-bit output, 2 fp32 residual, 4 SwiGLU, 8 16-bit residual)
};
Tuning& tuning();

}  // namespace vitg

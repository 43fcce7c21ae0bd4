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

// Host-side kernel-selection knobs (vitg_set_option / environment VITG_*).
// PRODUCT library: FOUR keys, each a choice between the measured default and one cross-check kernel set (what tests/test_gpu_e2e.py's
// test_full_size_invariants runs the same video through); everything that was an A/B switch of rounds 1-3 - and every kernel variant that lost
// its A/B - lives in the profiling library only (RESEARCH(field) below is the constant 0 in the product, so the host code folds and a stray
// option or environment variable cannot select a variant the product does not ship).
struct Tuning {
    // ---- product keys (include/videoitg_hip.h) -------------------------------------------------------------------------------
    int gemm_tile;     // 0 auto | 128 the 128x128 register-staged kernels everywhere (cross-check set) | 256 never the 128-tile kernel where a 256-wide one applies
    int gemm_cfg;      // 0 auto (choose_cfg per shape) | 2 256x128x32 | 6 256x256x64 | 9 256x384x64 | 10 384x256x64 | 12 320x256x64 (SwiGLU / fp32-residual GEMMs; others fall to 256x256) : forces ONE shipped tile shape (tests reach every shipped kernel on small shapes)
    int attn_ab;       // hd 72: 0 auto (half-tile-out-of-phase kernel when 256-query blocks fill the chip) | 2 off (per-tile kernel: cross-check)
    int attn_sp;       // hd 128: 0 auto (8-wave software-pipelined kernel) | 2 per-tile kernel (cross-check)   [4 = the one-wave-per-SIMD kernel, csrc/attention_w4.hip: profiling library only since round 6]
    // ---- research knobs: libvideoitg_hip_prof.so only (include/videoitg_hip_profiling.h) -------------------------------------
    int gemm_flags;    // A/B and ablation bits (see the kernels)
    int gemm_stagger;  // fp32-residual GEMMs in the tile loop: start the tile groups a third of a tile period apart (0 off, 1 = estimated period, > 1 = ns);
                       // +6 % on the out-proj shape alone, nothing measurable in the step (profiles/r02_gemm_timeline.md)
    int gemm_group;    // M-tiles per tile-order group (0 = 4)
    int gemm_pref;     // forced tile shape per epilogue mode as three decimal digits <16-bit out><fp32 residual><SwiGLU>, digit 0 auto | 1 256x256 | 2 256x384 | 3 384x256
    int gemm_mi;       // MFMA shape of the 256-wide tiles: 0 / 16 v_mfma_f32_16x16x32 (gemm_mi16.hip) | 32 v_mfma_f32_32x32x16 (gemm256.hip; lost by 4-14 %, profiles/r03_gemm_mi16.md)
    int gemm_mi_modes; // epilogue modes that take the 16x16x32 kernels when gemm_mi selects them, bit0 16-bit output, bit1 fp32 residual, bit2 SwiGLU (0 = all)
    int attn_waves;    // 0 auto | 4 | 8
    int attn_pipe;     // 1 cross-tile pipelined hd-72 variant (measured equal to the plain loop)
    int attn_dma;      // 0 auto (DMA for hd 72, registers for hd 128) | 1 LDS-DMA K/V staging | 2 register staging
    int gemm_wm4;      // 384x256 tile of gemm_mi16.hip as a 4 x 2 wave grid (96 x 128 wave tiles): 0 library default | -1 never | bit mask of epilogue modes
                       // (+3.7 % at long K, -1.7 % at K = 1152, nothing in the step)
    // research VALUES of the product keys (profiling library only): gemm_cfg 1/3/4/5/7/8/11/13; attn_ab 1 forced on | 3 16 waves per block | 4 PV on 32x32x16 |
    // 5 VALU row sums | 6 pre-scaled Q without the shift column; attn_sp 1 8-wave kernel forced on for both head sizes | 3 hd 128 with PV on 16x16x32 (3-4 % slower) | 5 / 6 / 7 / 100+ research forms of attention_w4.hip
};
Tuning& tuning();
#if VITG_PROFILING_BUILD
#define RESEARCH(field) (::vitg::tuning().field)
#else
#define RESEARCH(field) 0
#endif

}  // namespace vitg

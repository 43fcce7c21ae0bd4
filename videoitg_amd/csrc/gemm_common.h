// Shared between the GEMM kernels (gemm.hip: 128x128 tile, gemm256.hip: 256xBN ring-buffered tile).
#pragma once
#include "common.h"
#include "../../include/videoitg_hip.h"

namespace vitg {

struct GemmArgs {
    const void* A; int lda;
    const void* W; int ldw;
    int M, N, K;
    const float* bias;        // [N] fp32 or null
    int act;                  // VITG_ACT_*
    void* out; int ldo;       // mode STORE16/SWIGLU: T16*, mode F32: float*
    const float* res; int ldr; int res_mod;   // mode F32: out = acc + bias + res[(m % res_mod), n] (res may be null)
    int tiles_m, tiles_n;
};

enum { MODE_STORE16 = 0, MODE_F32 = 1, MODE_SWIGLU = 2 };

// XCD-contiguous remap (block b runs on XCD b % 8) followed by grouping GM row-tiles: the blocks resident on one
// XCD at a time form a compact super-tile that shares operand panels in that XCD's L2.
__device__ __forceinline__ void block_to_tile(int bid, int tiles_m, int tiles_n, int gm, int& tm, int& tn) {
    const int nwg = tiles_m * tiles_n;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    const int per_group = gm * tiles_n;
    const int grp = bid / per_group;
    const int first_m = grp * gm;
    const int gsz = min(tiles_m - first_m, gm);
    const int in_g = bid - grp * per_group;
    tm = first_m + in_g % gsz;
    tn = in_g / gsz;
}

// Epilogue for one 32x32 accumulator pair position: lane holds C[m][n .. n+3] (4 consecutive output features of one
// token row) in acc[q4*4 .. q4*4+3].
template <typename T, int MODE>
__device__ __forceinline__ void epilogue_quad(const GemmArgs& g, int m, int n, f32x4 v) {
    if (g.bias) v += *(const f32x4*)(g.bias + n);
    if (g.act == VITG_ACT_GELU_TANH) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = gelu_tanh_f(v[e]);
    } else if (g.act == VITG_ACT_GELU_ERF) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = gelu_erf_f(v[e]);
    }
    if constexpr (MODE == MODE_F32) {
        if (g.res) {
            const int rm = g.res_mod > 0 ? m % g.res_mod : m;
            v += *(const f32x4*)(g.res + (size_t)rm * g.ldr + n);
        }
        *(f32x4*)((float*)g.out + (size_t)m * g.ldo + n) = v;
    } else {
        typename T16<T>::vec4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = T16<T>::from_f32(v[e]);
        *(typename T16<T>::vec4*)((T*)g.out + (size_t)m * g.ldo + n) = o;
    }
}

template <typename T>
__device__ __forceinline__ void epilogue_swiglu_quad(const GemmArgs& g, int m, int no, f32x4 gate, f32x4 up) {
    typename T16<T>::vec4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = T16<T>::from_f32(silu_f(gate[e]) * up[e]);
    *(typename T16<T>::vec4*)((T*)g.out + (size_t)m * g.ldo + no) = v;
}

int launch_gemm256(const GemmArgs& g, int dtype, int mode, hipStream_t s);   // gemm256.hip
bool gemm256_applicable(const GemmArgs& g, int mode);

}  // namespace vitg

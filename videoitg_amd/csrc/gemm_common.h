// Shared between the GEMM kernels (gemm.hip: 128x128 tile, gemm256.hip: 256xBN ring-buffered tile).
#pragma once
#include <type_traits>
#include "common.h"
#include "../../include/videoitg_hip.h"

#ifndef VITG_PROFILING_BUILD
#define VITG_PROFILING_BUILD 0     // build.sh -DVITG_PROFILING_BUILD=1: compile the ablation / A-B switches into the kernels (see gemm256.hip)
#endif

namespace vitg {

struct GemmArgs {
    const void* A; int lda;
    const void* W; int ldw;
    int M, N, K;
    const float* bias;        // [N] fp32 or null
    int act;                  // VITG_ACT_*
    void* out; int ldo;       // mode STORE16/SWIGLU: T16*, mode F32: float*
    const float* res; int ldr; int res_mod;   // mode F32: out = acc + bias + res[(m % res_mod), n] (res may be null)
    const void* res16; int ldr16;             // mode STORE16 (vitg_gemm_res16): out = round16(acc + bias + res16[m, n]) - the 16-bit residual stream of the
                                              // tower; `res` (fp32, res_mod) is accepted there too (patch embedding + position table)
    int tiles_m, tiles_n;
    int group_m;              // gemm256: M-tiles per group of the XCD-aware tile order
    int stagger_ns;           // persistent kernel: tile period estimate for de-phasing the XCDs (0 = off)
    unsigned long long* stamps; // profiling build: per-block timeline (vitg_gemm_timeline_buffer), nullptr = off
    int stamp_blocks;         // blocks that record (the first stamp_blocks of the grid), 4 tiles x 2 wave groups x 32 slots each
    int stagger_blocks;       // gemm_wf, two blocks per CU: blocks of the initial fill that take part in the de-phasing
    int flags;                // bit0: non-temporal output stores, bit1: (profiling) skip output stores
    const void* Wf;           // fragment-packed copy of W (vitg_pack_weight_fragments) or null
    // fused normalisation (round 2; NRM instantiations only).  PRODUCER = MODE_F32 with the residual in the accumulator: besides
    // out = acc + bias (+ residual) it writes xh[m,n] = round16(out[m,n] * nrm_gamma[n]) - the operand of the GEMM that follows the
    // next LayerNorm / RMSNorm - and, per wave column block, the partial row statistics (mean, M2) of out.  CONSUMER = 16-bit output
    // modes fed with xh: out = epilogue(rstd[m] * (acc - mean[m] * nrm_colsum[n]) + bias) with (mean, rstd) from the finalised stats.
    const float* nrm_gamma;   // producer: weight of the norm that follows [N]
    void* nrm_xh; int nrm_ldxh;   // producer: 16-bit copy
    float* nrm_part;          // producer: [M][nrm_P][2] partial (mean, M2) over the wave's NT*32 columns
    int nrm_P;                // producer: partials per row = N / (columns per wave block), set by the host entry
    const float* nrm_stats;   // consumer: [M][2] (mean, rstd)
    const float* nrm_colsum;  // consumer: s[n] = sum_k gamma[k] W[n,k] (LayerNorm) or null (RMSNorm: no mean term)
    unsigned* xcd_sync;       // profiling build, gemm_flags bit26: 8 per-XCD arrival counters (128 B apart), zeroed before the launch
    int sk_units, sk_kt;      // stream-K (vitg_gemm_streamk, gemm_mi16.hip): K-tile units per block / K-tiles per output tile; 0 = off
    int split_k;              // MODE_F32 only: gridDim.y = split_k blocks share one output tile, block y accumulates its K range and
    long split_stride;        // writes the raw fp32 partial tile to out + y * split_stride floats (vitg_gemm_splitk reduces them)
    // patch embedding with the im2col inside the A-tile loader (vitg_patch_embed16; gemm_mi16.hip PATCH instantiations): A = pixel_values
    // [T, 3, R, R] (16-bit, NCHW), row m = (frame, py, px) reads its p x p x 3 patch straight from there
    int pt_R, pt_G, pt_P;     // image side, patches per side (R / p), patches per frame (G * G); pt_R == 0: off
    int pt_p, pt_pairs;       // patch side p (8..16), (channel, ky) pairs = 3 p
    long split_off;           // SPLIT instantiations of gemm_mi16.hip (vitg_gemm_split16): element offset of the lo plane in `out`
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

// 16-B output store with a cache policy (profiling build A/B, gemm_flags bits 22/23): wave-uniform 64-bit base + 32-bit lane byte offset.
//   0 plain (the line stays in the XCD's L2) | 1 sc1 (write-through, the L2 drops the line) | 2 nt | 3 sc0 sc1
// An output tile is never read again by the launch that writes it, so every line it leaves in the L2 displaces operand panels the
// other blocks of the XCD are about to fetch (MI355X_MICROARCH.md "stores of each flavour").
__device__ __forceinline__ void store16_policy(char* base, unsigned off, u32x4 v, int pol) {
    if (pol == 1) asm volatile("global_store_dwordx4 %0, %1, %2 sc1\n\ts_nop 1" ::"v"(off), "v"(v), "s"(base) : "memory");
    else if (pol == 2) asm volatile("global_store_dwordx4 %0, %1, %2 nt\n\ts_nop 1" ::"v"(off), "v"(v), "s"(base) : "memory");
    else if (pol == 3) asm volatile("global_store_dwordx4 %0, %1, %2 sc0 sc1\n\ts_nop 1" ::"v"(off), "v"(v), "s"(base) : "memory");
    else *(u32x4*)(base + off) = v;
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
        if (g.res) {                                     // vitg_gemm_res16 forms (a lane reads the quad it is about to overwrite: in place is safe)
            const int rm = g.res_mod > 0 ? m % g.res_mod : m;
            v += *(const f32x4*)(g.res + (size_t)rm * g.ldr + n);
        }
        if (g.res16) {
            const typename T16<T>::vec4 r = *(const typename T16<T>::vec4*)((const T*)g.res16 + (size_t)m * g.ldr16 + n);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += T16<T>::to_f32(r[e]);
        }
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


// ------------------------------------------------------------------------------------------------------------------
// LDS-staged epilogue of one wave's (MT*32) x (NT*32) accumulator tile.  The MFMA C layout gives a lane 4 consecutive
// features of ONE token row, i.e. a store instruction would touch 32 different rows with 16 bytes each.  Instead the
// wave transposes its tile through a private LDS region (XOR-swizzled 16-B chunks) and writes whole rows: 16 B per
// lane, consecutive lanes on consecutive addresses, full 128-B lines.  Bias / activation / SwiGLU are applied in
// registers before staging; the fp32 residual is added on the coalesced side.
//   acc[nt][mt][r]: m = m_base + mt*32 + (lane&31), n = n_base + nt*32 + 8*(r>>2) + 4*(lane>>5) + (r&3)
// `stage` = this wave's private LDS (>= MTN*32*(NT*64 [+16]) bytes for 16-bit output - the +16 applies when NT is not a power of
// two - and >= 32*NT*128 bytes for fp32 output).
// Requires out/res leading dimensions and n_base to keep 16-B alignment (checked by the host code).
// ------------------------------------------------------------------------------------------------------------------
// R16 (MODE_F32 only; gemm_fp8.hip, round 4): the residual is a 16-bit row of type T (g.res16, ld g.ldr16) and so is the output (g.out, ld g.ldo) -
// out16 = round16(acc + bias + res16), the tower's 16-bit residual stream on the fp8 weight path (vitg_gemm_fp8_res16).
template <typename T, int MODE, int MT, int NT, int MT0 = 0, int MTN = MT, bool NRM = false, bool R16 = false>
__device__ __forceinline__ void epilogue_staged(const GemmArgs& g, f32x16 (&acc)[NT][MT], char* stage, int m_base, int n_base,
                                                int lane) {
    // processes accumulator row-tiles mt in [MT0, MT0 + MTN); m_base is the row of mt = MT0
    const int frow = lane & 31, fh = lane >> 5;
    // NRM consumer: this lane's rows carry (mean, rstd) of the norm folded into this GEMM
    float nmu[MTN], nrs[MTN];
    if constexpr (NRM && MODE != MODE_F32) {
#pragma unroll
        for (int i = 0; i < MTN; ++i) {
            const f32x2 st = *(const f32x2*)(g.nrm_stats + 2 * (size_t)min(m_base + i * 32 + frow, g.M - 1));
            nmu[i] = st[0]; nrs[i] = st[1];
        }
    }
    if constexpr (MODE == MODE_F32) {
        constexpr int CH = NT * 8;                        // 16-B chunks (4 floats) per row
        constexpr int RB = NT * 128;                      // row bytes
        constexpr int SWZ = (CH & -CH) - 1;               // XOR mask: largest power of two dividing CH, minus 1 (NT = 3: 24 chunks -> 7).
        // a 16-B write is served 16 lanes (rows) at a time against 16 slots per cycle: RB = 256 puts every row on the same slots (swizzle
        // by row & 15); RB = 384 alternates by row parity, so the swizzle takes the next three row bits (rows r and r + 8 collided before)
        auto swz = [](int row) { return RB % 256 == 0 ? (row & SWZ) : ((row >> 1) & SWZ); };
                                                          // (Padding the row instead, as the 16-bit path does, costs registers here:
                                                          // the 256x384 fp32 kernel went from 19 to 256 spills.)
        // the activation / bias tests are made once (two copies of the loop), not per accumulator quad: a run-time test in the quad loop is a
        // scalar branch per quad and cuts the tile into basic blocks the scheduler cannot overlap (profiles/r03_gemm_epilogue.md)
        auto body32 = [&](auto plain_c, auto in_c) {
        constexpr bool PLAIN = decltype(plain_c)::value;      // no bias, no activation (the decoder's o / down projections)
        constexpr bool IN = decltype(in_c)::value;            // this wave's tile lies inside M x N: no bounds predicates
#pragma unroll
        for (int mt = MT0; mt < MT0 + MTN; ++mt) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[nt][mt][q4 * 4 + e];
                    const int n = n_base + nt * 32 + q4 * 8 + fh * 4;
                    if constexpr (!PLAIN) {
                    if (g.bias && n < g.N) v += *(const f32x4*)(g.bias + n);
                    if (g.act == VITG_ACT_GELU_TANH) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = gelu_tanh_f(v[e]);
                    } else if (g.act == VITG_ACT_GELU_ERF) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = gelu_erf_f(v[e]);
                    }
                    }
                    const int c = nt * 8 + q4 * 2 + fh;
                    *(f32x4*)(stage + frow * RB + ((c ^ swz(frow)) << 4)) = v;
                }
            // same wave reads back what it wrote: LDS ops of one wave complete in order, no barrier needed
#pragma unroll
            for (int i = 0; i < 32 * CH / 64; ++i) {
                const int idx = i * 64 + lane;
                const int row = idx / CH, c = idx % CH;
                f32x4 v = *(const f32x4*)(stage + row * RB + ((c ^ swz(row)) << 4));
                const int m = m_base + (mt - MT0) * 32 + row, n = n_base + c * 4;
                if constexpr (R16) {
                    if (IN || (m < g.M && n < g.N)) {
                        const char* const rb = (const char*)((const T*)g.res16 + (size_t)m_base * g.ldr16 + n_base);
                        const typename T16<T>::vec4 r = *(const typename T16<T>::vec4*)(rb + (unsigned)(((mt - MT0) * 32 + row) * g.ldr16 + c * 4) * 2u);
                        typename T16<T>::vec4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = T16<T>::from_f32(v[e] + T16<T>::to_f32(r[e]));
                        char* const ob16 = (char*)((T*)g.out + (size_t)m_base * g.ldo + n_base);
                        *(typename T16<T>::vec4*)(ob16 + (unsigned)(((mt - MT0) * 32 + row) * g.ldo + c * 4) * 2u) = o;
                    }
                } else
                if (IN || (m < g.M && n < g.N)) {
                    if (g.res) {
                        const int rm = g.res_mod > 0 ? m % g.res_mod : m;
                        v += *(const f32x4*)(g.res + (size_t)rm * g.ldr + n);
                    }
                    // wave-uniform tile origin (64-bit, scalar) + 32-bit in-tile offset: saddr store, no 64-bit VALU per row
                    char* const ob = (char*)((float*)g.out + (size_t)m_base * g.ldo + n_base);
                    const unsigned boff_ = (unsigned)(((mt - MT0) * 32 + row) * g.ldo + c * 4) * 4u;
                    if (VITG_PROFILING_BUILD && (g.flags & 0xC00000)) {
                        u32x4 vv; __builtin_memcpy(&vv, &v, 16);
                        store16_policy(ob, boff_, vv, (g.flags >> 22) & 3);
                    } else *(f32x4*)(ob + boff_) = v;
                }
            }
        }
        };   // body32
        const bool in32 = m_base + MTN * 32 <= g.M && n_base + NT * 32 <= g.N;
        if (g.bias == nullptr && g.act == VITG_ACT_NONE) { if (in32) body32(std::true_type{}, std::true_type{}); else body32(std::true_type{}, std::false_type{}); }
        else body32(std::false_type{}, std::false_type{});
    } else {
        constexpr int NTO = MODE == MODE_SWIGLU ? 1 : NT;  // output 32-column tiles per wave
        constexpr int CH = NTO * 4;                        // 16-B chunks (8 halfs) per row
        constexpr bool POW2 = (CH & (CH - 1)) == 0;
        constexpr int RB = NTO * 64 + (POW2 ? 0 : 16);
        constexpr int SWZ = POW2 ? CH - 1 : 0;
        // rows are RB = 64 / 128 B apart, so 4 / 2 consecutive rows fill the 256 B the LDS serves per cycle: the XOR has to take the row
        // bits ABOVE those.  (With the low bits, as round 1 had it, the 8-B staging writes of rows r, r + 8, r + 16, r + 24 met in the
        // same banks: 4-way conflicts on 48 writes per wave and tile; now 2-way, the minimum for 8-B writes at this row pitch.)
        constexpr int SH = CH == 4 ? 2 : CH == 8 ? 1 : 0;
        const int n_out = MODE == MODE_SWIGLU ? n_base / 2 : n_base;
        const int n_lim = MODE == MODE_SWIGLU ? g.N / 2 : g.N;
        auto body16 = [&](auto act_c, auto in_c) {
        constexpr int ACT = decltype(act_c)::value;           // chosen once per tile (see body32)
        constexpr bool IN = decltype(in_c)::value;            // this wave's tile lies inside M x N: no bounds predicates
#pragma unroll
        for (int mt = MT0; mt < MT0 + MTN; ++mt)
#pragma unroll
            for (int nt = 0; nt < NTO; ++nt)
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    f32x4 v;
                    if constexpr (MODE == MODE_SWIGLU) {
                        if constexpr (NRM) {                   // RMSNorm folded in: gate and up are both scaled by the row's rstd
                            const float rs = nrs[mt - MT0];
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = silu_f(rs * acc[0][mt][q4 * 4 + e]) * (rs * acc[NT - 1][mt][q4 * 4 + e]);
                        } else {
                            const f32x2 lo = silu_mul_f2(f32x2{acc[0][mt][q4 * 4], acc[0][mt][q4 * 4 + 1]}, f32x2{acc[NT - 1][mt][q4 * 4], acc[NT - 1][mt][q4 * 4 + 1]});
                            const f32x2 hi = silu_mul_f2(f32x2{acc[0][mt][q4 * 4 + 2], acc[0][mt][q4 * 4 + 3]}, f32x2{acc[NT - 1][mt][q4 * 4 + 2], acc[NT - 1][mt][q4 * 4 + 3]});
                            v = f32x4{lo[0], lo[1], hi[0], hi[1]};
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = acc[nt][mt][q4 * 4 + e];
                        const int n = n_base + nt * 32 + q4 * 8 + fh * 4;
                        if constexpr (NRM) {
                            f32x4 sn = {0.f, 0.f, 0.f, 0.f};
                            if (g.nrm_colsum && n < g.N) sn = *(const f32x4*)(g.nrm_colsum + n);
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = nrs[mt - MT0] * (v[e] - nmu[mt - MT0] * sn[e]);
                        }
                        if (g.bias && (IN || n < g.N)) v += *(const f32x4*)(g.bias + n);
                        if constexpr (ACT == VITG_ACT_GELU_TANH) {
                            const f32x2 lo = gelu_tanh_f2(f32x2{v[0], v[1]}), hi = gelu_tanh_f2(f32x2{v[2], v[3]});
                            v = f32x4{lo[0], lo[1], hi[0], hi[1]};
                        } else if constexpr (ACT == VITG_ACT_GELU_ERF) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = gelu_erf_f(v[e]);
                        }
                    }
                    typename T16<T>::vec4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = T16<T>::from_f32(v[e]);
                    const int row = (mt - MT0) * 32 + frow, c = nt * 4 + q4;
                    *(typename T16<T>::vec4*)(stage + row * RB + ((c ^ ((row >> SH) & SWZ)) << 4) + fh * 8) = o;
                }
#pragma unroll
        for (int i = 0; i < MTN * 32 * CH / 64; ++i) {
            const int idx = i * 64 + lane;
            const int row = idx / CH, c = idx % CH;
            const u32x4 v = *(const u32x4*)(stage + row * RB + ((c ^ ((row >> SH) & SWZ)) << 4));
            const int m = m_base + row, n = n_out + c * 8;
            if (IN || m < g.M) {
                // wave-uniform tile origin (64-bit, scalar) + 32-bit byte offset: saddr stores, no 64-bit VALU per row.  The non-temporal
                // variant (gemm_flags bit0, measured +-0) and the store-skipping ablation (bit1) exist only in a profiling build.
                char* const ob = (char*)((T*)g.out + (size_t)m_base * g.ldo + n_out);
                const unsigned boff = (unsigned)(row * g.ldo + c * 8) * 2u;
                if (VITG_PROFILING_BUILD && (g.flags & 2)) continue;
                if (IN || n + 8 <= n_lim) {
                    if (VITG_PROFILING_BUILD && (g.flags & 1)) __builtin_nontemporal_store(v, (u32x4*)(ob + boff));
                    else if (VITG_PROFILING_BUILD && (g.flags & 0xC00000)) store16_policy(ob, boff, v, (g.flags >> 22) & 3);
                    else *(u32x4*)(ob + boff) = v;
                } else if (n + 4 <= n_lim) *(u32x2*)(ob + boff) = u32x2{v[0], v[1]};
            }
        }
        };   // body16
        const bool in16 = m_base + MTN * 32 <= g.M && n_base + NT * 32 <= g.N;
        typedef std::integral_constant<int, VITG_ACT_GELU_TANH> AT_;
        typedef std::integral_constant<int, VITG_ACT_NONE> AN_;
        if (MODE != MODE_SWIGLU && g.act == VITG_ACT_GELU_TANH) { if (in16) body16(AT_{}, std::true_type{}); else body16(AT_{}, std::false_type{}); }
        else if (MODE != MODE_SWIGLU && g.act == VITG_ACT_GELU_ERF) body16(std::integral_constant<int, VITG_ACT_GELU_ERF>{}, std::false_type{});
        else if (in16) body16(AN_{}, std::true_type{});
        else body16(AN_{}, std::false_type{});
    }
}

int launch_gemm256(const GemmArgs& g, int dtype, int mode, hipStream_t s);   // gemm256.hip
bool gemm256_applicable(const GemmArgs& g, int mode);
int gemm256_pick_cfg(const GemmArgs& g, int mode);                                // the tile shape auto mode takes for this shape (2 / 6 / 9 / 10)
int gemm256_norm_cols(const GemmArgs& g);                                       // gemm256.hip
int launch_gemm_mi16_streamk(const GemmArgs& g, int dtype, int blocks, hipStream_t s);   // gemm_mi16.hip: raw fp32 partial tiles of a stream-K partition
int launch_gemm_mi16(const GemmArgs& g, int dtype, int mode, int cfg, hipStream_t s);   // gemm_mi16.hip: the same tiles on v_mfma_f32_16x16x32
int launch_gemm_w4(const GemmArgs& g, int dtype, int mode, hipStream_t s);      // gemm_w4.hip
bool gemm_w4_applicable(const GemmArgs& g, int mode);
int launch_gemm_wf(const GemmArgs& g, int dtype, int mode, hipStream_t s);      // gemm_wf.hip
bool gemm_wf_applicable(const GemmArgs& g, int mode);
int launch_gemm_pe(const GemmArgs& g, int dtype, int mode, hipStream_t s);               // gemm_pe.hip
bool gemm_pe_applicable(const GemmArgs& g, int mode);

}  // namespace vitg

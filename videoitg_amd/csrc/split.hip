// Operand-split precision tier (round 6): every MFMA operand that the default path rounds to ONE 16-bit value is carried as a PAIR of fp16 values
//   x = hi + lo,   hi = fp16(x),   lo = fp16(x - hi)            (22 of fp32's 24 mantissa bits; fp16 x fp16 products are exact in the fp32 accumulator)
// so that a forward whose weights are 16-bit exact (the reference's own model.half() weights, /root/reference/eagle/model/builder.py:59) reproduces the
// fp32 CPU reference (north_star: |dscore| <= 1e-3) on weights with massive-activation channels too, where 11-bit operands miss it
// (profiles/r05_heavy_tail.md: 2.9e-3 at 512 x 448^2).  Price: 2 x the MFMA work in the GEMMs, 3 x in attention.
//
// Layout: a split operand of payload width D is ONE row-major fp16 matrix [M, 2 * Dp] (Dp = D padded to 64): columns [0, Dp) hold hi, [Dp, 2 Dp) hold lo,
// pad columns zero.  A GEMM consumes it as an ordinary A operand with K = 2 Dp against the weight laid twice side by side [W | W] (K-concatenation:
// sum_k hi_k w_k + sum_k lo_k w_k in one accumulator) - the main loop of gemm_mi16.hip is untouched; only its 16-bit-output epilogues gained a form that
// writes (hi, lo) pairs (vitg_gemm_split16).  This file holds the other producers of split operands: LayerNorm / RMSNorm of the fp32 stream, RoPE, the
// bilinear token pool - and the attention kernel that consumes split Q / K / V and produces a split O:
//   S = Qh Kh^T + Qh Kl^T + Ql Kh^T          (the lo x lo term is 2^-22 relative: dropped)
//   P = exp2(S - m) in fp32, split into (Ph, Pl);   O += Ph Vh + Ph Vl + Pl Vh
// One kernel per head size, transposed flash attention as csrc/attention.hip's per-tile kernel (same fragment maps), tuned for correctness first.
#include "attn_common.h"
#include <math.h>
#include "../../include/videoitg_hip.h"

#define CHECK_LAUNCH() (hipGetLastError() == hipSuccess ? 0 : VITG_ERR_LAUNCH)

namespace vitg {

// (a macro: vector elements cannot bind to references)
// The empty asm pins x as ONE fp32 value in a register.  Without it hipcc (-ffp-contract=fast) fuses the multiply that produces x into each use separately:
// the stored hi became cvt(fl32(a * b)) - two roundings - while lo was taken against v_fma_mix_f16(a, b, 0) - one rounding of the exact product - and near a
// tie the two "hi" are different neighbours: (hi, lo) then sums to x +- ulp(hi), an fp16-sized error in one element out of ~10^4 (found by
// tools/r06/split_attn_debug.py: errors of exactly 2^-13 .. 2^-17 in single output elements).
#define SPLIT2(x_, hi_, lo_) do { float sx_ = (x_); asm volatile("" : "+v"(sx_)); const f16_t sh_ = (f16_t)sx_; (hi_) = sh_; (lo_) = (f16_t)(sx_ - (float)sh_); } while (0)

// ---- LayerNorm / RMSNorm of fp32 rows -> split rows.  One wave per row, row in registers (D <= 64 * 4 * NV) --------------------------------------
template <bool RMS, int NV>
__global__ __launch_bounds__(256) void norm_split_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ w, const float* __restrict__ bia,
                                                         f16_t* __restrict__ out, int ldo, int lo_off, long M, int D, float eps) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= M) return;
    f32x4 v[NV];
    const float* xr = x + row * ldx;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        v[i] = c < D ? *(const f32x4*)(xr + c) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
        s += RMS ? (v[i][0] * v[i][0] + v[i][1] * v[i][1] + v[i][2] * v[i][2] + v[i][3] * v[i][3]) : (v[i][0] + v[i][1] + v[i][2] + v[i][3]);
    s = wave_sum(s);
    float mean = 0.f, rstd;
    if (RMS) rstd = rsqrtf(s / D + eps);
    else {
        mean = s / D;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * 64 + lane) * 4;
            if (c < D) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float d = v[i][e] - mean; q += d * d; }
            }
        }
        q = wave_sum(q);
        rstd = rsqrtf(q / D + eps);
    }
    f16_t* orow = out + row * ldo;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c >= lo_off) continue;                           // lo_off = padded payload width
        f32x4 r = {0.f, 0.f, 0.f, 0.f};
        if (c < D) {
            r = (v[i] - mean) * rstd * *(const f32x4*)(w + c);
            if (!RMS) r += *(const f32x4*)(bia + c);
        }
        f16x4 h, l;
#pragma unroll
        for (int e = 0; e < 4; ++e) SPLIT2(r[e], h[e], l[e]);
        *(f16x4*)(orow + c) = h;
        *(f16x4*)(orow + lo_off + c) = l;
    }
}

// ---- RoPE in place on split q | k heads: x = hi + lo in fp32, rotate, split again ---------------------------------------------------------------
__global__ void rope_split_kernel(f16_t* __restrict__ qkv, int ld, long lo_off, const float* __restrict__ cs, const float* __restrict__ sn, long N, int heads, int hd) {
    const int half = hd >> 1, per = half >> 2;
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= N * heads * per) return;
    const int j = (int)(gid % per), hh = (int)((gid / per) % heads);
    const long n = gid / ((long)per * heads);
    f16_t* p = qkv + n * ld + (long)hh * hd + j * 4;
    const f16x4 ah = *(f16x4*)p, bh = *(f16x4*)(p + half), al = *(f16x4*)(p + lo_off), bl = *(f16x4*)(p + lo_off + half);
    const f32x4 c = *(const f32x4*)(cs + n * half + j * 4), s = *(const f32x4*)(sn + n * half + j * 4);
    f16x4 rah, ral, rbh, rbl;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float x1 = (float)ah[e] + (float)al[e], x2 = (float)bh[e] + (float)bl[e];
        SPLIT2(x1 * c[e] - x2 * s[e], rah[e], ral[e]);
        SPLIT2(x2 * c[e] + x1 * s[e], rbh[e], rbl[e]);
    }
    *(f16x4*)p = rah; *(f16x4*)(p + half) = rbh;
    *(f16x4*)(p + lo_off) = ral; *(f16x4*)(p + lo_off + half) = rbl;
}

// ---- bilinear token pool (mlp_proj.py:57-68 semantics, the arithmetic of elementwise.hip's pool_kernel) of fp32 features -> split rows ------------
__global__ void pool_split_kernel(const float* __restrict__ x, int ldx, f16_t* __restrict__ out, int ldo, int lo_off, int T, int G, int HW, int D,
                                  const int* __restrict__ cmap, int Pc) {
    const long orow = blockIdx.x;
    const int t = (int)(orow / (HW * HW)), o = (int)(orow % (HW * HW));
    const int oy = o / HW, ox = o % HW;
    const float sc = (float)G / (float)HW;
    float sy = __fsub_rn(__fmul_rn(oy + 0.5f, sc), 0.5f), sx = __fsub_rn(__fmul_rn(ox + 0.5f, sc), 0.5f);
    sy = sy < 0.f ? 0.f : sy;
    sx = sx < 0.f ? 0.f : sx;
    const int y0 = (int)sy, x0 = (int)sx;
    const int y1 = min(y0 + 1, G - 1), x1 = min(x0 + 1, G - 1);
    const float wy1 = sy - y0, wx1 = sx - x0, wy0 = 1.f - wy1, wx0 = 1.f - wx1;
    // dense input: G*G rows per frame; compact input (cmap != null): only the Pc rows the pool reads, at cmap[y*G+x] (as pool_kernel)
    const float* base = x + (long)t * (cmap ? Pc : G * G) * ldx;
    int r00 = y0 * G + x0, r01 = y0 * G + x1, r10 = y1 * G + x0, r11 = y1 * G + x1;
    if (cmap) {
        r00 = cmap[r00]; r01 = cmap[r01]; r10 = cmap[r10]; r11 = cmap[r11];
        if ((r00 | r01 | r10 | r11) < 0) {                   // a token the host did not list: poison the row, never dereference
            for (int c = threadIdx.x; c < 2 * lo_off; c += blockDim.x) out[orow * ldo + c] = (f16_t)__builtin_nanf("");
            return;
        }
    }
    const float* p00 = base + (long)r00 * ldx;
    const float* p01 = base + (long)r01 * ldx;
    const float* p10 = base + (long)r10 * ldx;
    const float* p11 = base + (long)r11 * ldx;
    for (int c = threadIdx.x * 4; c < lo_off; c += blockDim.x * 4) {
        f32x4 r = {0.f, 0.f, 0.f, 0.f};
        if (c < D) {
            if (HW == G) r = *(const f32x4*)(p00 + c);       // no pooling: the token itself (weights 1, 0 - but 0 * inf must not appear)
            else r = wy0 * (wx0 * *(const f32x4*)(p00 + c) + wx1 * *(const f32x4*)(p01 + c)) + wy1 * (wx0 * *(const f32x4*)(p10 + c) + wx1 * *(const f32x4*)(p11 + c));
        }
        f16x4 h, l;
#pragma unroll
        for (int e = 0; e < 4; ++e) SPLIT2(r[e], h[e], l[e]);
        *(f16x4*)(out + orow * ldo + c) = h;
        *(f16x4*)(out + orow * ldo + lo_off + c) = l;
    }
}

// ---- attention on split operands ----------------------------------------------------------------------------------------------------------------
struct AttnSplitArgs {
    const f16_t* q; const f16_t* k; const f16_t* v; f16_t* o;
    long q_lo, k_lo, v_lo, o_lo;     // element offsets of the lo planes
    long q_bs, k_bs, v_bs, o_bs;     // batch strides (elements)
    int q_rs, k_rs, v_rs, o_rs;      // row strides
    int q_hs, k_hs, v_hs, o_hs;      // head strides
    int S, Sk, group, kv_heads, nqb;
    float scale_log2e;
};

constexpr int AS_NW = 4;             // waves per block: 128 queries

template <int D>
__global__ __launch_bounds__(AS_NW * 64, 2) void attn_split_kernel(AttnSplitArgs a) {
    typedef f16_t T;
    typedef f16x8 vec8;
    typedef AttnCfg<D> C;
    constexpr int THREADS = AS_NW * 64, BQ = AS_NW * 32;
    constexpr int KS = C::DK / 16, DT = C::DV / 32;
    constexpr int KT_BYTES = ATT_BK * C::LDK * 2, VT_BYTES = ATT_BK * C::LDV * 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];     // [K hi | K lo | V hi | V lo]
    char* const kh_s = smem;
    char* const kl_s = smem + KT_BYTES;
    char* const vh_s = smem + 2 * KT_BYTES;
    char* const vl_s = smem + 2 * KT_BYTES + VT_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ql = lane & 31, h = lane >> 5;
    const int L = xcd_contiguous_block(blockIdx.x, gridDim.x);
    const int qb = L % a.nqb;
    const int hg = (L / a.nqb) % a.group;
    const int kvh = (L / (a.nqb * a.group)) % a.kv_heads;
    const int b = L / (a.nqb * a.group * a.kv_heads);
    const int head = kvh * a.group + hg;
    const int q0 = qb * BQ + wave * 32;
    const T* Q = a.q + b * a.q_bs + (long)head * a.q_hs;
    const T* K = a.k + b * a.k_bs + (long)kvh * a.k_hs;
    const T* V = a.v + b * a.v_bs + (long)kvh * a.v_hs;

    vec8 qh[KS], qlo[KS];
    {
        const T* qp = Q + (long)min(q0 + ql, a.S - 1) * a.q_rs;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int d0 = ks * 16 + h * 8;
            vec8 z;
#pragma unroll
            for (int e = 0; e < 8; ++e) z[e] = (T)0.f;
            qh[ks] = d0 < D ? *(const vec8*)(qp + d0) : z;
            qlo[ks] = d0 < D ? *(const vec8*)(qp + a.q_lo + d0) : z;
        }
    }
    // staging: thread -> (row, 16-B chunk) of a 64-row tile, the LDS images of attention.hip (K: k_lds_off, V: row-major, hd 128 XOR-swizzled 64-B units)
    constexpr int NCH = ATT_BK * C::CH;
    constexpr int PER = (NCH + THREADS - 1) / THREADS;
    // hd 72 prefetches the next tile's four planes into registers while the current tile computes (48 VGPRs; hd 128 has none to spare: 255 in use).
    // Worth 0.7 % of the tier's step in three same-call alternations (1 432 -> 1 421 ms, gpurun_out/r06_c23): the kernel is bound by its 3x MFMA stream
    // and the softmax beside it, not by staging latency.
#ifndef VITG_SPLIT_PREFETCH
#define VITG_SPLIT_PREFETCH 1
#endif
    constexpr bool PREFETCH = D == 72 && VITG_SPLIT_PREFETCH;
    u32x4 rg[PREFETCH ? PER : 1][4];
    auto load_tile = [&](int kt, u32x4 (&r)[PREFETCH ? PER : 1][4], int i) {
        const int c = tid + i * THREADS;
        const int row = c / C::CH, ch = c % C::CH;
        const long gr = min((long)kt * ATT_BK + row, (long)a.Sk - 1);          // rows past the last key: any finite filler (their scores are masked)
        const T* kp = K + gr * a.k_rs + ch * 8;
        const T* vp = V + gr * a.v_rs + ch * 8;
        r[PREFETCH ? i : 0][0] = *(const u32x4*)kp; r[PREFETCH ? i : 0][1] = *(const u32x4*)(kp + a.k_lo);
        r[PREFETCH ? i : 0][2] = *(const u32x4*)vp; r[PREFETCH ? i : 0][3] = *(const u32x4*)(vp + a.v_lo);
    };
    auto store_tile = [&](const u32x4 (&r)[PREFETCH ? PER : 1][4], int i) {
        const int c = tid + i * THREADS;
        const int row = c / C::CH, ch = c % C::CH;
        const int lk = k_lds_off<D>(row, ch);
        const int vch = D == 128 ? (ch ^ ((row & 3) << 2)) : ch;
        const int lv = row * (C::LDV * 2) + (vch << 4);
        *(u32x4*)(kh_s + lk) = r[PREFETCH ? i : 0][0]; *(u32x4*)(kl_s + lk) = r[PREFETCH ? i : 0][1];
        *(u32x4*)(vh_s + lv) = r[PREFETCH ? i : 0][2]; *(u32x4*)(vl_s + lv) = r[PREFETCH ? i : 0][3];
    };
    auto prefetch = [&](int kt) {                            // PREFETCH: issue the global loads of tile kt (consumed by publish() one iteration later)
#pragma unroll
        for (int i = 0; i < PER; ++i)
            if ((NCH % THREADS == 0) || tid + i * THREADS < NCH) load_tile(kt, rg, i);
    };
    // staging: thread -> (row, 16-B chunk) of a 64-row tile, the LDS images of attention.hip (K: k_lds_off, V: row-major, hd 128 XOR-swizzled 64-B units)
    auto stage = [&](int kt) {
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            if ((NCH % THREADS == 0) || tid + i * THREADS < NCH) {
                if constexpr (!PREFETCH) load_tile(kt, rg, i);
                store_tile(rg, i);
            }
        }
        if constexpr (D == 72) {
            if (tid < ATT_BK) {                                            // K pad chunk (d 72..79) meets Q's zero pad: must be finite -> zero
                const u32x4 z = {0u, 0u, 0u, 0u};
                *(u32x4*)(kh_s + k_lds_off<D>(tid, 9)) = z;
                *(u32x4*)(kl_s + k_lds_off<D>(tid, 9)) = z;
            }
            // V pad columns (72..95) only feed output rows d >= 72, which are never stored (an MFMA output element depends on its own V column only)
        }
    };

    f32x16 o[DT];
#pragma unroll
    for (int i = 0; i < DT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
    float m_run = -1e30f, l_run = 0.f;
    const int nkt = ceil_div(a.Sk, ATT_BK);
    int koff[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) koff[ks] = k_lds_off<D>(ql, ks * 2 + h);
    const int voff = (4 * h + ((lane & 15) >> 2)) * (C::LDV * 2) + ((((lane >> 4) & 1) * 16) + 4 * (lane & 3)) * 2;
    int voff_dt[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) voff_dt[dt] = voff + (D == 128 ? ((dt ^ ((lane & 15) >> 2)) * 64) : dt * 64);

    if constexpr (PREFETCH) prefetch(0);
    for (int kt = 0; kt < nkt; ++kt) {
        __syncthreads();                                  // every wave is done with the previous tile
        stage(kt);
        if constexpr (PREFETCH) { if (kt + 1 < nkt) prefetch(kt + 1); }      // lands while this tile computes
        __syncthreads();
        // ---- S^T = Kh Qh^T + Kh Ql^T + Kl Qh^T : two 32-key tiles (small terms first) -------------------------------------------------
        f32x16 s[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f32x16 acc = {};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const vec8 kl = *(const vec8*)(kl_s + koff[ks] + t * 32 * (C::LDK * 2));
                const vec8 kh = *(const vec8*)(kh_s + koff[ks] + t * 32 * (C::LDK * 2));
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, qh[ks], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qlo[ks], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qh[ks], acc, 0, 0, 0);
            }
            s[t] = acc;
        }
        if (kt * ATT_BK + ATT_BK > a.Sk) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kt * ATT_BK + t * 32 + 8 * (r >> 2) + 4 * h + (r & 3);
                    if (key >= a.Sk) s[t][r] = -INFINITY;
                }
        }
        float mx = s[0][0];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[t][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64)) * a.scale_log2e;
        const float m_new = fmaxf(m_run, mx);
        if (m_new != m_run) {
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            l_run *= alpha;
#pragma unroll
            for (int i = 0; i < DT; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
            m_run = m_new;
        }
        // P is split as 2^12 p (p <= 1: exact scaling, no overflow): an unscaled lo would be subnormal for every p < 2^-3 and carry an ABSOLUTE quantum of
        // 2^-25 - against p ~ 1e-4 (flat attention over 12 860 keys) that is 12 bits, not 22.  O accumulates 2^12 sum p V; the epilogue takes the factor out.
        vec8 ph[4], pl[4];
        float psum = 0.f;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[t][r], a.scale_log2e, -m_run));
                psum += p;
                SPLIT2(p * 4096.0f, ph[t * 2 + (r >> 3)][r & 7], pl[t * 2 + (r >> 3)][r & 7]);
            }
        l_run += psum;
        // ---- O^T += Vh^T Ph^T + Vh^T Pl^T + Vl^T Ph^T ------------------------------------------------------------------------------------
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int krow = (kk >> 1) * 32 + (kk & 1) * 16;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const int off = voff_dt[dt] + krow * (C::LDV * 2);
                const s16x4 h0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(vh_s + off));
                const s16x4 h1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(vh_s + off + 8 * (C::LDV * 2)));
                const s16x4 l0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(vl_s + off));
                const s16x4 l1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(vl_s + off + 8 * (C::LDV * 2)));
                const vec8 vh = bitcast<vec8>(__builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7));
                const vec8 vl = bitcast<vec8>(__builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7));
                f32x16 acc = o[dt];
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, ph[kk], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, pl[kk], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, ph[kk], acc, 0, 0, 0);
                o[dt] = acc;
            }
        }
    }
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = (1.0f / l_tot) * (1.0f / 4096.0f);
    const int qrow = q0 + ql;
    if (qrow < a.S) {
        T* op = a.o + b * a.o_bs + (long)head * a.o_hs + (long)qrow * a.o_rs;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int d = dt * 32 + 8 * g4 + 4 * h;
                if (d < D) {
                    f16x4 wh, wl;
#pragma unroll
                    for (int e = 0; e < 4; ++e) SPLIT2(o[dt][g4 * 4 + e] * inv, wh[e], wl[e]);
                    *(f16x4*)(op + d) = wh;
                    *(f16x4*)(op + a.o_lo + d) = wl;
                }
            }
    }
}

template <int D>
static int launch_attn_split(AttnSplitArgs a, int batch, int heads, hipStream_t s) {
    typedef AttnCfg<D> C;
    a.nqb = ceil_div(a.S, AS_NW * 32);
    const size_t lds = 2 * (ATT_BK * C::LDK * 2 + ATT_BK * C::LDV * 2);
    static std::atomic<unsigned> done_mask{0};
    const int cur_dev = current_device();
    if (!device_done(done_mask, cur_dev)) {
        (void)hipFuncSetAttribute((const void*)attn_split_kernel<D>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        mark_device_done(done_mask, cur_dev);
    }
    hipLaunchKernelGGL((attn_split_kernel<D>), dim3(a.nqb * heads * batch), dim3(AS_NW * 64), lds, s, a);
    return CHECK_LAUNCH();
}

}  // namespace vitg

using namespace vitg;

extern "C" int vitg_norm_split(const float* x, int ldx, const float* w, const float* b, void* out, int ldo, int lo_off, long M, int D, float eps,
                               void* stream) {
    if (M <= 0) return 0;
    if (!x || !w || !out) return VITG_ERR_ARG;
    if (D % 4 || ldo % 4 || ldx % 4 || lo_off % 4) return VITG_ERR_ALIGN;
    if (((uintptr_t)x & 15) || ((uintptr_t)w & 15) || (b && ((uintptr_t)b & 15)) || ((uintptr_t)out & 7)) return VITG_ERR_ALIGN;
    if (lo_off < D || ldo < 2 * lo_off) return VITG_ERR_SHAPE;
    const int need = ceil_div(lo_off, 256);
    dim3 blk(256), grid((unsigned)((M + 3) / 4));
    hipStream_t s = (hipStream_t)stream;
#define NS(NV) do { if (b) hipLaunchKernelGGL((norm_split_kernel<false, NV>), grid, blk, 0, s, x, ldx, w, b, (f16_t*)out, ldo, lo_off, M, D, eps); \
                    else hipLaunchKernelGGL((norm_split_kernel<true, NV>), grid, blk, 0, s, x, ldx, w, b, (f16_t*)out, ldo, lo_off, M, D, eps); } while (0)
    if (need <= 1) NS(1); else if (need <= 2) NS(2); else if (need <= 5) NS(5); else if (need <= 8) NS(8); else if (need <= 16) NS(16); else return VITG_ERR_SHAPE;
#undef NS
    return CHECK_LAUNCH();
}

extern "C" int vitg_rope_split(void* qkv, int ld, long lo_off, const float* cos_t, const float* sin_t, long N, int heads, int head_dim, void* stream) {
    if (N <= 0 || heads <= 0) return 0;
    if (!qkv || !cos_t || !sin_t) return VITG_ERR_ARG;
    if (head_dim % 8 || ld % 4 || lo_off % 4) return VITG_ERR_ALIGN;
    const long total = N * heads * (head_dim / 8);
    hipLaunchKernelGGL(rope_split_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (f16_t*)qkv, ld, lo_off, cos_t, sin_t, N,
                       heads, head_dim);
    return CHECK_LAUNCH();
}

extern "C" int vitg_bilinear_pool_split(const float* x, int ldx, void* out, int ldo, int lo_off, int T, int G, int HW, int D, const int* cmap, int Pc,
                                        void* stream) {
    if (T <= 0) return 0;
    if (!x || !out) return VITG_ERR_ARG;
    if (cmap && (HW == G || Pc <= 0)) return VITG_ERR_ARG;
    if (D % 4 || ldo % 4 || ldx % 4 || lo_off % 4) return VITG_ERR_ALIGN;
    if (HW <= 0 || HW > G || lo_off < D || ldo < 2 * lo_off) return VITG_ERR_SHAPE;
    hipLaunchKernelGGL(pool_split_kernel, dim3((unsigned)((long)T * HW * HW)), dim3(256), 0, (hipStream_t)stream, x, ldx, (f16_t*)out, ldo, lo_off, T, G, HW, D, cmap, Pc);
    return CHECK_LAUNCH();
}

extern "C" int vitg_attention_split(const void* q, const void* k, const void* v, void* o, long q_lo, long k_lo, long v_lo, long o_lo, int batch, int heads,
                                    int kv_heads, int S, int Sk, int head_dim, long q_bs, long k_bs, long v_bs, long o_bs, int q_rs, int k_rs, int v_rs,
                                    int o_rs, int q_hs, int k_hs, int v_hs, int o_hs, float scale, void* stream) {
    if (batch <= 0 || heads <= 0 || S <= 0) return 0;
    if (!q || !k || !v || !o) return VITG_ERR_ARG;
    if (Sk <= 0 || kv_heads <= 0 || heads % kv_heads) return VITG_ERR_SHAPE;
    if (head_dim != 72 && head_dim != 128) return VITG_ERR_SHAPE;
    if ((o_rs | o_hs) % 4 || (o_lo % 4)) return VITG_ERR_ALIGN;
    if ((q_rs | k_rs | v_rs | q_hs | k_hs | v_hs) % 8 || (q_bs | k_bs | v_bs | q_lo | k_lo | v_lo) % 8) return VITG_ERR_ALIGN;
    if ((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) & 15) || ((uintptr_t)o & 7)) return VITG_ERR_ALIGN;      // 16-B fragment loads, 8-B output stores
    AttnSplitArgs a;
    a.q = (const f16_t*)q; a.k = (const f16_t*)k; a.v = (const f16_t*)v; a.o = (f16_t*)o;
    a.q_lo = q_lo; a.k_lo = k_lo; a.v_lo = v_lo; a.o_lo = o_lo;
    a.q_bs = q_bs; a.k_bs = k_bs; a.v_bs = v_bs; a.o_bs = o_bs;
    a.q_rs = q_rs; a.k_rs = k_rs; a.v_rs = v_rs; a.o_rs = o_rs;
    a.q_hs = q_hs; a.k_hs = k_hs; a.v_hs = v_hs; a.o_hs = o_hs;
    a.S = S; a.Sk = Sk; a.group = heads / kv_heads; a.kv_heads = kv_heads; a.nqb = 0;
    a.scale_log2e = scale * 1.4426950408889634f;
    return head_dim == 72 ? launch_attn_split<72>(a, batch, heads, (hipStream_t)stream) : launch_attn_split<128>(a, batch, heads, (hipStream_t)stream);
}

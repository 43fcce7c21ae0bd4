// Software-pipelined non-causal flash attention forward for gfx950 (same math, layouts and transposed formulation as
// attention.hip; read that file's header first).
//
// Why a second loop structure.  In the per-tile kernel every wave runs  QK^T (MFMA) -> softmax (VALU) -> PV (MFMA)  and then
// meets the other waves of the block at the tile barrier, so all waves of a block are in the same phase at the same time:
// the matrix pipe idles while everybody exponentiates and the VALU idles while everybody multiplies.  PMC on the ViT shape:
// MFMA busy 43 %, VALU busy ~55 %, sum ~100 % - the two pipes take turns.  Removing 40 % of the VALU instructions, or halving
// the LDS traffic per MFMA (64 queries per wave), changed nothing (profiles/r01_attention.md), which rules out VALU issue and
// LDS bandwidth as the limit and leaves the missing overlap.
//
// Here one barrier interval t holds work of THREE tiles that is mutually independent inside a wave:
//      MFMA stream :  S(t+1) = K(t+1) Q^T           and          O += V(t-1)^T P(t-1)^T
//      VALU stream :  P(t) = exp2(S(t) * c - m)     (row max, exponent, row sum, 16-bit pack)
// and the two streams are interleaved instruction by instruction (sched_group_barrier: 1 MFMA, its LDS fragment reads, a few
// VALU ops, repeat), so each wave keeps both pipes busy by itself, whatever phase its neighbours are in.
//
// Consequences:
//   * K is staged two tiles ahead and V one tile behind: interval t reads K(t+1) and V(t-1) while the DMA engine fills
//     K(t+2) and V(t).  Two K buffers + two V buffers = the same LDS footprint as the per-tile kernel.
//   * The rescale of O by alpha = exp2(m_old - m_new) has to follow the PV(t-1) MFMAs of its interval (P(t-1) is relative to
//     m_old), so it sits after the interleaved block.  It is also made RARE: the running max is only moved when some query
//     of the wave saw its max grow by more than 2^8 (FA-4 style lazy rescale).  Until then P is computed against the stale
//     max, i.e. P <= 256 - exact in real arithmetic (numerator and denominator carry the same factor), harmless in fp32
//     accumulators and inside fp16/bf16 range for the packed P.  With the eager update a 32-query wave rescaled on ~95 % of
//     the tiles (any one query finding a new max is enough).
//   * Registers: two score tiles + two packed P tiles live at once -> ~200 (hd 72) / ~230 (hd 128) VGPRs, 2 waves per SIMD.
#include "attn_common.h"
#include "../../include/videoitg_hip.h"
#include <type_traits>

#ifndef VITG_PROFILING_BUILD
#define VITG_PROFILING_BUILD 0     // the ablation switches below (tools/attn_ablate_sp.py) are compiled in only with -DVITG_PROFILING_BUILD=1:
#endif                             // a run-time test inside the tile loop is a scalar branch per fragment read
static constexpr bool PROF_BUILD = VITG_PROFILING_BUILD != 0;

namespace vitg {

constexpr float LAZY_RESCALE_LOG2 = 8.0f;

// PV16 (round 3, hd 128; OPT-IN - measured slower than the 32x32x16 form, see attention_sp_dispatch): O^T += V^T P on v_mfma_f32_16x16x32 (eight 16-row d-tiles x 2 key blocks x 2 query tiles = 32 MFMAs of 16 cycles instead
// of 16 of 32: same pipe cycles, the cheaper shape under the power cap - tools/ubench/mfma_shape_power.hip).  As in attention_ab.hip: P is re-laid
// for the 16x16x32 B operand in registers (v_permlane32_swap + v_permlane16_swap per VGPR pair; here spread over the first four slices of the
// interval that CONSUMES it), the V^T fragment of lane group g reads keys 32 kb + 16 (g >> 1) + 4 (g & 1) + {0..3, 8..11}.  V keeps its 256-B rows;
// its chunk swizzle gains row bit 2 (the two lane groups of a half-wave read rows r..r+3 and r+4..r+7, which the round-2 swizzle put on the same
// banks).  One PV "step" of the interleaved stream = one V^T fragment + its two MFMAs, so the slice structure is unchanged.
template <typename T, int D, int NW, bool PV16 = false>
__global__ __launch_bounds__(NW * 64, 2) void attn_sp_kernel(AttnArgs a) {
    adopt_device_key_count(a);
    static_assert(!PV16 || D == 128, "PV16 is written for head dim 128");
    typedef typename T16<T>::vec8 vec8;
    typedef AttnCfg<D> C;
    constexpr int KS = C::DK / 16;           // QK^T k-steps per 32-key half
    constexpr int DT = PV16 ? C::DV / 16 : C::DV / 32;   // PV d-tiles (of 16 / 32 rows)
    constexpr int KT_BYTES = ATT_BK * C::LDK * 2, VT_BYTES = ATT_BK * C::LDV * 2;
    constexpr int V_BASE = 2 * KT_BYTES;     // LDS: [K0][K1][V0][V1]
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ql = lane & 31, h = lane >> 5;
    const int L = xcd_contiguous_block(blockIdx.x, gridDim.x);
    const int qb = L % a.nqb;
    const int hg = (L / a.nqb) % a.group;
    const int kvh = (L / (a.nqb * a.group)) % a.kv_heads;
    const int b = L / (a.nqb * a.group * a.kv_heads);
    const int head = kvh * a.group + hg;
    const int q0 = qb * (NW * 32) + wave * 32;

    const T* Q = (const T*)a.q + b * a.q_bs + (long)head * a.q_hs;
    const T* K = (const T*)a.k + b * a.k_bs + (long)kvh * a.k_hs;
    const T* V = (const T*)a.v + b * a.v_bs + (long)kvh * a.v_hs;

    // ---- Q fragments (B operand): lane (q = ql, h) holds d = 16 ks + 8 h .. +7 ------------------------------------------
    vec8 qf[KS];
    {
        const T* qp = Q + (long)min(q0 + ql, a.S - 1) * a.q_rs;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int d0 = ks * 16 + h * 8;
            vec8 t;
            if (d0 < D) t = *(const vec8*)(qp + d0);
            else {
#pragma unroll
                for (int e = 0; e < 8; ++e) t[e] = (T)0.f;
            }
            qf[ks] = t;
        }
    }

    // ---- LDS-DMA staging: one global_load_lds_dwordx4 fills 1 KiB lane-linearly; the source address carries the layout ---
    constexpr int K_CPR = C::LDK * 2 / 16, V_CPR = C::LDV * 2 / 16;       // 16-B slots per row = DMA instructions per tile
    constexpr int K_PW = (K_CPR + NW - 1) / NW, V_PW = (V_CPR + NW - 1) / NW;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    auto dma_src = [&](int ii, int cpr, int rs, bool is_k, int row_lim) -> unsigned {
        const int j = ii * 64 + lane;
        int row = j / cpr, c = j % cpr;
        if (D == 128) c = is_k ? (c ^ (row & 15)) : (c ^ ((row & 3) << 2) ^ (PV16 ? (((row >> 2) & 1) << 1) : 0));
        else c = min(c, C::CH - 1);          // pad slots re-read the row's last data chunk (finite filler)
        row = min(row, row_lim);
        return (unsigned)(row * rs + c * 8) * 2u;
    };
    unsigned dk_off[K_PW], dv_off[V_PW];
#pragma unroll
    for (int i = 0; i < K_PW; ++i) dk_off[i] = dma_src(wave_u + NW * i, K_CPR, a.k_rs, true, ATT_BK - 1);
#pragma unroll
    for (int i = 0; i < V_PW; ++i) dv_off[i] = dma_src(wave_u + NW * i, V_CPR, a.v_rs, false, ATT_BK - 1);
#ifndef VITG_SP_BUF_DMA
#define VITG_SP_BUF_DMA 1      // inside the step: 2.404 -> 2.298 ms per launch (+4.6 %), step 631.1 -> 628.3 ms, three alternations (gpurun_out/r05_spbuf)
#endif
    // hd 128, VITG_SP_BUF_DMA: staging through buffer descriptors whose extent ends with the last key row (attn_common.h: lds_dma16_buf) - no row clamp, no
    // tail branch, three instructions per piece at the head of the interval instead of ~15 (the one-wave-per-SIMD kernel's staging, csrc/attention_w4.hip)
    constexpr bool BUF_DMA = VITG_SP_BUF_DMA && VITG_ASM_DMA && D == 128;
    u32x4 k_rsrc = {}, v_rsrc = {};
    int dk_offb[K_PW], dv_offb[V_PW];
    unsigned lds_w = 0;
    int k_tile_bytes = 0, v_tile_bytes = 0;
    if constexpr (BUF_DMA) {
        k_rsrc = lds_dma_rsrc(K, ((long)(a.Sk - 1) * a.k_rs + D) * 2);
        v_rsrc = lds_dma_rsrc(V, ((long)(a.Sk - 1) * a.v_rs + D) * 2);
        auto src = [&](int ii, int cpr, int rs, bool is_k) -> int {
            const int j = ii * 64 + lane;
            const int row = j / cpr;
            int c = j % cpr;
            c = is_k ? (c ^ (row & 15)) : (c ^ ((row & 3) << 2) ^ (PV16 ? (((row >> 2) & 1) << 1) : 0));
            return (row * rs + c * 8) * 2;
        };
#pragma unroll
        for (int i = 0; i < K_PW; ++i) dk_offb[i] = src(wave_u + NW * i, K_CPR, a.k_rs, true);
#pragma unroll
        for (int i = 0; i < V_PW; ++i) dv_offb[i] = src(wave_u + NW * i, V_CPR, a.v_rs, false);
        lds_w = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)smem) + wave_u * 1024u;
        k_tile_bytes = ATT_BK * a.k_rs * 2;
        v_tile_bytes = ATT_BK * a.v_rs * 2;
    }
    auto issue_k = [&](int kt, int buf) {
        if constexpr (BUF_DMA) {
#pragma unroll
            for (int i = 0; i < K_PW; ++i)
                if (wave_u + NW * i < K_CPR) lds_dma16_buf(k_rsrc, lds_w + buf * KT_BYTES + NW * i * 1024, dk_offb[i] + kt * k_tile_bytes);
            return;
        }
        char* kb = smem + buf * KT_BYTES;
        const char* Kt = (const char*)(K + (long)kt * ATT_BK * a.k_rs);
        const bool tail = kt * ATT_BK + ATT_BK > a.Sk;          // wave-uniform: clamp rows past the last key
#pragma unroll
        for (int i = 0; i < K_PW; ++i) {
            const int ii = wave_u + NW * i;
            if (ii < K_CPR) {
                const unsigned off = tail ? dma_src(ii, K_CPR, a.k_rs, true, a.Sk - 1 - kt * ATT_BK) : dk_off[i];
                lds_dma16(Kt, off, kb + ii * 1024);
            }
        }
    };
    auto issue_v = [&](int kt, int buf) {
        if constexpr (BUF_DMA) {
#pragma unroll
            for (int i = 0; i < V_PW; ++i)
                if (wave_u + NW * i < V_CPR) lds_dma16_buf(v_rsrc, lds_w + V_BASE + buf * VT_BYTES + NW * i * 1024, dv_offb[i] + kt * v_tile_bytes);
            return;
        }
        char* vb = smem + V_BASE + buf * VT_BYTES;
        const char* Vt = (const char*)(V + (long)kt * ATT_BK * a.v_rs);
        const bool tail = kt * ATT_BK + ATT_BK > a.Sk;
#pragma unroll
        for (int i = 0; i < V_PW; ++i) {
            const int ii = wave_u + NW * i;
            if (ii < V_CPR) {
                const unsigned off = tail ? dma_src(ii, V_CPR, a.v_rs, false, a.Sk - 1 - kt * ATT_BK) : dv_off[i];
                lds_dma16(Vt, off, vb + ii * 1024);
            }
        }
    };

    // lane-dependent LDS read bases; inside the loop every fragment address is one of these + an immediate
    const char* kbase[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) kbase[ks] = smem + k_lds_off<D>(ql, ks * 2 + h);      // + t*32 rows: row & 15 unchanged
    const int g16 = lane >> 4;
    const int voff = (4 * h + ((lane & 15) >> 2)) * (C::LDV * 2) + ((((lane >> 4) & 1) * 16) + 4 * (lane & 3)) * 2;
    const char* vbase[DT];
    if constexpr (PV16) {
        // lane (d16 = lane & 15, g): rows 16 (g >> 1) + 4 (g & 1) + ((lane & 15) >> 2) (+ 32 kb, + 8 for the second read); 16-B chunk of d-tile dt:
        // (2 dt + ((lane & 3) >> 1)) ^ swizzle(row), swizzle = ((row & 3) << 2) ^ (((row >> 2) & 1) << 1) - a function of the lane only
        const int vrow = 16 * (g16 >> 1) + 4 * (g16 & 1) + ((lane & 15) >> 2);
        const int xs = (((lane & 15) >> 2) << 2) ^ ((g16 & 1) << 1);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
            vbase[dt] = smem + V_BASE + vrow * (C::LDV * 2) + (((2 * dt + ((lane & 3) >> 1)) ^ xs) << 4) + (lane & 1) * 8;
    } else {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) vbase[dt] = smem + V_BASE + voff + (D == 128 ? ((dt ^ ((lane & 15) >> 2)) * 64) : dt * 64);
    }

    // accumulators: 32x32 layout o[dt][r] = O[q][32 dt + 8 (r >> 2) + 4 h + (r & 3)], or (PV16) o16[dt][qt][r] = O[16 qt + (lane & 15)][16 dt + 4 (lane >> 4) + r]
    f32x16 o[PV16 ? 1 : DT];
    f32x4 o16[PV16 ? DT : 1][2];
#pragma unroll
    for (int i = 0; i < (PV16 ? 1 : DT); ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
#pragma unroll
    for (int i = 0; i < (PV16 ? DT : 1); ++i) { o16[i][0] = f32x4{0.f, 0.f, 0.f, 0.f}; o16[i][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    float m_run = -1e30f, l_run = 0.f;
    const int nkt = ceil_div(a.Sk, ATT_BK);

    // S = K Q^T for one 64-key tile out of K buffer `kbuf`
    auto qk_mfma = [&](int kbuf, f32x16 (&s)[2]) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const vec8 kf = *(const vec8*)(kbase[ks] + (kbuf * KT_BYTES + t * 32 * (C::LDK * 2)));
                s[t] = ks == 0 ? T16<T>::mfma32(kf, qf[0], f32x16{}) : T16<T>::mfma32(kf, qf[ks], s[t]);
            }
    };
    // PV16: VGPR index i (0..3) of the register pair (p[2kb], p[2kb+1]) -> B operands of query tile 0 / 1 (rows [(2kb,h0) (2kb,h1) (2kb+1,h0) (2kb+1,h1)])
    auto swap_p = [&](vec8 (&p)[4], int kb, int i) {
        u32x4 X = bitcast<u32x4>(p[2 * kb]), Y = bitcast<u32x4>(p[2 * kb + 1]);
        const u32x2 t = __builtin_amdgcn_permlane32_swap(X[i], Y[i], false, false);
        const u32x2 u = __builtin_amdgcn_permlane16_swap(t[0], t[1], false, false);
        X[i] = u[0]; Y[i] = u[1];
        p[2 * kb] = bitcast<vec8>(X); p[2 * kb + 1] = bitcast<vec8>(Y);
    };
    auto v16_frag = [&](int vbuf, int kb, int dt) -> vec8 {
        const char* a0 = vbase[dt] + (vbuf * VT_BYTES + kb * 32 * (C::LDV * 2));
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(a0));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(a0 + 8 * (C::LDV * 2)));
        return bitcast<vec8>(__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    };
    // O += V^T P^T for one tile out of V buffer `vbuf`; k-step kk covers keys (kk>>1)*32 + (kk&1)*16 + {8*(j>>2) + 4*h + (j&3)}
    // (the drain after the last interval; PV16: p is still in the softmax layout here and is re-laid first)
    auto pv_mfma = [&](int vbuf, vec8 (&p)[4]) {
        if constexpr (PV16) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int i = 0; i < 4; ++i) swap_p(p, kb, i);
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    const vec8 vf = v16_frag(vbuf, kb, dt);
                    o16[dt][0] = T16<T>::mfma16(vf, p[2 * kb], o16[dt][0]);
                    o16[dt][1] = T16<T>::mfma16(vf, p[2 * kb + 1], o16[dt][1]);
                }
        } else {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const char* a0 = vbase[dt] + (vbuf * VT_BYTES + ((kk >> 1) * 32 + (kk & 1) * 16) * (C::LDV * 2));
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(a0));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(a0 + 8 * (C::LDV * 2)));
                o[dt] = T16<T>::mfma32(bitcast<vec8>(__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7)), p[kk], o[dt]);
            }
        }
    };

    // One barrier interval.  PAR = t & 1.  s_cur = S(t) (in), s_nxt = S(t+1) (out), p_cur = P(t) (out), p_prv = P(t-1) (in).
    auto interval = [&](int t, auto par_c, auto qk_c, auto pv_c, f32x16 (&s_cur)[2], f32x16 (&s_nxt)[2], vec8 (&p_cur)[4],
                        vec8 (&p_prv)[4]) {
        constexpr int PAR = decltype(par_c)::value;
        constexpr bool HAS_QK = decltype(qk_c)::value, HAS_PV = decltype(pv_c)::value;
        if (!(PROF_BUILD && (a.ablate & 8))) {                             // (ablate: profiling only, tools/attn_ablate.py)
            issue_v(t, PAR);                               // V(t)   -> read in interval t+1
            if (t + 2 < nkt) issue_k(t + 2, PAR);          // K(t+2) -> read in interval t+1   (K(t) in that buffer is dead)
        }
        // only the interval without a next tile (HAS_QK false) can hold the last, possibly partial, tile
        if (!HAS_QK && t * ATT_BK + ATT_BK > a.Sk) {       // partial tile: lane (q, h) s[tt][r] is key t*64 + tt*32 + 8*(r>>2) + 4*h + (r&3)
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (t * ATT_BK + tt * 32 + 8 * (r >> 2) + 4 * h + (r & 3) >= a.Sk) s_cur[tt][r] = -INFINITY;
        }
        // ---- one basic block from here to the rescale test ------------------------------------------------------------
        // MFMA i of the interval (QK^T steps first, then PV steps) is followed by slice i of the softmax VALU stream; a
        // sched_barrier after every slice keeps the machine scheduler from re-bunching either stream.  The LDS fragment for
        // MFMA i+2 is requested in slice i (three fragment registers in rotation).
        constexpr int NQK = HAS_QK ? 2 * KS : 0, NPV = HAS_PV ? (PV16 ? 2 * DT : 4 * DT) : 0, NM = NQK + NPV;
        static_assert(!PV16 || !HAS_PV || NQK == 0 || NQK >= 4, "the P re-layout of PV16 rides the first four slices, ahead of the first PV step");
        auto frag = [&](int i) -> vec8 {
            if (i < NQK) return *(const vec8*)(kbase[i % KS] + ((PAR ^ 1) * KT_BYTES + (i / KS) * 32 * (C::LDK * 2)));
            const int j = i - NQK, kk = j / DT, dt = j % DT;
            if constexpr (PV16) return v16_frag(PAR ^ 1, kk, dt);          // kk = key block (0 / 1), one fragment feeds both query tiles
            const char* a0 = vbase[dt] + ((PAR ^ 1) * VT_BYTES + ((kk >> 1) * 32 + (kk & 1) * 16) * (C::LDV * 2));
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(a0));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(a0 + 8 * (C::LDV * 2)));
            return bitcast<vec8>(__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
        };
        auto mfma_step = [&](int i, vec8 f) {
            if (i < NQK) {
                const int tt = i / KS, ks = i % KS;
                s_nxt[tt] = ks == 0 ? T16<T>::mfma32(f, qf[0], f32x16{}) : T16<T>::mfma32(f, qf[ks], s_nxt[tt]);
            } else {
                const int j = i - NQK;
                if constexpr (PV16) {
                    o16[j % DT][0] = T16<T>::mfma16(f, p_prv[2 * (j / DT)], o16[j % DT][0]);
                    o16[j % DT][1] = T16<T>::mfma16(f, p_prv[2 * (j / DT) + 1], o16[j % DT][1]);
                } else
                o[j % DT] = T16<T>::mfma32(f, p_prv[j / DT], o[j % DT]);
            }
        };
        float mx, m_new = m_run, m_use = m_run;
        uint64_t grew = 0;
        f32x2 psum2 = {0.f, 0.f};
        const f32x2 sc2 = {a.scale_log2e, a.scale_log2e};
        auto row_max = [&](int half) {                      // slices 0 and 1
            if (half == 0) mx = s_cur[0][0];
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s_cur[half][r]);
            if (half == 1) {
                const u32x2 sw = __builtin_amdgcn_permlane32_swap(bitcast<uint32_t>(mx), bitcast<uint32_t>(mx), false, false);
                mx = fmaxf(bitcast<float>(sw[0]), bitcast<float>(sw[1])) * a.scale_log2e;      // both halves of the query; scale > 0
                m_new = fmaxf(m_run, mx);
                grew = __builtin_amdgcn_ballot_w64(m_new > m_run + LAZY_RESCALE_LOG2);           // wave-uniform
                m_use = grew ? m_new : m_run;
                // opaque: otherwise the exponent stream is speculated on m_run ahead of the row max (out of the slices) and
                // recomputed in the rare path
                asm volatile("" : "+v"(m_use));
            }
        };
        auto exp_pair = [&](int e) {                        // scores 2e, 2e+1 of the tile (e = 0..15)
            const int tt = e >> 3, r = (e & 7) * 2;
            const f32x2 nm2 = {-m_use, -m_use};
            f32x2 v = {s_cur[tt][r], s_cur[tt][r + 1]};
            v = __builtin_elementwise_fma(v, sc2, nm2);
            v = f32x2{__builtin_amdgcn_exp2f(v[0]), __builtin_amdgcn_exp2f(v[1])};
            psum2 += v;
            p_cur[tt * 2 + (r >> 3)][r & 7] = T16<T>::from_f32(v[0]);
            p_cur[tt * 2 + (r >> 3)][(r & 7) + 1] = T16<T>::from_f32(v[1]);
            // opaque uses: without them the optimiser sinks the whole stream below the rescale branch (its results are only
            // consumed after it), i.e. out of the slices again
            asm volatile("" : "+v"(psum2));
            if ((e & 3) == 3) asm volatile("" : "+v"(p_cur[tt * 2 + (r >> 3)]));
        };
        if constexpr (NM == 0) {
            row_max(0);
            row_max(1);
#pragma unroll
            for (int e = 0; e < 16; ++e) exp_pair(e);
        } else {
            if constexpr (PV16 && HAS_PV && NQK == 0) {                  // (last interval of a sequence: no QK^T steps to hide the re-layout behind)
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int q = 0; q < 4; ++q) swap_p(p_prv, kb, q);
            }
            vec8 fr[3];
            fr[0] = frag(0);
            fr[1] = frag(1);
#pragma unroll
            for (int i = 0; i < NM; ++i) {
                if constexpr (PV16 && HAS_PV && NQK > 0) {
                    if (i < 4) { swap_p(p_prv, i >> 1, (i & 1) * 2); swap_p(p_prv, i >> 1, (i & 1) * 2 + 1); }
                }
                if (i + 2 < NM && !(PROF_BUILD && (a.ablate & 16) && i + 2 > NQK)) fr[(i + 2) % 3] = frag(i + 2);      // (ablate bit4: profiling - V^T fragments not re-read)
                mfma_step(i, fr[i % 3]);
                if (i < 2) row_max(i);
                else {
                    const int lo = (i - 2) * 16 / (NM - 2), hi = (i - 1) * 16 / (NM - 2);
#pragma unroll
                    for (int e = lo; e < hi; ++e) exp_pair(e);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // The rescale test below uses an opaque copy of `grew`: if the compiler can prove it is the same condition it threads
        // the branch upwards and clones the exponent stream into the rare path, which splits the interleaved block.
        uint64_t grew_late = grew;
        asm volatile("" : "+s"(grew_late));
        if (grew_late) {                                   // after the PV(t-1) MFMAs: P(t-1) was relative to the old max
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            l_run *= alpha;
            if constexpr (PV16) {
                const u32x2 aa = __builtin_amdgcn_permlane16_swap(bitcast<uint32_t>(alpha), bitcast<uint32_t>(alpha), false, false);
                const float a0 = bitcast<float>(aa[0]), a1 = bitcast<float>(aa[1]);
#pragma unroll
                for (int i = 0; i < DT; ++i) { o16[i][0] *= a0; o16[i][1] *= a1; }
            } else {
#pragma unroll
            for (int i = 0; i < DT; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
            }
            m_run = m_new;
        }
        l_run += psum2[0] + psum2[1];
        { lds_dma_drain(); __syncthreads(); }                                   // drains vmcnt too: K(t+2) / V(t) have landed, K(t+1)/V(t-1) reads are done
    };

    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 1> I1;
    typedef std::true_type Y;
    typedef std::false_type N;

    // ---- prologue: K(0), K(1) in flight; S(0) computed without overlap ----------------------------------------------------
    issue_k(0, 0);
    if (nkt > 1) issue_k(1, 1);
    { lds_dma_drain(); __syncthreads(); }
    f32x16 sA[2], sB[2];
    vec8 pA[4], pB[4];
    qk_mfma(0, sA);
    { lds_dma_drain(); __syncthreads(); }                                       // every wave is done with K(0) before interval 0 refills buffer 0
    // Even intervals: S(t) in sA, P(t) -> pA; odd intervals: sB / pB.  The loop body is a fixed (odd, even) pair so that no
    // register set changes role across the back edge.
    if (nkt == 1) {
        interval(0, I0{}, N{}, N{}, sA, sB, pA, pB);
        pv_mfma(0, pA);
    } else {
        interval(0, I0{}, Y{}, N{}, sA, sB, pA, pB);
        int t = 1;
        for (; t + 2 < nkt; t += 2) {
            interval(t, I1{}, Y{}, Y{}, sB, sA, pB, pA);
            interval(t + 1, I0{}, Y{}, Y{}, sA, sB, pA, pB);
        }
        if (t + 1 < nkt) {                                 // two intervals left: t (odd), t + 1 = nkt - 1 (even)
            interval(t, I1{}, Y{}, Y{}, sB, sA, pB, pA);
            interval(t + 1, I0{}, N{}, Y{}, sA, sB, pA, pB);
            pv_mfma(0, pA);                                // drain: PV of the last tile (V(nkt-1) landed at the last barrier)
        } else {                                           // one left: t = nkt - 1 (odd)
            interval(t, I1{}, N{}, Y{}, sB, sA, pB, pA);
            pv_mfma(1, pB);
        }
    }

    // ---- epilogue: lane (q, h) holds O[q][dt*32 + 8*(r>>2) + 4*h + (r&3)] -------------------------------------------------
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    if constexpr (PV16) {
        const u32x2 ii = __builtin_amdgcn_permlane16_swap(bitcast<uint32_t>(inv), bitcast<uint32_t>(inv), false, false);
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            const float iq = bitcast<float>(ii[qt]);
            const int qrow = q0 + 16 * qt + (lane & 15);
            if (qrow < a.S) {
                T* op = (T*)a.o + b * a.o_bs + (long)head * a.o_hs + (long)qrow * a.o_rs;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    typename T16<T>::vec4 w;
#pragma unroll
                    for (int e = 0; e < 4; ++e) w[e] = T16<T>::from_f32(o16[dt][qt][e] * iq);
                    *(typename T16<T>::vec4*)(op + dt * 16 + 4 * g16) = w;
                }
            }
        }
    } else {
    const int qrow = q0 + ql;
    if (qrow < a.S) {
        T* op = (T*)a.o + b * a.o_bs + (long)head * a.o_hs + (long)qrow * a.o_rs;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int d = dt * 32 + 8 * g4 + 4 * h;
                if (d < D) {
                    typename T16<T>::vec4 w;
#pragma unroll
                    for (int e = 0; e < 4; ++e) w[e] = T16<T>::from_f32(o[dt][g4 * 4 + e] * inv);
                    *(typename T16<T>::vec4*)(op + d) = w;
                }
            }
    }
    }
}

template <typename T, int D, int NW, bool PV16 = false>
static int launch_sp(AttnArgs a, int batch, int heads, hipStream_t s) {
    a.nqb = ceil_div(a.S, NW * 32);
    typedef AttnCfg<D> C;
    const size_t lds = 2 * (ATT_BK * C::LDK * 2 + ATT_BK * C::LDV * 2);
    static std::atomic<unsigned> done_mask{0};   // per device: the attribute belongs to the device's copy of the kernel
    const int cur_dev = current_device();
    if (!device_done(done_mask, cur_dev)) {
        (void)hipFuncSetAttribute((const void*)attn_sp_kernel<T, D, NW, PV16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        mark_device_done(done_mask, cur_dev);
    }
    hipLaunchKernelGGL((attn_sp_kernel<T, D, NW, PV16>), dim3(a.nqb * heads * batch), dim3(NW * 64), lds, s, a);
    return (int)hipGetLastError();
}

// called by vitg_attention (attention.hip).  pv16 (attn_sp = 3, hd 128 with 8 waves): PV on v_mfma_f32_16x16x32 - built, bit-for-bit tested and
// measured 3-4 % SLOWER than PV on 32x32x16 here (2.50-2.54 against 2.43 ms on the LLM shape, profiles/r03_attention.md): there is no padding to drop
// at hd 128, the 16x16 shape issues at ~17 instead of 16 cycles and the re-layout adds 16 swaps to a hand-interleaved stream.  Opt-in, not the default.
int attention_sp_dispatch(const AttnArgs& a, int batch, int heads, int head_dim, int dtype, int nw, bool pv16, hipStream_t s) {
#if !VITG_PROFILING_BUILD
    // PRODUCT library: head dim 128 only (the hd-72 form and PV on 16x16x32 lost their A/Bs: profiling library)
    (void)pv16;
    if (head_dim != 128) return -1;
    if (dtype == DT_BF16) return nw == 4 ? launch_sp<bf16_t, 128, 4>(a, batch, heads, s) : launch_sp<bf16_t, 128, 8>(a, batch, heads, s);
    return nw == 4 ? launch_sp<f16_t, 128, 4>(a, batch, heads, s) : launch_sp<f16_t, 128, 8>(a, batch, heads, s);
#else
    if (head_dim == 128 && nw == 8 && pv16)
        return dtype == DT_BF16 ? launch_sp<bf16_t, 128, 8, true>(a, batch, heads, s) : launch_sp<f16_t, 128, 8, true>(a, batch, heads, s);
#define SP(T, D) (nw == 3 ? launch_sp<T, D, 3>(a, batch, heads, s) : nw == 4 ? launch_sp<T, D, 4>(a, batch, heads, s) : launch_sp<T, D, 8>(a, batch, heads, s))
    if (dtype == DT_BF16) return head_dim == 72 ? SP(bf16_t, 72) : SP(bf16_t, 128);
    return head_dim == 72 ? SP(f16_t, 72) : SP(f16_t, 128);
#undef SP
#endif
}

}  // namespace vitg

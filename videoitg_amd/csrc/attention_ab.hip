// Non-causal flash attention forward for gfx950 with the two waves of every SIMD running HALF A TILE OUT OF PHASE (same math,
// layouts and transposed formulation as attention.hip; read that file's header first).
//
// In the per-tile kernel all eight waves of a block run  QK^T (MFMA) -> softmax (VALU) -> PV (MFMA)  between the same two tile
// barriers, so the two waves that share a SIMD want the matrix pipe at the same time and the VALU at the same time.  Here waves
// 0..NW/2-1 (group A) keep that order, while waves NW/2..NW-1 (group B: wave w+NW/2 sits on the SIMD of wave w) run, inside the
// barrier interval of tile t,
//        softmax(t-1) (VALU)  ->  PV(t-1) (MFMA)  ->  QK^T(t) (MFMA)
// i.e. they carry the scores of a tile across the barrier and finish it one interval later:
//        A:  MFMA | VALU | MFMA            B:  VALU | MFMA | MFMA
// the first two slots are complementary on every SIMD.  No extra synchronisation: still one barrier per tile.
// Cost: V(t-1) must outlive interval t, so V is a ring of THREE tiles (K stays double-buffered): 2 x 11 + 3 x 12 KiB = 58.5 KiB
// per block at hd 72, two blocks per CU as before; group B holds its 32 score registers across the barrier, which does not
// raise the peak (the softmax of either group is the widest point).
#include "attn_common.h"
#include "../../include/videoitg_hip.h"
#include <type_traits>

namespace vitg {

constexpr float AB_LAZY_RESCALE_LOG2 = 8.0f;      // see attention_sp.hip: running max only moves on a > 2^8 jump

// PV16 (round 3, the default): O^T += V^T P on v_mfma_f32_16x16x32 instead of 32x32x16.
//   * head dim 72 is 4.5 d-tiles of 16: five 16-row tiles (80 rows) instead of three 32-row tiles (96) - 20 MFMAs of 16 pipe cycles per 64-key
//     tile and wave instead of 12 of 32 (320 against 384 cycles; QK^T stays at 10 x 32: d pads to 80 there either way);
//   * the 16x16x32 shape draws less power per flop (tools/ubench/mfma_shape_power.hip: an MFMA-only kernel sustains 1.92 against 1.68 PFLOP/s
//     inside the same package power cap) - and this kernel runs on that cap (profiles/r02_power_trace.txt).
// P leaves the softmax in the 32x32 C layout (lane (q, h): 8 keys 16 kk + {4h..4h+3, 8+4h..8+4h+3} per 16-key block kk).  The 16x16x32 B operand
// wants 16 queries x 4 lane groups g of 8 k-slots each: v_permlane32_swap + v_permlane16_swap of the register pair (p[2kb], p[2kb+1]) give, for
// query tile 0 / 1, the rows [ (2kb, h0), (2kb, h1), (2kb+1, h0), (2kb+1, h1) ] - no LDS round trip, 2 swaps per pair of VGPRs.  The contraction
// order is free as long as both operands agree: lane group g of the V^T fragment reads keys 32 kb + 16 (g >> 1) + 4 (g & 1) + {0..3, 8..11}
// with two ds_read_b64_tr_b16.  V rows sit 160 B apart in LDS (80 elements: 72 + one pad chunk): rows r .. r+7 then cover all 64 banks exactly once,
// which the two lane groups of a half-wave (h0 | h1 rows) need to stay conflict-free (192-B rows cannot give that).
// Row sums on the matrix pipe (PV16): the pad chunk of every V row (columns 72..79 of the 80) holds [1, 0, ..., 0] - written once per block, the DMA
// skips those lanes - so column 72 of O^T += V^T P accumulates l = sum_k P[k] in the MFMAs that run anyway (d-tile 4, lane group 2, register 0).
// The softmax then has no row-sum adds and no l rescale of its own (the lazy rescale of the accumulators covers that column like any other); l is the
// sum of the SAME 16-bit-rounded P the numerator uses.
// LSUM = that form (the default with PV16); false = the softmax adds its row sums on the VALU (attn_ab = 5: A/B and cross-check).
// QS (scores leave the matrix pipe ready for exp2; needs Q pre-multiplied by scale * log2(e) - vitg_attention variant bit 1 - which the engine folds
// into the q rows of the tower's qkv weights): the same pad column on the QK^T side.  K's pad chunk (d 72..79) holds [1, 0, ..], Q's pad element 72
// holds -shift of its query, so S' = K Q'^T - shift comes out of the MFMAs that run anyway and P = exp2(S') needs neither the scale multiply nor
// the max subtraction (one packed FMA per score pair less).  Softmax is shift-invariant per query, so the shift only has to be CONSISTENT between
// the numerator, the row sum (same P) and the accumulator rescale (alpha = exp2(old shift - new shift)): it is the running max rounded to 16 bits,
// moved - as before - only when a tile's scores exceed it by more than 2^8, and then that one tile subtracts the step on the VALU.
// NKT_C (round 6, VERDICT r5 item 4: "a compile-time S = 1024 specialisation with no tail / key-count path for the R = 448 tower"): > 0 = the number of
// 64-key tiles is this constant - no device key count, no tail mask, the tile loops run a known trip count (+3.3 % on the launch inside the step:
// profiles/r06_attention_hd72.md).  Bit-identical to the run-time form.
template <typename T, int D, int NW, bool PV16, bool LSUM = PV16, bool QS = false, int NKT_C = 0>
__global__ __launch_bounds__(NW * 64, 4) void attn_ab_kernel(AttnArgs a) {
    static_assert(!LSUM || PV16, "the row-sum column rides the 16x16x32 PV form");
    static_assert(!QS || LSUM, "the shift column comes with the row-sum column");
    if constexpr (NKT_C == 0) adopt_device_key_count(a);
    typedef typename T16<T>::vec8 vec8;
    typedef AttnCfg<D> C;
    constexpr int KS = C::DK / 16;           // QK^T k-steps per 32-key half
    constexpr int LDV = PV16 ? 80 : C::LDV;
    constexpr int DT = PV16 ? 5 : C::DV / 32;   // PV d-tiles (of 16 / 32 rows)
    constexpr int KT_BYTES = ATT_BK * C::LDK * 2, VT_BYTES = ATT_BK * LDV * 2;
    constexpr int V_BASE = 2 * KT_BYTES;     // LDS: [K0][K1][V0][V1][V2]
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
    constexpr int K_CPR = C::LDK * 2 / 16, V_CPR = LDV * 2 / 16;          // 16-B slots per row = DMA instructions per tile
    constexpr int K_PW = (K_CPR + NW - 1) / NW, V_PW = (V_CPR + NW - 1) / NW;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    auto dma_src = [&](int ii, int cpr, int rs, bool is_k, int row_lim) -> unsigned {
        const int j = ii * 64 + lane;
        int row = j / cpr, c = j % cpr;
        if (D == 128) c = is_k ? (c ^ (row & 15)) : (c ^ ((row & 3) << 2));
        else c = min(c, C::CH - 1);          // pad slots re-read the row's last data chunk (finite filler)
        row = min(row, row_lim);
        return (unsigned)(row * rs + c * 8) * 2u;
    };
    unsigned dk_off[K_PW], dv_off[V_PW];
    bool dv_pad[V_PW];                       // PV16: this lane's slot is the pad chunk of its V row (holds the constant [1, 0, ..]: never written by the DMA)
    bool dk_pad[K_PW];                       // QS: the same for K (chunks 9, 10 of the 11)
#pragma unroll
    for (int i = 0; i < K_PW; ++i) {
        dk_off[i] = dma_src(wave_u + NW * i, K_CPR, a.k_rs, true, ATT_BK - 1);
        dk_pad[i] = QS && ((wave_u + NW * i) * 64 + lane) % K_CPR >= C::CH;
    }
#pragma unroll
    for (int i = 0; i < V_PW; ++i) {
        dv_off[i] = dma_src(wave_u + NW * i, V_CPR, a.v_rs, false, ATT_BK - 1);
        dv_pad[i] = LSUM && ((wave_u + NW * i) * 64 + lane) % V_CPR >= C::CH;
    }
#ifndef VITG_AB_BUF_DMA
#define VITG_AB_BUF_DMA 1      // inside the step: 2.957 -> 2.932 ms per launch (+0.9 %), three alternations (gpurun_out/r05_abbuf); the kernel is VALU-bound
#endif
    // VITG_AB_BUF_DMA: staging through buffer descriptors whose extent ends with the last key row (attn_common.h: lds_dma16_buf; rows past it read as
    // zeros) - no row clamp, no tail branch: three instructions per piece where the clamped global form took ~15
    constexpr bool BUF_DMA = VITG_AB_BUF_DMA && VITG_ASM_DMA;
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
            if (D == 128) c = is_k ? (c ^ (row & 15)) : (c ^ ((row & 3) << 2));
            else c = min(c, C::CH - 1);
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
                if (wave_u + NW * i < K_CPR) {
                    if (!dk_pad[i]) lds_dma16_buf(k_rsrc, lds_w + buf * KT_BYTES + NW * i * 1024, dk_offb[i] + kt * k_tile_bytes);
                }
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
                if (!dk_pad[i])
                    lds_dma16(Kt, off, kb + ii * 1024);
            }
        }
    };
    auto issue_v = [&](int kt, int buf) {
        if constexpr (BUF_DMA) {
#pragma unroll
            for (int i = 0; i < V_PW; ++i)
                if (wave_u + NW * i < V_CPR) {
                    if (!dv_pad[i]) lds_dma16_buf(v_rsrc, lds_w + V_BASE + buf * VT_BYTES + NW * i * 1024, dv_offb[i] + kt * v_tile_bytes);
                }
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
                if (!dv_pad[i])              // (the LDS side of the DMA is lane-linear whatever EXEC is: masked lanes leave their 16 B untouched)
                    lds_dma16(Vt, off, vb + ii * 1024);
            }
        }
    };

    // lane-dependent LDS read bases; inside the loop every fragment address is one of these + an immediate
    const char* kbase[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) kbase[ks] = smem + k_lds_off<D>(ql, ks * 2 + h);      // + t*32 rows: row & 15 unchanged
    const int g16 = lane >> 4;
    const int voff = PV16 ? (16 * (g16 >> 1) + 4 * (g16 & 1) + ((lane & 15) >> 2)) * (LDV * 2) + (4 * (lane & 3)) * 2
                          : (4 * h + ((lane & 15) >> 2)) * (LDV * 2) + ((((lane >> 4) & 1) * 16) + 4 * (lane & 3)) * 2;
    const char* vbase[PV16 ? 1 : DT];
    if constexpr (PV16) vbase[0] = smem + V_BASE + voff;
    else {
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
    const int nkt = NKT_C > 0 ? NKT_C : ceil_div(a.Sk, ATT_BK);

    // runtime-selected buffers (K ring of 2, V ring of 3): per-interval base pointers, fragment offsets stay immediates
    auto qk_mfma = [&](int kbuf, f32x16 (&s)[2]) {
        const int ko = kbuf * KT_BYTES;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const vec8 kf = *(const vec8*)(kbase[ks] + ko + t * 32 * (C::LDK * 2));
                s[t] = ks == 0 ? T16<T>::mfma32(kf, qf[0], f32x16{}) : T16<T>::mfma32(kf, qf[ks], s[t]);
            }
    };
    auto pv_mfma = [&](int vbuf, const vec8 (&p)[4]) {
        const int vo = vbuf * VT_BYTES;
        if constexpr (PV16) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                // B operands of query tile 0 / 1 for the 32 keys of block kb: rows [ (2kb, h0), (2kb, h1), (2kb+1, h0), (2kb+1, h1) ]
                const u32x4 X = bitcast<u32x4>(p[2 * kb]), Y = bitcast<u32x4>(p[2 * kb + 1]);
                u32x4 P0, P1;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const u32x2 t = __builtin_amdgcn_permlane32_swap(X[i], Y[i], false, false);      // [X0 X1 Y0 Y1], [X2 X3 Y2 Y3]  (rows of 16 lanes)
                    const u32x2 u = __builtin_amdgcn_permlane16_swap(t[0], t[1], false, false);      // [X0 X2 Y0 Y2], [X1 X3 Y1 Y3]
                    P0[i] = u[0]; P1[i] = u[1];
                }
                const vec8 pb0 = bitcast<vec8>(P0), pb1 = bitcast<vec8>(P1);
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    const char* a0 = vbase[0] + vo + kb * 32 * (LDV * 2) + dt * 32;
                    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(a0));
                    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(a0 + 8 * (LDV * 2)));
                    const vec8 vf = bitcast<vec8>(__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
                    o16[dt][0] = T16<T>::mfma16(vf, pb0, o16[dt][0]);
                    o16[dt][1] = T16<T>::mfma16(vf, pb1, o16[dt][1]);
                }
            }
        } else {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const char* a0 = vbase[dt] + vo + ((kk >> 1) * 32 + (kk & 1) * 16) * (LDV * 2);
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(a0));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(a0 + 8 * (LDV * 2)));
                o[dt] = T16<T>::mfma32(bitcast<vec8>(__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7)), p[kk], o[dt]);
            }
        }
    };
    auto mask_tail = [&](int t, f32x16 (&s)[2]) {          // last, partial tile: lane (q, h) s[tt][r] is key t*64 + tt*32 + 8*(r>>2) + 4*h + (r&3)
        if constexpr (NKT_C > 0) return;                   // (the key count is NKT_C * 64: there is no partial tile)
        if (t * ATT_BK + ATT_BK > a.Sk) {
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (t * ATT_BK + tt * 32 + 8 * (r >> 2) + 4 * h + (r & 3) >= a.Sk) s[tt][r] = -INFINITY;
        }
    };
    // online softmax of one tile (lane-local + one exchange with lane^32), lazy rescale; P -> p (16-bit, MFMA B operand layout)
    float shift = 0.f;                       // QS: what the Q pad column subtracts from this query's scores (a 16-bit value)
    bool started = false;                    // QS: wave-uniform - a tile has been accumulated (the first tile always sets the shift)
    auto softmax = [&](f32x16 (&s)[2], vec8 (&p)[4]) {
        if constexpr (QS) {
            float mx = s[0][0];              // scores are already in log2 units and relative to `shift`
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[tt][r]);
            const u32x2 sw = __builtin_amdgcn_permlane32_swap(bitcast<uint32_t>(mx), bitcast<uint32_t>(mx), false, false);
            mx = fmaxf(bitcast<float>(sw[0]), bitcast<float>(sw[1]));
            if (!started || __builtin_amdgcn_ballot_w64(mx > AB_LAZY_RESCALE_LOG2) != 0) {
                const float ns = T16<T>::to_f32(T16<T>::from_f32(shift + (started ? fmaxf(mx, 0.f) : mx)));      // new shift: never lowered, 16-bit exact
                const float delta = ns - shift;
                const float alpha = started ? __builtin_amdgcn_exp2f(-delta) : 0.f;
                const u32x2 aa = __builtin_amdgcn_permlane16_swap(bitcast<uint32_t>(alpha), bitcast<uint32_t>(alpha), false, false);
                const float a0 = bitcast<float>(aa[0]), a1 = bitcast<float>(aa[1]);
#pragma unroll
                for (int i = 0; i < DT; ++i) { o16[i][0] *= a0; o16[i][1] *= a1; }
#pragma unroll
                for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) s[tt][r] -= delta;                 // this tile was computed against the old shift
                shift = ns;
                started = true;
                if (h == 1) qf[KS - 1][0] = T16<T>::from_f32(-ns);                  // element d = 72 of this query's Q row
            }
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const float v0 = __builtin_amdgcn_exp2f(s[tt][r]), v1 = __builtin_amdgcn_exp2f(s[tt][r + 1]);
                    p[tt * 2 + (r >> 3)][r & 7] = T16<T>::from_f32(v0);
                    p[tt * 2 + (r >> 3)][(r & 7) + 1] = T16<T>::from_f32(v1);
                }
            return;
        }
        float mx = s[0][0];
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[tt][r]);
        const u32x2 sw = __builtin_amdgcn_permlane32_swap(bitcast<uint32_t>(mx), bitcast<uint32_t>(mx), false, false);
        mx = fmaxf(bitcast<float>(sw[0]), bitcast<float>(sw[1])) * a.scale_log2e;      // scale > 0
        const float m_new = fmaxf(m_run, mx);
        if (__builtin_amdgcn_ballot_w64(m_new > m_run + AB_LAZY_RESCALE_LOG2) != 0) {
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            if constexpr (!LSUM) l_run *= alpha;
            if constexpr (PV16) {
                // the accumulators of query tile 0 / 1 sit in ALL four lane rows: row 0 (= row 2) / row 1 (= row 3) holds those queries' factor
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
        f32x2 psum2 = {0.f, 0.f};
        const f32x2 sc2 = {a.scale_log2e, a.scale_log2e};
        const f32x2 nm2 = {-m_run, -m_run};
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                f32x2 v = {s[tt][r], s[tt][r + 1]};
                v = __builtin_elementwise_fma(v, sc2, nm2);
                v = f32x2{__builtin_amdgcn_exp2f(v[0]), __builtin_amdgcn_exp2f(v[1])};
                if constexpr (!LSUM) psum2 += v;
                p[tt * 2 + (r >> 3)][r & 7] = T16<T>::from_f32(v[0]);
                p[tt * 2 + (r >> 3)][(r & 7) + 1] = T16<T>::from_f32(v[1]);
            }
        if constexpr (!LSUM) l_run += psum2[0] + psum2[1];
    };

    const bool group_b = wave_u >= NW / 2;
    f32x16 s[2];
    vec8 p[4];
    if constexpr (LSUM) {                    // the constant pad chunk of the 3 x 64 V rows
        for (int r = tid; r < 3 * ATT_BK; r += NW * 64) {
            vec8 one;
#pragma unroll
            for (int e = 0; e < 8; ++e) one[e] = T16<T>::from_f32(e == 0 ? 1.f : 0.f);
            *(vec8*)(smem + V_BASE + (r / ATT_BK) * VT_BYTES + (r % ATT_BK) * (LDV * 2) + C::CH * 16) = one;
        }
    }
    if constexpr (QS) {                      // the constant pad chunk (d 72..79) of the 2 x 64 K rows
        for (int r = tid; r < 2 * ATT_BK; r += NW * 64) {
            vec8 one;
#pragma unroll
            for (int e = 0; e < 8; ++e) one[e] = T16<T>::from_f32(e == 0 ? 1.f : 0.f);
            *(vec8*)(smem + (r / ATT_BK) * KT_BYTES + k_lds_off<D>(r % ATT_BK, C::CH)) = one;
        }
    }
    issue_k(0, 0);
    issue_v(0, 0);
    { lds_dma_drain(); __syncthreads(); }
    // Two separate loops (not one loop with a role branch inside): with the branch inside, the register allocator keeps both
    // roles' accumulator assignments alive across the back edge (247 VGPRs instead of 131).  Both loops execute the same
    // number of barriers.  (sched_barriers: the phases must not be interleaved by the compiler - overlapping QK^T(t) with the
    // softmax of the carried tile would need a second score buffer.)
    auto issue_next = [&](int t, int vb_next) {            // K(t-1) and V(t-2) are dead for both groups since the last barrier
        if (t + 1 < nkt) {
            issue_k(t + 1, (t + 1) & 1);
            issue_v(t + 1, vb_next);
        }
    };
    if (!group_b) {
        int vb_cur = 0;                                    // t % 3 without a division in the loop
        for (int t = 0; t < nkt; ++t) {
            const int vb_next = vb_cur == 2 ? 0 : vb_cur + 1;
            issue_next(t, vb_next);
            qk_mfma(t & 1, s);
            mask_tail(t, s);
            __builtin_amdgcn_sched_barrier(0);
            softmax(s, p);
            __builtin_amdgcn_sched_barrier(0);
            pv_mfma(vb_cur, p);
            vb_cur = vb_next;
            { lds_dma_drain(); __syncthreads(); }                               // drains vmcnt too: tile t+1 has landed
        }
    } else {
        int vb_prev = 0, vb_cur = 0;
        for (int t = 0; t < nkt; ++t) {
            const int vb_next = vb_cur == 2 ? 0 : vb_cur + 1;
            issue_next(t, vb_next);
            if (t > 0) {
                softmax(s, p);                             // scores of tile t-1, carried across the barrier
                __builtin_amdgcn_sched_barrier(0);
                pv_mfma(vb_prev, p);
                __builtin_amdgcn_sched_barrier(0);
            }
            qk_mfma(t & 1, s);
            mask_tail(t, s);
            vb_prev = vb_cur;
            vb_cur = vb_next;
            { lds_dma_drain(); __syncthreads(); }
        }
        softmax(s, p);                                     // finish the last tile (V(nkt-1) is still resident: nothing was issued after it)
        __builtin_amdgcn_sched_barrier(0);
        pv_mfma(vb_prev, p);
    }

    // ---- epilogue: lane (q, h) holds O[q][dt*32 + 8*(r>>2) + 4*h + (r&3)] -------------------------------------------------
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    if constexpr (PV16) {
        const u32x2 ii = __builtin_amdgcn_permlane16_swap(bitcast<uint32_t>(inv), bitcast<uint32_t>(inv), false, false);
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            // LSUM: l of query 16 qt + (lane & 15) = column 72 of its O row: d-tile 4, lane group 2, register 0
            const float iq = LSUM ? 1.0f / __shfl(o16[DT - 1][qt][0], 32 + (lane & 15), 64) : bitcast<float>(ii[qt]);
            const int qrow = q0 + 16 * qt + (lane & 15);
            if (qrow < a.S) {
                T* op = (T*)a.o + b * a.o_bs + (long)head * a.o_hs + (long)qrow * a.o_rs;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    const int d = dt * 16 + 4 * g16;
                    if (d < D) {
                        typename T16<T>::vec4 w;
#pragma unroll
                        for (int e = 0; e < 4; ++e) w[e] = T16<T>::from_f32(o16[dt][qt][e] * iq);
                        *(typename T16<T>::vec4*)(op + d) = w;
                    }
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

template <typename T, int D, int NW, bool PV16, bool LSUM = PV16, bool QS = false, int NKT_C = 0>
static int launch_ab(AttnArgs a, int batch, int heads, hipStream_t s) {
    a.nqb = ceil_div(a.S, NW * 32);
    typedef AttnCfg<D> C;
    const size_t lds = 2 * (ATT_BK * C::LDK * 2) + 3 * (ATT_BK * (PV16 ? 80 : C::LDV) * 2);
    static bool done_dev[16] = {false};          // per device: the attribute belongs to the device's copy of the kernel
    int cur_dev = 0;
    (void)hipGetDevice(&cur_dev);
    if (cur_dev >= 16 || !done_dev[cur_dev]) {
        (void)hipFuncSetAttribute((const void*)attn_ab_kernel<T, D, NW, PV16, LSUM, QS, NKT_C>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (cur_dev < 16) done_dev[cur_dev] = true;
    }
    hipLaunchKernelGGL((attn_ab_kernel<T, D, NW, PV16, LSUM, QS, NKT_C>), dim3(a.nqb * heads * batch), dim3(NW * 64), lds, s, a);
    return (int)hipGetLastError();
}

// called by vitg_attention (attention.hip); hd 72 only (hd 128 runs the software-pipelined kernel)
// waves = 8: 256 queries per block, two blocks per CU.  waves = 16: 512 queries per block, ONE block per CU sharing one K/V ring - the same
// 4 waves per SIMD (each SIMD holds two group-A and two group-B waves), half the K/V bytes streamed out of L2 per query.
// pv32 (attn_ab = 4, A/B and cross-check): the round-2 form with PV on v_mfma_f32_32x32x16 (head dim padded to 96 there).
// form (attn_ab): 0 default | 4 PV on 32x32x16 (round 2) | 5 PV on 16x16x32 with the row sums on the VALU (round 3 before the row-sum column)
// qs: Q arrives pre-multiplied by scale * log2(e) (a.scale_log2e == 1) and the default form may take the scores straight from the matrix pipe;
// form 6 = pre-scaled Q WITHOUT that (the multiply by 1 and the max subtraction stay on the VALU: A/B)
int attention_ab_dispatch(const AttnArgs& a, int batch, int heads, int dtype, int waves, int form, bool qs, hipStream_t s) {
    const bool pv32 = form == 4;
    // round 6: 1024 keys - the 32 x 32 grid of the R = 448 tower, BASELINE configs[1] - as a compile-time 16 tiles: no device key
    // count, no tail-mask test per tile, a known trip count.  Same-call alternation inside the step, three rounds (gpurun_out/r06_attn72/ab.txt): 2.946 / 2.952 /
    // 2.963 -> 2.851 / 2.855 / 2.864 ms per launch (+3.3 %, 0.335 -> 0.347 of peak), step 631.3 / 633.0 / 632.7 -> 629.3 / 630.6 / 631.6 ms; bit-identical.
    if (qs && form == 0 && waves == 8 && a.Sk == 16 * ATT_BK && a.sk_dev == nullptr)
        return dtype == DT_BF16 ? launch_ab<bf16_t, 72, 8, true, true, true, 16>(a, batch, heads, s) : launch_ab<f16_t, 72, 8, true, true, true, 16>(a, batch, heads, s);
    if (qs && form == 0 && waves == 8)
        return dtype == DT_BF16 ? launch_ab<bf16_t, 72, 8, true, true, true>(a, batch, heads, s) : launch_ab<f16_t, 72, 8, true, true, true>(a, batch, heads, s);
#if VITG_PROFILING_BUILD   // the forms that lost their A/B (profiles/r03_attention.md): profiling library only
    if (form == 5) return dtype == DT_BF16 ? launch_ab<bf16_t, 72, 8, true, false>(a, batch, heads, s) : launch_ab<f16_t, 72, 8, true, false>(a, batch, heads, s);
    if (pv32) return dtype == DT_BF16 ? launch_ab<bf16_t, 72, 8, false>(a, batch, heads, s) : launch_ab<f16_t, 72, 8, false>(a, batch, heads, s);
    if (waves == 16) return dtype == DT_BF16 ? launch_ab<bf16_t, 72, 16, true>(a, batch, heads, s) : launch_ab<f16_t, 72, 16, true>(a, batch, heads, s);
#else
    (void)pv32; (void)waves;
#endif
    return dtype == DT_BF16 ? launch_ab<bf16_t, 72, 8, true>(a, batch, heads, s) : launch_ab<f16_t, 72, 8, true>(a, batch, heads, s);
}

}  // namespace vitg

// Non-causal (bidirectional) flash attention forward for gfx950, two instantiations:
//   D=72  (SigLIP: 16 heads, seq 729, batch = frames)     -> padded to 80 (QK^T) / 96 (PV) in LDS only
//   D=128 (Qwen2: 28 q heads / 4 kv heads GQA, seq ~12.9K, batch 1)
//
// Everything is computed TRANSPOSED so that one lane owns one query column:
//   S^T[key][q] = K[key][:] . Q[q][:]     MFMA A = K tile (LDS, ds_read_b128), B = Q (registers, loaded once)
//   O^T[d][q]  += V^T[d][key] . P^T[key][q]   MFMA A = V^T via ds_read_b64_tr_b16 on the row-major V tile,
//                                              B = P (this lane's own exp'd scores, packed to 16 bit)
// so the running max / sum / rescale of the online softmax are lane-local (one exchange with lane^32 per
// tile), and P never leaves registers.  The key -> MFMA k-slot assignment is a fixed permutation used
// identically for P (B operand) and V (A operand), which a contraction is invariant to.
//
// Block = 4 waves x 32 queries = 128 queries; KV tile = 64 keys, double-buffered in LDS.
#include "attn_common.h"
#include <math.h>
#include "../../include/videoitg_hip.h"
#include <type_traits>
#include <stdlib.h>

namespace vitg {

template <typename T, int D, bool TR, int NW, bool ABL = false, bool PIPE = false, bool DMA = false>
__global__ __launch_bounds__(NW * 64, (NW == 8 && D == 72 && DMA) ? 4 : 2) void attn_kernel(AttnArgs a) {
    adopt_device_key_count(a);   // (256,3) for hd72 spills: 548 -> 428 TF/s
    const int ablate = ABL ? a.ablate : 0;      // ablation switches exist only in the profiling instantiation
    constexpr int ATT_THREADS = NW * 64, ATT_BQ = NW * 32;
    typedef typename T16<T>::vec8 vec8;
    typedef AttnCfg<D> C;
    constexpr int KS = C::DK / 16;           // QK^T k-steps
    constexpr int DT = C::DV / 32;           // PV d-tiles
    constexpr int KT_BYTES = ATT_BK * C::LDK * 2;
    constexpr int VT_BYTES = ATT_BK * C::LDV * 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // layout: [buf0: K | V][buf1: K | V]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ql = lane & 31, h = lane >> 5;
    // Block -> (batch, kv head, head-in-group, q-block).  Hardware block b runs on XCD b % 8 (private 4 MiB L2 each), so
    // the ids are first made XCD-contiguous: all blocks that read the same K/V (the q-blocks of one kv head) then sit on
    // ONE XCD and re-read K/V from its L2 instead of 8 XCDs each fetching their own copy from HBM (measured before the
    // remap: ViT attention fetched 19.6 GB per launch = 8x the K/V bytes, 4.2 TB/s, HBM-bound).
    int L;
    {
        const int nwg = gridDim.x, bid = blockIdx.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int qb = L % a.nqb;
    const int hg = (L / a.nqb) % a.group;
    const int kvh = (L / (a.nqb * a.group)) % a.kv_heads;
    const int b = L / (a.nqb * a.group * a.kv_heads);
    const int head = kvh * a.group + hg;
    const int q0 = qb * ATT_BQ + wave * 32;

    const T* Q = (const T*)a.q + b * a.q_bs + (long)head * a.q_hs;
    const T* K = (const T*)a.k + b * a.k_bs + (long)kvh * a.k_hs;
    const T* V = (const T*)a.v + b * a.v_bs + (long)kvh * a.v_hs;

    // ---- Q fragments (B operand): lane (q = ql, h) holds d = 16 ks + 8 h .. +7 -------------
    vec8 qf[KS];
    {
        const int qrow = min(q0 + ql, a.S - 1);
        const T* qp = Q + (long)qrow * a.q_rs;
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

    // ---- staging assignment ---------------------------------------------------------------------------
    // K tile: 64 rows x CH chunks(16 B);  V tile: 64 rows x CH chunks
    constexpr int NCH = ATT_BK * C::CH;                      // chunks per tile (1024 or 576)
    constexpr int PER = (NCH + ATT_THREADS - 1) / ATT_THREADS;
    u32x4 rk[PER], rv[PER];
    // staging geometry is loop-invariant: per-thread (row, chunk) -> global element offsets and LDS byte offsets are
    // computed once; per tile only a wave-uniform base advances (the row clamp is needed in the last tile only)
    int g_row[PER], gk_off[PER], gv_off[PER], lk_off[PER], lv_off[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int c = tid + i * ATT_THREADS;
        const int row = c / C::CH, ch = c % C::CH;
        g_row[i] = row;
        gk_off[i] = row * a.k_rs + ch * 8;
        gv_off[i] = row * a.v_rs + ch * 8;
        lk_off[i] = k_lds_off<D>(row, ch);
        // D=128: V rows are 256 B = one LDS bank row, so the 4 key rows of a tr-read block would collide 4-way;
        // XOR the 64-B unit index with (row & 3) (read side: voff_dt)
        const int vch = D == 128 ? (ch ^ ((row & 3) << 2)) : ch;
        lv_off[i] = row * (C::LDV * 2) + (vch << 4);
    }
    auto load_one = [&](int kt, bool do_k, bool do_v) {
        const T* Kt = K + (long)kt * ATT_BK * a.k_rs;
        const T* Vt = V + (long)kt * ATT_BK * a.v_rs;
        const bool tail = kt * ATT_BK + ATT_BK > a.Sk;          // wave-uniform
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            if ((NCH % ATT_THREADS == 0) || tid + i * ATT_THREADS < NCH) {
                int ko = gk_off[i], vo = gv_off[i];
                if (tail) {
                    const int r = min(g_row[i], a.Sk - 1 - kt * ATT_BK);
                    ko += (r - g_row[i]) * a.k_rs;
                    vo += (r - g_row[i]) * a.v_rs;
                }
                if (do_k) rk[i] = *(const u32x4*)(Kt + ko);
                if (do_v) rv[i] = *(const u32x4*)(Vt + vo);
            }
        }
    };
    auto load_kv = [&](int kt) { load_one(kt, true, true); };
    auto store_one = [&](int kbuf, int vbuf, bool do_k, bool do_v) {
        char* kb = smem + kbuf * (KT_BYTES + VT_BYTES);
        char* vb = smem + vbuf * (KT_BYTES + VT_BYTES) + KT_BYTES;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            if ((NCH % ATT_THREADS == 0) || tid + i * ATT_THREADS < NCH) {
                if (do_k) *(u32x4*)(kb + lk_off[i]) = rk[i];
                if (do_v) *(u32x4*)(vb + lv_off[i]) = rv[i];
            }
        }
        if constexpr (D == 72) {
            // zero the K pad chunk (d 72..79) once per buffer fill: 64 rows
            if (do_k && tid < ATT_BK) {
                u32x4 z = {0u, 0u, 0u, 0u};
                *(u32x4*)(kb + k_lds_off<D>(tid, 9)) = z;
            }
        }
    };
    auto store_kv = [&](int buf) { store_one(buf, buf, true, true); };

    // ---- DMA staging (global_load_lds_dwordx4): no staging registers, no ds_write --------------------------------------
    // One instruction fills 1 KiB of LDS lane-linearly = 64 consecutive 16-B chunks of the tile image; chunk j lives at row
    // j / CPR, slot j % CPR.  The per-lane SOURCE address carries the layout: hd 128 reads global chunk slot ^ swizzle(row);
    // hd 72 rows are padded in LDS (K 11 slots, V 12 slots for 9 data chunks): pad slots re-read the row's last data chunk
    // (finite filler: the K pad only ever meets Q's zero pad, the V pad only feeds output rows that are never stored).
    constexpr int K_CPR = C::LDK * 2 / 16, V_CPR = C::LDV * 2 / 16;       // = DMA instructions per tile (64 rows * CPR / 64 lanes)
    constexpr int K_PW = (K_CPR + NW - 1) / NW, V_PW = (V_CPR + NW - 1) / NW;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    unsigned dk_off[K_PW], dv_off[V_PW];
    auto dma_src = [&](int ii, int cpr, int rs, bool is_k, int row_lim) -> unsigned {
        const int j = ii * 64 + lane;
        int row = j / cpr, c = j % cpr;
        if (D == 128) c = is_k ? (c ^ (row & 15)) : (c ^ ((row & 3) << 2));
        else c = min(c, C::CH - 1);
        row = min(row, row_lim);
        return (unsigned)(row * rs + c * 8) * 2u;
    };
    if constexpr (DMA) {
#pragma unroll
        for (int i = 0; i < K_PW; ++i) dk_off[i] = dma_src(wave_u + NW * i, K_CPR, a.k_rs, true, ATT_BK - 1);
#pragma unroll
        for (int i = 0; i < V_PW; ++i) dv_off[i] = dma_src(wave_u + NW * i, V_CPR, a.v_rs, false, ATT_BK - 1);
    }
    auto issue_dma = [&](int kt, int buf) {
        char* kb = smem + buf * (KT_BYTES + VT_BYTES);
        char* vb = kb + KT_BYTES;
        const char* Kt = (const char*)(K + (long)kt * ATT_BK * a.k_rs);
        const char* Vt = (const char*)(V + (long)kt * ATT_BK * a.v_rs);
        const bool tail = kt * ATT_BK + ATT_BK > a.Sk;          // wave-uniform: clamp rows past the last key
        const int lim = a.Sk - 1 - kt * ATT_BK;
#pragma unroll
        for (int i = 0; i < K_PW; ++i) {
            const int ii = wave_u + NW * i;
            if (ii < K_CPR) {
                const unsigned off = tail ? dma_src(ii, K_CPR, a.k_rs, true, lim) : dk_off[i];
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Kt + off),
                                                 (__attribute__((address_space(3))) void*)(kb + ii * 1024), 16, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < V_PW; ++i) {
            const int ii = wave_u + NW * i;
            if (ii < V_CPR) {
                const unsigned off = tail ? dma_src(ii, V_CPR, a.v_rs, false, lim) : dv_off[i];
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Vt + off),
                                                 (__attribute__((address_space(3))) void*)(vb + ii * 1024), 16, 0, 0);
            }
        }
    };

    f32x16 o[DT];
#pragma unroll
    for (int i = 0; i < DT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
    float m_run = -1e30f, l_run = 0.f;

    const int nkt = ceil_div(a.Sk, ATT_BK);
    if constexpr (DMA) issue_dma(0, 0);
    else { load_kv(0); store_kv(0); }
    __syncthreads();

    // lane-dependent LDS offsets, hoisted: inside the loop every fragment address is one of these + an immediate
    int koff[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) koff[ks] = k_lds_off<D>(ql, ks * 2 + h);      // + t*32 rows: row&15 unchanged
    // V via ds_read_b64_tr_b16: 16-lane group g = lane>>4 -> column block 16*(g&1); lane p = lane&15 supplies the address
    // of row (p>>2), columns 4*(p&3)..+3 of a 4x16 block; the lane receives column (lane&15), rows 0..3.
    const int voff = (4 * h + ((lane & 15) >> 2)) * (C::LDV * 2) + ((((lane >> 4) & 1) * 16) + 4 * (lane & 3)) * 2;
    int voff_dt[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) voff_dt[dt] = voff + (D == 128 ? ((dt ^ ((lane & 15) >> 2)) * 64) : dt * 64);

    const char* kbase[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) kbase[ks] = smem + koff[ks];
    const char* vbase[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) vbase[dt] = smem + voff_dt[dt];

    auto tile = [&](int kt, auto buf_c) {
        constexpr int BUF = decltype(buf_c)::value;
        const bool more = kt + 1 < nkt && !(ablate & 8);
        if (more) { if constexpr (DMA) issue_dma(kt + 1, BUF ^ 1); else load_kv(kt + 1); }
        const char* kb = smem + BUF * (KT_BYTES + VT_BYTES);
        const char* vb = kb + KT_BYTES;

        // ---- S^T = K Q^T : two 32-key tiles ------------------------------------------------------------
        f32x16 s[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            if (ABL && (ablate & 1)) {
#pragma unroll
                for (int r = 0; r < 16; ++r) s[t][r] = 0.f;
            } else {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const vec8 kf = *(const vec8*)(kbase[ks] + (BUF * (KT_BYTES + VT_BYTES) + t * 32 * (C::LDK * 2)));
                    // first k-step accumulates onto a literal zero (inline constant C operand): no 32 v_mov to clear s
                    s[t] = ks == 0 ? T16<T>::mfma32(kf, qf[0], f32x16{}) : T16<T>::mfma32(kf, qf[ks], s[t]);
                }
            }
        }
        // lane (q = ql, h): s[t][r] is key  kt*64 + t*32 + 8*(r>>2) + 4*h + (r&3)
        if (kt * ATT_BK + ATT_BK > a.Sk) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kt * ATT_BK + t * 32 + 8 * (r >> 2) + 4 * h + (r & 3);
                    if (key >= a.Sk) s[t][r] = -INFINITY;
                }
        }
        // ---- online softmax (base-2), lane-local + one exchange with lane^32 ----------------------------
        // (scores are scaled by softmax_scale*log2(e) in fp32 here, not by rounding a pre-scaled Q to 16 bit)
        float mx = s[0][0];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[t][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64)) * a.scale_log2e;      // scale > 0
        const float m_new = fmaxf(m_run, mx);
        // exact deferral: when no query of this wave saw its running max grow, alpha == 1 for every lane and the
        // rescale of O and l (the widest VALU block of the tile) is skipped
        if (__builtin_amdgcn_ballot_w64(m_new != m_run) != 0) {
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            l_run *= alpha;
#pragma unroll
            for (int i = 0; i < DT; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
            m_run = m_new;
        }
        // The softmax block is VALU-issue bound (PMC: VALU busy > MFMA busy), so the exponent arguments and the row sum
        // use packed fp32 (v_pk_fma_f32 / v_pk_add_f32: two scores per issue); the 32 v_exp_f32 stay scalar.
        f32x2 psum2 = {0.f, 0.f};
        const f32x2 sc2 = {a.scale_log2e, a.scale_log2e};
        const f32x2 nm2 = {-m_run, -m_run};
        vec8 pf[4];                      // k-step kk = 2*t + u uses s[t][8u .. 8u+7]
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                f32x2 v = {s[t][r], s[t][r + 1]};
                if (!(ABL && (ablate & 2))) {
                    v = __builtin_elementwise_fma(v, sc2, nm2);
                    v = f32x2{__builtin_amdgcn_exp2f(v[0]), __builtin_amdgcn_exp2f(v[1])};
                }
                psum2 += v;
                pf[t * 2 + (r >> 3)][r & 7] = T16<T>::from_f32(v[0]);
                pf[t * 2 + (r >> 3)][(r & 7) + 1] = T16<T>::from_f32(v[1]);
            }
        l_run += psum2[0] + psum2[1];

        // ---- O^T += V^T P^T ----------------------------------------------------------------------------
        // k-step kk (16 keys: tile t = kk>>1, half u = kk&1): slot (h, j) <-> key t*32 + u*16 + 8*(j>>2) + 4*h + (j&3)
        // A operand lane (d = dt*32 + ql, h) needs V[key(h, j)][d] for j = 0..7: two groups of 4 consecutive keys.
        if (!(ablate & 4))
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int krow = (kk >> 1) * 32 + (kk & 1) * 16;             // + 4*h inside voff
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                vec8 vf;
                if constexpr (TR) {
                    // one hoisted base pointer per d-tile + compile-time offsets (they fold into the instruction's offset field)
                    const char* a0 = vbase[dt] + (BUF * (KT_BYTES + VT_BYTES) + KT_BYTES + krow * (C::LDV * 2));
                    const char* a1 = a0 + 8 * (C::LDV * 2);
                    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(a0));
                    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(a1));
                    vf = bitcast<vec8>(__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
                } else {
                    const T* vt = (const T*)vb;
                    const int d = dt * 32 + ql;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int row = krow + 4 * h + 8 * (j >> 2) + (j & 3);
                        const int dd = D == 128 ? (d ^ ((row & 3) << 5)) : d;      // same 64-B-unit swizzle, in elements
                        vf[j] = vt[row * C::LDV + dd];
                    }
                }
                o[dt] = T16<T>::mfma32(vf, pf[kk], o[dt]);
            }
        }

        if constexpr (!DMA) { if (more) store_kv(BUF ^ 1); }
        __syncthreads();                                 // with DMA in flight this also drains vmcnt (next tile landed)
    };
    if constexpr (!PIPE) {
        for (int kt = 0; kt < nkt; kt += 2) {
            tile(kt, std::integral_constant<int, 0>{});
            if (kt + 1 < nkt) tile(kt + 1, std::integral_constant<int, 1>{});
        }
    } else {
        // Software pipeline across KV tiles: the QK^T MFMAs of tile t+1 are issued BEFORE the softmax VALU work of tile t,
        // so the matrix pipe computes the next scores while this wave exponentiates the current ones (within one wave MFMA
        // and VALU only overlap when independent work is adjacent in program order).  K is therefore staged one tile
        // ahead of V: iteration t reads K(t+1) [buffer BUF^1] and V(t) [buffer BUF], and fills K(t+2) -> BUF, V(t+1) -> BUF^1.
        auto qk = [&](const char* kb, f32x16 (&s)[2]) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
#pragma unroll
                for (int r = 0; r < 16; ++r) s[t][r] = 0.f;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const vec8 kf = *(const vec8*)(kb + koff[ks] + t * 32 * (C::LDK * 2));
                    s[t] = T16<T>::mfma32(kf, qf[ks], s[t]);
                }
            }
        };
        auto tile_p = [&](int kt, auto buf_c, f32x16 (&s)[2], f32x16 (&sn)[2]) {
            constexpr int BUF = decltype(buf_c)::value;
            const bool has1 = kt + 1 < nkt, has2 = kt + 2 < nkt;
            if (has2) load_one(kt + 2, true, false);
            if (has1) load_one(kt + 1, false, true);
            const char* vb = smem + BUF * (KT_BYTES + VT_BYTES) + KT_BYTES;
            if (has1) qk(smem + (BUF ^ 1) * (KT_BYTES + VT_BYTES), sn);
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
            if (__builtin_amdgcn_ballot_w64(m_new != m_run) != 0) {
                const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
                l_run *= alpha;
#pragma unroll
                for (int i = 0; i < DT; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
                m_run = m_new;
            }
            float psum = 0.f;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {                 // exp of one 16-key step, then its PV MFMAs: VALU of step kk+1
                vec8 pf;                                     // issues under the MFMAs of step kk
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float p = __builtin_amdgcn_exp2f(fmaf(s[kk >> 1][(kk & 1) * 8 + j], a.scale_log2e, -m_run));
                    psum += p;
                    pf[j] = T16<T>::from_f32(p);
                }
                const int krow = (kk >> 1) * 32 + (kk & 1) * 16;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    const char* a0 = vb + voff_dt[dt] + krow * (C::LDV * 2);
                    const char* a1 = a0 + 8 * (C::LDV * 2);
                    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(a0));
                    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(a1));
                    s16x8 both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    o[dt] = T16<T>::mfma32(bitcast<vec8>(both), pf, o[dt]);
                }
            }
            l_run += psum;
            if (has2 || has1) store_one(BUF, BUF ^ 1, has2, has1);
            __syncthreads();
        };
        // prologue: K(0), V(0) are staged (above); K(1) goes to buffer 1 before the first score tile is needed
        if (nkt > 1) {
            load_one(1, true, false);
            store_one(1, 1, true, false);
            __syncthreads();
        }
        f32x16 sA[2], sB[2];
        qk(smem, sA);
        for (int kt = 0; kt < nkt; kt += 2) {
            tile_p(kt, std::integral_constant<int, 0>{}, sA, sB);
            if (kt + 1 < nkt) tile_p(kt + 1, std::integral_constant<int, 1>{}, sB, sA);
        }
    }

    // ---- epilogue: lane (q, h) holds O[q][dt*32 + 8*(r>>2) + 4*h + (r&3)] -------------------------------
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
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

template <typename T, int D, bool TR, int NW, bool ABL = false, bool PIPE = false, bool DMA = false>
static int launch_attn(AttnArgs a, int batch, int heads, hipStream_t s) {
    a.nqb = ceil_div(a.S, NW * 32);
    typedef AttnCfg<D> C;
    const size_t lds = 2 * (ATT_BK * C::LDK * 2 + ATT_BK * C::LDV * 2);
    static std::atomic<unsigned> done_mask{0};   // per device: the attribute belongs to the device's copy of the kernel
    const int cur_dev = current_device();
    if (!device_done(done_mask, cur_dev)) {
        (void)hipFuncSetAttribute((const void*)attn_kernel<T, D, TR, NW, ABL, PIPE, DMA>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        mark_device_done(done_mask, cur_dev);
    }
    dim3 grid(ceil_div(a.S, NW * 32) * heads * batch);
    hipLaunchKernelGGL((attn_kernel<T, D, TR, NW, ABL, PIPE, DMA>), grid, dim3(NW * 64), lds, s, a);
    return (int)hipGetLastError();
}

int attention_sp_dispatch(const AttnArgs& a, int batch, int heads, int head_dim, int dtype, int nw, bool pv16, hipStream_t s);   // attention_sp.hip
int attention_ab_dispatch(const AttnArgs& a, int batch, int heads, int dtype, int waves, int form, bool qs, hipStream_t s);                 // attention_ab.hip
int attention_w4_dispatch(const AttnArgs& a, int batch, int heads, int dtype, int form, hipStream_t s);                                      // attention_w4.hip

}  // namespace vitg

using namespace vitg;

static int attention_entry(const void* q, const void* k, const void* v, void* o, int batch, int heads, int kv_heads,
                           int S, int Sk, const int* sk_dev, int head_dim, long q_bs, long k_bs, long v_bs, long o_bs, int q_rs, int k_rs,
                           int v_rs, int o_rs, int q_hs, int k_hs, int v_hs, int o_hs, float scale, int dtype,
                           int variant, void* stream) {
    if (batch <= 0 || heads <= 0 || S <= 0) return 0;
    if (Sk <= 0) return VITG_ERR_SHAPE;
    if (kv_heads <= 0 || heads % kv_heads) return VITG_ERR_SHAPE;
    if (head_dim != 72 && head_dim != 128) return VITG_ERR_SHAPE;
    if ((q_rs | k_rs | v_rs | o_rs | q_hs | k_hs | v_hs | o_hs) % 4) return VITG_ERR_ALIGN;
    if ((q_rs | k_rs | v_rs | q_hs | k_hs | v_hs) % 8 || (q_bs | k_bs | v_bs) % 8) return VITG_ERR_ALIGN;
    if (dtype != DT_BF16 && dtype != DT_F16) return VITG_ERR_DTYPE;
    // the shipped kernels stage K / V through 32-bit buffer descriptors with per-lane byte offsets computed in 32-bit SIGNED arithmetic
    // (csrc/attn_common.h: lds_dma_rsrc; attention_sp.hip / attention_ab.hip: dk_offb + kt * tile bytes): one (batch, kv head) slice - INCLUDING the rows
    // the last 64-row tile stages past the last key, which must land in the descriptor's zero-filled range and not wrap - has to end below 2 GiB
    // (the largest slice of the path is 118 MB)
    if ((((long)((Sk + 63) / 64 * 64 - 1) * (k_rs > v_rs ? k_rs : v_rs) + head_dim) * 2) >= (1L << 31)) return VITG_ERR_SHAPE;
    AttnArgs a;
    a.q = q; a.k = k; a.v = v; a.o = o;
    a.q_bs = q_bs; a.k_bs = k_bs; a.v_bs = v_bs; a.o_bs = o_bs;
    a.q_rs = q_rs; a.k_rs = k_rs; a.v_rs = v_rs; a.o_rs = o_rs;
    a.q_hs = q_hs; a.k_hs = k_hs; a.v_hs = v_hs; a.o_hs = o_hs;
    a.S = S; a.Sk = Sk; a.sk_dev = sk_dev; a.group = heads / kv_heads; a.kv_heads = kv_heads; a.nqb = 0;
    a.scale_log2e = scale * 1.4426950408889634f;
    hipStream_t s = (hipStream_t)stream;
    int rc;
    a.ablate = variant >> 4;             // profiling ablations (WRONG results): only in a process that set VITG_PROFILING=1
    variant &= 15;
    if (a.ablate) {
#if VITG_PROFILING_BUILD
        static const bool profiling = [] { const char* e = getenv("VITG_PROFILING"); return e && atoi(e) != 0; }();
        if (!profiling) return VITG_ERR_ARG;
#else
        return VITG_ERR_ARG;              // the product library has no ablation instantiations
#endif
    }
    const bool tr = (variant & 1) == 0;  // variant bit 0 = scalar LDS reads of V (cross-check path for the tr-read)
    // variant bit 1: q holds Q * scale * log2(e) already (the producing projection carries the factor in its weights; `scale` is then only
    // documentation): every kernel exponentiates K Q^T in base 2 with a unit scale, and the default hd-72 kernel takes the scores from the matrix
    // pipe ready for exp2 (attention_ab.hip, QS)
    const bool q_prescaled = (variant & 2) != 0;
    if (q_prescaled) a.scale_log2e = 1.0f;
    // Defaults from same-call A/B (profiles/r01_attention.md): 8 waves / 256 queries per block when that still yields enough
    // blocks to fill the chip, else 4; K/V staging by LDS-DMA for hd 72 (+16 %: no staging registers -> 3-4 waves/SIMD) and
    // through registers for hd 128 (DMA measured 3-6 % slower there).  attn_waves / attn_dma (1 = DMA, 2 = registers) pin them.
    const int nw_env = RESEARCH(attn_waves);
    int nw = nw_env ? nw_env : 8;
    if (!nw_env) {
        // 256-query blocks run one per CU, 128-query blocks two per CU: with r = blocks / slots rounds of work the chip is r / ceil(r)
        // busy.  The block size with the fuller last round wins (measured on the per-rank shapes of the context-parallel LLM,
        // tools/attn_cp_shapes.py: 1660 queries x 28 heads -> 196 big blocks on 256 CUs = 0.77 beat 364 small ones on 512 slots = 0.71,
        // 344 vs 371 us; the previous rule "fewer than 384 big blocks -> small blocks" picked the slower one there; 6460 queries: 0.95 vs
        // 0.93, 1120 vs 1203 us); a tie goes to the small blocks (3260 queries: 644 vs 684 us), long launches to the big ones.
        const int cus = device_cu_count(current_device());
        const double r8 = (double)ceil_div(S, 256) * heads * batch / cus, r4 = (double)ceil_div(S, 128) * heads * batch / (2.0 * cus);
        const double e8 = r8 / ceil(r8), e4 = r4 / ceil(r4);
        if (r8 < 4.0 && e4 >= e8) nw = 4;
    }
#if VITG_PROFILING_BUILD
    if (a.ablate && tuning().attn_sp != 1) {   // profiling instantiations (fp16 only)
        if (dtype != DT_F16) return VITG_ERR_ARG;
        rc = head_dim == 72 ? launch_attn<f16_t, 72, true, 4, true>(a, batch, heads, s) : launch_attn<f16_t, 128, true, 8, true>(a, batch, heads, s);
        return rc == 0 ? 0 : VITG_ERR_LAUNCH;
    }
#endif
    // hd 128, profiling library only since round 6: attn_sp = 4 selects the one-wave-per-SIMD kernel (attention_w4.hip) - +4.9 % over the 8-wave
    // software-pipelined kernel when the two alternate in isolation, but 3.7 % SLOWER inside the step (2.49 against 2.40 ms per launch, step 635.1 against
    // 630.9 ms in same-call alternation of bench.py, gpurun_out/r05_step_ab).  It lost its A/B where it runs, so - like every other variant that did - it
    // left the product library; 5 / 6 / 7 / 100+ are its research forms.
#if VITG_PROFILING_BUILD
    {
        const int sp = tuning().attn_sp;
        if (tr && head_dim == 128 && !a.ablate && (sp == 4 || sp == 5 || sp == 6 || sp == 7 || sp >= 100)) {
            rc = attention_w4_dispatch(a, batch, heads, dtype, sp == 4 ? 0 : sp, s);
            return rc == 0 ? 0 : VITG_ERR_LAUNCH;
        }
    }
#endif
#define GO(T, D) (!tr ? launch_attn<T, D, false, 4>(a, batch, heads, s) : nw == 4 ? launch_attn<T, D, true, 4>(a, batch, heads, s) : launch_attn<T, D, true, 8>(a, batch, heads, s))
#if !VITG_PROFILING_BUILD
    // PRODUCT library: the measured defaults plus ONE cross-check set (attn_sp = 2 / attn_ab = 2: the per-tile kernels of this file).
    //   hd 128 -> software-pipelined 8-wave loop (attention_sp.hip; 1023 vs 916 TF/s on the LLM shape against the per-tile loop);
    //   hd 72  -> half-tile-out-of-phase kernel (attention_ab.hip) when 256-query blocks fill the chip, else the per-tile loop with LDS-DMA staging.
    // The variants that lost their A/B (hd-72 cross-tile pipelined loop, hd-72 software-pipelined loop, DMA staging at hd 128, 16-wave blocks, PV on
    // 32x32x16, VALU row sums, hd-128 PV on 16x16x32) are instantiated in the profiling library only.
    if (tr && head_dim == 128 && tuning().attn_sp == 0) {
        rc = attention_sp_dispatch(a, batch, heads, head_dim, dtype, nw, false, s);
        return rc == 0 ? 0 : VITG_ERR_LAUNCH;
    }
    if (tr && head_dim == 72 && tuning().attn_ab == 0 && nw == 8) {
        rc = attention_ab_dispatch(a, batch, heads, dtype, 8, 0, q_prescaled, s);
        return rc == 0 ? 0 : VITG_ERR_LAUNCH;
    }
    if (tr && head_dim == 72)
        rc = dtype == DT_BF16 ? (nw == 4 ? launch_attn<bf16_t, 72, true, 4, false, false, true>(a, batch, heads, s) : launch_attn<bf16_t, 72, true, 8, false, false, true>(a, batch, heads, s))
                              : (nw == 4 ? launch_attn<f16_t, 72, true, 4, false, false, true>(a, batch, heads, s) : launch_attn<f16_t, 72, true, 8, false, false, true>(a, batch, heads, s));
    else if (dtype == DT_BF16) rc = head_dim == 72 ? GO(bf16_t, 72) : GO(bf16_t, 128);
    else rc = head_dim == 72 ? GO(f16_t, 72) : GO(f16_t, 128);
    return rc == 0 ? 0 : VITG_ERR_LAUNCH;
#else
    // hd 72: optional cross-tile software-pipelined variant (QK^T of tile t+1 issued before the softmax of tile t).  Measured
    // equal to the plain loop (549 vs 555 TF/s, profiles/r01_attention.md), so it is opt-in: attn_pipe = 1
    // Software-pipelined loop (attention_sp.hip): default for hd 128 (LLM shape 1023 vs 916 TF/s); for hd 72 the per-tile
    // kernel at 4 waves/SIMD is still ahead (630 vs 597 TF/s), profiles/r01_attention.md.  attn_sp = 1 / 2 pins it on / off.
    if (tr && (tuning().attn_sp == 1 || ((tuning().attn_sp == 0 || tuning().attn_sp == 3) && head_dim == 128 && !a.ablate))) {
        rc = attention_sp_dispatch(a, batch, heads, head_dim, dtype, nw, tuning().attn_sp == 3, s);
        return rc == 0 ? 0 : VITG_ERR_LAUNCH;
    }
    // hd 72 default: the half-tile-out-of-phase kernel (attention_ab.hip, +5 % over the per-tile loop in same-call A/B) whenever
    // 256-query blocks fill the chip and no other variant is pinned; attn_ab = 1 / 2 forces it on / off
    const bool ab_auto = tuning().attn_ab == 0 && !nw_env && nw == 8 && !RESEARCH(attn_pipe) && !RESEARCH(attn_dma) && tuning().attn_sp != 1;
    if (tr && head_dim == 72 && !a.ablate && (tuning().attn_ab == 1 || tuning().attn_ab >= 3 || ab_auto) && tuning().attn_sp != 1) {
        rc = attention_ab_dispatch(a, batch, heads, dtype, tuning().attn_ab == 3 ? 16 : 8, tuning().attn_ab >= 4 ? tuning().attn_ab : 0, q_prescaled, s);
        return rc == 0 ? 0 : VITG_ERR_LAUNCH;
    }
    const bool pipe72 = tr && head_dim == 72 && nw == 4 && RESEARCH(attn_pipe) == 1;
    const bool dma = tr && !pipe72 && (RESEARCH(attn_dma) == 1 || (RESEARCH(attn_dma) == 0 && head_dim == 72));
    if (pipe72) rc = dtype == DT_BF16 ? launch_attn<bf16_t, 72, true, 4, false, true>(a, batch, heads, s)
                                      : launch_attn<f16_t, 72, true, 4, false, true>(a, batch, heads, s);
    else if (dma) {
#define GD(T) (head_dim == 72 ? (nw == 4 ? launch_attn<T, 72, true, 4, false, false, true>(a, batch, heads, s) : launch_attn<T, 72, true, 8, false, false, true>(a, batch, heads, s)) \
                              : (nw == 4 ? launch_attn<T, 128, true, 4, false, false, true>(a, batch, heads, s) : launch_attn<T, 128, true, 8, false, false, true>(a, batch, heads, s)))
        rc = dtype == DT_BF16 ? GD(bf16_t) : GD(f16_t);
#undef GD
    } else if (dtype == DT_BF16) rc = head_dim == 72 ? GO(bf16_t, 72) : GO(bf16_t, 128);
    else rc = head_dim == 72 ? GO(f16_t, 72) : GO(f16_t, 128);
    return rc == 0 ? 0 : VITG_ERR_LAUNCH;
#endif
#undef GO
}

extern "C" int vitg_attention(const void* q, const void* k, const void* v, void* o, int batch, int heads, int kv_heads,
                              int S, int Sk, int head_dim, long q_bs, long k_bs, long v_bs, long o_bs, int q_rs, int k_rs,
                              int v_rs, int o_rs, int q_hs, int k_hs, int v_hs, int o_hs, float scale, int dtype,
                              int variant, void* stream) {
    return attention_entry(q, k, v, o, batch, heads, kv_heads, S, Sk, nullptr, head_dim, q_bs, k_bs, v_bs, o_bs, q_rs, k_rs, v_rs, o_rs,
                           q_hs, k_hs, v_hs, o_hs, scale, dtype, variant, stream);
}

extern "C" int vitg_attention_dyn(const void* q, const void* k, const void* v, void* o, int batch, int heads, int kv_heads,
                                  int S, int Sk_max, const int* kv_len_dev, int head_dim, long q_bs, long k_bs, long v_bs, long o_bs,
                                  int q_rs, int k_rs, int v_rs, int o_rs, int q_hs, int k_hs, int v_hs, int o_hs, float scale,
                                  int dtype, int variant, void* stream) {
    if (!kv_len_dev) return VITG_ERR_ARG;
    return attention_entry(q, k, v, o, batch, heads, kv_heads, S, Sk_max, kv_len_dev, head_dim, q_bs, k_bs, v_bs, o_bs, q_rs, k_rs, v_rs,
                           o_rs, q_hs, k_hs, v_hs, o_hs, scale, dtype, variant, stream);
}

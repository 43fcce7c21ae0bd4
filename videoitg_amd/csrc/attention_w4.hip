// One-wave-per-SIMD software-pipelined flash attention forward for gfx950, head dim 128 (same math, LDS layouts and transposed
// formulation as attention_sp.hip; read that file's header first).
//
// What changes against attention_sp.hip.  There a block is 8 waves x 32 queries, two waves per SIMD inside 256 VGPRs: every LDS
// fragment (one ds_read_b128 of K, two ds_read_b64_tr_b16 of V) feeds ONE 32-cycle MFMA and can only be requested one MFMA ahead
// (three fragment registers are all the 256-register budget leaves), so each wave waits on LDS latency at every step and relies
// on its SIMD partner - which runs the same stream - to fill the hole: MFMA busy 54 % (profiles/r04_mfma_util.md).
// Here a block is 4 waves x 64 queries (the same 256 queries, the same 64 KiB of LDS, half the LDS read traffic), one wave per
// SIMD with the whole 512-register file:
//   * a wave owns TWO 32-query blocks; each K / V^T fragment is read once and feeds two MFMAs (64 pipe cycles per fragment);
//   * fragments are requested PF steps (= 2 PF MFMAs, >= 128 pipe cycles) ahead, more than the LDS latency, so the stream needs
//     no partner to hide it;
//   * per MFMA gap the wave issues ~4 softmax VALU + <= 1 DS instruction - inside the 5 single-issue slots a 32-cycle MFMA
//     covers (MI355X_MICROARCH.md, per-instruction constants).
// One barrier interval t holds, as in attention_sp.hip,  S(t+1) = K(t+1) Q^T  and  O += V(t-1)^T P(t-1)^T  on the matrix pipe and
// P(t) = exp2(S(t) c - m) on the VALU, for both query blocks of the wave; lazy rescale (2^8) per query block.
#include "attn_common.h"
#include "../../include/videoitg_hip.h"
#include <type_traits>

namespace vitg {

// Lazy rescale threshold: the running max moves only when some query of the block saw it grow by more than 2^12 (the 8-wave kernel: 2^8).  P <= 2^12 is
// exact in real arithmetic and well inside fp16 / bf16 range; the rescale is the one thing this kernel does much worse than the 8-wave one (O lives in the
// accumulator file: read, multiply, write back = 384 instructions per query block with nobody else on the SIMD), and on peaky scores it fired often
// enough to turn +5 % into -9 % (tools/r05/attn_data_dep.py).
constexpr float W4_LAZY_RESCALE_LOG2 = 12.0f;

// The MFMAs are inline assembly so that their register FILES are this file's choice: with 512 registers per wave hipcc selects the
// accumulator-file form for every MFMA builtin, i.e. the score tiles would land in AGPRs and be copied out for the softmax (64
// v_accvgpr_read per tile and query block - measured in the first build of this kernel).  Here O (128 registers) and the Q fragments (64)
// live in the accumulator file, where only MFMAs touch them; scores, P, K / V^T fragments stay in the arch VGPRs.
// Hazards hipcc does not pad for an asm statement (cdna_hip_programming.md 5.7 item 2): an MFMA's D -> a non-MFMA reader needs 12 wait
// states for these 8-pass shapes.  Inside the interval every such reader is >= 2 MFMAs away by construction; the three places where it is
// not (after the prologue's S(0), before the rescale of a drain, before the epilogue) carry a settle statement.
// The *_pad forms open with s_nop 1: hipcc's own copies into an operand (v_accvgpr_mov / v_accvgpr_write when it re-homes a tuple, as it does on the
// one-tile path) may sit directly in front of the statement, and a VALU / accvgpr write -> MFMA read needs 2 wait states.  They are used wherever the
// statement is not inside the main loop; the main loop's statements are audited in the built code instead (tools/isa_mfma_hazard_audit.py).
template <typename T> struct W4Mfma;
template <> struct W4Mfma<f16_t> {
    typedef f16x8 vec8;
    static __device__ __forceinline__ void s_init_pad(f32x16& s, vec8 k, vec8 q) { asm("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(s) : "v"(k), "a"(q)); }
    static __device__ __forceinline__ void s_acc_pad(f32x16& s, vec8 k, vec8 q) { asm("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(s) : "v"(k), "a"(q)); }
    static __device__ __forceinline__ void o_acc_pad(f32x16& o, vec8 v, vec8 p) { asm("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(o) : "v"(v), "v"(p)); }
    static __device__ __forceinline__ void s_init(f32x16& s, vec8 k, vec8 q) { asm("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(s) : "v"(k), "a"(q)); }
    static __device__ __forceinline__ void s_acc(f32x16& s, vec8 k, vec8 q) { asm("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(s) : "v"(k), "a"(q)); }
    static __device__ __forceinline__ void o_acc(f32x16& o, vec8 v, vec8 p) { asm("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(o) : "v"(v), "v"(p)); }
};
template <> struct W4Mfma<bf16_t> {
    typedef bf16x8 vec8;
    static __device__ __forceinline__ void s_init_pad(f32x16& s, vec8 k, vec8 q) { asm("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(s) : "v"(k), "a"(q)); }
    static __device__ __forceinline__ void s_acc_pad(f32x16& s, vec8 k, vec8 q) { asm("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(s) : "v"(k), "a"(q)); }
    static __device__ __forceinline__ void o_acc_pad(f32x16& o, vec8 v, vec8 p) { asm("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(o) : "v"(v), "v"(p)); }
    static __device__ __forceinline__ void s_init(f32x16& s, vec8 k, vec8 q) { asm("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(s) : "v"(k), "a"(q)); }
    static __device__ __forceinline__ void s_acc(f32x16& s, vec8 k, vec8 q) { asm("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(s) : "v"(k), "a"(q)); }
    static __device__ __forceinline__ void o_acc(f32x16& o, vec8 v, vec8 p) { asm("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(o) : "v"(v), "v"(p)); }
};
// The settle statements NAME the registers they guard: the MFMA statements are not volatile (hipcc may move them), and only a data dependence keeps
// every producer above the pad and every reader below it (a bare s_nop was passed by the last MFMA of the prologue in the first build).
__device__ __forceinline__ void settle_s(f32x16 (&s0)[2], f32x16 (&s1)[2]) { asm volatile("s_nop 15" : "+v"(s0[0]), "+v"(s0[1]), "+v"(s1[0]), "+v"(s1[1])); }
__device__ __forceinline__ void settle_o(f32x16 (&o)[2][4]) {
    asm volatile("s_nop 15" : "+a"(o[0][0]), "+a"(o[0][1]), "+a"(o[0][2]), "+a"(o[0][3]), "+a"(o[1][0]), "+a"(o[1][1]), "+a"(o[1][2]), "+a"(o[1][3]));
}

template <typename T, int PF, int ABL = 0, bool PP = true, int DMA0 = 8>      // DMA0: first step that issues a staging piece; PP: the tt = 1 score tiles ping-pong between two register sets (see interval); ABL: timing ablations (WRONG results; profiling build only): 1 no exponent stream, 2 no K/V staging, 4 no MFMAs, 8 no fragment reads, 16 no barrier
__global__ __launch_bounds__(256, 1) void attn_w4_kernel(AttnArgs a) {
    adopt_device_key_count(a);
    constexpr int D = 128, NW = 4, QB = 2;
    typedef typename T16<T>::vec8 vec8;
    typedef AttnCfg<D> C;
    constexpr int KS = C::DK / 16;           // QK^T k-steps per 32-key half
    constexpr int DT = C::DV / 32;           // PV d-tiles of 32 rows
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
    const int q0 = qb * (NW * 32 * QB) + wave * (32 * QB);

    const T* Q = (const T*)a.q + b * a.q_bs + (long)head * a.q_hs;
    const T* K = (const T*)a.k + b * a.k_bs + (long)kvh * a.k_hs;
    const T* V = (const T*)a.v + b * a.v_bs + (long)kvh * a.v_hs;

    // ---- Q fragments (B operand): lane (q = ql, h) of query block u holds d = 16 ks + 8 h .. +7 ---------------------------
    vec8 qf[QB][KS];
#pragma unroll
    for (int u = 0; u < QB; ++u) {
        const T* qp = Q + (long)min(q0 + 32 * u + ql, a.S - 1) * a.q_rs;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[u][ks] = *(const vec8*)(qp + ks * 16 + h * 8);
    }
    // into the accumulator file HERE (the copies hipcc makes for an "a" operand are not padded against the asm MFMA that reads them)
#pragma unroll
    for (int u = 0; u < QB; ++u)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+a"(qf[u][ks]));

    // ---- LDS-DMA staging through BUFFER loads: 4 K pieces + 4 V pieces of 1 KiB per wave and tile ----------------------------
    // The descriptor's extent ends with the last key row, so rows past Sk (the tail of the last tile, or a K tile past the end) read as
    // zeros in hardware - no clamp, no tail branch, 3 instructions per piece (m0, the load, nothing else) - and each piece is issued from
    // its own step of the interleaved stream (one LDS-DMA instruction costs the issuing wave 60-180 cycles, MI355X_MICROARCH.md; eight of them
    // back to back at the head of the interval were 17 % of this kernel's time with nobody else on the SIMD to fill them).
    constexpr int K_CPR = C::LDK * 2 / 16, V_CPR = C::LDV * 2 / 16;
    constexpr int K_PW = K_CPR / NW, V_PW = V_CPR / NW;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    // descriptor words by hand (an "s" operand of the asm statements below): base, stride 0, extent in bytes, raw 32-bit-offset format
    auto uniform_rsrc = [&](const T* base, int rs) -> u32x4 {
        const uintptr_t pb = (uintptr_t)base;
        u32x4 r;
        r[0] = __builtin_amdgcn_readfirstlane((unsigned)pb);
        r[1] = __builtin_amdgcn_readfirstlane((unsigned)(pb >> 32)) & 0xffffu;
        r[2] = __builtin_amdgcn_readfirstlane((unsigned)(((long)(a.Sk - 1) * rs + D) * 2));
        r[3] = 0x00020000u;
        return r;
    };
    const u32x4 k_rsrc = uniform_rsrc(K, a.k_rs), v_rsrc = uniform_rsrc(V, a.v_rs);
    auto dma_src = [&](int ii, int cpr, int rs, bool is_k) -> int {
        const int j = ii * 64 + lane;
        int row = j / cpr, c = j % cpr;
        c = is_k ? (c ^ (row & 15)) : (c ^ ((row & 3) << 2));
        return (row * rs + c * 8) * 2;
    };
    int dk_off[K_PW], dv_off[V_PW];
#pragma unroll
    for (int i = 0; i < K_PW; ++i) dk_off[i] = dma_src(wave_u + NW * i, K_CPR, a.k_rs, true);
#pragma unroll
    for (int i = 0; i < V_PW; ++i) dv_off[i] = dma_src(wave_u + NW * i, V_CPR, a.v_rs, false);
    const int k_tile_bytes = ATT_BK * a.k_rs * 2, v_tile_bytes = ATT_BK * a.v_rs * 2;
    // piece j of interval t: j < 4 -> V(t) piece j into V buffer `buf`; j >= 4 -> K(kt) piece j - 4 into K buffer `buf` (a kt past the last
    // tile reads zeros into a dead buffer)
    // The loads are inline assembly because hipcc orders every later LDS read behind an LDS-DMA builtin it cannot disambiguate (s_waitcnt
    // vmcnt(0) in front of each ds_read - seen in the listing of this kernel with the builtin, and in front of the first V^T read of every
    // interval of attention_sp.hip): the wave would wait for its own staging of the NEXT tile before reading THIS one.  hipcc does not
    // count these loads; the interval ends with an explicit vmcnt(0) in front of its barrier (dma_drain).
    const unsigned lds_k0 = (unsigned)(uintptr_t)smem + wave_u * 1024u, lds_v0 = lds_k0 + V_BASE;
    auto dma16 = [&](const u32x4& rsrc, unsigned lds_dst, int voff, int soff) {
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                     :: "s"(lds_dst), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
    };
    auto dma_drain = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };
    auto issue_piece = [&](int j, int vt, int kt, int buf) {
        // the tile offset rides in the VGPR offset: the hardware's range check covers voffset (+ the immediate), NOT soffset
        if (j < V_PW) dma16(v_rsrc, lds_v0 + buf * VT_BYTES + NW * j * 1024, dv_off[j] + vt * v_tile_bytes, 0);
        else dma16(k_rsrc, lds_k0 + buf * KT_BYTES + NW * (j - V_PW) * 1024, dk_off[j - V_PW] + kt * k_tile_bytes, 0);
    };
    constexpr int NPIECE = K_PW + V_PW;
    auto issue_k = [&](int kt, int buf) {
#pragma unroll
        for (int j = V_PW; j < NPIECE; ++j) issue_piece(j, 0, kt, buf);
    };

    // lane-dependent LDS read bases; inside the loop every fragment address is one of these + an immediate
    const char* kbase[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) kbase[ks] = smem + k_lds_off<D>(ql, ks * 2 + h);
    const int voff = (4 * h + ((lane & 15) >> 2)) * (C::LDV * 2) + ((((lane >> 4) & 1) * 16) + 4 * (lane & 3)) * 2;
    const char* vbase[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) vbase[dt] = smem + V_BASE + voff + ((dt ^ ((lane & 15) >> 2)) * 64);

    // accumulators: o[u][dt][r] = O[q of block u][32 dt + 8 (r >> 2) + 4 h + (r & 3)]
    f32x16 o[QB][DT];
#pragma unroll
    for (int u = 0; u < QB; ++u)
#pragma unroll
        for (int i = 0; i < DT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[u][i][r] = 0.f;
    float m_run[QB], l_run[QB];
#pragma unroll
    for (int u = 0; u < QB; ++u) { m_run[u] = -1e30f; l_run[u] = 0.f; }
    const int nkt = ceil_div(a.Sk, ATT_BK);

    auto k_frag = [&](int kbuf, int tt, int ks) -> vec8 {
        return *(const vec8*)(kbase[ks] + (kbuf * KT_BYTES + tt * 32 * (C::LDK * 2)));
    };
    auto v_frag = [&](int vbuf, int kk, int dt) -> vec8 {
        const char* a0 = vbase[dt] + (vbuf * VT_BYTES + ((kk >> 1) * 32 + (kk & 1) * 16) * (C::LDV * 2));
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(a0));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(a0 + 8 * (C::LDV * 2)));
        return bitcast<vec8>(__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    };
    // S = K Q^T for one 64-key tile (prologue only: no overlap)
    auto qk_mfma = [&](int kbuf, f32x16 (&s0)[QB], f32x16 (&s1)[QB]) {
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const vec8 kf = k_frag(kbuf, tt, ks);
#pragma unroll
                for (int u = 0; u < QB; ++u) {
                    f32x16& d = tt == 0 ? s0[u] : s1[u];
                    if (ks == 0) W4Mfma<T>::s_init_pad(d, kf, qf[u][0]);
                    else W4Mfma<T>::s_acc_pad(d, kf, qf[u][ks]);
                }
            }
        settle_s(s0, s1);
    };
    // O += V^T P^T for one tile (drain after the last interval)
    auto pv_mfma = [&](int vbuf, vec8 (&p)[QB][4]) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const vec8 vf = v_frag(vbuf, kk, dt);
#pragma unroll
                for (int u = 0; u < QB; ++u) W4Mfma<T>::o_acc_pad(o[u][dt], vf, p[u][kk]);
            }
        settle_o(o);
    };

    // One barrier interval.  PAR = t & 1.  s = S(t) on entry and S(t+1) on exit (IN PLACE), p_cur = P(t) (out), p_prv = P(t-1) (in).
    // Step order: the 16 PV(t-1) steps FIRST, the 16 QK^T(t+1) steps second.  The exponent stream retires S(t) tile by tile in the order
    // (u, tt) = (0,0) (1,0) (0,1) (1,1) and is done with both tt = 0 tiles before step 16, with the tt = 1 tiles before step 24 - exactly
    // when the QK^T steps of S(t+1) start writing those registers.  At no point are two score tiles alive, which is what keeps scores + P
    // + fragments inside the arch VGPRs (a second score array put the allocator 60 registers over: first build of this kernel).
    auto interval = [&](int t, auto par_c, auto qk_c, auto pv_c, auto pad_c, f32x16 (&s0)[QB], f32x16 (&s1c)[QB], f32x16 (&s1n)[QB], vec8 (&p_cur)[QB][4],
                        vec8 (&p_prv)[QB][4]) {
        // S(t) tile (u, tt): tt = 0 -> s0[u] (overwritten in place by S(t+1)), tt = 1 -> s1c[u]; S(t+1)'s tt = 1 tiles go to s1n[u] (PP: the other
        // register set, so the exponents of the tt = 1 tiles may run until the interval's last step; !PP: the same registers)
        auto sc = [&](int u, int tt) -> f32x16& { return tt == 0 ? s0[u] : s1c[u]; };
        constexpr int PAR = decltype(par_c)::value;
        constexpr bool PAD = decltype(pad_c)::value;      // outside the main loop: padded MFMA statements (see W4Mfma)
        constexpr bool HAS_QK = decltype(qk_c)::value, HAS_PV = decltype(pv_c)::value;
        if (!HAS_QK && t * ATT_BK + ATT_BK > a.Sk) {   // partial last tile: s[u][tt][r] is key t*64 + tt*32 + 8*(r>>2) + 4*h + (r&3)
#pragma unroll
            for (int u = 0; u < QB; ++u)
#pragma unroll
                for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (t * ATT_BK + tt * 32 + 8 * (r >> 2) + 4 * h + (r & 3) >= a.Sk) sc(u, tt)[r] = -INFINITY;
        }
        constexpr int NQK = HAS_QK ? 2 * KS : 0, NPV = HAS_PV ? 4 * DT : 0, NM = NQK + NPV;
        auto frag = [&](int i) -> vec8 {
            if (i < NPV) return v_frag(PAR ^ 1, i / DT, i % DT);
            const int j = i - NPV;
            return k_frag(PAR ^ 1, j / KS, j % KS);
        };
        auto mfma_step = [&](int i, vec8 f) {
            if (i < NPV) {
#pragma unroll
                for (int u = 0; u < QB; ++u) {
                    if constexpr (PAD) W4Mfma<T>::o_acc_pad(o[u][i % DT], f, p_prv[u][i / DT]);
                    else W4Mfma<T>::o_acc(o[u][i % DT], f, p_prv[u][i / DT]);
                }
            } else {
                const int j = i - NPV, tt = j / KS, ks = j % KS;
#pragma unroll
                for (int u = 0; u < QB; ++u) {
                    f32x16& d = tt == 0 ? s0[u] : s1n[u];
                    if constexpr (PAD) {
                        if (ks == 0) W4Mfma<T>::s_init_pad(d, f, qf[u][0]);
                        else W4Mfma<T>::s_acc_pad(d, f, qf[u][ks]);
                    } else {
                        if (ks == 0) W4Mfma<T>::s_init(d, f, qf[u][0]);
                        else W4Mfma<T>::s_acc(d, f, qf[u][ks]);
                    }
                }
            }
        };
        float mx[QB], m_use[QB];
        uint64_t grew[QB];
        float psum[QB][2];
#pragma unroll
        for (int u = 0; u < QB; ++u) { m_use[u] = m_run[u]; grew[u] = 0; psum[u][0] = psum[u][1] = 0.f; mx[u] = 0.f; }
        auto row_max = [&](int u, int half) {
            if (half == 0) mx[u] = s0[u][0];
#pragma unroll
            for (int r = 0; r < 16; ++r) mx[u] = fmaxf(mx[u], sc(u, half)[r]);
            if (half == 1) {
                const u32x2 sw = __builtin_amdgcn_permlane32_swap(bitcast<uint32_t>(mx[u]), bitcast<uint32_t>(mx[u]), false, false);
                const float m = fmaxf(bitcast<float>(sw[0]), bitcast<float>(sw[1])) * a.scale_log2e;      // scale > 0
                const float m_new = fmaxf(m_run[u], m);
                grew[u] = __builtin_amdgcn_ballot_w64(m_new > m_run[u] + W4_LAZY_RESCALE_LOG2);            // wave-uniform
                m_use[u] = grew[u] ? m_new : m_run[u];
                asm volatile("" : "+v"(m_use[u]));       // opaque: keeps the exponent stream behind the row max (see attention_sp.hip)
            }
        };
        // The exponent stream as a three-stage pipeline over consecutive slices (a wave alone on its SIMD has nobody to fill the
        // fma -> exp -> add / cvt dependency bubbles): slice i scales the pairs of slice i+1, exponentiates its own, and sums / packs those of
        // slice i-1.  e = 0..31: tile (u, tt) = (e >> 3) & 1, e >> 4; scores 2 (e & 7), + 1 of that tile.
        float xs[16 * QB][2];
        auto st_scale = [&](int e) {
            const int u = (e >> 3) & 1, tt = e >> 4, r = (e & 7) * 2;
            // plain C on purpose (no asm): hipcc pads the hazards of instructions it can see - a v_exp_f32 result consumed by the next VALU
            // instruction needs a wait state, and an asm v_add_f32 placed straight behind its v_exp_f32 read stale lanes (second build of this
            // pipeline).  The single-issue forms the stream wants beside MFMAs (MI355X_MICROARCH.md "price of one filler": v_pk_fma_f32 /
            // v_pk_add_f32 cost more than the two plain instructions they replace) come from building this file with -fno-slp-vectorize.
            xs[e][0] = __builtin_fmaf(sc(u, tt)[r], a.scale_log2e, -m_use[u]);
            xs[e][1] = __builtin_fmaf(sc(u, tt)[r + 1], a.scale_log2e, -m_use[u]);
        };
        auto st_exp = [&](int e) {
            xs[e][0] = __builtin_amdgcn_exp2f(xs[e][0]);
            xs[e][1] = __builtin_amdgcn_exp2f(xs[e][1]);
        };
        auto st_pack = [&](int e) {
            const int u = (e >> 3) & 1, tt = e >> 4, r = (e & 7) * 2;
            psum[u][0] += xs[e][0];
            psum[u][1] += xs[e][1];
            asm volatile("" : "+v"(psum[u][0]), "+v"(psum[u][1]));      // opaque: left alone, hipcc sinks all 64 row-sum adds below the last MFMA of the interval
            p_cur[u][tt * 2 + (r >> 3)][r & 7] = T16<T>::from_f32(xs[e][0]);
            p_cur[u][tt * 2 + (r >> 3)][(r & 7) + 1] = T16<T>::from_f32(xs[e][1]);
            if ((e & 3) == 3) asm volatile("" : "+v"(p_cur[u][tt * 2 + (r >> 3)]));   // opaque use: keeps the stream inside its slices
        };
        constexpr int NSM = QB * 2, NE = 16 * QB;
        // full interval: the pairs are spread over slices NSM .. EXP_END-1 (their scale stage runs one slice earlier, their pack stage one
        // later); EXP_END = the step at which the tt = 1 tiles are overwritten
        constexpr bool FULL = NPV > 0 && NQK > 0;
        constexpr int EXP_END = (FULL && !PP) ? NPV + KS : NM;
        // pairs retired through slice i.  Full interval: the tt = 0 tiles (pairs 0..15) go over the PV steps - they must be gone when the first
        // QK^T step overwrites them - and the tt = 1 tiles over the QK^T steps (PP) or up to the step that overwrites them (!PP)
        auto exp_hi = [&](int i) {
            if (i < NSM) return 0;
            if (i >= EXP_END) return NE;
            if (FULL && PP) return i < NPV ? (i + 1 - NSM) * (NE / 2) / (NPV - NSM) : NE / 2 + (i + 1 - NPV) * (NE / 2) / (NM - NPV);
            return (i + 1 - NSM) * NE / (EXP_END - NSM);
        };
        if constexpr (FULL && !PP) {
            static_assert((NPV - NSM) * NE / (EXP_END - NSM) >= NE / 2, "both tt = 0 tiles must be retired before the first QK^T step");
        }
        auto valu_slice = [&](int i) {
            if (i < NSM) row_max(i >> 1, i & 1);
#pragma unroll
            for (int e = exp_hi(i); e < exp_hi(i + 1); ++e) st_scale(e);
#pragma unroll
            for (int e = exp_hi(i - 1); e < exp_hi(i); ++e) st_exp(e);
#pragma unroll
            for (int e = exp_hi(i - 2); e < exp_hi(i - 1); ++e) st_pack(e);
        };
        if constexpr (!HAS_PV) {
            // no PV steps to retire S(t) behind (interval 0, or a one-tile sequence): staging, softmax, then the QK^T steps on their own
            if (!(ABL & 2)) {
#pragma unroll
                for (int j = 0; j < NPIECE; ++j) issue_piece(j, t, t + 2, PAR);
            }
#pragma unroll
            for (int u = 0; u < QB; ++u) { row_max(u, 0); row_max(u, 1); }
#pragma unroll
            for (int e = 0; e < NE; ++e) { st_scale(e); st_exp(e); st_pack(e); }
            if constexpr (HAS_QK) {
                __builtin_amdgcn_sched_barrier(0);
                qk_mfma(PAR ^ 1, s0, s1n);
            }
        } else {
            vec8 fr[PF + 1];
#pragma unroll
            for (int i = 0; i < PF && i < NM; ++i) fr[i] = frag(i);
#pragma unroll
            for (int i = 0; i < NM; ++i) {
                if (i >= (FULL ? DMA0 : 0) && i < (FULL ? DMA0 : 0) + NPIECE && !(ABL & 2)) issue_piece(i - (FULL ? DMA0 : 0), t, t + 2, PAR);       // V(t), K(t+2): read in interval t+1
                if (i + PF < NM && !((ABL & 8) && i + PF > PF + 1)) fr[(i + PF) % (PF + 1)] = frag(i + PF);
                if (!(ABL & 4)) mfma_step(i, fr[i % (PF + 1)]);
                else asm volatile("" :: "v"(fr[i % (PF + 1)]));
                if (!(ABL & 1) || i < NSM) valu_slice(i);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int i = NM; i < EXP_END + 1; ++i)       // (an interval without QK^T steps ends before the pipeline has drained)
                if (!(ABL & 1)) valu_slice(i);
        }
        uint64_t grew_late = grew[0] | grew[1];
        asm volatile("" : "+s"(grew_late));              // opaque copy: no branch threading into the interleaved block
        if (__builtin_expect(grew_late != 0, 0)) {        // after the PV(t-1) MFMAs: P(t-1) was relative to the old max
            if constexpr (!HAS_QK) settle_o(o);            // (with QK^T steps behind them the PV MFMAs are 32 MFMAs away)
#pragma unroll
            for (int u = 0; u < QB; ++u) {
                const float alpha = __builtin_amdgcn_exp2f(m_run[u] - m_use[u]);      // 1 for a block whose max did not move
                l_run[u] *= alpha;
                // one accumulator tuple at a time, pinned by the volatile statements: left to itself hipcc reads all 128 accumulators out at
                // once and spills the interval's live registers around this (rare) block - into the main loop
#pragma unroll
                for (int i = 0; i < DT; ++i) {
                    asm volatile("" : "+a"(o[u][i]));
                    f32x16 w = o[u][i];
#pragma unroll
                    for (int r = 0; r < 16; ++r) w[r] *= alpha;
                    o[u][i] = w;
                    asm volatile("" : "+a"(o[u][i]));
                }
                m_run[u] = m_use[u];
            }
        }
#pragma unroll
        for (int u = 0; u < QB; ++u) l_run[u] += psum[u][0] + psum[u][1];
        dma_drain();                                     // this wave's pieces of K(t+2) / V(t) have landed ...
        if (!(ABL & 16)) __syncthreads();                // ... everybody's have, and everybody is done reading K(t+1) / V(t-1)
    };

    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 1> I1;
    typedef std::true_type Y;
    typedef std::false_type N;

    // ---- prologue: K(0), K(1) in flight; S(0) computed without overlap ----------------------------------------------------
    issue_k(0, 0);
    issue_k(1, 1);                                     // (zeros when there is no second tile)
    dma_drain();
    __syncthreads();
    f32x16 s0[QB], s1A[QB];
    f32x16 s1B_[PP ? QB : 1];
    f32x16 (&s1B)[QB] = *reinterpret_cast<f32x16 (*)[QB]>(PP ? (void*)&s1B_ : (void*)&s1A);      // !PP: one register set, in place
    vec8 pA[QB][4], pB[QB][4];
    qk_mfma(0, s0, s1A);
    __syncthreads();
    // even intervals write P into pA, odd ones into pB; the loop body is a fixed (odd, even) pair so that no register set changes role
    if (nkt == 1) {
        interval(0, I0{}, N{}, N{}, Y{}, s0, s1A, s1B, pA, pB);
        pv_mfma(0, pA);
    } else {
        interval(0, I0{}, Y{}, N{}, Y{}, s0, s1A, s1B, pA, pB);
        int t = 1;
        for (; t + 2 < nkt; t += 2) {
            interval(t, I1{}, Y{}, Y{}, N{}, s0, s1B, s1A, pB, pA);
            interval(t + 1, I0{}, Y{}, Y{}, N{}, s0, s1A, s1B, pA, pB);
        }
        if (t + 1 < nkt) {
            interval(t, I1{}, Y{}, Y{}, Y{}, s0, s1B, s1A, pB, pA);
            interval(t + 1, I0{}, N{}, Y{}, Y{}, s0, s1A, s1B, pA, pB);
            pv_mfma(0, pA);
        } else {
            interval(t, I1{}, N{}, Y{}, Y{}, s0, s1B, s1A, pB, pA);
            pv_mfma(1, pB);
        }
    }

    // ---- epilogue: lane (q, h) of block u holds O[q][dt*32 + 8*(r>>2) + 4*h + (r&3)] --------------------------------------
#pragma unroll
    for (int u = 0; u < QB; ++u) {
        const float l_tot = l_run[u] + __shfl_xor(l_run[u], 32, 64);
        const float inv = 1.0f / l_tot;
        const int qrow = q0 + 32 * u + ql;
        if (qrow < a.S) {
            T* op = (T*)a.o + b * a.o_bs + (long)head * a.o_hs + (long)qrow * a.o_rs;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    typename T16<T>::vec4 w;
#pragma unroll
                    for (int e = 0; e < 4; ++e) w[e] = T16<T>::from_f32(o[u][dt][g4 * 4 + e] * inv);
                    *(typename T16<T>::vec4*)(op + dt * 32 + 8 * g4 + 4 * h) = w;
                }
        }
    }
}

template <typename T, int PF, int ABL = 0, bool PP = true, int DMA0 = 8>
static int launch_w4(AttnArgs a, int batch, int heads, hipStream_t s) {
    a.nqb = ceil_div(a.S, 256);
    typedef AttnCfg<128> C;
    const size_t lds = 2 * (ATT_BK * C::LDK * 2 + ATT_BK * C::LDV * 2);
    static std::atomic<unsigned> done_mask{0};
    const int cur_dev = current_device();
    if (!device_done(done_mask, cur_dev)) {
        (void)hipFuncSetAttribute((const void*)attn_w4_kernel<T, PF, ABL, PP, DMA0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        mark_device_done(done_mask, cur_dev);
    }
    hipLaunchKernelGGL((attn_w4_kernel<T, PF, ABL, PP, DMA0>), dim3(a.nqb * heads * batch), dim3(256), lds, s, a);
    return (int)hipGetLastError();
}

// called by vitg_attention (attention.hip): head dim 128, 256-query blocks.  form 0 = the shipped kernel (fragments requested 2 steps ahead, tt = 1
// score tiles ping-pong, staging pieces from step 8 on); profiling library: 5 in-place score tiles | 6 / 7 staging from step 0 / 16 | 100 + bits
// timing ablations (tools/r05/attn_w4_ablate.py)
int attention_w4_dispatch(const AttnArgs& a, int batch, int heads, int dtype, int form, hipStream_t s) {
#if VITG_PROFILING_BUILD
    if (form >= 100 && dtype == DT_F16) {
        switch (form - 100) {
            case 1: return launch_w4<f16_t, 2, 1, true, 8>(a, batch, heads, s);
            case 2: return launch_w4<f16_t, 2, 2, true, 8>(a, batch, heads, s);
            case 4: return launch_w4<f16_t, 2, 4, true, 8>(a, batch, heads, s);
            case 8: return launch_w4<f16_t, 2, 8, true, 8>(a, batch, heads, s);
            case 16: return launch_w4<f16_t, 2, 16, true, 8>(a, batch, heads, s);
            case 18: return launch_w4<f16_t, 2, 18, true, 8>(a, batch, heads, s);
            case 5: return launch_w4<f16_t, 2, 5, true, 8>(a, batch, heads, s);
            case 19: return launch_w4<f16_t, 2, 19, true, 8>(a, batch, heads, s);
            default: return -1;
        }
    }
    if (form == 5) return dtype == DT_BF16 ? launch_w4<bf16_t, 2, 0, false, 8>(a, batch, heads, s) : launch_w4<f16_t, 2, 0, false, 8>(a, batch, heads, s);
    if (form == 6) return dtype == DT_BF16 ? launch_w4<bf16_t, 2, 0, true, 0>(a, batch, heads, s) : launch_w4<f16_t, 2, 0, true, 0>(a, batch, heads, s);
    if (form == 7) return dtype == DT_BF16 ? launch_w4<bf16_t, 2, 0, true, 16>(a, batch, heads, s) : launch_w4<f16_t, 2, 0, true, 16>(a, batch, heads, s);
#endif
    (void)form;
    return dtype == DT_BF16 ? launch_w4<bf16_t, 2, 0, true, 8>(a, batch, heads, s) : launch_w4<f16_t, 2, 0, true, 8>(a, batch, heads, s);
}

}  // namespace vitg

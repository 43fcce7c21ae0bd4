// C[M,N] = A[M,K] * W[N,K]^T with the WEIGHT operand taken straight from global memory into MFMA fragments (no LDS), gfx950.
//
// Why.  profiles/r01_gemm_shapes.md section 9: the 256x256 phased kernel (gemm256.hip, cfg 6) runs at 95 % of the MFMA ceiling
// with its operand stream switched off and at 57-59 % with it - the LDS port is shared, in bursts, between the fragment reads
// of the MFMA phase and the operand writes of the LDS-DMA.  Weights are static, so they can be stored once in the order in which
// a wave consumes them: for every (32-row weight tile, 16-deep k-step) one contiguous 1-KiB block holding, lane by lane, the
// 16 bytes that lane feeds to v_mfma_f32_32x32x16 (vitg_pack_weight_fragments).  A wave then fetches a whole fragment with ONE
// perfectly coalesced global_load_dwordx4 and the weight operand never touches LDS: LDS traffic per 64-deep K-tile drops from
// 256 KiB (64 written + 192 read) to 160 KiB (32 written + 128 read); the texture path carries 96 instead of 64 KiB (the two
// waves that share a weight tile fetch it twice), still under its 64 B/clk.
//
// Structure: 256x256x64 tile, 8 waves (2 M x 4 N, wave tile 128x64), the phased schedule of gemm256.hip (every k-step is a
// fragment-read interval and an MFMA interval, waves 4-7 one barrier behind waves 0-3).  Activations: LDS-DMA into a ring of FOUR
// 32-KiB stages, three K-tiles ahead.  Weights: a ring of eight k-steps of fragments in registers (64 VGPRs), refilled at the end
// of every MFMA interval.  Both streams share the wave's in-order vmcnt queue; every tile issues the same sequence
// (4 DMA pieces, then 2 fragment loads after each of its 4 MFMA intervals), so all waits are constants:
//   fragments of k-step k were issued 8 k-steps ago, 22 operations back;
//   the DMA pieces of tile kt+1 were issued two tiles ago, 30 (waves 4-7, before their last barrier) / 32 (waves 0-3) back.
// The prologue issues the pattern of two tiles without compute to fill both rings; fragment loads past the end of K re-read the
// last k-step (their slots stay allocated, the values are never used) so that the pattern - and with it the counts - stays
// uniform; in the last three tiles, where no DMA is issued any more, the waits fall back to conservative counts.
#include "gemm_common.h"
#include <stdlib.h>
#include <type_traits>

namespace vitg {

constexpr int WF_BN = 256, WF_BK = 64, WF_ASTAGES = 4, WF_RING = 8;

// WF_BM = 256: one 8-wave block per CU (2 M x 4 N waves), the round-1 kernel.
// WF_BM = 128 (round 2, gemm_flags bit25): FOUR waves per block (1 M x 4 N, the same 128x64 wave tile) and TWO independent blocks per
// CU - the activation ring is 4 x 16 KiB per block, so two fit the 160 KiB of LDS now that the weights bypass it.  The two blocks of
// a CU are not in phase, so one block's epilogue / prologue (27 % of an 18-K-tile tile) runs under the other's main loop instead of
// leaving the matrix pipe idle (profiles/r02_mfma_util.md).  Per wave nothing changes (4 DMA pieces + 8 fragment loads per K-tile:
// the counted waits below hold for both shapes); there is simply no second wave group inside the block.
// De-phasing of the two co-resident blocks (WF_BM = 128): identical tiles take identical time, so two blocks that start together
// stay in phase and reach their epilogues together - which is exactly what the second block was meant to avoid.  Every block of the
// initial fill draws a ticket per physical CU (XCC id + HW_ID se/sh/cu bits); the odd ticket sleeps stagger_ns before it starts.
// Later blocks inherit the phase of the block they replace.
__device__ unsigned wf_tickets[4096];

template <typename T, int MODE, int ABL = 0, int WF_BM = 256>      // ABL (profiling, wrong results): 1 no fragment loads, 2 no DMA in steady state
__global__ __launch_bounds__(2 * WF_BM, 2) void gemm_wf_kernel(GemmArgs g) {
    constexpr int NWAVES = WF_BM / 32;
    if constexpr (WF_BM == 128) {
        if (g.stagger_ns > 0 && (int)blockIdx.x < g.stagger_blocks) {
            __shared__ int odd;
            if (threadIdx.x == 0) {
                const unsigned hw = __builtin_amdgcn_s_getreg((7 << 11) | (8 << 6) | 4);      // HW_REG_HW_ID bits [15:8]: cu / sh / se
                const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20);                // HW_REG_XCC_ID bits [3:0]
                odd = atomicAdd(&wf_tickets[((xcc & 15) << 8) | (hw & 255)], 1u) & 1;
            }
            __syncthreads();
            if (odd) {
                const unsigned long long t0 = wall_clock64();                                   // 100 MHz
                const unsigned long long ticks = (unsigned long long)g.stagger_ns / 10;
                while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
            }
        }
    }
    typedef typename T16<T>::vec8 vec8;
    constexpr int MT = 4, NT = 2, KS = 4, ROWB = 128, CPR = 8, RPI = 8;
    constexpr int A_BYTES = WF_BM * ROWB;                // 32 KiB per stage
    constexpr int A_INSTR = A_BYTES / 1024 / NWAVES;     // 4 DMA pieces per wave per K-tile
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;

    int tm, tn;
    block_to_tile(blockIdx.x, g.tiles_m, g.tiles_n, g.group_m, tm, tn);
    const int m0 = tm * WF_BM, n0 = tn * WF_BN;

    // ---- activation stream: LDS-DMA, wave-uniform base + per-lane 32-bit offsets (swizzle on the source address) ------------
    const int drow = lane / CPR, dslot = lane % CPR;
    const char* a_base = (const char*)g.A + (size_t)m0 * g.lda * 2;
    unsigned a_cur[A_INSTR];
#pragma unroll
    for (int j = 0; j < A_INSTR; ++j) {
        const int R = (wave * A_INSTR + j) * RPI + drow;
        a_cur[j] = (unsigned)(min(R, g.M - 1 - m0) * g.lda + ((dslot ^ ((R >> 1) & 7)) << 3)) * 2u;
    }
    auto issue_a = [&](int kt) {
        char* st = smem + (kt & (WF_ASTAGES - 1)) * A_BYTES;
        const char* ab = a_base + (size_t)kt * ROWB;
#pragma unroll
        for (int j = 0; j < A_INSTR; ++j)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ab + a_cur[j]),
                                             (__attribute__((address_space(3))) void*)(st + (wave * A_INSTR + j) * 1024), 16, 0, 0);
    };

    // ---- weight stream: fragment-packed copy, 1 KiB per (32-row tile, k-step); this wave owns tiles ntile0, ntile0 + 1 ---------
    const int nk16 = g.K / 16;
    // wave-uniform (SGPR pair); a wave whose 64 columns lie past N re-reads the last pair of tiles (its results are not stored)
    const int ntile0 = min((n0 + wc * 64) / 32, ceil_div(g.N, 64) * 2 - 2);
    const char* wf_base = (const char*)g.Wf + ((size_t)ntile0 * nk16) * 1024;
    {   // the "s" operands below need the compiler to KNOW this is wave-uniform
        const unsigned long long wb = (unsigned long long)wf_base;
        wf_base = (const char*)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(wb >> 32)) << 32) |
                                (unsigned)__builtin_amdgcn_readfirstlane((int)wb));
    }
    const unsigned wf_lane = lane * 16, wf_tile = (unsigned)nk16 * 1024u;
    vec8 wq[WF_RING][NT];
    // asm loads: the compiler must not track these (it would wait for them in ITS order); the counted waits below do
    auto load_w = [&](int kstep, vec8 (&slot)[NT]) {
        const int kc = min(max(kstep, 0), nk16 - 1);         // past the end (and in the first virtual tile): a harmless re-read
        const unsigned vo = wf_lane + (unsigned)kc * 1024u;
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(slot[0]) : "v"(vo), "s"(wf_base));
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(slot[1]) : "v"(vo + wf_tile), "s"(wf_base));
    };

    const int frow = lane & 31, fh = lane >> 5;
    f32x16 acc[NT][MT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    unsigned a_off[KS];
    {
        const int sw = (frow >> 1) & 7;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) a_off[ks] = (unsigned)(size_t)smem + (wr * 128 + frow) * ROWB + (((ks * 2 + fh) ^ sw) << 4);
    }
    vec8 fa[MT];
    const int nk = g.K / WF_BK;
    const bool grpB = wr == 1;

    // TAIL = one of the last tiles (kt + 3 may be >= nk): no DMA issued any more, so the counted waits fall back to conservative
    // values (vmcnt(14) is valid - only stricter - whenever vmcnt(22) is).  The steady-state instance is branch-free apart from
    // the wave-group test: a taken scalar branch per interval costs more than the waits it selects.
    auto tile = [&](int kt, auto par_c, auto tail_c) {
        constexpr int PAR = decltype(par_c)::value;
        constexpr bool TAIL = decltype(tail_c)::value;
        const unsigned so = (unsigned)((kt & (WF_ASTAGES - 1)) * A_BYTES);
#pragma unroll
        for (int pp = 0; pp < KS; ++pp) {
            // ---- read interval --------------------------------------------------------------------------------------------
#pragma unroll
            for (int i = 0; i < MT; ++i)
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[i]) : "v"(a_off[pp] + so), "n"(i * 32 * ROWB));
            __builtin_amdgcn_sched_barrier(0);
            if (pp == 0 && (!TAIL || kt + 3 < nk) && !(ABL & 2)) issue_a(kt + 3);          // stage of tile kt-1: free since interval 8kt
            __builtin_amdgcn_sched_barrier(0);
            if (pp == KS - 1 && grpB && (!TAIL || kt + 1 < nk)) {             // tile kt+1 landed (my pieces), before my last barrier
                if (TAIL) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(30)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
            if (TAIL) asm volatile("s_waitcnt vmcnt(14) lgkmcnt(0)" ::: "memory");   // fragments of this k-step (issued 8 k-steps ago)
            else asm volatile("s_waitcnt vmcnt(22) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            // ---- MFMA interval -----------------------------------------------------------------------------------------------
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[nt][mt] = T16<T>::mfma32(wq[PAR * 4 + pp][nt], fa[mt], acc[nt][mt]);
            __builtin_amdgcn_sched_barrier(0);
            if (!(ABL & 1)) load_w(kt * KS + pp + WF_RING, wq[PAR * 4 + pp]);               // refill the slot just consumed, 8 k-steps ahead
            __builtin_amdgcn_sched_barrier(0);
            if (pp == KS - 1 && !grpB && (!TAIL || kt + 1 < nk)) {
                if (TAIL) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // prologue = D(0), then the issue pattern of two tiles (D(1) + k-steps 0-3, D(2) + k-steps 4-7) so that the counts above hold
    // from tile 0 on.  (No dummy loads here: an asm load whose result is never read leaves its destination free for the
    // compiler to reuse while the load is still in flight.)
    issue_a(0);
    if (nk > 1) issue_a(1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int pp = 0; pp < KS; ++pp) load_w(pp, wq[pp]);
    __builtin_amdgcn_sched_barrier(0);
    if (nk > 2) issue_a(2);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int pp = 0; pp < KS; ++pp) load_w(KS + pp, wq[KS + pp]);
    __builtin_amdgcn_sched_barrier(0);
    if (nk > 2) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");     // D(0) landed: 4 + 8 + 4 + 8 operations are newer
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (grpB) __builtin_amdgcn_s_barrier();              // waves 4-7 run one barrier behind waves 0-3 from here on
    __builtin_amdgcn_sched_barrier(0);
    using std::integral_constant;
    int kt = 0;
    for (; kt + 5 <= nk; kt += 2) {                      // kt + 1 + 3 < nk: DMA three tiles ahead still to be issued in both
        tile(kt, integral_constant<int, 0>{}, integral_constant<bool, false>{});
        tile(kt + 1, integral_constant<int, 1>{}, integral_constant<bool, false>{});
    }
    for (; kt < nk; kt += 2) {
        tile(kt, integral_constant<int, 0>{}, integral_constant<bool, true>{});
        if (kt + 1 < nk) tile(kt + 1, integral_constant<int, 1>{}, integral_constant<bool, true>{});
    }
    __builtin_amdgcn_sched_barrier(0);
    if (!grpB) __builtin_amdgcn_s_barrier();             // pair the extra barrier of waves 4-7
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // outstanding (unused) fragment prefetches
    __builtin_amdgcn_s_barrier();                        // every wave is done reading the ring before it is reused

    // ---- epilogue (gemm_common.h) -------------------------------------------------------------------------------------------------
    const int m_base = m0 + wr * 128, n_base = n0 + wc * 64;
    if constexpr (MODE == MODE_F32) {
        constexpr int SB = 32 * NT * 128;
        epilogue_staged<T, MODE_F32, MT, NT>(g, acc, smem + wave * SB, m_base, n_base, lane);
    } else {
        constexpr int SB = MT * 32 * (MODE == MODE_SWIGLU ? 1 : NT) * 64;
        static_assert(NWAVES * SB <= WF_ASTAGES * A_BYTES, "epilogue staging must fit in the ring");
        epilogue_staged<T, MODE, MT, NT>(g, acc, smem + wave * SB, m_base, n_base, lane);
    }
}

// Weight re-layout: out block b = ntile * (K/16) + kstep holds, for lane l, W[ntile*32 + (l & 31)][kstep*16 + 8*(l >> 5) .. +7]
// (rows past N are zero).  One thread per 16-byte chunk.
__global__ __launch_bounds__(256) void pack_wf_kernel(const char* __restrict__ W, int ldw, int N, int nk16, int ntiles, char* __restrict__ out) {
    const long c = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)ntiles * nk16 * 64;
    if (c >= total) return;
    const int l = (int)(c & 63);
    const long b = c >> 6;
    const int ks = (int)(b % nk16), nt = (int)(b / nk16);
    const int row = nt * 32 + (l & 31);
    u32x4 v = {0u, 0u, 0u, 0u};
    if (row < N) v = *(const u32x4*)(W + ((size_t)row * ldw + ks * 16 + (l >> 5) * 8) * 2);
    *(u32x4*)(out + c * 16) = v;
}

template <typename T, int MODE, int ABL = 0, int WF_BM = 256>
static int launch_wf_tm(GemmArgs g, hipStream_t s) {
    constexpr int LDS = WF_ASTAGES * WF_BM * WF_BK * 2;
    static int done_dev = -1;
    int cur_dev = 0;
    (void)hipGetDevice(&cur_dev);
    if (cur_dev != done_dev) {
        (void)hipFuncSetAttribute((const void*)gemm_wf_kernel<T, MODE, ABL, WF_BM>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        done_dev = cur_dev;
    }
    g.tiles_m = ceil_div(g.M, WF_BM);
    if (WF_BM == 128) {
        static int cus = 0;
        if (!cus) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, cur_dev);
        g.stagger_blocks = 2 * cus;                      // the initial fill: two blocks per CU
        // half of what a 128 x 256 tile takes while it shares its CU (~1.1 us per K-tile) unless gemm_stagger gives ns explicitly
        const int st = RESEARCH(gemm_stagger);
        g.stagger_ns = st > 1 ? st : st == 1 ? (g.K / 64) * 550 + 3000 : 0;
    }
    hipLaunchKernelGGL((gemm_wf_kernel<T, MODE, ABL, WF_BM>), dim3(g.tiles_m * g.tiles_n), dim3(2 * WF_BM), LDS, s, g);
    return (int)hipGetLastError();
}

bool gemm_wf_applicable(const GemmArgs& g, int mode) {
    if (!g.Wf || g.K % WF_BK || g.M < 256 || g.N < 256) return false;
    if (g.lda % 8 || g.ldo % 8 || g.N % 8) return false;
    if (mode == MODE_SWIGLU && g.N % 64) return false;
    return true;
}

int launch_gemm_wf(const GemmArgs& g_in, int dtype, int mode, hipStream_t s) {
    GemmArgs g = g_in;
    g.tiles_m = ceil_div(g.M, 256);
    g.tiles_n = ceil_div(g.N, WF_BN);
    g.group_m = RESEARCH(gemm_group) > 0 ? RESEARCH(gemm_group) : 4;
    const int abl = (g.flags >> 16) & 3;                 // profiling only (gemm.hip masks these bits in product runs)
    if (abl && mode == MODE_STORE16 && dtype == DT_BF16) {
        switch (abl) {
            case 1: return launch_wf_tm<bf16_t, MODE_STORE16, 1>(g, s);
            case 2: return launch_wf_tm<bf16_t, MODE_STORE16, 2>(g, s);
            case 3: return launch_wf_tm<bf16_t, MODE_STORE16, 3>(g, s);
            default: break;
        }
    }
    if (RESEARCH(gemm_flags) & 0x2000000) {               // bit25: two independent 4-wave blocks per CU (128 x 256 tiles)
#define WF_GO2(T)                                                                                    \
    (mode == MODE_STORE16 ? launch_wf_tm<T, MODE_STORE16, 0, 128>(g, s)                              \
                          : mode == MODE_F32 ? launch_wf_tm<T, MODE_F32, 0, 128>(g, s) : launch_wf_tm<T, MODE_SWIGLU, 0, 128>(g, s))
        return dtype == DT_BF16 ? WF_GO2(bf16_t) : WF_GO2(f16_t);
#undef WF_GO2
    }
#define WF_GO(T)                                                                                     \
    (mode == MODE_STORE16 ? launch_wf_tm<T, MODE_STORE16>(g, s)                                      \
                          : mode == MODE_F32 ? launch_wf_tm<T, MODE_F32>(g, s) : launch_wf_tm<T, MODE_SWIGLU>(g, s))
    return dtype == DT_BF16 ? WF_GO(bf16_t) : WF_GO(f16_t);
#undef WF_GO
}

}  // namespace vitg

using namespace vitg;

extern "C" int vitg_weight_fragments_bytes(int N, int K, long* bytes) {
    if (!bytes || N <= 0 || K <= 0 || K % 16) return VITG_ERR_ARG;
    *bytes = (long)ceil_div(N, 64) * 2 * (K / 16) * 1024;          // weight tiles in pairs of 32 rows (a wave owns 64 rows)
    return 0;
}

extern "C" int vitg_pack_weight_fragments(const void* W, int ldw, int N, int K, void* out, void* stream) {
    if (!W || !out || N <= 0 || K <= 0 || K % 16 || ldw % 8 || ldw < K) return VITG_ERR_ARG;
    const int ntiles = ceil_div(N, 64) * 2, nk16 = K / 16;
    const long total = (long)ntiles * nk16 * 64;
    hipLaunchKernelGGL(pack_wf_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const char*)W, ldw, N, nk16,
                       ntiles, (char*)out);
    return hipGetLastError() == hipSuccess ? 0 : VITG_ERR_LAUNCH;
}

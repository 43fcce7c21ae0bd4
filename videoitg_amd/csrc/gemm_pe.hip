// Persistent 256x256x64 GEMM with a DEFERRED epilogue (16-bit output; bias and activation at the tile boundary), gfx950.
//
// Why.  profiles/r01_gemm_shapes.md sections 13-14: on the K = 1152 shapes of the SigLIP tower a 256x256 tile spends ~12 us outside
// its main loop (block hand-over, first DMA round trip, accumulator -> LDS transpose -> stores) against 15-22 us inside, and with
// one block per CU nothing overlaps that time.  Here one block per CU walks a list of tiles; the operand ring simply continues
// across tile boundaries (the last K-tile of tile i prefetches K-tile 0 of tile i+1), and the finished tile is NOT stored at the
// tile boundary: its accumulators are biased, rounded and packed to 16 bits (48 registers per lane + one 4-KiB staging window per
// wave) and trickled out - LDS transpose, full-row 16-B stores - from the read intervals of the next tile's first twelve K-tiles.
// At the boundary the matrix pipe only waits for the packing (~0.5 us per wave group).
//
// Main loop = the phased schedule of gemm256.hip (cfg 6): every k-step is a fragment-read interval and an MFMA interval, waves 4-7
// one barrier behind waves 0-3, DMA pieces of the next K-tile spread over the first three read intervals, landing waits at the end
// of the K-tile.  The first twelve K-tiles of every tile are straight-line code so that the trickle costs no branches and the packed
// registers die one row-tile at a time.  LDS: 2 x 64 KiB ring + 8 x 4 KiB staging = 160 KiB.
// Same tile, wave tile, k-order, MFMA and rounding as cfg 6: bit-identical results.
#include "gemm_common.h"
#include <stdlib.h>
#include <type_traits>

namespace vitg {

constexpr int PE_THREADS = 512, PE_BM = 256, PE_BN = 256, PE_BK = 64, PE_UNROLL_KT = 12;
constexpr int PE_ACT_SWIGLU = 100;                    // ACT value: SwiGLU mode (silu(gate) * up on the [32 gate | 32 up] row interleave, no bias)

template <typename T, int ACT>
__global__ __launch_bounds__(PE_THREADS, 2) void gemm_pe_kernel(GemmArgs g) {
    typedef typename T16<T>::vec8 vec8;
    typedef typename T16<T>::vec4 vec4;
    constexpr int MT = 4, NT = 2, KS = 4, ROWB = 128, CPR = 8, RPI = 8;
    constexpr int A_BYTES = PE_BM * ROWB, STAGE = 2 * A_BYTES, PIECES = 4;        // 4 A + 4 W DMA pieces per wave per K-tile
    constexpr bool SW = ACT == PE_ACT_SWIGLU;
    constexpr int NTO = SW ? 1 : NT, CH = NTO * 4, RB = NTO * 64;                 // output 32-column tiles per wave, 16-B chunks / bytes per staged row
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int frow = lane & 31, fh = lane >> 5;
    const int drow = lane / CPR, dslot = lane % CPR;
    char* const stg = smem + 2 * STAGE + wave * 4096;    // wave-private staging window: 32 rows x 128 B

    const int nwg = g.tiles_m * g.tiles_n;
    const int n_my = (nwg - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;      // grid <= nwg: at least one

    // ---- operand stream: wave-uniform tile base + per-lane 32-bit offsets; ONE live set, switched to the next tile when the ring wraps
    const char* a_base;
    const char* w_base;
    unsigned a_cur[PIECES], w_cur[PIECES];
    auto set_src = [&](int b, int& m0, int& n0) {
        int tm, tn;
        block_to_tile(b, g.tiles_m, g.tiles_n, g.group_m, tm, tn);
        m0 = tm * PE_BM;
        n0 = tn * PE_BN;
        a_base = (const char*)g.A + (size_t)m0 * g.lda * 2;
        w_base = (const char*)g.W + (size_t)n0 * g.ldw * 2;
#pragma unroll
        for (int j = 0; j < PIECES; ++j) {
            const int R = (wave * PIECES + j) * RPI + drow;
            const unsigned sw = (unsigned)((dslot ^ ((R >> 1) & 7)) << 3);
            a_cur[j] = ((unsigned)(min(R, g.M - 1 - m0) * g.lda) + sw) * 2u;
            w_cur[j] = ((unsigned)(min(R, g.N - 1 - n0) * g.ldw) + sw) * 2u;
        }
    };
    auto issue_piece = [&](unsigned so, int kt, int j) {                  // j-th of 8 pieces of K-tile kt into the stage at byte offset so
        if (j < PIECES)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_base + (size_t)kt * ROWB + a_cur[j]),
                                             (__attribute__((address_space(3))) void*)(smem + so + (wave * PIECES + j) * 1024), 16, 0, 0);
        else
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(w_base + (size_t)kt * ROWB + w_cur[j - PIECES]),
                                             (__attribute__((address_space(3))) void*)(smem + so + A_BYTES + (wave * PIECES + j - PIECES) * 1024), 16, 0, 0);
    };

    int a_off[KS], w_off[KS];
    {
        const int sw = (frow >> 1) & 7;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int ch = ((ks * 2 + fh) ^ sw) << 4;
            a_off[ks] = (wr * 128 + frow) * ROWB + ch;
            w_off[ks] = A_BYTES + (wc * 64 + frow) * ROWB + ch;
        }
    }

    f32x16 acc[NT][MT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    vec8 fa[MT], fw[NT];
    u32x2 pk[MT - 1][NTO][4];                            // row-tiles 1..3 of the previous tile, packed 16-bit (row-tile 0 goes straight to LDS)
#pragma unroll
    for (int i = 0; i < MT - 1; ++i)
#pragma unroll
        for (int j = 0; j < NTO; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) pk[i][j][q] = u32x2{0u, 0u};

    const int nk = g.K / PE_BK;
    const bool grpB = wr == 1;
    int cm0, cn0, nm0 = 0, nn0 = 0;                      // current / next tile origin
    int pm_base = 0, pn_base = 0, pM = 0;                // previous tile (being trickled): wave origin, row bound (0 = nothing to store)
    set_src((int)blockIdx.x, cm0, cn0);
#pragma unroll
    for (int j = 0; j < 2 * PIECES; ++j) issue_piece(0u, 0, j);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (grpB) __builtin_amdgcn_s_barrier();              // waves 4-7 run one barrier behind waves 0-3 from here on
    __builtin_amdgcn_sched_barrier(0);

    // ---- trickle steps (wave-private LDS window: LDS operations of one wave execute in order, no barrier needed) -----------------
    auto stage_write = [&](const u32x2 (&p)[NTO][4]) {   // one 32-row tile of packed outputs -> staging, 16-B chunks XOR-swizzled by row
#pragma unroll
        for (int nt = 0; nt < NTO; ++nt)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) *(u32x2*)(stg + frow * RB + (((nt * 4 + q4) ^ (frow & (CH - 1))) << 4) + fh * 8) = p[nt][q4];
    };
    const int n_lim = SW ? g.N / 2 : g.N;
    auto stage_store = [&](int mt) {                     // staging -> global, full rows of the wave's output (128 B, SwiGLU 64 B)
#pragma unroll
        for (int i = 0; i < 32 * CH / 64; ++i) {
            const int idx = i * 64 + lane, row = idx / CH, c = idx % CH;
            const u32x4 v = *(const u32x4*)(stg + row * RB + ((c ^ (row & (CH - 1))) << 4));
            const int m = pm_base + mt * 32 + row, n = pn_base + c * 8;
            if (m < pM && n < n_lim) *(u32x4*)((T*)g.out + (size_t)m * g.ldo + n) = v;
        }
    };

    unsigned s = 0;                                      // flat K-tile counter: ring stage = s & 1
    for (int it = 0; it < n_my; ++it) {
        const bool has_next = it + 1 < n_my;
        // one k-step = two intervals.  KT >= 0: compile-time K-tile index (straight-line part, with the trickle); KT < 0: runtime kt
        auto kstep = [&](auto kt_c, auto pp_c, int kt) {
            constexpr int KT = decltype(kt_c)::value, PP = decltype(pp_c)::value;
            const unsigned so = (s & 1u) * STAGE, sn = ((s + 1u) & 1u) * STAGE;
            const bool more = kt + 1 < nk || (has_next && !(g.flags & 0x800000));   // bit23 (A/B): no prefetch across the tile boundary
            // ---- read interval ----------------------------------------------------------------------------------------------
            if constexpr (KT >= 1 && KT % 3 == 1 && PP == 0) {          // K-tiles 1, 4, 7, 10: row-tile (KT-1)/3 leaves for global memory
                stage_store((KT - 1) / 3);
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (KT >= 3 && KT % 3 == 0 && PP == 2) {          // K-tiles 3, 6, 9: next packed row-tile into the window
                stage_write(pk[KT / 3 - 1]);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int i = 0; i < NT; ++i) fw[i] = *(const vec8*)(smem + so + w_off[PP] + i * (32 * ROWB));
#pragma unroll
            for (int i = 0; i < MT; ++i) fa[i] = *(const vec8*)(smem + so + a_off[PP] + i * (32 * ROWB));
            __builtin_amdgcn_sched_barrier(0);
            if (PP < 3 && more) {
                const int kn = kt + 1 < nk ? kt + 1 : 0;                // K-tile 0 of the next tile once the ring wraps
#pragma unroll
                for (int j = 0; j < 2 * PIECES; ++j)
                    if (j / 3 == PP) issue_piece(sn, kn, j);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (PP == KS - 1 && more && grpB) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            // ---- MFMA interval ----------------------------------------------------------------------------------------------
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[nt][mt] = T16<T>::mfma32(fw[nt], fa[mt], acc[nt][mt]);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            if (PP == KS - 1 && more && !grpB) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        };
        auto ktile = [&](auto kt_c, int kt) {
            if (kt == nk - 1 && has_next) set_src((int)blockIdx.x + (it + 1) * (int)gridDim.x, nm0, nn0);   // no DMA of this tile left
            kstep(kt_c, std::integral_constant<int, 0>{}, kt);
            kstep(kt_c, std::integral_constant<int, 1>{}, kt);
            kstep(kt_c, std::integral_constant<int, 2>{}, kt);
            kstep(kt_c, std::integral_constant<int, 3>{}, kt);
            ++s;
        };
#define PE_KT(N) ktile(std::integral_constant<int, N>{}, N);
        PE_KT(0) PE_KT(1) PE_KT(2) PE_KT(3) PE_KT(4) PE_KT(5) PE_KT(6) PE_KT(7) PE_KT(8) PE_KT(9) PE_KT(10) PE_KT(11)
#undef PE_KT
        // bias for this wave's 64 columns: 8 quads per lane, fetched now (six or more K-tiles ahead of their use; pk is dead here)
        f32x4 bq[NT][4];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const int n = cn0 + wc * 64 + nt * 32 + q4 * 8 + fh * 4;
                bq[nt][q4] = (!SW && g.bias && n < g.N) ? *(const f32x4*)(g.bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        for (int kt = PE_UNROLL_KT; kt < nk; ++kt) ktile(std::integral_constant<int, -1>{}, kt);

        // ---- tile boundary: bias, round, pack; row-tile 0 straight into the staging window; accumulators back to zero --------------
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            u32x2 p[NTO][4];
#pragma unroll
            for (int nt = 0; nt < NTO; ++nt)
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    vec4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float v;
                        if constexpr (SW) {
                            v = silu_f(acc[0][mt][q4 * 4 + e]) * acc[1][mt][q4 * 4 + e];
                            acc[1][mt][q4 * 4 + e] = 0.f;
                        } else {
                            v = acc[nt][mt][q4 * 4 + e] + bq[nt][q4][e];
                            if (ACT == VITG_ACT_GELU_TANH) v = gelu_tanh_f(v);
                            else if (ACT == VITG_ACT_GELU_ERF) v = gelu_erf_f(v);
                        }
                        o[e] = T16<T>::from_f32(v);
                        acc[nt][mt][q4 * 4 + e] = 0.f;
                    }
                    p[nt][q4] = bitcast<u32x2>(o);
                }
            if (mt == 0) stage_write(p);
            else {
#pragma unroll
                for (int nt = 0; nt < NTO; ++nt)
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) pk[mt - 1][nt][q4] = p[nt][q4];
            }
        }
        pm_base = cm0 + wr * 128;
        pn_base = SW ? (cn0 + wc * 64) / 2 : cn0 + wc * 64;
        pM = g.M;
        if (g.flags & 0x400000) {                        // A/B switch: persistent, but the tile is stored at the boundary (no trickle)
            stage_store(0);
#pragma unroll
            for (int mt = 1; mt < MT; ++mt) {
                stage_write(pk[mt - 1]);
                stage_store(mt);
            }
            pM = 0;
        }
        cm0 = nm0;
        cn0 = nn0;
        if ((g.flags & 0x800000) && has_next) {          // A/B: "software relaunch" - the next tile starts like a fresh block would
            if (!grpB) __builtin_amdgcn_s_barrier();     // re-align the two wave groups
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 2 * PIECES; ++j) issue_piece((s & 1u) * STAGE, 0, j);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (grpB) __builtin_amdgcn_s_barrier();
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    // ---- last tile of this block: flush synchronously ------------------------------------------------------------------------------
    stage_store(0);
#pragma unroll
    for (int mt = 1; mt < MT; ++mt) {
        stage_write(pk[mt - 1]);
        stage_store(mt);
    }
    if (!grpB) __builtin_amdgcn_s_barrier();             // pair the extra barrier of waves 4-7
}

bool gemm_pe_applicable(const GemmArgs& g, int mode) {
    if (mode != MODE_STORE16 && mode != MODE_SWIGLU) return false;
    if (mode == MODE_SWIGLU && g.N % 64) return false;
    if (g.K % PE_BK || g.K / PE_BK < PE_UNROLL_KT) return false;
    if (g.M < 256 || g.N < 256 || g.lda % 8 || g.ldw % 8 || g.ldo % 8 || g.N % 8) return false;
    return true;
}

template <typename T, int ACT>
static int launch_pe_t(const GemmArgs& g, hipStream_t s) {
    constexpr int LDS = 2 * 2 * PE_BM * PE_BK * 2 + 8 * 4096;        // 160 KiB
    static int done_dev = -1, cus = 256;
    int cur_dev = 0;
    (void)hipGetDevice(&cur_dev);
    if (cur_dev != done_dev) {
        (void)hipFuncSetAttribute((const void*)gemm_pe_kernel<T, ACT>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, cur_dev);
        done_dev = cur_dev;
    }
    const int grid = min(cus, g.tiles_m * g.tiles_n);
    hipLaunchKernelGGL((gemm_pe_kernel<T, ACT>), dim3(grid), dim3(PE_THREADS), LDS, s, g);
    return (int)hipGetLastError();
}

int launch_gemm_pe(const GemmArgs& g_in, int dtype, int mode, hipStream_t s) {
    GemmArgs g = g_in;
    g.tiles_m = ceil_div(g.M, PE_BM);
    g.tiles_n = ceil_div(g.N, PE_BN);
    g.group_m = RESEARCH(gemm_group) > 0 ? RESEARCH(gemm_group) : 4;
#define PE_GO(T)                                                                                         \
    (mode == MODE_SWIGLU ? launch_pe_t<T, PE_ACT_SWIGLU>(g, s) : g.act == VITG_ACT_GELU_TANH ? launch_pe_t<T, VITG_ACT_GELU_TANH>(g, s)                              \
                                 : g.act == VITG_ACT_GELU_ERF ? launch_pe_t<T, VITG_ACT_GELU_ERF>(g, s) : launch_pe_t<T, VITG_ACT_NONE>(g, s))
    return dtype == DT_BF16 ? PE_GO(bf16_t) : PE_GO(f16_t);
#undef PE_GO
}

}  // namespace vitg

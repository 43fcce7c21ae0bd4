// C[M,N] = A[M,K] * W[N,K]^T, 256 x 256 x 32 tile, FOUR waves (2 M x 2 N), each wave owns a 128 x 128 output tile held in
// 256 accumulator registers (16 MFMA 32x32 tiles -> AGPRs), one wave per SIMD (gfx950).
//
// Why.  With eight waves and 128x64 wave tiles (gemm256.hip) every 16-deep k-step reads 4 + 2 operand fragments from LDS for
// 8 MFMAs: 0.75 KiB of ds_read per MFMA, plus 0.25 KiB/MFMA of LDS-DMA writes = 1 KiB of LDS traffic per MFMA.  A 32x32x16
// MFMA occupies its SIMD for 32 clocks, four SIMDs -> 128 B/clk per CU: exactly the LDS port's peak (measured:
// tools/ubench/mfma_valu_overlap.hip, one ds_read_b128 per MFMA slice already costs 13 ns against a 16 ns MFMA).  Both pipes
// would have to be 100 % busy at once; in practice the loop settles at ~59 % of the MFMA ceiling.  A 128x128 wave tile reads
// 4 + 4 fragments for 16 MFMAs: 0.5 KiB/MFMA (+0.25 DMA) = 75 % of the port at full MFMA rate, which leaves the matrix pipe
// as the only saturated resource.  The price is occupancy: 256 accumulators + 64 fragment registers -> one wave per SIMD, so
// all latency hiding is explicit (fragment reads one k-step ahead, DMA three K-tiles ahead, the tile barrier placed in front
// of the last k-step's MFMAs so the next tile's first fragments are in flight while the matrix pipe drains).
//
// Ring of 4 stages x (256 + 256 rows x 64 B) = 128 KiB LDS; per K-tile and wave: 8 LDS-DMA instructions, 16 ds_read_b128,
// 32 MFMAs, ONE s_barrier.  LDS image and swizzle as gemm256.hip (64-B rows, chunk ^ ((row >> 3) & 3)).
#include "gemm_common.h"

namespace vitg {

constexpr int W4_THREADS = 256, W4_BM = 256, W4_BN = 256, W4_BK = 32;

template <typename T, int MODE, int W4_STAGES>
__global__ __launch_bounds__(W4_THREADS, 1) void gemm_w4_kernel(GemmArgs g) {
    typedef typename T16<T>::vec8 vec8;
    constexpr int MT = 4, NT = 4, KS = 2, ROWB = 64, CPR = 4, RPI = 16;
    constexpr int A_BYTES = W4_BM * ROWB, W_BYTES = W4_BN * ROWB, STAGE = A_BYTES + W_BYTES;      // 16 + 16 KiB
    constexpr int A_INSTR = A_BYTES / 1024 / 4, W_INSTR = W_BYTES / 1024 / 4, LPT = A_INSTR + W_INSTR;   // 4 + 4 per wave per K-tile
    constexpr int DIST = W4_STAGES - 1;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;

    int tm, tn;
    block_to_tile(blockIdx.x, g.tiles_m, g.tiles_n, 8, tm, tn);
    const int m0 = tm * W4_BM, n0 = tn * W4_BN;

    // ---- DMA sources: wave-uniform 64-bit tile base (SGPRs) + per-lane 32-bit byte offset (one VGPR per instruction) --------
    const int drow = lane / CPR, dslot = lane % CPR;
    const char* a_base = (const char*)g.A + (size_t)m0 * g.lda * 2;
    const char* w_base = (const char*)g.W + (size_t)n0 * g.ldw * 2;
    unsigned a_cur[A_INSTR], w_cur[W_INSTR];
#pragma unroll
    for (int j = 0; j < A_INSTR; ++j) {
        const int R = (wave * A_INSTR + j) * RPI + drow;                 // row inside the 256-row tile
        a_cur[j] = (unsigned)(min(R, g.M - 1 - m0) * g.lda + ((dslot ^ ((R >> 3) & 3)) << 3)) * 2u;
    }
#pragma unroll
    for (int j = 0; j < W_INSTR; ++j) {
        const int R = (wave * W_INSTR + j) * RPI + drow;
        w_cur[j] = (unsigned)(min(R, g.N - 1 - n0) * g.ldw + ((dslot ^ ((R >> 3) & 3)) << 3)) * 2u;
    }
    auto issue = [&](int kt) {
        char* st = smem + (kt % W4_STAGES) * STAGE;
        const char* ab = a_base + (size_t)kt * ROWB;
        const char* wb = w_base + (size_t)kt * ROWB;
#pragma unroll
        for (int j = 0; j < A_INSTR; ++j)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ab + a_cur[j]),
                                             (__attribute__((address_space(3))) void*)(st + (wave * A_INSTR + j) * 1024), 16, 0, 0);
#pragma unroll
        for (int j = 0; j < W_INSTR; ++j)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wb + w_cur[j]),
                                             (__attribute__((address_space(3))) void*)(st + A_BYTES + (wave * W_INSTR + j) * 1024), 16, 0, 0);
    };

    const int frow = lane & 31, fh = lane >> 5;
    f32x16 acc[NT][MT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment byte offsets inside a stage: row*64 + ((chunk ^ ((row>>3)&3)) << 4); the swizzle term is the same for row + 32*i
    unsigned a_off[KS], w_off[KS];
    {
        const int sw = (frow >> 3) & 3;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int ch = ((ks * 2 + fh) ^ sw) << 4;
            a_off[ks] = (unsigned)(size_t)smem + (wr * 128 + frow) * ROWB + ch;
            w_off[ks] = (unsigned)(size_t)smem + A_BYTES + (wc * 128 + frow) * ROWB + ch;
        }
    }
    // One wave per SIMD: nothing but this wave's own instruction stream keeps the matrix pipe fed, so every LDS read, DMA
    // issue, wait and barrier is placed BETWEEN MFMAs (an MFMA occupies the pipe for 32 clocks but issues in a few).  All of it
    // is inline asm / builtins pinned by sched_barrier: hipcc's waitcnt pass never sees the reads, completion is enforced by
    // the hand-counted lgkmcnt waits (LDS returns in order).
    // Fragment read order of a k-step (8 x ds_read_b128): fw[0], fa[0], fa[1], fa[2], fa[3], fw[1], fw[2], fw[3]; MFMA order
    // nt-major, so MFMAs 0-3 need reads 0-4, MFMAs 4-7 read 5, 8-11 read 6, 12-15 read 7.
    vec8 fa0[MT], fw0[NT], fa1[MT], fw1[NT];
#define W4_RD1(FA, FW, OFFA, OFFW, SO, J)                                                                                  \
    {                                                                                                                      \
        if ((J) == 0 || (J) >= 5)                                                                                          \
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(FW[(J) == 0 ? 0 : (J) - 4]) : "v"(OFFW + SO), "n"(((J) == 0 ? 0 : (J) - 4) * 32 * ROWB)); \
        else                                                                                                               \
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(FA[(J) - 1]) : "v"(OFFA + SO), "n"(((J) - 1) * 32 * ROWB));   \
    }
#define W4_LGKM(N) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory")
#define W4_MFMA(FA, FW, I) acc[(I) / MT][(I) % MT] = T16<T>::mfma32(FW[(I) / MT], FA[(I) % MT], acc[(I) / MT][(I) % MT])
#define W4_PIN() __builtin_amdgcn_sched_barrier(0)

    auto issue_one = [&](int kt, int j) {                // j-th DMA instruction of K-tile kt (j compile-time after unroll)
        char* st = smem + (kt % W4_STAGES) * STAGE;
        if (j < A_INSTR)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_base + (size_t)kt * ROWB + a_cur[j]),
                                             (__attribute__((address_space(3))) void*)(st + (wave * A_INSTR + j) * 1024), 16, 0, 0);
        else
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(w_base + (size_t)kt * ROWB + w_cur[j - A_INSTR]),
                                             (__attribute__((address_space(3))) void*)(st + A_BYTES + (wave * W_INSTR + j - A_INSTR) * 1024), 16, 0, 0);
    };

    const int nk = g.K / W4_BK;
#pragma unroll
    for (int t = 0; t < DIST; ++t)
        if (t < nk) issue(t);
    // tile 0 landed (this wave: all but the newer tiles' loads; then every wave)
    auto wait_newer = [&](int newer) {                   // all of this wave's DMA except the `newer` most recent tiles
        if (newer >= 3 && DIST > 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * LPT) : "memory");
        else if (newer == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPT) : "memory");
        else if (newer == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPT) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    wait_newer(min(nk, DIST) - 1);
    __builtin_amdgcn_s_barrier();
    W4_PIN();
#pragma unroll
    for (int j = 0; j < 8; ++j) W4_RD1(fa0, fw0, a_off[0], w_off[0], 0u, j)
    W4_PIN();

    for (int kt = 0; kt < nk; ++kt) {
        const unsigned so = (unsigned)((kt % W4_STAGES) * STAGE);
        const unsigned so_next = (unsigned)(((kt + 1) % W4_STAGES) * STAGE);
        const bool dma = kt + DIST < nk && !(g.flags & 64);                 // stage (kt+3)&3 held tile kt-1: free since the barrier of iteration kt-1
        const bool more = kt + 1 < nk;
        // ---- k-step 0: 16 MFMAs on fa0/fw0; after MFMA 0-7 one k-step-1 fragment read each, after MFMA 8-15 one DMA each ---
        W4_LGKM(3);                                      // reads 0-4 of this k-step (issued during the previous k-step 1)
        W4_PIN();
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (i == 4) W4_LGKM(6);                      // fw0[1]: may still fly: fw0[2], fw0[3] + 4 new reads
            if (i == 8) W4_LGKM(9);                      // fw0[2]: fw0[3] + 8 new reads
            if (i == 12) W4_LGKM(8);                     // fw0[3]: 8 new reads
            W4_PIN();
            W4_MFMA(fa0, fw0, i);
            W4_PIN();
            if (i < 8) W4_RD1(fa1, fw1, a_off[1], w_off[1], so, i)
            else if (dma) issue_one(kt + DIST, i - 8);
            W4_PIN();
        }
        // ---- k-step 1: 16 MFMAs on fa1/fw1; the tile barrier after MFMA 1, then the first fragments of tile kt+1 ------------
        W4_LGKM(3);
        W4_PIN();
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            W4_MFMA(fa1, fw1, i);
            W4_PIN();
            if (i == 1) {
                W4_LGKM(0);                              // all my reads of tile kt are done (fw1[1..3] were issued 10+ MFMAs ago)
                if (more) {
                    const int newer = min(nk - 2 - kt, DIST - 1);                // tiles issued after kt+1 that may still fly
                    if (!(g.flags & 4)) wait_newer(newer);                    // (flags bits 2, 3, 6: profiling only, wrong results)
                    if (!(g.flags & 8)) __builtin_amdgcn_s_barrier();        // tile kt+1 landed everywhere; everyone is done reading tile kt
                }
                W4_PIN();
            }
            if (i >= 2 && i < 10 && more) W4_RD1(fa0, fw0, a_off[0], w_off[0], so_next, i - 2)
            W4_PIN();
        }
    }
#undef W4_RD1
#undef W4_LGKM
#undef W4_MFMA
#undef W4_PIN

    // ---- epilogue (gemm_common.h): transposed through wave-private LDS into full-row stores --------------------------------
    __builtin_amdgcn_s_barrier();                        // every wave is done reading the ring before it is reused
    const int m_base = m0 + wr * 128, n_base = n0 + wc * 128;
    if constexpr (MODE == MODE_F32) {
        // (no residual pre-load into the accumulators here: 64 row loads per lane on top of 256 live accumulators spill; the
        // residual is added on the coalesced side of the staged epilogue instead)
        constexpr int SB = 32 * NT * 128;                // 16 KiB per wave
        epilogue_staged<T, MODE_F32, MT, NT>(g, acc, smem + wave * SB, m_base, n_base, lane);
    } else if constexpr (MODE == MODE_SWIGLU) {
        // weight rows are interleaved [32 gate | 32 up]: the wave's four 32-wide tiles are two (gate, up) pairs
        constexpr int SB = MT * 32 * 64;                 // 8 KiB per wave and pair
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
            f32x16 pair[2][MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) { pair[0][mt] = acc[2 * pr][mt]; pair[1][mt] = acc[2 * pr + 1][mt]; }
            epilogue_staged<T, MODE_SWIGLU, MT, 2>(g, pair, smem + (wave * 2 + pr) * SB, m_base, n_base + 64 * pr, lane);
        }
    } else {
        constexpr int SB = MT * 32 * NT * 64;            // 32 KiB per wave
        static_assert(4 * SB <= W4_STAGES * STAGE, "epilogue staging must fit in the ring");
        epilogue_staged<T, MODE, MT, NT>(g, acc, smem + wave * SB, m_base, n_base, lane);
    }
}

template <typename T, int MODE, int STAGES>
static int launch_w4_tms(const GemmArgs& g, hipStream_t s) {
    constexpr int LDS = STAGES * (W4_BM + W4_BN) * W4_BK * 2;
    static int done_dev = -1;            // per device: the attribute belongs to the device's copy of the kernel
    int cur_dev = 0;
    (void)hipGetDevice(&cur_dev);
    if (cur_dev != done_dev) {
        (void)hipFuncSetAttribute((const void*)gemm_w4_kernel<T, MODE, STAGES>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        done_dev = cur_dev;
    }
    hipLaunchKernelGGL((gemm_w4_kernel<T, MODE, STAGES>), dim3(g.tiles_m * g.tiles_n), dim3(W4_THREADS), LDS, s, g);
    return (int)hipGetLastError();
}
template <typename T, int MODE>
static int launch_w4_tm(const GemmArgs& g, hipStream_t s) {
    return (RESEARCH(gemm_flags) & 128) ? launch_w4_tms<T, MODE, 5>(g, s) : launch_w4_tms<T, MODE, 4>(g, s);     // bit7: 5-stage ring (160 KiB)
}

bool gemm_w4_applicable(const GemmArgs& g, int mode) {
    if (g.K % W4_BK || g.K < 3 * W4_BK) return false;
    if (g.lda % 8 || g.ldw % 8 || g.ldo % 8) return false;
    if (mode == MODE_SWIGLU && g.N % 64) return false;
    return true;
}

int launch_gemm_w4(const GemmArgs& g_in, int dtype, int mode, hipStream_t s) {
    GemmArgs g = g_in;
    g.tiles_m = ceil_div(g.M, W4_BM);
    g.tiles_n = ceil_div(g.N, W4_BN);
#define W4_GO(T)                                                                                     \
    (mode == MODE_STORE16 ? launch_w4_tm<T, MODE_STORE16>(g, s)                                      \
                          : mode == MODE_F32 ? launch_w4_tm<T, MODE_F32>(g, s) : launch_w4_tm<T, MODE_SWIGLU>(g, s))
    return dtype == DT_BF16 ? W4_GO(bf16_t) : W4_GO(f16_t);
#undef W4_GO
}

}  // namespace vitg

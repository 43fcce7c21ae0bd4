// C[M,N] = A[M,K] * W[N,K]^T, 256 x BN x 32 tile (BN = 256 or 128), 8 waves (2 M x 4 N), ring-buffered LDS fed by
// LDS-DMA (global_load_lds_dwordx4) with loads in flight ACROSS the per-K-tile barrier (gfx950).
//
//   ring of 4 stages, prefetch distance 3 K-tiles:
//     iteration t:  s_waitcnt vmcnt(<loads of the 2 newer tiles>)   -> this wave's DMA for tile t has landed
//                   s_barrier                                       -> every wave's has; all waves left compute(t-1)
//                   issue DMA of tile t+3 into stage (t+3)&3 == (t-1)&3   (free: nobody reads tile t-1 any more)
//                   compute(t) from stage t&3: ds_read_b128 fragments + v_mfma_f32_32x32x16
//   One raw s_barrier per K-tile, never a vmcnt(0) in the steady state, no VGPR round trip and no ds_write for
//   staging.  (__syncthreads() would drain the DMA queue: cdna_hip_programming.md §5 "glds vs register staging".)
//
// LDS image of one operand tile: [rows][4 chunks x 16 B] (64-B rows), chunk index XOR ((row>>3)&3): conflict-free for
// the ds_read_b128 lane groups with the 32x32x16 fragment pattern (row = lane&31, chunk = 2*ks + lane>>5).  The DMA
// writes LDS lane-linearly, so the swizzle is applied to the per-lane SOURCE address (rule 21 of the guide).
//
// MFMA "A" operand = weight rows, "B" operand = activation rows (as in gemm.hip): a lane's accumulator holds 4
// consecutive output features of one token.
#include "gemm_common.h"
#include <stdlib.h>
#include <type_traits>

namespace vitg {

constexpr int G2_BM = 256, G2_THREADS = 512;

// Two shipped configurations (chosen per shape by launch256_t):
//   <BN=256, WM=2, STAGES=4>: wave tile 128x64, 128 KiB LDS, 1 block/CU  - best steady state (large K)
//   <BN=128, WM=4, STAGES=3>: wave tile  64x64,  72 KiB LDS, 2 blocks/CU - one block's prologue/epilogue overlaps the
//                                                                          other's main loop (small K, heavy epilogues)
#ifndef VITG_PROFILING_BUILD
#define VITG_PROFILING_BUILD 0     // build.sh -DVITG_PROFILING_BUILD=1 compiles the ablation switches (gemm_flags bits 1-4, 6, 9, 15, 18) into the
#endif                             // main loop; in the product build they are compiled out: every test of g.flags in an interval is a scalar
                                   // branch, and eight of them per read interval cost the phased loop 5-10 % (profiles/r01_gemm_shapes.md section 16)
template <typename T, int MODE, int BN, int WM, int STAGES, int BK, bool ASMR = false, int PH = 0, int BM = G2_BM, bool NRM = false>
__global__ __launch_bounds__(G2_THREADS, (BN >= 256 ? 2 : 4)) void gemm256_kernel(GemmArgs g_) {
    typedef typename T16<T>::vec8 vec8;
    GemmArgs g = g_;
    if constexpr (MODE == MODE_F32) {
        // split-K (small-M shapes of the context-parallel LLM: 98 tiles on 256 CUs): block y of a tile takes K-tiles [k0, k0 + cnt)
        // and stores its raw fp32 partial into its own buffer; everything below just sees a shorter K and another output pointer
        if (g.split_k > 1) {
            const int sp = blockIdx.y;
            const int nk_all = g.K / BK, per = nk_all / g.split_k, rem = nk_all % g.split_k;
            const int k0 = sp * per + min(sp, rem);
            g.A = (const char*)g.A + (size_t)k0 * BK * 2;
            g.W = (const char*)g.W + (size_t)k0 * BK * 2;
            g.K = (per + (sp < rem ? 1 : 0)) * BK;
            g.out = (float*)g.out + (size_t)sp * g.split_stride;
        }
    }
    constexpr bool PROF = VITG_PROFILING_BUILD != 0;
    constexpr int WN = 8 / WM;
    constexpr int MT = BM / WM / 32;                  // 32-row activation tiles per wave
    constexpr int NT = BN / WN / 32;                     // 32-wide weight tiles per wave
    constexpr int KS = BK / 16;                          // 16-deep MFMA k-steps per K-tile
    constexpr int ROWB = BK * 2;                         // bytes per LDS row (64 or 128)
    constexpr int CPR = BK / 8;                          // 16-B chunks per row
    constexpr int RPI = 1024 / ROWB;                     // rows covered by one DMA instruction (1 KiB)
    constexpr int A_BYTES = BM * ROWB;
    constexpr int W_BYTES = BN * ROWB;
    constexpr int STAGE = A_BYTES + W_BYTES;
    constexpr int A_INSTR = A_BYTES / 1024 / 8;          // DMA instructions per wave per K-tile
    constexpr int W_INSTR = W_BYTES / 1024 / 8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // Tile loop.  Launched with one block per tile this runs once; launched with one block per CU (launch256: "software relaunch",
    // default for long tile lists; gemm_flags bit24 turns it off) the block walks every gridDim.x-th tile of the same XCD-aware order and
    // simply starts over - sources, accumulators, first DMA - where the hardware would have launched a fresh block.  Measured on
    // the persistent kernel of gemm_pe.hip: this alone is worth +8-11 % on long-K shapes (profiles/r01_gemm_shapes.md section 15).
    const int nwg_ = g.tiles_m * g.tiles_n;
    if constexpr (MODE == MODE_F32) {
        // fp32-residual shapes in the tile loop: every tile is [residual tile -> accumulators: HBM read burst][main loop: operand
        // stream only][stores: HBM write burst] and all CUs walk these phases in lock-step (residual phase 28 us at 3.6 TB/s where a
        // lone CU needs 9 us, profiles/r02_gemm_timeline.md).  Starting the tile groups a third of a tile period apart spreads the
        // bursts; the blocks of one group (group_m x tiles_n tiles sharing operand panels in L2) keep a common phase.
        if (g.stagger_ns > 0) {
            int tm0, tn0;
            block_to_tile(blockIdx.x, g.tiles_m, g.tiles_n, g.group_m, tm0, tn0);
            const int ph = (tm0 / g.group_m) % 3;
            if (ph) {
                const unsigned long long t0 = wall_clock64();                                  // 100 MHz
                const unsigned long long ticks = (unsigned long long)g.stagger_ns * ph / 30;   // ns -> 10-ns ticks, thirds
                while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
            }
        }
    }
    for (int vb = blockIdx.x; vb < nwg_; vb += gridDim.x) {
    // per-lane quantities are re-derived from an opaque copy of the thread id in every pass: hoisted out of the loop they would all
    // stay live through the main loop (72-90 spilled registers on the wide tiles)
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WN, wc = wave % WN;
    int tm, tn;
    block_to_tile(vb, g.tiles_m, g.tiles_n, g.group_m, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;
    // Timeline (profiling build, vitg_gemm_timeline_buffer): waves 0 and 4 of the first stamp_blocks blocks record the 100 MHz
    // real-time counter at the tile's phase boundaries and after every K-tile, for their first four tiles.
    // slots: 0 tile start, 1 first K-tile landed, 2+kt end of K-tile kt (kt < 24), 28 epilogue begins, 29 epilogue stores issued,
    // 30 tile done (LDS drained, barrier), 31 shader-clock counter at tile start
    const int tile_i_ = (vb - (int)blockIdx.x) / (int)gridDim.x;
    auto stamp = [&](int slot) {
        if constexpr (PROF) {
            if (g.stamps && (int)blockIdx.x < g.stamp_blocks && tile_i_ < 4 && (wave & 3) == 0 && lane == 0)
                g.stamps[(((size_t)blockIdx.x * 2 + (wave >> 2)) * 4 + tile_i_) * 32 + slot] =
                    slot == 31 ? __builtin_amdgcn_s_memtime() : __builtin_amdgcn_s_memrealtime();
        }
    };
    stamp(0);
    stamp(31);

    // ---- DMA assignment: one instruction = RPI rows x ROWB bytes; lane -> (row = lane / CPR, LDS chunk slot = lane % CPR)
    // chunk swizzle: BK=32: (row>>3)&3, BK=64: (row>>1)&7 -- both invariant under row += 32 and conflict-free for the
    // ds_read_b128 lane groups with the 32x32x16 fragment pattern.
    auto swz = [](int row) { return BK == 32 ? ((row >> 3) & 3) : ((row >> 1) & 7); };
    const int drow = lane / CPR, dslot = lane % CPR;
    // source = wave-uniform tile base (SGPR pair) + per-lane 32-bit byte offset: the DMA takes the saddr + voffset form and the
    // K-tile advance is one scalar add (per-lane 64-bit pointers cost two v_lshl_add_u64 per piece in every read interval)
    const char* const a_base = (const char*)g.A + (size_t)m0 * g.lda * 2;
    const char* const w_base = (const char*)g.W + (size_t)n0 * g.ldw * 2;
    unsigned a_cur[A_INSTR], w_cur[W_INSTR];
#pragma unroll
    for (int j = 0; j < A_INSTR; ++j) {
        const int R = (wave * A_INSTR + j) * RPI + drow;                // row within the activation tile
        a_cur[j] = ((unsigned)(min(R, g.M - 1 - m0) * g.lda) + (unsigned)((dslot ^ swz(R)) << 3)) * 2u;
    }
#pragma unroll
    for (int j = 0; j < W_INSTR; ++j) {
        const int R = (wave * W_INSTR + j) * RPI + drow;
        w_cur[j] = ((unsigned)(min(R, g.N - 1 - n0) * g.ldw) + (unsigned)((dslot ^ swz(R)) << 3)) * 2u;
    }
    constexpr int LPT = A_INSTR + W_INSTR;               // DMA instructions per thread per K-tile
    auto issue_one = [&](int kt, int j) {                // j-th DMA instruction of K-tile kt (j compile-time after unroll)
        if (PROF && kt > 0 && (g.flags & 64)) return;            // profiling only (wrong results): operand stream off after the first K-tile
        char* st = smem + (kt % STAGES) * STAGE;
        const size_t koff = (size_t)kt * ROWB;
        // the lane offset goes through an opaque copy: otherwise the compiler hoists base + lane offset out of the K loop as a 64-bit
        // per-lane pointer and pays a v_lshl_add_u64 per piece instead of using the saddr form
        const char* ka = a_base + koff;
        const char* kw = w_base + koff;
        unsigned off = j < A_INSTR ? a_cur[j] : w_cur[j - A_INSTR];
        asm volatile("" : "+v"(off));
        if (j < A_INSTR)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ka + off),
                                             (__attribute__((address_space(3))) void*)(st + (wave * A_INSTR + j) * 1024), 16, 0, 0);
        else
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(kw + off),
                                             (__attribute__((address_space(3))) void*)(st + A_BYTES + (wave * W_INSTR + j - A_INSTR) * 1024), 16, 0, 0);
    };
    auto issue = [&](int kt) {
#pragma unroll
        for (int j = 0; j < LPT; ++j) issue_one(kt, j);
    };
    // Register-staged operand stream (PH == 3): the same 1-KiB pieces, but global_load_dwordx4 -> VGPR -> ds_write_b128 instead
    // of LDS-DMA.  Why: with the DMA the main loop loses 40 % (cfg 6 reaches 1858 TF/s on 8192^3 with the operand stream switched
    // off after the first K-tile, 1128 with it), and tools/ubench/lds_port_share.hip shows LDS-DMA writes starving behind
    // ds_read traffic; a ds_write_b128 moves 1 KiB in 8 LDS cycles.  Two register sets: tile kt+2 is in flight from global memory
    // (a whole tile of latency budget) while tile kt+1 is being written into the free LDS stage.
    u32x4 stg[PH == 3 ? 2 : 1][PH == 3 ? LPT : 1];
    auto load_one = [&](int kt, int j, u32x4 (&set)[PH == 3 ? LPT : 1]) {
        const size_t koff = (size_t)kt * ROWB;
        set[j] = *(const u32x4*)(j < A_INSTR ? a_base + koff + a_cur[j] : w_base + koff + w_cur[j - A_INSTR]);
    };
    auto store_one = [&](int kt, int j, const u32x4 (&set)[PH == 3 ? LPT : 1]) {
        char* st = smem + (kt % STAGES) * STAGE;
        char* dst = j < A_INSTR ? st + (wave * A_INSTR + j) * 1024 : st + A_BYTES + (wave * W_INSTR + j - A_INSTR) * 1024;
        *(u32x4*)(dst + lane * 16) = set[j];
    };

    const int frow = lane & 31, fh = lane >> 5;
    // Accumulators start at zero -- or, for the fp32 residual update without activation, AT THE RESIDUAL: the tile's
    // res[(m % res_mod), n] is loaded straight into the accumulator registers before the first DMA is issued, so the read
    // half of the read-modify-write overlaps the main loop instead of sitting (latency-bound, 256 KiB per tile through the
    // CU's vector-memory path) in the epilogue; the epilogue then only adds the bias and stores.  (These loads are older
    // than every DMA in the vmcnt queue, so the counted DMA waits still imply them.)
    const bool acc_from_res = MODE == MODE_F32 && g.res != nullptr && g.act == VITG_ACT_NONE;
    f32x16 acc[NT][MT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    if constexpr (MODE == MODE_F32) {
        if (acc_from_res) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int m = m0 + wr * (MT * 32) + mt * 32 + frow;
                const int rm = g.res_mod > 0 ? m % g.res_mod : m;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) {
                        const int n = n0 + wc * (NT * 32) + nt * 32 + q4 * 8 + fh * 4;
                        if (m < g.M && n < g.N) {
                            const f32x4 v = *(const f32x4*)(g.res + (size_t)rm * g.ldr + n);
#pragma unroll
                            for (int e = 0; e < 4; ++e) acc[nt][mt][q4 * 4 + e] = v[e];
                        }
                    }
            }
        }
    }

    // fragment byte offsets inside a tile: row*64 + ((chunk ^ ((row>>3)&3)) << 4); rows = base + 32*i + frow
    int a_off[KS], w_off[KS];
    {
        const int sw = swz(frow);                        // same for row and row + 32*i
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int ch = ((ks * 2 + fh) ^ sw) << 4;
            a_off[ks] = (wr * (MT * 32) + frow) * ROWB + ch;
            w_off[ks] = A_BYTES + (wc * (NT * 32) + frow) * ROWB + ch;
        }
    }

    const int nk = g.K / BK;
    constexpr int DIST = STAGES - 1;                     // prefetch distance in K-tiles
    if constexpr (PH == 3) {
        // tile 0 -> LDS through registers, tile 1 -> register set 1 (in flight)
#pragma unroll
        for (int j = 0; j < LPT; ++j) load_one(0, j, stg[0]);
        if (nk > 1) {
#pragma unroll
            for (int j = 0; j < LPT; ++j) load_one(1, j, stg[1]);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPT) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
#pragma unroll
        for (int j = 0; j < LPT; ++j) store_one(0, j, stg[0]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else {
    issue(0);
    if (DIST > 1 && nk > 1) issue(1);
    if (DIST > 2 && nk > 2) issue(2);
    }

    // Software pipeline at 16-deep k-step granularity with two named fragment sets: the MFMAs of one k-step run while
    // the ds_read_b128 of the next k-step are in flight (LDS latency off the MFMA critical path).
    vec8 fa0[MT], fw0[NT], fa1[MT], fw1[NT];
    auto wait_tile = [&](int kt) {                       // this wave's DMA of tile kt has landed
        if (PROF && (g.flags & 4)) return;                         // profiling only: results are wrong, shows the DMA-wait share
        const int newer = min(nk - 1 - kt, DIST - 1);    // tiles issued after kt that may still be in flight
        if (DIST > 2 && newer == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPT) : "memory");
        else if (DIST > 1 && newer == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPT) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
// Fragment reads.  ASMR: the ds_read_b128 are inline asm, invisible to hipcc's waitcnt pass, and completion is enforced
// by hand-counted `s_waitcnt lgkmcnt(N)` (G2_WAIT_OLDER): hipcc otherwise waits lgkmcnt(0) in front of the second MFMA
// block of the loop (conservative loop-carried merge), i.e. it waits for the reads just issued for the NEXT k-step and
// defeats the software pipeline.  LDS returns in order, so "all but the newest (NT+MT) reads" is exact.
#define G2_READ(FA, FW, KT, KS_)                                                                              \
    {                                                                                                         \
        if constexpr (ASMR) {                                                                                 \
            const unsigned wb_ = (unsigned)(size_t)(smem + ((KT) % STAGES) * STAGE) + (unsigned)w_off[KS_];    \
            const unsigned ab_ = (unsigned)(size_t)(smem + ((KT) % STAGES) * STAGE) + (unsigned)a_off[KS_];    \
            _Pragma("unroll") for (int i = 0; i < NT; ++i)                                                    \
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(FW[i]) : "v"(wb_), "n"(i * 32 * ROWB));   \
            _Pragma("unroll") for (int i = 0; i < MT; ++i)                                                    \
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(FA[i]) : "v"(ab_), "n"(i * 32 * ROWB));   \
        } else {                                                                                              \
            const char* st_ = smem + ((KT) % STAGES) * STAGE;                                                 \
            _Pragma("unroll") for (int i = 0; i < NT; ++i) FW[i] = *(const vec8*)(st_ + w_off[KS_] + i * (32 * ROWB)); \
            _Pragma("unroll") for (int i = 0; i < MT; ++i) FA[i] = *(const vec8*)(st_ + a_off[KS_] + i * (32 * ROWB)); \
        }                                                                                                     \
    }
// all LDS reads except the newest NT+MT (the set issued last) have returned
#define G2_WAIT_OLDER()                                                                                       \
    if constexpr (ASMR) {                                                                                     \
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(NT + MT) : "memory");                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
    }
#define G2_MMA(FA, FW)                                                                                        \
    __builtin_amdgcn_s_setprio(1);                                                                            \
    _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)                                                         \
        _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) acc[nt][mt] = T16<T>::mfma32(FW[nt], FA[mt], acc[nt][mt]); \
    __builtin_amdgcn_s_setprio(0);

    if constexpr (PH) {
        // ---- phased schedule (cfg 6): strict MFMA / load alternation between the two waves of every SIMD ----------------
        // Waves 0-3 (group A, rows 0-127) and waves 4-7 (group B, rows 128-255) share the SIMDs pairwise.  Every k-step is
        // two barrier-delimited intervals, [fragment reads + DMA issue] then [8 MFMAs]; group B runs ONE barrier behind
        // group A, so in every interval exactly one wave per SIMD issues MFMAs back-to-back (no contention for the matrix
        // pipe) while its sibling fills fragments / issues DMA.  Intervals (global barrier index i):
        //   A: R_j in interval 2j, M_j in 2j+1;   B: R_j in 2j+1, M_j in 2j+2       (j = 4*kt + p)
        // Stage (kt+1)&1 is last read by B's R_{4(kt-1)+3} in interval 8kt-1, so DMA for tile kt+1 may be issued from
        // interval 8kt on (A: p = 0,1,2 -> 8kt+0,+2,+4; B: 8kt+1,+3,+5); it must have landed before A's R_{4(kt+1)} in interval
        // 8kt+8: both groups drain vmcnt at the END of interval 8kt+7 (A after M_3, B after R_3).
        static_assert(PH == 0 || (STAGES == 2 && BK == 64), "phased schedule is written for the ring-2, 64-deep configurations");
        const bool grpB = wr == 1;
        if constexpr (PH != 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        stamp(1);
        if (grpB) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (PH == 3) {
        // Same interval structure as PH == 1; in its first three R phases of tile kt a wave writes its pieces of tile kt+1 from
        // the register set that was loaded during tile kt-1 into stage (kt+1)&1 (free since interval 8kt, see above), after having
        // re-armed the OTHER set with the global loads of tile kt+2.  The LDS writes are drained (lgkmcnt) before the barrier that
        // ends the tile; global loads stay in flight across it.
        auto tile = [&](int kt, auto par_c) {
            constexpr int PAR = decltype(par_c)::value;                 // kt & 1: tile kt+1 lives in stg[PAR ^ 1], tile kt+2 goes to stg[PAR]
            const bool more = kt + 1 < nk, more2 = kt + 2 < nk;
#pragma unroll
            for (int pp = 0; pp < KS; ++pp) {
                G2_READ(fa0, fw0, kt, pp)
                __builtin_amdgcn_sched_barrier(0);
                if (pp == 0 && more) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // tile kt+1 has been in flight for a whole tile
                    if (more2) {
#pragma unroll
                        for (int j = 0; j < LPT; ++j) load_one(kt + 2, j, stg[PAR]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (more && pp < 3) {
#pragma unroll
                    for (int j = 0; j < LPT; ++j)
                        if (j / ((LPT + 2) / 3) == pp) store_one(kt + 1, j, stg[PAR ^ 1]);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (pp == KS - 1 && more && grpB) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // my LDS writes of tile kt+1 are done
                __builtin_amdgcn_s_barrier();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                G2_MMA(fa0, fw0)
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        for (int kt = 0; kt < nk; kt += 2) {
            tile(kt, std::integral_constant<int, 0>{});
            if (kt + 1 < nk) tile(kt + 1, std::integral_constant<int, 1>{});
        }
        } else if constexpr (PH == 1) {
#pragma unroll 1
        for (int kt = 0; kt < nk; ++kt) {
            const bool more = kt + 1 < nk;
#pragma unroll
            for (int pp = 0; pp < KS; ++pp) {
                G2_READ(fa0, fw0, kt, pp)
                __builtin_amdgcn_sched_barrier(0);
                if (more && pp < 3) {
                    // Pieces of tile kt+1: spread over the first three R phases (LPT = 8: 3 + 3 + 2).  Front-loading them (all in the first
                    // R phase, 7 + 3 for the 10-piece tiles; flags bit9) wins 2-10 % on every shape when one GEMM is timed in a loop, but
                    // LOSES 1 % inside the real step (GEMM 1027 -> 1014 TF/s, alternating runs, both dtypes), so the spread stays.
                    const bool front = PROF && ((g.flags & 512) || ((g.flags & 32768) && g.K >= 2048));       // bit15: front-load long-K shapes only (A/B)
#pragma unroll
                    for (int j = 0; j < LPT; ++j)
                        if (front ? (LPT <= 8 ? pp == 0 : (j < 7 ? pp == 0 : pp == 1)) : (j / ((LPT + 2) / 3) == pp)) issue_one(kt + 1, j);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (pp == KS - 1 && more && grpB && !(PROF && (g.flags & 4))) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (flags bit2: profiling, wrong results)
                __builtin_amdgcn_s_barrier();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                G2_MMA(fa0, fw0)
                __builtin_amdgcn_sched_barrier(0);
                if (pp == KS - 1 && more && !grpB && !(PROF && (g.flags & 4))) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            }
            if (kt < 24) stamp(2 + kt);
        }
        } else {
        // PH == 2: two k-steps (16 MFMAs, ~512 pipe cycles) per interval, 4 barriers per K-tile.  A: R(kt,q) in interval
        // 4kt+2q, M in 4kt+2q+1; B one interval later.  Stage (kt+1)&1 is free from interval 4kt on; it must be complete
        // before interval 4kt+4: drain at the end of 4kt+3 (A after its M(kt,1), B after its R(kt,1)).  B therefore issues all
        // its DMA in its first R phase (interval 4kt+1); A splits 4 + 4 over intervals 4kt and 4kt+2.
        for (int kt = 0; kt < nk; ++kt) {
            const bool more = kt + 1 < nk;
#pragma unroll
            for (int q = 0; q < KS / 2; ++q) {
                G2_READ(fa0, fw0, kt, 2 * q)
                G2_READ(fa1, fw1, kt, 2 * q + 1)
                __builtin_amdgcn_sched_barrier(0);
                if (more) {
#pragma unroll
                    for (int j = 0; j < LPT; ++j)
                        if (grpB ? (q == 0) : (j / 4 == q)) issue_one(kt + 1, j);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (q == KS / 2 - 1 && more && grpB) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                G2_MMA(fa0, fw0)
                G2_MMA(fa1, fw1)
                __builtin_amdgcn_sched_barrier(0);
                if (q == KS / 2 - 1 && more && !grpB) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        }
        if (!grpB) __builtin_amdgcn_s_barrier();          // pair group B's extra barrier
    } else {
    wait_tile(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (DIST < nk) issue(DIST);
    G2_READ(fa0, fw0, 0, 0)
    // The two waves that share a SIMD (wave i and wave i+4) issue their DMA bursts at DIFFERENT points of the
    // iteration -- right after the barrier vs. after the first k-step pair -- so that one wave's DMA issue (tens of
    // cycles per instruction) is covered by its sibling's MFMAs instead of both stalling the matrix pipe together.
    const bool early = wave < 4;
    for (int kt = 0; kt < nk; ++kt) {
#pragma unroll
        for (int kp = 0; kp < KS / 2; ++kp) {            // k-step pair (2kp, 2kp+1): F0 holds step 2kp on entry
            G2_READ(fa1, fw1, kt, 2 * kp + 1)
            __builtin_amdgcn_sched_barrier(0);
            G2_WAIT_OLDER()                              // F0 landed (F1 may be in flight)
            G2_MMA(fa0, fw0)
            __builtin_amdgcn_sched_barrier(0);
            if (kp == 0 && !early && kt + DIST < nk && kt > 0) {
                issue(kt + DIST);                        // late half: stage (kt+DIST)%STAGES == (kt-1)%STAGES, free since barrier kt
                __builtin_amdgcn_sched_barrier(0);
            }
            if (kp + 1 < KS / 2) {
                G2_READ(fa0, fw0, kt, 2 * kp + 2)
                __builtin_amdgcn_sched_barrier(0);
                G2_WAIT_OLDER()                          // F1 landed
            } else if (kt + 1 < nk) {
                wait_tile(kt + 1);
                // every ds_read of stage kt (incl. F1 above) must have completed before another wave may overwrite it
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (!(PROF && (g.flags & 8))) __builtin_amdgcn_s_barrier();      // flags&8: profiling only (wrong results)
                __builtin_amdgcn_sched_barrier(0);
                if (early && kt + 1 + DIST < nk) issue(kt + 1 + DIST);   // into stage (kt+1+DIST) % STAGES == kt % STAGES
                G2_READ(fa0, fw0, kt + 1, 0)
                __builtin_amdgcn_sched_barrier(0);
            } else if (ASMR) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // last k-step of the last tile: F1 landed
                __builtin_amdgcn_sched_barrier(0);
            }
            G2_MMA(fa1, fw1)
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    }   // !PH
#undef G2_READ
#undef G2_MMA
#undef G2_WAIT_OLDER

    // ---- epilogue ---------------------------------------------------------------------------------------------
    stamp(28);
    if (MODE == MODE_F32 && acc_from_res && !(PROF && (g.flags & 16))) {
        // the residual is already in the accumulator: write-only fp32 tile through the LDS transpose -> full-row stores
        // (+1.5...4 % over per-lane 16-B stores; flags bit4 selects the direct form for A/B)
        __builtin_amdgcn_s_barrier();
        GemmArgs ge = g;
        ge.res = nullptr;
        constexpr int SB = 32 * NT * 128;
        if constexpr (NRM) {
            // ---- fused-norm PRODUCER: final values (residual + acc + bias) are in registers in the MFMA layout --------------------
            const int frow_ = lane & 31, fh_ = lane >> 5;
            if (g.bias) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) {
                        const int n = n0 + wc * (NT * 32) + nt * 32 + q4 * 8 + fh_ * 4;
                        if (n < g.N) {
                            const f32x4 bv = *(const f32x4*)(g.bias + n);
#pragma unroll
                            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                                for (int e = 0; e < 4; ++e) acc[nt][mt][q4 * 4 + e] += bv[e];
                        }
                    }
                ge.bias = nullptr;
            }
            // partial row statistics over this wave's NT*32 columns: a row's values sit in lanes l and l^32; (mean, M2 about
            // that mean) per partial, combined by norm_stats_finalize with Chan's formula (no E[x^2] - mean^2 cancellation)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                float sm = 0.f;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) sm += acc[nt][mt][r];
                sm += __shfl_xor(sm, 32, 64);
                const float mean = sm * (1.0f / (NT * 32));
                float q = 0.f;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) { const float d = acc[nt][mt][r] - mean; q += d * d; }
                q += __shfl_xor(q, 32, 64);
                const int m = m0 + wr * (MT * 32) + mt * 32 + frow_;
                // nrm_P = N / (NT*32) real partials per row (host); a wave column block past N (last tile of a ragged N) has none
                const int pidx = tn * WN + wc;
                if (fh_ == 0 && m < g.M && pidx < g.nrm_P) *(f32x2*)(g.nrm_part + ((size_t)m * g.nrm_P + pidx) * 2) = f32x2{mean, q};
            }
            epilogue_staged<T, MODE_F32, MT, NT>(ge, acc, smem + wave * SB, m0 + wr * (MT * 32), n0 + wc * (NT * 32), lane);
            // 16-bit copy scaled by the NEXT norm's weight: the operand of the GEMM behind that norm
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const int n = n0 + wc * (NT * 32) + nt * 32 + q4 * 8 + fh_ * 4;
                    f32x4 gv = {0.f, 0.f, 0.f, 0.f};
                    if (n < g.N) gv = *(const f32x4*)(g.nrm_gamma + n);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[nt][mt][q4 * 4 + e] *= gv[e];
                }
            GemmArgs gx = ge;
            gx.out = g.nrm_xh; gx.ldo = g.nrm_ldxh; gx.act = VITG_ACT_NONE;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                  // the 16-bit staging windows of the waves overlap their fp32 windows
            constexpr int HALF = (MT / 2) * 32 * (NT * 64 + ((NT & (NT - 1)) ? 16 : 0));
            static_assert(MT % 2 == 0 && 8 * HALF <= STAGES * STAGE, "16-bit staging of the fused-norm producer must fit in the ring");
            epilogue_staged<T, MODE_STORE16, MT, NT, 0, MT / 2>(gx, acc, smem + wave * HALF, m0 + wr * (MT * 32), n0 + wc * (NT * 32), lane);
            epilogue_staged<T, MODE_STORE16, MT, NT, MT / 2, MT / 2>(gx, acc, smem + wave * HALF, m0 + wr * (MT * 32) + (MT / 2) * 32,
                                                                       n0 + wc * (NT * 32), lane);
        } else {
        epilogue_staged<T, MODE_F32, MT, NT>(ge, acc, smem + wave * SB, m0 + wr * (MT * 32), n0 + wc * (NT * 32), lane);
        }
    } else if constexpr (MODE == MODE_F32) {
        // fp32 residual update: a lane already owns 16 contiguous bytes per accumulator quad; direct read-modify-write
        // measured faster than staging (the staged form serialises LDS round trip + residual load latency per pass).
        const int frow_ = lane & 31, fh_ = lane >> 5;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int m = m0 + wr * (MT * 32) + mt * 32 + frow_;
            if (m >= g.M) continue;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const int n = n0 + wc * (NT * 32) + nt * 32 + q4 * 8 + fh_ * 4;
                    if (n >= g.N) continue;
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[nt][mt][q4 * 4 + e];
                    if (acc_from_res) {                  // residual already inside the accumulator: bias + store only
                        if (g.bias) v += *(const f32x4*)(g.bias + n);
                        *(f32x4*)((float*)g.out + (size_t)m * g.ldo + n) = v;
                    } else {
                        epilogue_quad<T, MODE>(g, m, n, v);
                    }
                }
        }
    } else {
        // 16-bit outputs: transposed through wave-private LDS into full-row 16-B stores (gemm_common.h)
        __builtin_amdgcn_s_barrier();                    // every wave is done reading the ring before it is reused
        constexpr int STAGE_BYTES = MT * 32 * (MODE == MODE_SWIGLU ? 1 : NT) * 64;
        if constexpr (8 * STAGE_BYTES <= STAGES * STAGE) {
            // flags bit18 (profiling, wrong results): every tile stores over tile (0,0) - same store instructions, but the output
            // stays in L2 (no HBM write-back stream, no eviction of operand panels): separates issue cost from memory-side cost
            const int m0e = (PROF && (g.flags & 0x40000)) ? 0 : m0, n0e = (PROF && (g.flags & 0x40000)) ? 0 : n0;
            epilogue_staged<T, MODE, MT, NT, 0, MT, NRM>(g, acc, smem + wave * STAGE_BYTES, m0e + wr * (MT * 32), n0e + wc * (NT * 32), lane);
        } else {
            // 256x384 tile: the wave's 128x96 outputs are staged in two 64-row halves (the same wave writes and reads its
            // private region, in order, so the second half may reuse it)
            constexpr int HALF = (MT / 2) * 32 * (NT * 64 + ((NT & (NT - 1)) ? 16 : 0));     // padded rows when NT is not a power of two
            static_assert(MT % 2 == 0 && 8 * HALF <= STAGES * STAGE, "epilogue staging must fit in the ring");
            epilogue_staged<T, MODE, MT, NT, 0, MT / 2, NRM>(g, acc, smem + wave * HALF, m0 + wr * (MT * 32), n0 + wc * (NT * 32), lane);
            epilogue_staged<T, MODE, MT, NT, MT / 2, MT / 2, NRM>(g, acc, smem + wave * HALF, m0 + wr * (MT * 32) + (MT / 2) * 32,
                                                                    n0 + wc * (NT * 32), lane);
        }
    }
    stamp(29);
    if constexpr (PROF) {
        // gemm_flags bit26 (A/B): keep the blocks of an XCD in the same ROUND of the tile list.  The 32 blocks of an XCD work on a
        // compact group_m x (32 / group_m) super-tile and share its operand panels through the 4 MiB L2 only while they are within a
        // few K-tiles of each other; free-running they drift apart and the same panel is fetched again and again (fc1: 8.6x its
        // operand bytes through L2 misses, profiles/r02_pmc_traffic_v9_fp16.md).  Timing only - no data is handed over, so relaxed
        // atomics without fences; the spin is bounded, so a block that is not co-resident can never hang the others.
        if ((g.flags & 0x4000000) && g.xcd_sync) {
            const int round_ = (vb - (int)blockIdx.x) / (int)gridDim.x;
            const int full_rounds = nwg_ / (int)gridDim.x;
            if (round_ + 1 < full_rounds && threadIdx.x == 0) {
                unsigned* c = g.xcd_sync + (blockIdx.x & 7) * 32;
                __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned target = (unsigned)((gridDim.x >> 3) * (round_ + 1));
                for (int spin = 0; spin < 4096 && __hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target; ++spin)
                    __builtin_amdgcn_s_sleep(8);
            }
        }
    }
    if (vb + (int)gridDim.x < nwg_) {                    // another tile follows: every wave is done with the ring / its staging region
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }
    stamp(30);
    }   // tile loop
}

#if VITG_PROFILING_BUILD
// ---------------------------------------------------------------------------------------------------------------------
// Persistent variant of the 256x256x64 ring-2 configuration: one block per CU walks a static, XCD-contiguous list of
// tiles.  The K-tile ring simply continues across tile boundaries (the last iteration of a tile prefetches K-tile 0 of the
// next one), so there is no per-tile block launch, no address set-up stall and no pipeline refill; the epilogue's stores
// drain while the next tile's first K-tile is already in LDS.  Requires an even number of K-tiles (stage parity).
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void linear_to_tile(int L, int tiles_m, int tiles_n, int gm, int& tm, int& tn) {
    const int per_group = gm * tiles_n;
    const int grp = L / per_group;
    const int first_m = grp * gm;
    const int gsz = min(tiles_m - first_m, gm);
    const int in_g = L - grp * per_group;
    tm = first_m + in_g % gsz;
    tn = in_g / gsz;
}

template <typename T, int MODE>
__global__ __launch_bounds__(G2_THREADS, 2) void gemm256p_kernel(GemmArgs g) {
    typedef typename T16<T>::vec8 vec8;
    constexpr int BN = 256, BK = 64, MT = 4, NT = 2, KS = 4, ROWB = 128, CPR = 8, RPI = 8;
    constexpr int A_BYTES = G2_BM * ROWB, W_BYTES = BN * ROWB, STAGE = A_BYTES + W_BYTES;
    constexpr int A_INSTR = 4, W_INSTR = 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const bool early = wave < 4;

    // ---- static tile schedule: XCD x = bid % 8 owns a contiguous chunk of the (grouped) tile order ------------------
    const int ntiles = g.tiles_m * g.tiles_n;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
    const int cq = ntiles >> 3, cr = ntiles & 7;
    const int chunk0 = xcd < cr ? xcd * (cq + 1) : cr * (cq + 1) + (xcd - cr) * cq;
    const int chunkn = cq + (xcd < cr ? 1 : 0);
    if (slot >= chunkn) return;                         // whole block exits together

    const int drow = lane / CPR, dslot = lane % CPR;
    // Source addressing = wave-uniform 64-bit tile base (SGPRs) + per-lane 32-bit byte offset inside the tile (one VGPR per
    // DMA instruction): half the address registers of per-lane pointers, and the saddr+voffset form needs no VALU.
    const char* a_base; const char* w_base;              // uniform: A + m0*lda*2, W + n0*ldw*2
    unsigned a_cur[A_INSTR], w_cur[W_INSTR];
    auto set_src = [&](int L, int& m0, int& n0) {
        int tm, tn;
        linear_to_tile(L, g.tiles_m, g.tiles_n, 8, tm, tn);
        m0 = tm * G2_BM;
        n0 = tn * BN;
        a_base = (const char*)g.A + (size_t)m0 * g.lda * 2;
        w_base = (const char*)g.W + (size_t)n0 * g.ldw * 2;
#pragma unroll
        for (int j = 0; j < A_INSTR; ++j) {
            const int R = (wave * A_INSTR + j) * RPI + drow;
            a_cur[j] = (unsigned)(min(R, g.M - 1 - m0) * g.lda + ((dslot ^ ((R >> 1) & 7)) << 3)) * 2u;
        }
#pragma unroll
        for (int j = 0; j < W_INSTR; ++j) {
            const int R = (wave * W_INSTR + j) * RPI + drow;
            w_cur[j] = (unsigned)(min(R, g.N - 1 - n0) * g.ldw + ((dslot ^ ((R >> 1) & 7)) << 3)) * 2u;
        }
    };
    auto issue = [&](int kt) {
        char* st = smem + (kt & 1) * STAGE;
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
    int a_off[KS], w_off[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int ch = ((ks * 2 + fh) ^ ((frow >> 1) & 7)) << 4;
        a_off[ks] = (wr * 128 + frow) * ROWB + ch;
        w_off[ks] = A_BYTES + (wc * 64 + frow) * ROWB + ch;
    }
    const int nk = g.K / BK;                            // even (checked by the host)

    // ONE live set of source offsets: it is switched to the next tile at the moment the ring wraps (the current tile has
    // no DMA left to issue by then); the epilogue only needs the tile origin (m0, n0), kept separately.
    int m0, n0, m0n = 0, n0n = 0;
    int it = slot;
    set_src(chunk0 + it, m0, n0);
    // De-phase the XCDs.  Every tile takes the same time, so without this all 256 CUs reach their epilogue together:
    // HBM idles during the main loops and then saturates on a 33 MB write burst (measured: time = MFMA time + output bytes /
    // ~3 TB/s, i.e. no overlap at all).  Delaying XCD x by x/8 of a tile period spreads the bursts evenly while the 32
    // blocks of one XCD stay in lockstep and keep sharing operand panels in their L2.
    if (g.stagger_ns > 0 && xcd > 0) {
        const unsigned long long t0 = wall_clock64();                      // 100 MHz
        const unsigned long long ticks = (unsigned long long)g.stagger_ns * xcd / 80;   // ns -> 10-ns ticks, x/8
        while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
    }
    issue(0);

#define GP_READ(FA, FW, KT, KS_)                                                                              \
    {                                                                                                         \
        const char* st_ = smem + ((KT) & 1) * STAGE;                                                          \
        _Pragma("unroll") for (int i = 0; i < NT; ++i) FW[i] = *(const vec8*)(st_ + w_off[KS_] + i * (32 * ROWB)); \
        _Pragma("unroll") for (int i = 0; i < MT; ++i) FA[i] = *(const vec8*)(st_ + a_off[KS_] + i * (32 * ROWB)); \
    }
#define GP_MMA(FA, FW)                                                                                        \
    __builtin_amdgcn_s_setprio(1);                                                                            \
    _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)                                                         \
        _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) acc[nt][mt] = T16<T>::mfma32(FW[nt], FA[mt], acc[nt][mt]); \
    __builtin_amdgcn_s_setprio(0);

    bool first = true;
    while (true) {
        const bool has_next = it + per_xcd < chunkn;
        f32x16 acc[NT][MT];
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int j = 0; j < MT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        vec8 fa0[MT], fw0[NT], fa1[MT], fw1[NT];

        // K-tile 0 of this tile has landed: the prologue waits for it below, and later tiles wait for it BEFORE the previous
        // tile's epilogue (vmcnt is in-order and counts stores, so waiting here would also drain the epilogue's stores;
        // waiting before them lets the stores retire underneath the first iteration instead).
        if (first) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        first = false;
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        issue(1);                                        // every wave (the late half starts staggering at kt = 1); nk >= 4
        GP_READ(fa0, fw0, 0, 0)
        for (int kt = 0; kt < nk; ++kt) {
#pragma unroll
            for (int kp = 0; kp < KS / 2; ++kp) {
                GP_READ(fa1, fw1, kt, 2 * kp + 1)
                __builtin_amdgcn_sched_barrier(0);
                GP_MMA(fa0, fw0)
                __builtin_amdgcn_sched_barrier(0);
                if (kp == 0 && !early && kt > 0) {       // late half: K-tile kt+1 (possibly K-tile 0 of the next tile)
                    if (kt + 1 < nk) issue(kt + 1);
                    else if (has_next) { set_src(chunk0 + it + per_xcd, m0n, n0n); issue(0); }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (kp + 1 < KS / 2) {
                    GP_READ(fa0, fw0, kt, 2 * kp + 2)
                    __builtin_amdgcn_sched_barrier(0);
                } else if (kt + 1 < nk) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                    if (early) {                          // early half: K-tile kt+2 (possibly K-tile 0 of the next tile)
                        if (kt + 2 < nk) issue(kt + 2);
                        else if (has_next) { set_src(chunk0 + it + per_xcd, m0n, n0n); issue(0); }
                    }
                    GP_READ(fa0, fw0, kt + 1, 0)
                    __builtin_amdgcn_sched_barrier(0);
                }
                GP_MMA(fa1, fw1)
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // nk == 1 cannot happen (nk even); for nk == 2 the late half issued nothing at kt == 0: handled because early/late
        // both cover "tile kt+1 during iteration kt" for kt >= 1 and the prologue/previous tile covered tile 1... see host check nk >= 4.

        // ---- epilogue (stage 1 = last K-tile's stage is free after the barrier; stage 0 holds the next tile's K-tile 0)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // next tile's K-tile 0 (issued during the last iteration) landed
        if constexpr (MODE == MODE_F32) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int m = m0 + wr * 128 + mt * 32 + frow;
                if (m >= g.M) continue;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) {
                        const int n = n0 + wc * 64 + nt * 32 + q4 * 8 + fh * 4;
                        if (n >= g.N) continue;
                        f32x4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = acc[nt][mt][q4 * 4 + e];
                        epilogue_quad<T, MODE>(g, m, n, v);
                    }
            }
        } else {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();               // all waves finished reading stage 1
            char* stg = smem + STAGE + wave * 8192;     // 8 KiB per wave inside stage 1
            epilogue_staged<T, MODE, MT, NT, 0, 2>(g, acc, stg, m0 + wr * 128, n0 + wc * 64, lane);
            epilogue_staged<T, MODE, MT, NT, 2, 2>(g, acc, stg, m0 + wr * 128 + 64, n0 + wc * 64, lane);
        }
        if (!has_next) break;
        it += per_xcd;
        m0 = m0n;
        n0 = n0n;
    }
#undef GP_READ
#undef GP_MMA
}

template <typename T, int MODE>
static int launch256p(GemmArgs g, hipStream_t s) {
    constexpr int lds = 2 * (G2_BM * 128 + 256 * 128);
    static int done_dev = -1;
    static int cus = 256;
    int cur_dev = 0;
    (void)hipGetDevice(&cur_dev);
    if (cur_dev != done_dev) {
        (void)hipFuncSetAttribute((const void*)gemm256p_kernel<T, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        hipDeviceProp_t p;
        if (hipGetDeviceProperties(&p, cur_dev) == hipSuccess) cus = p.multiProcessorCount & ~7;
        done_dev = cur_dev;
    }
    g.tiles_m = ceil_div(g.M, G2_BM);
    g.tiles_n = ceil_div(g.N, 256);
    const int ntiles = g.tiles_m * g.tiles_n;
    const int grid = ntiles >= cus ? cus : ((ntiles + 7) & ~7);
    // one tile period ~ 1.15 us per 64-deep K-tile (measured main-loop rate) + epilogue; only worth it with several rounds
    const int stag = RESEARCH(gemm_stagger);            // default off: measured no gain
    g.stagger_ns = (stag && ntiles >= 4 * cus) ? (g.K / 64) * 1150 + 4000 : 0;
    hipLaunchKernelGGL((gemm256p_kernel<T, MODE>), dim3(grid), dim3(G2_THREADS), lds, s, g);
    return (int)hipGetLastError();
}

#endif   // VITG_PROFILING_BUILD (persistent cfg 4)

template <typename T, int MODE, int BN, int WM, int STAGES, int BK, bool ASMR = false, int PH = 0, int BM = G2_BM, bool NRM = false>
static int launch256(GemmArgs g, hipStream_t s) {
    constexpr int lds = STAGES * (BM * BK * 2 + BN * BK * 2);
    static std::atomic<unsigned> done_mask{0};   // per device: the attribute belongs to the device's copy of the kernel
    const int cur_dev = current_device();
    if (!device_done(done_mask, cur_dev)) {
        (void)hipFuncSetAttribute((const void*)gemm256_kernel<T, MODE, BN, WM, STAGES, BK, ASMR, PH, BM, NRM>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        mark_device_done(done_mask, cur_dev);
    }
    g.tiles_m = ceil_div(g.M, BM);
    g.tiles_n = ceil_div(g.N, BN);
    // M-tiles per group of the tile order (the 32 blocks resident on an XCD then cover group_m x 32/group_m tiles sharing operand
    // panels in its L2): 4 measured best (+0.5 % on the step over 8; 16 and 32 lose 10-20 %, profiles/r01_gemm_shapes.md section 7)
    g.group_m = RESEARCH(gemm_group) > 0 ? RESEARCH(gemm_group) : 4;
    // one block per CU walking the tile list ("software relaunch") when there are at least two tiles per CU and one block fills a CU
    const int cus = device_cu_count(cur_dev);
    const int nwg = g.tiles_m * g.tiles_n;
    // default since round 2: in-step A/B in fp16, two alternations: 734.8 -> 732.7 ms, GEMM 1073 -> 1079 TF/s (gpurun_out/r02_ab_flags.log);
    // gemm_flags bit24 switches back to one hardware block per tile (A/B and the bit-identity test)
    const bool loop = BN >= 256 && nwg >= 2 * cus && !(RESEARCH(gemm_flags) & 0x1000000);
    const int splits = (MODE == MODE_F32 && g.split_k > 1) ? g.split_k : 1;
    // gemm_stagger (default off: +6 % on the out-proj shape timed alone, 725.8 / 731.8 vs 734.7 / 732.6 ms in the step, tools/ab_stagger.sh;
    // 1 = estimate the tile period, > 1 = tile period in ns): de-phase the tile groups of the fp32-residual shapes
    const int stag = RESEARCH(gemm_stagger);
    g.stagger_ns = (MODE == MODE_F32 && loop && splits == 1 && stag > 0) ? (stag > 1 ? stag : (g.K / 64) * 2400 + 40000) : 0;
#if VITG_PROFILING_BUILD
    g.xcd_sync = nullptr;
    if ((g.flags & 0x4000000) && loop && splits == 1 && (cus & 7) == 0) {
        static unsigned* sync_buf = nullptr;
        if (!sync_buf) (void)hipMalloc((void**)&sync_buf, 8 * 32 * sizeof(unsigned));
        if (sync_buf) { (void)hipMemsetAsync(sync_buf, 0, 8 * 32 * sizeof(unsigned), s); g.xcd_sync = sync_buf; }
    }
#endif
    hipLaunchKernelGGL((gemm256_kernel<T, MODE, BN, WM, STAGES, BK, ASMR, PH, BM, NRM>), dim3(loop && splits == 1 ? cus : nwg, splits), dim3(G2_THREADS), lds, s, g);
    return (int)hipGetLastError();
}

bool gemm256_applicable(const GemmArgs& g, int mode) {
    (void)mode;
    // staged epilogue: 16-B row stores need 16-B aligned rows (8 halfs / 4 floats)
    const bool aligned = g.ldo % 8 == 0 && (g.res == nullptr || g.ldr % 4 == 0) && g.N % 8 == 0;
    return g.M >= 256 && g.N >= 128 && g.K % 64 == 0 && aligned;
}


// Tile-shape selection.  Cost model: a launch runs ceil(tiles / CUs) rounds of one tile per CU (one block per CU for every
// configuration but cfg 2), a round costs tile area / relative main-loop efficiency.  The efficiencies are same-call A/B results
// (profiles/r01_gemm_shapes.md sections 5-8); what the model adds is the round quantisation, which decides the small-M shapes of
// the context-parallel LLM (M = 1660 at 8 GPUs: gate/up takes 3 rounds of 384x256 instead of 5 of 256x256, +24 % measured) and
// keeps the wide tiles off shapes where they would leave a nearly empty last round (down-proj at M = 6460: 256x384 -25 %).
//   cfg 6  256x256 phased                      1.00
//   cfg 2  256x128 (wave tile 64x64)           0.65 (fitted to the M = 1660 / 6460 measurements of tools/gemm_small_m.py)
//   cfg 9  256x384 (wave tile 128x96)          fp32-residual: 1.07 for K >= 4096, else 1.00; 16-bit outputs 1.00 (qkv: -1.3 % in an
//                                              isolated loop, +3.5 % inside the step - the step decides); no SwiGLU
//   cfg 10 384x256 (wave tile 192x64)          SwiGLU 1.07; 16-bit outputs 1.08 (fc1 / qkv: -3 % in an isolated loop but +1.3 % of the
//                                              WHOLE step in alternating bench.py runs - the step decides); fp32-residual unusable (spills)
// cfg 6 is kept unless another configuration is predicted at least 3 % faster.
// MFMA shape of the three wide tiles: v_mfma_f32_16x16x32 (gemm_mi16.hip) unless gemm_mi = 32 asks for the 32x32x16 kernels of this file
// (A/B, tests); gemm_mi_modes restricts it per epilogue mode (A/B)
static bool gemm_mi16_selected(int mode) {
    if (RESEARCH(gemm_mi) == 32) return false;
    const int mm = RESEARCH(gemm_mi_modes);
    return mm == 0 || ((mm >> mode) & 1);
}
static int choose_cfg(const GemmArgs& g, int mode) {
    const int cus = device_cu_count(current_device());
    auto cost = [&](int bm, int bn, double eff, int blocks_per_cu) {
        const long tiles = (long)ceil_div(g.M, bm) * ceil_div(g.N, bn);
        const long rounds = (tiles + (long)cus * blocks_per_cu - 1) / ((long)cus * blocks_per_cu);
        return (double)rounds * blocks_per_cu * bm * bn / eff;
    };
    const bool wide_ok = !(RESEARCH(gemm_flags) & 256);               // flags bit8: A/B switch (wide tiles off)
    double best = cost(256, 256, 1.00, 1);
    int cfg = 6;
    if (g.N < 256) return 2;
    {   // gemm_pref (A/B): force a tile shape per epilogue mode when the shape admits it
        const int pref = RESEARCH(gemm_pref), d = mode == MODE_STORE16 ? pref / 100 % 10 : mode == MODE_F32 ? pref / 10 % 10 : pref % 10;
        if (d == 1) return 6;
        if (d == 2 && mode != MODE_SWIGLU && g.N >= 384) return 9;
        if (d == 3 && g.M >= 384) return 10;
    }
    const double c2 = cost(256, 128, 0.65, 1);
    if (c2 < 0.97 * best) { best = c2; cfg = 2; }
    if (wide_ok && mode != MODE_SWIGLU && g.N >= 384) {
        // residual updates (fp32 stream, or the 16-bit stream of vitg_gemm_res16) behave alike: the wide-N tile pays for K >= 4096 (fc2, down)
        const bool resid = mode == MODE_F32 || (mode == MODE_STORE16 && (g.res || g.res16));
        // 16-bit outputs on the 16x16x32 kernels: the 128x96 wave tile leads on the tower's qkv / fc1 (1142 / 1024 against 1096 / 1041 for
        // 384x256 and 1042 / 994 for 256x256 TF/s, +0.7 % on the step over 384x256: profiles/r03_gemm_mi16.md)
        const double e = resid ? ((g.K >= 4096 || (RESEARCH(gemm_flags) & 8192)) ? 1.07 : 1.00)
                               : (gemm_mi16_selected(mode) ? 1.10 : ((RESEARCH(gemm_flags) & 2048) ? 0.90 : 1.00));   // flags bit11: A/B (16-bit outputs off the wide tile)
        const double c9 = cost(256, 384, e, 1);
        if (c9 < 0.97 * best) { best = c9; cfg = 9; }
    }
    if (wide_ok && g.M >= 384 && !(mode == MODE_F32 && (RESEARCH(gemm_flags) & 16384))) {
        // fp32-residual GEMMs on the tall tile since the saddr addressing freed its registers (it used to spill): o-proj at M = 12860
        // +6 %, and at M = 6460 (two context-parallel ranks) o / down fit ONE round of 384x256 tiles: +24 % / +48 % (tools/gemm_small_m.py).
        // flags bit12: A/B (16-bit outputs off the tall tile); bit14: A/B (fp32-residual GEMMs off it)
        const bool resid10 = mode == MODE_F32 || (mode == MODE_STORE16 && (g.res || g.res16));
        const double c10 = cost(384, 256, mode == MODE_SWIGLU ? 1.07 : resid10 ? 1.05 : (gemm_mi16_selected(mode) ? 1.05 : ((RESEARCH(gemm_flags) & 4096) ? 0.96 : 1.08)), 1);
        // the 3 % hysteresis protects the 256x256 default; between the two wide tiles the cheaper one wins outright (fc1: N = 4352 is 11.3 tiles of
        // 384 but 17 of 256 - 1033 against 1015 TF/s on the tall tile, while qkv, N = 9 x 384, stays on the wide one: 1131 against 1078)
        if (cfg == 9 ? c10 < best : c10 < 0.97 * best) { best = c10; cfg = 10; }
    }
    if (wide_ok && g.M > 256 && g.M <= 4096 && g.N >= 256 && (mode == MODE_SWIGLU || mode == MODE_F32) && gemm_mi16_selected(mode)) {
        // 320 x 256 (round 5): few-row shapes whose M pads badly on 256 / 384 rows - one context-parallel rank's 1600 rows are 5 x 320 exactly where
        // the 384-row tile computes 1920.  Wave tile 160 x 64: between the 128 x 64 of 256 x 256 and the 192 x 64 of 384 x 256 in fragment bytes per flop
        const double c12 = cost(320, 256, mode == MODE_SWIGLU ? 1.04 : 1.03, 1);
        if (c12 < 0.97 * best) { best = c12; cfg = 12; }
    }
    return cfg;
}

// split-K launches (vitg_gemm_splitk): rows per tile, 256 or 320.  With fewer tiles than CUs the launch is ONE round of (tiles x splits) blocks and its
// time is one block's K range times its rows: 98 tiles of 256 rows x 2 splits (M = 1660, N = 3584) run K / 2 per block, 84 tiles of 320 rows x 3 splits
// run K / 3 - 17 % less work on the critical path for 252 of 256 CUs busy instead of 196.
int splitk_tile_rows(int M, int N, int K, int splits, int cus) {
    if (M <= 256 || splits < 2 || !gemm_mi16_selected(MODE_F32)) return 256;
    auto cost = [&](int bm) {
        const long blocks = (long)ceil_div(M, bm) * ceil_div(N, 256) * splits;
        const long rounds = (blocks + cus - 1) / cus;
        return (double)rounds * bm * ceil_div(K / 64, splits);
    };
    return cost(320) < 0.97 * cost(256) ? 320 : 256;
}

int gemm256_pick_cfg(const GemmArgs& g, int mode);

template <typename T, int MODE, bool NRM = false>
static int launch256_m(const GemmArgs& g, hipStream_t s) {
    int cfg = tuning().gemm_cfg;
    if (!cfg) cfg = choose_cfg(g, MODE);
    if constexpr (NRM) {                     // fused-norm instantiations exist for the four shipped tile shapes only
        if (cfg == 10 && g.M >= 384) return launch256<T, MODE, 256, 2, 2, 64, false, 1, 384, true>(g, s);
        if (cfg == 9 && MODE != MODE_SWIGLU && g.N >= 384) return launch256<T, MODE, 384, 2, 2, 64, false, 1, G2_BM, true>(g, s);
        if (cfg == 2 || g.N < 256) return launch256<T, MODE, 128, 4, 3, 32, false, 0, G2_BM, true>(g, s);
        return launch256<T, MODE, 256, 2, 2, 64, false, 1, G2_BM, true>(g, s);
    }
    if (MODE == MODE_F32 && g.split_k > 1)                    // split-K partials: 256x256 tiles, or 320x256 where vitg_gemm_splitk_plan planned with them
        cfg = splitk_tile_rows(g.M, g.N, g.K, g.split_k, device_cu_count(current_device())) == 320 ? 12 : 6;
    // MFMA shape (round 3): the three shipped tile shapes exist on v_mfma_f32_16x16x32 as well (gemm_mi16.hip), which the package power
    // cap lets run 14 % faster than 32x32x16 on real operands (profiles/r03_mfma_shape.md)
    // (the residual forms of the 16-bit-output mode - vitg_gemm_res16 - exist in gemm_mi16.hip and the 128-tile kernel only; gemm_entry
    // never sends them here with the 256x128 tile)
    // (fp32 output WITH an activation - no caller in the engine - stays on this file's kernels: gemm_mi16.hip's fp32 epilogue carries none)
    if ((cfg == 6 || cfg == 9 || cfg == 10 || cfg == 12) && !(MODE == MODE_F32 && g.act != VITG_ACT_NONE) &&
        (gemm_mi16_selected(MODE) || (MODE == MODE_STORE16 && (g.res || g.res16))))
        return launch_gemm_mi16(g, std::is_same<T, bf16_t>::value ? DT_BF16 : DT_F16, MODE, cfg, s);
#if VITG_PROFILING_BUILD
    if (cfg == 8 && g.N >= 256 && gemm_w4_applicable(g, MODE)) return launch_gemm_w4(g, std::is_same<T, bf16_t>::value ? DT_BF16 : DT_F16, MODE, s);   // 4 waves x 128x128 (gemm_w4.hip)
#endif
#if !VITG_PROFILING_BUILD
    // PRODUCT library: this file ships ONE kernel, the 256x128x32 ring-3 tile (2 blocks / CU: the small-M shapes of a context-parallel rank, N < 256);
    // the wide tiles run on v_mfma_f32_16x16x32 (gemm_mi16.hip, above).  What else arrives here - fp32 output WITH an activation on a wide-tile
    // shape (no caller in the engine; gemm_mi16.hip's fp32 epilogue carries no activation) - runs on this kernel too: its staged epilogue has every form.
    return launch256<T, MODE, 128, 4, 3, 32>(g, s);
#else
    if (cfg == 10 && g.M >= 384) return launch256<T, MODE, 256, 2, 2, 64, false, 1, 384>(g, s);               // 384x256x64 phased (wave tile 192x64)
    if (cfg == 9 && MODE != MODE_SWIGLU && g.N >= 384) return launch256<T, MODE, 384, 2, 2, 64, false, 1>(g, s);   // 256x384x64 phased (wave tile 128x96)
    if (cfg == 2 || g.N < 256) return launch256<T, MODE, 128, 4, 3, 32>(g, s);
    // research configurations (profiles/r01_gemm_shapes.md): only in libvideoitg_hip_prof.so
    if (cfg == 4 && (g.K / 64) % 2 == 0 && g.K >= 256) return launch256p<T, MODE>(g, s);
    if (cfg == 1) return launch256<T, MODE, 256, 2, 4, 32>(g, s);
    if (cfg == 11) return launch256<T, MODE, 256, 2, 2, 64, false, 3>(g, s);    // 11 = phased, operands staged through registers instead of LDS-DMA
    if (cfg == 7) return launch256<T, MODE, 256, 2, 2, 64, false, 2>(g, s);     // 7 = phased, 2 k-steps per interval
    if (cfg == 5) return launch256<T, MODE, 256, 2, 2, 64, true>(g, s);       // 5 = inline-asm LDS reads + counted lgkmcnt (measured = cfg 3)
    if (cfg == 3) return launch256<T, MODE, 256, 2, 2, 64, false>(g, s);      // 3 = free-running ring-2 loop
    return launch256<T, MODE, 256, 2, 2, 64, false, 1>(g, s);                  // 6 = phased, 1 k-step per interval
#endif   // VITG_PROFILING_BUILD
#if 0
    // N = k*256 + 128 (SigLIP: 1152 = 4.5 tiles, 3456 = 13.5): a 256-wide last tile would be half padding.  Run the k*256
    // columns with the 256x256 kernel and the last 128 columns with the 256x128 kernel (two launches, same stream).
    // Measured: qkv +-0, out-proj -4 %, fc2 +2 % -> off by default (gemm_flags bit5 enables it).
    if (MODE != MODE_SWIGLU && g.N % 256 == 128 && g.N >= 384 && (RESEARCH(gemm_flags) & 32)) {
        GemmArgs a = g, b = g;
        a.N = g.N - 128;
        b.N = 128;
        b.W = (const char*)g.W + (size_t)a.N * g.ldw * 2;
        if (g.bias) b.bias = g.bias + a.N;
        b.out = (char*)g.out + (size_t)a.N * (MODE == MODE_F32 ? 4 : 2);
        if (g.res) b.res = g.res + a.N;
        const int rc = launch256<T, MODE, 256, 2, 2, 64, false>(a, s);
        if (rc) return rc;
        return launch256<T, MODE, 128, 4, 3, 32>(b, s);
    }
    return launch256<T, MODE, 256, 2, 2, 64, false>(g, s);
#endif
}

template <typename T>
static int launch256_t(const GemmArgs& g, int mode, hipStream_t s) {
#if VITG_PROFILING_BUILD
    // fused norm (measured SLOWER than the stand-alone norm kernel, profiles/r02_fused_norm.md): profiling library only
    if (g.nrm_gamma || g.nrm_stats) {        // producer (F32) or consumer (16-bit outputs)
        if (mode == MODE_SWIGLU) return launch256_m<T, MODE_SWIGLU, true>(g, s);
        if (mode == MODE_F32) return launch256_m<T, MODE_F32, true>(g, s);
        return launch256_m<T, MODE_STORE16, true>(g, s);
    }
#endif
    if (mode == MODE_SWIGLU) return launch256_m<T, MODE_SWIGLU>(g, s);
    if (mode == MODE_F32) return launch256_m<T, MODE_F32>(g, s);
    return launch256_m<T, MODE_STORE16>(g, s);
}

int gemm256_pick_cfg(const GemmArgs& g, int mode) {
    const int cfg = tuning().gemm_cfg;
    return cfg ? cfg : choose_cfg(g, mode);
}

// columns covered by one partial of the fused-norm producer for this shape (= BN / WN of the tile shape auto mode picks), 0 when the
// 256-tile kernels do not take the shape
int gemm256_norm_cols(const GemmArgs& g) {
    if (!gemm256_applicable(g, MODE_F32)) return 0;
    int cfg = tuning().gemm_cfg;
    if (!cfg) cfg = choose_cfg(g, MODE_F32);
    if (cfg == 9 && g.N >= 384) return 96;
    return 64;                                // 256x256, 384x256 (4 waves across 256 columns) and 256x128 (2 waves across 128)
}

int launch_gemm256(const GemmArgs& g, int dtype, int mode, hipStream_t s) {
    return dtype == DT_BF16 ? launch256_t<bf16_t>(g, mode, s) : launch256_t<f16_t>(g, mode, s);
}

}  // namespace vitg

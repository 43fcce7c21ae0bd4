// C[M,N] = A[M,K] * W[N,K]^T on v_mfma_f32_16x16x32 (round 3).  Same block tiles, LDS-DMA operand ring, phased schedule and fused
// epilogues as gemm256.hip's shipped configurations (256x256 / 256x384 / 384x256 x 64, 8 waves as 2 (M) x 4 (N)), with the
// 32x32x16 MFMA replaced by the 16x16x32 one.
//
// Why: on MI355X the matrix pipes run against the 1400 W package power cap, and what the cap lets them sustain depends on the MFMA
// SHAPE.  An MFMA-only kernel on N(0,1) fp16 operands holds 1.68 PFLOP/s with 32x32x16 and 1.92 PFLOP/s with 16x16x32 (bf16: 1.83 vs
// 2.05; tools/ubench/mfma_shape_power.hip, profiles/r03_mfma_shape.md) - the same flops, the same accumulator registers, +14 % rate at
// the same watts.  (hipBLASLt's gfx950 assembly kernel, MT256x256x64_MI16x16x1, is on that shape and was ahead of gemm256.hip by
// exactly that margin on the large-N shapes, profiles/r02_gemm_timeline.md.)
//
// What changes against gemm256.hip:
//   * fragments: a lane holds 8 consecutive k of ONE row - row = lane & 15, 16-B chunk (lane >> 4) of the 64-B (32-deep) k-step.  The
//     LDS image ([rows][8 chunks x 16 B], chunk index XOR ((row >> 1) & 7), written lane-linearly by global_load_lds_dwordx4 from
//     swizzled SOURCE addresses) is the one gemm256.hip uses: it is conflict-free for this read pattern too (each ds_read_b128 lane
//     group covers the 16 rows once, 8 distinct swizzled chunks per row parity).
//   * accumulators: f32x4 per 16x16 tile; lane holds features n0 + 4 (lane >> 4) .. + 3 of token m0 + (lane & 15)  (MFMA "A" = weight
//     rows, "B" = activation rows, as in gemm256.hip: 4 consecutive output features of one token per lane).
//   * schedule: a K-tile (64 deep) is 2 k-steps x (MT / MC) row chunks = NI intervals pairs [fragment reads + DMA issue | MC x NT MFMAs]
//     delimited by raw s_barriers; the weight fragments of a k-step are read once and kept over its row chunks.  Waves 4-7 run one
//     barrier behind waves 0-3, so on every SIMD one wave owns the matrix pipe while its sibling reads / issues DMA (the phased
//     schedule of gemm256.hip, cfg 6).  16 MFMAs of 16x16x32 = 256 pipe cycles per interval, as 8 of 32x32x16 there.
// Epilogues (bias, GELU tanh / erf, SwiGLU, fp32 residual preloaded into the accumulators, 16-bit or fp32 output) are restated for
// the 16x16 accumulator layout: 32 rows at a time through a wave-private LDS window into full-row 16-B stores.
// Late round 3 (profiles/r03_gemm_epilogue.md: 0.457 -> 0.50 of the MFMA roofline from this alone): every decision an epilogue needs - which
// activation, bias or not, interior or edge tile, which residual pointers - is taken ONCE per tile and selects a straight-line copy of the
// stage loop (lambda templates below); nothing is tested per accumulator quad.  16-bit-output GEMMs start their accumulators at the bias.
#include "gemm_common.h"
#include <type_traits>

namespace vitg {

constexpr int G16_THREADS = 512;
constexpr int WM4_DEFAULT_MODES = 0;   // epilogue modes whose 384x256 tile runs the 4 x 2 wave grid by default (gemm_wm4 = 0); see launch16_m
constexpr int MODE_RES16 = 3;      // internal: MODE_STORE16 with a residual preloaded into the accumulators (vitg_gemm_res16).  Its own instantiation:
                                   // compiled into the plain 16-bit-output kernels the preload cost the wide tiles 15-20 spilled registers

// SK (stream-K, vitg_gemm_streamk): for fp32-residual GEMMs with FEWER output tiles than CUs and a long K (o / down projection of one
// context-parallel rank: 98 tiles of 256x256 on 256 CUs).  The tiles' K-tiles are laid end to end (tile-major) and cut into equal runs of
// sk_units, one run per block: a block works on the tail of one tile and the head of the next (at most two segments, sk_units <= K-tiles
// per tile) and stores each segment's raw fp32 partial tile into its own workspace slot [block][segment][256][256]; a row-parallel kernel
// then adds a tile's segments in block order, the bias and the residual (fixed order, no atomics: run-to-run bit-identical).  Every CU gets
// the same number of K-tiles - split-K x2 left 60 of 256 CUs idle on these shapes.
// WM (wave rows of the 8-wave block, 2 or 4; wave columns WN = 8 / WM).  The 384x256 tile as 2 x 4 gives a wave 192 x 64 (fragment bytes per flop
// (192 + 64) / (192 * 64) = 0.0208); as 4 x 2 it gives 96 x 128 (0.0182, the figure of the 256x384 tile's 128 x 96) - the wide tiles differ by
// 4-5 % on shapes both fit, and that is their LDS read traffic (profiles/r03_gemm_mi16.md).  The phased schedule is unchanged: waves 4-7 run
// one barrier behind waves 0-3 (wave w and w + 4 share a SIMD).
// PATCH (round 4, vitg_patch_embed16): the patch-embedding GEMM reads its A operand - the im2col matrix [T * P, 3 p^2] - straight out of
// pixel_values [T, 3, R, R] (16-bit NCHW): no im2col kernel, no [T * P, 640] operand written and read back (671 MB at 512 x 448^2).  The K axis is
// re-laid as (c, ky) PAIRS of 16 slots (the weight is re-laid to match once at load, PackedWeights.patch_wf): a K-tile of 64 = 4 pairs, one
// 128-B LDS row = 4 x [kx 0..7 | kx p-8..p-1].  Each 16-B DMA chunk is then 8 CONSECUTIVE pixels of one image row - a legal (4-B aligned)
// global_load_lds_dwordx4 source - and the second chunk of a pair re-reads kx p-8..7, whose weight slots are zero: no chunk ever crosses the
// end of its pixel row, so nothing is read outside the tensor and no value depends on what lies behind a patch.  The per-lane source offset
// is (row's pixel base) + (pair's row/channel offset, recomputed per K-tile: ~7 VALU per DMA instruction on a GEMM of 11 K-tiles).
// SPLIT (round 6, vitg_gemm_split16: the operand-split precision tier, csrc/split.hip): the 16-bit-output epilogues write every value as a PAIR
// hi = fp16(v) at out[m, n] and lo = fp16(v - hi) at out[m, n + split_off] - the A operand of the next GEMM in K-concatenated form.  Two passes of
// the same stage loop (the activation is recomputed); the main loop is the same code.  Separate instantiations: the default kernels do not carry it.
template <typename T, int MODE, int BM, int BN, bool SK = false, int WM = 2, bool PATCH = false, bool SPLIT = false>
__global__ __launch_bounds__(G16_THREADS, 2) void gemm16_kernel(GemmArgs g_) {
    static_assert(!SK || MODE == MODE_F32, "stream-K partials are fp32 tiles");
    static_assert(!SPLIT || MODE == MODE_STORE16 || MODE == MODE_SWIGLU, "split pairs are a 16-bit-output form");
    static_assert(!PATCH || !SK, "the patch loader is not combined with stream-K");
    typedef typename T16<T>::vec8 vec8;
    GemmArgs g = g_;
    constexpr bool RES16 = MODE == MODE_RES16;
    constexpr int EMODE = RES16 ? MODE_STORE16 : MODE;   // epilogue form
    if constexpr (MODE == MODE_F32) {
        if (g.split_k > 1) {                              // split-K: block y takes its own K range into its own partial buffer (gemm256.hip)
            const int sp = blockIdx.y;
            const int nk_all = g.K / 64, per = nk_all / g.split_k, rem = nk_all % g.split_k;
            const int k0 = sp * per + min(sp, rem);
            g.A = (const char*)g.A + (size_t)k0 * 128;
            g.W = (const char*)g.W + (size_t)k0 * 128;
            g.K = (per + (sp < rem ? 1 : 0)) * 64;
            g.out = (float*)g.out + (size_t)sp * g.split_stride;
        }
    }
    constexpr int WN = 8 / WM;
    static_assert(WM == 2 || WM == 4, "wave grid");
    constexpr int MT = BM / WM / 16;                     // 16-row activation tiles per wave (8 or 12)
    constexpr int NT = BN / WN / 16;                     // 16-wide weight tiles per wave (4 or 6)
    // activation tiles per interval: 4 (16 MFMAs at NT = 4), except the SwiGLU GEMM on the 192-row wave tile, which runs intervals of 6 (24 MFMAs,
    // 8 barriers per K-tile instead of 12): gate/up +4.2 % (1272 -> 1326 TF/s), step +0.3 %; the same on fc1 / o-proj: -1 % / +-0 (tools/archive/r03_calls.sh.txt (ab_mc6))
    // the 96-row wave tile (WM = 4 on 384 rows: MT = 6, NT = 8) runs intervals of 3 activation tiles x 8 weight tiles = 24 MFMAs, 8 barriers per K-tile
    // the 160-row wave tile (320 x 256, round 5: M = 1600 = 5 x 320 rows of one context-parallel rank): MT = 10 as two chunks of 5 = 20 MFMAs, 8 barriers per K-tile
    constexpr int MC = MT == 10 ? 5 : MT == 6 ? 3 : (MODE == MODE_SWIGLU && MT == 12) ? 6 : 4;
    constexpr int NC = MT / MC;                          // row chunks per k-step
    constexpr int NI = 2 * NC;                           // intervals pairs per K-tile (2 k-steps of 32)
    constexpr int ROWB = 128;                            // bytes per LDS row: 64 k x 2 B
    constexpr int A_BYTES = BM * ROWB, W_BYTES = BN * ROWB, STAGE = A_BYTES + W_BYTES;
    constexpr int A_INSTR = A_BYTES / 1024 / 8, W_INSTR = W_BYTES / 1024 / 8, LPT = A_INSTR + W_INSTR;
    static_assert(MT % MC == 0, "row chunks");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int nwg_ = g.tiles_m * g.tiles_n;
    int sk_t0 = 0, sk_k0 = 0, sk_c0 = 0, sk_c1 = 0;      // stream-K: first tile, first K-tile in it, K-tiles in it, K-tiles in the next tile
    if constexpr (SK) {
        const long total = (long)nwg_ * g.sk_kt;
        const long u0 = (long)blockIdx.x * g.sk_units, u1 = min(u0 + (long)g.sk_units, total);
        if (u0 >= total) return;
        sk_t0 = (int)(u0 / g.sk_kt); sk_k0 = (int)(u0 % g.sk_kt);
        sk_c0 = (int)min((long)(g.sk_kt - sk_k0), u1 - u0);
        sk_c1 = (int)(u1 - u0) - sk_c0;
    }
    const int item_end = SK ? (sk_c1 > 0 ? 2 : 1) : nwg_, item_step = SK ? 1 : (int)gridDim.x;
    // (Requesting the first operand stage of the NEXT tile before this tile's epilogue - so that the wait at the next tile's start does not cover
    // the drain of the output stores and a DMA round trip back to back - was built and measured: 669.7-670.4 against 668.8-671.0 ms per step,
    // nothing, at 10-23 more spilled registers per tile.  Not kept; gpurun_out/r03_c21.)
    for (int vb = SK ? 0 : (int)blockIdx.x; vb < item_end; vb += item_step) {
    // per-lane quantities re-derived from an opaque copy of the thread id in every pass (hoisted they would stay live through the loop)
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WN, wc = wave % WN;
    int tm, tn;
    if constexpr (SK) { const int t = sk_t0 + vb; tm = t % g.tiles_m; tn = t / g.tiles_m; }
    else block_to_tile(vb, g.tiles_m, g.tiles_n, g.group_m, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;

    // ---- DMA assignment (as gemm256.hip): one instruction = 8 rows x 128 B; lane -> (row = lane / 8, LDS chunk slot = lane % 8);
    // source = wave-uniform tile base + per-lane 32-bit byte offset (saddr form), source chunk = slot ^ ((row >> 1) & 7)
    const int drow = lane >> 3, dslot = lane & 7;
    const int kfirst = SK ? (vb == 0 ? sk_k0 : 0) : 0;   // first K-tile of this work item
    const char* const a_base = (const char*)g.A + (size_t)m0 * g.lda * 2 + (size_t)kfirst * 128;
    const char* const w_base = (const char*)g.W + (size_t)n0 * g.ldw * 2 + (size_t)kfirst * 128;
    unsigned a_cur[A_INSTR], w_cur[W_INSTR];
    int a_pair[PATCH ? A_INSTR : 1];                     // PATCH: which of the K-tile's four (c, ky) pairs this lane's chunk belongs to
#pragma unroll
    for (int j = 0; j < A_INSTR; ++j) {
        const int R = (wave * A_INSTR + j) * 8 + drow;
        if constexpr (PATCH) {
            const int m = m0 + min(R, g.M - 1 - m0);
            const int t = m / g.pt_P, rem = m - t * g.pt_P, py = rem / g.pt_G, px = rem - py * g.pt_G;
            const int sc = dslot ^ ((R >> 1) & 7);       // source chunk 0..7 of the 128-B row: pair sc >> 1, half sc & 1
            a_cur[j] = (unsigned)((t * 3 * g.pt_R + py * g.pt_p) * g.pt_R + px * g.pt_p + (sc & 1) * (g.pt_p - 8)) * 2u;   // (elements < 2^31: host check)
            a_pair[j] = sc >> 1;
        } else {
            a_cur[j] = ((unsigned)(min(R, g.M - 1 - m0) * g.lda) + (unsigned)((dslot ^ ((R >> 1) & 7)) << 3)) * 2u;
        }
    }
#pragma unroll
    for (int j = 0; j < W_INSTR; ++j) {
        const int R = (wave * W_INSTR + j) * 8 + drow;
        w_cur[j] = ((unsigned)(min(R, g.N - 1 - n0) * g.ldw) + (unsigned)((dslot ^ ((R >> 1) & 7)) << 3)) * 2u;
    }
    auto issue_one = [&](int kt, int j) {
        char* st = smem + (kt & 1) * STAGE;
        const size_t koff = (size_t)kt * ROWB;
        const char* ka = a_base + koff;
        const char* kw = w_base + koff;
        unsigned off = j < A_INSTR ? a_cur[j] : w_cur[j - A_INSTR];
        if constexpr (PATCH) {
            if (j < A_INSTR) {                           // (pairs past 3 p - 1 are K padding: zero weight rows, any finite pixels)
                const int pair = min(kt * 4 + a_pair[j], g.pt_pairs - 1);
                const int c = (pair >= g.pt_p ? 1 : 0) + (pair >= 2 * g.pt_p ? 1 : 0), ky = pair - c * g.pt_p;
                off += (unsigned)((c * g.pt_R + ky) * g.pt_R * 2);
                ka = (const char*)g.A;
            }
        }
        asm volatile("" : "+v"(off));                    // keeps base + offset in the saddr form (no per-lane 64-bit pointer)
        if (j < A_INSTR)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ka + off),
                                             (__attribute__((address_space(3))) void*)(st + (wave * A_INSTR + j) * 1024), 16, 0, 0);
        else
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(kw + off),
                                             (__attribute__((address_space(3))) void*)(st + A_BYTES + (wave * W_INSTR + j - A_INSTR) * 1024), 16, 0, 0);
    };

    const int r16 = lane & 15, fq = lane >> 4;
    // Accumulators: zero, or - fp32 residual update without activation - the residual tile itself (the read half of the
    // read-modify-write then overlaps the main loop; these loads are older than every DMA in the vmcnt queue)
    // (MODE_RES16 = vitg_gemm_res16: the 16-bit residual stream of the tower, or the fp32 position table, into a 16-bit output)
    const bool acc_from_res = (MODE == MODE_F32 && g.res != nullptr && g.act == VITG_ACT_NONE) || RES16;
    // 16-bit output with a bias and no residual: the accumulators START at the bias (a lane's four features are the same for all its row tiles), so
    // the epilogue has no add - the initialisation costs the same moves as zeroing.  gemm_flags bit27 keeps the add in the epilogue (A/B)
    const bool bias_init = MODE == MODE_STORE16 && g.bias != nullptr && !(g.flags & 0x8000000);
    // (the 16-bit-residual form keeps its bias add in the epilogue: adding it to the preloaded residual measured 0.5 % SLOWER in the step - 670.7-671.9
    // against 667.2-668.3 ms, gpurun_out/r03_c20 - the preload is the load-heavy phase of those tiles)
    // (the bias quads are requested here - older than every DMA in the vmcnt queue - and moved into the accumulators behind the wait for the first
    // stage, so the first DMA round trip covers them)
    f32x4 acc[NT][MT];
    f32x4 bq[MODE == MODE_STORE16 ? NT : 1];
    if constexpr (MODE == MODE_STORE16) {
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const int nb = n0 + wc * (NT * 16) + i * 16 + fq * 4;
            bq[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (bias_init && nb < g.N) bq[i] = *(const f32x4*)(g.bias + nb);
        }
    } else {
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int j = 0; j < MT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // Interior tiles (all of BM x BN inside M x N - every tile but the last row / column of tiles) take straight-line forms of the residual preload
    // and of the output stages: no per-quad bounds predicate, no per-quad test of which residual pointers are set.  gemm_flags bit29 = off (A/B).
    const bool interior = !(g.flags & 0x20000000) && m0 + BM <= g.M && n0 + BN <= g.N;
    if constexpr (MODE == MODE_F32 || RES16) {
        if (acc_from_res) {
            if (RES16 && interior && g.res16 != nullptr && g.res == nullptr) {          // the tower's 16-bit stream
                const T* const rb = (const T*)g.res16 + (size_t)(m0 + wr * (MT * 16) + r16) * g.ldr16 + (n0 + wc * (NT * 16) + fq * 4);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const typename T16<T>::vec4 r = *(const typename T16<T>::vec4*)(rb + (size_t)(mt * 16) * g.ldr16 + nt * 16);
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[nt][mt][e] = T16<T>::to_f32(r[e]);
                    }
            } else if (MODE == MODE_F32 && interior && g.res_mod == 0) {                 // the decoder's fp32 stream (g.res is set: acc_from_res)
                const float* const rb = g.res + (size_t)(m0 + wr * (MT * 16) + r16) * g.ldr + (n0 + wc * (NT * 16) + fq * 4);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[nt][mt] = *(const f32x4*)(rb + (size_t)(mt * 16) * g.ldr + nt * 16);
            } else {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int m = m0 + wr * (MT * 16) + mt * 16 + r16;
                const int rm = g.res_mod > 0 ? m % g.res_mod : m;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int n = n0 + wc * (NT * 16) + nt * 16 + fq * 4;
                    if (m < g.M && n < g.N) {
                        if (g.res) acc[nt][mt] = *(const f32x4*)(g.res + (size_t)rm * g.ldr + n);
                        if constexpr (RES16) {
                            if (g.res16) {
                                const typename T16<T>::vec4 r = *(const typename T16<T>::vec4*)((const T*)g.res16 + (size_t)m * g.ldr16 + n);
#pragma unroll
                                for (int e = 0; e < 4; ++e) acc[nt][mt][e] += T16<T>::to_f32(r[e]);
                            }
                        }
                    }
                }
            }
            }
        }
    }

    // fragment byte offsets inside a stage: row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4), chunk = 4 ks + (lane >> 4); rows = base + 16 i + r16
    int a_off[2], w_off[2];
    {
        const int sw = (r16 >> 1) & 7;                   // the same for row + 16 i
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int ch = ((ks * 4 + fq) ^ sw) << 4;
            a_off[ks] = (wr * (MT * 16) + r16) * ROWB + ch;
            w_off[ks] = A_BYTES + (wc * (NT * 16) + r16) * ROWB + ch;
        }
    }

    const int nk = SK ? (vb == 0 ? sk_c0 : sk_c1) : g.K / 64;
#pragma unroll
    for (int j = 0; j < LPT; ++j) issue_one(0, j);

    vec8 fw[NT], fa[MC];
    const bool grpB = wave >= 4;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (MODE == MODE_STORE16) {
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int j = 0; j < MT; ++j) acc[i][j] = bq[i];
    }
    __builtin_amdgcn_s_barrier();
    if (grpB) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // Intervals (global barrier index i), group A: R_j in 2j, M_j in 2j+1; group B one later (j = NI kt + ph).  Stage (kt+1)&1 is last
    // read by B's R_{NI kt - 1} in interval 2 NI kt - 1 (B retires those reads - lgkmcnt(0) - before the barrier that ends it), so the
    // DMA of tile kt+1 may be issued from interval 2 NI kt on: spread over the first NI-1 read phases of tile kt.  It has to be
    // complete before A's R_{NI (kt+1)}: both groups drain vmcnt at the END of interval 2 NI (kt+1) - 1 (A after its last MFMA phase,
    // B after its last read phase).
#pragma unroll 1
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = kt + 1 < nk;
        const char* st_ = smem + (kt & 1) * STAGE;
#pragma unroll
        for (int ph = 0; ph < NI; ++ph) {
            const int ks = ph / NC, c = ph % NC;
            if (c == 0) {
#pragma unroll
                for (int i = 0; i < NT; ++i) fw[i] = *(const vec8*)(st_ + w_off[ks] + i * (16 * ROWB));
            }
#pragma unroll
            for (int i = 0; i < MC; ++i) fa[i] = *(const vec8*)(st_ + a_off[ks] + (c * MC + i) * (16 * ROWB));
            __builtin_amdgcn_sched_barrier(0);
            if (more && ph < NI - 1) {
#pragma unroll
                for (int j = 0; j < LPT; ++j)
                    if ((j * (NI - 1)) / LPT == ph) issue_one(kt + 1, j);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (ph == NI - 1 && grpB) {
                if (more) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // this stage is rewritten from the next interval on
            }
            __builtin_amdgcn_s_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int i = 0; i < MC; ++i) acc[nt][c * MC + i] = T16<T>::mfma16(fw[nt], fa[i], acc[nt][c * MC + i]);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            if (ph == NI - 1 && more && !grpB) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (!grpB) __builtin_amdgcn_s_barrier();             // pairs group B's extra barrier

    // ---- epilogue: 32 rows (two accumulator row-tiles) at a time through a wave-private LDS window -> full-row 16-B stores -------
    __builtin_amdgcn_s_barrier();                        // every wave is done reading the ring before it is reused
    const int mw = m0 + wr * (MT * 16), nw = n0 + wc * (NT * 16);
    if constexpr (EMODE == MODE_F32) {
        constexpr int CH = NT * 4;                        // 16-B chunks (4 floats) per row: 16 or 24
        constexpr int RB = NT * 64;
        constexpr int SWZ = (CH & -CH) - 1;               // 15 or 7
        auto swz = [](int row) { return RB % 256 == 0 ? (row & SWZ) : ((row >> 1) & SWZ); };
        char* const stage = smem + wave * (32 * RB);
        // (no activation in this epilogue: fp32 output with an activation runs on gemm256.hip's kernels, launch256_m)
        auto stages32 = [&](auto in_c, auto bias_c) {
        constexpr bool IN = decltype(in_c)::value;           // interior tile (or a stream-K partial tile): no bounds predicates, residual already in the accumulators
        constexpr bool BIAS = decltype(bias_c)::value;
#pragma unroll
        for (int mp = 0; mp < MT / 2; ++mp) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    f32x4 v = acc[nt][mp * 2 + h];
                    if constexpr (BIAS) {
                        const int nb = nw + nt * 16 + fq * 4;      // (the bias is re-read per quad from L1: holding NT quads of it over the epilogue spilled the wide tiles)
                        if (IN || nb < g.N) v += *(const f32x4*)(g.bias + nb);
                    }
                    const int row = h * 16 + r16, c = nt * 4 + fq;
                    *(f32x4*)(stage + row * RB + ((c ^ swz(row)) << 4)) = v;
                }
            // the same wave reads back what it wrote: LDS operations of one wave complete in order
            // stream-K: the segment's raw partial goes to its workspace slot as a whole BM x BN tile (rows past M hold finite filler nobody reads)
            const int om = SK ? wr * (MT * 16) : mw, on = SK ? wc * (NT * 16) : nw, oM = SK ? BM : g.M, oN = SK ? BN : g.N, oldo = SK ? BN : g.ldo;
            float* const obase = SK ? (float*)g.out + ((size_t)blockIdx.x * 2 + vb) * (size_t)(BM * BN) : (float*)g.out;
            char* const ob = (char*)(obase + (size_t)(om + mp * 32) * oldo + on);
#pragma unroll
            for (int i = 0; i < 32 * CH / 64; ++i) {
                const int idx = i * 64 + lane;
                const int row = idx / CH, c = idx % CH;
                f32x4 v = *(const f32x4*)(stage + row * RB + ((c ^ swz(row)) << 4));
                const int m = om + mp * 32 + row, n = on + c * 4;
                if (IN || (m < oM && n < oN)) {
                    if (!IN && !SK && g.res && !acc_from_res) {
                        const int rm = g.res_mod > 0 ? m % g.res_mod : m;
                        v += *(const f32x4*)(g.res + (size_t)rm * g.ldr + n);
                    }
                    *(f32x4*)(ob + (unsigned)(row * oldo + c * 4) * 4u) = v;
                }
            }
        }
        };   // stages32
        typedef std::true_type Y_;
        typedef std::false_type N_;
        const bool in32 = (SK || interior) && (SK || !g.res || acc_from_res);
        if (SK || g.bias == nullptr) { if (in32) stages32(Y_{}, N_{}); else stages32(N_{}, N_{}); }
        else { if (in32) stages32(Y_{}, Y_{}); else stages32(N_{}, Y_{}); }
    } else {
        constexpr int NTO = EMODE == MODE_SWIGLU ? NT / 2 : NT;        // output 16-column tiles per wave
        constexpr int CH = NTO * 2;                                   // 16-B chunks (8 halfs) per row: 4, 8 or 12
        constexpr bool POW2 = (CH & (CH - 1)) == 0;
        constexpr int RB = NTO * 32 + (POW2 ? 0 : 16);                // padded rows when the chunk count is not a power of two
        constexpr int SWZ = POW2 ? CH - 1 : 0;
        constexpr int SH = CH == 4 ? 2 : CH == 8 ? 1 : 0;             // XOR takes the row bits above those that already spread over 256 B
        char* const stage = smem + wave * (32 * RB);
        const int n_out = EMODE == MODE_SWIGLU ? nw / 2 : nw;
        const int n_lim = EMODE == MODE_SWIGLU ? g.N / 2 : g.N;
        // the activation is selected ONCE per tile (three copies of the stage loop), not per accumulator quad: a run-time test inside the quad loop
        // is a scalar branch per quad - 48-72 per tile - and cuts the stage into basic blocks the scheduler cannot overlap
        // 16-bit-residual form: the bias joins the accumulators in ONE pass here, behind the main loop (NT quad loads, then register adds), instead of a
        // 16-B load per accumulator quad inside the stages (the stores between the stages keep the compiler from reusing them): step 669.2 -> 666.7 ms
        // (+0.4 %, bit-identical; gpurun_out/r03_c42).  gemm_flags bit26 = per quad (A/B)
        bool res_bias_done = false;
        if constexpr (RES16) {
            if (g.bias != nullptr && !(g.flags & 0x4000000)) {   // (A/B: gemm_flags bit26 = per quad in the stages)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int nb = nw + nt * 16 + fq * 4;
                    f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
                    if (nb < g.N) b4 = *(const f32x4*)(g.bias + nb);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) acc[nt][mt] += b4;
                }
                res_bias_done = true;
            }
        }
        const bool has_bias = EMODE != MODE_SWIGLU && g.bias != nullptr && !(MODE == MODE_STORE16 && bias_init) && !res_bias_done;      // a bias is still to be added here
#ifndef VITG_SPLIT_ONEPASS
#define VITG_SPLIT_ONEPASS 1      // SPLIT: both planes from ONE pass over the accumulators (a second wave-private LDS window for lo); 0 = two passes (A/B:
                                  // the tier's step 1 504 -> 1 400 ms in three same-call alternations, bit-identical - gpurun_out/r06_c16/onepass_ab.txt)
#endif
        auto stages_hl = [&](auto act_c, auto in_c, auto bias_c, auto lo_c) {
        constexpr bool LO = decltype(lo_c)::value;           // SPLIT, two-pass form: the pass that writes lo = fp16(v - fp16(v)) to out + split_off
        constexpr bool BOTH = SPLIT && VITG_SPLIT_ONEPASS;   // SPLIT, one-pass form: hi to the wave's window, lo to its second window (behind all first windows)
        char* const stage_lo = stage + (G16_THREADS / 64) * (32 * RB);
        static_assert(!BOTH || 2 * (G16_THREADS / 64) * 32 * RB <= 2 * (BM + BN) * 128, "two staging windows per wave fit the operand ring");
        constexpr int ACT = decltype(act_c)::value;
        constexpr bool IN = decltype(in_c)::value;           // interior tile: no bounds predicates
        constexpr int BIAS = decltype(bias_c)::value;        // 1 / 0: a bias is / is not added here (interior forms); -1: tested per quad (edge tiles)
#pragma unroll
        for (int mp = 0; mp < MT / 2; ++mp) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int nt = 0; nt < NTO; ++nt) {
                    f32x4 v;
                    if constexpr (EMODE == MODE_SWIGLU) {
                        // weight rows are interleaved [32 gate | 32 up] (engine.pack_llm_layer): of every 64 features of the wave, tiles 0-1 gate, 2-3 up
                        const int gt = (nt >> 1) * 4 + (nt & 1);
                        const f32x4 ga = acc[gt][mp * 2 + h], up = acc[gt + 2][mp * 2 + h];
                        if constexpr (ACT == -1) {               // (A/B: the scalar form)
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = silu_f(ga[e]) * up[e];
                        } else {
                            const f32x2 lo = silu_mul_f2(f32x2{ga[0], ga[1]}, f32x2{up[0], up[1]}), hi = silu_mul_f2(f32x2{ga[2], ga[3]}, f32x2{up[2], up[3]});
                            v = f32x4{lo[0], lo[1], hi[0], hi[1]};
                        }
                    } else {
                        v = acc[nt][mp * 2 + h];
                        // (bias re-read per quad from L1 instead of NT quads held over the epilogue - held across the main loop OR across the stages they
                        // spill the wide tiles; the 16-bit-output form has it in its accumulators already)
                        const int nb = nw + nt * 16 + fq * 4;
                        if constexpr (BIAS == 1) v += *(const f32x4*)(g.bias + nb);
                        else if constexpr (BIAS == -1) { if (has_bias && nb < g.N) v += *(const f32x4*)(g.bias + nb); }
                        if constexpr (ACT == VITG_ACT_GELU_TANH) {
                            const f32x2 lo = gelu_tanh_f2(f32x2{v[0], v[1]}), hi = gelu_tanh_f2(f32x2{v[2], v[3]});
                            v = f32x4{lo[0], lo[1], hi[0], hi[1]};
                        } else if constexpr (ACT == VITG_ACT_GELU_ERF) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = gelu_erf_f(v[e]);
                        } else if constexpr (ACT == -1) {        // (A/B, gemm_flags bit28: the activation tested per quad, scalar forms - the round-2 epilogue)
                            if (g.act == VITG_ACT_GELU_TANH) {
#pragma unroll
                                for (int e = 0; e < 4; ++e) v[e] = gelu_tanh_f(v[e]);
                            } else if (g.act == VITG_ACT_GELU_ERF) {
#pragma unroll
                                for (int e = 0; e < 4; ++e) v[e] = gelu_erf_f(v[e]);
                            }
                        }
                    }
                    if constexpr (SPLIT) {
                        // pin each value as ONE fp32 number: the hi pass and the lo pass are separate copies of this code, and -ffp-contract=fast fuses the
                        // last multiply / add of the activation into the 16-bit conversion differently per use (v_fma_mix_f16: one rounding of the exact
                        // product, against cvt(fl32(product))): near a tie the hi this pass subtracts would not be the hi the other pass stored (csrc/split.hip)
#pragma unroll
                        for (int e = 0; e < 4; ++e) { float t_ = v[e]; asm volatile("" : "+v"(t_)); v[e] = t_; }
                    }
                    typename T16<T>::vec4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = T16<T>::from_f32(v[e]);
                    const int row = h * 16 + r16, c = nt * 2 + (fq >> 1);
                    if constexpr (BOTH) {
                        typename T16<T>::vec4 o2;
#pragma unroll
                        for (int e = 0; e < 4; ++e) o2[e] = T16<T>::from_f32(v[e] - T16<T>::to_f32(o[e]));
                        *(typename T16<T>::vec4*)(stage_lo + row * RB + ((c ^ ((row >> SH) & SWZ)) << 4) + (fq & 1) * 8) = o2;
                    } else if constexpr (LO) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = T16<T>::from_f32(v[e] - T16<T>::to_f32(o[e]));
                    }
                    *(typename T16<T>::vec4*)(stage + row * RB + ((c ^ ((row >> SH) & SWZ)) << 4) + (fq & 1) * 8) = o;
                }
#pragma unroll
            for (int plane = 0; plane < (BOTH ? 2 : 1); ++plane) {
            const char* const sp = plane ? stage_lo : stage;
            char* const ob = (char*)((T*)g.out + ((LO || plane) ? (size_t)g.split_off : (size_t)0) + (size_t)(mw + mp * 32) * g.ldo + n_out);
#pragma unroll
            for (int i = 0; i < 32 * CH / 64; ++i) {
                const int idx = i * 64 + lane;
                const int row = idx / CH, c = idx % CH;
                const u32x4 v = *(const u32x4*)(sp + row * RB + ((c ^ ((row >> SH) & SWZ)) << 4));
                const int m = mw + mp * 32 + row, n = n_out + c * 8;
                if (IN || m < g.M) {
                    const unsigned boff = (unsigned)(row * g.ldo + c * 8) * 2u;
                    if (IN || n + 8 <= n_lim) *(u32x4*)(ob + boff) = v;
                    else if (n + 4 <= n_lim) *(u32x2*)(ob + boff) = u32x2{v[0], v[1]};
                }
            }
            }
        }
        };   // stages_hl
        auto stages = [&](auto act_c, auto in_c, auto bias_c) {
            stages_hl(act_c, in_c, bias_c, std::false_type{});
            if constexpr (SPLIT && !VITG_SPLIT_ONEPASS) stages_hl(act_c, in_c, bias_c, std::true_type{});
        };
        typedef std::true_type IY;
        typedef std::false_type IN_;
        typedef std::integral_constant<int, 1> B1;
        typedef std::integral_constant<int, 0> B0;
        typedef std::integral_constant<int, -1> BR;
        typedef std::integral_constant<int, VITG_ACT_GELU_TANH> AT;
        typedef std::integral_constant<int, VITG_ACT_NONE> AN;
        if (g.flags & 0x10000000) stages(std::integral_constant<int, -1>{}, IN_{}, BR{});
        else if (EMODE != MODE_SWIGLU && g.act == VITG_ACT_GELU_TANH) {
            if (!interior) stages(AT{}, IN_{}, BR{});
            else if (g.flags & 0x40000000) stages(AT{}, IY{}, BR{});          // (A/B: bias tested per quad)
            else if (has_bias) stages(AT{}, IY{}, B1{});
            else stages(AT{}, IY{}, B0{});
        } else if (EMODE != MODE_SWIGLU && g.act == VITG_ACT_GELU_ERF) stages(std::integral_constant<int, VITG_ACT_GELU_ERF>{}, IN_{}, BR{});
        else if (!interior) stages(AN{}, IN_{}, BR{});
        else if (g.flags & 0x40000000) stages(AN{}, IY{}, BR{});              // (A/B: bias tested per quad)
        else if (EMODE != MODE_SWIGLU && has_bias) stages(AN{}, IY{}, B1{});
        else stages(AN{}, IY{}, B0{});
    }
    if (vb + item_step < item_end) {                     // another tile follows: every wave is done with its staging window
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }
    }   // tile loop
}

template <typename T, int MODE, int BM, int BN, int WM = 2, bool SPLIT = false>
static int launch16(GemmArgs g, hipStream_t s) {
    constexpr int lds = 2 * (BM + BN) * 128;
    static int done_dev[16] = {0};                       // per device: the attribute belongs to the device's copy of the kernel
    int cur_dev = 0;
    (void)hipGetDevice(&cur_dev);
    if (cur_dev >= 16 || !done_dev[cur_dev]) {
        (void)hipFuncSetAttribute((const void*)gemm16_kernel<T, MODE, BM, BN, false, WM, false, SPLIT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (cur_dev < 16) done_dev[cur_dev] = 1;
    }
    g.tiles_m = ceil_div(g.M, BM);
    g.tiles_n = ceil_div(g.N, BN);
    g.group_m = RESEARCH(gemm_group) > 0 ? RESEARCH(gemm_group) : 4;
    static int cus_dev[16] = {0};
    int cus = cur_dev < 16 ? cus_dev[cur_dev] : 0;
    if (!cus) {
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, cur_dev);
        if (cur_dev < 16) cus_dev[cur_dev] = cus;
    }
    const int nwg = g.tiles_m * g.tiles_n;
    // one block per CU walking the tile list when there are at least two tiles per CU (gemm256.hip: "software relaunch")
    const bool loop = nwg >= 2 * cus && !(RESEARCH(gemm_flags) & 0x1000000);
    const int splits = (MODE == MODE_F32 && g.split_k > 1) ? g.split_k : 1;
    hipLaunchKernelGGL((gemm16_kernel<T, MODE, BM, BN, false, WM, false, SPLIT>), dim3(loop && splits == 1 ? cus : nwg, splits), dim3(G16_THREADS), lds, s, g);
    return (int)hipGetLastError();
}

// the operand-split tier's 16-bit-output GEMMs (fp16 only): one tile shape per epilogue mode and shape class
int launch_gemm_mi16_split(const GemmArgs& g, int mode, hipStream_t s) {
    if (mode == MODE_SWIGLU) return g.M >= 384 ? launch16<f16_t, MODE_SWIGLU, 384, 256, 2, true>(g, s) : launch16<f16_t, MODE_SWIGLU, 256, 256, 2, true>(g, s);
    return g.N >= 384 ? launch16<f16_t, MODE_STORE16, 256, 384, 2, true>(g, s) : launch16<f16_t, MODE_STORE16, 256, 256, 2, true>(g, s);
}

#if VITG_PROFILING_BUILD   // stream-K lost its A/B against split-K x2 on the per-rank shapes (profiles/r03_context_parallel.md): profiling library only
template <typename T>
static int launch16_sk(GemmArgs g, int blocks, hipStream_t s) {
    constexpr int lds = 2 * (256 + 256) * 128;
    static std::atomic<unsigned> done_mask{0};
    const int cur_dev = current_device();
    if (!device_done(done_mask, cur_dev)) {
        (void)hipFuncSetAttribute((const void*)gemm16_kernel<T, MODE_F32, 256, 256, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        mark_device_done(done_mask, cur_dev);
    }
    g.tiles_m = ceil_div(g.M, 256);
    g.tiles_n = ceil_div(g.N, 256);
    g.group_m = 4;
    hipLaunchKernelGGL((gemm16_kernel<T, MODE_F32, 256, 256, true>), dim3(blocks), dim3(G16_THREADS), lds, s, g);
    return (int)hipGetLastError();
}

int launch_gemm_mi16_streamk(const GemmArgs& g, int dtype, int blocks, hipStream_t s) {
    return dtype == DT_BF16 ? launch16_sk<bf16_t>(g, blocks, s) : launch16_sk<f16_t>(g, blocks, s);
}

static int wm4_modes() {
    const int v = RESEARCH(gemm_wm4);
    return v < 0 ? 0 : v == 0 ? WM4_DEFAULT_MODES : v;
}
#endif

template <typename T, int MODE>
static int launch16_m(const GemmArgs& g, int cfg, hipStream_t s) {
    // 320 x 256 (round 5): the rows of one context-parallel rank (M = 1600 = 5 x 320; 1660 on the rank that carries the prompt) pad to 1920 on the
    // 384-row tile; gate / up (SwiGLU) and the split-K o / down projections (fp32 residual) take it - choose_cfg / splitk_tile_rows decide
    if constexpr (MODE == MODE_SWIGLU || MODE == MODE_F32) {
        if (cfg == 12 && g.M > 256) return launch16<T, MODE, 320, 256>(g, s);
    }
    if (cfg == 10 && g.M >= 384) {
        // gemm_wm4 (A/B, bit per epilogue mode: 1 16-bit output, 2 fp32 residual, 4 SwiGLU, 8 16-bit residual): the 4 x 2 wave grid (96 x 128 wave tiles)
#if VITG_PROFILING_BUILD   // (+3.7 % at long K, -1.7 % at K = 1152, nothing in the step: profiling library only)
        constexpr int bit = MODE == MODE_STORE16 ? 1 : MODE == MODE_F32 ? 2 : MODE == MODE_SWIGLU ? 4 : 8;
        if (wm4_modes() & bit) return launch16<T, MODE, 384, 256, 4>(g, s);
#endif
        return launch16<T, MODE, 384, 256>(g, s);
    }
    if constexpr (MODE != MODE_SWIGLU) {
        if (cfg == 9 && g.N >= 384) return launch16<T, MODE, 256, 384>(g, s);
    }
    return launch16<T, MODE, 256, 256>(g, s);
}

template <typename T>
static int launch16_t(const GemmArgs& g, int mode, int cfg, hipStream_t s) {
    if (mode == MODE_STORE16 && (g.res || g.res16)) return launch16_m<T, MODE_RES16>(g, cfg, s);
    if (mode == MODE_SWIGLU) return launch16_m<T, MODE_SWIGLU>(g, cfg, s);
    if (mode == MODE_F32) return launch16_m<T, MODE_F32>(g, cfg, s);
    return launch16_m<T, MODE_STORE16>(g, cfg, s);
}

// cfg: the tile shape gemm256.hip's choose_cfg picked (6 = 256x256, 9 = 256x384, 10 = 384x256, 12 = 320x256)
int launch_gemm_mi16(const GemmArgs& g, int dtype, int mode, int cfg, hipStream_t s) {
    return dtype == DT_BF16 ? launch16_t<bf16_t>(g, mode, cfg, s) : launch16_t<f16_t>(g, mode, cfg, s);
}

// patch embedding with the im2col inside the A-tile loader: the 256x384 tile with the 16-bit-output + fp32 position-table epilogue (MODE_RES16)
template <typename T>
static int launch16_patch(GemmArgs g, hipStream_t s) {
    constexpr int BM = 256, BN = 384, lds = 2 * (BM + BN) * 128;
    static std::atomic<unsigned> done_mask{0};
    const int cur_dev = current_device();
    if (!device_done(done_mask, cur_dev)) {
        (void)hipFuncSetAttribute((const void*)gemm16_kernel<T, MODE_RES16, BM, BN, false, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        mark_device_done(done_mask, cur_dev);
    }
    g.tiles_m = ceil_div(g.M, BM);
    g.tiles_n = ceil_div(g.N, BN);
    g.group_m = 4;
    const int cus = device_cu_count(cur_dev), nwg = g.tiles_m * g.tiles_n;
    hipLaunchKernelGGL((gemm16_kernel<T, MODE_RES16, BM, BN, false, 2, true>), dim3(nwg >= 2 * cus ? cus : nwg), dim3(G16_THREADS), lds, s, g);
    return (int)hipGetLastError();
}

}  // namespace vitg

// Operand-split precision tier (csrc/split.hip): out[m, n] = hi, out[m, n + lo_off] = lo of v = act(A W^T + bias) (mode STORE16) or of
// silu(gate) * up (mode SWIGLU, N = 2 x output width), v = hi + lo to 22 bits.  A / W are ordinary fp16 operands - a split A operand arrives
// K-concatenated ([hi | lo], K = 2 Kp) against W laid twice side by side, which this entry neither knows nor checks.  fp16 only.
extern "C" int vitg_gemm_split16(const void* A, int lda, const void* W, int ldw, int M, int N, int K, int mode, const float* bias, int act,
                                 void* out, int ldo, long lo_off, void* stream) {
    using namespace vitg;
    if (M <= 0 || N <= 0) return 0;
    if (!A || !W || !out) return VITG_ERR_ARG;
    if (K <= 0 || K % 64) return VITG_ERR_SHAPE;
    if (mode != VITG_GEMM_STORE16 && mode != VITG_GEMM_SWIGLU) return VITG_ERR_ARG;
    if (act != VITG_ACT_NONE && act != VITG_ACT_GELU_TANH && act != VITG_ACT_GELU_ERF) return VITG_ERR_ARG;
    if (mode == VITG_GEMM_SWIGLU && (N % 64 || act != VITG_ACT_NONE || bias)) return VITG_ERR_SHAPE;
    if ((lda % 8) || (ldw % 8) || (N % 8) || (ldo % 8) || (lo_off % 8)) return VITG_ERR_ALIGN;
    if (((uintptr_t)A & 15) || ((uintptr_t)W & 15) || ((uintptr_t)out & 15)) return VITG_ERR_ALIGN;
    const int n_out = mode == VITG_GEMM_SWIGLU ? N / 2 : N;
    if (lo_off < n_out || ldo < lo_off + n_out) return VITG_ERR_SHAPE;       // the two planes of a row must not overlap
    GemmArgs g = {};
    g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.M = M; g.N = N; g.K = K;
    g.bias = bias; g.act = act; g.out = out; g.ldo = ldo; g.split_k = 1; g.split_off = lo_off;
    const int rc = launch_gemm_mi16_split(g, mode, (hipStream_t)stream);
    return rc == 0 ? 0 : VITG_ERR_LAUNCH;
}

// x[T * P, N] (16-bit) = im2col(pixel_values) @ Wf^T + bias + pos[(m % P), :]   - SigLIP's patch_embedding + position table
// ([3P] SiglipVisionEmbeddings, reached from /root/reference/eagle/model/multimodal_encoder/clip_encoder.py:146) with the im2col done by the
// GEMM's own operand loader (PATCH above).  Wf: [N, ldw >= 64 ceil(3 p / 4)] in the PAIR layout - k = 16 (c p + ky) + slot, slot 0..7 = kx 0..7,
// slot 8..15 = kx p-8..p-1 with the slots of the re-read kx (p-8..7) ZERO, rows of K padding zero (videoitg_amd.engine.PackedWeights.patch_wf).
// Returns VITG_ERR_SHAPE for geometries the loader does not cover (the caller then runs vitg_patchify + vitg_gemm_res16).
extern "C" int vitg_patch_embed16(const void* pix, int T, int R, int patch, const void* Wf, int ldw, int N, int dtype, const float* bias,
                                  void* out, int ldo, const float* pos, int ldp, void* stream) {
    using namespace vitg;
    if (T <= 0) return 0;
    if (!pix || !Wf || !out) return VITG_ERR_ARG;
    if (dtype != DT_BF16 && dtype != DT_F16) return VITG_ERR_DTYPE;
    if (patch < 8 || patch > 16 || (patch & 1) || R < patch || (R & 1)) return VITG_ERR_SHAPE;     // 4-B aligned chunk sources need even p and R
    const int G = R / patch, P = G * G, pairs = 3 * patch, K = (pairs + 3) / 4 * 64;
    if (N < 256 || (long)T * 3 * R * R * 2 >= (1l << 32) || (long)T * P >= (1l << 31)) return VITG_ERR_SHAPE;
    if (ldw < K || (ldw % 8) || (ldo % 4) || (N % 4) || (pos && (ldp % 4))) return VITG_ERR_ALIGN;
    if (((uintptr_t)pix & 3) || ((uintptr_t)Wf & 15) || ((uintptr_t)out & 7)) return VITG_ERR_ALIGN;   // the loader's 16-B chunks start on 4-B pixel pairs
    if (ldo < N || (pos && ldp < N)) return VITG_ERR_SHAPE;                                            // rows of `out` / `pos` hold N values
    GemmArgs g = {};
    g.A = pix; g.lda = 0; g.W = Wf; g.ldw = ldw; g.M = T * P; g.N = N; g.K = K;
    g.bias = bias; g.act = VITG_ACT_NONE; g.out = out; g.ldo = ldo; g.res = pos; g.ldr = ldp; g.res_mod = pos ? P : 0;
    g.split_k = 1;
    g.pt_R = R; g.pt_G = G; g.pt_P = P; g.pt_p = patch; g.pt_pairs = pairs;
    const int rc = dtype == DT_BF16 ? launch16_patch<bf16_t>(g, (hipStream_t)stream) : launch16_patch<f16_t>(g, (hipStream_t)stream);
    return rc == 0 ? 0 : VITG_ERR_LAUNCH;
}

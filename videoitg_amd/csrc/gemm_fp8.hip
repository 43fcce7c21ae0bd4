// FP8 (OCP e4m3fn) weight path:  C[M,N] = diag(sa) * (A8[M,K] * W8[N,K]^T) * diag(sw)  with the fused epilogues of gemm256.hip.
//
//   A8: activations quantised PER TOKEN   (row m:  a = A8[m,:] * sa[m],  sa[m] = amax_m / 448), produced by quant_rows_fp8 /
//       the fused norm kernels below;
//   W8: weights quantised PER OUTPUT CHANNEL at load time (row n:  w = W8[n,:] * sw[n]).
// The matrix pipe runs v_mfma_scale_f32_32x32x64_f8f6f4 (the only fp8 MFMA of gfx950 that runs at twice the 16-bit rate:
// 64 deep, 16 passes; the carried-forward 32x32x16_fp8_fp8 runs at the bf16 rate) with both block scales pinned to 2^0 - the
// real scales are per row / per column of the OUTPUT, so they factor out of the K sum and are applied once to the fp32
// accumulators in the epilogue.
//
// Structure = the phased 256xBN kernel of gemm256.hip measured in BYTES: an LDS row is 128 B (= 128 fp8 of K), a k-step is
// 64 B of it (one MFMA, 32 B per lane = two ds_read_b128), so per K-tile there are 2 k-steps of 8 MFMAs x 64 cycles where the
// 16-bit kernel has 4 k-steps of 8 MFMAs x 32 cycles: same LDS bytes per MFMA cycle, same DMA bytes per MFMA cycle, same
// barrier cadence - and twice the flops.  Operands by LDS-DMA (saddr form), 2-stage ring, waves 4-7 one barrier behind waves 0-3
// so that one wave per SIMD owns the matrix pipe while its sibling reads fragments / issues DMA.
//
// Fragment layout of the f8f6f4 MFMA (32x32x64, fp8): lane l = (row l & 31, h = l >> 5) holds K bytes [16 h, +16) and [32 + 16 h, +16)
// of the 64-deep step in its registers 0-3 / 4-7, for A and B alike (see the fragment offsets below).
// C/D layout is the shape's (same as 32x32x16): lane (token = l & 31, h = l >> 5) holds features 8 (r >> 2) + 4 h + (r & 3).
#include "gemm_common.h"
#include <type_traits>

namespace vitg {

struct Gemm8Args {
    GemmArgs g;               // A / W are fp8 bytes; lda / ldw in BYTES (multiples of 16); K a multiple of 128
    const float* sa;          // [M] per-token activation scales
    const float* sw;          // [N] per-output-channel weight scales (row order of W8, i.e. interleaved for SwiGLU)
    // MX block scales (one E8M0 byte per 32 consecutive K elements of a row; the scaled MFMA consumes them directly):
    // layout [K/128][ld][4] bytes: the four block scales of one 128-deep K-tile of row m sit in ONE dword at ((kt * ld + m) * 4), so the 32
    // token rows of a fragment read one 128-B line per K-tile (a row-major [M][K/32] array cost 32 lines per load instruction)
    const unsigned char* a_mx; int ld_amx;   // MXA kernels: A8 is block-scaled; ld_amx >= M rows per K-tile plane; sa unused
    unsigned char* o_mx; int ld_omx;         // MXO kernels: the output is e4m3 bytes [M][ldo] + block scales in the same layout (the next GEMM's A8 / a_mx)
};

typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x4v __attribute__((ext_vector_type(4)));

constexpr int F8_THREADS = 512, F8_ROWB = 128, F8_KS = 2, F8_STAGES = 2;

__device__ __forceinline__ unsigned pack4_fp8(float a, float b, float c, float d);      // (defined with the row quantisers below)

// MXA: the activation operand carries MX block scales (ga.a_mx) instead of a per-token scale.  MXO (MODE_STORE16 only): the output
// leaves as e4m3 bytes + MX block scales computed in the epilogue - what the producer of the NEXT GEMM's operand would otherwise write
// as 16 bits and a stand-alone quantiser pass would read back (fc1 -> fc2: 1.8 ms per tower layer).
template <typename T, int MODE, int BM, int BN, bool MXA = false, bool MXO = false>
__global__ __launch_bounds__(F8_THREADS, 2) void gemm_fp8_kernel(Gemm8Args ga) {
    static_assert(!MXO || MODE != MODE_F32, "MX output exists for the 16-bit-output modes (plain and SwiGLU)");
    const GemmArgs& g = ga.g;
    constexpr int WM = 2, WN = 4;
    constexpr int MT = BM / WM / 32, NT = BN / WN / 32;
    constexpr int CPR = F8_ROWB / 16;                   // 8 chunks of 16 B per LDS row
    constexpr int RPI = 1024 / F8_ROWB;                 // 8 rows per 1-KiB DMA instruction
    constexpr int A_BYTES = BM * F8_ROWB, W_BYTES = BN * F8_ROWB, STAGE = A_BYTES + W_BYTES;
    constexpr int A_INSTR = A_BYTES / 1024 / 8, W_INSTR = W_BYTES / 1024 / 8, LPT = A_INSTR + W_INSTR;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // one block per CU walking the tile list when the launcher found at least two tiles per CU (round 4; as gemm_mi16.hip): gridDim.x < tiles
    const int nwg_ = g.tiles_m * g.tiles_n;
    for (int vb = (int)blockIdx.x; vb < nwg_; vb += (int)gridDim.x) {
    // per-lane quantities re-derived from an opaque copy of the thread id in every pass (hoisted they would stay live through the loop)
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WN, wc = wave % WN;
    int tm, tn;
    block_to_tile(vb, g.tiles_m, g.tiles_n, g.group_m, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;

    // ---- LDS-DMA: one instruction = 8 rows x 128 B; lane -> (row = lane / 8, slot = lane % 8); XOR swizzle on the SOURCE chunk
    auto swz = [](int row) { return (row >> 1) & 7; };
    const int drow = lane / CPR, dslot = lane % CPR;
    const char* const a_base = (const char*)g.A + (size_t)m0 * g.lda;
    const char* const w_base = (const char*)g.W + (size_t)n0 * g.ldw;
    unsigned a_cur[A_INSTR], w_cur[W_INSTR];
#pragma unroll
    for (int j = 0; j < A_INSTR; ++j) {
        const int R = (wave * A_INSTR + j) * RPI + drow;
        a_cur[j] = (unsigned)(min(R, g.M - 1 - m0) * g.lda) + (unsigned)((dslot ^ swz(R)) << 4);
    }
#pragma unroll
    for (int j = 0; j < W_INSTR; ++j) {
        const int R = (wave * W_INSTR + j) * RPI + drow;
        w_cur[j] = (unsigned)(min(R, g.N - 1 - n0) * g.ldw) + (unsigned)((dslot ^ swz(R)) << 4);
    }
    auto issue_one = [&](int kt, int j) {
        char* st = smem + (kt & 1) * STAGE;
        const size_t koff = (size_t)kt * F8_ROWB;
        const char* ka = a_base + koff;
        const char* kw = w_base + koff;
        unsigned off = j < A_INSTR ? a_cur[j] : w_cur[j - A_INSTR];
        asm volatile("" : "+v"(off));                  // keeps the saddr + 32-bit voffset form (see gemm256.hip)
        if (j < A_INSTR)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ka + off),
                                             (__attribute__((address_space(3))) void*)(st + (wave * A_INSTR + j) * 1024), 16, 0, 0);
        else
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(kw + off),
                                             (__attribute__((address_space(3))) void*)(st + A_BYTES + (wave * W_INSTR + j - A_INSTR) * 1024), 16, 0, 0);
    };

    f32x16 acc[NT][MT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment byte offsets: row*128 + ((chunk ^ swz(row)) << 4).  Hardware K order of the 64-deep fp8 step (probed with one-hot
    // operands and block scales, tools/ubench/mfma_mx_kmap.hip): registers 0-3 of lane (row, h) are k = 16 h .. +15 of the FIRST
    // 32-block, registers 4-7 are k = 32 + 16 h .. +15 of the SECOND, and the block scale of the first / second block is taken from
    // lanes 0-31 / 32-63.  So the two 16-B reads of a lane are chunks 4 ks + h and 4 ks + 2 + h of the row.  (Without block scales any
    // K order that is the same for both operands would do - round 2's first version read 32 contiguous bytes per lane.)
    const int frow = lane & 31, fh = lane >> 5;
    int a_off[F8_KS][2], w_off[F8_KS][2];
    {
        const int sw = swz(frow);                       // invariant under row += 32
#pragma unroll
        for (int ks = 0; ks < F8_KS; ++ks)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int ch = ((ks * 4 + u * 2 + fh) ^ sw) << 4;
                a_off[ks][u] = (wr * (MT * 32) + frow) * F8_ROWB + ch;
                w_off[ks][u] = A_BYTES + (wc * (NT * 32) + frow) * F8_ROWB + ch;
            }
    }
    const int nk = g.K / F8_ROWB;
#pragma unroll
    for (int j = 0; j < LPT; ++j) issue_one(0, j);

    i32x8 fa[MT], fw[NT];
    const int one = 0x7F7F7F7F;                         // E8M0 127 = 2^0 in every byte: block scales off
    // MXA: per K-tile (128 = 4 blocks of 32) one dword of scale bytes per token row; lane (row, h) needs bytes 2 pp + h of it for k-step
    // pp -> shifted right by 8 h once, then opsel 0 / 2 picks the byte.  The dword of the NEXT K-tile is fetched with its DMA.
    unsigned sc_off[MXA ? MT : 1];
    int sc_cur[MXA ? MT : 1], sc_nxt[MXA ? MT : 1];
    const unsigned char* const sc_base = MXA ? ga.a_mx + (size_t)m0 * 4 : nullptr;
    const size_t sc_plane = MXA ? (size_t)ga.ld_amx * 4 : 0;        // bytes per K-tile plane
    if constexpr (MXA) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            sc_off[mt] = (unsigned)(min(wr * (MT * 32) + mt * 32 + frow, g.M - 1 - m0) * 4);
            sc_cur[mt] = (int)(*(const unsigned*)(sc_base + sc_off[mt]) >> (8 * fh));
        }
    }
    const bool grpB = wr == 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (grpB) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = kt + 1 < nk;
#pragma unroll
        for (int pp = 0; pp < F8_KS; ++pp) {
            const char* st_ = smem + (kt & 1) * STAGE;
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                const i32x4v lo = *(const i32x4v*)(st_ + w_off[pp][0] + i * (32 * F8_ROWB));
                const i32x4v hi = *(const i32x4v*)(st_ + w_off[pp][1] + i * (32 * F8_ROWB));
                fw[i] = i32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const i32x4v lo = *(const i32x4v*)(st_ + a_off[pp][0] + i * (32 * F8_ROWB));
                const i32x4v hi = *(const i32x4v*)(st_ + a_off[pp][1] + i * (32 * F8_ROWB));
                fa[i] = i32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
            __builtin_amdgcn_sched_barrier(0);
            if (more && pp == 0) {
                // the next K-tile goes into the other stage (free since the sibling group's last read, one interval ago): ALL pieces in
                // the FIRST read phase.  A K-tile is only two intervals here, and the pieces must have landed at the end of the second:
                // issuing half of them in the second read phase made group B wait for its DMA right after issuing it (matrix pipe
                // 47 % busy on 8192^3 at 1.93 GHz, i.e. not even power-limited; tools/profile_fp8_gemm.sh)
#pragma unroll
                for (int j = 0; j < LPT; ++j) issue_one(kt + 1, j);
                if constexpr (MXA) {
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {          // raw dwords: any use here would make the compiler wait for the DMA pieces issued above as well
                        unsigned off = sc_off[mt];
                        asm volatile("" : "+v"(off));               // saddr form: uniform base + K-tile advance in SGPRs, 32-bit lane offset
                        sc_nxt[mt] = (int)*(const unsigned*)(sc_base + (size_t)(kt + 1) * sc_plane + off);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (pp == F8_KS - 1 && more && grpB) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            // The MFMA intrinsic is a pure function of registers: without the two fences below LLVM sinks all 16 MFMAs of a K-tile past
            // the barriers to the bottom of the loop (seen in the ISA: `s_setprio 1; s_setprio 0` with nothing between them), both wave
            // groups then fire their MFMAs at the same time and nobody reads fragments meanwhile - matrix pipe 47 % busy at 1.93 GHz.
            // An empty asm with the accumulators as in/out operands pins "computed before here" / "not started before here".
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) asm volatile("" : "+v"(acc[nt][mt]));
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    if constexpr (!MXA) acc[nt][mt] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fw[nt], fa[mt], acc[nt][mt], 0, 0, 0, one, 0, one);
                    else if (pp == 0) acc[nt][mt] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fw[nt], fa[mt], acc[nt][mt], 0, 0, 0, one, 0, sc_cur[mt]);
                    else acc[nt][mt] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fw[nt], fa[mt], acc[nt][mt], 0, 0, 0, one, 2, sc_cur[mt]);
                }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) asm volatile("" : "+v"(acc[nt][mt]));
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            if (pp == F8_KS - 1 && more && !grpB) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (MXA) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                // (landed: both groups passed their vmcnt(0) of this K-tile.)  The opaque copy pins the shift HERE: left alone, LLVM sinks it
                // into the next iteration behind that iteration's DMA issue and guards it with vmcnt(0) - a full DMA round trip per K-tile
                sc_cur[mt] = (int)((unsigned)sc_nxt[mt] >> (8 * fh));
                asm volatile("" : "+v"(sc_cur[mt]));
            }
        }
    }
    if (!grpB) __builtin_amdgcn_s_barrier();            // pairs group B's extra barrier

    // ---- dequantise: acc[nt][mt][r] *= sa[token] * sw[feature] ------------------------------------------------------------
    {
        float sam[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) sam[mt] = MXA ? 1.0f : ga.sa[min(m0 + wr * (MT * 32) + mt * 32 + frow, g.M - 1)];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const int n = n0 + wc * (NT * 32) + nt * 32 + q4 * 8 + fh * 4;
                f32x4 s = {0.f, 0.f, 0.f, 0.f};
                if (n < g.N) s = *(const f32x4*)(ga.sw + n);          // N % 4 == 0 (checked by the host)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[nt][mt][q4 * 4 + e] *= sam[mt] * s[e];
            }
    }
    // ---- epilogue: the staged full-row stores of gemm_common.h (bias, activation, SwiGLU, fp32 residual) ----------------------
    __builtin_amdgcn_s_barrier();                        // every wave is done reading the ring before it is reused
    if constexpr (MODE == MODE_F32) {
        constexpr int SB = 32 * NT * 128;
        static_assert(8 * SB <= F8_STAGES * STAGE, "fp32 staging must fit in the ring");
        if (g.res16) epilogue_staged<T, MODE_F32, MT, NT, 0, MT, false, true>(g, acc, smem + wave * SB, m0 + wr * (MT * 32), n0 + wc * (NT * 32), lane);   // 16-bit residual stream
        else epilogue_staged<T, MODE_F32, MT, NT>(g, acc, smem + wave * SB, m0 + wr * (MT * 32), n0 + wc * (NT * 32), lane);
    } else if constexpr (MXO) {
        // ---- MX output: bias + activation (or SwiGLU) in fp32, then per (token row, 32-column output block = one MFMA tile): amax over the
        // lane's 16 values and its lane ^ 32 partner's 16, E8M0 exponent e = ceil(log2(amax / 448)) (no saturation), q = e4m3(v * 2^-e);
        // bytes staged through wave-private LDS into 16-B row stores, one scale byte per block.  Columns >= N_out are written as zeros
        // (they are the K padding of the next GEMM).
        static_assert(NT == 2, "MX epilogue is written for the 128 x 64 wave tile");
        constexpr int NTO = MODE == MODE_SWIGLU ? 1 : NT;   // 32-column output blocks per wave
        constexpr int RB = NTO * 32;                        // staged bytes per row
        char* stg = smem + wave * (MT * 32 * RB);
        const int m_base = m0 + wr * (MT * 32);
        const int n_base = MODE == MODE_SWIGLU ? (n0 + wc * (NT * 32)) / 2 : n0 + wc * (NT * 32);     // first OUTPUT column of this wave
        const int n_lim = MODE == MODE_SWIGLU ? g.N / 2 : g.N;
        unsigned char* const ob = (unsigned char*)g.out;
        auto mx_rows = [&](auto act_c) {
        constexpr int ACT = decltype(act_c)::value;           // chosen once per tile, not per accumulator quad (gemm_common.h: body32)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            unsigned sc2 = 0;
#pragma unroll
            for (int nt = 0; nt < NTO; ++nt) {
                float v[16];
                float amax = 0.f;
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const int n = n_base + nt * 32 + q4 * 8 + fh * 4;
                    f32x4 t;
                    if constexpr (MODE == MODE_SWIGLU) {
                        const f32x2 lo = silu_mul_f2(f32x2{acc[0][mt][q4 * 4], acc[0][mt][q4 * 4 + 1]}, f32x2{acc[NT - 1][mt][q4 * 4], acc[NT - 1][mt][q4 * 4 + 1]});
                        const f32x2 hi = silu_mul_f2(f32x2{acc[0][mt][q4 * 4 + 2], acc[0][mt][q4 * 4 + 3]}, f32x2{acc[NT - 1][mt][q4 * 4 + 2], acc[NT - 1][mt][q4 * 4 + 3]});
                        t = f32x4{lo[0], lo[1], hi[0], hi[1]};
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) t[e] = acc[nt][mt][q4 * 4 + e];
                        if (g.bias && n < g.N) t += *(const f32x4*)(g.bias + n);
                        if constexpr (ACT == VITG_ACT_GELU_TANH) {
                            const f32x2 lo = gelu_tanh_f2(f32x2{t[0], t[1]}), hi = gelu_tanh_f2(f32x2{t[2], t[3]});
                            t = f32x4{lo[0], lo[1], hi[0], hi[1]};
                        } else if constexpr (ACT == VITG_ACT_GELU_ERF) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) t[e] = gelu_erf_f(t[e]);
                        }
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float x = n < n_lim ? t[e] : 0.f;
                        v[q4 * 4 + e] = x;
                        amax = fmaxf(amax, fabsf(x));
                    }
                }
                amax = fmaxf(amax, __shfl_xor(amax, 32, 64));
                const unsigned bits = __float_as_uint(amax * (1.0f / 448.0f));
                int ex = (int)((bits >> 23) & 255u) - 127 + ((bits & 0x7fffffu) ? 1 : 0);
                ex = min(max(ex, -126), 127);                // amax = 0 (or denormal): any scale will do, the bytes are zero; amax = inf / NaN (exponent 255):
                                                             // clamped so that the shift below stays defined - inv becomes 0 and inf * 0 = NaN / NaN * 0 = NaN
                                                             // reach the e4m3 bytes as NaN instead of a garbage scale silently rescaling the block
                const float inv = __uint_as_float((unsigned)(127 - ex) << 23);
                sc2 |= (unsigned)(ex + 127) << (8 * nt);
                const int row = mt * 32 + frow;
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4)
                    *(unsigned*)(stg + row * RB + nt * 32 + q4 * 8 + fh * 4) =
                        pack4_fp8(v[q4 * 4] * inv, v[q4 * 4 + 1] * inv, v[q4 * 4 + 2] * inv, v[q4 * 4 + 3] * inv);
            }
            const int m = m_base + mt * 32 + frow;
            if (fh == 0 && m < g.M && n_base < g.ldo) {          // this wave's NTO blocks are a quarter / half of a K-tile dword of the consumer
                unsigned char* sp = ga.o_mx + ((size_t)(n_base / 128) * ga.ld_omx + m) * 4 + (n_base / 32) % 4;
                if constexpr (NTO == 2) *(unsigned short*)sp = (unsigned short)sc2;
                else *sp = (unsigned char)sc2;
            }
        }
        };   // mx_rows
        if (MODE != MODE_SWIGLU && g.act == VITG_ACT_GELU_TANH) mx_rows(std::integral_constant<int, VITG_ACT_GELU_TANH>{});
        else if (MODE != MODE_SWIGLU && g.act == VITG_ACT_GELU_ERF) mx_rows(std::integral_constant<int, VITG_ACT_GELU_ERF>{});
        else mx_rows(std::integral_constant<int, VITG_ACT_NONE>{});
        // the same wave reads back what it wrote (LDS ops of one wave complete in order): 16-B chunks
#pragma unroll
        for (int i = 0; i < MT * 32 * (RB / 16) / 64; ++i) {
            const int idx = i * 64 + lane;
            const int row = idx / (RB / 16), c = idx % (RB / 16);
            const u32x4 q = *(const u32x4*)(stg + row * RB + c * 16);
            const int m = m_base + row, n = n_base + c * 16;
            if (m < g.M && n + 16 <= g.ldo) *(u32x4*)(ob + (size_t)m * g.ldo + n) = q;
        }
    } else {
        constexpr int STAGE_BYTES = MT * 32 * (MODE == MODE_SWIGLU ? 1 : NT) * 64;
        static_assert(8 * STAGE_BYTES <= F8_STAGES * STAGE, "16-bit staging must fit in the ring");
        epilogue_staged<T, MODE, MT, NT>(g, acc, smem + wave * STAGE_BYTES, m0 + wr * (MT * 32), n0 + wc * (NT * 32), lane);
    }
    // the next tile's first DMA stage overwrites the staging windows: every wave's LDS reads of this tile have to be retired first
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    }   // tile loop
}

template <typename T, int MODE, int BM, int BN, bool MXA = false, bool MXO = false>
static int launch_fp8(Gemm8Args ga, hipStream_t s) {
    constexpr int lds = F8_STAGES * (BM + BN) * F8_ROWB;
    static std::atomic<unsigned> done_mask{0};   // per device: the attribute belongs to the device's copy of the kernel
    const int cur_dev = current_device();
    if (!device_done(done_mask, cur_dev)) {
        (void)hipFuncSetAttribute((const void*)gemm_fp8_kernel<T, MODE, BM, BN, MXA, MXO>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        mark_device_done(done_mask, cur_dev);
    }
    ga.g.tiles_m = ceil_div(ga.g.M, BM);
    ga.g.tiles_n = ceil_div(ga.g.N, BN);
    ga.g.group_m = RESEARCH(gemm_group) > 0 ? RESEARCH(gemm_group) : 4;
    const int nwg = ga.g.tiles_m * ga.g.tiles_n, cus = device_cu_count(cur_dev);
    // one block per CU walking the tile list when there are at least two tiles per CU (as gemm_mi16.hip; round 4: fc1 +5 %, qkv +2-4 %, the fp8 job
    // +0.5 % against one hardware block per tile, bit-identical; profiles/r04_experiments.md section 7)
    const bool loop = nwg >= 2 * cus;
    hipLaunchKernelGGL((gemm_fp8_kernel<T, MODE, BM, BN, MXA, MXO>), dim3(loop ? cus : nwg), dim3(F8_THREADS), lds, s, ga);
    return (int)hipGetLastError();
}

template <typename T>
static int launch_fp8_t(const Gemm8Args& ga, int mode, hipStream_t s) {
    // MX forms: the two the tower's MLP needs - a block-scaled activation operand into the fp32-residual GEMM (fc2) and the plain
    // 16-bit-output GEMM emitting block-scaled e4m3 instead (fc1); vitg_gemm_fp8_mx refuses the other combinations
    if (ga.a_mx) return launch_fp8<T, MODE_F32, 256, 256, true, false>(ga, s);
    if (ga.o_mx) return mode == MODE_SWIGLU ? launch_fp8<T, MODE_SWIGLU, 256, 256, false, true>(ga, s) : launch_fp8<T, MODE_STORE16, 256, 256, false, true>(ga, s);
    if (mode == MODE_SWIGLU) return launch_fp8<T, MODE_SWIGLU, 256, 256>(ga, s);
    if (mode == MODE_F32) return launch_fp8<T, MODE_F32, 256, 256>(ga, s);
    return launch_fp8<T, MODE_STORE16, 256, 256>(ga, s);
}

// ---- row quantisers ---------------------------------------------------------------------------------------------------------
// e4m3fn: max finite 448.  scale = amax / 448 (1 for an all-zero row), q = RNE(x / scale) via v_cvt_pk_fp8_f32 (OCP on gfx950).
__device__ __forceinline__ unsigned pack4_fp8(float a, float b, float c, float d) {
    a = fminf(fmaxf(a, -448.f), 448.f); b = fminf(fmaxf(b, -448.f), 448.f);
    c = fminf(fmaxf(c, -448.f), 448.f); d = fminf(fmaxf(d, -448.f), 448.f);
    int v = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
    v = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, v, true);
    return (unsigned)v;
}

template <typename TI> __device__ __forceinline__ f32x4 load4(const TI* p);
template <> __device__ __forceinline__ f32x4 load4<float>(const float* p) { return *(const f32x4*)p; }
template <> __device__ __forceinline__ f32x4 load4<bf16_t>(const bf16_t* p) {
    const bf16x4 v = *(const bf16x4*)p;
    return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}
template <> __device__ __forceinline__ f32x4 load4<f16_t>(const f16_t* p) {
    const f16x4 v = *(const f16x4*)p;
    return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}

__device__ __forceinline__ float block_max_256(float v, float* red) {
    v = wave_max(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) red[wave] = v;
    __syncthreads();
    v = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    return v;
}

// one block (256 threads) per row: pass 1 row amax, pass 2 re-reads the (cache-hot) row and writes fp8; pad columns [D, ldo) zeroed
template <typename TI>
__global__ __launch_bounds__(256) void quant_rows_kernel(const TI* __restrict__ x, long ldx, unsigned char* __restrict__ out, long ldo,
                                                         float* __restrict__ scale, int D) {
    __shared__ float red[4];
    const long row = blockIdx.x;
    const TI* xr = x + row * ldx;
    float amax = 0.f;
    for (int c = threadIdx.x * 4; c < D; c += 1024) {
        const f32x4 v = load4<TI>(xr + c);
        amax = fmaxf(amax, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
    }
    amax = block_max_256(amax, red);
    const float sc = amax > 0.f ? amax * (1.0f / 448.0f) : 1.0f;
    const float inv = 1.0f / sc;
    if (threadIdx.x == 0) scale[row] = sc;
    unsigned* orow = (unsigned*)(out + row * ldo);
    for (int c = threadIdx.x * 4; c < ldo; c += 1024) {
        unsigned q = 0;
        if (c < D) {
            const f32x4 v = load4<TI>(xr + c);
            q = pack4_fp8(v[0] * inv, v[1] * inv, v[2] * inv, v[3] * inv);
        }
        orow[c >> 2] = q;
    }
}

// rows of up to 8192 values: ONE WAVE per row, the row stays in registers (single read, no LDS, no block barrier); 4 rows per block
template <typename TI, int VPL>       // VPL = f32x4 groups per lane: D <= 256 * VPL
__global__ __launch_bounds__(256) void quant_rows_wave_kernel(const TI* __restrict__ x, long ldx, unsigned char* __restrict__ out, long ldo,
                                                              float* __restrict__ scale, long M, int D) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const TI* xr = x + row * ldx;
    f32x4 v[VPL];
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c = (i * 64 + lane) * 4;
        v[i] = c < D ? load4<TI>(xr + c) : f32x4{0.f, 0.f, 0.f, 0.f};
        amax = fmaxf(amax, fmaxf(fmaxf(fabsf(v[i][0]), fabsf(v[i][1])), fmaxf(fabsf(v[i][2]), fabsf(v[i][3]))));
    }
    amax = wave_max(amax);
    const float sc = amax > 0.f ? amax * (1.0f / 448.0f) : 1.0f;
    const float inv = 1.0f / sc;
    if (lane == 0) scale[row] = sc;
    unsigned* orow = (unsigned*)(out + row * ldo);
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < ldo) orow[c >> 2] = c < D ? pack4_fp8(v[i][0] * inv, v[i][1] * inv, v[i][2] * inv, v[i][3] * inv) : 0u;
    }
}

// LayerNorm / RMSNorm fused with the per-token quantiser: y = norm(x) * w (+ b) in fp32, then amax / fp8 as above.  One WAVE per
// row (like norm_kernel in elementwise.hip), the row in registers: a single HBM read, no LDS, no block barrier; D <= 256 * VPL.
template <bool RMS, int VPL, typename TI = float>
__global__ __launch_bounds__(256) void norm_quant_kernel(const TI* __restrict__ x, long ldx, const float* __restrict__ w,
                                                         const float* __restrict__ b, unsigned char* __restrict__ out, long ldo,
                                                         float* __restrict__ scale, long M, int D, float eps) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const TI* xr = x + row * ldx;
    f32x4 v[VPL];
    float s = 0.f, ss = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c = (i * 64 + lane) * 4;
        v[i] = c < D ? load4<TI>(xr + c) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) { s += v[i][e]; ss += v[i][e] * v[i][e]; }
    }
    float mean = 0.f, rstd;
    if (RMS) rstd = rsqrtf(wave_sum(ss) / D + eps);
    else {
        mean = wave_sum(s) / D;
        // two-pass variance on the register-resident row (the one-pass form cancels badly on the residual stream's outlier channels)
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int c = (i * 64 + lane) * 4;
            if (c < D) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float d = v[i][e] - mean; q += d * d; }
            }
        }
        rstd = rsqrtf(wave_sum(q) / D + eps);
    }
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < D) {
            const f32x4 wv = *(const f32x4*)(w + c);
            f32x4 bv = {0.f, 0.f, 0.f, 0.f};
            if (!RMS) bv = *(const f32x4*)(b + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[i][e] = RMS ? wv[e] * (v[i][e] * rstd) : (v[i][e] - mean) * rstd * wv[e] + bv[e];
                amax = fmaxf(amax, fabsf(v[i][e]));
            }
        }
    }
    amax = wave_max(amax);
    const float sc = amax > 0.f ? amax * (1.0f / 448.0f) : 1.0f;
    const float inv = 1.0f / sc;
    if (lane == 0) scale[row] = sc;
    unsigned* orow = (unsigned*)(out + row * ldo);
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < ldo) orow[c >> 2] = c < D ? pack4_fp8(v[i][0] * inv, v[i][1] * inv, v[i][2] * inv, v[i][3] * inv) : 0u;
    }
}

}  // namespace vitg

using namespace vitg;

extern "C" int vitg_gemm_fp8(const void* A8, int lda, const float* sa, const void* W8, int ldw, const float* sw, int M, int N, int K,
                             int out_dtype, int mode, const float* bias, int act, void* out, int ldo, const float* res, int ldr,
                             int res_mod, void* stream) {
    if (M <= 0 || N <= 0) return 0;
    if (K <= 0 || K % F8_ROWB != 0) return VITG_ERR_SHAPE;
    if ((lda % 16) || (ldw % 16) || lda < K || ldw < K) return VITG_ERR_ALIGN;
    if (!A8 || !W8 || !sa || !sw || !out) return VITG_ERR_ARG;
    if (mode < 0 || mode > 2) return VITG_ERR_ARG;
    if (mode == VITG_GEMM_SWIGLU && (N % 64 != 0)) return VITG_ERR_SHAPE;
    if (N % 8 != 0 || ldo % 8 != 0 || (res && ldr % 4 != 0)) return VITG_ERR_ALIGN;
    if (out_dtype != DT_BF16 && out_dtype != DT_F16) return VITG_ERR_DTYPE;     // type of the 16-bit outputs (ignored for VITG_GEMM_F32)
    Gemm8Args ga = {};
    GemmArgs& g = ga.g;
    g.A = A8; g.lda = lda; g.W = W8; g.ldw = ldw; g.M = M; g.N = N; g.K = K;
    g.bias = bias; g.act = act; g.out = out; g.ldo = ldo; g.res = res; g.ldr = ldr; g.res_mod = res_mod;
    g.tiles_m = g.tiles_n = 0; g.stagger_ns = 0; g.group_m = 4; g.Wf = nullptr; g.flags = 0; g.split_k = 1; g.split_stride = 0;
    ga.sa = sa; ga.sw = sw;
    const int rc = out_dtype == DT_BF16 ? launch_fp8_t<bf16_t>(ga, mode, (hipStream_t)stream) : launch_fp8_t<f16_t>(ga, mode, (hipStream_t)stream);
    return rc == 0 ? 0 : VITG_ERR_LAUNCH;
}

// 16-bit residual stream on the fp8 weight path (round 4; include/videoitg_hip.h): out16 = round16(scales * (A8 W8^T) + bias + res16), A8 with per-token
// scales (a_mx == NULL) or MX block scales (a_mx set: fc2 behind an MX-producing fc1)
extern "C" int vitg_gemm_fp8_res16(const void* A8, int lda, const float* sa, const void* a_mx, int ld_amx, const void* W8, int ldw, const float* sw,
                                   int M, int N, int K, int dtype, const float* bias, void* out, int ldo, const void* res16, int ldr16, void* stream) {
    if (M <= 0 || N <= 0) return 0;
    if (K <= 0 || K % F8_ROWB != 0) return VITG_ERR_SHAPE;
    if ((lda % 16) || (ldw % 16) || lda < K || ldw < K) return VITG_ERR_ALIGN;
    if (!A8 || !W8 || !sw || !out || !res16 || (!sa && !a_mx)) return VITG_ERR_ARG;
    if (a_mx && ld_amx < M) return VITG_ERR_SHAPE;
    if (N % 8 != 0 || ldo % 4 != 0 || ldr16 % 4 != 0 || ldo < N || ldr16 < N) return VITG_ERR_ALIGN;
    if (dtype != DT_BF16 && dtype != DT_F16) return VITG_ERR_DTYPE;
    Gemm8Args ga = {};
    GemmArgs& g = ga.g;
    g.A = A8; g.lda = lda; g.W = W8; g.ldw = ldw; g.M = M; g.N = N; g.K = K;
    g.bias = bias; g.act = VITG_ACT_NONE; g.out = out; g.ldo = ldo; g.res = nullptr; g.res16 = res16; g.ldr16 = ldr16;
    g.group_m = 4; g.split_k = 1;
    ga.sa = sa; ga.sw = sw; ga.a_mx = (const unsigned char*)a_mx; ga.ld_amx = ld_amx;
    const int rc = dtype == DT_BF16 ? launch_fp8_t<bf16_t>(ga, MODE_F32, (hipStream_t)stream) : launch_fp8_t<f16_t>(ga, MODE_F32, (hipStream_t)stream);
    return rc == 0 ? 0 : VITG_ERR_LAUNCH;
}

// MX forms of vitg_gemm_fp8 (see include/videoitg_hip.h): exactly one of a_mx (consumer, mode F32) / o_mx (producer, mode STORE16)
extern "C" int vitg_gemm_fp8_mx(const void* A8, int lda, const float* sa, const void* a_mx, int ld_amx, const void* W8, int ldw, const float* sw,
                                int M, int N, int K, int out_dtype, int mode, const float* bias, int act, void* out, int ldo, void* o_mx,
                                int ld_omx, const float* res, int ldr, void* stream) {
    if (M <= 0 || N <= 0) return 0;
    if (K <= 0 || K % F8_ROWB != 0) return VITG_ERR_SHAPE;
    if ((lda % 16) || (ldw % 16) || lda < K || ldw < K) return VITG_ERR_ALIGN;
    if (!A8 || !W8 || !sw || !out) return VITG_ERR_ARG;
    if ((a_mx != nullptr) == (o_mx != nullptr)) return VITG_ERR_ARG;
    if (out_dtype != DT_BF16 && out_dtype != DT_F16) return VITG_ERR_DTYPE;
    if (a_mx) {                                         // block-scaled activations into the fp32-residual GEMM
        if (mode != VITG_GEMM_F32 || act != VITG_ACT_NONE) return VITG_ERR_ARG;
        if (ld_amx < M) return VITG_ERR_ALIGN;
        if (N % 8 != 0 || ldo % 8 != 0 || (res && ldr % 4 != 0)) return VITG_ERR_ALIGN;
    } else {                                            // e4m3 + block scales out
        if ((mode != VITG_GEMM_STORE16 && mode != VITG_GEMM_SWIGLU) || !sa) return VITG_ERR_ARG;
        if (mode == VITG_GEMM_SWIGLU && N % 64 != 0) return VITG_ERR_SHAPE;
        if (N % 8 != 0 || ldo % 128 != 0 || ldo < (mode == VITG_GEMM_SWIGLU ? N / 2 : N) || ld_omx < M) return VITG_ERR_ALIGN;
    }
    Gemm8Args ga = {};
    GemmArgs& g = ga.g;
    g.A = A8; g.lda = lda; g.W = W8; g.ldw = ldw; g.M = M; g.N = N; g.K = K;
    g.bias = bias; g.act = act; g.out = out; g.ldo = ldo; g.res = res; g.ldr = ldr; g.res_mod = 0;
    g.tiles_m = g.tiles_n = 0; g.stagger_ns = 0; g.group_m = 4; g.Wf = nullptr; g.flags = 0; g.split_k = 1; g.split_stride = 0;
    ga.sa = sa; ga.sw = sw; ga.a_mx = (const unsigned char*)a_mx; ga.ld_amx = ld_amx; ga.o_mx = (unsigned char*)o_mx; ga.ld_omx = ld_omx;
    const int rc = out_dtype == DT_BF16 ? launch_fp8_t<bf16_t>(ga, mode, (hipStream_t)stream) : launch_fp8_t<f16_t>(ga, mode, (hipStream_t)stream);
    return rc == 0 ? 0 : VITG_ERR_LAUNCH;
}

extern "C" int vitg_quant_rows_fp8(const void* x, long ldx, int in_dtype, void* out8, long ldo, float* scale, long M, int D,
                                   void* stream) {
    if (M <= 0) return 0;
    if (D <= 0 || D % 4 || ldx % 4 || ldo % 16 || ldo < D) return VITG_ERR_ALIGN;
    hipStream_t s = (hipStream_t)stream;
    if (ldo <= 8192) {                   // one wave per row, row in registers
        dim3 g4((unsigned)((M + 3) / 4)), b4(256);
#define QW(T, V) hipLaunchKernelGGL((quant_rows_wave_kernel<T, V>), g4, b4, 0, s, (const T*)x, ldx, (unsigned char*)out8, ldo, scale, M, D)
#define QWD(T) do { if (ldo <= 1280) QW(T, 5); else if (ldo <= 2048) QW(T, 8); else if (ldo <= 4608) QW(T, 18); else QW(T, 32); } while (0)
        if (in_dtype == DT_F32) QWD(float);
        else if (in_dtype == DT_BF16) QWD(bf16_t);
        else if (in_dtype == DT_F16) QWD(f16_t);
        else return VITG_ERR_DTYPE;
#undef QWD
#undef QW
        return hipGetLastError() == hipSuccess ? 0 : VITG_ERR_LAUNCH;
    }
    dim3 grid((unsigned)M), blk(256);
    if (in_dtype == DT_F32) hipLaunchKernelGGL((quant_rows_kernel<float>), grid, blk, 0, s, (const float*)x, ldx, (unsigned char*)out8, ldo, scale, D);
    else if (in_dtype == DT_BF16) hipLaunchKernelGGL((quant_rows_kernel<bf16_t>), grid, blk, 0, s, (const bf16_t*)x, ldx, (unsigned char*)out8, ldo, scale, D);
    else if (in_dtype == DT_F16) hipLaunchKernelGGL((quant_rows_kernel<f16_t>), grid, blk, 0, s, (const f16_t*)x, ldx, (unsigned char*)out8, ldo, scale, D);
    else return VITG_ERR_DTYPE;
    return hipGetLastError() == hipSuccess ? 0 : VITG_ERR_LAUNCH;
}

// vitg_norm_quant_fp8 on 16-bit rows (the tower's 16-bit residual stream): LayerNorm (b != NULL) / RMSNorm of x16 [M, ldx] -> e4m3 + per-token scales
extern "C" int vitg_norm_quant_fp8_16(const void* x, long ldx, int in_dtype, const float* w, const float* b, void* out8, long ldo, float* scale,
                                      long M, int D, float eps, void* stream) {
    if (M <= 0) return 0;
    if (D <= 0 || D > 4096 || ldo > 4096 || ldo < D) return VITG_ERR_SHAPE;
    if (D % 4 || ldx % 4 || ldo % 16) return VITG_ERR_ALIGN;
    if (!w || !x) return VITG_ERR_ARG;
    if (in_dtype != DT_BF16 && in_dtype != DT_F16) return VITG_ERR_DTYPE;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid((unsigned)((M + 3) / 4)), blk(256);
#define NQ16(R, V) do { if (in_dtype == DT_F16) hipLaunchKernelGGL((norm_quant_kernel<R, V, f16_t>), grid, blk, 0, s, (const f16_t*)x, ldx, w, b, (unsigned char*)out8, ldo, scale, M, D, eps); \
                        else hipLaunchKernelGGL((norm_quant_kernel<R, V, bf16_t>), grid, blk, 0, s, (const bf16_t*)x, ldx, w, b, (unsigned char*)out8, ldo, scale, M, D, eps); } while (0)
    if (b) { if (ldo <= 1280) NQ16(false, 5); else NQ16(false, 16); }
    else { if (ldo <= 1280) NQ16(true, 5); else NQ16(true, 16); }
#undef NQ16
    return hipGetLastError() == hipSuccess ? 0 : VITG_ERR_LAUNCH;
}

extern "C" int vitg_norm_quant_fp8(const float* x, long ldx, const float* w, const float* b, void* out8, long ldo, float* scale,
                                   long M, int D, float eps, void* stream) {
    if (M <= 0) return 0;
    if (D <= 0 || D > 4096 || ldo > 4096 || ldo < D) return VITG_ERR_SHAPE;          // (the row lives in registers: hidden sizes up to 4096)
    if (D % 4 || ldx % 4 || ldo % 16) return VITG_ERR_ALIGN;
    if (!w) return VITG_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid((unsigned)((M + 3) / 4)), blk(256);
#define NQ(R, V) hipLaunchKernelGGL((norm_quant_kernel<R, V>), grid, blk, 0, s, x, ldx, w, b, (unsigned char*)out8, ldo, scale, M, D, eps)
    if (b) { if (ldo <= 1280) NQ(false, 5); else NQ(false, 16); }
    else { if (ldo <= 1280) NQ(true, 5); else NQ(true, 16); }
#undef NQ
    return hipGetLastError() == hipSuccess ? 0 : VITG_ERR_LAUNCH;
}

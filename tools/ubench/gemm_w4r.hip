// gemm_w4r: go / no-go probe for the one-wave-per-SIMD GEMM main loop.  Stand-alone (no torch): hipcc --offload-arch=gfx950 -O3.
//
//   C[M,N] (fp16) = A[M,K] (fp16, row-major) * W[N,K]^T (fp16, row-major), fp32 accumulation on v_mfma_f32_16x16x32_f16.
//   256 x 256 x 64 block tile, FOUR waves (2 x 2), each wave a 128 x 128 output tile in 256 accumulator registers (the AGPR half of the
//   512-register file), one wave per SIMD.  LDS fragment bytes per flop: (128 + 128) / (128 * 128) = 0.0156 against 0.0182 - 0.0208 for the
//   shipped 8-wave tiles (gemm_mi16.hip).
//
// What differs from round 1's gemm_w4.hip (1064 TF/s; 1479 with the operand stream removed): the operands do NOT come through LDS-DMA.  An
// LDS-DMA piece costs its issuing wave >= 60 cycles of issue (MI355X_MICROARCH.md, "LDS-DMA piece ... issue cost"), and with one wave per SIMD
// nothing else issues MFMAs meanwhile: 16 pieces per K-tile and wave = half the K-tile's 2048 MFMA cycles.  Here the operand tile of K-tile
// kt + 2 is requested with plain global_load_dwordx4 into 64 staging registers (the registers a 512-register wave has to spare), written to
// the LDS stage of kt + 1 with ds_write_b128 one K-tile later, and the only barrier of a K-tile sits behind its MFMAs.
//
//   per wave and K-tile: 128 MFMAs (2048 pipe cycles), 32 ds_read_b128, 16 ds_write_b128, 16 global_load_dwordx4, 1 s_barrier.
//
// Usage: gemm_w4r M N K [reps] [check]     (M, N multiples of 256, K of 64)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
#include <type_traits>

typedef _Float16 f16;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int BM = 256, BN = 256, BK = 64, ROWB = 128;
constexpr int A_BYTES = BM * ROWB, W_BYTES = BN * ROWB, STAGE = A_BYTES + W_BYTES;      // 32 + 32 KiB; two stages = 128 KiB
constexpr int MT = 8, NT = 8;                                                            // 16-row / 16-column MFMA tiles per wave

// VARIANT: ablation bits (results WRONG when any of 1..15 is set; timing only): 1 no operand requests in the loop | 2 no deposits | 4 no K-tile barrier |
// 8 no fragment reads in the loop | 16 no output stores
#ifndef VARIANT
#define VARIANT 0
#endif
#ifndef SCHED
#define SCHED 1
#endif

__device__ __forceinline__ void tile_of(int bid, int tiles_m, int tiles_n, int& tm, int& tn) {
    // XCD-aware (block b runs on XCD b % 8): every XCD walks a contiguous run of the group-major tile list, groups of 4 M-tiles
    const int nwg = tiles_m * tiles_n, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    const int gm = 4, per_group = gm * tiles_n, grp = bid / per_group, first_m = grp * gm;
    const int gsz = min(tiles_m - first_m, gm), in_g = bid - grp * per_group;
    tm = first_m + in_g % gsz;
    tn = in_g / gsz;
}

__global__ __launch_bounds__(256, 1) void gemm_w4r_kernel(const f16* __restrict__ A, const f16* __restrict__ W, f16* __restrict__ C, int M, int N, int K,
                                                          int tiles_m, int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int r16 = lane & 15, fq = lane >> 4;
    const int nk = K / BK;
    for (int vb = blockIdx.x; vb < tiles_m * tiles_n; vb += gridDim.x) {
        int tm, tn;
        tile_of(vb, tiles_m, tiles_n, tm, tn);
        const int m0 = tm * BM, n0 = tn * BN;
        // ---- operand requests: one instruction = 8 rows x 128 B (lane -> row lane / 8, 16-B chunk lane % 8); wave w owns rows [64 w, 64 w + 64) of both tiles
        const int drow = lane >> 3, dch = lane & 7;
        const char* ap = (const char*)A + ((size_t)(m0 + wave * 64 + drow) * K + dch * 8) * 2;
        const char* wp = (const char*)W + ((size_t)(n0 + wave * 64 + drow) * K + dch * 8) * 2;
        const size_t rstep = (size_t)8 * K * 2;
        // LDS byte offset of this lane's chunk inside a stage: row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); row = 64 w + 8 j + drow
        int st_off[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int row = wave * 64 + j * 8 + drow;
            st_off[j] = row * ROWB + ((dch ^ ((row >> 1) & 7)) << 4);
        }
        u32x4 sa[8], sw[8];
        auto request = [&](int kt) {
#pragma unroll
            for (int j = 0; j < 8; ++j) sa[j] = *(const u32x4*)(ap + j * rstep + (size_t)kt * ROWB);
#pragma unroll
            for (int j = 0; j < 8; ++j) sw[j] = *(const u32x4*)(wp + j * rstep + (size_t)kt * ROWB);
        };
        auto deposit = [&](int kt) {
            char* st = smem + (kt & 1) * STAGE;
#pragma unroll
            for (int j = 0; j < 8; ++j) *(u32x4*)(st + st_off[j]) = sa[j];
#pragma unroll
            for (int j = 0; j < 8; ++j) *(u32x4*)(st + A_BYTES + st_off[j]) = sw[j];
        };
        // fragment offsets: rows base + 16 i + r16 share the swizzle term of r16
        int a_off[2], w_off[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int ch = ((ks * 4 + fq) ^ ((r16 >> 1) & 7)) << 4;
            a_off[ks] = (wr * 128 + r16) * ROWB + ch;
            w_off[ks] = A_BYTES + (wc * 128 + r16) * ROWB + ch;
        }
        f32x4 acc[NT][MT];
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int j = 0; j < MT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        f16x8 fa0[MT], fw0[NT], fa1[MT], fw1[NT];
        // The MFMAs are inline asm with the accumulator TIED ("+a": destination = third source, in the AGPR half): written with the builtin, hipcc
        // gives every MFMA a destination different from its accumulator input and shuffles 200+ v_accvgpr_mov / read / write through each K-tile
        // (and spills 177 registers) - what sank gemm_w4.hip's compiler-scheduled body in round 1.  Everything else (ds_read / ds_write / global
        // loads, and therefore every s_waitcnt) stays with the compiler; sched_barrier(0) after each slot pins the source order.
#define MFMA(c, a, b) asm("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b))
        // fragment read #i of a k-step, in the order the MFMA sweep (nt outer, mt inner) first needs them: fw[0], fa[0..7], fw[1..7]
        auto frag_read = [&](const char* st, int ks, int i, f16x8* fa, f16x8* fw) {
            if (i == 0) fw[0] = *(const f16x8*)(st + w_off[ks]);
            else if (i <= 8) fa[i - 1] = *(const f16x8*)(st + a_off[ks] + (i - 1) * (16 * ROWB));
            else fw[i - 8] = *(const f16x8*)(st + w_off[ks] + (i - 8) * (16 * ROWB));
        };

        __builtin_amdgcn_s_barrier();                       // the previous tile's epilogue is done with the LDS
        request(0);
        deposit(0);
        request(nk > 1 ? 1 : 0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int i = 0; i < 16; ++i) frag_read(smem, 0, i, fa0, fw0);
#if SCHED == 0
#pragma unroll 1
        for (int kt = 0; kt < nk; ++kt) {
            const char* st = smem + (kt & 1) * STAGE;
            char* const so = smem + ((kt + 1) & 1) * STAGE;
            const size_t k2 = (size_t)min(kt + 2, nk - 1) * ROWB;          // past the end: a harmless re-request of the last tile (no branch in the body)
            __builtin_amdgcn_sched_barrier(0);
            // ---- phase A: 64 MFMAs on k-step 0; slots 0..30 (even): the 16 fragment reads of k-step 1; slots 32..62 (even): the 16 deposits of tile kt + 1
#pragma unroll
            for (int i = 0; i < 64; ++i) {
                if (!(VARIANT & 8) && i < 32 && i % 2 == 0) frag_read(st, 1, i / 2, fa1, fw1);
                if (!(VARIANT & 2) && i >= 32 && i % 2 == 0) {
                    const int j = (i - 32) / 2;
                    if (j < 8) *(u32x4*)(so + st_off[j]) = sa[j];
                    else *(u32x4*)(so + A_BYTES + st_off[j - 8]) = sw[j - 8];
                }
                MFMA(acc[i / 8][i % 8], fw0[i / 8], fa0[i % 8]);
                __builtin_amdgcn_sched_barrier(0);
            }
            // ---- phase B: 64 MFMAs on k-step 1; slots 0..45 (every third): the 16 requests of tile kt + 2; then the K-tile's barrier; the last 16 slots
            // carry the fragment reads of the next tile's k-step 0
#pragma unroll
            for (int i = 0; i < 48; ++i) {
                if (!(VARIANT & 1) && i % 3 == 0) {
                    const int j = i / 3;
                    if (j < 8) sa[j] = *(const u32x4*)(ap + j * rstep + k2);
                    else sw[j - 8] = *(const u32x4*)(wp + (j - 8) * rstep + k2);
                }
                MFMA(acc[i / 8][i % 8], fw1[i / 8], fa1[i % 8]);
                __builtin_amdgcn_sched_barrier(0);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // this wave's deposits are in the LDS
            if (!(VARIANT & 4)) __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 48; i < 64; ++i) {
                if (!(VARIANT & 8)) frag_read(so, 0, i - 48, fa0, fw0);
                MFMA(acc[i / 8][i % 8], fw1[i / 8], fa1[i % 8]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#else
        // SCHED 1: the four waves of a block run the K-tile in lockstep (one barrier per K-tile), so an instruction placed in one slot is issued by all four
        // at the same moment - and the LDS write path (ds_write_b128: 13 cycles per wave-instruction for the CU) and the texture path then serve four
        // queued requests while every wave's MFMA stream waits behind its own (measured: the 16 deposits of a K-tile cost 25 % of the loop, the 16
        // requests 16 %; gpurun_out/r05_c3).  Here wave W deposits staging register j in slot 7 j + 2 W and re-requests it in the next slot: no two waves
        // write (or request) in the same slot, a staging register is free again one slot after its deposit, and both streams are spread over the K-tile.
        // 32-bit lane offsets against wave-uniform bases (saddr form): no 64-bit VALU address arithmetic per request.
        // one lane offset serves all 16 requests (the 8-row step j * 8 K * 2 goes into the wave-uniform base), two serve all 16 deposits (the swizzle term of
        // row 64 w + 8 j + drow depends on j's parity only; the rest is an immediate offset)
        unsigned g_lo = (unsigned)((drow * K + dch * 8) * 2);
        int l_lo[2];
#pragma unroll
        for (int p2 = 0; p2 < 2; ++p2) {
            const int row = wave * 64 + p2 * 8 + drow;
            l_lo[p2] = row * ROWB + ((dch ^ ((row >> 1) & 7)) << 4);
        }
        const char* const abase = (const char*)A + (size_t)(m0 + wave * 64) * K * 2;
        const char* const wbase = (const char*)W + (size_t)(n0 + wave * 64) * K * 2;
#define SLOT(i, DW)                                                                                                              \
        {                                                                                                                                \
            if (!(VARIANT & 8) && (i) < 32 && (i) % 2 == 0) frag_read(st, 1, (i) / 2, fa1, fw1);                                         \
            if (!(VARIANT & 8) && (i) >= 112) frag_read(so, 0, (i) - 112, fa0, fw0);                                                     \
            if (!(VARIANT & 2) && (i) >= (DW) && ((i) - (DW)) % 7 == 0 && ((i) - (DW)) / 7 < 16) {                                       \
                const int j_ = ((i) - (DW)) / 7;                                                                                        \
                if (j_ < 8) *(u32x4*)(so + l_lo[j_ & 1] + ((j_ & 7) >> 1) * (16 * ROWB)) = sa[j_ & 7];                                   \
                else *(u32x4*)(so + A_BYTES + l_lo[j_ & 1] + ((j_ & 7) >> 1) * (16 * ROWB)) = sw[j_ & 7];                                \
            }                                                                                                                            \
            if (!(VARIANT & 1) && (i) >= (DW) + 1 && ((i) - (DW) - 1) % 7 == 0 && ((i) - (DW) - 1) / 7 < 16) {                           \
                const int j_ = ((i) - (DW) - 1) / 7;                                                                                    \
                unsigned off_ = g_lo;                                                                                                    \
                asm volatile("" : "+v"(off_));                                                                                           \
                if (j_ < 8) sa[j_ & 7] = *(const u32x4*)(ak + (size_t)(j_ & 7) * rstep + off_);                                          \
                else sw[j_ & 7] = *(const u32x4*)(wk + (size_t)(j_ & 7) * rstep + off_);                                                 \
            }                                                                                                                            \
            if ((i) < 64) MFMA(acc[((i) & 63) / 8][(i) % 8], fw0[((i) & 63) / 8], fa0[(i) % 8]);                                         \
            else MFMA(acc[((i) & 63) / 8][(i) % 8], fw1[((i) & 63) / 8], fa1[(i) % 8]);                                                  \
            __builtin_amdgcn_sched_barrier(0);                                                                                           \
        }
#define KLOOP(DW)                                                                                                                \
        _Pragma("unroll 1") for (int kt = 0; kt < nk; ++kt) {                                                                        \
            const char* st = smem + (kt & 1) * STAGE;                                                                                    \
            char* const so = smem + ((kt + 1) & 1) * STAGE;                                                                              \
            const size_t k2 = (size_t)min(kt + 2, nk - 1) * ROWB;                                                                        \
            const char* const ak = abase + k2;                                                                                           \
            const char* const wk = wbase + k2;                                                                                           \
            __builtin_amdgcn_sched_barrier(0);                                                                                           \
            _Pragma("unroll") for (int i = 0; i < 56; ++i) SLOT(i, DW)                                                                   \
            _Pragma("unroll") for (int i = 56; i < 112; ++i) SLOT(i, DW)                                                                 \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                           \
            if (!(VARIANT & 4)) __builtin_amdgcn_s_barrier();                                                                            \
            __builtin_amdgcn_sched_barrier(0);                                                                                           \
            _Pragma("unroll") for (int i = 112; i < 128; ++i) SLOT(i, DW)                                                                \
        }
        if (wave == 0) { KLOOP(0) }
        else if (wave == 1) { KLOOP(2) }
        else if (wave == 2) { KLOOP(4) }
        else { KLOOP(6) }
#endif
        asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");             // the last MFMAs' results have left the pipe before the compiler reads the accumulators
        // ---- epilogue: the wave's 128 x 128 tile through its own 32 KiB + pad LDS window -> full 256-B row segments ----------------
        __builtin_amdgcn_s_barrier();
        constexpr int ER = 272;                             // padded row pitch of the window (128 fp16 = 256 B + 16)
        char* const win = smem + wave * (128 * ER);
        int tid_e = threadIdx.x;                            // lane quantities of the epilogue from an opaque copy: hoisted above the K-loop they would stay live through it
        asm volatile("" : "+v"(tid_e));
        const int lane_e = tid_e & 63, r16e = lane_e & 15, fqe = lane_e >> 4;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const f32x4 v = acc[nt][mt];
                f16x4 h = {(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
                *(f16x4*)(win + (mt * 16 + r16e) * ER + (nt * 16 + fqe * 4) * 2) = h;
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const int erow = lane_e >> 4, ech = lane_e & 15;    // 4 rows x 16 chunks of 16 B per instruction
        char* const cb = (char*)C + ((size_t)(m0 + wr * 128) * N + n0 + wc * 128) * 2;
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const int row = i * 4 + erow;
            const u32x4 v = *(const u32x4*)(win + row * ER + ech * 16);
            if (!(VARIANT & 16) || v[0] == 0x12345678u) *(u32x4*)(cb + (size_t)row * N * 2 + ech * 16) = v;
        }
    }
}

static uint32_t rng_state = 12345u;
static inline float rnd() { rng_state = rng_state * 1664525u + 1013904223u; return ((rng_state >> 8) & 0xFFFF) / 65536.0f * 2.0f - 1.0f; }

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 8192, N = argc > 2 ? atoi(argv[2]) : 8192, K = argc > 3 ? atoi(argv[3]) : 8192;
    const int reps = argc > 4 ? atoi(argv[4]) : 20, check = argc > 5 ? atoi(argv[5]) : 1;
    if (M % 256 || N % 256 || K % 64) { fprintf(stderr, "M, N multiples of 256, K of 64\n"); return 2; }
    std::vector<f16> ha((size_t)M * K), hw((size_t)N * K);
    for (auto& v : ha) v = (f16)rnd();
    const float wscale = 1.0f / sqrtf((float)K);
    for (auto& v : hw) v = (f16)(rnd() * wscale);
    f16 *A, *W, *C;
    CHECK(hipMalloc(&A, ha.size() * 2)); CHECK(hipMalloc(&W, hw.size() * 2)); CHECK(hipMalloc(&C, (size_t)M * N * 2));
    CHECK(hipMemcpy(A, ha.data(), ha.size() * 2, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(W, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
    const int lds = 2 * STAGE > 4 * 128 * 272 ? 2 * STAGE : 4 * 128 * 272;
    CHECK(hipFuncSetAttribute((const void*)gemm_w4r_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    int cus = 256;
    CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    const int tiles_m = M / 256, tiles_n = N / 256, grid = tiles_m * tiles_n < cus ? tiles_m * tiles_n : cus;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(gemm_w4r_kernel, dim3(grid), dim3(256), lds, 0, A, W, C, M, N, K, tiles_m, tiles_n);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(gemm_w4r_kernel, dim3(grid), dim3(256), lds, 0, A, W, C, M, N, K, tiles_m, tiles_n);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    printf("gemm_w4r sched " "%d variant %d  %d x %d x %d fp16: %.4f ms  %.1f TFLOP/s\n", SCHED, VARIANT, M, N, K, ms, 2.0 * M * N * K / (ms * 1e-3) / 1e12);
    if (check) {
        std::vector<f16> hc((size_t)M * N);
        CHECK(hipMemcpy(hc.data(), C, hc.size() * 2, hipMemcpyDeviceToHost));
        double worst = 0;
        int bad = 0;
        for (int s = 0; s < 4096; ++s) {                    // sampled entries (every tile position class: rows / columns drawn over the whole matrix)
            rng_state = rng_state * 1664525u + 1013904223u; const int m = (rng_state >> 4) % M;
            rng_state = rng_state * 1664525u + 1013904223u; const int n = (rng_state >> 4) % N;
            double ref = 0;
            for (int k = 0; k < K; ++k) ref += (double)(float)ha[(size_t)m * K + k] * (double)(float)hw[(size_t)n * K + k];
            const double d = fabs((double)(float)hc[(size_t)m * N + n] - ref), tol = 4e-3 + 2e-3 * fabs(ref);
            if (d > worst) worst = d;
            if (d > tol) { if (bad < 5) fprintf(stderr, "mismatch C[%d][%d] = %f, want %f\n", m, n, (double)(float)hc[(size_t)m * N + n], ref); ++bad; }
        }
        printf("check: %d of 4096 sampled entries off, worst |d| = %.4g -> %s\n", bad, worst, bad ? "FAIL" : "ok");
        if (bad) return 1;
    }
    return 0;
}

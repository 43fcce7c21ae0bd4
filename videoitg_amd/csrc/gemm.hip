// C[M,N] = A[M,K] * W[N,K]^T with fused epilogues, 16-bit inputs, fp32 MFMA accumulation (gfx950).
//
// Both operands are K-contiguous (activations row-major, nn.Linear weights [out,in]), which is the
// natural MFMA layout: no transposes anywhere.  Tile 128x128x64, 4 waves (2x2), each wave a 64x64
// sub-tile as 2x2 v_mfma_f32_32x32x16.  The MFMA "A" operand is the WEIGHT tile and the "B" operand the
// ACTIVATION tile, so the accumulator of a lane holds 4 consecutive output features of one token row:
// epilogue loads/stores are 8-16 B per lane and bias / residual reads vectorise.
//
// LDS: [128 rows][8 x 16-B chunks] per operand, chunk index XOR ((row>>1)&7): conflict-free for the
// ds_read_b128 lane groups (MI355X_MICROARCH.md §LDS).  Double-buffered, one barrier per K-tile; global
// loads of tile t+1 are issued before the MFMAs of tile t and written to LDS after them.
//
// Work-group order: XCD-contiguous remap (block b runs on XCD b%8) + 8-row-tile groups, so the ~64
// concurrently resident blocks of an XCD form an 8x8 super-tile that shares operand panels in its L2.
#include "gemm_common.h"
#include <stdlib.h>

namespace vitg {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int GEMM_THREADS = 256;



__device__ __forceinline__ int lds_off(int row, int chunk) {        // bytes within one operand tile
    return row * (BK * 2) + ((chunk ^ ((row >> 1) & 7)) << 4);
}

template <typename T, int MODE>
__global__ __launch_bounds__(GEMM_THREADS, 2) void gemm128_kernel(GemmArgs g) {
    typedef typename T16<T>::vec8 vec8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // [buf][A|W][128*128B]
    constexpr int TILE_BYTES = BM * BK * 2;                        // 16 KiB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    int tm, tn;
    block_to_tile(blockIdx.x, g.tiles_m, g.tiles_n, 8, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;

    // ---- staging assignment: 1024 chunks per operand tile, 4 per thread ---------------------
    const char* Ab = (const char*)g.A;
    const char* Wb = (const char*)g.W;
    const char* a_src[4];
    const char* w_src[4];
    int st_off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = tid + i * GEMM_THREADS;
        const int row = c >> 3, kc = c & 7;
        const int am = min(m0 + row, g.M - 1), wn_ = min(n0 + row, g.N - 1);
        a_src[i] = Ab + ((size_t)am * g.lda + kc * 8) * 2;
        w_src[i] = Wb + ((size_t)wn_ * g.ldw + kc * 8) * 2;
        st_off[i] = lds_off(row, kc);
    }

    f32x16 acc[2][2];   // [nt][mt]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    u32x4 ra[4], rw[4];
    const int nk = g.K / BK;

    auto load_tile = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ra[i] = *(const u32x4*)(a_src[i] + (size_t)kt * (BK * 2));
            rw[i] = *(const u32x4*)(w_src[i] + (size_t)kt * (BK * 2));
        }
    };
    auto store_tile = [&](int buf) {
        char* base = smem + buf * (2 * TILE_BYTES);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *(u32x4*)(base + st_off[i]) = ra[i];
            *(u32x4*)(base + TILE_BYTES + st_off[i]) = rw[i];
        }
    };

    const int frow = lane & 31, fh = lane >> 5;
    auto compute = [&](int buf) {
        const char* As = smem + buf * (2 * TILE_BYTES);
        const char* Ws = As + TILE_BYTES;
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            vec8 fa[2], fw[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                fa[t] = *(const vec8*)(As + lds_off(wm * 64 + t * 32 + frow, ks * 2 + fh));
                fw[t] = *(const vec8*)(Ws + lds_off(wn * 64 + t * 32 + frow, ks * 2 + fh));
            }
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) acc[nt][mt] = T16<T>::mfma32(fw[nt], fa[mt], acc[nt][mt]);
        }
    };

    load_tile(0);
    store_tile(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = kt + 1 < nk;
        if (more) load_tile(kt + 1);
        compute(kt & 1);
        if (more) store_tile((kt + 1) & 1);
        __syncthreads();
    }

    // ---- epilogue: lane holds C[m][n..n+3] for 4 column groups per 32x32 tile ----------------
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int m = m0 + wm * 64 + mt * 32 + frow;
        if (m >= g.M) continue;
        if constexpr (MODE == MODE_SWIGLU) {
            // weight rows interleaved [32 gate | 32 up]: tile nt=0 gate, nt=1 up, same lane/register
            T* out = (T*)g.out;
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const int nn = n0 + wn * 64 + q4 * 8 + fh * 4;       // column in the interleaved N space (gate part)
                const int no = (n0 + wn * 64) / 2 + q4 * 8 + fh * 4;  // output column
                if (nn >= g.N) continue;
                typename T16<T>::vec4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float gt = acc[0][mt][q4 * 4 + e], up = acc[1][mt][q4 * 4 + e];
                    v[e] = T16<T>::from_f32(silu_f(gt) * up);
                }
                *(typename T16<T>::vec4*)(out + (size_t)m * g.ldo + no) = v;
            }
        } else {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const int n = n0 + wn * 64 + nt * 32 + q4 * 8 + fh * 4;
                    if (n >= g.N) continue;
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[nt][mt][q4 * 4 + e];
                    if (g.bias) v += *(const f32x4*)(g.bias + n);
                    if (g.act == VITG_ACT_GELU_TANH) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = gelu_tanh_f(v[e]);
                    } else if (g.act == VITG_ACT_GELU_ERF) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = gelu_erf_f(v[e]);
                    }
                    if constexpr (MODE == MODE_F32) {
                        if (g.res) {
                            const int rm = g.res_mod > 0 ? m % g.res_mod : m;
                            v += *(const f32x4*)(g.res + (size_t)rm * g.ldr + n);
                        }
                        *(f32x4*)((float*)g.out + (size_t)m * g.ldo + n) = v;
                    } else {
                        if (g.res) {                     // vitg_gemm_res16 forms: fp32 table (patch embedding + positions) ...
                            const int rm = g.res_mod > 0 ? m % g.res_mod : m;
                            v += *(const f32x4*)(g.res + (size_t)rm * g.ldr + n);
                        }
                        if (g.res16) {                   // ... or the 16-bit residual row (a lane reads the quad it overwrites: in place is safe)
                            const typename T16<T>::vec4 r = *(const typename T16<T>::vec4*)((const T*)g.res16 + (size_t)m * g.ldr16 + n);
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] += T16<T>::to_f32(r[e]);
                        }
                        typename T16<T>::vec4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = T16<T>::from_f32(v[e]);
                        *(typename T16<T>::vec4*)((T*)g.out + (size_t)m * g.ldo + n) = o;
                    }
                }
        }
    }
}

template <typename T>
static int launch(const GemmArgs& g, int mode, hipStream_t s) {
    const int grid = g.tiles_m * g.tiles_n;
    const size_t lds = 2 * 2 * BM * BK * 2;   // 64 KiB
    if (mode == MODE_STORE16) hipLaunchKernelGGL((gemm128_kernel<T, MODE_STORE16>), dim3(grid), dim3(GEMM_THREADS), lds, s, g);
    else if (mode == MODE_F32) hipLaunchKernelGGL((gemm128_kernel<T, MODE_F32>), dim3(grid), dim3(GEMM_THREADS), lds, s, g);
    else hipLaunchKernelGGL((gemm128_kernel<T, MODE_SWIGLU>), dim3(grid), dim3(GEMM_THREADS), lds, s, g);
    return (int)hipGetLastError();
}

}  // namespace vitg

using namespace vitg;

static void set_attrs() {
    static std::atomic<unsigned> done_mask{0};   // per device: the attribute belongs to the device's copy of the kernel
    const int cur_dev = current_device();
    if (device_done(done_mask, cur_dev)) return;
    const int lds = 2 * 2 * BM * BK * 2;
#define SETA(T, Mo) hipFuncSetAttribute((const void*)gemm128_kernel<T, Mo>, hipFuncAttributeMaxDynamicSharedMemorySize, lds)
    SETA(bf16_t, MODE_STORE16); SETA(bf16_t, MODE_F32); SETA(bf16_t, MODE_SWIGLU);
    SETA(f16_t, MODE_STORE16); SETA(f16_t, MODE_F32); SETA(f16_t, MODE_SWIGLU);
#undef SETA
    mark_device_done(done_mask, cur_dev);
}

#if VITG_PROFILING_BUILD
static unsigned long long* g_stamp_buf = nullptr;      // vitg_gemm_timeline_buffer
static int g_stamp_blocks = 0;
#endif

struct NormFusion {               // fused normalisation around a GEMM (see GemmArgs::nrm_*): producer or consumer fields, never both
    const float* gamma = nullptr; void* xh = nullptr; int ldxh = 0; float* part = nullptr; int P = 0;
    const float* stats = nullptr; const float* colsum = nullptr;
};

static int gemm_entry(const void* A, int lda, const void* W, int ldw, const void* Wf, int M, int N, int K, int dtype,
                      int mode, const float* bias, int act, void* out, int ldo,
                      const float* res, int ldr, int res_mod, void* stream, const NormFusion* nf = nullptr,
                      const void* res16 = nullptr, int ldr16 = 0) {
    if (M <= 0 || N <= 0) return 0;
    if (K <= 0 || K % BK != 0) return VITG_ERR_SHAPE;
    if ((lda % 8) || (ldw % 8)) return VITG_ERR_ALIGN;
    if (mode < 0 || mode > 2) return VITG_ERR_ARG;
    if (mode == VITG_GEMM_SWIGLU && (N % 64 != 0)) return VITG_ERR_SHAPE;
    if (mode != VITG_GEMM_SWIGLU && (N % 4 != 0 || ldo % 4 != 0)) return VITG_ERR_ALIGN;
    if (dtype != DT_BF16 && dtype != DT_F16) return VITG_ERR_DTYPE;
    set_attrs();
    GemmArgs g = {};
    g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.M = M; g.N = N; g.K = K;
    g.bias = bias; g.act = act; g.out = out; g.ldo = ldo; g.res = res; g.ldr = ldr; g.res_mod = res_mod;
    g.tiles_m = ceil_div(M, BM); g.tiles_n = ceil_div(N, BN); g.stagger_ns = 0; g.group_m = 4; g.Wf = Wf;
    g.split_k = 1; g.split_stride = 0;
    g.res16 = res16; g.ldr16 = ldr16;
#if VITG_PROFILING_BUILD
    g.stamps = g_stamp_buf; g.stamp_blocks = g_stamp_blocks;
#endif
    if (nf) {                                // (callers exist in the profiling build only)
        g.nrm_gamma = nf->gamma; g.nrm_xh = nf->xh; g.nrm_ldxh = nf->ldxh; g.nrm_part = nf->part; g.nrm_P = nf->P;
        g.nrm_stats = nf->stats; g.nrm_colsum = nf->colsum;
        // the fused forms live in the 256-tile kernels' staged epilogues only
        if (tuning().gemm_tile == 128 || !gemm256_applicable(g, mode)) return VITG_ERR_SHAPE;
        const int rc = launch_gemm256(g, dtype, mode, (hipStream_t)stream);
        return rc == 0 ? 0 : VITG_ERR_LAUNCH;
    }
    // bits 1-4 and 6 of gemm_flags are profiling ablations that produce WRONG results; they only take effect in a process that
    // opted in with VITG_PROFILING=1 (tools/ set it), so a stray option or environment variable cannot corrupt a product run
    static const bool profiling = [] { const char* e = getenv("VITG_PROFILING"); return e && atoi(e) != 0; }();
    g.flags = profiling ? RESEARCH(gemm_flags) : (RESEARCH(gemm_flags) & ~(2 | 4 | 8 | 16 | 64 | 0xF0000));
    hipStream_t s = (hipStream_t)stream;
    const int force = tuning().gemm_tile;          // 0 auto | 128 | 256 (vitg_set_option / VITG_GEMM_TILE)
    int rc;
    // (only the three tile configurations that exist in gemm_mi16.hip - 6 / 9 / 10, with the shape conditions launch256_m puts on
    // 9 and 10 - carry the 16-bit-output residual forms; any other forced gemm_cfg must NOT reach gemm256.hip's own epilogue, which would drop the residual)
    const int cfg16 = (mode == VITG_GEMM_STORE16 && (res || res16) && force != 128 && gemm256_applicable(g, mode)) ? gemm256_pick_cfg(g, mode) : 0;
    const bool res16_on_mi16 = (cfg16 == 6 || (cfg16 == 9 && N >= 384) || (cfg16 == 10 && M >= 384)) && (ldr16 % 4 == 0) && N >= 256;
    // Kernel choice by tile count (measured on the per-rank shapes of the context-parallel LLM, tools/gemm_small_m.py):
    // >= 160 tiles of 256x256 -> 256x256x64 ring-2 (1 block/CU); fewer -> 256x128x32 ring-3 at 2 blocks/CU (twice the tiles,
    // 512 slots) which beat both the 256x256 and the 128x128 kernel there; M < 256 / unaligned cases -> 128x128.
#if VITG_PROFILING_BUILD
    if (Wf && force != 128 && gemm_wf_applicable(g, mode)) rc = launch_gemm_wf(g, dtype, mode, s);
    // gemm_pe.hip (cfg 13: persistent blocks, deferred epilogue; same bits as the other 256-tile kernels) is opt-in: gemm_cfg 13
    // forces it, gemm_flags bit20 puts auto mode on it for 16-bit-output GEMMs with >= 2 tiles per CU (bit21: SwiGLU too).  It was
    // briefly the default for qkv / fc1 (+2.9 % on the step) until its real advantage - saddr DMA addressing and no run-time
    // switches inside an interval - was ported to gemm256.hip; since then the 384x256 tile wins again (721.6 vs 713.9 frames/s,
    // alternating runs; profiles/r01_gemm_shapes.md section 16).
    else if ((tuning().gemm_cfg == 13 ||
              (tuning().gemm_cfg == 0 && (RESEARCH(gemm_flags) & 0x100000) && act != VITG_ACT_GELU_ERF &&
               (mode != VITG_GEMM_SWIGLU || (RESEARCH(gemm_flags) & 0x200000)) && ceil_div(M, 256) * ceil_div(N, 256) >= 512)) &&
             force != 128 && gemm_pe_applicable(g, mode))
        rc = launch_gemm_pe(g, dtype, mode, s);
    else
#endif
    // 16-bit output + residual (vitg_gemm_res16): the 256-wide tiles of gemm_mi16.hip, or - where auto mode would take the 256x128
    // tile or the shape is small / unaligned - the 128-tile kernel; gemm256.hip's own epilogues do not carry these forms
    if (mode == VITG_GEMM_STORE16 && (res || res16) && !res16_on_mi16)
        rc = dtype == DT_BF16 ? launch<bf16_t>(g, mode, s) : launch<f16_t>(g, mode, s);
    else
    if (force != 128 && gemm256_applicable(g, mode)) rc = launch_gemm256(g, dtype, mode, s);
    else rc = dtype == DT_BF16 ? launch<bf16_t>(g, mode, s) : launch<f16_t>(g, mode, s);
    return rc == 0 ? 0 : VITG_ERR_LAUNCH;
}

extern "C" int vitg_gemm(const void* A, int lda, const void* W, int ldw, int M, int N, int K, int dtype,
                         int mode, const float* bias, int act, void* out, int ldo,
                         const float* res, int ldr, int res_mod, void* stream) {
    return gemm_entry(A, lda, W, ldw, nullptr, M, N, K, dtype, mode, bias, act, out, ldo, res, ldr, res_mod, stream);
}

extern "C" int vitg_gemm_res16(const void* A, int lda, const void* W, int ldw, int M, int N, int K, int dtype, const float* bias,
                               void* out, int ldo, const void* res16, int ldr16, const float* res32, int ldr32, int res_mod,
                               void* stream) {
    if (res16 && (ldr16 % 4)) return VITG_ERR_ALIGN;
    if (res32 && (ldr32 % 4)) return VITG_ERR_ALIGN;
    return gemm_entry(A, lda, W, ldw, nullptr, M, N, K, dtype, VITG_GEMM_STORE16, bias, VITG_ACT_NONE, out, ldo, res32, ldr32, res_mod, stream,
                      nullptr, res16, ldr16);
}

#if VITG_PROFILING_BUILD
// Timeline of the 256-tile kernel (gemm256.hip `stamp`): buf holds blocks x 2 wave groups x 4 tiles x 32 slots of u64; nullptr = off.
extern "C" int vitg_gemm_timeline_buffer(void* buf, int blocks) {
    g_stamp_buf = (unsigned long long*)buf; g_stamp_blocks = buf ? blocks : 0;
    return 0;
}
#endif

#if VITG_PROFILING_BUILD
extern "C" int vitg_gemm_wf(const void* A, int lda, const void* W, int ldw, const void* Wf, int M, int N, int K, int dtype,
                            int mode, const float* bias, int act, void* out, int ldo,
                            const float* res, int ldr, int res_mod, void* stream) {
    return gemm_entry(A, lda, W, ldw, Wf, M, N, K, dtype, mode, bias, act, out, ldo, res, ldr, res_mod, stream);
}
#endif

// ---- deterministic split-K for fp32-residual GEMMs with few output tiles ---------------------------------------------------------
namespace vitg {
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, long split_stride, int splits, int ldw,
                                                            const float* __restrict__ bias, const float* __restrict__ res, int ldr,
                                                            float* __restrict__ out, int ldo, int N) {
    const long m = blockIdx.x;
    for (int n = threadIdx.x * 4; n < N; n += 1024) {
        f32x4 v = *(const f32x4*)(ws + m * ldw + n);                       // fixed order: partial 0, 1, 2, ... then bias, then residual
        for (int s = 1; s < splits; ++s) v += *(const f32x4*)(ws + s * split_stride + m * ldw + n);
        if (bias) v += *(const f32x4*)(bias + n);
        if (res) v += *(const f32x4*)(res + m * ldr + n);
        *(f32x4*)(out + m * ldo + n) = v;
    }
}
// 16-bit output form (round 5: the Q and K|V projections of one context-parallel rank - 98 / 28 tiles on 256 CUs): out16 = round16(partials + bias)
template <typename T>
__global__ __launch_bounds__(256) void splitk_reduce16_kernel(const float* __restrict__ ws, long split_stride, int splits, int ldw,
                                                              const float* __restrict__ bias, T* __restrict__ out, int ldo, int N) {
    const long m = blockIdx.x;
    for (int n = threadIdx.x * 4; n < N; n += 1024) {
        f32x4 v = *(const f32x4*)(ws + m * ldw + n);
        for (int s = 1; s < splits; ++s) v += *(const f32x4*)(ws + s * split_stride + m * ldw + n);
        if (bias) v += *(const f32x4*)(bias + n);
        typename T16<T>::vec4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = T16<T>::from_f32(v[e]);
        *(typename T16<T>::vec4*)(out + m * ldo + n) = o;
    }
}
}  // namespace vitg

namespace vitg { int splitk_tile_rows(int M, int N, int K, int splits, int cus); }

extern "C" int vitg_gemm_splitk_plan(int M, int N, int K, int* splits, long* workspace_bytes) {
    if (!splits || !workspace_bytes) return VITG_ERR_ARG;
    const int cus = device_cu_count(current_device());
    // worth it only when one round of tiles leaves most of the chip idle and every split still has a real K loop.  Planned for both tile heights the
    // split-K launch can take (256 x 256, 320 x 256 - gemm256.hip splitk_tile_rows makes the same choice at launch time): the cheaper critical path wins
    auto plan = [&](int bm, double* cost) {
        const long tiles = (long)ceil_div(M, bm) * ceil_div(N, 256);
        int s = 1;
        if (M >= 256 && N >= 256 && N % 8 == 0 && tiles * 2 <= cus && K >= 2048) {
            s = (int)(cus / tiles);
            if (s > 4) s = 4;
            while (s > 1 && K / 64 / s < 16) --s;
        }
        *cost = (double)bm * ceil_div(K / 64, s) * ceil_div((int)(tiles * s), cus);
        return s;
    };
    double c256, c320;
    int sp = plan(256, &c256);
    if (sp > 1 && M > 256) {
        const int sp320 = plan(320, &c320);
        if (sp320 > 1 && c320 < 0.97 * c256 && vitg::splitk_tile_rows(M, N, K, sp320, cus) == 320) sp = sp320;
    }
    *splits = sp;
    *workspace_bytes = sp > 1 ? (long)sp * M * N * 4 : 0;
    return 0;
}

extern "C" int vitg_gemm_splitk(const void* A, int lda, const void* W, int ldw, int M, int N, int K, int dtype, const float* bias,
                                float* out, int ldo, const float* res, int ldr, float* workspace, int splits, void* stream) {
    if (M <= 0 || N <= 0) return 0;
    if (splits < 2 || !workspace) return VITG_ERR_ARG;
    if (N % 4 || ldo % 4 || (res && ldr % 4)) return VITG_ERR_ALIGN;
    if (K / BK < splits) return VITG_ERR_SHAPE;
    set_attrs();
    GemmArgs g = {};
    g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.M = M; g.N = N; g.K = K;
    g.bias = nullptr; g.act = VITG_ACT_NONE; g.out = workspace; g.ldo = N; g.res = nullptr; g.ldr = 0; g.res_mod = 0;
    g.tiles_m = ceil_div(M, BM); g.tiles_n = ceil_div(N, BN); g.stagger_ns = 0; g.group_m = 4; g.Wf = nullptr; g.flags = 0;
    g.split_k = splits; g.split_stride = (long)M * N;
    if (K <= 0 || K % BK != 0) return VITG_ERR_SHAPE;
    if ((lda % 8) || (ldw % 8)) return VITG_ERR_ALIGN;
    if (dtype != DT_BF16 && dtype != DT_F16) return VITG_ERR_DTYPE;
    if (!gemm256_applicable(g, VITG_GEMM_F32)) return VITG_ERR_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    if (launch_gemm256(g, dtype, VITG_GEMM_F32, s) != 0) return VITG_ERR_LAUNCH;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)M), dim3(256), 0, s, workspace, g.split_stride, splits, N, bias, res, ldr, out, ldo, N);
    return hipGetLastError() == hipSuccess ? 0 : VITG_ERR_LAUNCH;
}

// out16[M, N] = round16(A W^T + bias) through `splits` fp32 K-partials in `workspace` (>= splits * M * N floats), fixed summation order: the split-K form of
// vitg_gemm(mode = VITG_GEMM_STORE16, act = none) for few-tile shapes with a long K (plan: vitg_gemm_splitk_plan).
extern "C" int vitg_gemm_splitk16(const void* A, int lda, const void* W, int ldw, int M, int N, int K, int dtype, const float* bias, void* out,
                                  int ldo, float* workspace, int splits, void* stream) {
    if (M <= 0 || N <= 0) return 0;
    if (splits < 2 || !workspace || !out) return VITG_ERR_ARG;
    if (N % 4 || ldo % 4 || ldo < N || ((uintptr_t)out & 7)) return VITG_ERR_ALIGN;
    if (K <= 0 || K % BK != 0 || K / BK < splits) return VITG_ERR_SHAPE;
    if ((lda % 8) || (ldw % 8)) return VITG_ERR_ALIGN;
    if (dtype != DT_BF16 && dtype != DT_F16) return VITG_ERR_DTYPE;
    set_attrs();
    GemmArgs g = {};
    g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.M = M; g.N = N; g.K = K;
    g.bias = nullptr; g.act = VITG_ACT_NONE; g.out = workspace; g.ldo = N; g.res = nullptr; g.ldr = 0; g.res_mod = 0;
    g.tiles_m = ceil_div(M, BM); g.tiles_n = ceil_div(N, BN); g.stagger_ns = 0; g.group_m = 4; g.Wf = nullptr; g.flags = 0;
    g.split_k = splits; g.split_stride = (long)M * N;
    if (!gemm256_applicable(g, VITG_GEMM_F32)) return VITG_ERR_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    if (launch_gemm256(g, dtype, VITG_GEMM_F32, s) != 0) return VITG_ERR_LAUNCH;
    if (dtype == DT_BF16) hipLaunchKernelGGL(splitk_reduce16_kernel<bf16_t>, dim3((unsigned)M), dim3(256), 0, s, workspace, g.split_stride, splits, N, bias, (bf16_t*)out, ldo, N);
    else hipLaunchKernelGGL(splitk_reduce16_kernel<f16_t>, dim3((unsigned)M), dim3(256), 0, s, workspace, g.split_stride, splits, N, bias, (f16_t*)out, ldo, N);
    return hipGetLastError() == hipSuccess ? 0 : VITG_ERR_LAUNCH;
}

#if VITG_PROFILING_BUILD   // (lost its A/B against split-K x2: profiling library only since round 4)
// ---- stream-K for fp32-residual GEMMs with fewer output tiles than CUs (round 3; kernel side: gemm_mi16.hip, SK) -----------------------------
namespace vitg {
// out[m, n] = sum over the segments of tile (m / 256, n / 256), in block order, + bias[n] + res[m, n].  Tile t owns K-tile units [t KT, (t+1) KT) of the
// tile-major unit line; block b holds units [b U, (b+1) U): its slot 0 is the tile it starts in, slot 1 the next one.
__global__ __launch_bounds__(256) void streamk_reduce_kernel(const float* __restrict__ ws, int tiles_m, int KT, int U, const float* __restrict__ bias,
                                                             const float* __restrict__ res, int ldr, float* __restrict__ out, int ldo, int M, int N) {
    const int t = blockIdx.x >> 2, rq = blockIdx.x & 3;            // 4 blocks per tile, 64 rows each; a wave per row, 4 columns per lane
    const int tm = t % tiles_m, tn = t / tiles_m;
    const long u_lo = (long)t * KT, u_hi = u_lo + KT - 1;
    const int b0 = (int)(u_lo / U), b1 = (int)(u_hi / U);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int n = tn * 256 + lane * 4;
    for (int r = rq * 64 + wv; r < rq * 64 + 64; r += 4) {
        const int m = tm * 256 + r;
        if (m >= M || n >= N) continue;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        for (int b = b0; b <= b1; ++b) {
            const int slot = ((long)b * U / KT == t) ? 0 : 1;
            v += *(const f32x4*)(ws + ((size_t)b * 2 + slot) * 65536 + (size_t)r * 256 + lane * 4);
        }
        if (bias) v += *(const f32x4*)(bias + n);
        if (res) v += *(const f32x4*)(res + (size_t)m * ldr + n);
        *(f32x4*)(out + (size_t)m * ldo + n) = v;
    }
}
}  // namespace vitg

// U = K-tile units per block, blocks = ceil(total / U); 0 blocks = stream-K does not apply to this shape
static void streamk_partition(int M, int N, int K, int* U, int* blocks) {
    *U = 0; *blocks = 0;
    if (M < 256 || N < 256 || K <= 0 || K % 64 || N % 4) return;
    const int cus = device_cu_count(current_device());
    const long tiles = (long)ceil_div(M, 256) * ceil_div(N, 256);
    const int KT = K / 64;
    // fewer tiles than three quarters of the CUs, and enough K per tile that a block's run (<= one tile's worth: at most two segments) is long
    if (tiles * 4 > (long)cus * 3 || KT < 32) return;
    const long total = tiles * KT;
    int u = (int)((total + cus - 1) / cus);
    if (u < 16) u = 16;
    if (u > KT) return;
    *U = u;
    *blocks = (int)((total + u - 1) / u);
}

extern "C" int vitg_gemm_streamk_plan(int M, int N, int K, int* blocks, long* workspace_bytes) {
    if (!blocks || !workspace_bytes) return VITG_ERR_ARG;
    int U = 0;
    streamk_partition(M, N, K, &U, blocks);
    *workspace_bytes = (long)*blocks * 2 * 65536 * 4;
    return 0;
}

extern "C" int vitg_gemm_streamk(const void* A, int lda, const void* W, int ldw, int M, int N, int K, int dtype, const float* bias, float* out,
                                 int ldo, const float* res, int ldr, float* workspace, int blocks, void* stream) {
    if (M <= 0 || N <= 0) return 0;
    if (blocks < 1 || !workspace) return VITG_ERR_ARG;
    if (N % 4 || ldo % 4 || (res && ldr % 4) || (lda % 8) || (ldw % 8)) return VITG_ERR_ALIGN;
    if (K <= 0 || K % 64 != 0) return VITG_ERR_SHAPE;
    if (dtype != DT_BF16 && dtype != DT_F16) return VITG_ERR_DTYPE;
    int U = 0, want = 0;
    streamk_partition(M, N, K, &U, &want);
    if (want != blocks) return VITG_ERR_ARG;                       // the partition is the plan's: blocks and workspace size come from it
    GemmArgs g = {};
    g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.M = M; g.N = N; g.K = K;
    g.act = VITG_ACT_NONE; g.out = workspace; g.ldo = 256; g.split_k = 1;
    g.sk_kt = K / 64; g.sk_units = U;
    const long tiles = (long)ceil_div(M, 256) * ceil_div(N, 256);
    hipStream_t s = (hipStream_t)stream;
    if (launch_gemm_mi16_streamk(g, dtype, blocks, s) != 0) return VITG_ERR_LAUNCH;
    hipLaunchKernelGGL(streamk_reduce_kernel, dim3((unsigned)(tiles * 4)), dim3(256), 0, s, workspace, ceil_div(M, 256), g.sk_kt, U, bias, res, ldr,
                       out, ldo, M, N);
    return hipGetLastError() == hipSuccess ? 0 : VITG_ERR_LAUNCH;
}
#endif   // stream-K

#if VITG_PROFILING_BUILD
// ---- LayerNorm / RMSNorm folded into the GEMMs on either side of it (round 2; measured slower than the stand-alone norm pass, kept
//      in the profiling library: include/videoitg_hip_profiling.h, profiles/r02_fused_norm.md) ------------------------------------------
namespace vitg {
// one thread per row: Chan-combine the P equal-sized partials (mean_p, M2_p) -> (mean, rstd); RMSNorm: (0, rsqrt(E[x^2] + eps))
__global__ __launch_bounds__(256) void norm_stats_finalize_kernel(const float* __restrict__ part, long M, int P, int cols, float eps, int rms,
                                                                  float* __restrict__ stats) {
    const long m = (long)blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    const f32x2* p = (const f32x2*)(part + (size_t)m * P * 2);
    float sm = 0.f;
    for (int i = 0; i < P; ++i) sm += p[i][0];
    const float mean = sm / P;
    float m2 = 0.f;
    for (int i = 0; i < P; ++i) { const float d = p[i][0] - mean; m2 += p[i][1] + cols * d * d; }
    const float D = (float)P * cols;
    f32x2 o;
    if (rms) { o[0] = 0.f; o[1] = rsqrtf((m2 + D * mean * mean) / D + eps); }
    else { o[0] = mean; o[1] = rsqrtf(m2 / D + eps); }
    *(f32x2*)(stats + 2 * m) = o;
}
}  // namespace vitg

extern "C" int vitg_gemm_norm_plan(int M, int N, int K, int* cols_per_partial) {
    if (!cols_per_partial) return VITG_ERR_ARG;
    GemmArgs g = {};
    g.M = M; g.N = N; g.K = K; g.ldo = N; g.ldr = N; g.res = (const float*)1;
    int cols = (K > 0 && K % 64 == 0) ? gemm256_norm_cols(g) : 0;
    if (cols && N % cols) cols = 0;                      // every partial must cover `cols` real columns
    *cols_per_partial = cols;
    return 0;
}

extern "C" int vitg_gemm_norm_producer(const void* A, int lda, const void* W, int ldw, int M, int N, int K, int dtype, const float* bias,
                                       float* out, int ldo, const float* res, int ldr, int res_mod, const float* next_gamma, void* xh,
                                       int ldxh, float* partials, void* stream) {
    if (!res || !next_gamma || !xh || !partials) return VITG_ERR_ARG;
    if (ldxh % 8) return VITG_ERR_ALIGN;
    int cols = 0;
    (void)vitg_gemm_norm_plan(M, N, K, &cols);
    if (!cols) return VITG_ERR_SHAPE;
    NormFusion nf;
    nf.gamma = next_gamma; nf.xh = xh; nf.ldxh = ldxh; nf.part = partials; nf.P = N / cols;
    return gemm_entry(A, lda, W, ldw, nullptr, M, N, K, dtype, VITG_GEMM_F32, bias, VITG_ACT_NONE, out, ldo, res, ldr, res_mod, stream, &nf);
}

extern "C" int vitg_norm_stats_finalize(const float* partials, long M, int P, int cols, float eps, int rms, float* stats, void* stream) {
    if (M <= 0) return 0;
    if (!partials || !stats || P <= 0 || cols <= 0) return VITG_ERR_ARG;
    hipLaunchKernelGGL(norm_stats_finalize_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, (hipStream_t)stream, partials, M, P, cols,
                       eps, rms, stats);
    return hipGetLastError() == hipSuccess ? 0 : VITG_ERR_LAUNCH;
}

extern "C" int vitg_gemm_norm_consumer(const void* XH, int lda, const void* W, int ldw, int M, int N, int K, int dtype, int mode,
                                       const float* bias, int act, void* out, int ldo, const float* stats, const float* colsum,
                                       void* stream) {
    if (!stats) return VITG_ERR_ARG;
    if (mode != VITG_GEMM_STORE16 && mode != VITG_GEMM_SWIGLU) return VITG_ERR_ARG;
    if (mode == VITG_GEMM_SWIGLU && colsum) return VITG_ERR_ARG;       // RMSNorm only in front of the SwiGLU GEMM
    NormFusion nf;
    nf.stats = stats; nf.colsum = colsum;
    return gemm_entry(XH, lda, W, ldw, nullptr, M, N, K, dtype, mode, bias, act, out, ldo, nullptr, 0, 0, stream, &nf);
}
#endif   // VITG_PROFILING_BUILD (fused norm)

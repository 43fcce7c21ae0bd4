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
#include "common.h"
#include "../../include/videoitg_hip.h"

namespace vitg {

struct AttnArgs {
    const void* q; const void* k; const void* v; void* o;
    long q_bs, k_bs, v_bs, o_bs;     // batch strides (elements)
    int q_rs, k_rs, v_rs, o_rs;      // row (token) strides (elements)
    int q_hs, k_hs, v_hs, o_hs;      // head strides (elements)
    int S;                           // number of queries
    int Sk;                          // number of keys (== S unless the keys were all-gathered from other ranks)
    int group;                       // q heads per kv head
    float scale_log2e;               // softmax scale * log2(e)
};

constexpr int ATT_BQ = 128, ATT_BK = 64, ATT_THREADS = 256;

template <int D> struct AttnCfg;
template <> struct AttnCfg<128> { static constexpr int DK = 128, DV = 128, LDK = 128, LDV = 128, CH = 16; };
// K rows padded to 80 (zero d 72..79), V rows padded to 96 (pad never stored to the output)
template <> struct AttnCfg<72> { static constexpr int DK = 80, DV = 96, LDK = 80, LDV = 96, CH = 9; };

template <int D> __device__ __forceinline__ int k_lds_off(int row, int chunk) {   // bytes
    if constexpr (D == 128) return row * 256 + ((chunk ^ (row & 15)) << 4);
    else return row * 160 + (chunk << 4);
}

template <typename T, int D, bool TR>
__global__ __launch_bounds__(ATT_THREADS, 2) void attn_kernel(AttnArgs a) {
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
    const int head = blockIdx.y, b = blockIdx.z;
    const int kvh = head / a.group;
    const int q0 = blockIdx.x * ATT_BQ + wave * 32;

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
    auto load_kv = [&](int kt) {
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int c = tid + i * ATT_THREADS;
            if (c < NCH) {
                const int row = c / C::CH, ch = c % C::CH;
                const int key = min(kt * ATT_BK + row, a.Sk - 1);
                rk[i] = *(const u32x4*)(K + (long)key * a.k_rs + ch * 8);
                rv[i] = *(const u32x4*)(V + (long)key * a.v_rs + ch * 8);
            }
        }
    };
    auto store_kv = [&](int buf) {
        char* kb = smem + buf * (KT_BYTES + VT_BYTES);
        char* vb = kb + KT_BYTES;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int c = tid + i * ATT_THREADS;
            if (c < NCH) {
                const int row = c / C::CH, ch = c % C::CH;
                *(u32x4*)(kb + k_lds_off<D>(row, ch)) = rk[i];
                *(u32x4*)(vb + row * (C::LDV * 2) + (ch << 4)) = rv[i];
            }
        }
        if constexpr (D == 72) {
            // zero the K pad chunk (d 72..79) once per buffer fill: 64 rows
            if (tid < ATT_BK) {
                u32x4 z = {0u, 0u, 0u, 0u};
                *(u32x4*)(kb + k_lds_off<D>(tid, 9)) = z;
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
    load_kv(0);
    store_kv(0);
    __syncthreads();

    for (int kt = 0; kt < nkt; ++kt) {
        const bool more = kt + 1 < nkt;
        if (more) load_kv(kt + 1);
        const char* kb = smem + (kt & 1) * (KT_BYTES + VT_BYTES);
        const char* vb = kb + KT_BYTES;

        // ---- S^T = K Q^T : two 32-key tiles ------------------------------------------------------------
        f32x16 s[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[t][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const vec8 kf = *(const vec8*)(kb + k_lds_off<D>(t * 32 + ql, ks * 2 + h));
                s[t] = T16<T>::mfma32(kf, qf[ks], s[t]);
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
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        m_run = m_new;
        float psum = 0.f;
        vec8 pf[4];                      // k-step kk = 2*t + u uses s[t][8u .. 8u+7]
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = __builtin_amdgcn_exp2f(fmaf(s[t][r], a.scale_log2e, -m_new));
                psum += p;
                pf[t * 2 + (r >> 3)][r & 7] = T16<T>::from_f32(p);
            }
        l_run = l_run * alpha + psum;
#pragma unroll
        for (int i = 0; i < DT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[i][r] *= alpha;

        // ---- O^T += V^T P^T ----------------------------------------------------------------------------
        // k-step kk (16 keys: tile t = kk>>1, half u = kk&1): slot (h, j) <-> key t*32 + u*16 + 8*(j>>2) + 4*h + (j&3)
        // A operand lane (d = dt*32 + ql, h) needs V[key(h, j)][d] for j = 0..7: two groups of 4 consecutive keys.
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int kbase = (kk >> 1) * 32 + (kk & 1) * 16 + 4 * h;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                vec8 vf;
                if constexpr (TR) {
                    // 16-lane group g = lane>>4: column block 16*(g&1); lane p = lane&15 supplies the address of
                    // row (p>>2), columns 4*(p&3)..+3 of the 4x16 block; result: lane gets column (lane&15), rows 0..3.
                    const int p = lane & 15, cb = ((lane >> 4) & 1) * 16;
                    const int col = dt * 32 + cb + 4 * (p & 3);
                    const char* a0 = vb + (kbase + (p >> 2)) * (C::LDV * 2) + col * 2;
                    const char* a1 = a0 + 8 * (C::LDV * 2);
                    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(a0));
                    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(a1));
                    s16x8 both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    vf = bitcast<vec8>(both);
                } else {
                    const T* vt = (const T*)vb;
                    const int d = dt * 32 + ql;
#pragma unroll
                    for (int j = 0; j < 8; ++j) vf[j] = vt[(kbase + 8 * (j >> 2) + (j & 3)) * C::LDV + d];
                }
                o[dt] = T16<T>::mfma32(vf, pf[kk], o[dt]);
            }
        }

        if (more) store_kv((kt + 1) & 1);
        __syncthreads();
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

template <typename T, int D, bool TR>
static int launch_attn(const AttnArgs& a, int batch, int heads, hipStream_t s) {
    typedef AttnCfg<D> C;
    const size_t lds = 2 * (ATT_BK * C::LDK * 2 + ATT_BK * C::LDV * 2);
    static bool done = false;
    if (!done) {
        hipFuncSetAttribute((const void*)attn_kernel<T, D, TR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        done = true;
    }
    dim3 grid(ceil_div(a.S, ATT_BQ), heads, batch);
    hipLaunchKernelGGL((attn_kernel<T, D, TR>), grid, dim3(ATT_THREADS), lds, s, a);
    return (int)hipGetLastError();
}

}  // namespace vitg

using namespace vitg;

extern "C" int vitg_attention(const void* q, const void* k, const void* v, void* o, int batch, int heads, int kv_heads,
                              int S, int Sk, int head_dim, long q_bs, long k_bs, long v_bs, long o_bs, int q_rs, int k_rs,
                              int v_rs, int o_rs, int q_hs, int k_hs, int v_hs, int o_hs, float scale, int dtype,
                              int variant, void* stream) {
    if (batch <= 0 || heads <= 0 || S <= 0) return 0;
    if (Sk <= 0) return VITG_ERR_SHAPE;
    if (kv_heads <= 0 || heads % kv_heads) return VITG_ERR_SHAPE;
    if (head_dim != 72 && head_dim != 128) return VITG_ERR_SHAPE;
    if ((q_rs | k_rs | v_rs | o_rs | q_hs | k_hs | v_hs | o_hs) % 4) return VITG_ERR_ALIGN;
    if ((q_rs | k_rs | v_rs | q_hs | k_hs | v_hs) % 8 || (q_bs | k_bs | v_bs) % 8) return VITG_ERR_ALIGN;
    if (dtype != DT_BF16 && dtype != DT_F16) return VITG_ERR_DTYPE;
    AttnArgs a;
    a.q = q; a.k = k; a.v = v; a.o = o;
    a.q_bs = q_bs; a.k_bs = k_bs; a.v_bs = v_bs; a.o_bs = o_bs;
    a.q_rs = q_rs; a.k_rs = k_rs; a.v_rs = v_rs; a.o_rs = o_rs;
    a.q_hs = q_hs; a.k_hs = k_hs; a.v_hs = v_hs; a.o_hs = o_hs;
    a.S = S; a.Sk = Sk; a.group = heads / kv_heads;
    a.scale_log2e = scale * 1.4426950408889634f;
    hipStream_t s = (hipStream_t)stream;
    int rc;
    const bool tr = variant != 1;     // variant 1 = scalar LDS reads of V (cross-check path for the tr-read)
#define GO(T, D) (tr ? launch_attn<T, D, true>(a, batch, heads, s) : launch_attn<T, D, false>(a, batch, heads, s))
    if (dtype == DT_BF16) rc = head_dim == 72 ? GO(bf16_t, 72) : GO(bf16_t, 128);
    else rc = head_dim == 72 ? GO(f16_t, 72) : GO(f16_t, 128);
#undef GO
    return rc == 0 ? 0 : VITG_ERR_LAUNCH;
}

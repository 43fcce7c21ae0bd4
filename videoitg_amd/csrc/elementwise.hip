// HBM-bound kernels of the selector forward: patchify (im2col), LayerNorm, RMSNorm, RoPE, bilinear token
// pool, embedding gather, per-frame mean-pool + score head, sigmoid + stable descending rank.
// All loads/stores are 8-16 B per lane; reductions are wave-level (64 lanes) with fp32 statistics.
#include "common.h"
#include <stdlib.h>
#include "../../include/videoitg_hip.h"

namespace vitg {

template <typename T> __device__ __forceinline__ float ld_as_f32(const void* p, long i);
template <> __device__ __forceinline__ float ld_as_f32<float>(const void* p, long i) { return ((const float*)p)[i]; }
template <> __device__ __forceinline__ float ld_as_f32<bf16_t>(const void* p, long i) { return (float)((const bf16_t*)p)[i]; }
template <> __device__ __forceinline__ float ld_as_f32<f16_t>(const void* p, long i) { return (float)((const f16_t*)p)[i]; }

// ------------------------------------------------------------------------------------------------
// patchify: pixel_values [T,3,R,R] (NCHW) -> A[T*G*G, Kpad], k = c*ps*ps + ky*ps + kx (the order of the
// conv weight [Dv,3,ps,ps] flattened), zero-filled k >= 3*ps*ps.  One wave per output row.
// ------------------------------------------------------------------------------------------------
template <typename TI, typename TO>
__global__ void patchify_kernel(const void* __restrict__ pix, TO* __restrict__ out, int T, int R, int ps, int G,
                                int Kpad, long rows) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int P = G * G;
    const int t = (int)(row / P), pi = (int)(row % P);
    const int py = pi / G, px = pi % G;
    const int KK = 3 * ps * ps;
    const long img = (long)t * 3 * R * R;
    for (int k = lane; k < Kpad; k += 64) {
        float v = 0.f;
        if (k < KK) {
            const int c = k / (ps * ps), rem = k % (ps * ps);
            const int ky = rem / ps, kx = rem % ps;
            v = ld_as_f32<TI>(pix, img + ((long)c * R + (py * ps + ky)) * R + (px * ps + kx));
        }
        out[row * Kpad + k] = T16<TO>::from_f32(v);
    }
}

// Coalesced form (round 3; R % 8 == 0): one block per (frame, patch row).  The 3 x ps pixel rows the patch row needs are read as whole
// rows (16-B loads, consecutive lanes on consecutive addresses: full lines instead of the 28-B segments one patch row offers) into LDS as
// 16-bit [c][ky][x]; the G im2col rows leave as 16-B stores, consecutive lanes on consecutive chunks of one output row.  HBM traffic is the
// algorithmic minimum: every pixel read once, every operand element written once.
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void patchify_rows_kernel(const TI* __restrict__ pix, TO* __restrict__ out, int R, int ps, int G, int Kpad) {
    extern __shared__ __attribute__((aligned(16))) char psm[];
    TO* lds = (TO*)psm;                                    // [3 * ps][R]
    const int t = blockIdx.x / G, py = blockIdx.x % G;
    const int rows = 3 * ps, cpr = R / 8;                  // 8-element chunks per pixel row
#pragma unroll 4                                          // four rows of loads in flight per thread (the loop is load-latency bound otherwise)
    for (int i = threadIdx.x; i < rows * cpr; i += blockDim.x) {
        const int rr = i / cpr, ch = i % cpr;
        const int c = rr / ps, ky = rr % ps;
        const TI* src = pix + (((long)t * 3 + c) * R + (py * ps + ky)) * R + ch * 8;
        typename T16<TO>::vec8 v;
        if constexpr (sizeof(TI) == 4) {
            const f32x4 a = *(const f32x4*)src, b = *(const f32x4*)(src + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] = T16<TO>::from_f32(a[e]); v[e + 4] = T16<TO>::from_f32(b[e]); }
        } else {
            const typename T16<TI>::vec8 a = *(const typename T16<TI>::vec8*)src;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = T16<TO>::from_f32(T16<TI>::to_f32(a[e]));
        }
        *(typename T16<TO>::vec8*)(lds + rr * R + ch * 8) = v;
    }
    __syncthreads();
    const int KK = 3 * ps * ps, kc = Kpad / 8;             // 8-element chunks per output row
    TO* orow = out + ((long)t * G * G + (long)py * G) * Kpad;
    for (int i = threadIdx.x; i < G * kc; i += blockDim.x) {
        const int px = i / kc, k0 = (i % kc) * 8;
        typename T16<TO>::vec8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = k0 + e;
            TO x = T16<TO>::from_f32(0.f);
            if (k < KK) {
                const int c = k / (ps * ps), rem = k % (ps * ps);
                x = lds[(c * ps + rem / ps) * R + px * ps + rem % ps];
            }
            v[e] = x;
        }
        *(typename T16<TO>::vec8*)(orow + (long)px * Kpad + k0) = v;
    }
}

// ------------------------------------------------------------------------------------------------
// LayerNorm / RMSNorm over fp32 rows -> 16-bit (or fp32) rows with row stride ldo >= D; columns D..ldo-1
// are written as zero so the result can feed a GEMM whose K is padded to a multiple of 64.
// One wave per row, row cached in registers (D <= 64*4*NV).
// ------------------------------------------------------------------------------------------------
template <typename TO, bool RMS, int NV, bool PF>
__global__ void norm_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ w, const float* __restrict__ bia,
                            void* __restrict__ out, int ldo, int out_f32, long M, int D, float eps) {
    // One wave per row, rows grid-strided over all waves of the launch; the NEXT row's loads are issued before the current
    // row's reductions so that every wave always has 2 rows of loads in flight (HBM latency hidden without relying on
    // occupancy alone).
    const int lane = threadIdx.x & 63;
    const long nwaves = (long)gridDim.x * (blockDim.x >> 6);
    long row = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= M) return;
    // PF = false (every wave has exactly one row): no second register set, so occupancy stays high for short launches
    f32x4 v[NV], nx[PF ? NV : 1];
    auto load_row = [&](auto& dst, long r) {
        const float* xr = x + r * ldx;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * 64 + lane) * 4;
            if (c < D) dst[i] = *(const f32x4*)(xr + c);
            else dst[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    load_row(v, row);
    for (; row < M; row += nwaves) {
        const long next = row + nwaves;
        if constexpr (PF) { if (next < M) load_row(nx, next); }
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i)
            s += RMS ? (v[i][0] * v[i][0] + v[i][1] * v[i][1] + v[i][2] * v[i][2] + v[i][3] * v[i][3])
                     : (v[i][0] + v[i][1] + v[i][2] + v[i][3]);
        s = wave_sum(s);
        float mean = 0.f, rstd;
        if (RMS) {
            rstd = rsqrtf(s / D + eps);
        } else {
            mean = s / D;
            float q = 0.f;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = (i * 64 + lane) * 4;
                if (c < D) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { const float d = v[i][e] - mean; q += d * d; }
                }
            }
            q = wave_sum(q);
            rstd = rsqrtf(q / D + eps);
        }
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * 64 + lane) * 4;
            if (c >= ldo) continue;
            f32x4 r = {0.f, 0.f, 0.f, 0.f};
            if (c < D) {
                const f32x4 ww = *(const f32x4*)(w + c);
                r = (v[i] - mean) * rstd * ww;
                if (!RMS) r += *(const f32x4*)(bia + c);
            }
            if (out_f32) *(f32x4*)((float*)out + row * ldo + c) = r;
            else {
                typename T16<TO>::vec4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = T16<TO>::from_f32(r[e]);
                *(typename T16<TO>::vec4*)((TO*)out + row * ldo + c) = o;
            }
        }
        if constexpr (PF) {
#pragma unroll
            for (int i = 0; i < NV; ++i) v[i] = nx[i];
        }
    }
}

// LayerNorm over 16-bit rows (the tower's 16-bit residual stream, round 3: the reference's own arithmetic - model.half(),
// /root/reference/eagle/model/builder.py:59) -> 16-bit rows of the same type: 2 B read + 2 B written per element instead of 4 + 2.  One wave per row, 8 elements (16 B) per lane and chunk, row in registers, fp32 statistics, next row prefetched.
template <typename T, int NV, bool PF>
__global__ void ln16_kernel(const T* __restrict__ x, int ldx, const float* __restrict__ w, const float* __restrict__ bia,
                            T* __restrict__ out, int ldo, long M, int D, float eps) {
    typedef typename T16<T>::vec8 vec8;
    const int lane = threadIdx.x & 63;
    const long nwaves = (long)gridDim.x * (blockDim.x >> 6);
    long row = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= M) return;
    vec8 v[NV], nx[PF ? NV : 1];
    auto load_row = [&](auto& dst, long r) {
        const T* xr = x + r * ldx;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * 64 + lane) * 8;
            if (c < D) dst[i] = *(const vec8*)(xr + c);
            else dst[i] = vec8{};
        }
    };
    load_row(v, row);
    for (; row < M; row += nwaves) {
        const long next = row + nwaves;
        if constexpr (PF) { if (next < M) load_row(nx, next); }
        float f[NV][8];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
            for (int e = 0; e < 8; ++e) { f[i][e] = T16<T>::to_f32(v[i][e]); s += f[i][e]; }
        s = wave_sum(s);
        const float mean = s / D;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * 64 + lane) * 8;
            if (c < D) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float d = f[i][e] - mean; q += d * d; }
            }
        }
        q = wave_sum(q);
        const float rstd = rsqrtf(q / D + eps);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * 64 + lane) * 8;
            if (c >= ldo) continue;
            vec8 o = vec8{};
            if (c < D) {
                const f32x4 w0 = *(const f32x4*)(w + c), w1 = *(const f32x4*)(w + c + 4);
                const f32x4 b0 = *(const f32x4*)(bia + c), b1 = *(const f32x4*)(bia + c + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    o[e] = T16<T>::from_f32((f[i][e] - mean) * rstd * w0[e] + b0[e]);
                    o[e + 4] = T16<T>::from_f32((f[i][e + 4] - mean) * rstd * w1[e] + b1[e]);
                }
            }
            *(vec8*)(out + row * ldo + c) = o;
        }
        if constexpr (PF) {
#pragma unroll
            for (int i = 0; i < NV; ++i) v[i] = nx[i];
        }
    }
}

// Round 4: FOUR rows per wave, 16 lanes per row, for D = 128 NC exactly (the tower's 1152 = 9 x 128).  Every load / store instruction moves 4 x 256 B
// with all 64 lanes (the one-row-per-wave form above runs its third 1-KiB chunk with 16 of 64 lanes) and the row reductions stay inside a DPP row of
// 16 lanes (4 rotate-adds instead of 6 cross-lane steps); 9 KiB per wave in flight (+ the prefetched next four rows).  Same arithmetic as ln16_kernel
// (fp32 two-pass statistics, one rounding of the result); the sums run in another order.  Measured (tools/archive/r04_calls.sh.txt (call18), gpurun_out/r04_c18): stand-alone
// 465-486 us against 520-545 (5.0-5.2 against 4.4-4.65 TB/s; a torch copy of the same bytes on that box: 443 us = 5.46 TB/s), inside the step 0.443-0.451
// against 0.502-0.509 ms per launch, +0.45 % frames/s in three alternations.
__device__ __forceinline__ float row16_sum(float v) {
#define ROR(n) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + (n), 0xf, 0xf, false))
    ROR(8); ROR(4); ROR(2); ROR(1);
#undef ROR
    return v;
}
template <typename T, int NC, bool PF>
__global__ __launch_bounds__(256) void ln16r4_kernel(const T* __restrict__ x, int ldx, const float* __restrict__ w, const float* __restrict__ bia,
                              T* __restrict__ out, int ldo, long M, int D, float eps) {
    typedef typename T16<T>::vec8 vec8;
    const int lane = threadIdx.x & 63, sub = lane & 15, rsel = lane >> 4;
    const long nq = (long)gridDim.x * (blockDim.x >> 6);
    long q = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long MQ = (M + 3) >> 2;
    if (q >= MQ) return;
    vec8 v[NC], nx[PF ? NC : 1];
    auto load_rows = [&](auto& dst, long qq) {
        const long r = min(qq * 4 + rsel, M - 1);
        const T* xr = x + r * ldx + sub * 8;
#pragma unroll
        for (int i = 0; i < NC; ++i) dst[i] = *(const vec8*)(xr + i * 128);
    };
    load_rows(v, q);
    const float invD = 1.0f / D;
    for (; q < MQ; q += nq) {
        const long next = q + nq;
        if constexpr (PF) { if (next < MQ) load_rows(nx, next); }
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NC; ++i)
#pragma unroll
            for (int e = 0; e < 8; ++e) s += T16<T>::to_f32(v[i][e]);
        const float mean = row16_sum(s) * invD;
        float qq = 0.f;
#pragma unroll
        for (int i = 0; i < NC; ++i)
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = T16<T>::to_f32(v[i][e]) - mean; qq += d * d; }
        const float rstd = rsqrtf(row16_sum(qq) * invD + eps);
        const long row = q * 4 + rsel;
        if (row < M) {
            T* orow = out + row * ldo + sub * 8;
#pragma unroll
            for (int i = 0; i < NC; ++i) {
                const int c = i * 128 + sub * 8;
                const f32x4 w0 = *(const f32x4*)(w + c), w1 = *(const f32x4*)(w + c + 4);
                const f32x4 b0 = *(const f32x4*)(bia + c), b1 = *(const f32x4*)(bia + c + 4);
                vec8 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    o[e] = T16<T>::from_f32((T16<T>::to_f32(v[i][e]) - mean) * rstd * w0[e] + b0[e]);
                    o[e + 4] = T16<T>::from_f32((T16<T>::to_f32(v[i][e + 4]) - mean) * rstd * w1[e] + b1[e]);
                }
                *(vec8*)(orow + i * 128) = o;
            }
            for (int c = D + sub * 8; c < ldo; c += 128) *(vec8*)(out + row * ldo + c) = vec8{};
        }
        if constexpr (PF) {
#pragma unroll
            for (int i = 0; i < NC; ++i) v[i] = nx[i];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// RoPE (rotate_half layout) in place on the q and k heads of a packed qkv row:
//   x[i] <- x[i] cos_i - x[i+hd/2] sin_i ;  x[i+hd/2] <- x[i+hd/2] cos_i + x[i] sin_i
// cos/sin tables fp32 [N, hd/2] (built on the host in fp32 like Qwen2RotaryEmbedding).
// One thread handles 4 consecutive i of one (token, head).
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void rope_kernel(T* __restrict__ qkv, int ld, const float* __restrict__ cs, const float* __restrict__ sn,
                            long N, int heads, int hd) {
    const int half = hd >> 1, per = half >> 2;           // threads per head
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = N * heads * per;
    if (gid >= total) return;
    const int j = (int)(gid % per);
    const int hh = (int)((gid / per) % heads);
    const long n = gid / ((long)per * heads);
    T* p = qkv + n * ld + (long)hh * hd + j * 4;
    typedef typename T16<T>::vec4 v4;
    v4 a = *(v4*)p, b = *(v4*)(p + half);
    const f32x4 c = *(const f32x4*)(cs + n * half + j * 4), s = *(const f32x4*)(sn + n * half + j * 4);
    v4 ra, rb;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float x1 = T16<T>::to_f32(a[e]), x2 = T16<T>::to_f32(b[e]);
        ra[e] = T16<T>::from_f32(x1 * c[e] - x2 * s[e]);
        rb[e] = T16<T>::from_f32(x2 * c[e] + x1 * s[e]);
    }
    *(v4*)p = ra;
    *(v4*)(p + half) = rb;
}

// ------------------------------------------------------------------------------------------------
// Bilinear token pool G x G -> HW x HW (F.interpolate bilinear, align_corners=False, no antialias) over
// fp32 tower features [T, G*G, D] -> 16-bit [T*HW*HW, ldo] (pad columns zeroed).
// src = (dst + 0.5) * G/HW - 0.5 clamped at 0; i0 = floor(src), i1 = min(i0+1, G-1), w1 = src - i0.
// ------------------------------------------------------------------------------------------------
template <typename TI> __device__ __forceinline__ f32x4 ld4_as_f32(const TI* p);
template <> __device__ __forceinline__ f32x4 ld4_as_f32<float>(const float* p) { return *(const f32x4*)p; }
template <> __device__ __forceinline__ f32x4 ld4_as_f32<f16_t>(const f16_t* p) { const f16x4 v = *(const f16x4*)p; return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]}; }
template <> __device__ __forceinline__ f32x4 ld4_as_f32<bf16_t>(const bf16_t* p) { const bf16x4 v = *(const bf16x4*)p; return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]}; }

template <typename TO, typename TI = float>
__global__ void pool_kernel(const TI* __restrict__ x, int ldx, TO* __restrict__ out, int ldo, int T, int G, int HW,
                            int D, const int* __restrict__ cmap, int Pc) {
    const long orow = blockIdx.x;                 // t*HW*HW + oy*HW + ox
    const int t = (int)(orow / (HW * HW)), o = (int)(orow % (HW * HW));
    const int oy = o / HW, ox = o % HW;
    const float sc = (float)G / (float)HW;
    // separately rounded multiply and subtract (no FMA contraction): SelectorEngine.pool_sources evaluates the same expression
    // with torch fp32 ops on the host to decide which rows exist in a compact input, and the two must agree at integer boundaries
    float sy = __fsub_rn(__fmul_rn(oy + 0.5f, sc), 0.5f), sx = __fsub_rn(__fmul_rn(ox + 0.5f, sc), 0.5f);
    sy = sy < 0.f ? 0.f : sy;
    sx = sx < 0.f ? 0.f : sx;
    const int y0 = (int)sy, x0 = (int)sx;
    const int y1 = min(y0 + 1, G - 1), x1 = min(x0 + 1, G - 1);
    const float wy1 = sy - y0, wx1 = sx - x0, wy0 = 1.f - wy1, wx0 = 1.f - wx1;
    // dense input: G*G rows per frame; compact input (cmap != null): only the Pc rows the pool reads, at cmap[y*G+x]
    const TI* base = x + (long)t * (cmap ? Pc : G * G) * ldx;
    const int i00 = y0 * G + x0, i01 = y0 * G + x1, i10 = y1 * G + x0, i11 = y1 * G + x1;
    int r00 = i00, r01 = i01, r10 = i10, r11 = i11;
    if (cmap) {
        // a token the host did not list (cmap -1) must never be dereferenced: read row 0 instead and poison the output row with
        // NaN so that the disagreement is loud rather than silent
        r00 = cmap[i00]; r01 = cmap[i01]; r10 = cmap[i10]; r11 = cmap[i11];
        if ((r00 | r01 | r10 | r11) < 0) {
            for (int c = threadIdx.x; c < ldo; c += blockDim.x) out[orow * ldo + c] = T16<TO>::from_f32(__builtin_nanf(""));
            return;
        }
    }
    const TI* p00 = base + (long)r00 * ldx;
    const TI* p01 = base + (long)r01 * ldx;
    const TI* p10 = base + (long)r10 * ldx;
    const TI* p11 = base + (long)r11 * ldx;
    for (int c = threadIdx.x * 4; c < ldo; c += blockDim.x * 4) {
        f32x4 r = {0.f, 0.f, 0.f, 0.f};
        if (c < D) {
            // same association as ATen's upsample_bilinear2d: w_y0*(w_x0*a + w_x1*b) + w_y1*(w_x0*c + w_x1*d)
            r = wy0 * (wx0 * ld4_as_f32<TI>(p00 + c) + wx1 * ld4_as_f32<TI>(p01 + c)) +
                wy1 * (wx0 * ld4_as_f32<TI>(p10 + c) + wx1 * ld4_as_f32<TI>(p11 + c));
        }
        typename T16<TO>::vec4 ov;
#pragma unroll
        for (int e = 0; e < 4; ++e) ov[e] = T16<TO>::from_f32(r[e]);
        *(typename T16<TO>::vec4*)(out + orow * ldo + c) = ov;
    }
}

// out[(b*n_idx + i), :] = src[(b*rows_per_batch + idx[i]), :]   (16-byte granules; used to compact the rows the token pool
// reads before the last tower layer's Q / out-proj / MLP)
__global__ void gather_rows_kernel(const char* __restrict__ src, long ld_src_b, const int* __restrict__ idx, int n_idx,
                                   int rows_per_batch, char* __restrict__ out, long ld_out_b, int row_bytes) {
    const long orow = blockIdx.x;
    const int b = (int)(orow / n_idx), i = (int)(orow % n_idx);
    const char* s = src + ((long)b * rows_per_batch + idx[i]) * ld_src_b;
    char* d = out + orow * ld_out_b;
    for (int c = threadIdx.x * 16; c < row_bytes; c += blockDim.x * 16) *(u32x4*)(d + c) = *(const u32x4*)(s + c);
}

// fp32 rows -> 16-bit rows (no pooling branch HW == G), pad columns zeroed
template <typename TO>
__global__ void cast_rows_kernel(const float* __restrict__ x, int ldx, TO* __restrict__ out, int ldo, int D) {
    const long row = blockIdx.x;
    for (int c = threadIdx.x * 4; c < ldo; c += blockDim.x * 4) {
        f32x4 r = {0.f, 0.f, 0.f, 0.f};
        if (c < D) r = *(const f32x4*)(x + row * ldx + c);
        typename T16<TO>::vec4 ov;
#pragma unroll
        for (int e = 0; e < 4; ++e) ov[e] = T16<TO>::from_f32(r[e]);
        *(typename T16<TO>::vec4*)(out + row * ldo + c) = ov;
    }
}

// ------------------------------------------------------------------------------------------------
// embedding gather: ids[n] -> out_f32[row0 + n, :] = table[ids[n], :]
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void embed_kernel(const long long* __restrict__ ids, const T* __restrict__ table, int D,
                             float* __restrict__ out, int ldo) {
    const long id = ids[blockIdx.x];
    const T* src = table + id * D;
    float* dst = out + (long)blockIdx.x * ldo;
    for (int c = threadIdx.x * 4; c < D; c += blockDim.x * 4) {
        const typename T16<T>::vec4 v = *(const typename T16<T>::vec4*)(src + c);
        f32x4 r;
#pragma unroll
        for (int e = 0; e < 4; ++e) r[e] = T16<T>::to_f32(v[e]);
        *(f32x4*)(dst + c) = r;
    }
}

// ------------------------------------------------------------------------------------------------
// score head: logits[t] = mean_{j<tok}(h[row0 + t*tok + j, :]) . w + b      (h fp32, w fp32)
// one block (256 threads) per frame
// ------------------------------------------------------------------------------------------------
__global__ void pool_score_kernel(const float* __restrict__ h, int ldh, int tok, int D, const float* __restrict__ w,
                                  const float* __restrict__ b, float* __restrict__ logits) {
    __shared__ float part[4];
    const int t = blockIdx.x;
    const float* base = h + (long)t * tok * ldh;
    float acc = 0.f;
    for (int c = threadIdx.x * 4; c < D; c += blockDim.x * 4) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        for (int j = 0; j < tok; ++j) s += *(const f32x4*)(base + (long)j * ldh + c);
        const f32x4 ww = *(const f32x4*)(w + c);
        acc += s[0] * ww[0] + s[1] * ww[1] + s[2] * ww[2] + s[3] * ww[3];
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) logits[t] = (part[0] + part[1] + part[2] + part[3]) / tok + b[0];
}

// ------------------------------------------------------------------------------------------------
// rank: scores = sigmoid(logits); order = stable argsort descending (ties: lower position first).
// Single block bitonic sort of 64-bit keys (~orderable(score) << 32 | index) in LDS; n <= 8192.
// ------------------------------------------------------------------------------------------------
__global__ void rank_kernel(const float* __restrict__ logits, int n, int npow2, float* __restrict__ scores,
                            int* __restrict__ order) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned long long* keys = (unsigned long long*)smem;
    for (int i = threadIdx.x; i < npow2; i += blockDim.x) {
        unsigned long long k = ~0ull;                       // padding sorts last
        if (i < n) {
            const float s = 1.0f / (1.0f + expf(-logits[i]));
            scores[i] = s;
            unsigned u = __float_as_uint(s);
            u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);   // ascending-orderable
            u = ~u;                                            // descending
            k = ((unsigned long long)u << 32) | (unsigned)i;
        }
        keys[i] = k;
    }
    __syncthreads();
    for (int sz = 2; sz <= npow2; sz <<= 1)
        for (int st = sz >> 1; st > 0; st >>= 1) {
            for (int i = threadIdx.x; i < npow2; i += blockDim.x) {
                const int j = i ^ st;
                if (j > i) {
                    const unsigned long long a = keys[i], b = keys[j];
                    const bool up = (i & sz) == 0;
                    if ((a > b) == up) { keys[i] = b; keys[j] = a; }
                }
            }
            __syncthreads();
        }
    for (int i = threadIdx.x; i < n; i += blockDim.x) order[i] = (int)(keys[i] & 0xffffffffu);
}

}  // namespace vitg

using namespace vitg;

#define CHECK_LAUNCH() (hipGetLastError() == hipSuccess ? 0 : VITG_ERR_LAUNCH)

extern "C" int vitg_patchify(const void* pix, int pix_dtype, void* out, int T, int R, int patch, int Kpad, int dtype,
                             void* stream) {
    if (T <= 0) return 0;
    if (R < patch) return VITG_ERR_SHAPE;          // R need not be a multiple of the patch: like the stride-14 VALID convolution it replaces,
                                                   // the kernel ignores the last R % patch rows and columns (SigLIP-so400m: 384 = 27 * 14 + 6)
    if (Kpad < 3 * patch * patch) return VITG_ERR_SHAPE;
    const int G = R / patch;
    const long rows = (long)T * G * G;
    hipStream_t s = (hipStream_t)stream;
    // coalesced whole-row form when pixel rows are 16-B granular (every production resolution: 384, 448, 224 ...); the per-element kernel
    // below keeps the odd sizes of the test fixtures (62, 70)
    const size_t rows_lds = (size_t)3 * patch * R * 2;
#if VITG_PROFILING_BUILD
    static const bool per_element = [] { const char* e = getenv("VITG_PATCHIFY_ROWS"); return e && atoi(e) == 0; }();      // A/B knob (profiling / experiment builds)
#else
    constexpr bool per_element = false;
#endif
    if (R % 8 == 0 && Kpad % 8 == 0 && rows_lds <= 64 * 1024 && !per_element) {
        dim3 rblk(256), rgrid((unsigned)(T * G));
#define PR(TI, TO) hipLaunchKernelGGL((patchify_rows_kernel<TI, TO>), rgrid, rblk, rows_lds, s, (const TI*)pix, (TO*)out, R, patch, G, Kpad)
        if (dtype == DT_BF16) {
            if (pix_dtype == DT_F32) PR(float, bf16_t); else if (pix_dtype == DT_BF16) PR(bf16_t, bf16_t); else if (pix_dtype == DT_F16) PR(f16_t, bf16_t); else return VITG_ERR_DTYPE;
        } else if (dtype == DT_F16) {
            if (pix_dtype == DT_F32) PR(float, f16_t); else if (pix_dtype == DT_BF16) PR(bf16_t, f16_t); else if (pix_dtype == DT_F16) PR(f16_t, f16_t); else return VITG_ERR_DTYPE;
        } else return VITG_ERR_DTYPE;
#undef PR
        return CHECK_LAUNCH();
    }
    dim3 blk(256), grid((unsigned)((rows + 3) / 4));
#define PF(TI, TO) hipLaunchKernelGGL((patchify_kernel<TI, TO>), grid, blk, 0, s, pix, (TO*)out, T, R, patch, G, Kpad, rows)
    if (dtype == DT_BF16) {
        if (pix_dtype == DT_F32) PF(float, bf16_t); else if (pix_dtype == DT_BF16) PF(bf16_t, bf16_t); else if (pix_dtype == DT_F16) PF(f16_t, bf16_t); else return VITG_ERR_DTYPE;
    } else if (dtype == DT_F16) {
        if (pix_dtype == DT_F32) PF(float, f16_t); else if (pix_dtype == DT_BF16) PF(bf16_t, f16_t); else if (pix_dtype == DT_F16) PF(f16_t, f16_t); else return VITG_ERR_DTYPE;
    } else return VITG_ERR_DTYPE;
#undef PF
    return CHECK_LAUNCH();
}

template <bool RMS>
static int norm_dispatch(const float* x, int ldx, const float* w, const float* b, void* out, int ldo, int out_dtype,
                         long M, int D, float eps, hipStream_t s) {
    if (M <= 0) return 0;
    if (D % 4 || ldo % 4 || ldx % 4) return VITG_ERR_ALIGN;
    if (ldo < D) return VITG_ERR_SHAPE;
    const int need = ceil_div(ldo, 256);
    const long want = (M + 3) / 4;
    // 16384 blocks measured 5-6 % faster than 4096 (0.757 -> 0.710 ms on the ViT LN), +0.5 % on the step; VITG_NORM_GRID: A/B knob of the profiling / experiment builds
#if VITG_PROFILING_BUILD
    static const int cap = [] { const char* e = getenv("VITG_NORM_GRID"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 16384; }();
#else
    constexpr int cap = 16384;
#endif
    dim3 blk(256), grid((unsigned)(want < cap ? want : cap));     // rows grid-strided over the resident waves
    const int of32 = out_dtype == DT_F32;
#define NK(TO, NV) do { if (want > cap) hipLaunchKernelGGL((norm_kernel<TO, RMS, NV, true>), grid, blk, 0, s, x, ldx, w, b, out, ldo, of32, M, D, eps); \
                        else hipLaunchKernelGGL((norm_kernel<TO, RMS, NV, false>), grid, blk, 0, s, x, ldx, w, b, out, ldo, of32, M, D, eps); } while (0)
#define NKS(TO) do { if (need <= 1) NK(TO, 1); else if (need <= 2) NK(TO, 2); else if (need <= 5) NK(TO, 5); else if (need <= 8) NK(TO, 8); else if (need <= 16) NK(TO, 16); else return VITG_ERR_SHAPE; } while (0)
    if (out_dtype == DT_F16) NKS(f16_t); else if (out_dtype == DT_BF16 || out_dtype == DT_F32) NKS(bf16_t); else return VITG_ERR_DTYPE;
#undef NKS
#undef NK
    return CHECK_LAUNCH();
}

extern "C" int vitg_layernorm(const float* x, int ldx, const float* w, const float* b, void* out, int ldo, int out_dtype,
                              long M, int D, float eps, void* stream) {
    return norm_dispatch<false>(x, ldx, w, b, out, ldo, out_dtype, M, D, eps, (hipStream_t)stream);
}

extern "C" int vitg_rmsnorm(const float* x, int ldx, const float* w, void* out, int ldo, int out_dtype, long M, int D,
                            float eps, void* stream) {
    return norm_dispatch<true>(x, ldx, w, nullptr, out, ldo, out_dtype, M, D, eps, (hipStream_t)stream);
}

extern "C" int vitg_layernorm16(const void* x, int ldx, const float* w, const float* b, void* out, int ldo, int dtype, long M, int D,
                                float eps, void* stream) {
    if (M <= 0) return 0;
    if (D % 8 || ldo % 8 || ldx % 8) return VITG_ERR_ALIGN;
    if (ldo < D) return VITG_ERR_SHAPE;
    const int need = ceil_div(ldo, 512);
    // (VITG_LN16_CAP / VITG_LN16_BLOCK: A/B knobs for the grid cap and the block size - profiling / experiment builds only; defaults measured,
    // profiles/r03_residual16.md)
#if VITG_PROFILING_BUILD
    static const long cap_env = [] { const char* e = getenv("VITG_LN16_CAP"); return e ? atol(e) : 0L; }();
    static const int blk_env = [] { const char* e = getenv("VITG_LN16_BLOCK"); return e ? atoi(e) : 0; }();
#else
    constexpr long cap_env = 0;
    constexpr int blk_env = 0;
#endif
    const int threads = (blk_env == 128 || blk_env == 512 || blk_env == 1024) ? blk_env : 256;
    const int wpb = threads / 64;
    if (D == 9 * 128 && b && !cap_env && !blk_env) {           // the tower's width: four rows per wave (ln16r4_kernel)
        constexpr int wpb4 = 4;
        const long want4 = ((M + 3) / 4 + wpb4 - 1) / wpb4, cap4 = 8192;
        dim3 blk4(256), grid4((unsigned)(want4 < cap4 ? want4 : cap4));
        hipStream_t s4 = (hipStream_t)stream;
#define LK4(T) do { if (want4 > cap4) hipLaunchKernelGGL((ln16r4_kernel<T, 9, true>), grid4, blk4, 0, s4, (const T*)x, ldx, w, b, (T*)out, ldo, M, D, eps); \
                    else hipLaunchKernelGGL((ln16r4_kernel<T, 9, false>), grid4, blk4, 0, s4, (const T*)x, ldx, w, b, (T*)out, ldo, M, D, eps); } while (0)
        if (dtype == DT_F16) LK4(f16_t); else if (dtype == DT_BF16) LK4(bf16_t); else return VITG_ERR_DTYPE;
#undef LK4
        return CHECK_LAUNCH();
    }
    const long want = (M + wpb - 1) / wpb;
    const long cap = cap_env > 0 ? cap_env : 16384;
    dim3 blk(threads), grid((unsigned)(want < cap ? want : cap));
    hipStream_t s = (hipStream_t)stream;
#define LK(T, NV) do { if (want > cap) hipLaunchKernelGGL((ln16_kernel<T, NV, true>), grid, blk, 0, s, (const T*)x, ldx, w, b, (T*)out, ldo, M, D, eps); \
                        else hipLaunchKernelGGL((ln16_kernel<T, NV, false>), grid, blk, 0, s, (const T*)x, ldx, w, b, (T*)out, ldo, M, D, eps); } while (0)
#define LKS(T) do { if (need <= 1) LK(T, 1); else if (need <= 3) LK(T, 3); else if (need <= 8) LK(T, 8); else return VITG_ERR_SHAPE; } while (0)
    if (dtype == DT_F16) LKS(f16_t); else if (dtype == DT_BF16) LKS(bf16_t); else return VITG_ERR_DTYPE;
#undef LKS
#undef LK
    return CHECK_LAUNCH();
}

extern "C" int vitg_rope(void* qkv, int ld, const float* cos_t, const float* sin_t, long N, int heads, int head_dim,
                         int dtype, void* stream) {
    if (N <= 0 || heads <= 0) return 0;
    if (head_dim % 8 || ld % 4) return VITG_ERR_ALIGN;
    const long total = N * heads * (head_dim / 8);
    dim3 blk(256), grid((unsigned)((total + 255) / 256));
    hipStream_t s = (hipStream_t)stream;
    if (dtype == DT_BF16) hipLaunchKernelGGL((rope_kernel<bf16_t>), grid, blk, 0, s, (bf16_t*)qkv, ld, cos_t, sin_t, N, heads, head_dim);
    else if (dtype == DT_F16) hipLaunchKernelGGL((rope_kernel<f16_t>), grid, blk, 0, s, (f16_t*)qkv, ld, cos_t, sin_t, N, heads, head_dim);
    else return VITG_ERR_DTYPE;
    return CHECK_LAUNCH();
}

extern "C" int vitg_gather_rows(const void* src, long ld_src_bytes, const int* idx, int n_idx, int rows_per_batch, int batch,
                                void* out, long ld_out_bytes, int row_bytes, void* stream) {
    if (batch <= 0 || n_idx <= 0) return 0;
    if (row_bytes % 16 || ld_src_bytes % 16 || ld_out_bytes % 16) return VITG_ERR_ALIGN;
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((long)batch * n_idx)), dim3(256), 0, (hipStream_t)stream, (const char*)src,
                       ld_src_bytes, idx, n_idx, rows_per_batch, (char*)out, ld_out_bytes, row_bytes);
    return CHECK_LAUNCH();
}

extern "C" int vitg_bilinear_pool(const float* x, int ldx, void* out, int ldo, int T, int G, int HW, int D, int dtype,
                                  const int* cmap, int Pc, void* stream) {
    if (T <= 0) return 0;
    if (D % 4 || ldo % 4 || ldx % 4 || ldo < D) return VITG_ERR_ALIGN;
    if (HW <= 0 || HW > G) return VITG_ERR_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    dim3 blk(256);
    if (HW == G) {
        if (cmap) return VITG_ERR_ARG;
        dim3 grid((unsigned)((long)T * G * G));
        if (dtype == DT_BF16) hipLaunchKernelGGL((cast_rows_kernel<bf16_t>), grid, blk, 0, s, x, ldx, (bf16_t*)out, ldo, D);
        else if (dtype == DT_F16) hipLaunchKernelGGL((cast_rows_kernel<f16_t>), grid, blk, 0, s, x, ldx, (f16_t*)out, ldo, D);
        else return VITG_ERR_DTYPE;
    } else {
        dim3 grid((unsigned)((long)T * HW * HW));
        if (dtype == DT_BF16) hipLaunchKernelGGL((pool_kernel<bf16_t>), grid, blk, 0, s, x, ldx, (bf16_t*)out, ldo, T, G, HW, D, cmap, Pc);
        else if (dtype == DT_F16) hipLaunchKernelGGL((pool_kernel<f16_t>), grid, blk, 0, s, x, ldx, (f16_t*)out, ldo, T, G, HW, D, cmap, Pc);
        else return VITG_ERR_DTYPE;
    }
    return CHECK_LAUNCH();
}

extern "C" int vitg_bilinear_pool16(const void* x, int ldx, void* out, int ldo, int T, int G, int HW, int D, int dtype,
                                    const int* cmap, int Pc, void* stream) {
    if (T <= 0) return 0;
    if (D % 4 || ldo % 4 || ldx % 4 || ldo < D) return VITG_ERR_ALIGN;
    if (HW <= 0 || HW > G) return VITG_ERR_SHAPE;
    if (HW == G && cmap) return VITG_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    // (HW == G runs the same kernel: every output token then reads its own source token with weight 1 and the pad columns are zeroed)
    dim3 blk(256), grid((unsigned)((long)T * HW * HW));
    if (dtype == DT_BF16) hipLaunchKernelGGL((pool_kernel<bf16_t, bf16_t>), grid, blk, 0, s, (const bf16_t*)x, ldx, (bf16_t*)out, ldo, T, G, HW, D, cmap, Pc);
    else if (dtype == DT_F16) hipLaunchKernelGGL((pool_kernel<f16_t, f16_t>), grid, blk, 0, s, (const f16_t*)x, ldx, (f16_t*)out, ldo, T, G, HW, D, cmap, Pc);
    else return VITG_ERR_DTYPE;
    return CHECK_LAUNCH();
}

extern "C" int vitg_embed_gather(const long long* ids, int n, const void* table, int D, float* out, int ldo, int dtype,
                                 void* stream) {
    if (n <= 0) return 0;
    if (D % 4 || ldo % 4) return VITG_ERR_ALIGN;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == DT_BF16) hipLaunchKernelGGL((embed_kernel<bf16_t>), dim3(n), dim3(256), 0, s, ids, (const bf16_t*)table, D, out, ldo);
    else if (dtype == DT_F16) hipLaunchKernelGGL((embed_kernel<f16_t>), dim3(n), dim3(256), 0, s, ids, (const f16_t*)table, D, out, ldo);
    else return VITG_ERR_DTYPE;
    return CHECK_LAUNCH();
}

extern "C" int vitg_pool_score(const float* h, int ldh, int frames, int tokens_per_frame, int D, const float* w,
                               const float* b, float* logits, void* stream) {
    if (frames <= 0) return 0;
    if (D % 4 || ldh % 4 || tokens_per_frame <= 0) return VITG_ERR_ALIGN;
    hipLaunchKernelGGL(pool_score_kernel, dim3(frames), dim3(256), 0, (hipStream_t)stream, h, ldh, tokens_per_frame, D, w, b, logits);
    return CHECK_LAUNCH();
}


// ---- grounding loss (forward value only): BCE-with-logits with the reference's capped positive weight --------------------
// pos_weight = min(5, sqrt(neg / max(1, pos))), loss = mean_i [ (1 - y_i) x_i + (1 + (pw - 1) y_i) softplus(-x_i) ]
// (torch's binary_cross_entropy_with_logits formula; /root/reference/eagle/model/language_model/grounding_qwen2.py:159-169).
namespace vitg {
__global__ __launch_bounds__(1024) void bce_loss_kernel(const float* __restrict__ x, const float* __restrict__ y, int n, float* __restrict__ out) {
    __shared__ float red[32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float pos = 0.f;
    for (int i = tid; i < n; i += blockDim.x) pos += y[i];
    pos = wave_sum(pos);
    if (lane == 0) red[wave] = pos;
    __syncthreads();
    pos = 0.f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) pos += red[w];
    __syncthreads();
    const float neg = (float)n - pos;
    const float pw = fminf(5.0f, sqrtf(neg / fmaxf(1.0f, pos)));
    float acc = 0.f;
    for (int i = tid; i < n; i += blockDim.x) {
        const float xi = x[i], yi = y[i];
        const float sp = log1pf(expf(-fabsf(xi))) + fmaxf(-xi, 0.0f);       // softplus(-x), stable
        acc += (1.0f - yi) * xi + (1.0f + (pw - 1.0f) * yi) * sp;
    }
    acc = wave_sum(acc);
    if (lane == 0) red[wave] = acc;
    __syncthreads();
    if (tid == 0) {
        float t = 0.f;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += red[w];
        out[0] = t / (float)n;
        out[1] = pw;
    }
}
}  // namespace vitg

extern "C" int vitg_bce_loss(const float* logits, const float* labels, int n, float* out2, void* stream) {
    if (n <= 0 || !logits || !labels || !out2) return VITG_ERR_ARG;
    hipLaunchKernelGGL(vitg::bce_loss_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, logits, labels, n, out2);
    return hipGetLastError() == hipSuccess ? 0 : VITG_ERR_LAUNCH;
}

extern "C" int vitg_rank(const float* logits, int n, float* scores, int* order, void* stream) {
    if (n <= 0) return 0;
    if (n > 8192) return VITG_ERR_SHAPE;
    int p2 = 1;
    while (p2 < n) p2 <<= 1;
    static std::atomic<unsigned> done_mask{0};
    const int cur_dev = current_device();
    if (!device_done(done_mask, cur_dev)) { (void)hipFuncSetAttribute((const void*)rank_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 8192 * 8); mark_device_done(done_mask, cur_dev); }
    hipLaunchKernelGGL(rank_kernel, dim3(1), dim3(1024), (size_t)p2 * 8, (hipStream_t)stream, logits, n, p2, scores, order);
    return CHECK_LAUNCH();
}

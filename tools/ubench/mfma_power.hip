// Is the MFMA rate on MI355X data dependent?  MFMA-only kernel (no memory traffic in the timed loop), 2 waves per SIMD, the
// register shape of the GEMM main loop (8 accumulators of 32x32 per wave), operands taken from a pool of 4 A and 4 B fragments
// held in registers.  Three fillings of the pool:
//   zeros            - no multiplier input ever toggles
//   one random frag  - every MFMA sees the same (random) operands
//   random pool      - consecutive MFMAs see different random operands, like a GEMM on real data
// The average shader clock of block 0 (s_memtime cycles per 100-MHz s_memrealtime tick) is printed next to each rate.
// Each case runs back to back for several hundred ms so that the power management settles; the per-repeat rates are printed.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_power.hip -o tools/ubench/mfma_power
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int BF> __global__ __launch_bounds__(512, 2) void k(const uint4* pool, float* out, int iters, long long* clk) {
    const long long c0 = clock64(), w0 = wall_clock64();          // shader-clock cycles (s_memtime) / constant 100 MHz (s_memrealtime)
    // pool: 8 fragments x 64 lanes x 16 bytes per wave slot (all waves read the same 8 KiB: L2 hit, outside the timed loop)
    uint4 fr[8];
    for (int i = 0; i < 8; ++i) fr[i] = pool[i * 64 + (threadIdx.x & 63)];
    f32x16 c[2][4];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) c[i][j][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 2; ++u)                       // two "k-steps": A fragments (0,1) then (2,3); B fragments 4..7
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (BF) c[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(bf16x8*)&fr[u * 2 + i], *(bf16x8*)&fr[4 + j], c[i][j], 0, 0, 0);
                    else c[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*(f16x8*)&fr[u * 2 + i], *(f16x8*)&fr[4 + j], c[i][j], 0, 0, 0);
                }
    }
    float s = 0;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += c[i][j][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (clk && blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = clock64() - c0; clk[1] = wall_clock64() - w0; }
}

static unsigned short rnd16(int bf) {                    // a finite 16-bit float in [-1, 1)
    const float v = (float)rand() / RAND_MAX * 2.f - 1.f;
    if (bf) { unsigned u; memcpy(&u, &v, 4); return (unsigned short)(u >> 16); }
    _Float16 h = (_Float16)v; unsigned short r; memcpy(&r, &h, 2); return r;
}

template <int BF> void run(const char* name, int fill) {
    std::vector<unsigned short> h(8 * 64 * 8);
    for (int f = 0; f < 8; ++f)
        for (int i = 0; i < 512; ++i)
            h[f * 512 + i] = fill == 0 ? 0 : fill == 1 ? (f == 0 || f == 4 ? rnd16(BF) : 0) : rnd16(BF);
    if (fill == 1)                                       // the same random fragment in every A slot / every B slot
        for (int f = 0; f < 8; ++f)
            if (f != 0 && f != 4) memcpy(&h[f * 512], &h[(f < 4 ? 0 : 4) * 512], 1024);
    uint4* pool; float* out; long long* clk; long long hclk[2];
    hipMalloc(&clk, 16);
    hipMalloc(&pool, h.size() * 2); hipMalloc(&out, 256 * 512 * 4);
    hipMemcpy(pool, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    const int iters = 40000, blocks = 256;               // 16 MFMAs per iteration per wave
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<BF><<<blocks, 512>>>(pool, out, 16, nullptr);
    printf("%-5s %-16s", BF ? "bf16" : "fp16", name);
    for (int rep = 0; rep < 6; ++rep) {
        hipEventRecord(a);
        k<BF><<<blocks, 512>>>(pool, out, iters, clk);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        const double flops = (double)blocks * 8 * iters * 16 * 32768.0;
        hipMemcpy(hclk, clk, 16, hipMemcpyDeviceToHost);
        printf("  %6.0f @%4.0f MHz", flops / ms / 1e9, (double)hclk[0] / (double)hclk[1] * 100.0);
    }
    printf("  TFLOP/s\n");
    hipFree(pool); hipFree(out);
}
int main() {
    for (int round = 0; round < 2; ++round) {
        run<1>("zeros", 0); run<1>("one random frag", 1); run<1>("random pool", 2);
        run<0>("zeros", 0); run<0>("one random frag", 1); run<0>("random pool", 2);
    }
    return 0;
}

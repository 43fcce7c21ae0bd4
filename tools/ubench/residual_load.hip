// How fast can all 256 CUs pull their 256 x 384 fp32 residual tiles into registers at the same moment (the prologue of the fp32-residual
// GEMMs: 28 us per tile in the full-grid timeline against 9 us on a nearly idle chip, profiles/r02_gemm_timeline.md)?  Two lane -> address
// maps over the same bytes of a [524288, 1152] fp32 matrix, 8 waves per block, wave tile 128 x 96 as in the kernel:
//   A  accumulator layout: an instruction covers 32 rows x 32 B (lane = row, lane>>5 picks the 16-B half)  -> quarter-line requests
//   B  coalesced: an instruction covers 64 consecutive 16-B chunks of the wave tile, row-major (24 chunks = 384 B per row)
// Prints GB/s over the whole grid for both, and for the write direction (the epilogue's stores use B already).
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/residual_load.hip -o tools/ubench/residual_load && tools/ubench/residual_load
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int PAT>
__global__ __launch_bounds__(512, 1) void k(const float* __restrict__ x, float* __restrict__ out, int ld, int tiles_m, int tiles_n) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wr = wave >> 2, wc = wave & 3;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int t = blockIdx.x; t < tiles_m * tiles_n; t += gridDim.x) {
        const int tm = t / tiles_n, tn = t % tiles_n;
        const float* base = x + (size_t)(tm * 256 + wr * 128) * ld + tn * 384 + wc * 96;
        f32x4 v[48];
        if (PAT == 0) {
            const int frow = lane & 31, fh = lane >> 5;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 3; ++nt)
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4)
                        v[(mt * 3 + nt) * 4 + q4] = *(const f32x4*)(base + (size_t)(mt * 32 + frow) * ld + nt * 32 + q4 * 8 + fh * 4);
        } else {
#pragma unroll
            for (int i = 0; i < 48; ++i) {
                const int idx = i * 64 + lane, row = idx / 24, c = idx % 24;
                v[i] = *(const f32x4*)(base + (size_t)row * ld + c * 4);
            }
        }
#pragma unroll
        for (int i = 0; i < 48; ++i) acc += v[i];
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[threadIdx.x] = acc[0];
}
int main() {
    const int M = 524288, N = 1152;
    float *x, *o;
    hipMalloc(&x, (size_t)M * N * 4); hipMalloc(&o, 4096);
    hipMemset(x, 0, (size_t)M * N * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep)
        for (int pat = 0; pat < 2; ++pat)
            for (int grid : {256, 64}) {
                hipEventRecord(e0);
                if (pat == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(512), 0, 0, x, o, N, M / 256, N / 384);
                else hipLaunchKernelGGL(k<1>, dim3(grid), dim3(512), 0, 0, x, o, N, M / 256, N / 384);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                printf("pattern %s grid %3d: %.3f ms  %.2f TB/s  (%.1f us per 256x384 tile and CU)\n", pat ? "B coalesced  " : "A accumulator", grid, ms,
                       (double)M * N * 4 / ms / 1e9, ms * 1e3 / ((M / 256) * (N / 384) / (double)grid));
            }
    return 0;
}

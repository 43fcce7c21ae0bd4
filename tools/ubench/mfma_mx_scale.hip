// Semantics probe for the block-scale operands of v_mfma_scale_f32_32x32x64_f8f6f4 (MX: one E8M0 scale per 32 K-elements).
// One wave, A = B = all 1.0 (e4m3 0x38), so C[i][j] = sum over the two 32-deep K blocks of scale_a(i, blk) * scale_b(j, blk) * 32.
// Experiments: (1) lane L gets scale 2^1 in byte 0 of scale_b, everything else 2^0 -> which C entries change tells which (column, K block)
// lane L's scale applies to; (2) the same for scale_a; (3) scale bytes 1..3 with opsel 1..3.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/mfma_mx_scale.hip -o tools/ubench/mfma_mx_scale && tools/ubench/mfma_mx_scale
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int OPA, int OPB>
__global__ void k(const unsigned* sa, const unsigned* sb, float* out) {
    const int lane = threadIdx.x;
    i32x8 ones;
    for (int i = 0; i < 8; ++i) ones[i] = 0x38383838;
    f32x16 c;
    for (int i = 0; i < 16; ++i) c[i] = 0.f;
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(ones, ones, c, 0, 0, OPA, (int)sa[lane], OPB, (int)sb[lane]);
    for (int r = 0; r < 16; ++r) out[lane * 16 + r] = c[r];
}
// C layout of the 32x32 MFMA: lane l, register r -> row = 8*(r>>2) + 4*(l>>5) + (r&3), column = l & 31
static void decode(const std::vector<float>& h, float C[32][32]) {
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 16; ++r) C[8 * (r >> 2) + 4 * (l >> 5) + (r & 3)][l & 31] = h[l * 16 + r];
}
template <int OPA, int OPB>
static void run(const char* what, int lane_a, int lane_b, int byte) {
    std::vector<unsigned> sa(64, 0x7f7f7f7fu), sb(64, 0x7f7f7f7fu);
    if (lane_a >= 0) sa[lane_a] = (sa[lane_a] & ~(0xffu << (8 * byte))) | (0x80u << (8 * byte));
    if (lane_b >= 0) sb[lane_b] = (sb[lane_b] & ~(0xffu << (8 * byte))) | (0x80u << (8 * byte));
    unsigned *da, *db; float* dout;
    hipMalloc(&da, 256); hipMalloc(&db, 256); hipMalloc(&dout, 64 * 16 * 4);
    hipMemcpy(da, sa.data(), 256, hipMemcpyHostToDevice); hipMemcpy(db, sb.data(), 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL((k<OPA, OPB>), dim3(1), dim3(64), 0, 0, da, db, dout);
    std::vector<float> h(64 * 16);
    hipMemcpy(h.data(), dout, h.size() * 4, hipMemcpyDeviceToHost);
    float C[32][32]; decode(h, C);
    int nrows = 0, ncols = 0, r0 = -1, c0 = -1; float v = 0;
    for (int i = 0; i < 32; ++i) { bool any = false; for (int j = 0; j < 32; ++j) if (C[i][j] != 64.f) { any = true; v = C[i][j]; c0 = j; } if (any) { ++nrows; r0 = i; } }
    for (int j = 0; j < 32; ++j) { bool any = false; for (int i = 0; i < 32; ++i) if (C[i][j] != 64.f) any = true; if (any) ++ncols; }
    printf("%-58s: baseline C = %.0f; changed rows %d (last %d) cols %d (last %d), changed value %.0f\n", what, C[31][31] == 64.f ? 64.f : C[31][31], nrows, r0, ncols, c0, v);
    hipFree(da); hipFree(db); hipFree(dout);
}
int main() {
    run<0, 0>("no scale change", -1, -1, 0);
    run<0, 0>("scale_b lane 5 byte0 = 2^1, opsel 0", -1, 5, 0);
    run<0, 0>("scale_b lane 37 byte0 = 2^1, opsel 0", -1, 37, 0);
    run<0, 0>("scale_a lane 5 byte0 = 2^1, opsel 0", 5, -1, 0);
    run<0, 0>("scale_a lane 37 byte0 = 2^1, opsel 0", 37, -1, 0);
    run<0, 0>("scale_b lane 5 byte1 = 2^1, opsel 0 (should be unused)", -1, 5, 1);
    run<0, 1>("scale_b lane 5 byte1 = 2^1, opsel_b 1", -1, 5, 1);
    run<0, 2>("scale_b lane 5 byte2 = 2^1, opsel_b 2", -1, 5, 2);
    run<0, 3>("scale_b lane 5 byte3 = 2^1, opsel_b 3", -1, 5, 3);
    run<2, 0>("scale_a lane 5 byte2 = 2^1, opsel_a 2", 5, -1, 2);
    return 0;
}

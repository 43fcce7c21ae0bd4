// Which lane's block scale multiplies which of a lane's 32 operand bytes in v_mfma_scale_f32_32x32x64_f8f6f4 (fp8)?
// A = all 1.0; B = zero except ONE byte (lane L, byte p) = 1.0; scale_b = 2^1 on lanes 0-31 and 2^0 on lanes 32-63 (then the other way
// round).  C[:, L & 31] is then exactly the scale that was applied to that element: tells which 32-lane half's scale owns (L >> 5, p).
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/mfma_mx_kmap.hip -o tools/ubench/mfma_mx_kmap && tools/ubench/mfma_mx_kmap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstring>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(const unsigned char* bbytes, const unsigned* sb, float* out) {
    const int lane = threadIdx.x;
    i32x8 ones, b;
    for (int i = 0; i < 8; ++i) { ones[i] = 0x38383838; b[i] = ((const int*)bbytes)[lane * 8 + i]; }
    f32x16 c;
    for (int i = 0; i < 16; ++i) c[i] = 0.f;
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(ones, b, c, 0, 0, 0, 0x7f7f7f7f, 0, (int)sb[lane]);
    for (int r = 0; r < 16; ++r) out[lane * 16 + r] = c[r];
}
int main() {
    unsigned char* db; unsigned* ds; float* dout;
    hipMalloc(&db, 64 * 32); hipMalloc(&ds, 256); hipMalloc(&dout, 64 * 16 * 4);
    for (int L : {5, 37})
        for (int p : {0, 3, 4, 15, 16, 19, 20, 31}) {
            float res[2];
            for (int which = 0; which < 2; ++which) {
                std::vector<unsigned char> bb(64 * 32, 0); bb[L * 32 + p] = 0x38;
                std::vector<unsigned> sb(64);
                for (int l = 0; l < 64; ++l) sb[l] = ((l < 32) == (which == 0)) ? 0x80808080u : 0x7f7f7f7fu;
                hipMemcpy(db, bb.data(), bb.size(), hipMemcpyHostToDevice); hipMemcpy(ds, sb.data(), 256, hipMemcpyHostToDevice);
                hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, db, ds, dout);
                std::vector<float> h(64 * 16); hipMemcpy(h.data(), dout, h.size() * 4, hipMemcpyDeviceToHost);
                res[which] = h[(L & 31) * 16 + 0];      // lane (L & 31), register 0: row 0, column L & 31
            }
            printf("element (lane %2d, byte %2d): C = %.0f with lanes 0-31 scaled x2, %.0f with lanes 32-63 scaled x2 -> owned by the scale of lanes %s\n",
                   L, p, res[0], res[1], res[0] == 2.f ? "0-31" : res[1] == 2.f ? "32-63" : "?");
        }
    return 0;
}

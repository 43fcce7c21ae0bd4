// Which XCD does hardware block b land on?  The tile orders of gemm256.hip / attention*.hip assume XCD = b % 8 (round-robin
// dispatch).  Records HW_REG_XCC_ID per block for a few grid sizes and occupancies and prints the share of blocks that follow b % 8.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/xcc_map.hip -o tools/ubench/xcc_map && tools/ubench/xcc_map
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void rec(int* out, int spin) {
    const int x = __builtin_amdgcn_s_getreg((3 << 11) | 20);      // HW_REG_XCC_ID, bits [3:0]
    if (threadIdx.x == 0) out[blockIdx.x] = x;
    for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(8);
}
int main() {
    for (int threads : {256, 512})
        for (int lds : {0, 160 * 1024})
            for (int n : {256, 2048, 6144, 23222}) {
                int* d; hipMalloc(&d, n * 4);
                hipFuncSetAttribute((const void*)rec, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                hipLaunchKernelGGL(rec, dim3(n), dim3(threads), lds, 0, d, 2000);
                hipDeviceSynchronize();
                std::vector<int> h(n); hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost);
                int ok = 0, cnt[16] = {0};
                for (int b = 0; b < n; ++b) { ok += (h[b] == b % 8); cnt[h[b] & 15]++; }
                printf("threads %d lds %6d blocks %6d: XCC == b%%8 for %.1f %%; per-XCC counts", threads, lds, n, 100.0 * ok / n);
                for (int i = 0; i < 8; ++i) printf(" %d", cnt[i]);
                printf("; first 16:"); for (int b = 0; b < 16; ++b) printf(" %d", h[b]);
                printf("\n");
                hipFree(d);
            }
    return 0;
}

// Library / device identification entry points of the C ABI.
#include <hip/hip_runtime.h>
#include <string.h>
#include "../../include/videoitg_hip.h"
#include "common.h"
#include <stdlib.h>

extern "C" int vitg_version(void) { return 100; }

extern "C" int vitg_device_info(char* name, int name_len) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return VITG_ERR_LAUNCH;
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, dev) != hipSuccess) return VITG_ERR_LAUNCH;
    if (name && name_len > 0) {
        strncpy(name, p.gcnArchName, (size_t)name_len - 1);
        name[name_len - 1] = 0;
    }
    return p.multiProcessorCount;
}

namespace vitg {
static int env_int(const char* k) { const char* e = getenv(k); return e ? atoi(e) : 0; }
// values a PRODUCT key may take (the research values of these keys select variants only the profiling library instantiates)
static bool product_value_ok(const char* key, int v) {
    if (!strcmp(key, "gemm_tile")) return v == 0 || v == 128 || v == 256;
    if (!strcmp(key, "gemm_cfg")) return v == 0 || v == 2 || v == 6 || v == 9 || v == 10 || v == 12;
    if (!strcmp(key, "attn_ab")) return v == 0 || v == 2;
    if (!strcmp(key, "attn_sp")) return v == 0 || v == 2;
    return false;
}
static int env_product(const char* env, const char* key) { const int v = env_int(env); return (VITG_PROFILING_BUILD || product_value_ok(key, v)) ? v : 0; }
Tuning& tuning() {
    static Tuning t = [] {
        Tuning x = {};
        x.gemm_tile = env_product("VITG_GEMM_TILE", "gemm_tile"); x.gemm_cfg = env_product("VITG_GEMM_CFG", "gemm_cfg");
        x.attn_ab = env_product("VITG_ATTN_AB", "attn_ab"); x.attn_sp = env_product("VITG_ATTN_SP", "attn_sp");
#if VITG_PROFILING_BUILD
        x.gemm_flags = env_int("VITG_GEMM_FLAGS"); x.gemm_stagger = env_int("VITG_GEMM_STAGGER"); x.gemm_group = env_int("VITG_GEMM_GROUP");
        x.gemm_pref = env_int("VITG_GEMM_PREF"); x.gemm_mi = env_int("VITG_GEMM_MI"); x.gemm_mi_modes = env_int("VITG_GEMM_MI_MODES");
        x.attn_waves = env_int("VITG_ATTN_WAVES"); x.attn_pipe = env_int("VITG_ATTN_PIPE"); x.attn_dma = env_int("VITG_ATTN_DMA");
        x.gemm_wm4 = env_int("VITG_GEMM_WM4");
#endif
        return x;
    }();
    return t;
}
}  // namespace vitg

extern "C" int vitg_set_option(const char* key, int value) {
    if (!key) return VITG_ERR_ARG;
    vitg::Tuning& t = vitg::tuning();
#if !VITG_PROFILING_BUILD
    if (!vitg::product_value_ok(key, value)) return VITG_ERR_ARG;      // not a product key, or a research value of one
#endif
    if (!strcmp(key, "gemm_tile")) t.gemm_tile = value;
    else if (!strcmp(key, "gemm_cfg")) t.gemm_cfg = value;
    else if (!strcmp(key, "attn_sp")) t.attn_sp = value;
    else if (!strcmp(key, "attn_ab")) t.attn_ab = value;
#if VITG_PROFILING_BUILD
    else if (!strcmp(key, "gemm_flags")) t.gemm_flags = value;
    else if (!strcmp(key, "gemm_stagger")) t.gemm_stagger = value;
    else if (!strcmp(key, "gemm_group")) t.gemm_group = value;
    else if (!strcmp(key, "gemm_mi")) t.gemm_mi = value;
    else if (!strcmp(key, "gemm_pref")) t.gemm_pref = value;
    else if (!strcmp(key, "gemm_mi_modes")) t.gemm_mi_modes = value;
    else if (!strcmp(key, "gemm_wm4")) t.gemm_wm4 = value;
    else if (!strcmp(key, "attn_waves")) t.attn_waves = value;
    else if (!strcmp(key, "attn_pipe")) t.attn_pipe = value;
    else if (!strcmp(key, "attn_dma")) t.attn_dma = value;
#endif
    else return VITG_ERR_ARG;
    return 0;
}

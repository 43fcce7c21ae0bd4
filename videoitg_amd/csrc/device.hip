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
Tuning& tuning() {
    static Tuning t = {env_int("VITG_GEMM_TILE"), env_int("VITG_GEMM_CFG"), env_int("VITG_GEMM_FLAGS"), env_int("VITG_GEMM_STAGGER"), env_int("VITG_GEMM_GROUP"), env_int("VITG_GEMM_PREF"), env_int("VITG_GEMM_MI"), env_int("VITG_GEMM_MI_MODES"),
                       env_int("VITG_ATTN_WAVES"), env_int("VITG_ATTN_PIPE"), env_int("VITG_ATTN_DMA"), env_int("VITG_ATTN_AB"), env_int("VITG_ATTN_SP"), env_int("VITG_GEMM_WM4")};
    return t;
}
}  // namespace vitg

extern "C" int vitg_set_option(const char* key, int value) {
    if (!key) return VITG_ERR_ARG;
    vitg::Tuning& t = vitg::tuning();
    if (!strcmp(key, "gemm_tile")) t.gemm_tile = value;
    else if (!strcmp(key, "gemm_cfg")) t.gemm_cfg = value;
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
    else if (!strcmp(key, "attn_sp")) t.attn_sp = value;
    else if (!strcmp(key, "attn_ab")) t.attn_ab = value;
    else return VITG_ERR_ARG;
    return 0;
}

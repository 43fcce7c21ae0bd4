// Library / device identification entry points of the C ABI.
#include <hip/hip_runtime.h>
#include <string.h>
#include "../../include/videoitg_hip.h"

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

// C-ABI plumbing: thread-local error message, version, device queries.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include "../../include/mc_hip.h"

static thread_local char g_err[512] = "";

void mc_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* mc_last_error(void) { return g_err; }
extern "C" int mc_abi_version(void) { return MC_ABI_VERSION; }
// the 16-bit storage element this build of the library was instantiated on (MC_DTYPE_BF16: libmc_hip.so, MC_DTYPE_F16: libmc_hip_f16.so)
#ifdef MC_STORAGE_F16
extern "C" int mc_storage_dtype(void) { return MC_DTYPE_F16; }
#else
extern "C" int mc_storage_dtype(void) { return MC_DTYPE_BF16; }
#endif

extern "C" int mc_device_info(int* cu_count, int64_t* hbm_bytes, char* arch, int arch_len) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) { mc_set_error("hipGetDevice: %s", hipGetErrorString(e)); return 2; }
    hipDeviceProp_t pr;
    e = hipGetDeviceProperties(&pr, dev);
    if (e != hipSuccess) { mc_set_error("hipGetDeviceProperties: %s", hipGetErrorString(e)); return 2; }
    if (cu_count) *cu_count = pr.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = (int64_t)pr.totalGlobalMem;
    if (arch && arch_len > 0) snprintf(arch, arch_len, "%s", pr.gcnArchName);
    return 0;
}

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

// HIP streams restricted to a subset of the CUs (hipExtStreamCreateWithCUMask): bit i of the mask = CU i in the driver's enumeration, which
// deals consecutive bits round-robin over the 8 XCDs (bit i -> XCD i % 8), so a contiguous bit range of a multiple of 8 takes the same number
// of CUs from every XCD.  first_cu / n_cus select bits [first_cu, first_cu + n_cus).  Used by generate_pipelined's "decode_cus" mode: the
// HBM-bound decode chain on a few CUs of its own beside the MFMA-bound prefill on the rest.
extern "C" int mc_stream_create_cu_range(int first_cu, int n_cus, void** stream) {
    if (!stream || first_cu < 0 || n_cus <= 0) { mc_set_error("mc_stream_create_cu_range: bad arguments"); return 1; }
    int cus = 0;
    if (mc_device_info(&cus, nullptr, nullptr, 0)) return 2;
    if (first_cu + n_cus > cus) { mc_set_error("mc_stream_create_cu_range: CUs [%d, %d) of %d", first_cu, first_cu + n_cus, cus); return 1; }
    uint32_t mask[32] = {0};
    const int words = (cus + 31) / 32;
    if (words > 32) { mc_set_error("mc_stream_create_cu_range: %d CUs", cus); return 1; }
    for (int i = first_cu; i < first_cu + n_cus; ++i) mask[i >> 5] |= 1u << (i & 31);
    hipStream_t s = nullptr;
    hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)words, mask);
    if (e != hipSuccess) { mc_set_error("hipExtStreamCreateWithCUMask: %s", hipGetErrorString(e)); return 2; }
    *stream = (void*)s;
    return 0;
}

extern "C" int mc_stream_destroy(void* stream) {
    if (!stream) return 0;
    hipError_t e = hipStreamDestroy((hipStream_t)stream);
    if (e != hipSuccess) { mc_set_error("hipStreamDestroy: %s", hipGetErrorString(e)); return 2; }
    return 0;
}

// ABI bookkeeping entry points of libthr_hip.so.
#include <string.h>

#include "thr_common.hpp"

extern "C" int thr_abi_version(void) { return THR_ABI_VERSION; }

extern "C" const char* thr_error_string(int code) {
    switch (code) {
        case THR_OK: return "ok";
        case THR_ERR_INVALID: return "invalid argument";
        case THR_ERR_UNSUPPORTED: return "unsupported shape";
        case THR_ERR_WORKSPACE: return "workspace too small";
        case THR_ERR_CAPACITY: return "on-chip capacity exceeded";
        default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown error";
    }
}

extern "C" int thr_device_info(int* h_compute_units, int64_t* h_hbm_bytes, char* h_arch,
                               int h_arch_len) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, dev);
    if (e != hipSuccess) return (int)e;
    if (h_compute_units) *h_compute_units = prop.multiProcessorCount;
    if (h_hbm_bytes) *h_hbm_bytes = (int64_t)prop.totalGlobalMem;
    if (h_arch && h_arch_len > 0) {
        strncpy(h_arch, prop.gcnArchName, (size_t)h_arch_len - 1);
        h_arch[h_arch_len - 1] = 0;
    }
    return THR_OK;
}

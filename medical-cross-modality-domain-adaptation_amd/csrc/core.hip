// core.hip — error channel, ABI version and device query of libpnp_hip.so.
#include <stdarg.h>
#include <string.h>
#include "pnp_common.h"

static thread_local char g_err[512] = "";

void pnp_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" {

int pnp_abi_version(void) { return 1; }

const char* pnp_last_error(void) { return g_err; }

int pnp_device_info(int device, int* cu_count, int* clock_khz, int* lds_bytes, char* arch, int arch_len) {
    hipDeviceProp_t p;
    hipError_t e = hipGetDeviceProperties(&p, device);
    if (e != hipSuccess) {
        pnp_set_error("pnp_device_info: %s", hipGetErrorString(e));
        return PNP_ELAUNCH;
    }
    if (cu_count) *cu_count = p.multiProcessorCount;
    if (clock_khz) *clock_khz = p.clockRate;
    if (lds_bytes) *lds_bytes = (int)p.maxSharedMemoryPerMultiProcessor;
    if (arch && arch_len > 0) {
        strncpy(arch, p.gcnArchName, (size_t)arch_len - 1);
        arch[arch_len - 1] = 0;
    }
    return PNP_OK;
}

}  // extern "C"

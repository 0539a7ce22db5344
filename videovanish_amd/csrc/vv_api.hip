// Host-side plumbing of the C ABI: error string, version, device probe.
#include <stdarg.h>
#include <string.h>
#include "vv_common.h"

static thread_local char g_err[512] = "";

void vv_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int vv_abi_version(void) { return VV_ABI_VERSION; }
extern "C" const char* vv_last_error(void) { return g_err; }

extern "C" int vv_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) VV_FAIL(VV_E_LAUNCH, "hipGetDeviceCount: %s", hipGetErrorString(e));
    return n;
}

extern "C" int vv_device_name(int dev, char* buf, int buflen) {
    hipDeviceProp_t prop;
    hipError_t e = hipGetDeviceProperties(&prop, dev);
    if (e != hipSuccess) VV_FAIL(VV_E_LAUNCH, "hipGetDeviceProperties(%d): %s", dev, hipGetErrorString(e));
    snprintf(buf, buflen, "%s|%s|cus=%d", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    return VV_OK;
}

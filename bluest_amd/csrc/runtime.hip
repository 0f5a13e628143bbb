// runtime.hip -- error state, ABI version, device queries of libbluest_hip.so (include/bluest_hip.h, top).
#include "common.hpp"

static thread_local std::string g_last_error;

int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

int require_gpu()
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        return fail(BLUEST_ERR_NOGPU, "no HIP device visible (hipGetDeviceCount: %s); libbluest_hip has no CPU path",
                    e == hipSuccess ? "0 devices" : hipGetErrorString(e));
    }
    return BLUEST_OK;
}

extern "C" int bluest_abi_version(void) { return BLUEST_ABI_VERSION; }
extern "C" const char *bluest_last_error(void) { return g_last_error.c_str(); }

extern "C" int bluest_device_count(int *count)
{
    if (!count) return fail(BLUEST_ERR_ARG, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { (void)hipGetLastError(); n = 0; }
    *count = n;
    return BLUEST_OK;
}

extern "C" int bluest_device_name(char *buf, int buflen)
{
    if (!buf || buflen <= 0) return fail(BLUEST_ERR_ARG, "bad buffer");
    int rc = require_gpu();
    if (rc) return rc;
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    hipDeviceProp_t p;
    HIP_TRY(hipGetDeviceProperties(&p, dev));
    snprintf(buf, buflen, "%s (%s, %d CUs)", p.name, p.gcnArchName, p.multiProcessorCount);
    return BLUEST_OK;
}


// Error reporting and device queries shared by the C-ABI entry points.
#include "common.h"
#include <cstring>

namespace mf {

static thread_local char g_error[512] = "";

char *error_buffer() { return g_error; }

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
    return code;
}

int device_cu_count()
{
    static thread_local int cached_dev = -1, cached = 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (dev != cached_dev) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
            n = 256;
        cached = n;
        cached_dev = dev;
    }
    return cached;
}

}  // namespace mf

extern "C" int mf_version(void) { return 1; }

extern "C" const char *mf_last_error_string(void) { return mf::error_buffer(); }

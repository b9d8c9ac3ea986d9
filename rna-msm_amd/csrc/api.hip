// Error plumbing, version and device probes of the C ABI.
#include "common.h"
#include <string.h>

namespace rnamsm {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
}  // namespace rnamsm

extern "C" int rnamsm_version(void) { return RNAMSM_VERSION; }
extern "C" const char* rnamsm_last_error(void) { return rnamsm::g_err; }
extern "C" int rnamsm_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

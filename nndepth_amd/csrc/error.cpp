// Thread-local error string + trivial entry points of the C-ABI.
#include "common.h"

namespace nnd {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace nnd

extern "C" {
int nnd_version(void) { return NND_VERSION; }
const char* nnd_last_error(void) { return nnd::g_err; }
int nnd_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}
}

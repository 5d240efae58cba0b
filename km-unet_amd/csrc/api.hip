// Version + thread-local error string of the C ABI (include/kmunet_hip.h).
#include "common.h"

namespace kmu {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace kmu

extern "C" int kmu_version(void) { return KMU_ABI_VERSION; }
extern "C" const char* kmu_last_error(void) { return kmu::g_err; }

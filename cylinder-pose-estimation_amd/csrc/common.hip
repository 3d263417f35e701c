// Error string + version for libcpe_hip.so.
#include "cpe_internal.h"
#include <stdarg.h>

namespace cpe {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace cpe

extern "C" int32_t cpe_version(void) { return CPE_VERSION; }
extern "C" const char *cpe_last_error_string(void) { return cpe::g_err; }

#include <stdarg.h>
#include "common.h"

namespace psn {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace psn

extern "C" const char* psn_last_error(void) { return psn::g_err; }
extern "C" int psn_version(void) { return 100; }

// Library-level entry points: ABI version and last-error text.
#include <stdarg.h>
#include <string.h>
#include "common.h"

namespace somi {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace somi

extern "C" int somi_abi_version(void) { return SOMI_ABI_VERSION; }
extern "C" const char *somi_last_error(void) { return somi::g_err; }

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
// sizeof of the descriptor structs as this library was compiled: a binding checks its own mirror against them (which = 0: somi_conv_desc,
// 1: somi_loss_desc)
extern "C" size_t somi_sizeof_desc(int which) { return which == 0 ? sizeof(somi_conv_desc) : which == 1 ? sizeof(somi_loss_desc) : 0; }

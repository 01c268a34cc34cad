// Library-level entry points: version and thread-local error string.
#include "common.h"
#include <stdarg.h>

namespace pp {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace pp

extern "C" int pp_version(void) { return 100; }  // 0.1.0
extern "C" const char* pp_last_error_string(void) { return pp::g_err; }

// ABI self-description so that foreign-language bindings can verify their struct layouts.
extern "C" int pp_sizeof_conv_args(void) { return (int)sizeof(pp_conv_args_t); }
extern "C" int pp_sizeof_attn_args(void) { return (int)sizeof(pp_attn_args_t); }

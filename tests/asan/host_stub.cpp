// Test infrastructure: the two runtime symbols the host parsers use (defined in gk_runtime.hip for the product),
// so that gk_bamread.cpp / gk_sampack.cpp / gk_textout.cpp can be built as a CPU-only library with
// AddressSanitizer + UndefinedBehaviorSanitizer (tests/test_sanitized_host.py).
#include <cstdarg>
#include <cstdio>

static thread_local char g_err[512] = "";

void gk_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* gk_last_error(void) { return g_err; }

// Test infrastructure: stands in for the GPU half of the search so that the HOST half (csrc/gk_hostsearch.cpp:
// gk_search_run -- first occurrences, cuts, stable ranking, the bounded and the exact branch -- and the site verdicts)
// can run on the CPU, under sanitizers, against the oracle (tests/test_host_search.py).  The four device entry points
// the host half calls are forwarded to callbacks the test registers (numpy implementations of their contracts in
// include/graphkir_hip.h); "device addresses" are then plain host addresses.  Built with `hipcc --cuda-host-only`.
#include <cstdarg>
#include <cstdio>

#include "gk_common.h"

static thread_local char g_err[512] = "";
void gk_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" {

const char* gk_last_error(void) { return g_err; }
hipError_t hipSetDevice(int) { return hipSuccess; }

typedef int (*maxsum_fn)(gk_dptr, int64_t, int64_t, const int32_t*, int32_t, int32_t, const int32_t*, int32_t, double*);
typedef int (*fraction_fn)(gk_dptr, int64_t, int64_t, const int32_t*, int32_t, int32_t, double*);
typedef int (*setsum_fn)(gk_dptr, int64_t, int64_t, const int32_t*, int32_t, int32_t, double*, double*);
typedef int (*bound_fn)(gk_dptr, int64_t, int64_t, gk_dptr, const int32_t*, int32_t, int32_t, const int32_t*, int32_t,
                        const uint8_t*, int32_t, int32_t, uint32_t*, int32_t*, uint32_t*);
static maxsum_fn g_maxsum;
static fraction_fn g_fraction;
static setsum_fn g_setsum;
static bound_fn g_bound;

void gk_test_device_calls(maxsum_fn a, fraction_fn b, setsum_fn c, bound_fn d) {
  g_maxsum = a; g_fraction = b; g_setsum = c; g_bound = d;
}

gk_ctx* gk_test_ctx(void) {
  static gk_ctx* ctx = new gk_ctx();
  return ctx;
}

int gk_maxsum(gk_ctx*, gk_dptr d_L, int64_t n_rows, int64_t ld, const int32_t* ids, int32_t n_sets, int32_t c_prev,
              const int32_t* cols, int32_t n_cols, double* out) {
  return g_maxsum(d_L, n_rows, ld, ids, n_sets, c_prev, cols, n_cols, out);
}
int gk_fraction(gk_ctx*, gk_dptr d_L, int64_t n_rows, int64_t ld, const int32_t* ids, int32_t n_sets, int32_t c,
                double* frac_out) {
  return g_fraction(d_L, n_rows, ld, ids, n_sets, c, frac_out);
}
int gk_setsum(gk_ctx*, gk_dptr d_L, int64_t n_rows, int64_t ld, const int32_t* ids, int32_t n_sets, int32_t c,
              double* value_out, double* frac_out) {
  return g_setsum(d_L, n_rows, ld, ids, n_sets, c, value_out, frac_out);
}
int gk_bound_step(gk_ctx*, gk_dptr d_miss8, int64_t ldm, int64_t n_rows, gk_dptr d_msum, const int32_t* ids,
                  int32_t n_sets, int32_t c_prev, const int32_t* cols, int32_t n_cols, const uint8_t* first,
                  int32_t top_n, int32_t cap, uint32_t* hdr_out, int32_t* idx_out, uint32_t* m_out) {
  return g_bound(d_miss8, ldm, n_rows, d_msum, ids, n_sets, c_prev, cols, n_cols, first, top_n, cap, hdr_out, idx_out, m_out);
}

}  // extern "C"

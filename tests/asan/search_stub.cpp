// Test infrastructure: stands in for the GPU half of the search so that the HOST half (csrc/gk_hostsearch.cpp:
// gk_search_run -- first occurrences, cuts, stable ranking, the bounded and the exact branch -- and the site verdicts)
// can run on the CPU, under sanitizers, against the oracle (tests/test_host_search.py).  The four device entry points
// the host half calls are forwarded to callbacks the test registers (numpy implementations of their contracts in
// include/graphkir_hip.h); "device addresses" are then plain host addresses.  Built with `hipcc --cuda-host-only`.
// The host half queues its device calls and collects them after a wait (gk_calls.h): here `enqueue` runs the callback
// at once and keeps the result, `collect` hands it over -- the same contract, without a stream.
#include <cstdarg>
#include <cstdio>

#include "gk_calls.h"
#include "gk_lut.h"

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

// ---- the two-phase forms (gk_calls.h) on the same callbacks
hipError_t gk_fetch_wait(gk_ctx*) { return hipSuccess; }
hipError_t gk_fetch_queue(gk_ctx*, void*, const void*, size_t) { return hipSuccess; }
hipError_t gk_fetch_mark(gk_ctx*, uint64_t* mark) { *mark = 0; return hipSuccess; }
hipError_t gk_fetch_wait_mark(gk_ctx*, uint64_t) { return hipSuccess; }
void gk_fetch_cancel(gk_ctx*) {}
void gk_pool_free(gk_ctx*, void*) {}

struct StubBound { uint32_t hdr[4]; std::vector<int32_t> idx; std::vector<uint32_t> mm; };
static thread_local StubBound t_bound;

int gk_bound_enqueue(gk_ctx* ctx, gk_dptr d_miss8, int64_t ldm, int64_t n_rows, gk_dptr d_msum, const int32_t* ids,
                     int32_t n_sets, int32_t c_prev, const int32_t* cols, int32_t n_cols, const uint8_t* first,
                     int32_t top_n, int32_t cap, GkBoundCall& call) {
  call.cap = cap;
  t_bound.idx.assign((size_t)cap, 0);
  t_bound.mm.assign((size_t)cap, 0);
  return gk_bound_step(ctx, d_miss8, ldm, n_rows, d_msum, ids, n_sets, c_prev, cols, n_cols, first, top_n, cap, t_bound.hdr,
                       t_bound.idx.data(), t_bound.mm.data());
}
void gk_bound_collect(gk_ctx*, GkBoundCall& call, uint32_t* hdr_out, int32_t* idx_out, uint32_t* m_out) {
  memcpy(hdr_out, t_bound.hdr, sizeof(t_bound.hdr));
  const uint32_t n = std::min<uint32_t>(t_bound.hdr[2], (uint32_t)call.cap);
  std::copy(t_bound.idx.begin(), t_bound.idx.begin() + n, idx_out);
  std::copy(t_bound.mm.begin(), t_bound.mm.begin() + n, m_out);
}

int gk_shares_enqueue(gk_ctx* ctx, const GkTable& L, int64_t n_rows, const int32_t* ids, int32_t n_sets, int32_t c,
                      bool with_value, GkSumCall& call) {
  const gk_dptr d_L = L.d;
  const int64_t ld = L.ld;
  call.n_sets = n_sets; call.c = c; call.with_value = with_value;
  call.back.assign((size_t)n_sets * (c + 1), 0.0);          // [values | shares]
  double* value = call.back.data();
  double* frac = call.back.data() + n_sets;
  return with_value ? gk_setsum(ctx, d_L, n_rows, ld, ids, n_sets, c, value, frac)
                    : gk_fraction(ctx, d_L, n_rows, ld, ids, n_sets, c, frac);
}
void gk_shares_collect(gk_ctx*, GkSumCall& call, double* value_out, double* frac_out) {
  if (value_out && call.with_value) std::copy(call.back.begin(), call.back.begin() + call.n_sets, value_out);
  if (frac_out) std::copy(call.back.begin() + call.n_sets, call.back.end(), frac_out);
}

int gk_colsum_enqueue(gk_ctx* ctx, const GkTable& L, int64_t n_rows, const int32_t* cols, int32_t n_cols,
                      GkSumCall& call) {
  call.back.assign((size_t)n_cols, 0.0);
  return gk_maxsum(ctx, L.d, n_rows, L.ld, nullptr, 1, 0, cols, n_cols, call.back.data());
}
void gk_colsum_collect(gk_ctx*, GkSumCall& call, double* out) { std::copy(call.back.begin(), call.back.end(), out); }
int gk_expand_table(gk_ctx*, const GkTable&, int64_t, int32_t, gk_dptr, int64_t) { return GK_ERR_NO_DEVICE; }   // index tables: device only
hipError_t gk_pool_malloc(gk_ctx*, void**, size_t) { return hipErrorOutOfMemory; }

// gk_sample_search's table phase has no CPU stand-in (it is the compatibility kernel): the entry points exist so that the
// library loads, and fail when called
extern "C" {
int gk_compat_log_miss(gk_ctx*, gk_tab*, gk_dptr, int64_t, gk_dptr, int32_t, int32_t, gk_dptr, int32_t, int32_t, int32_t,
                       gk_lut*, gk_dptr, gk_dptr, int64_t, gk_dptr) { gk_set_error("no device in the host-only build"); return GK_ERR_NO_DEVICE; }
int gk_compat_log(gk_ctx*, gk_tab*, gk_dptr, int64_t, gk_dptr, int32_t, int32_t, gk_dptr, int32_t, int32_t, int32_t, gk_lut*,
                  gk_dptr) { gk_set_error("no device in the host-only build"); return GK_ERR_NO_DEVICE; }
int gk_miss_colsum(gk_ctx*, gk_dptr, int64_t, int32_t, gk_dptr) { return GK_ERR_NO_DEVICE; }
int gk_compat_index(gk_ctx*, gk_tab*, gk_dptr, int64_t, gk_dptr, int32_t, int32_t, gk_dptr, int32_t, int32_t, int32_t, gk_lut*,
                    gk_dptr, gk_dptr, int64_t, gk_dptr) { return GK_ERR_NO_DEVICE; }
int gk_compat_patch(gk_ctx*, gk_lut*, gk_dptr, int64_t, int32_t, gk_dptr, int64_t, gk_dptr) { gk_set_error("no device in the host-only build"); return GK_ERR_NO_DEVICE; }
int gk_lut_known(gk_lut*, int32_t* n) { *n = 0; return GK_OK; }
int gk_lut_resolve(gk_lut*, gk_log10_fn, int32_t*, int32_t*, int32_t*) { return GK_ERR_NO_DEVICE; }
int gk_lut_resolve_stored(gk_lut*, gk_log10_fn, int32_t*, int32_t*, int32_t*) { return GK_ERR_NO_DEVICE; }
}

// Device calls of a search step in two halves: `enqueue` queues the parameter copy, the kernels and the copy of the
// result on the context's stream and returns at once; `collect`, after ONE stream synchronisation that may cover the
// calls of many genes (gk_fetch_wait), unpacks the result and releases the temporaries.  The blocking entry points of
// the C ABI (gk_bound_step, gk_setsum, gk_fraction, gk_maxsum) are enqueue + wait + collect.
#pragma once
#include <vector>

#include "gk_common.h"

// ---- gk_bound.hip: integer bound of a step (gk_bound_step)
struct GkBoundCall {
  std::vector<char> back;          // [SelState | idx[cap] | m[cap]] as fetched
  std::vector<void*> temps;        // pool blocks of the call
  int32_t cap = 0;
  size_t state_bytes = 0;
};
int gk_bound_enqueue(gk_ctx* ctx, gk_dptr d_miss8, int64_t ldm, int64_t n_rows, gk_dptr d_msum, const int32_t* ids,
                     int32_t n_sets, int32_t c_prev, const int32_t* cols, int32_t n_cols, const uint8_t* first,
                     int32_t top_n, int32_t cap, GkBoundCall& call);
// hdr_out[4] = candidates, cut, selected, 0; idx_out / m_out hold min(selected, cap) entries
void gk_bound_collect(gk_ctx* ctx, GkBoundCall& call, uint32_t* hdr_out, int32_t* idx_out, uint32_t* m_out);

// ---- gk_search.hip: exact float64 sums with numpy's tree
// A gene's table of log-likelihoods, column-major [allele][ld]: float64 values (vals == nullptr), or uint16 dense
// indices into the value table's array `vals` (the index form, gk_compat_index).
struct GkTable {
  gk_dptr d = 0;
  int64_t ld = 0;
  const double* vals = nullptr;
  bool indexed() const { return vals != nullptr; }
};
struct GkSumCall {
  std::vector<double> back;        // results as fetched (per set: c shares [+ value]; or one sum per column)
  std::vector<void*> temps;
  int64_t n_rows = 0;
  int32_t n_sets = 0, c = 0;
  bool with_value = false;
  bool leafwise = false;           // served by setsum_leaves (every column through LDS once), else by tiles of sets
};
// value + shares (value_out != nullptr at collect) or shares only of the given sets (gk_setsum / gk_fraction)
int gk_shares_enqueue(gk_ctx* ctx, const GkTable& L, int64_t n_rows, const int32_t* ids, int32_t n_sets, int32_t c,
                      bool with_value, GkSumCall& call);
void gk_shares_collect(gk_ctx* ctx, GkSumCall& call, double* value_out, double* frac_out);
// column sums log_probs[:, cols].sum(axis=0) (gk_maxsum with no previous sets)
int gk_colsum_enqueue(gk_ctx* ctx, const GkTable& L, int64_t n_rows, const int32_t* cols, int32_t n_cols,
                      GkSumCall& call);
// the float64 form of an index table (for the exact (max,+) kernel): d_L double [n_allele][ld], queued on the stream
int gk_expand_table(gk_ctx* ctx, const GkTable& L, int64_t n_rows, int32_t n_allele, gk_dptr d_L, int64_t ld);
void gk_colsum_collect(gk_ctx* ctx, GkSumCall& call, double* out);

static inline void gk_release(gk_ctx* ctx, std::vector<void*>& temps) {
  for (void* p : temps) gk_pool_free(ctx, p);
  temps.clear();
}

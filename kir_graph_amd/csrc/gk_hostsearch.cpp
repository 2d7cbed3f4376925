// Host half of the greedy multi-allele search, native: AlleleTyping.addCandidate for every copy-number step of
// one gene (typing_mulit_allele.py:478-598), plus the per-position homozygosity verdict (835-857).
//
// The device side of a step is gk_maxsum / gk_bound_step / gk_setsum / gk_fraction; what the reference does
// around them -- first occurrences of allele multisets (uniqueAllele 456-476), the sort of the scores and the
// top_n cut (567), per-allele sums (571), the stable three-key ranking (rankScore 202-214) -- ran as numpy
// under the interpreter lock of the calling process.  Here it runs in the calling thread without that lock,
// so the gene threads of ONE process keep the GPU busy.
//
// Ranks must be the reference's, ties included: wherever the reference calls numpy.argsort (an unstable sort
// whose tie order is numpy's own), this code calls back into the host language (gk_argsort_fn: the Python
// binding passes numpy.argsort itself; a few hundred values per gene, or the whole score table in the rare
// exact redo); stable orders (numpy.lexsort, Python sorted) are std::stable_sort with the same keys.
// Reductions over the <= 8 alleles of a set follow numpy's add.reduce for a contiguous inner axis
// (sequential below 8 elements, the 8-accumulator tree at 8).
#include <algorithm>
#include <cmath>
#include <memory>
#include <numeric>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <chrono>
#include <sys/syscall.h>
#include <unistd.h>

#include "gk_calls.h"
#include "gk_lut.h"

namespace {

struct Step {
  int n = 0;                 // alleles per set
  int bounded = 0;           // 1: served by the integer bound, 0: float64 sums for every candidate
  std::vector<double> value, sum_indv, frac;
  std::vector<int32_t> ids;  // [rows][n]
  int64_t rows() const { return (int64_t)value.size(); }
};

// numpy add.reduce over a contiguous run of n <= 8 doubles
inline double sum_small(const double* a, int n) {
  if (n == 8) return ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
  double r = 0.0;
  for (int i = 0; i < n; ++i) r += a[i];
  return r;
}

struct Key3 { double k1, k2, k3; };

// np.lexsort((k3, k2, k1)): stable, by k1 then k2 then k3
void lexsort3(const std::vector<Key3>& k, std::vector<int64_t>& order) {
  order.resize(k.size());
  std::iota(order.begin(), order.end(), 0);
  std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t b) {
    if (k[a].k1 != k[b].k1) return k[a].k1 < k[b].k1;
    if (k[a].k2 != k[b].k2) return k[a].k2 < k[b].k2;
    return k[a].k3 < k[b].k3;
  });
}

// First-occurrence mask over the candidates prev[t] + [cols[a]] in (t-major, a-minor) order
// (uniqueAllele 456-476 on the stacked id table).  One- and two-allele previous sets are answered from
// position tables: candidate (t, a) repeats an earlier one exactly when swapping the new allele with a member e
// of prev[t] gives a previous set that sits earlier in the list (with e itself among the offered alleles), or
// when prev[t] already occurred earlier.  Larger sets are hashed.
void first_of_sets(const int32_t* prev, int T, int k, const int32_t* cols, int A, int n_allele,
                   std::vector<uint8_t>& first) {
  first.assign((size_t)T * A, 0);
  bool unique_cols = true;
  {
    std::vector<char> seen((size_t)n_allele, 0);
    for (int a = 0; a < A && unique_cols; ++a) {
      if (seen[cols[a]]) unique_cols = false;
      seen[cols[a]] = 1;
    }
  }
  if (k <= 2 && n_allele <= 2048 && unique_cols) {
    std::vector<char> offered((size_t)n_allele, 0);
    for (int a = 0; a < A; ++a) offered[cols[a]] = 1;
    if (k == 1) {
      std::vector<int32_t> pos((size_t)n_allele, T);
      for (int t = T - 1; t >= 0; --t) pos[prev[t]] = t;
      for (int t = 0; t < T; ++t) {
        const int p = prev[t];
        if (pos[p] != t) continue;
        const bool free_pass = !offered[p];
        uint8_t* row = first.data() + (size_t)t * A;
        for (int a = 0; a < A; ++a) row[a] = (free_pass || pos[cols[a]] >= t) ? 1 : 0;
      }
      return;
    }
    std::vector<int32_t> pos((size_t)n_allele * n_allele, T);
    for (int t = T - 1; t >= 0; --t) {
      const int lo = std::min(prev[2 * t], prev[2 * t + 1]), hi = std::max(prev[2 * t], prev[2 * t + 1]);
      pos[(size_t)lo * n_allele + hi] = t;
      pos[(size_t)hi * n_allele + lo] = t;
    }
    for (int t = 0; t < T; ++t) {
      const int lo = std::min(prev[2 * t], prev[2 * t + 1]), hi = std::max(prev[2 * t], prev[2 * t + 1]);
      if (pos[(size_t)lo * n_allele + hi] != t) continue;
      const int32_t* by_hi = pos.data() + (size_t)hi * n_allele;   // pairs {hi, x}
      const int32_t* by_lo = pos.data() + (size_t)lo * n_allele;   // pairs {lo, x}
      const bool lo_off = offered[lo], hi_off = offered[hi];
      uint8_t* row = first.data() + (size_t)t * A;
      for (int a = 0; a < A; ++a) {
        const int c = cols[a];
        const bool dup = (by_hi[c] < t && lo_off) || (by_lo[c] < t && hi_off);
        row[a] = dup ? 0 : 1;
      }
    }
    return;
  }
  // general: hash the sorted multisets in list order
  std::unordered_set<std::string> seen;
  seen.reserve((size_t)T * A * 2);
  std::vector<int32_t> key((size_t)k + 1);
  for (int t = 0; t < T; ++t)
    for (int a = 0; a < A; ++a) {
      std::copy(prev + (size_t)t * k, prev + (size_t)(t + 1) * k, key.begin());
      key[k] = cols[a];
      std::sort(key.begin(), key.end());
      std::string s((const char*)key.data(), key.size() * sizeof(int32_t));
      first[(size_t)t * A + a] = seen.insert(std::move(s)).second ? 1 : 0;
    }
}

// the ranking tail shared by both kinds of step: rows (ids, value, frac for the contenders) -> first top_n rows
// under the stable order (-value, -sum of per-allele sums, unevenness).  `contend` indexes into the head rows.
struct Head {
  int c = 0;
  std::vector<int32_t> ids;        // [rows][c]
  std::vector<double> value;       // [rows]
  std::vector<double> sum_indv;    // [rows][c]
  std::vector<double> key2;        // -sum(sum_indv)
};

void fill_sums(Head& h, const double* colsum) {
  const int64_t n = (int64_t)h.value.size();
  h.sum_indv.resize((size_t)n * h.c);
  h.key2.resize((size_t)n);
  for (int64_t i = 0; i < n; ++i) {
    double* s = h.sum_indv.data() + (size_t)i * h.c;
    for (int j = 0; j < h.c; ++j) s[j] = colsum[h.ids[(size_t)i * h.c + j]];
    h.key2[i] = -sum_small(s, h.c);
  }
}

// rows not worse than the top_n-th row on (-value, key2): only they can make the cut (unevenness is the last key)
void contenders(const Head& h, int top_n, std::vector<int64_t>& contend) {
  const int64_t n = (int64_t)h.value.size();
  contend.clear();
  if (n <= top_n) {
    contend.resize((size_t)n);
    std::iota(contend.begin(), contend.end(), 0);
    return;
  }
  std::vector<int64_t> order((size_t)n);
  std::iota(order.begin(), order.end(), 0);
  std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t b) {
    const double ka = -h.value[a], kb = -h.value[b];
    if (ka != kb) return ka < kb;
    return h.key2[a] < h.key2[b];
  });
  const int64_t b = order[top_n - 1];
  const double b1 = -h.value[b], b2 = h.key2[b];
  for (int64_t i = 0; i < n; ++i) {
    const double k1 = -h.value[i];
    if (k1 < b1 || (k1 == b1 && h.key2[i] <= b2)) contend.push_back(i);
  }
}

double unevenness(const double* f, int c) {
  const double mean = sum_small(f, c) / (double)c;
  double d[8];
  for (int j = 0; j < c; ++j) d[j] = std::fabs(f[j] - mean);
  return sum_small(d, c);
}

}  // namespace

struct gk_search {
  int n_allele = 0;
  std::vector<double> colsum;
  std::vector<Step> steps;
  // launch geometries for the roofline accounting (kir_graph_amd/roofmodel.py): 7 numbers per device call --
  // kind (0 maxsum / column sums, 1 minsum, 2 set sums by tiles, 3 set sums leaf by leaf), then the model's arguments
  std::vector<int64_t> log;
  int64_t distinct(const int32_t* ids, size_t n) const {
    std::vector<char> seen((size_t)n_allele, 0);
    int64_t d = 0;
    for (size_t i = 0; i < n; ++i)
      if (!seen[ids[i]]) { seen[ids[i]] = 1; ++d; }
    return d;
  }
  void note(int64_t kind, int64_t a, int64_t b, int64_t c, int64_t d, int64_t e, int64_t f) {
    const int64_t row[7] = {kind, a, b, c, d, e, f};
    log.insert(log.end(), row, row + 7);
  }
};

extern "C" {

int gk_maxsum(gk_ctx* ctx, gk_dptr d_L, int64_t n_rows, int64_t ld, const int32_t* ids, int32_t n_sets, int32_t c_prev,
              const int32_t* cols, int32_t n_cols, double* out);
int gk_fraction(gk_ctx* ctx, gk_dptr d_L, int64_t n_rows, int64_t ld, const int32_t* ids, int32_t n_sets, int32_t c,
                double* frac_out);
int gk_compat_log_miss(gk_ctx* ctx, gk_tab* tab, gk_dptr d_rows, int64_t n_rows, gk_dptr d_vflag, int32_t vbeg,
                       int32_t vend, gk_dptr d_mask, int32_t words, int32_t n_allele, int32_t keep_empty, gk_lut* lut,
                       gk_dptr d_log, gk_dptr d_miss8, int64_t ldm, gk_dptr d_flags);
int gk_compat_log(gk_ctx* ctx, gk_tab* tab, gk_dptr d_rows, int64_t n_rows, gk_dptr d_vflag, int32_t vbeg,
                  int32_t vend, gk_dptr d_mask, int32_t words, int32_t n_allele, int32_t keep_empty, gk_lut* lut,
                  gk_dptr d_log);
int gk_miss_colsum(gk_ctx* ctx, gk_dptr d_miss8, int64_t ldm, int32_t n_cols, gk_dptr d_msum);
int gk_compat_patch(gk_ctx* ctx, gk_lut* lut, gk_dptr d_log, int64_t n_rows, int32_t n_allele, gk_dptr d_miss8, int64_t ldm,
                    gk_dptr d_flags);
int gk_compat_index(gk_ctx* ctx, gk_tab* tab, gk_dptr d_rows, int64_t n_rows, gk_dptr d_vflag, int32_t vbeg, int32_t vend,
                    gk_dptr d_mask, int32_t words, int32_t n_allele, int32_t keep_empty, gk_lut* lut, gk_dptr d_lidx,
                    gk_dptr d_miss8, int64_t ldm, gk_dptr d_flags);

}  // extern "C"

namespace {

// The search of ONE gene as a sequence of phases, each cut where the host needs something from the device:
//   colsums   -> (wait) -> first step
//   per further step:  prepare (first occurrences) -> bound enqueue -> (wait) -> selection -> set sums enqueue -> (wait)
//                      -> ranking; a step the bound cannot serve is done with float64 sums for every candidate (blocking).
// gk_search_run drives one of these with a wait after every enqueue; gk_sample_search drives all genes of a sample in
// lock-step, so that ONE wait covers the step of every gene (typing_mulit_allele.py:478-598 per gene).
struct GeneSearch {
  gk_ctx* ctx = nullptr;
  GkTable table;                // the gene's log-likelihoods as the sums read them: float64, or indices + value array
  gk_dptr d_L = 0;              // the float64 form (the exact (max,+) kernel streams it); 0 in index form until needed
  void* expanded = nullptr;     // pool block behind d_L when this search made it from the index form
  gk_dptr d_miss8 = 0, d_msum = 0;
  int64_t n_rows = 0, ld = 0, ldm = 0;
  int n_allele = 0, n_steps = 0, T = 0, A = 0;
  std::vector<int32_t> cols;    // the alleles offered at the step in hand
  // exon-first (typing_mulit_allele.py:740-746): every step of a candidate search offers the alleles of ONE exon group;
  // empty = every step offers `cols` as given to init
  std::vector<std::vector<int32_t>> step_cols;
  gk_argsort_fn argsort = nullptr;
  bool bound = false, unique_cols = true;
  std::unique_ptr<gk_search> S;
  // state of the step in flight
  std::vector<uint8_t> first;
  int64_t N = 0;
  GkBoundCall bcall;
  GkSumCall scall;
  Head sel;                     // sets selected by the bound, list order
  Step st;
  bool bound_in_flight = false, sums_in_flight = false, step_done = false;

  int init(gk_ctx* c, const GkTable& tbl, gk_dptr L, int64_t rows, int64_t ld_, int32_t n_allele_, gk_dptr miss8, int64_t ldm_,
           gk_dptr msum, const int32_t* cols_, int32_t n_cols, int32_t n_steps_, int32_t top_n, gk_argsort_fn fn) {
    GK_REQUIRE(c && tbl.d && (L || tbl.indexed()) && cols_ && fn, "null pointer");
    table = tbl;
    GK_REQUIRE(rows > 0 && ld_ >= rows && n_allele_ > 0 && n_cols > 0 && n_steps_ >= 0 && n_steps_ <= 8 && top_n >= 1,
               "bad search arguments");
    for (int a = 0; a < n_cols; ++a) GK_REQUIRE(cols_[a] >= 0 && cols_[a] < n_allele_, "candidate allele out of range");
    ctx = c; d_L = L; n_rows = rows; ld = ld_; n_allele = n_allele_; d_miss8 = miss8; ldm = ldm_; d_msum = msum;
    cols.assign(cols_, cols_ + n_cols);
    n_steps = n_steps_; T = top_n; A = n_cols; argsort = fn;
    bound = d_miss8 != 0 && d_msum != 0;
    S.reset(new gk_search());
    S->n_allele = n_allele;
    S->colsum.resize((size_t)n_allele);
    std::vector<char> seen((size_t)n_allele, 0);
    for (int a = 0; a < A && unique_cols; ++a) { if (seen[cols[a]]) unique_cols = false; seen[cols[a]] = 1; }
    return GK_OK;
  }

  // the columns of step `s` (0-based) become the ones in hand
  int use_step(int s) {
    if (step_cols.empty()) return GK_OK;
    const std::vector<int32_t>& c = step_cols[std::min<size_t>((size_t)s, step_cols.size() - 1)];
    GK_REQUIRE(!c.empty(), "a search step without candidate alleles");
    cols = c;
    A = (int)cols.size();
    unique_cols = true;
    std::vector<char> seen((size_t)n_allele, 0);
    for (int a = 0; a < A; ++a) {
      GK_REQUIRE(cols[a] >= 0 && cols[a] < n_allele, "candidate allele out of range");
      if (seen[cols[a]]) unique_cols = false;
      seen[cols[a]] = 1;
    }
    return GK_OK;
  }

  // ---- per-allele column sums = log_probs.sum(axis=0) (line 514), needed by every step (571)
  int colsum_enqueue() {
    std::vector<int32_t> every((size_t)n_allele);
    std::iota(every.begin(), every.end(), 0);
    int rc = gk_colsum_enqueue(ctx, table, n_rows, every.data(), n_allele, scall);
    if (rc) return rc;
    S->note(0, n_rows, 1, 0, n_allele, 0, 0);
    return GK_OK;
  }
  void colsum_collect() { gk_colsum_collect(ctx, scall, S->colsum.data()); }

  // ---- first allele (512-532): argsort(score)[::-1][:top_n]
  int first_step() {
    { const int r = use_step(0); if (r) return r; }
    const double* colsum = S->colsum.data();
    std::vector<double> score((size_t)A);
    for (int a = 0; a < A; ++a) score[a] = colsum[cols[a]];
    std::vector<int64_t> order((size_t)A);
    // numpy.argsort is an unstable sort: its order among EQUAL values is numpy's own, so it is asked (a call back
    // into the host language, which may have to wait for the interpreter lock) only when equal values exist;
    // distinct values have one ascending order, whoever sorts them.  NaN (a table read before its last pass) never
    // reaches std::sort: `<` is no strict weak order with it
    bool plain = true;
    for (int a = 0; a < A && plain; ++a) plain = score[a] == score[a];
    if (plain) {
      std::iota(order.begin(), order.end(), 0);
      std::sort(order.begin(), order.end(), [&](int64_t x, int64_t y) { return score[x] != score[y] ? score[x] < score[y] : x < y; });
      for (int a = 1; a < A && plain; ++a) plain = score[order[a - 1]] != score[order[a]];   // no tie
    }
    if (!plain) GK_REQUIRE(argsort(score.data(), A, order.data()) == 0, "host argsort failed");
    Step s1;
    s1.n = 1;
    const int keep = std::min(T, A);
    for (int i = 0; i < keep; ++i) {
      const int64_t a = order[(size_t)A - 1 - i];
      GK_REQUIRE(a >= 0 && a < A, "host argsort returned an index out of range");
      s1.value.push_back(score[a]);
      s1.sum_indv.push_back(score[a]);
      s1.ids.push_back(cols[a]);
      s1.frac.push_back(1.0);
    }
    S->steps.push_back(std::move(s1));
    return GK_OK;
  }

  bool more() const { return (int)S->steps.size() < n_steps; }

  // ---- a further step: first occurrences, then the integer bound when it applies
  int step_begin() {
    { const int r = use_step((int)S->steps.size()); if (r) return r; }
    const Step& prev = S->steps.back();
    const int k = prev.n;
    const int Tp = (int)prev.rows();
    first_of_sets(prev.ids.data(), Tp, k, cols.data(), A, n_allele, first);
    N = 0;
    for (uint8_t f : first) N += f;
    st = Step();
    st.n = k + 1;
    step_done = bound_in_flight = sums_in_flight = false;
    if (bound && N > 0 && unique_cols) {
      const int cap = 4 * T + 4096;
      int rc = gk_bound_enqueue(ctx, d_miss8, ldm, n_rows, d_msum, prev.ids.data(), Tp, k, cols.data(), A, first.data(), T,
                                cap, bcall);
      if (rc) return rc;
      S->note(1, n_rows, Tp, A, k == 1 ? S->distinct(prev.ids.data(), prev.ids.size()) : Tp, 0, 0);
      bound_in_flight = true;
    }
    return GK_OK;
  }

  // the selection is back: exact sums for the sets that can reach the cut
  int after_bound() {
    if (!bound_in_flight) return GK_OK;
    bound_in_flight = false;
    const Step& prev = S->steps.back();
    const int k = prev.n, c = k + 1;
    const int cap = bcall.cap;
    uint32_t hdr[4];
    std::vector<int32_t> idx((size_t)cap);
    std::vector<uint32_t> mm((size_t)cap);
    gk_bound_collect(ctx, bcall, hdr, idx.data(), mm.data());
    const int64_t n_sel = hdr[2];
    if (n_sel <= 0 || n_sel > cap) return GK_OK;                // a flood of ties at the cut: exact step
    idx.resize((size_t)n_sel);
    std::sort(idx.begin(), idx.end());                         // list order of the candidates
    sel = Head();
    sel.c = c;
    sel.ids.resize((size_t)n_sel * c);
    for (int64_t i = 0; i < n_sel; ++i) {
      const int t = idx[i] / A, a = idx[i] % A;
      std::copy(prev.ids.begin() + (size_t)t * k, prev.ids.begin() + (size_t)(t + 1) * k, sel.ids.begin() + (size_t)i * c);
      sel.ids[(size_t)i * c + k] = cols[a];
    }
    int rc = gk_shares_enqueue(ctx, table, n_rows, sel.ids.data(), (int32_t)n_sel, c, true, scall);
    if (rc) return rc;
    S->note(scall.leafwise ? 3 : 2, n_rows, n_sel, c, S->distinct(sel.ids.data(), sel.ids.size()), 0, 0);
    sums_in_flight = true;
    return GK_OK;
  }

  // exact values + shares of the selection are back: the reference's head, the ranking, the ambiguity tests
  int after_sums() {
    if (!sums_in_flight) return GK_OK;
    sums_in_flight = false;
    const double* colsum = S->colsum.data();
    const int c = sel.c;
    const int64_t n_sel = (int64_t)sel.ids.size() / c;
    std::vector<double> value((size_t)n_sel), frac((size_t)n_sel * c);
    gk_shares_collect(ctx, scall, value.data(), frac.data());
    // the reference's head: rows of the sorted table that reach the top_n-th value (567, then the cut to top_n)
    bool ok = true;
    std::vector<int64_t> head;
    const int64_t n_top = std::min<int64_t>(std::max<int64_t>(T, N / 5), N);
    for (double v : value)
      if (v != v) return GK_OK;          // a NaN sum (cannot happen on a settled table): the exact step, which asks numpy
    if (N > T) {
      std::vector<double> tmp(value);
      std::nth_element(tmp.begin(), tmp.begin() + (T - 1), tmp.end(), [](double x, double y) { return x > y; });
      const double v_cut = tmp[(size_t)T - 1];
      for (int64_t i = 0; i < n_sel; ++i)
        if (value[i] >= v_cut) head.push_back(i);
      if (n_top > T) ok = (int64_t)head.size() <= n_top;   // ties running past the N // 5 cut
      else ok = (int64_t)head.size() == T;                 // ties across the top_n cut
    } else {
      head.resize((size_t)n_sel);
      std::iota(head.begin(), head.end(), 0);
    }
    if (!ok) return GK_OK;
    Head hh;
    hh.c = c;
    for (int64_t i : head) {
      hh.value.push_back(value[i]);
      hh.ids.insert(hh.ids.end(), sel.ids.begin() + (size_t)i * c, sel.ids.begin() + (size_t)(i + 1) * c);
    }
    fill_sums(hh, colsum);
    std::vector<int64_t> contend;
    contenders(hh, T, contend);
    std::vector<Key3> keys(contend.size());
    for (size_t q = 0; q < contend.size(); ++q) {
      const int64_t i = contend[q];
      keys[q] = Key3{-hh.value[i], hh.key2[i], unevenness(frac.data() + (size_t)head[i] * c, c)};
    }
    std::vector<int64_t> sub;
    lexsort3(keys, sub);
    const size_t look = std::min<size_t>(sub.size(), (size_t)T + 1);
    for (size_t q = 1; q < look && ok; ++q) {
      const Key3 &x = keys[sub[q - 1]], &y = keys[sub[q]];
      if (x.k1 == y.k1 && x.k2 == y.k2 && x.k3 == y.k3) ok = false;   // their order would be argsort's
    }
    if (!ok) return GK_OK;
    const size_t keep = std::min<size_t>(sub.size(), (size_t)T);
    for (size_t q = 0; q < keep; ++q) {
      const int64_t i = contend[sub[q]];
      st.value.push_back(hh.value[i]);
      st.ids.insert(st.ids.end(), hh.ids.begin() + (size_t)i * c, hh.ids.begin() + (size_t)(i + 1) * c);
      st.sum_indv.insert(st.sum_indv.end(), hh.sum_indv.begin() + (size_t)i * c, hh.sum_indv.begin() + (size_t)(i + 1) * c);
      const double* f = frac.data() + (size_t)head[i] * c;
      st.frac.insert(st.frac.end(), f, f + c);
    }
    st.bounded = 1;
    step_done = true;
    return GK_OK;
  }

  // ---- float64 sums for every candidate (540-598 as written); blocking
  int exact_step() {
    const Step& prev = S->steps.back();
    const double* colsum = S->colsum.data();
    const int k = prev.n, c = k + 1;
    const int Tp = (int)prev.rows();
    int rc = GK_OK;
    if (!d_L) {     // index form: the (max,+) kernel streams float64 operands -- make them once for this gene
      GK_HIP(gk_pool_malloc(ctx, &expanded, (size_t)n_allele * (size_t)n_rows * sizeof(double)));
      rc = gk_expand_table(ctx, table, n_rows, n_allele, gk_addr(expanded), n_rows);
      if (rc) return rc;
      d_L = gk_addr(expanded);
      ld = n_rows;
    }
    std::vector<double> scores((size_t)Tp * A);
    rc = gk_maxsum(ctx, d_L, n_rows, ld, prev.ids.data(), Tp, k, cols.data(), A, scores.data());
    if (rc) return rc;
    {
      const int64_t d_prev = S->distinct(prev.ids.data(), prev.ids.size());
      const bool symmetric = k == 1 && Tp == A && Tp > 32 && d_prev == Tp && unique_cols;
      S->note(0, n_rows, Tp, k, A, k == 1 ? d_prev : Tp, symmetric ? 1 : 0);
    }
    std::vector<int64_t> where;            // flat index of the first occurrences, list order
    std::vector<double> score;
    where.reserve((size_t)N);
    score.reserve((size_t)N);
    for (int64_t i = 0; i < (int64_t)first.size(); ++i)
      if (first[i]) { where.push_back(i); score.push_back(scores[i]); }
    const int64_t n_keep = std::max<int64_t>(T, N / 5);
    std::vector<int64_t> order((size_t)N);
    if (N) GK_REQUIRE(argsort(score.data(), N, order.data()) == 0, "host argsort failed");
    std::vector<int64_t> top;              // argsort(score)[::-1][:n_keep]
    for (int64_t i = 0; i < std::min<int64_t>(n_keep, N); ++i) top.push_back(order[(size_t)N - 1 - i]);
    int64_t head = (int64_t)top.size();
    if ((int64_t)top.size() > T) {
      const double v_cut = score[top[(size_t)T - 1]];
      head = 0;
      for (int64_t i : top) head += score[i] >= v_cut;      // value is descending: a prefix
    }
    top.resize((size_t)head);
    Head hh;
    hh.c = c;
    for (int64_t i : top) {
      const int64_t flat = where[i];
      const int t = (int)(flat / A), a = (int)(flat % A);
      hh.value.push_back(score[i]);
      hh.ids.insert(hh.ids.end(), prev.ids.begin() + (size_t)t * k, prev.ids.begin() + (size_t)(t + 1) * k);
      hh.ids.push_back(cols[a]);
    }
    fill_sums(hh, colsum);
    std::vector<int64_t> contend;
    contenders(hh, T, contend);
    std::vector<int32_t> cids(contend.size() * c);
    for (size_t q = 0; q < contend.size(); ++q)
      std::copy(hh.ids.begin() + (size_t)contend[q] * c, hh.ids.begin() + (size_t)(contend[q] + 1) * c,
                cids.begin() + q * c);
    std::vector<double> frac(contend.size() * c);
    if (!contend.empty()) {
      rc = gk_fraction(ctx, d_L, n_rows, ld, cids.data(), (int32_t)contend.size(), c, frac.data());
      if (rc) return rc;
      S->note(2, n_rows, (int64_t)contend.size(), c, S->distinct(cids.data(), cids.size()), 0, 0);
    }
    std::vector<Key3> keys(contend.size());
    for (size_t q = 0; q < contend.size(); ++q)
      keys[q] = Key3{-hh.value[contend[q]], hh.key2[contend[q]], unevenness(frac.data() + q * c, c)};
    std::vector<int64_t> sub;
    lexsort3(keys, sub);
    const size_t keep = std::min<size_t>(sub.size(), (size_t)T);
    st = Step();
    st.n = c;
    for (size_t q = 0; q < keep; ++q) {
      const int64_t i = contend[sub[q]];
      st.value.push_back(hh.value[i]);
      st.ids.insert(st.ids.end(), hh.ids.begin() + (size_t)i * c, hh.ids.begin() + (size_t)(i + 1) * c);
      st.sum_indv.insert(st.sum_indv.end(), hh.sum_indv.begin() + (size_t)i * c, hh.sum_indv.begin() + (size_t)(i + 1) * c);
      st.frac.insert(st.frac.end(), frac.begin() + sub[q] * c, frac.begin() + (sub[q] + 1) * c);
    }
    step_done = true;
    return GK_OK;
  }

  void step_end() { S->steps.push_back(std::move(st)); }

  // calls still queued when an error ends the run: their temporaries go back to the pool
  void abandon() {
    gk_release(ctx, bcall.temps);
    gk_release(ctx, scall.temps);
    finish();
  }
  // the float64 copy this search made for itself (every kernel that reads it has been waited for)
  void finish() {
    if (expanded) { gk_pool_free(ctx, expanded); expanded = nullptr; d_L = 0; }
  }
};

int wait_stream(gk_ctx* ctx) {
  if (gk_fetch_wait(ctx) != hipSuccess) {
    gk_set_error("search: waiting for the stream failed: %s", hipGetErrorString(hipGetLastError()));
    return GK_ERR_HIP;
  }
  return GK_OK;
}

int wait_mark(gk_ctx* ctx, uint64_t mark) {
  if (gk_fetch_wait_mark(ctx, mark) != hipSuccess) {
    gk_set_error("search: waiting for a mark of the stream failed: %s", hipGetErrorString(hipGetLastError()));
    return GK_ERR_HIP;
  }
  return GK_OK;
}

int set_mark(gk_ctx* ctx, uint64_t* mark) {
  if (gk_fetch_mark(ctx, mark) != hipSuccess) {
    gk_set_error("search: recording a mark on the stream failed: %s", hipGetErrorString(hipGetLastError()));
    return GK_ERR_HIP;
  }
  return GK_OK;
}

// GK_TRACE=search: where the calling thread spends a gk_sample_search call (host work between the waits / the waits)
struct SearchClock {
  using clk = std::chrono::steady_clock;
  const bool on;
  const char* form;
  int genes;
  double host = 0, wait = 0;
  int n_wait = 0;
  clk::time_point mark = clk::now();
  const clk::time_point begin = mark;
  SearchClock(const char* f, int g) : on(gk_trace("search")), form(f), genes(g) {}
  void lap(bool waited) {
    if (!on) return;
    const auto now = clk::now();
    (waited ? wait : host) += std::chrono::duration<double, std::milli>(now - mark).count();
    n_wait += waited ? 1 : 0;
    mark = now;
  }
  ~SearchClock() {
    // begin / end on the monotonic clock (Python's time.perf_counter on Linux), for timelines across threads
    if (on) fprintf(stderr, "[gk_sample_search %s] thread %ld, %d genes: host %.2f ms, waits %.2f ms in %d waits; from %.6f to %.6f s\n", form,
                    (long)syscall(SYS_gettid), genes, host, wait, n_wait, std::chrono::duration<double>(begin.time_since_epoch()).count(),
                    std::chrono::duration<double>(clk::now().time_since_epoch()).count());
  }
};

}  // namespace

extern "C" {

int gk_search_run(gk_ctx* ctx, gk_dptr d_L, int64_t n_rows, int64_t ld, int32_t n_allele, gk_dptr d_miss8, int64_t ldm,
                  gk_dptr d_msum, const int32_t* cols, int32_t n_cols, int32_t n_steps, int32_t top_n,
                  gk_argsort_fn argsort, const double* colsum_in, gk_search** out) {
  gk_bind(ctx);
  GK_REQUIRE(out, "null pointer");
  GeneSearch g;
  int rc = g.init(ctx, GkTable{d_L, ld, nullptr}, d_L, n_rows, ld, n_allele, d_miss8, ldm, d_msum, cols, n_cols, n_steps, top_n,
                  argsort);
  if (rc) return rc;
  if (colsum_in) {
    std::copy(colsum_in, colsum_in + n_allele, g.S->colsum.begin());
  } else {
    rc = g.colsum_enqueue();
    if (rc == GK_OK) rc = wait_stream(ctx);
    if (rc) { g.abandon(); return rc; }
    g.colsum_collect();
  }
  rc = g.first_step();
  while (rc == GK_OK && g.more()) {
    rc = g.step_begin();
    if (rc == GK_OK && g.bound_in_flight) {
      rc = wait_stream(ctx);
      if (rc == GK_OK) rc = g.after_bound();
      if (rc == GK_OK && g.sums_in_flight) {
        rc = wait_stream(ctx);
        if (rc == GK_OK) rc = g.after_sums();
      }
    }
    if (rc == GK_OK && !g.step_done) rc = g.exact_step();
    if (rc == GK_OK) g.step_end();
  }
  if (rc) { g.abandon(); return rc; }
  g.finish();
  *out = g.S.release();
  return GK_OK;
}

}  // extern "C"

namespace {

/* All genes in LOCK-STEP: each phase is queued for every gene before ONE wait (about ten stream synchronisations per
 * sample).  The form for several streams, for index tables, and the one that settles the value table (repeated passes
 * while log10 values are being defined). */
int sample_search_lockstep(gk_ctx* ctx, const std::vector<gk_ctx*>& cx, gk_tab* tab, gk_dptr d_vflag, gk_lut* lut,
                           gk_gene_job* jobs, int32_t n_jobs, const std::vector<int>& live, gk_argsort_fn argsort,
                           gk_log10_fn log10_fn, gk_search** out) {
  int rc = GK_OK;
  std::vector<gk_ctx*> ctx_of((size_t)n_jobs, ctx);
  {
    std::vector<int> by_size(live);      // largest tables first, dealt round-robin: the streams carry similar loads
    std::stable_sort(by_size.begin(), by_size.end(), [&](int x, int y) {
      return jobs[x].n_rows * jobs[x].n_allele > jobs[y].n_rows * jobs[y].n_allele;
    });
    for (size_t k = 0; k < by_size.size(); ++k) ctx_of[by_size[k]] = cx[k % cx.size()];
  }
  auto wait_all = [&]() {
    int worst = GK_OK;
    for (gk_ctx* c : cx) { const int r = wait_stream(c); if (r) worst = r; }
    return worst;
  };
  SearchClock clock("lock-step", (int)live.size());
  auto lap = [&](bool waited) { clock.lap(waited); };
  // ---- phase 0: the compatibility tables (log-likelihoods through the value table + mismatch counts); a float64 table's
  // column sums are queued right behind the kernel that writes it -- the table is still in the Infinity Cache then, which
  // it is not any more once the tables of all genes (1.3 GB for a configs[1] sample) have been written
  std::vector<uint32_t> flags((size_t)n_jobs, 0);
  std::vector<std::unique_ptr<GeneSearch>> gs((size_t)n_jobs);
  std::vector<char> sums_queued((size_t)n_jobs, 0);
  auto start_search = [&](int i, gk_dptr d_L, bool indexed) {       // the gene's search object + its column sums, queued
    gk_gene_job& j = jobs[i];
    std::vector<int32_t> cols((size_t)j.n_allele);
    std::iota(cols.begin(), cols.end(), 0);
    if (gs[i]) gs[i]->abandon();                                    // a pass that is being repeated: its sums are void
    gs[i].reset(new GeneSearch());
    const GkTable tbl = indexed ? GkTable{j.d_lidx, j.ldm, lut->d_vals} : GkTable{d_L, j.n_rows, nullptr};
    int r = gs[i]->init(ctx_of[i], tbl, indexed ? 0 : d_L, j.n_rows, j.n_rows, j.n_allele, j.d_miss8, j.ldm, j.d_msum,
                        cols.data(), j.n_allele, j.n_steps, j.top_n, argsort);
    if (r == GK_OK) r = gs[i]->colsum_enqueue();
    sums_queued[i] = r == GK_OK;
    return r;
  };
  auto quit = [&](int code) {
    for (gk_ctx* c : cx) gk_fetch_cancel(c);
    for (auto& g : gs) if (g) g->abandon();
    return code;
  };
  // The tables are written until every product met had its log10 in the value table.  A kernel that stored NaN for a
  // product says so in its gene's flag word (bit 2), so only THAT gene's table is written again; a job without a flag
  // word (no mismatch table) is judged by the table's growth since its launch, as before.
  std::vector<int> dirty(live);
  std::vector<uint32_t> sticky((size_t)n_jobs, 0);      // bit 0 of a gene's flag word survives its patches
  bool settled = false;
  for (int pass = 0; pass < 64 && !settled; ++pass) {
    int32_t known_at_launch = 0;
    rc = gk_lut_known(lut, &known_at_launch);
    if (rc) return quit(rc);
    for (int i : dirty) {
      gk_gene_job& j = jobs[i];
      gk_ctx* const gc = ctx_of[i];
      if (pass > 0 && j.d_L && j.d_miss8 && j.d_flags && (flags[i] & 12u) == 4u) {
        // the entries that hold their product get their log10 (gk_compat_patch); everything else of the table stands
        j.patches++;
        sticky[i] |= flags[i] & 1u;
        rc = gk_compat_patch(gc, lut, j.d_L, j.n_rows, j.n_allele, j.d_miss8, j.ldm, j.d_flags);
        if (rc == GK_OK) rc = gk_miss_colsum(gc, j.d_miss8, j.ldm, j.n_allele, j.d_msum);
        if (rc == GK_OK && gk_fetch_queue(gc, &flags[i], gk_ptr<void>(j.d_flags), sizeof(uint32_t)) != hipSuccess) rc = GK_ERR_HIP;
        if (rc == GK_OK) rc = start_search(i, j.d_L, false);
        if (rc) return quit(rc);
        continue;
      }
      j.passes++;
      sticky[i] = 0;
      if (j.d_lidx && !j.d_L) {
        rc = gk_compat_index(gc, tab, j.d_rows, j.n_rows, d_vflag, j.vbeg, j.vend, j.d_mask, j.words, j.n_allele, 0, lut,
                             j.d_lidx, j.d_miss8, j.ldm, j.d_flags);
        if (rc == GK_OK) rc = gk_miss_colsum(gc, j.d_miss8, j.ldm, j.n_allele, j.d_msum);
        if (rc == GK_OK && gk_fetch_queue(gc, &flags[i], gk_ptr<void>(j.d_flags), sizeof(uint32_t)) != hipSuccess) rc = GK_ERR_HIP;
      } else if (j.d_miss8) {
        rc = gk_compat_log_miss(gc, tab, j.d_rows, j.n_rows, d_vflag, j.vbeg, j.vend, j.d_mask, j.words, j.n_allele, 0, lut,
                                j.d_L, j.d_miss8, j.ldm, j.d_flags);
        if (rc == GK_OK) rc = gk_miss_colsum(gc, j.d_miss8, j.ldm, j.n_allele, j.d_msum);
        if (rc == GK_OK && gk_fetch_queue(gc, &flags[i], gk_ptr<void>(j.d_flags), sizeof(uint32_t)) != hipSuccess) rc = GK_ERR_HIP;
        if (rc == GK_OK) rc = start_search(i, j.d_L, false);
      } else {
        rc = gk_compat_log(gc, tab, j.d_rows, j.n_rows, d_vflag, j.vbeg, j.vend, j.d_mask, j.words, j.n_allele, 0, lut, j.d_L);
        if (rc == GK_OK) rc = start_search(i, j.d_L, false);
      }
      if (rc) return quit(rc);
    }
    lap(false); rc = wait_all(); lap(true);          // every key these kernels claimed is stored
    if (rc) return quit(rc);
    int32_t n_new = 0, n_known = 0, n_undefined = 0;
    rc = gk_lut_resolve(lut, log10_fn, &n_new, &n_known, &n_undefined);
    if (rc) return quit(rc);
    std::vector<int> again;
    bool unflagged = false;
    for (int i : dirty) {
      if (jobs[i].d_flags) { if (flags[i] & 4u) again.push_back(i); }
      else unflagged = true;
    }
    if (unflagged && (n_known > known_at_launch || n_undefined != 0))
      for (int i : dirty) if (!jobs[i].d_flags) again.push_back(i);
    std::sort(again.begin(), again.end());
    settled = again.empty();
    dirty.swap(again);
  }
  if (!settled) {
    gk_set_error("log10 value table did not settle");
    return quit(GK_ERR_ASSERT);
  }
  // a gene whose indices do not fit 16 bits (the value table holds more than 65535 values) takes the float64 form, in a
  // block of this call's own; every value is defined by now, so one pass writes it
  std::vector<void*> own_L((size_t)n_jobs, nullptr);
  auto fail = [&](int code) {
    for (gk_ctx* c : cx) gk_fetch_cancel(c);        // copies still queued point into the searches that go away now
    for (auto& g : gs) if (g) g->abandon();
    for (int i = 0; i < n_jobs; ++i) gk_pool_free(ctx_of[i], own_L[i]);
    return code;
  };
  for (int i : live) {
    gk_gene_job& j = jobs[i];
    j.indexed = (j.d_lidx && !j.d_L && !(flags[i] & 2u)) ? 1 : 0;
    if (j.d_lidx && !j.d_L && !j.indexed) {
      gk_ctx* const gc = ctx_of[i];
      if (gk_pool_malloc(gc, &own_L[i], (size_t)j.n_allele * (size_t)j.n_rows * sizeof(double)) != hipSuccess) {
        gk_set_error("out of device memory for the float64 table of a gene");
        return fail(GK_ERR_HIP);
      }
      rc = gk_compat_log_miss(gc, tab, j.d_rows, j.n_rows, d_vflag, j.vbeg, j.vend, j.d_mask, j.words, j.n_allele, 0, lut,
                              gk_addr(own_L[i]), j.d_miss8, j.ldm, j.d_flags);
      if (rc == GK_OK) rc = gk_miss_colsum(gc, j.d_miss8, j.ldm, j.n_allele, j.d_msum);
      if (rc == GK_OK && gk_fetch_queue(gc, &flags[i], gk_ptr<void>(j.d_flags), sizeof(uint32_t)) != hipSuccess) rc = GK_ERR_HIP;
      if (rc == GK_OK) rc = wait_stream(gc);
      if (rc) return fail(rc);
      j.passes++;
    }
  }
  // ---- phase 1: the column sums that are not queued yet (index tables, float64 tables of this call's own), first step
  bool late = false;
  for (int i : live) {
    gk_gene_job& j = jobs[i];
    j.bound_ok = (j.d_miss8 && ((flags[i] | sticky[i]) & 1u) == 0) ? 1 : 0;
    if (!sums_queued[i]) {
      rc = start_search(i, own_L[i] ? gk_addr(own_L[i]) : j.d_L, j.indexed != 0);
      if (rc) return fail(rc);
      late = true;
    }
    if (!j.bound_ok) gs[i]->bound = false;        // a mismatch count near the underflow range / a very long row: exact steps
  }
  if (late) {
    lap(false); rc = wait_all(); lap(true);
    if (rc) return fail(rc);
  }
  for (int i : live) {
    gs[i]->colsum_collect();
    rc = gs[i]->first_step();
    if (rc) return fail(rc);
  }
  // ---- further steps, all genes that have one together
  for (;;) {
    std::vector<int> todo;
    for (int i : live) if (gs[i]->more()) todo.push_back(i);
    if (todo.empty()) break;
    for (int i : todo) { rc = gs[i]->step_begin(); if (rc) return fail(rc); }
    lap(false); rc = wait_all(); lap(true);
    if (rc) return fail(rc);
    for (int i : todo) { rc = gs[i]->after_bound(); if (rc) return fail(rc); }
    lap(false); rc = wait_all(); lap(true);
    if (rc) return fail(rc);
    for (int i : todo) { rc = gs[i]->after_sums(); if (rc) return fail(rc); }
    for (int i : todo) {
      if (!gs[i]->step_done) { rc = gs[i]->exact_step(); if (rc) return fail(rc); }
      gs[i]->step_end();
    }
  }
  for (int i : live) {
    gs[i]->finish();
    out[i] = gs[i]->S.release();
  }
  for (int i = 0; i < n_jobs; ++i) gk_pool_free(ctx_of[i], own_L[i]);
  lap(false);
  return GK_OK;
}

/* All genes PIPELINED on one stream: every gene is a sequence table -> [bound -> sums]* whose stages are cut where the
 * host needs a result; a stage ends with a mark (gk_fetch_mark), and the host takes the stages in the order they were
 * queued: waits for the mark, does the gene's host work, queues its next stage -- behind the stages of the other genes
 * that are still running.  The stream therefore always holds about one stage of every gene, and the host work of a
 * gene is hidden behind the kernels of the others (lock-step: the GPU idles while the host handles a phase of all
 * genes, ~3 ms of a 9 ms sample).  The value table settles PER GENE: a compatibility kernel that stored NaN for a product
 * without a log10 raises bit 2 of its gene's flag word, which comes back with the table's mark -- before any host sort
 * sees the column sums; that gene's table is written again after the new values are defined, the searches of the other
 * genes are not touched (a sample that brings one new value pays one more kernel of one gene). */
int sample_search_pipelined(gk_ctx* ctx, gk_tab* tab, gk_dptr d_vflag, gk_lut* lut, gk_gene_job* jobs, int32_t n_jobs,
                            const std::vector<int>& live, gk_argsort_fn argsort, gk_log10_fn log10_fn, gk_search** out) {
  SearchClock clock("pipelined", (int)live.size());
  std::vector<uint32_t> flags((size_t)n_jobs, 0);
  std::vector<std::unique_ptr<GeneSearch>> gs((size_t)n_jobs);
  enum { kTable, kBound, kSums, kOneSet };
  struct Item { int gene, stage; uint64_t mark; };
  std::deque<Item> queue;
  // Candidate searches whose every step offers ONE allele (exon groups of one member -- the usual case of a flood of tied
  // exon sets): step k has one candidate, the set of the first k alleles, so the search IS that chain of sets.  The exact
  // sums of the k-allele prefixes of all such searches of a gene are ONE set-sum call per k (the kernels and the tree of a
  // search step: same bits), instead of a bound + a selection + a set sum of one set per search and step.
  struct OneSetBatch { GkSumCall call; std::vector<int> deps; int k = 0; std::vector<int32_t> ids; };
  std::vector<std::unique_ptr<OneSetBatch>> batches;
  auto fail = [&](int code) {
    gk_fetch_cancel(ctx);                        // copies still queued point into the searches that go away now
    for (auto& g : gs) if (g) g->abandon();
    for (auto& b : batches) if (b) gk_release(ctx, b->call.temps);
    return code;
  };
  auto push = [&](int gene, int stage) {
    uint64_t m = 0;
    const int r = set_mark(ctx, &m);
    if (r == GK_OK) queue.push_back({gene, stage, m});
    return r;
  };
  // gene i's table, mismatch sums, flag word and column sums (right behind the kernel that wrote the table: it is still
  // in the Infinity Cache), then a mark
  std::vector<uint32_t> sticky((size_t)n_jobs, 0);      // bit 0 of a gene's flag word survives its patches
  auto write_table = [&](int i, bool patch) -> int {
    gk_gene_job& j = jobs[i];
    int rc;
    if (patch) {      // only the entries that hold their product are touched (gk_compat_patch)
      j.patches++;
      sticky[i] |= flags[i] & 1u;
      rc = gk_compat_patch(ctx, lut, j.d_L, j.n_rows, j.n_allele, j.d_miss8, j.ldm, j.d_flags);
    } else {
      j.passes++;
      sticky[i] = 0;
      rc = gk_compat_log_miss(ctx, tab, j.d_rows, j.n_rows, d_vflag, j.vbeg, j.vend, j.d_mask, j.words, j.n_allele, 0, lut,
                              j.d_L, j.d_miss8, j.ldm, j.d_flags);
    }
    if (rc == GK_OK) rc = gk_miss_colsum(ctx, j.d_miss8, j.ldm, j.n_allele, j.d_msum);
    if (rc == GK_OK && gk_fetch_queue(ctx, &flags[i], gk_ptr<void>(j.d_flags), sizeof(uint32_t)) != hipSuccess) rc = GK_ERR_HIP;
    if (rc) return rc;
    std::vector<int32_t> cols((size_t)j.n_allele);
    std::iota(cols.begin(), cols.end(), 0);
    if (gs[i]) gs[i]->abandon();                 // the pass before this one: its column sums are void
    gs[i].reset(new GeneSearch());
    rc = gs[i]->init(ctx, GkTable{j.d_L, j.n_rows, nullptr}, j.d_L, j.n_rows, j.n_rows, j.n_allele, j.d_miss8, j.ldm, j.d_msum,
                     cols.data(), j.n_allele, j.n_steps, j.top_n, argsort);
    if (rc == GK_OK) rc = gs[i]->colsum_enqueue();
    if (rc == GK_OK) rc = push(i, kTable);
    return rc;
  };
  // searches that read another job's table (exon-first: the candidate searches of a gene on its full table) start when
  // that table is final
  std::vector<std::vector<int>> dependents((size_t)n_jobs);
  for (int i : live)
    if (jobs[i].table_of >= 0) dependents[(size_t)jobs[i].table_of].push_back(i);
  int rc = GK_OK;
  for (int i : live) {
    if (jobs[i].table_of >= 0) continue;
    rc = write_table(i, false);
    if (rc) return fail(rc);
  }

  // gene i goes on until it has a stage in flight (queued here) or no step left
  auto advance = [&](int i, bool at_step_start) -> int {
    GeneSearch& g = *gs[i];
    for (;;) {
      if (at_step_start) {
        if (!g.more()) return GK_OK;
        const int r = g.step_begin();
        if (r) return r;
        if (g.bound_in_flight) return push(i, kBound);
      }
      if (!g.step_done) {          // the bound does not apply / could not settle the step: float64 sums of every candidate
        const int r = g.exact_step();
        if (r) return r;
      }
      g.step_end();
      at_step_start = true;
    }
  };
  while (!queue.empty()) {
    const Item it = queue.front();
    queue.pop_front();
    clock.lap(false);
    rc = wait_mark(ctx, it.mark);
    clock.lap(true);
    if (rc) return fail(rc);
    const int at = it.stage == kOneSet ? batches[(size_t)it.gene]->deps[0] : it.gene;
    gk_gene_job& j = jobs[at];
    GeneSearch& g = *gs[at];
    switch (it.stage) {
      case kTable: {
        if (flags[it.gene] & 4u) {
          // the kernel met a product without a log10 and left the product in its place (a sample that brings new values):
          // define what is stored by now -- this gene's kernel has completed, so its own keys are -- and patch THIS gene's
          // table (gk_compat_patch: one pass over it; the whole kernel again only for a product that could not mark
          // itself), behind the stages of the other genes, whose searches go on.  A key that a kernel still running on another
          // stream has claimed but not stored ends the resolver's batch; after a few fruitless passes the device is drained.
          if (j.passes + j.patches >= 64) {
            gk_set_error("log10 value table did not settle");
            rc = GK_ERR_ASSERT;
            break;
          }
          rc = (j.passes + j.patches < 4 ? gk_lut_resolve_stored : gk_lut_resolve)(lut, log10_fn, nullptr, nullptr, nullptr);
          if (rc == GK_OK) rc = write_table(it.gene, (flags[it.gene] & 8u) == 0);
          break;
        }
        j.bound_ok = ((flags[it.gene] | sticky[it.gene]) & 1u) == 0 ? 1 : 0;
        if (!j.bound_ok) g.bound = false;      // a mismatch count near the underflow range / a very long row: exact steps
        g.colsum_collect();
        if (j.n_steps > 0) {
          rc = g.first_step();
          if (rc == GK_OK) rc = advance(it.gene, true);
        }
        std::vector<int> one_set;                        // the dependents whose steps offer one allele each
        for (int d : dependents[(size_t)it.gene]) {      // the table is final: the searches that read it begin
          if (rc) break;
          gk_gene_job& jd = jobs[d];
          jd.bound_ok = j.bound_ok;
          bool single = jd.n_step_cols >= jd.n_steps && jd.n_steps >= 1;
          for (int q = 0; q < jd.n_steps && single; ++q) single = jd.step_cols_off[q + 1] - jd.step_cols_off[q] == 1;
          if (single) {
            gs[d].reset(new GeneSearch());
            gs[d]->S.reset(new gk_search());
            gs[d]->S->n_allele = j.n_allele;
            gs[d]->S->colsum = g.S->colsum;
            const int32_t a0 = jd.step_cols[jd.step_cols_off[0]];
            if (a0 < 0 || a0 >= j.n_allele) { gk_set_error("candidate allele out of range"); rc = GK_ERR_ARG; break; }
            Step s1;
            s1.n = 1;
            s1.value.push_back(g.S->colsum[(size_t)a0]);
            s1.sum_indv.push_back(g.S->colsum[(size_t)a0]);
            s1.ids.push_back(a0);
            s1.frac.push_back(1.0);
            gs[d]->S->steps.push_back(std::move(s1));
            if (jd.n_steps >= 2) one_set.push_back(d);
            continue;
          }
          gs[d].reset(new GeneSearch());
          std::vector<int32_t> every((size_t)j.n_allele);
          std::iota(every.begin(), every.end(), 0);
          rc = gs[d]->init(ctx, GkTable{j.d_L, j.n_rows, nullptr}, j.d_L, j.n_rows, j.n_rows, j.n_allele, j.d_miss8, j.ldm, j.d_msum,
                           every.data(), j.n_allele, jd.n_steps, jd.top_n, argsort);
          if (rc) break;
          if (!j.bound_ok) gs[d]->bound = false;
          gs[d]->S->colsum = g.S->colsum;
          if (jd.n_step_cols > 0) {
            gs[d]->step_cols.resize((size_t)jd.n_step_cols);
            for (int q = 0; q < jd.n_step_cols; ++q)
              gs[d]->step_cols[(size_t)q].assign(jd.step_cols + jd.step_cols_off[q], jd.step_cols + jd.step_cols_off[q + 1]);
          }
          if (jd.n_steps > 0) {
            rc = gs[d]->first_step();
            if (rc == GK_OK) rc = advance(d, true);
          }
        }
        for (int k = 2; k <= 8 && rc == GK_OK; ++k) {     // the k-allele prefixes of every one-set search, one call per k
          std::unique_ptr<OneSetBatch> b(new OneSetBatch());
          b->k = k;
          for (int d : one_set) {
            if (jobs[d].n_steps < k) continue;
            for (int q = 0; q < k; ++q) {
              const int32_t a = jobs[d].step_cols[jobs[d].step_cols_off[q]];
              if (a < 0 || a >= j.n_allele) { gk_set_error("candidate allele out of range"); rc = GK_ERR_ARG; }
              b->ids.push_back(a);
            }
            b->deps.push_back(d);
          }
          if (rc || b->deps.empty()) break;
          rc = gk_shares_enqueue(ctx, GkTable{j.d_L, j.n_rows, nullptr}, j.n_rows, b->ids.data(), (int32_t)b->deps.size(), k, true,
                                 b->call);
          if (rc) break;
          gs[b->deps[0]]->S->note(b->call.leafwise ? 3 : 2, j.n_rows, (int64_t)b->deps.size(), k, gs[b->deps[0]]->S->distinct(b->ids.data(), b->ids.size()), 0, 0);
          batches.push_back(std::move(b));
          rc = push((int)batches.size() - 1, kOneSet);
        }
        break;
      }
      case kOneSet: {
        OneSetBatch& b = *batches[(size_t)it.gene];
        const int k = b.k;
        std::vector<double> value(b.deps.size()), frac(b.deps.size() * (size_t)k);
        gk_shares_collect(ctx, b.call, value.data(), frac.data());
        for (size_t x = 0; x < b.deps.size(); ++x) {
          gk_search& S = *gs[b.deps[x]]->S;
          Step st;
          st.n = k;
          st.bounded = 1;
          st.value.push_back(value[x]);
          for (int q = 0; q < k; ++q) {
            const int32_t a = b.ids[x * (size_t)k + q];
            st.ids.push_back(a);
            st.sum_indv.push_back(S.colsum[(size_t)a]);
            st.frac.push_back(frac[x * (size_t)k + q]);
          }
          // the steps of a search arrive in order: the batches of a gene were queued k = 2, 3, ... on one stream
          S.steps.push_back(std::move(st));
        }
        break;
      }
      case kBound:
        rc = g.after_bound();
        if (rc == GK_OK) rc = g.sums_in_flight ? push(it.gene, kSums) : advance(it.gene, false);
        break;
      default:
        rc = g.after_sums();
        if (rc == GK_OK) rc = advance(it.gene, false);
        break;
    }
    if (rc) return fail(rc);
  }
  for (int i : live) {
    gs[i]->finish();
    out[i] = gs[i]->S.release();
  }
  clock.lap(false);
  return GK_OK;
}

}  // namespace

extern "C" {

/* The searches of ALL genes of a sample on the calling thread: the compatibility tables, then the column sums, then
 * step 2, 3, ... of every gene that has one -- pipelined on one stream (sample_search_pipelined)
 * or in lock-step (several streams, index tables, a value table that is still growing), so that one host
 * thread keeps the GPU fed (typing_mulit_allele.py:340-381 + 478-598 per gene; kir_typing.py:103-132 is the gene loop).
 * jobs[i] describes gene i (tables allocated by the caller); out[i] receives its search (gk_search_*), NULL for a
 * gene without rows.  `log10_fn` = numpy.log10 (value table, see gk_lut_resolve), `argsort` = numpy.argsort. */
int gk_sample_search(gk_ctx* ctx, gk_ctx** more_ctx, int32_t n_more, gk_tab* tab, gk_dptr d_vflag, gk_lut* lut,
                     gk_gene_job* jobs, int32_t n_jobs, gk_argsort_fn argsort, gk_log10_fn log10_fn, gk_search** out) {
  gk_bind(ctx);
  GK_REQUIRE(ctx && tab && lut && jobs && argsort && log10_fn && out && n_jobs >= 0, "null pointer");
  GK_REQUIRE(n_more >= 0 && (n_more == 0 || more_ctx), "bad context list");
  // the calling thread's contexts: gene i queues everything it does on ONE of them (its stream, its staging, its pool),
  // so the kernels of different genes overlap on the GPU while the host still makes one pass and one wait per phase
  std::vector<gk_ctx*> cx{ctx};
  for (int k = 0; k < n_more; ++k) {
    GK_REQUIRE(more_ctx[k] && more_ctx[k]->device == ctx->device, "contexts of one call share a device");
    cx.push_back(more_ctx[k]);
  }
  for (int i = 0; i < n_jobs; ++i) out[i] = nullptr;
  std::vector<int> live;
  bool any_special = false;      // table-only jobs, searches on another job's table, columns per step: pipelined form only
  for (int i = 0; i < n_jobs; ++i) {
    gk_gene_job& j = jobs[i];
    j.bound_ok = 0;
    j.passes = 0;
    j.indexed = 0;
    j.patches = 0;
    GK_REQUIRE(j.n_rows >= 0 && j.n_allele >= 0 && j.n_steps >= 0 && j.top_n >= 1, "bad gene job");
    GK_REQUIRE(j.table_of >= -1 && j.table_of < n_jobs && j.table_of != i, "bad table reference");
    GK_REQUIRE(j.n_step_cols >= 0 && (j.n_step_cols == 0 || (j.step_cols && j.step_cols_off)), "bad step columns");
    if (j.table_of >= 0) {       // a search on another job's table: that job writes it
      const gk_gene_job& o = jobs[j.table_of];
      GK_REQUIRE(o.table_of < 0, "a table reference must name a job that writes its own table");
      GK_REQUIRE(j.n_steps >= 1, "a search on another job's table needs steps");
      if (o.n_rows > 0 && o.n_allele > 0) { live.push_back(i); any_special = true; }
      continue;
    }
    if (j.n_steps == 0 || j.n_step_cols > 0) any_special = true;
    if (j.n_rows > 0 && j.n_allele > 0) {
      GK_REQUIRE(j.d_rows && (j.d_L || j.d_lidx) && j.d_mask && j.words >= 1, "gene job without tables");
      GK_REQUIRE(!j.d_miss8 || (j.d_msum && j.d_flags && j.ldm >= j.n_rows && j.ldm % 64 == 0), "bad mismatch table");
      GK_REQUIRE(!j.d_lidx || j.d_miss8, "the index form comes with the mismatch table (same stride)");
      live.push_back(i);
    }
  }
  if (live.empty()) return GK_OK;
  bool float_tables = true;
  for (int i : live) if (jobs[i].table_of < 0) float_tables = float_tables && jobs[i].d_L && !jobs[i].d_lidx;
  bool flagged = true;               // every table comes with the flag word that reports products without a log10
  for (int i : live) if (jobs[i].table_of < 0) flagged = flagged && jobs[i].d_miss8 && jobs[i].d_flags;
  if (n_more == 0 && float_tables && flagged)
    return sample_search_pipelined(ctx, tab, d_vflag, lut, jobs, n_jobs, live, argsort, log10_fn, out);
  GK_REQUIRE(!any_special, "table-only jobs, searches on another job's table and per-step columns need float64 tables "
                           "with mismatch tables on one stream (the pipelined form)");
  return sample_search_lockstep(ctx, cx, tab, d_vflag, lut, jobs, n_jobs, live, argsort, log10_fn, out);
}

int gk_search_steps(gk_search* s, int32_t* n_steps) {
  GK_REQUIRE(s && n_steps, "null pointer");
  *n_steps = (int32_t)s->steps.size();
  return GK_OK;
}

int gk_search_info(gk_search* s, int32_t step, int32_t* n, int64_t* rows, int32_t* bounded) {
  GK_REQUIRE(s && step >= 0 && step < (int32_t)s->steps.size() && n && rows && bounded, "bad search step");
  *n = s->steps[step].n;
  *rows = s->steps[step].rows();
  *bounded = s->steps[step].bounded;
  return GK_OK;
}

int gk_search_copy(gk_search* s, int32_t step, double* value, double* sum_indv, int32_t* ids, double* frac) {
  GK_REQUIRE(s && step >= 0 && step < (int32_t)s->steps.size(), "bad search step");
  const Step& st = s->steps[step];
  if (value) std::copy(st.value.begin(), st.value.end(), value);
  if (sum_indv) std::copy(st.sum_indv.begin(), st.sum_indv.end(), sum_indv);
  if (ids) std::copy(st.ids.begin(), st.ids.end(), ids);
  if (frac) std::copy(st.frac.begin(), st.frac.end(), frac);
  return GK_OK;
}

/* Every step of MANY searches in one call (exon-first adopts hundreds of candidate searches per sample: a call per search
 * and step was a third of the typing's host time).  totals[3] = steps, rows, cells (rows x set size) over all listed
 * searches; with the arrays given: meta = [steps of search 0 .. n - 1 | per step (set size, rows, bounded)] and the rows of
 * the steps back to back in (search, step) order. */
int gk_search_export(gk_search* const* s, int32_t n, int64_t* totals, int64_t* meta, double* value, double* sum_indv,
                     double* frac, int32_t* ids) {
  GK_REQUIRE(s && n >= 0 && totals, "null pointer");
  int64_t steps = 0, rows = 0, cells = 0;
  for (int q = 0; q < n; ++q) {
    GK_REQUIRE(s[q], "null search");
    for (const Step& st : s[q]->steps) { ++steps; rows += st.rows(); cells += st.rows() * st.n; }
  }
  totals[0] = steps; totals[1] = rows; totals[2] = cells;
  if (!meta) return GK_OK;
  GK_REQUIRE(value && sum_indv && frac && ids, "null pointer");
  int64_t* triple = meta + n;
  for (int q = 0; q < n; ++q) {
    meta[q] = (int64_t)s[q]->steps.size();
    for (const Step& st : s[q]->steps) {
      triple[0] = st.n; triple[1] = st.rows(); triple[2] = st.bounded;
      triple += 3;
      value = std::copy(st.value.begin(), st.value.end(), value);
      sum_indv = std::copy(st.sum_indv.begin(), st.sum_indv.end(), sum_indv);
      frac = std::copy(st.frac.begin(), st.frac.end(), frac);
      ids = std::copy(st.ids.begin(), st.ids.end(), ids);
    }
  }
  return GK_OK;
}

int gk_search_colsum(gk_search* s, double* out) {
  GK_REQUIRE(s && out, "null pointer");
  std::copy(s->colsum.begin(), s->colsum.end(), out);
  return GK_OK;
}

int gk_search_log(gk_search* s, int64_t* out, int64_t capacity, int64_t* n_out) {
  GK_REQUIRE(s && n_out, "null pointer");
  *n_out = (int64_t)s->log.size();
  if (out) std::copy(s->log.begin(), s->log.begin() + std::min<int64_t>(capacity, *n_out), out);
  return GK_OK;
}

int gk_search_destroy(gk_search* s) {
  delete s;
  return GK_OK;
}

// isHomozygous, lines 835-857, on flat observations: entry i = (position, label code, negative?, count).
// Per position the reference sums counts per label, skips positions with one label or only negative labels,
// keeps counts > 3, needs their total >= 20, and calls the position heterozygous when the second largest kept
// share is > 0.1 and > 1 / (2 cn).  *homozygous = 1 when no position is heterozygous.
namespace {
struct SiteObs { int64_t pos, label; int64_t cnt; };

// the verdict over flat (position, label, count) observations; label = code << 1 | negative
int sites_homozygous(std::vector<SiteObs>& obs, int32_t cn) {
  std::sort(obs.begin(), obs.end(), [](const SiteObs& a, const SiteObs& b) {
    return a.pos != b.pos ? a.pos < b.pos : a.label < b.label;
  });
  std::vector<int64_t> counts;
  for (size_t i = 0; i < obs.size();) {
    size_t j = i;
    counts.clear();
    int n_label = 0, n_positive = 0;
    while (j < obs.size() && obs[j].pos == obs[i].pos) {
      size_t q = j;
      int64_t c = 0;
      while (q < obs.size() && obs[q].pos == obs[i].pos && obs[q].label == obs[j].label) c += obs[q++].cnt;
      ++n_label;
      n_positive += (obs[j].label & 1) == 0;
      counts.push_back(c);
      j = q;
    }
    i = j;
    if (n_label <= 1 || n_positive == 0) continue;
    std::sort(counts.begin(), counts.end(), [](int64_t a, int64_t b) { return a > b; });
    int64_t total = 0;
    int kept = 0;
    for (int64_t c : counts)
      if (c > 3) { total += c; ++kept; }
    if (total < 20 || kept < 2) continue;
    // counts are descending: the kept ones are a prefix; the second largest share decides
    const double second = (double)counts[1] / (double)total;
    if (!(second > 0.1)) continue;          // `major` then has one entry (shares are descending too)
    if (second > (1.0 / (double)(cn * 2))) return 0;
  }
  return 1;
}
}  // namespace

int gk_site_verdict(const int64_t* pos, const int64_t* code, const uint8_t* negative, const int64_t* count, int64_t n,
                    int32_t cn, int32_t* homozygous) {
  GK_REQUIRE(homozygous && cn >= 1 && (n == 0 || (pos && code && negative && count)), "bad verdict arguments");
  *homozygous = 1;
  if (n == 0) return GK_OK;
  std::vector<SiteObs> obs((size_t)n);
  for (int64_t i = 0; i < n; ++i) obs[i] = SiteObs{pos[i], (code[i] << 1) | (negative[i] ? 1 : 0), count[i]};
  *homozygous = sites_homozygous(obs, cn);
  return GK_OK;
}

// The same verdict straight from the surviving tallies: ordinals into `keys` (index keys followed by the sample's
// novel keys), their positive / negative counts.  Deletions are skipped (typing_mulit_allele.py:829-831); the label
// of a variant is str(val) in the reference -- the substituted base, or the inserted string (label_of_insert[id]:
// a one-base insertion prints like a substitution of that base, longer ones get codes of their own).
int gk_site_verdict_tallies(const uint64_t* keys, int64_t n_keys, const int64_t* label_of_insert, int64_t n_insert,
                            const int32_t* ordinal, const uint32_t* positive, const uint32_t* negative, int64_t n,
                            int32_t cn, int32_t* homozygous) {
  GK_REQUIRE(homozygous && cn >= 1 && (n == 0 || (keys && ordinal && positive && negative)), "bad verdict arguments");
  *homozygous = 1;
  std::vector<SiteObs> obs;
  obs.reserve((size_t)n * 2);
  for (int64_t i = 0; i < n; ++i) {
    GK_REQUIRE(ordinal[i] >= 0 && ordinal[i] < n_keys, "tally ordinal outside the key table");
    const uint64_t k = keys[ordinal[i]];
    const uint32_t typ = (uint32_t)(k >> GK_KEY_TYP_SHIFT) & 3u;
    if (typ == GK_TYP_DEL) continue;
    const int64_t val = (int64_t)(k & GK_KEY_VAL_MASK);
    int64_t code = val;
    if (typ == GK_TYP_INS) {
      GK_REQUIRE(label_of_insert && n_insert > 0, "insertion without a label table");
      code = label_of_insert[std::min<int64_t>(val, n_insert - 1)];
    }
    const int64_t pos = (int64_t)((k >> GK_KEY_POS_SHIFT) & 0xFFFFFFu);
    if (positive[i]) obs.push_back(SiteObs{pos, code << 1, (int64_t)positive[i]});
    if (negative[i]) obs.push_back(SiteObs{pos, (code << 1) | 1, (int64_t)negative[i]});
  }
  if (!obs.empty()) *homozygous = sites_homozygous(obs, cn);
  return GK_OK;
}

int gk_site_verdict_genes(const uint64_t* keys, int64_t n_keys, const int64_t* label_of_insert, int64_t n_insert,
                          const int32_t* ordinal, const uint32_t* positive, const uint32_t* negative,
                          const int64_t* bounds, int32_t n_groups, const int32_t* cn, int32_t* homozygous) {
  GK_REQUIRE(bounds && cn && homozygous && n_groups >= 0, "bad verdict arguments");
  for (int32_t g = 0; g < n_groups; ++g) {
    const int64_t a0 = bounds[g], n = bounds[g + 1] - a0;
    GK_REQUIRE(a0 >= 0 && n >= 0, "tally groups must follow one another");
    homozygous[g] = 0;
    if (cn[g] <= 1) continue;
    const int rc = gk_site_verdict_tallies(keys, n_keys, label_of_insert, n_insert, n ? ordinal + a0 : nullptr,
                                           n ? positive + a0 : nullptr, n ? negative + a0 : nullptr, n, cn[g], &homozygous[g]);
    if (rc) return rc;
  }
  return GK_OK;
}

}  // extern "C"

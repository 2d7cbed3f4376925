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

#include "gk_common.h"

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
  // kind (0 maxsum, 1 minsum, 2 set sums), then the model's arguments
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
int gk_setsum(gk_ctx* ctx, gk_dptr d_L, int64_t n_rows, int64_t ld, const int32_t* ids, int32_t n_sets, int32_t c,
              double* value_out, double* frac_out);
int gk_bound_step(gk_ctx* ctx, gk_dptr d_miss8, int64_t ldm, int64_t n_rows, gk_dptr d_msum, const int32_t* ids,
                  int32_t n_sets, int32_t c_prev, const int32_t* cols, int32_t n_cols, const uint8_t* first,
                  int32_t top_n, int32_t cap, uint32_t* hdr_out, int32_t* idx_out, uint32_t* m_out);

int gk_search_run(gk_ctx* ctx, gk_dptr d_L, int64_t n_rows, int64_t ld, int32_t n_allele, gk_dptr d_miss8, int64_t ldm,
                  gk_dptr d_msum, const int32_t* cols, int32_t n_cols, int32_t n_steps, int32_t top_n,
                  gk_argsort_fn argsort, const double* colsum_in, gk_search** out) {
  gk_bind(ctx);
  GK_REQUIRE(ctx && d_L && cols && argsort && out, "null pointer");
  GK_REQUIRE(n_rows > 0 && ld >= n_rows && n_allele > 0 && n_cols > 0 && n_steps >= 1 && n_steps <= 8 && top_n >= 1,
             "bad search arguments");
  for (int a = 0; a < n_cols; ++a) GK_REQUIRE(cols[a] >= 0 && cols[a] < n_allele, "candidate allele out of range");
  const bool bound = d_miss8 != 0 && d_msum != 0;
  std::unique_ptr<gk_search> S(new gk_search());
  S->n_allele = n_allele;
  const int T = top_n, A = n_cols;

  // ---- per-allele column sums = log_probs.sum(axis=0) (line 514), needed by every step (571)
  S->colsum.resize((size_t)n_allele);
  if (colsum_in) {
    std::copy(colsum_in, colsum_in + n_allele, S->colsum.begin());
  } else {
    std::vector<int32_t> every((size_t)n_allele);
    std::iota(every.begin(), every.end(), 0);
    int rc = gk_maxsum(ctx, d_L, n_rows, ld, nullptr, 1, 0, every.data(), n_allele, S->colsum.data());
    if (rc) return rc;
    S->note(0, n_rows, 1, 0, n_allele, 0, 0);
  }
  const double* colsum = S->colsum.data();

  // ---- first allele (512-532): argsort(score)[::-1][:top_n]
  {
    std::vector<double> score((size_t)A);
    for (int a = 0; a < A; ++a) score[a] = colsum[cols[a]];
    std::vector<int64_t> order((size_t)A);
    GK_REQUIRE(argsort(score.data(), A, order.data()) == 0, "host argsort failed");
    Step st;
    st.n = 1;
    const int keep = std::min(T, A);
    for (int i = 0; i < keep; ++i) {
      const int64_t a = order[(size_t)A - 1 - i];
      GK_REQUIRE(a >= 0 && a < A, "host argsort returned an index out of range");
      st.value.push_back(score[a]);
      st.sum_indv.push_back(score[a]);
      st.ids.push_back(cols[a]);
      st.frac.push_back(1.0);
    }
    S->steps.push_back(std::move(st));
  }

  std::vector<uint8_t> first;
  for (int step = 2; step <= n_steps; ++step) {
    const Step& prev = S->steps.back();
    const int k = prev.n, c = k + 1;
    const int Tp = (int)prev.rows();
    first_of_sets(prev.ids.data(), Tp, k, cols, A, n_allele, first);
    int64_t N = 0;
    for (uint8_t f : first) N += f;
    Step st;
    st.n = c;
    bool done = false;
    bool unique_cols = true;
    {
      std::vector<char> seen((size_t)n_allele, 0);
      for (int a = 0; a < A && unique_cols; ++a) { if (seen[cols[a]]) unique_cols = false; seen[cols[a]] = 1; }
    }

    // ---------------- integer bound first (see gk_bound.hip): exact sums for the sets that can reach the cut
    if (bound && N > 0 && unique_cols) {
      const int cap = 4 * T + 4096;
      uint32_t hdr[4];
      std::vector<int32_t> idx((size_t)cap);
      std::vector<uint32_t> mm((size_t)cap);
      int rc = gk_bound_step(ctx, d_miss8, ldm, n_rows, d_msum, prev.ids.data(), Tp, k, cols, A, first.data(), T, cap,
                             hdr, idx.data(), mm.data());
      if (rc) return rc;
      S->note(1, n_rows, Tp, A, k == 1 ? S->distinct(prev.ids.data(), prev.ids.size()) : Tp, 0, 0);
      const int64_t n_sel = hdr[2];
      if (n_sel > 0 && n_sel <= cap) {
        idx.resize((size_t)n_sel);
        std::sort(idx.begin(), idx.end());                         // list order of the candidates
        Head h;
        h.c = c;
        h.ids.resize((size_t)n_sel * c);
        for (int64_t i = 0; i < n_sel; ++i) {
          const int t = idx[i] / A, a = idx[i] % A;
          std::copy(prev.ids.begin() + (size_t)t * k, prev.ids.begin() + (size_t)(t + 1) * k, h.ids.begin() + (size_t)i * c);
          h.ids[(size_t)i * c + k] = cols[a];
        }
        std::vector<double> value((size_t)n_sel), frac((size_t)n_sel * c);
        rc = gk_setsum(ctx, d_L, n_rows, ld, h.ids.data(), (int32_t)n_sel, c, value.data(), frac.data());
        if (rc) return rc;
        S->note(2, n_rows, n_sel, c, S->distinct(h.ids.data(), h.ids.size()), 0, 0);
        // the reference's head: rows of the sorted table that reach the top_n-th value (567, then the cut to top_n)
        bool ok = true;
        std::vector<int64_t> head;
        const int64_t n_top = std::min<int64_t>(std::max<int64_t>(T, N / 5), N);
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
        if (ok) {
          Head hh;
          hh.c = c;
          for (int64_t i : head) {
            hh.value.push_back(value[i]);
            hh.ids.insert(hh.ids.end(), h.ids.begin() + (size_t)i * c, h.ids.begin() + (size_t)(i + 1) * c);
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
          if (ok) {
            const size_t keep = std::min<size_t>(sub.size(), (size_t)T);
            for (size_t q = 0; q < keep; ++q) {
              const int64_t i = contend[sub[q]];
              st.value.push_back(hh.value[i]);
              st.ids.insert(st.ids.end(), hh.ids.begin() + (size_t)i * c, hh.ids.begin() + (size_t)(i + 1) * c);
              st.sum_indv.insert(st.sum_indv.end(), hh.sum_indv.begin() + (size_t)i * c,
                                 hh.sum_indv.begin() + (size_t)(i + 1) * c);
              const double* f = frac.data() + (size_t)head[i] * c;
              st.frac.insert(st.frac.end(), f, f + c);
            }
            st.bounded = 1;
            done = true;
          }
        }
      }
    }

    // ---------------- float64 sums for every candidate (540-598 as written)
    if (!done) {
      std::vector<double> table((size_t)Tp * A);
      int rc = gk_maxsum(ctx, d_L, n_rows, ld, prev.ids.data(), Tp, k, cols, A, table.data());
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
        if (first[i]) { where.push_back(i); score.push_back(table[i]); }
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
      for (size_t q = 0; q < keep; ++q) {
        const int64_t i = contend[sub[q]];
        st.value.push_back(hh.value[i]);
        st.ids.insert(st.ids.end(), hh.ids.begin() + (size_t)i * c, hh.ids.begin() + (size_t)(i + 1) * c);
        st.sum_indv.insert(st.sum_indv.end(), hh.sum_indv.begin() + (size_t)i * c,
                           hh.sum_indv.begin() + (size_t)(i + 1) * c);
        st.frac.insert(st.frac.end(), frac.begin() + sub[q] * c, frac.begin() + (sub[q] + 1) * c);
      }
    }
    S->steps.push_back(std::move(st));
  }
  *out = S.release();
  return GK_OK;
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

}  // extern "C"

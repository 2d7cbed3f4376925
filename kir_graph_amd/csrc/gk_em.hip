// EM strategy ("report" / "em"): typing_em.py:68-188.
//
//   gk_em_sets  getCandidateAllelePerRead 68-87 + getMostFreqAllele 90-104 as bit-set algebra:
//               mate set  = AND of the positive variants' allele rows, minus OR of the negative rows
//                           (empty when the mate has no positive variant);
//               pair set  = L & R when that is non-empty (alleles named twice), else L | R.
//   gk_em_run   hisatEMnp 107-188 on DISTINCT sets with multiplicities.  The reference builds a
//               dense 0/1 read x allele float matrix and sweeps it 3-4 times per iteration; reads
//               with equal candidate sets contribute identical rows, so the device keeps one row
//               per distinct set (weights = multiplicity) and runs the whole SQUAREM loop inside a
//               single workgroup (no launch per iteration).  Sums are evaluated in a fixed order, so
//               results are run-to-run deterministic; they agree with numpy's row-sequential sums to
//               rounding (tolerance 1e-5 relative per BASELINE.json north_star).
#include <algorithm>
#include <vector>

#include "gk_common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kMaxWords = 16;    // up to 512 alleles per gene
constexpr int kMaxAllele = kMaxWords * 32;

__global__ __launch_bounds__(kThreads) void em_sets_kernel(const int32_t* rows, int64_t n_rows, const uint32_t* off,
                                                           const uint32_t* ids, int vbeg, int vend,
                                                           const uint32_t* mask, int words, uint32_t* out) {
  const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (i >= n_rows) return;
  const int64_t row = rows[i];
  uint32_t side[2][kMaxWords];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    // list order in the CSR: lpv, rpv, lnv, rnv
    const uint32_t pb = off[4 * row + s], pe = off[4 * row + s + 1];
    const uint32_t nb = off[4 * row + 2 + s], ne = off[4 * row + 2 + s + 1];
    const bool any_pos = pe > pb;
#pragma unroll
    for (int w = 0; w < kMaxWords; ++w) side[s][w] = (any_pos && w < words) ? 0xFFFFFFFFu : 0u;
    for (uint32_t k = pb; k < pe; ++k) {
      const int v = (int)ids[k];
      const bool indexed = v >= vbeg && v < vend;
#pragma unroll
      for (int w = 0; w < kMaxWords; ++w)
        if (w < words) side[s][w] &= indexed ? mask[(int64_t)(v - vbeg) * words + w] : 0u;
    }
    if (any_pos) {
      for (uint32_t k = nb; k < ne; ++k) {
        const int v = (int)ids[k];
        if (v < vbeg || v >= vend) continue;
#pragma unroll
        for (int w = 0; w < kMaxWords; ++w)
          if (w < words) side[s][w] &= ~mask[(int64_t)(v - vbeg) * words + w];
      }
    }
  }
  uint32_t both = 0;
#pragma unroll
  for (int w = 0; w < kMaxWords; ++w) both |= side[0][w] & side[1][w];
#pragma unroll
  for (int w = 0; w < kMaxWords; ++w)
    if (w < words) out[i * words + w] = both ? (side[0][w] & side[1][w]) : (side[0][w] | side[1][w]);
}

// Distinct candidate sets and their multiplicities (the EM only needs those): open addressing on a
// 64-bit mix of the row; a slot is owned by the first row that claims it, later rows compare their
// words with the owner's and either add 1 to its count or probe on.
__device__ inline uint64_t mix_row(const uint32_t* r, int words) {
  uint64_t h = 0x9E3779B97F4A7C15ull;
  for (int w = 0; w < words; ++w) {
    h ^= r[w];
    h *= 0xff51afd7ed558ccdull;
    h ^= h >> 29;
  }
  return h;
}

__global__ __launch_bounds__(kThreads) void em_distinct_kernel(const uint32_t* __restrict__ sets, int64_t n_rows, int words,
                                                               int32_t* owner, uint32_t* count, uint32_t mask) {
  const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (i >= n_rows) return;
  const uint32_t* mine = sets + i * words;
  uint32_t s = (uint32_t)mix_row(mine, words) & mask;
  for (uint32_t probe = 0; probe <= mask; ++probe) {
    int32_t o = owner[s];
    if (o < 0) {
      const int32_t prev = atomicCAS(&owner[s], -1, (int32_t)i);
      o = prev < 0 ? (int32_t)i : prev;
    }
    const uint32_t* other = sets + (int64_t)o * words;
    bool same = true;
    for (int w = 0; w < words; ++w) same &= other[w] == mine[w];
    if (same) {
      atomicAdd(&count[s], 1u);
      return;
    }
    s = (s + 1) & mask;
  }
}

__global__ __launch_bounds__(kThreads) void em_distinct_emit(const uint32_t* __restrict__ sets, int words,
                                                             const int32_t* __restrict__ owner,
                                                             const uint32_t* __restrict__ count, uint32_t n_slots,
                                                             uint32_t max_out, uint32_t* n_out, uint32_t* sets_out,
                                                             uint32_t* count_out) {
  const uint32_t s = blockIdx.x * kThreads + threadIdx.x;
  if (s >= n_slots || owner[s] < 0) return;
  const uint32_t k = atomicAdd(n_out, 1u);
  if (k >= max_out) return;
  for (int w = 0; w < words; ++w) sets_out[(int64_t)k * words + w] = sets[(int64_t)owner[s] * words + w];
  count_out[k] = count[s];
}

struct EmShared {
  double p[kMaxAllele], p1[kMaxAllele], p2[kMaxAllele], p3[kMaxAllele];
  double scalar[4];
  int flag;
};

// next(p): q[a] = sum_u w_u * p[a] / (sum_{b in u} p[b]) over sets containing a, then normalise
__device__ void em_step(const uint32_t* sets, const double* weight, double* scale, int n_sets, int words, int n_allele,
                        const double* in, double* out, double* scalar) {
  const int tid = threadIdx.x;
  for (int u = tid; u < n_sets; u += kThreads) {
    double tot = 0.0;
    for (int w = 0; w < words; ++w) {
      uint32_t bits = sets[u * words + w];
      while (bits) {
        const int b = __ffs(bits) - 1;
        bits &= bits - 1;
        tot += in[w * 32 + b];
      }
    }
    scale[u] = tot != 0.0 ? weight[u] / tot : 0.0;
  }
  __syncthreads();
  for (int a = tid; a < n_allele; a += kThreads) {
    const int w = a >> 5;
    const uint32_t bit = 1u << (a & 31);
    double s = 0.0;
    for (int u = 0; u < n_sets; ++u)
      if (sets[u * words + w] & bit) s += scale[u];
    out[a] = in[a] * s;
  }
  __syncthreads();
  if (tid == 0) {
    double tot = 0.0;
    for (int a = 0; a < n_allele; ++a) tot += out[a];
    scalar[0] = tot;
  }
  __syncthreads();
  const double tot = scalar[0];
  for (int a = tid; a < n_allele; a += kThreads) out[a] = out[a] / tot;
  __syncthreads();
}

__device__ void em_solve(EmShared& sh, const uint32_t* sets, const double* weight, double* scale, int n_sets, int words,
                         int n_allele, int iter_max, double diff_threshold, double* prob_out, int* iters_out) {
  const int tid = threadIdx.x;
  for (int a = tid; a < n_allele; a += kThreads) sh.p3[a] = 1.0;
  __syncthreads();
  em_step(sets, weight, scale, n_sets, words, n_allele, sh.p3, sh.p, sh.scalar);
  int iters = 0;
  for (iters = 0; iters < iter_max; ++iters) {
    em_step(sets, weight, scale, n_sets, words, n_allele, sh.p, sh.p1, sh.scalar);
    em_step(sets, weight, scale, n_sets, words, n_allele, sh.p1, sh.p2, sh.scalar);
    if (tid == 0) {
      double rs = 0.0, vs = 0.0;
      for (int a = 0; a < n_allele; ++a) {
        const double r = sh.p1[a] - sh.p[a];
        const double v = sh.p2[a] - sh.p1[a] - r;
        rs += r * r;
        vs += v * v;
      }
      sh.scalar[1] = rs;
      sh.scalar[2] = vs;
    }
    __syncthreads();
    const double rs = sh.scalar[1], vs = sh.scalar[2];
    if (vs > 0.0) {
      const double g = -sqrt(rs / vs);
      for (int a = tid; a < n_allele; a += kThreads) {
        const double r = sh.p1[a] - sh.p[a];
        const double v = sh.p2[a] - sh.p1[a] - r;
        const double x = sh.p[a] - r * g * 2 + v * (g * g);
        sh.p3[a] = x > 0.0 ? x : 0.0;
      }
      __syncthreads();
      em_step(sets, weight, scale, n_sets, words, n_allele, sh.p3, sh.p1, sh.scalar);
    }
    if (tid == 0) {
      double d = 0.0;
      for (int a = 0; a < n_allele; ++a) d += fabs(sh.p[a] - sh.p1[a]);
      sh.flag = d <= diff_threshold;
    }
    __syncthreads();
    if (sh.flag) break;
    for (int a = tid; a < n_allele; a += kThreads) sh.p[a] = sh.p1[a];
    __syncthreads();
  }
  for (int a = tid; a < n_allele; a += kThreads) prob_out[a] = sh.p[a];
  if (tid == 0) *iters_out = iters;
}

__global__ __launch_bounds__(kThreads) void em_kernel(const uint32_t* sets, const double* weight, double* scale,
                                                      int n_sets, int words, int n_allele, int iter_max,
                                                      double diff_threshold, double* prob_out, int* iters_out) {
  __shared__ EmShared sh;
  em_solve(sh, sets, weight, scale, n_sets, words, n_allele, iter_max, diff_threshold, prob_out, iters_out);
}

// the EM of every gene of a sample in ONE launch: workgroup g solves gene g (gk_sample_em)
struct EmGene { int64_t sets_off, w_off, prob_off; int32_t n_sets, words, n_allele, pad; };
__global__ __launch_bounds__(kThreads) void em_kernel_genes(const EmGene* genes, const uint32_t* sets, const double* weight,
                                                            double* scale, int iter_max, double diff_threshold,
                                                            double* prob_out, int* iters_out) {
  __shared__ EmShared sh;
  const EmGene g = genes[blockIdx.x];
  em_solve(sh, sets + g.sets_off, weight + g.w_off, scale + g.w_off, g.n_sets, g.words, g.n_allele, iter_max,
           diff_threshold, prob_out + g.prob_off, iters_out + blockIdx.x);
}

}  // namespace

extern "C" {

int gk_em_sets(gk_ctx* ctx, gk_tab* tab, gk_dptr d_rows, int64_t n_rows, int32_t vbeg, int32_t vend, gk_dptr d_mask,
               int32_t words, gk_dptr d_sets_out) {
  gk_bind(ctx);
  GK_REQUIRE(ctx && tab, "null pointer");
  GK_REQUIRE(words >= 1 && words <= kMaxWords, "more than 512 alleles per gene are not supported by the EM kernel");
  if (!n_rows) return GK_OK;
  GK_PROF(ctx, GK_K_EM_SETS, GK_KERNEL(em_sets_kernel, dim3((unsigned)((n_rows + kThreads - 1) / kThreads)), dim3(kThreads), 0,
                     ctx->stream, gk_ptr<int32_t>(d_rows), n_rows, tab->d_off, tab->d_ids, vbeg, vend,
                     gk_ptr<uint32_t>(d_mask), words, gk_ptr<uint32_t>(d_sets_out)));
  GK_HIP(hipGetLastError());
  return GK_OK;
}

int gk_em_distinct(gk_ctx* ctx, gk_dptr d_sets, int64_t n_rows, int32_t words, int32_t max_out, uint32_t* sets_out,
                   uint32_t* count_out, int32_t* n_out) {
  gk_bind(ctx);
  GK_REQUIRE(ctx && sets_out && count_out && n_out && max_out > 0, "null pointer");
  GK_REQUIRE(words >= 1 && words <= kMaxWords && n_rows >= 0 && n_rows < (1ll << 31), "bad set geometry");
  *n_out = 0;
  if (!n_rows) return GK_OK;
  hipStream_t st = ctx->stream;
  uint32_t log2 = 10;
  while ((1ull << log2) < (uint64_t)n_rows * 2 && log2 < 28) ++log2;   // load factor <= 0.5 even if all rows differ
  const uint32_t n_slots = 1u << log2;
  int32_t* owner = nullptr;
  uint32_t *count = nullptr, *d_n = nullptr, *d_out_sets = nullptr, *d_out_count = nullptr;
  GK_HIP(gk_pool_malloc(ctx, (void**)&owner, (size_t)n_slots * sizeof(int32_t)));
  GK_HIP(gk_pool_malloc(ctx, (void**)&count, (size_t)n_slots * sizeof(uint32_t)));
  GK_HIP(gk_pool_malloc(ctx, (void**)&d_n, sizeof(uint32_t)));
  GK_HIP(gk_pool_malloc(ctx, (void**)&d_out_sets, (size_t)max_out * words * sizeof(uint32_t)));
  GK_HIP(gk_pool_malloc(ctx, (void**)&d_out_count, (size_t)max_out * sizeof(uint32_t)));
  GK_HIP(hipMemsetAsync(owner, 0xFF, (size_t)n_slots * sizeof(int32_t), st));
  GK_HIP(hipMemsetAsync(count, 0, (size_t)n_slots * sizeof(uint32_t), st));
  GK_HIP(hipMemsetAsync(d_n, 0, sizeof(uint32_t), st));
  GK_PROF(ctx, GK_K_EM_SETS,
          GK_KERNEL(em_distinct_kernel, dim3((unsigned)((n_rows + kThreads - 1) / kThreads)), dim3(kThreads), 0, st,
                             gk_ptr<uint32_t>(d_sets), n_rows, words, owner, count, n_slots - 1));
  GK_PROF(ctx, GK_K_EM_SETS,
          GK_KERNEL(em_distinct_emit, dim3((n_slots + kThreads - 1) / kThreads), dim3(kThreads), 0, st,
                             gk_ptr<uint32_t>(d_sets), words, owner, count, n_slots, (uint32_t)max_out, d_n, d_out_sets,
                             d_out_count));
  GK_HIP(hipGetLastError());
  uint32_t n = 0;
  GK_HIP(hipMemcpyAsync(&n, d_n, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  GK_HIP(hipStreamSynchronize(st));
  int rc = GK_OK;
  if (n > (uint32_t)max_out) {
    gk_set_error("%u distinct candidate sets exceed the output capacity %d", n, max_out);
    rc = GK_ERR_CAPACITY;
  } else if (n) {
    GK_HIP(hipMemcpyAsync(sets_out, d_out_sets, (size_t)n * words * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    GK_HIP(hipMemcpyAsync(count_out, d_out_count, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    GK_HIP(hipStreamSynchronize(st));
  }
  *n_out = (int32_t)n;
  gk_pool_free(ctx, owner); gk_pool_free(ctx, count); gk_pool_free(ctx, d_n);
  gk_pool_free(ctx, d_out_sets); gk_pool_free(ctx, d_out_count);
  return rc;
}

int gk_em_run(gk_ctx* ctx, const uint32_t* sets, const double* weight, int32_t n_sets, int32_t words, int32_t n_allele,
              int32_t iter_max, double diff_threshold, double* prob_out, int32_t* iters_out) {
  gk_bind(ctx);
  GK_REQUIRE(ctx && sets && weight && prob_out && iters_out, "null pointer");
  GK_REQUIRE(n_sets > 0 && words >= 1 && words <= kMaxWords && n_allele >= 1 && n_allele <= words * 32,
             "bad EM geometry");
  hipStream_t st = ctx->stream;
  uint32_t* d_sets = nullptr;
  double *d_w = nullptr, *d_scale = nullptr, *d_prob = nullptr;
  int* d_it = nullptr;
  GK_HIP(gk_pool_malloc(ctx, (void**)&d_sets, (size_t)n_sets * words * sizeof(uint32_t)));
  GK_HIP(gk_pool_malloc(ctx, (void**)&d_w, (size_t)n_sets * sizeof(double)));
  GK_HIP(gk_pool_malloc(ctx, (void**)&d_scale, (size_t)n_sets * sizeof(double)));
  GK_HIP(gk_pool_malloc(ctx, (void**)&d_prob, (size_t)n_allele * sizeof(double)));
  GK_HIP(gk_pool_malloc(ctx, (void**)&d_it, sizeof(int)));
  GK_HIP(hipMemcpyAsync(d_sets, sets, (size_t)n_sets * words * sizeof(uint32_t), hipMemcpyHostToDevice, st));
  GK_HIP(hipMemcpyAsync(d_w, weight, (size_t)n_sets * sizeof(double), hipMemcpyHostToDevice, st));
  GK_PROF(ctx, GK_K_EM_RUN, GK_KERNEL(em_kernel, dim3(1), dim3(kThreads), 0, st, d_sets, d_w, d_scale, n_sets, words, n_allele, iter_max,
                     diff_threshold, d_prob, d_it));
  GK_HIP(hipGetLastError());
  GK_HIP(hipMemcpyAsync(prob_out, d_prob, (size_t)n_allele * sizeof(double), hipMemcpyDeviceToHost, st));
  GK_HIP(hipMemcpyAsync(iters_out, d_it, sizeof(int), hipMemcpyDeviceToHost, st));
  GK_HIP(hipStreamSynchronize(st));
  gk_pool_free(ctx,d_sets); gk_pool_free(ctx,d_w); gk_pool_free(ctx,d_scale); gk_pool_free(ctx,d_prob); gk_pool_free(ctx,d_it);
  return GK_OK;
}

/* The EM strategy for ALL genes of a sample in one call on the calling thread and the context's one stream
 * (kir_typing.py:163-195 is the reference's gene loop, typing_em.py:68-188 the work per gene): candidate sets and their
 * distinct forms of every gene queued together (one wait for the counts, one for the sets), the host half -- the
 * ascending order numpy.unique gives the sets, the reads naming each allele, the empty set dropped -- in C++, and the
 * SQUAREM loops of all genes in ONE launch (a workgroup per gene).  jobs[i]: the gene's rows (NH == 1 pairs), variant span,
 * bit rows; prob_out / count_out: n_allele entries per job, one after the other; per job the distinct sets and the SQUAREM
 * steps come back.  GK_ERR_CAPACITY when a gene has more than 2^18 distinct sets (the caller takes the per-gene calls). */
int gk_sample_em(gk_ctx* ctx, gk_tab* tab, gk_em_job* jobs, int32_t n_jobs, int32_t iter_max, double diff_threshold,
                 double* prob_out, int64_t* count_out) {
  gk_bind(ctx);
  GK_REQUIRE(ctx && tab && jobs && prob_out && count_out && n_jobs >= 0, "null pointer");
  hipStream_t st = ctx->stream;
  struct Work {
    uint32_t *d_sets = nullptr, *d_out_sets = nullptr, *d_out_count = nullptr, *d_n = nullptr, *count = nullptr;
    int32_t* owner = nullptr;
    uint32_t n = 0, cap = 0;
    std::vector<uint32_t> sets, mult;
    std::vector<int64_t> order;
  };
  std::vector<Work> work((size_t)n_jobs);
  std::vector<void*> temps;
  auto take = [&](void** p, size_t bytes) -> hipError_t {
    hipError_t e = gk_pool_malloc(ctx, p, bytes ? bytes : 16);
    if (e == hipSuccess) temps.push_back(*p);
    return e;
  };
  auto done = [&](int rc) { for (void* p : temps) gk_pool_free(ctx, p); return rc; };
  int64_t out_off = 0;
  std::vector<int64_t> prob_off((size_t)n_jobs, 0);
  for (int i = 0; i < n_jobs; ++i) {
    gk_em_job& j = jobs[i];
    j.n_distinct = 0; j.iterations = 0;
    GK_REQUIRE(j.words >= 1 && j.words <= kMaxWords && j.n_allele >= 0 && j.n_allele <= j.words * 32 && j.n_rows >= 0 &&
               j.n_rows < (1ll << 31), "bad EM job");
    prob_off[i] = out_off;
    for (int a = 0; a < j.n_allele; ++a) { prob_out[out_off + a] = 0.0; count_out[out_off + a] = 0; }
    out_off += j.n_allele;
  }
  // ---- phase 1: candidate sets + distinct sets of every gene, queued
  for (int i = 0; i < n_jobs; ++i) {
    gk_em_job& j = jobs[i];
    if (!j.n_rows || !j.n_allele) continue;
    Work& w = work[i];
    uint32_t log2 = 10;
    while ((1ull << log2) < (uint64_t)j.n_rows * 2 && log2 < 28) ++log2;
    const uint32_t n_slots = 1u << log2;
    w.cap = (uint32_t)std::min<int64_t>(j.n_rows, 1ll << 18);
    if (take((void**)&w.d_sets, (size_t)j.n_rows * j.words * sizeof(uint32_t)) != hipSuccess ||
        take((void**)&w.owner, (size_t)n_slots * sizeof(int32_t)) != hipSuccess ||
        take((void**)&w.count, (size_t)n_slots * sizeof(uint32_t)) != hipSuccess ||
        take((void**)&w.d_n, sizeof(uint32_t)) != hipSuccess ||
        take((void**)&w.d_out_sets, (size_t)w.cap * j.words * sizeof(uint32_t)) != hipSuccess ||
        take((void**)&w.d_out_count, (size_t)w.cap * sizeof(uint32_t)) != hipSuccess) {
      gk_set_error("out of device memory for the candidate sets of a gene");
      return done(GK_ERR_HIP);
    }
    const unsigned blocks = (unsigned)((j.n_rows + kThreads - 1) / kThreads);
    GK_PROF(ctx, GK_K_EM_SETS, GK_KERNEL(em_sets_kernel, dim3(blocks), dim3(kThreads), 0, st, gk_ptr<int32_t>(j.d_rows), j.n_rows,
                                         tab->d_off, tab->d_ids, j.vbeg, j.vend, gk_ptr<uint32_t>(j.d_mask), j.words, w.d_sets));
    hipMemsetAsync(w.owner, 0xFF, (size_t)n_slots * sizeof(int32_t), st);
    hipMemsetAsync(w.count, 0, (size_t)n_slots * sizeof(uint32_t), st);
    hipMemsetAsync(w.d_n, 0, sizeof(uint32_t), st);
    GK_PROF(ctx, GK_K_EM_SETS, GK_KERNEL(em_distinct_kernel, dim3(blocks), dim3(kThreads), 0, st, w.d_sets, j.n_rows, j.words,
                                         w.owner, w.count, n_slots - 1));
    GK_PROF(ctx, GK_K_EM_SETS, GK_KERNEL(em_distinct_emit, dim3((n_slots + kThreads - 1) / kThreads), dim3(kThreads), 0, st, w.d_sets,
                                         j.words, w.owner, w.count, n_slots, w.cap, w.d_n, w.d_out_sets, w.d_out_count));
    if (gk_fetch_queue(ctx, &w.n, w.d_n, sizeof(uint32_t)) != hipSuccess) { gk_fetch_cancel(ctx); return done(GK_ERR_HIP); }
  }
  if (hipGetLastError() != hipSuccess || gk_fetch_wait(ctx) != hipSuccess) {
    gk_fetch_cancel(ctx);
    gk_set_error("sample EM: %s", hipGetErrorString(hipGetLastError()));
    return done(GK_ERR_HIP);
  }
  for (int i = 0; i < n_jobs; ++i) {
    Work& w = work[i];
    if (w.n > w.cap) {
      gk_set_error("%u distinct candidate sets exceed the capacity %u of the one-call EM", w.n, w.cap);
      return done(GK_ERR_CAPACITY);
    }
    if (!w.n) continue;
    w.sets.resize((size_t)w.n * jobs[i].words);
    w.mult.resize(w.n);
    if (gk_fetch_queue(ctx, w.sets.data(), w.d_out_sets, w.sets.size() * sizeof(uint32_t)) != hipSuccess ||
        gk_fetch_queue(ctx, w.mult.data(), w.d_out_count, w.mult.size() * sizeof(uint32_t)) != hipSuccess) {
      gk_fetch_cancel(ctx);
      return done(GK_ERR_HIP);
    }
  }
  if (gk_fetch_wait(ctx) != hipSuccess) { gk_fetch_cancel(ctx); return done(GK_ERR_HIP); }
  // ---- phase 2 (host): numpy.unique's order, the reads naming each allele, the empty set dropped
  std::vector<EmGene> genes;
  std::vector<int> gene_job;
  std::vector<uint32_t> all_sets;
  std::vector<double> all_w;
  for (int i = 0; i < n_jobs; ++i) {
    Work& w = work[i];
    gk_em_job& j = jobs[i];
    j.n_distinct = (int32_t)w.n;
    if (!w.n) continue;
    const int words = j.words;
    w.order.resize(w.n);
    for (uint32_t u = 0; u < w.n; ++u) w.order[u] = u;
    std::sort(w.order.begin(), w.order.end(), [&](int64_t x, int64_t y) {
      const uint32_t *a = w.sets.data() + (size_t)x * words, *b = w.sets.data() + (size_t)y * words;
      for (int q = 0; q < words; ++q)
        if (a[q] != b[q]) return a[q] < b[q];
      return false;
    });
    EmGene g{(int64_t)all_sets.size(), (int64_t)all_w.size(), prob_off[i], 0, words, j.n_allele, 0};
    for (int64_t u : w.order) {
      const uint32_t* row = w.sets.data() + (size_t)u * words;
      bool any = false;
      for (int q = 0; q < words; ++q) {
        uint32_t bits = row[q];
        any |= bits != 0;
        while (bits) {
          const int b = __builtin_ctz(bits);
          bits &= bits - 1;
          const int a = q * 32 + b;
          if (a < j.n_allele) count_out[prob_off[i] + a] += (int64_t)w.mult[(size_t)u];
        }
      }
      if (!any) continue;
      all_sets.insert(all_sets.end(), row, row + words);
      all_w.push_back((double)w.mult[(size_t)u]);
      g.n_sets++;
    }
    if (g.n_sets) { genes.push_back(g); gene_job.push_back(i); }
  }
  if (genes.empty()) return done(GK_OK);
  // ---- phase 3: every gene's SQUAREM loop in one launch
  EmGene* d_genes = nullptr;
  uint32_t* d_all_sets = nullptr;
  double *d_all_w = nullptr, *d_scale = nullptr, *d_prob = nullptr;
  int* d_it = nullptr;
  if (take((void**)&d_genes, genes.size() * sizeof(EmGene)) != hipSuccess ||
      take((void**)&d_all_sets, all_sets.size() * sizeof(uint32_t)) != hipSuccess ||
      take((void**)&d_all_w, all_w.size() * sizeof(double)) != hipSuccess ||
      take((void**)&d_scale, all_w.size() * sizeof(double)) != hipSuccess ||
      take((void**)&d_prob, (size_t)std::max<int64_t>(out_off, 1) * sizeof(double)) != hipSuccess ||
      take((void**)&d_it, genes.size() * sizeof(int)) != hipSuccess) {
    gk_set_error("out of device memory for the EM of a sample");
    return done(GK_ERR_HIP);
  }
  hipMemcpyAsync(d_genes, genes.data(), genes.size() * sizeof(EmGene), hipMemcpyHostToDevice, st);
  hipMemcpyAsync(d_all_sets, all_sets.data(), all_sets.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st);
  hipMemcpyAsync(d_all_w, all_w.data(), all_w.size() * sizeof(double), hipMemcpyHostToDevice, st);
  hipMemsetAsync(d_prob, 0, (size_t)std::max<int64_t>(out_off, 1) * sizeof(double), st);
  GK_PROF(ctx, GK_K_EM_RUN, GK_KERNEL(em_kernel_genes, dim3((unsigned)genes.size()), dim3(kThreads), 0, st, d_genes, d_all_sets, d_all_w,
                                      d_scale, iter_max, diff_threshold, d_prob, d_it));
  std::vector<int> iters(genes.size(), 0);
  std::vector<double> probs((size_t)out_off, 0.0);
  hipMemcpyAsync(probs.data(), d_prob, (size_t)out_off * sizeof(double), hipMemcpyDeviceToHost, st);
  hipMemcpyAsync(iters.data(), d_it, genes.size() * sizeof(int), hipMemcpyDeviceToHost, st);
  if (hipGetLastError() != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
    gk_set_error("sample EM: %s", hipGetErrorString(hipGetLastError()));
    return done(GK_ERR_HIP);
  }
  for (size_t k = 0; k < genes.size(); ++k) {
    const int i = gene_job[k];
    jobs[i].iterations = iters[k];
    for (int a = 0; a < jobs[i].n_allele; ++a) prob_out[prob_off[i] + a] = probs[(size_t)(prob_off[i] + a)];
  }
  return done(GK_OK);
}

}  // extern "C"


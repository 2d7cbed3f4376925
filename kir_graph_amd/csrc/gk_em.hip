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

__global__ __launch_bounds__(kThreads) void em_kernel(const uint32_t* sets, const double* weight, double* scale,
                                                      int n_sets, int words, int n_allele, int iter_max,
                                                      double diff_threshold, double* prob_out, int* iters_out) {
  __shared__ EmShared sh;
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

}  // extern "C"

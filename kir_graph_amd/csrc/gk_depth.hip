// Per-position read depth of the filtered, uniquely mapped pairs.
//
// Replaces `samtools depth -aa {name}.no_multi.bam` (samtools_utils.py:9-14, main.py:152-158): the
// reference rewrites the filter-passing pairs with NH == 1 to a BAM and lets samtools count, for
// every backbone position, the reads whose aligned (M) bases cover it; deletions are not counted,
// soft clips are not aligned, mates are counted independently (no overlap removal).
// Here: one thread per mate adds +1 / -1 at the ends of every M run into a difference array over
// the concatenated backbones, one exclusive scan turns it into depths.
#include "gk_common.h"

namespace {

constexpr int kThreads = 256;

__global__ __launch_bounds__(kThreads) void depth_mark(const gk_mate* __restrict__ mates,
                                                       const gk_mate_wide* __restrict__ wide,
                                                       const int32_t* __restrict__ pair_src,
                                                       const uint8_t* __restrict__ pair_nh, int64_t n_valid,
                                                       int multiple, const int64_t* __restrict__ gene_off, int n_gene,
                                                       uint32_t* __restrict__ diff) {
  const int64_t t = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (t >= 2 * n_valid) return;
  const int64_t i = t >> 1;
  if (!multiple && pair_nh[i] != 1) return;
  const gk_mate& m = mates[2 * (int64_t)pair_src[i] + (t & 1)];
  if (m.ref >= n_gene) return;
  const int64_t base = gene_off[m.ref], len = gene_off[m.ref + 1] - base;
  int64_t cur = m.pos0;
  auto run = [&](uint32_t op, uint32_t n) {
    if (op == GK_CIG_M) {
      const int64_t a = cur < 0 ? 0 : cur, b = cur + n > len ? len : cur + n;
      if (b > a) {
        atomicAdd(&diff[base + a], 1u);
        atomicAdd(&diff[base + b], 0xFFFFFFFFu);   // -1 (mod 2^32); diff has one slot past the end
      }
      cur += n;
    } else if (op == GK_CIG_D) {
      cur += n;
    }
  };
  if (m.n_cig == GK_SPILLED) {   // the pair is in the wide array (gk_mate_wide): its CIGAR is there
    if (!wide) return;
    const gk_mate_wide& x = wide[2 * (int64_t)m.ins[0] + (t & 1)];
    const int n_cig = x.n_cig < GK_WIDE_CIG ? x.n_cig : GK_WIDE_CIG;
    for (int c = 0; c < n_cig; ++c) run(x.cig[c] & 15u, x.cig[c] >> 4);
    return;
  }
  const int n_cig = m.n_cig < GK_MAX_CIG ? m.n_cig : GK_MAX_CIG;
  for (int c = 0; c < n_cig; ++c) run(m.cig[c] & 15u, m.cig[c] >> 4);
}

__global__ __launch_bounds__(kThreads) void depth_finish(const uint32_t* excl, const uint32_t* diff, int64_t n,
                                                         uint32_t* depth /* may alias diff */) {
  const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (i < n) depth[i] = excl[i] + diff[i];
}

}  // namespace

extern "C" int gk_depth(gk_ctx* ctx, gk_tab* tab, gk_dptr d_mates, int32_t multiple, const int64_t* gene_off,
                        int32_t n_gene, uint32_t* depth_out) {
  gk_bind(ctx);
  GK_REQUIRE(ctx && tab && gene_off && depth_out && n_gene > 0, "bad depth arguments");
  GK_REQUIRE(tab->d_pair_src, "depth needs a tabulation made from packed records");
  const int64_t total = gene_off[n_gene];
  hipStream_t st = ctx->stream;
  uint32_t *diff = nullptr, *scan = nullptr;
  int64_t* d_off = nullptr;
  GK_HIP(gk_pool_malloc(ctx, (void**)&diff, (size_t)(total + 1) * sizeof(uint32_t)));
  GK_HIP(gk_pool_malloc(ctx, (void**)&scan, (size_t)(total + 1) * sizeof(uint32_t)));
  GK_HIP(gk_pool_malloc(ctx, (void**)&d_off, (size_t)(n_gene + 1) * sizeof(int64_t)));
  GK_HIP(hipMemsetAsync(diff, 0, (size_t)(total + 1) * sizeof(uint32_t), st));
  GK_HIP(hipMemcpyAsync(d_off, gene_off, (size_t)(n_gene + 1) * sizeof(int64_t), hipMemcpyHostToDevice, st));
  if (tab->n_valid)
    GK_KERNEL(depth_mark, dim3((unsigned)((2 * tab->n_valid + kThreads - 1) / kThreads)), dim3(kThreads), 0, st,
                       gk_ptr<const gk_mate>(d_mates), tab->d_wide, tab->d_pair_src, tab->d_pair_nh, tab->n_valid, multiple, d_off,
                       n_gene, diff);
  GK_HIP(hipMemcpyAsync(scan, diff, (size_t)(total + 1) * sizeof(uint32_t), hipMemcpyDeviceToDevice, st));
  int rc = gk_scan_u32(ctx, scan, total + 1, nullptr);
  if (rc) return rc;
  GK_KERNEL(depth_finish, dim3((unsigned)((total + kThreads - 1) / kThreads)), dim3(kThreads), 0, st, scan, diff,
                     total, diff);
  GK_HIP(hipGetLastError());
  GK_HIP(hipMemcpyAsync(depth_out, diff, (size_t)total * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  GK_HIP(hipStreamSynchronize(st));
  gk_pool_free(ctx, diff);
  gk_pool_free(ctx, scan);
  gk_pool_free(ctx, d_off);
  return GK_OK;
}

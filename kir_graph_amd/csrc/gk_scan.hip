// Exclusive scan and stable compaction on the context stream (wave64 shuffles + LDS).
#include "gk_common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kItems = 8;
constexpr int kTile = kThreads * kItems;

__device__ inline uint32_t wave_incl_scan(uint32_t v, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint32_t o = __shfl_up(v, d, 64);
    if (lane >= d) v += o;
  }
  return v;
}

// each block scans kTile elements exclusively in place and emits its total
__global__ __launch_bounds__(kThreads) void scan_tiles(uint32_t* data, int64_t n, uint32_t* tile_sum) {
  __shared__ uint32_t wave_tot[kThreads / 64];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int64_t base = (int64_t)blockIdx.x * kTile + (int64_t)tid * kItems;
  uint32_t v[kItems];
  uint32_t sum = 0;
#pragma unroll
  for (int k = 0; k < kItems; ++k) {
    int64_t i = base + k;
    v[k] = i < n ? data[i] : 0u;
    sum += v[k];
  }
  uint32_t incl = wave_incl_scan(sum, lane);
  if (lane == 63) wave_tot[wid] = incl;
  __syncthreads();
  uint32_t off = 0;
  for (int w = 0; w < wid; ++w) off += wave_tot[w];
  uint32_t run = off + incl - sum;
#pragma unroll
  for (int k = 0; k < kItems; ++k) {
    int64_t i = base + k;
    if (i < n) data[i] = run;
    run += v[k];
  }
  if (tid == kThreads - 1) tile_sum[blockIdx.x] = off + incl;
}

__global__ __launch_bounds__(kThreads) void add_tile_offsets(uint32_t* data, int64_t n, const uint32_t* tile_off) {
  const int64_t base = (int64_t)blockIdx.x * kTile + (int64_t)threadIdx.x * kItems;
  const uint32_t off = tile_off[blockIdx.x];
#pragma unroll
  for (int k = 0; k < kItems; ++k) {
    int64_t i = base + k;
    if (i < n) data[i] += off;
  }
}


__global__ __launch_bounds__(kThreads) void flags_to_u32(const uint32_t* flag, uint32_t* out, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (i < n) out[i] = flag[i] != 0;
}

__global__ __launch_bounds__(kThreads) void scatter_selected(const uint32_t* flag, const uint32_t* pos,
                                                             const int32_t* values, int64_t n, int32_t* out) {
  int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (i < n && flag[i]) out[pos[i]] = values ? values[i] : (int32_t)i;
}

// ---- two-launch stable compaction for up to kCompactMaxBlocks * kCompactTile items: per-block
// counts, then every block sums the counts before it (<= 4096 loads) and scatters its own items.
constexpr int kCompactItems = 4;
constexpr int kCompactTile = kThreads * kCompactItems;   // 1024 items per block
constexpr int kCompactMaxBlocks = 4096;

__global__ __launch_bounds__(kThreads) void compact_count(const uint32_t* __restrict__ flag, int64_t n,
                                                          uint32_t* __restrict__ block_count) {
  __shared__ uint32_t wave_tot[kThreads / 64];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int64_t base = (int64_t)blockIdx.x * kCompactTile + (int64_t)tid * kCompactItems;
  uint32_t c = 0;
#pragma unroll
  for (int k = 0; k < kCompactItems; ++k) c += (base + k < n && flag[base + k]) ? 1u : 0u;
  const uint32_t incl = wave_incl_scan(c, lane);
  if (lane == 63) wave_tot[wid] = incl;
  __syncthreads();
  if (tid == 0) block_count[blockIdx.x] = wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
}

__global__ __launch_bounds__(kThreads) void compact_scatter(const uint32_t* __restrict__ flag,
                                                            const int32_t* __restrict__ values, int64_t n,
                                                            const uint32_t* __restrict__ block_count,
                                                            int32_t* __restrict__ out, uint32_t* __restrict__ total) {
  __shared__ uint32_t wave_tot[kThreads / 64];
  __shared__ uint32_t wave_pre[kThreads / 64];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  // items kept by the blocks before this one
  uint32_t before = 0;
  for (int b = tid; b < (int)blockIdx.x; b += kThreads) before += block_count[b];
  before = wave_incl_scan(before, lane);
  if (lane == 63) wave_pre[wid] = before;
  const int64_t base = (int64_t)blockIdx.x * kCompactTile + (int64_t)tid * kCompactItems;
  bool keep[kCompactItems];
  uint32_t c = 0;
#pragma unroll
  for (int k = 0; k < kCompactItems; ++k) {
    keep[k] = base + k < n && flag[base + k];
    c += keep[k] ? 1u : 0u;
  }
  const uint32_t incl = wave_incl_scan(c, lane);
  if (lane == 63) wave_tot[wid] = incl;
  __syncthreads();
  uint32_t pos = wave_pre[0] + wave_pre[1] + wave_pre[2] + wave_pre[3] + incl - c;
  for (int w = 0; w < wid; ++w) pos += wave_tot[w];
#pragma unroll
  for (int k = 0; k < kCompactItems; ++k)
    if (keep[k]) out[pos++] = values ? values[base + k] : (int32_t)(base + k);
  if (blockIdx.x == gridDim.x - 1 && tid == kThreads - 1) *total = pos;
}

}  // namespace

static int scan_rec(gk_ctx* ctx, uint32_t* d, int64_t n, uint32_t* sums_area, uint32_t* d_total) {
  // sums_area: scratch large enough for all levels
  int64_t tiles = (n + kTile - 1) / kTile;
  if (tiles <= 0) {
    if (d_total) GK_HIP(hipMemsetAsync(d_total, 0, sizeof(uint32_t), ctx->stream));
    return GK_OK;
  }
  GK_PROF(ctx, "scan_tiles", GK_KERNEL(scan_tiles, dim3((unsigned)tiles), dim3(kThreads), 0, ctx->stream, d, n, sums_area));
  if (tiles > 1) {
    int rc = scan_rec(ctx, sums_area, tiles, sums_area + tiles, d_total);
    if (rc) return rc;
    GK_PROF(ctx, "add_tile_offsets", GK_KERNEL(add_tile_offsets, dim3((unsigned)tiles), dim3(kThreads), 0, ctx->stream, d, n, sums_area));
  } else if (d_total) {
    GK_HIP(hipMemcpyAsync(d_total, sums_area, sizeof(uint32_t), hipMemcpyDeviceToDevice, ctx->stream));
  }
  GK_HIP(hipGetLastError());
  return GK_OK;
}

int gk_scan_u32(gk_ctx* ctx, uint32_t* d_data, int64_t n, uint32_t* d_total) {
  gk_bind(ctx);
  // scratch for tile sums of every level: tiles + tiles/kTile + ... <= tiles + tiles/1024 + 64
  int64_t tiles = (n + kTile - 1) / kTile;
  size_t need = (size_t)(tiles + tiles / 1024 + 4096) * sizeof(uint32_t);
  uint32_t* area = nullptr;
  GK_HIP(gk_pool_malloc(ctx, (void**)&area, need));
  const int rc = scan_rec(ctx, d_data, n, area, d_total);
  gk_pool_free(ctx, area);   // stream-ordered reuse
  return rc;
}

// The compaction queued on the stream, its total left on the device (*d_total_out points into a temporary): nothing is
// waited for.  `temps` are the pool blocks to give back once the stream has passed the compaction (stream-ordered reuse:
// at once is fine too, as gk_compact does).
int gk_compact_enqueue(gk_ctx* ctx, const uint32_t* d_flag, const int32_t* d_values, int64_t n, int32_t* d_out,
                       uint32_t** d_total_out, std::vector<void*>& temps) {
  gk_bind(ctx);
  *d_total_out = nullptr;
  if (n <= 0) return GK_OK;
  const int64_t blocks2 = (n + kCompactTile - 1) / kCompactTile;
  if (blocks2 <= kCompactMaxBlocks) {
    uint32_t* cnt = nullptr;
    GK_HIP(gk_pool_malloc(ctx, (void**)&cnt, (size_t)(blocks2 + 1) * sizeof(uint32_t)));
    temps.push_back(cnt);
    GK_PROF(ctx, "compact_count", GK_KERNEL(compact_count, dim3((unsigned)blocks2), dim3(kThreads), 0, ctx->stream,
                                             d_flag, n, cnt));
    GK_PROF(ctx, "compact_scatter", GK_KERNEL(compact_scatter, dim3((unsigned)blocks2), dim3(kThreads), 0, ctx->stream,
                                             d_flag, d_values, n, cnt, d_out, cnt + blocks2));
    GK_HIP(hipGetLastError());
    *d_total_out = cnt + blocks2;
    return GK_OK;
  }
  uint32_t* pos = nullptr;
  GK_HIP(gk_pool_malloc(ctx, (void**)&pos, (size_t)(n + 1) * sizeof(uint32_t)));
  temps.push_back(pos);
  unsigned blocks = (unsigned)((n + kThreads - 1) / kThreads);
  GK_PROF(ctx, "flags_to_u32", GK_KERNEL(flags_to_u32, dim3(blocks), dim3(kThreads), 0, ctx->stream, d_flag, pos, n));
  int rc = gk_scan_u32(ctx, pos, n, pos + n);
  if (rc) return rc;
  GK_PROF(ctx, "scatter_selected", GK_KERNEL(scatter_selected, dim3(blocks), dim3(kThreads), 0, ctx->stream, d_flag, pos, d_values, n, d_out));
  GK_HIP(hipGetLastError());
  *d_total_out = pos + n;
  return GK_OK;
}

int gk_compact(gk_ctx* ctx, const uint32_t* d_flag, const int32_t* d_values, int64_t n, int32_t* d_out,
               int64_t* n_out) {
  if (n <= 0) {
    if (n_out) *n_out = 0;
    return GK_OK;
  }
  std::vector<void*> temps;
  uint32_t* d_total = nullptr;
  int rc = gk_compact_enqueue(ctx, d_flag, d_values, n, d_out, &d_total, temps);
  uint32_t total = 0;
  if (rc == GK_OK && gk_fetch(ctx, &total, d_total, sizeof(uint32_t)) != hipSuccess) {
    gk_set_error("compaction: fetching the count failed");
    rc = GK_ERR_HIP;
  }
  for (void* t : temps) gk_pool_free(ctx, t);
  if (rc == GK_OK && n_out) *n_out = total;
  return rc;
}

// Likelihood search reductions with numpy's exact summation tree.
//
//   gk_maxsum    AlleleTyping.addCandidate   typing_mulit_allele.py:514 (column sums) and 540-542
//                (sum over reads of max(log_probs[:, a], allele_prob[:, t]))
//   gk_fraction  typing_mulit_allele.py:575-580 (share of reads owned by each allele of a set)
//   gk_setmax    typing_mulit_allele.py:569     (allele_prob of a set = row-wise max of its columns)
//
// Why a tree: candidate sets tie mathematically all the time and the reference separates them by
// float64 rounding noise, so ranks are only reproducible with numpy's add.reduce order
// (note to a12-a14): rows are cut into 8192-row chunks; a chunk is summed by
// recursive halving (n2 = n/2 - (n/2)%8) down to blocks of <= 128 rows; a block uses 8 strided
// accumulators combined as ((0+1)+(2+3))+((4+5)+(6+7)) plus a sequential tail; chunk sums are
// accumulated sequentially.
//
// Mapping to CDNA4: a (max,+) contraction has no multiply, so MFMA does not apply; the kernel is
// f64-VALU bound.  The 8 strided accumulators of numpy's block loop are spread over 8 LANES
// (lane j owns rows == j mod 8), which leaves one accumulator register per output and lets every
// lane carry a TT x TA register tile.  The recursion stack lives in the same 8 lanes (lane s holds
// stack slot s), so the whole tree runs without dynamically indexed registers.  L is column-major
// [allele][row]: row blocks are staged through LDS with coalesced 512-byte wave loads, the previous
// set's row-wise max is formed while staging (no R x T temporary in HBM, the reference
// materialises T x R x A).
#include <algorithm>

#include "gk_common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kBlockRows = 128;   // numpy PW_BLOCKSIZE
constexpr int kChunkRows = 8192;  // numpy buffer size for add.reduce
constexpr int kLD = kBlockRows + 8;  // LDS column stride (doubles): +64 B de-phases the 4 allele groups
constexpr int TT = 4, TA = 4;
constexpr int kTileT = 8 * TT;    // 32 sets per workgroup
constexpr int kTileA = 4 * TA;    // 16 candidate columns per workgroup
constexpr int kMaxC = 8;          // alleles per set (copy number) supported

struct Op { int32_t kind, a, b, c; };   // kind 0: LEAF(start=a, len=b, slot=c); kind 1: ADD(slot a += slot b)

__device__ inline double vmax(double a, double b) {
  double r;
  asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));   // fmax() would add two canonicalising v_max
  return r;
}

__device__ inline double group_sum8(double v) {
  // ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) over the 8 lanes of a group; every lane gets the result
  v += __shfl_xor(v, 1, 64);
  v += __shfl_xor(v, 2, 64);
  v += __shfl_xor(v, 4, 64);
  return v;
}

__global__ __launch_bounds__(kThreads) void maxsum_chunks(const double* __restrict__ L, int64_t n_rows, int64_t ld,
                                                          const int32_t* __restrict__ ids, int n_sets, int c_prev,
                                                          const int32_t* __restrict__ cols, int n_cols,
                                                          const Op* __restrict__ ops_full, int n_ops_full,
                                                          const Op* __restrict__ ops_tail, int n_ops_tail,
                                                          int n_chunks, double* __restrict__ partial) {
  __shared__ double Pt[kTileT * kLD];
  __shared__ double Lt[kTileA * kLD];
  __shared__ int32_t p_col[kTileT * kMaxC];
  __shared__ int32_t l_col[kTileA];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int j = lane & 7, g = lane >> 3, gt = g >> 2, ga = g & 3;
  const int tiles_a = (n_cols + kTileA - 1) / kTileA;
  const int tile_t = blockIdx.x / tiles_a, tile_a = blockIdx.x % tiles_a;
  const int t0 = tile_t * kTileT, a0 = tile_a * kTileA;
  const int chunk = blockIdx.y;
  const int64_t chunk_row0 = (int64_t)chunk * kChunkRows;
  const bool tail = (chunk == n_chunks - 1) && (n_rows - chunk_row0 < kChunkRows);
  const Op* ops = tail ? ops_tail : ops_full;
  const int n_ops = tail ? n_ops_tail : n_ops_full;

  for (int i = tid; i < kTileT * kMaxC; i += kThreads) {
    const int t = t0 + i / kMaxC, k = i % kMaxC;
    p_col[i] = (t < n_sets && k < c_prev) ? ids[t * c_prev + k] : -1;
  }
  if (tid < kTileA) l_col[tid] = (a0 + tid < n_cols) ? cols[a0 + tid] : -1;
  __syncthreads();

  const int t_loc = (wid * 2 + gt) * TT;   // first of TT consecutive sets of this lane
  double st[TT][TA];
#pragma unroll
  for (int x = 0; x < TT; ++x)
#pragma unroll
    for (int y = 0; y < TA; ++y) st[x][y] = 0.0;

  for (int o = 0; o < n_ops; ++o) {
    const Op op = ops[o];
    if (op.kind == 0) {
      const int64_t r0 = chunk_row0 + op.a;
      const int len = op.b;
      // ---- stage rows [r0, r0+len) : candidate columns and the previous sets' row-wise max
      __syncthreads();
      for (int i = tid; i < kTileA * kBlockRows; i += kThreads) {
        const int col = i >> 7, row = i & (kBlockRows - 1);
        const int cidx = l_col[col];
        double v = 0.0;
        if (row < len && cidx >= 0) v = L[(int64_t)cidx * ld + r0 + row];
        Lt[col * kLD + row] = v;
      }
      for (int i = tid; i < kTileT * kBlockRows; i += kThreads) {
        const int t = i >> 7, row = i & (kBlockRows - 1);
        double v = -__builtin_huge_val();
        if (row < len) {
          for (int k = 0; k < c_prev; ++k) {
            const int cidx = p_col[t * kMaxC + k];
            if (cidx >= 0) v = vmax(v, L[(int64_t)cidx * ld + r0 + row]);
          }
        }
        Pt[t * kLD + row] = v;
      }
      __syncthreads();
      // ---- numpy pairwise block: 8 strided accumulators = 8 lanes
      double acc[TT][TA];
      const int n8 = len - (len & 7);
      if (len < 8) {
#pragma unroll
        for (int x = 0; x < TT; ++x)
#pragma unroll
          for (int y = 0; y < TA; ++y) acc[x][y] = 0.0;
        for (int r = 0; r < len; ++r) {
#pragma unroll
          for (int x = 0; x < TT; ++x) {
            const double p = Pt[(t_loc + x) * kLD + r];
#pragma unroll
            for (int y = 0; y < TA; ++y) acc[x][y] += vmax(p, Lt[(ga + 4 * y) * kLD + r]);
          }
        }
      } else {
        {
          double p[TT], l[TA];
#pragma unroll
          for (int x = 0; x < TT; ++x) p[x] = Pt[(t_loc + x) * kLD + j];
#pragma unroll
          for (int y = 0; y < TA; ++y) l[y] = Lt[(ga + 4 * y) * kLD + j];
#pragma unroll
          for (int x = 0; x < TT; ++x)
#pragma unroll
            for (int y = 0; y < TA; ++y) acc[x][y] = vmax(p[x], l[y]);
        }
        for (int r = 8 + j; r < n8; r += 8) {
          double p[TT], l[TA];
#pragma unroll
          for (int x = 0; x < TT; ++x) p[x] = Pt[(t_loc + x) * kLD + r];
#pragma unroll
          for (int y = 0; y < TA; ++y) l[y] = Lt[(ga + 4 * y) * kLD + r];
#pragma unroll
          for (int x = 0; x < TT; ++x)
#pragma unroll
            for (int y = 0; y < TA; ++y) acc[x][y] += vmax(p[x], l[y]);
        }
#pragma unroll
        for (int x = 0; x < TT; ++x)
#pragma unroll
          for (int y = 0; y < TA; ++y) acc[x][y] = group_sum8(acc[x][y]);
        for (int r = n8; r < len; ++r) {   // sequential tail, same in every lane
#pragma unroll
          for (int x = 0; x < TT; ++x) {
            const double p = Pt[(t_loc + x) * kLD + r];
#pragma unroll
            for (int y = 0; y < TA; ++y) acc[x][y] += vmax(p, Lt[(ga + 4 * y) * kLD + r]);
          }
        }
      }
      const bool mine = (j == op.c);
#pragma unroll
      for (int x = 0; x < TT; ++x)
#pragma unroll
        for (int y = 0; y < TA; ++y) st[x][y] = mine ? acc[x][y] : st[x][y];
    } else {
      const int src = (lane & ~7) | op.b;
      const bool mine = (j == op.a);
#pragma unroll
      for (int x = 0; x < TT; ++x)
#pragma unroll
        for (int y = 0; y < TA; ++y) {
          const double other = __shfl(st[x][y], src, 64);
          st[x][y] = mine ? st[x][y] + other : st[x][y];
        }
    }
  }
  if (j == 0) {
#pragma unroll
    for (int x = 0; x < TT; ++x) {
      const int t = t0 + t_loc + x;
#pragma unroll
      for (int y = 0; y < TA; ++y) {
        const int a = a0 + ga + 4 * y;
        if (t < n_sets && a < n_cols) partial[((int64_t)chunk * n_sets + t) * n_cols + a] = st[x][y];
      }
    }
  }
}

// chunk sums are accumulated sequentially (numpy's outer reduce loop)
__global__ __launch_bounds__(kThreads) void combine_chunks(const double* partial, int64_t n_out, int n_chunks,
                                                           double scale_div, double* out) {
  const int64_t o = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (o >= n_out) return;
  double r = partial[o];
  for (int q = 1; q < n_chunks; ++q) r += partial[(int64_t)q * n_out + o];
  out[o] = scale_div != 0.0 ? r / scale_div : r;
}

// abundance share: one 8-lane group per allele set, rows read straight from HBM/L2
__global__ __launch_bounds__(kThreads) void fraction_chunks(const double* __restrict__ L, int64_t n_rows, int64_t ld,
                                                            const int32_t* __restrict__ ids, int n_sets, int c,
                                                            const Op* __restrict__ ops_full, int n_ops_full,
                                                            const Op* __restrict__ ops_tail, int n_ops_tail,
                                                            int n_chunks, double* __restrict__ partial) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int j = lane & 7;
  const int k = blockIdx.x * (kThreads / 8) + (tid >> 3);
  const int chunk = blockIdx.y;
  const int64_t chunk_row0 = (int64_t)chunk * kChunkRows;
  const bool tail = (chunk == n_chunks - 1) && (n_rows - chunk_row0 < kChunkRows);
  const Op* ops = tail ? ops_tail : ops_full;
  const int n_ops = tail ? n_ops_tail : n_ops_full;
  const bool live = k < n_sets;
  const double* colp[kMaxC];
#pragma unroll
  for (int q = 0; q < kMaxC; ++q) colp[q] = L + (int64_t)((live && q < c) ? ids[k * c + q] : 0) * ld;
  double st[kMaxC];
#pragma unroll
  for (int q = 0; q < kMaxC; ++q) st[q] = 0.0;

  auto terms = [&](int64_t r, double* out_t) {
    double v[kMaxC];
    double best = -__builtin_huge_val();
#pragma unroll
    for (int q = 0; q < kMaxC; ++q) {
      v[q] = q < c ? colp[q][r] : -__builtin_huge_val();
      if (q < c) best = vmax(best, v[q]);
    }
    int cnt = 0;
#pragma unroll
    for (int q = 0; q < kMaxC; ++q) cnt += (q < c && v[q] == best) ? 1 : 0;
    const double share = 1.0 / (double)cnt;
#pragma unroll
    for (int q = 0; q < kMaxC; ++q) out_t[q] = (q < c && v[q] == best) ? share : 0.0;
  };

  for (int o = 0; o < n_ops; ++o) {
    const Op op = ops[o];
    if (op.kind == 0) {
      const int64_t r0 = chunk_row0 + op.a;
      const int len = op.b;
      const int n8 = len - (len & 7);
      double acc[kMaxC], t[kMaxC];
      if (len < 8) {
#pragma unroll
        for (int q = 0; q < kMaxC; ++q) acc[q] = 0.0;
        for (int r = 0; r < len; ++r) {
          terms(r0 + r, t);
#pragma unroll
          for (int q = 0; q < kMaxC; ++q) acc[q] += t[q];
        }
      } else {
        terms(r0 + j, acc);
        for (int r = 8 + j; r < n8; r += 8) {
          terms(r0 + r, t);
#pragma unroll
          for (int q = 0; q < kMaxC; ++q) acc[q] += t[q];
        }
#pragma unroll
        for (int q = 0; q < kMaxC; ++q) acc[q] = group_sum8(acc[q]);
        for (int r = n8; r < len; ++r) {
          terms(r0 + r, t);
#pragma unroll
          for (int q = 0; q < kMaxC; ++q) acc[q] += t[q];
        }
      }
      const bool mine = (j == op.c);
#pragma unroll
      for (int q = 0; q < kMaxC; ++q) st[q] = mine ? acc[q] : st[q];
    } else {
      const int src = (lane & ~7) | op.b;
      const bool mine = (j == op.a);
#pragma unroll
      for (int q = 0; q < kMaxC; ++q) {
        const double other = __shfl(st[q], src, 64);
        st[q] = mine ? st[q] + other : st[q];
      }
    }
  }
  if (j == 0 && live) {
#pragma unroll
    for (int q = 0; q < kMaxC; ++q)
      if (q < c) partial[((int64_t)chunk * n_sets + k) * c + q] = st[q];
  }
}

__global__ __launch_bounds__(kThreads) void setmax_kernel(const double* __restrict__ L, int64_t n_rows, int64_t ld,
                                                          const int32_t* __restrict__ ids, int n_sets, int c,
                                                          double* __restrict__ P) {
  const int t = blockIdx.y;
  for (int64_t r = (int64_t)blockIdx.x * kThreads + threadIdx.x; r < n_rows; r += (int64_t)gridDim.x * kThreads) {
    double v = -__builtin_huge_val();
    for (int k = 0; k < c; ++k) v = vmax(v, L[(int64_t)ids[t * c + k] * ld + r]);
    P[(int64_t)t * ld + r] = v;
  }
}

// numpy pairwise_sum recursion for one chunk, as a post-order stack program
void build_plan(int start, int n, int slot, std::vector<Op>& ops) {
  if (n <= kBlockRows) {
    ops.push_back(Op{0, start, n, slot});
    return;
  }
  int n2 = n / 2;
  n2 -= n2 % 8;
  build_plan(start, n2, slot, ops);
  build_plan(start + n2, n - n2, slot + 1, ops);
  ops.push_back(Op{1, slot, slot + 1, 0});
}

struct Plans {
  Op* d_ops = nullptr;   // full plan followed by tail plan
  int n_full = 0, n_tail = 0, n_chunks = 0;
};

int make_plans(gk_ctx* ctx, int64_t n_rows, Plans& p) {
  std::vector<Op> full, tail;
  build_plan(0, kChunkRows, 0, full);
  p.n_chunks = (int)((n_rows + kChunkRows - 1) / kChunkRows);
  const int rem = (int)(n_rows - (int64_t)(p.n_chunks - 1) * kChunkRows);
  if (rem < kChunkRows) build_plan(0, rem, 0, tail);
  p.n_full = (int)full.size();
  p.n_tail = (int)tail.size();
  std::vector<Op> all(full);
  all.insert(all.end(), tail.begin(), tail.end());
  GK_HIP(hipMalloc((void**)&p.d_ops, all.size() * sizeof(Op)));
  GK_HIP(hipMemcpyAsync(p.d_ops, all.data(), all.size() * sizeof(Op), hipMemcpyHostToDevice, ctx->stream));
  GK_HIP(hipStreamSynchronize(ctx->stream));
  return GK_OK;
}

}  // namespace

extern "C" {

int gk_maxsum(gk_ctx* ctx, gk_dptr d_L, int64_t n_rows, int64_t ld, const int32_t* ids, int32_t n_sets, int32_t c_prev,
              const int32_t* cols, int32_t n_cols, double* out) {
  GK_REQUIRE(ctx && cols && out && n_rows > 0 && ld >= n_rows && n_cols > 0, "bad maxsum arguments");
  GK_REQUIRE(c_prev >= 0 && c_prev <= kMaxC, "copy number beyond supported set size");
  GK_REQUIRE(n_sets >= 1 && (c_prev == 0 || ids), "missing previous sets");
  Plans pl;
  int rc = make_plans(ctx, n_rows, pl);
  if (rc) return rc;
  hipStream_t st = ctx->stream;
  int32_t *d_ids = nullptr, *d_cols = nullptr;
  double *d_partial = nullptr, *d_out = nullptr;
  const int64_t n_out = (int64_t)n_sets * n_cols;
  GK_HIP(hipMalloc((void**)&d_ids, (size_t)std::max<int64_t>(1, (int64_t)n_sets * c_prev) * sizeof(int32_t)));
  GK_HIP(hipMalloc((void**)&d_cols, (size_t)n_cols * sizeof(int32_t)));
  GK_HIP(hipMalloc((void**)&d_partial, (size_t)n_out * pl.n_chunks * sizeof(double)));
  GK_HIP(hipMalloc((void**)&d_out, (size_t)n_out * sizeof(double)));
  if (c_prev) GK_HIP(hipMemcpyAsync(d_ids, ids, (size_t)n_sets * c_prev * sizeof(int32_t), hipMemcpyHostToDevice, st));
  GK_HIP(hipMemcpyAsync(d_cols, cols, (size_t)n_cols * sizeof(int32_t), hipMemcpyHostToDevice, st));
  const int tiles_t = (n_sets + kTileT - 1) / kTileT, tiles_a = (n_cols + kTileA - 1) / kTileA;
  hipLaunchKernelGGL(maxsum_chunks, dim3((unsigned)(tiles_t * tiles_a), (unsigned)pl.n_chunks), dim3(kThreads), 0, st,
                     gk_ptr<double>(d_L), n_rows, ld, d_ids, n_sets, c_prev, d_cols, n_cols, pl.d_ops, pl.n_full,
                     pl.d_ops + pl.n_full, pl.n_tail, pl.n_chunks, d_partial);
  hipLaunchKernelGGL(combine_chunks, dim3((unsigned)((n_out + kThreads - 1) / kThreads)), dim3(kThreads), 0, st,
                     d_partial, n_out, pl.n_chunks, 0.0, d_out);
  GK_HIP(hipGetLastError());
  GK_HIP(hipMemcpyAsync(out, d_out, (size_t)n_out * sizeof(double), hipMemcpyDeviceToHost, st));
  GK_HIP(hipStreamSynchronize(st));
  hipFree(d_ids); hipFree(d_cols); hipFree(d_partial); hipFree(d_out); hipFree(pl.d_ops);
  return GK_OK;
}

int gk_fraction(gk_ctx* ctx, gk_dptr d_L, int64_t n_rows, int64_t ld, const int32_t* ids, int32_t n_sets, int32_t c,
                double* frac_out) {
  GK_REQUIRE(ctx && ids && frac_out && n_rows > 0 && ld >= n_rows && n_sets > 0, "bad fraction arguments");
  GK_REQUIRE(c >= 1 && c <= kMaxC, "copy number beyond supported set size");
  Plans pl;
  int rc = make_plans(ctx, n_rows, pl);
  if (rc) return rc;
  hipStream_t st = ctx->stream;
  int32_t* d_ids = nullptr;
  double *d_partial = nullptr, *d_out = nullptr;
  const int64_t n_out = (int64_t)n_sets * c;
  GK_HIP(hipMalloc((void**)&d_ids, (size_t)n_out * sizeof(int32_t)));
  GK_HIP(hipMalloc((void**)&d_partial, (size_t)n_out * pl.n_chunks * sizeof(double)));
  GK_HIP(hipMalloc((void**)&d_out, (size_t)n_out * sizeof(double)));
  GK_HIP(hipMemcpyAsync(d_ids, ids, (size_t)n_out * sizeof(int32_t), hipMemcpyHostToDevice, st));
  const int per_block = kThreads / 8;
  hipLaunchKernelGGL(fraction_chunks, dim3((unsigned)((n_sets + per_block - 1) / per_block), (unsigned)pl.n_chunks),
                     dim3(kThreads), 0, st, gk_ptr<double>(d_L), n_rows, ld, d_ids, n_sets, c, pl.d_ops, pl.n_full,
                     pl.d_ops + pl.n_full, pl.n_tail, pl.n_chunks, d_partial);
  hipLaunchKernelGGL(combine_chunks, dim3((unsigned)((n_out + kThreads - 1) / kThreads)), dim3(kThreads), 0, st,
                     d_partial, n_out, pl.n_chunks, (double)n_rows, d_out);
  GK_HIP(hipGetLastError());
  GK_HIP(hipMemcpyAsync(frac_out, d_out, (size_t)n_out * sizeof(double), hipMemcpyDeviceToHost, st));
  GK_HIP(hipStreamSynchronize(st));
  hipFree(d_ids); hipFree(d_partial); hipFree(d_out); hipFree(pl.d_ops);
  return GK_OK;
}

int gk_setmax(gk_ctx* ctx, gk_dptr d_L, int64_t n_rows, int64_t ld, const int32_t* ids, int32_t n_sets, int32_t c,
              gk_dptr d_P) {
  GK_REQUIRE(ctx && ids && n_rows > 0 && ld >= n_rows && n_sets > 0 && c >= 1 && c <= kMaxC, "bad setmax arguments");
  hipStream_t st = ctx->stream;
  int32_t* d_ids = nullptr;
  GK_HIP(hipMalloc((void**)&d_ids, (size_t)n_sets * c * sizeof(int32_t)));
  GK_HIP(hipMemcpyAsync(d_ids, ids, (size_t)n_sets * c * sizeof(int32_t), hipMemcpyHostToDevice, st));
  int64_t want = (n_rows + kThreads - 1) / kThreads;
  unsigned bx = (unsigned)(want < 1024 ? want : 1024);
  hipLaunchKernelGGL(setmax_kernel, dim3(bx, (unsigned)n_sets), dim3(kThreads), 0, st, gk_ptr<double>(d_L), n_rows, ld,
                     d_ids, n_sets, c, gk_ptr<double>(d_P));
  GK_HIP(hipGetLastError());
  GK_HIP(hipStreamSynchronize(st));
  hipFree(d_ids);
  return GK_OK;
}

}  // extern "C"

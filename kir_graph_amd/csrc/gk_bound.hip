// Integer bound of a likelihood search step: which candidate sets can reach the top_n cut.
//
//   gk_miss_colsum / gk_bound_step      AlleleTyping.addCandidate   typing_mulit_allele.py:534-567
//
// The reference scores every candidate set S (a previous set + one more allele) with
// value(S) = sum_r max_{a in S} log_probs[r, a] in float64, sorts ALL of them and keeps the best top_n
// (plus ties).  For one read every allele shares the number of listed variants n_r, and
// log_probs[r, a] = (n_r - m) log10(.999) - 3 m up to 1e-12, m = miss[r, a] = how many of them disagree with
// the allele ("integer reformulation"), so
//     value(S) = const - 2.99957 * M(S) + noise,   M(S) = sum_r min_{a in S} miss[r, a],  |noise| < 1e-4,
// and two sets with different M are ordered by M alone.  This file computes M for all T x A candidates on
// packed bytes and returns the sets with M <= M_T (the T-th smallest among first occurrences): only those
// can be among the reference's top_n-by-value, and only for those the exact float64 sums are formed
// afterwards (gk_setsum, with numpy's summation tree).  Everything whose order the float noise decides --
// sets with equal M -- is inside that selection, so ranks and ties stay the reference's.
//
// Mapping to CDNA4: min(a, b) = (a + b - |a - b|) / 2, so M = (sum P + sum m - SAD) / 2 and the inner loop is
// ONE v_sad_u8 (four reads) with accumulate per 4 (read, set, allele) triples -- 8x fewer instructions than
// the f64 max + add of the exact kernel, on 8x fewer bytes.  Workgroup = 64 sets x 64 alleles x one slice of
// the reads; operands are staged through LDS as 16-byte words ([column][8 words + 1 pad]: the b128 reads of
// 8 consecutive columns fall on disjoint banks), a lane owns a strided 4 x 4 block of outputs and reads
// 8 words per 64 SADs.  Integer sums are associative, so the reads are cut into as many slices as fill the
// GPU; a second small kernel adds the slices, and a two-level radix select over three more one-element-per-
// thread kernels finds M_T and compacts the selection.
#include <algorithm>

#include "gk_calls.h"

namespace {

constexpr int kThreads = 256;
constexpr int kBT = 64, kBA = 64;   // sets x alleles per workgroup
constexpr int kW = 8;               // staged 16-byte words (of 16 reads) per column and block
constexpr int kLd = kW + 1;         // padded words per staged column
constexpr int kStage = (kBT + kBA) * kW / kThreads;   // words fetched per thread and block (4)
constexpr uint32_t kNotFirst = 0xFFFFFFFFu;
constexpr int kBins = 2048;

__device__ inline uint32_t sad4(uint32_t a, uint32_t b, uint32_t acc) { return __builtin_amdgcn_sad_u8(a, b, acc); }

__device__ inline uint32_t min_u8x4(uint32_t a, uint32_t b) {
  uint32_t r = 0;
#pragma unroll
  for (int k = 0; k < 32; k += 8) r |= min((a >> k) & 255u, (b >> k) & 255u) << k;
  return r;
}

// msum[a] = sum over the reads of column a (u8 table [n_cols][ldm], rows past the end are zero)
__global__ __launch_bounds__(kThreads) void colsum_u8(const uint8_t* __restrict__ m, int64_t ldm, uint32_t* __restrict__ out) {
  const uint4* col = reinterpret_cast<const uint4*>(m + (int64_t)blockIdx.x * ldm);
  const int64_t n16 = ldm / 16;
  uint32_t acc = 0;
  for (int64_t i = threadIdx.x; i < n16; i += kThreads) {
    const uint4 w = col[i];
    acc = sad4(w.x, 0, acc); acc = sad4(w.y, 0, acc); acc = sad4(w.z, 0, acc); acc = sad4(w.w, 0, acc);
  }
  __shared__ uint32_t part[kThreads];
  part[threadIdx.x] = acc;
  __syncthreads();
  for (int s = kThreads / 2; s > 0; s >>= 1) {
    if (threadIdx.x < s) part[threadIdx.x] += part[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[blockIdx.x] = part[0];
}

// P[t][r] = min over the c columns of set t (allele_prob of a set in mismatch counts, line 569), psum[t] = its sum
__global__ __launch_bounds__(kThreads) void setmin_u8(const uint8_t* __restrict__ m, int64_t ldm,
                                                      const int32_t* __restrict__ ids, int c, uint8_t* __restrict__ P,
                                                      uint32_t* __restrict__ psum) {
  const int t = blockIdx.x;
  const int64_t n16 = ldm / 16;
  uint4* dst = reinterpret_cast<uint4*>(P + (int64_t)t * ldm);
  uint32_t acc = 0;
  for (int64_t i = threadIdx.x; i < n16; i += kThreads) {
    uint4 w = reinterpret_cast<const uint4*>(m + (int64_t)ids[t * c] * ldm)[i];
    for (int k = 1; k < c; ++k) {
      const uint4 v = reinterpret_cast<const uint4*>(m + (int64_t)ids[t * c + k] * ldm)[i];
      w.x = min_u8x4(w.x, v.x); w.y = min_u8x4(w.y, v.y); w.z = min_u8x4(w.z, v.z); w.w = min_u8x4(w.w, v.w);
    }
    dst[i] = w;
    acc = sad4(w.x, 0, acc); acc = sad4(w.y, 0, acc); acc = sad4(w.z, 0, acc); acc = sad4(w.w, 0, acc);
  }
  __shared__ uint32_t part[kThreads];
  part[threadIdx.x] = acc;
  __syncthreads();
  for (int s = kThreads / 2; s > 0; s >>= 1) {
    if (threadIdx.x < s) part[threadIdx.x] += part[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) psum[t] = part[0];
}

// partial[slice][t][a] = sum over the slice's reads of |P_t[r] - m_a[r]|
__global__ __launch_bounds__(kThreads, 2) void minsum_sad(const uint8_t* __restrict__ Pbase, int64_t ldp,
                                                          const int32_t* __restrict__ pcol, int n_sets,
                                                          const uint8_t* __restrict__ Cbase, int64_t ldc,
                                                          const int32_t* __restrict__ cols, int n_cols, int64_t n16,
                                                          int blocks_per_slice, int tiles_a,
                                                          uint32_t* __restrict__ partial, uint32_t* zero_state,
                                                          int n_zero_words) {
  __shared__ uint4 lds[(kBT + kBA) * kLd];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  // the selection state of the kernels that follow this one on the stream starts at zero (was a fill of its own)
  if (blockIdx.x == 0 && blockIdx.y == 0)
    for (int w = tid; w < n_zero_words; w += kThreads) zero_state[w] = 0;
  const int tile_t = blockIdx.x / tiles_a, tile_a = blockIdx.x % tiles_a;
  const int t0 = tile_t * kBT, c0 = tile_a * kBA;
  const int64_t blk0 = (int64_t)blockIdx.y * blocks_per_slice;
  const int64_t n_blk = (n16 + kW - 1) / kW;
  const int64_t blk1 = min<int64_t>(blk0 + blocks_per_slice, n_blk);

  // staging: thread -> word (tid & 7) of columns (tid >> 3) + 32 q; columns 0..63 are the sets, 64..127 the alleles
  const int sw = tid & (kW - 1);
  const uint4* src[kStage];
#pragma unroll
  for (int q = 0; q < kStage; ++q) {
    const int c = (tid >> 3) + 32 * q;
    int64_t base;
    if (c < kBT) {
      const int t = t0 + c;
      base = (int64_t)(t < n_sets ? (pcol ? pcol[t] : t) : 0) * ldp;
      src[q] = reinterpret_cast<const uint4*>(Pbase + base);
    } else {
      const int a = c0 + c - kBT;
      base = (int64_t)(a < n_cols ? cols[a] : cols[0]) * ldc;
      src[q] = reinterpret_cast<const uint4*>(Cbase + base);
    }
  }
  uint4 pre[kStage];
  auto fetch = [&](int64_t blk) {
    const int64_t w = blk * kW + sw;
#pragma unroll
    for (int q = 0; q < kStage; ++q) pre[q] = w < n16 ? src[q][w] : make_uint4(0, 0, 0, 0);
  };

  const int tx = lane & 7, ty = lane >> 3;
  const int sb = (wid >> 1) * 32 + ty, cb = kBT + (wid & 1) * 32 + tx;   // + 8 x, + 8 y
  uint32_t acc[4][4];
#pragma unroll
  for (int x = 0; x < 4; ++x)
#pragma unroll
    for (int y = 0; y < 4; ++y) acc[x][y] = 0;

  if (blk0 < blk1) fetch(blk0);
  for (int64_t blk = blk0; blk < blk1; ++blk) {
    __syncthreads();
#pragma unroll
    for (int q = 0; q < kStage; ++q) lds[((tid >> 3) + 32 * q) * kLd + sw] = pre[q];
    __syncthreads();
    if (blk + 1 < blk1) fetch(blk + 1);
#pragma unroll
    for (int i = 0; i < kW; ++i) {
      uint4 p[4], c[4];
#pragma unroll
      for (int x = 0; x < 4; ++x) p[x] = lds[(sb + 8 * x) * kLd + i];
#pragma unroll
      for (int y = 0; y < 4; ++y) c[y] = lds[(cb + 8 * y) * kLd + i];
#pragma unroll
      for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y) {
          uint32_t a = acc[x][y];
          a = sad4(p[x].x, c[y].x, a);
          a = sad4(p[x].y, c[y].y, a);
          a = sad4(p[x].z, c[y].z, a);
          a = sad4(p[x].w, c[y].w, a);
          acc[x][y] = a;
        }
    }
  }
  uint32_t* out = partial + (int64_t)blockIdx.y * n_sets * n_cols;
#pragma unroll
  for (int x = 0; x < 4; ++x) {
    const int t = t0 + sb + 8 * x;
#pragma unroll
    for (int y = 0; y < 4; ++y) {
      const int a = c0 + cb - kBT + 8 * y;
      if (t < n_sets && a < n_cols) out[(int64_t)t * n_cols + a] = acc[x][y];
    }
  }
}

// Selection state of one step in HBM (zeroed before the kernels run): the candidates' count and range, the two
// histogram levels of the radix select, the number of sets selected.
struct SelState {
  uint32_t count, inv_min, max, selected;   // inv_min = ~min, so that zero is the identity of atomicMax
  uint32_t cut, pad[3];
  uint32_t hist1[kBins], hist2[kBins];
};

// M[t][a] = (psum[t] + msum[col a] - sum of the slices) / 2 for first occurrences, kNotFirst otherwise
__global__ __launch_bounds__(kThreads) void minsum_finish(const uint32_t* __restrict__ partial, int n_slices,
                                                          int64_t n_out, int n_cols, const uint32_t* __restrict__ psum,
                                                          const int32_t* __restrict__ pcol,
                                                          const uint32_t* __restrict__ msum,
                                                          const int32_t* __restrict__ cols,
                                                          const uint8_t* __restrict__ first, uint32_t* __restrict__ M,
                                                          SelState* __restrict__ st) {
  const int64_t o = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  uint32_t v = kNotFirst;
  if (o < n_out && first[o]) {
    uint32_t sad = 0;
    for (int s = 0; s < n_slices; ++s) sad += partial[(int64_t)s * n_out + o];
    const int t = (int)(o / n_cols), a = (int)(o % n_cols);
    const uint32_t sp = pcol ? msum[pcol[t]] : psum[t];
    v = (sp + msum[cols[a]] - sad) >> 1;
  }
  if (o < n_out) M[o] = v;
  // count / min / max of the candidates for the select passes below (one atomic triple per workgroup)
  uint32_t cnt = v != kNotFirst ? 1u : 0u, inv_lo = v != kNotFirst ? ~v : 0u, hi = v != kNotFirst ? v : 0u;
  for (int off = 32; off > 0; off >>= 1) {
    cnt += __shfl_xor(cnt, off, 64);
    inv_lo = max(inv_lo, (uint32_t)__shfl_xor(inv_lo, off, 64));
    hi = max(hi, (uint32_t)__shfl_xor(hi, off, 64));
  }
  __shared__ uint32_t part[3][kThreads / 64];
  if ((threadIdx.x & 63) == 0) { part[0][threadIdx.x >> 6] = cnt; part[1][threadIdx.x >> 6] = inv_lo; part[2][threadIdx.x >> 6] = hi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < kThreads / 64; ++w) { cnt += part[0][w]; inv_lo = max(inv_lo, part[1][w]); hi = max(hi, part[2][w]); }
    if (cnt) {
      atomicAdd(&st->count, cnt);
      atomicMax(&st->inv_min, inv_lo);
      atomicMax(&st->max, hi);
    }
  }
}

constexpr int kSelBlocks = 64;   // workgroups of the strided select passes

// The candidates at or below the T-th smallest M (T = top_n), by a two-level radix select that every
// workgroup can finish on its own: the totals of a step lie in a narrow band, so values are reduced to
// d = (M - min) >> sh0 with sh0 chosen so that d has at most 22 bits (sh0 = 0 unless the band is wider than
// 4 M: then the cut is rounded UP to a multiple of 2^sh0 -- a superset, which the caller accepts);
// level 1 counts d >> 11, level 2 counts d & 2047 inside the level-1 bin that holds rank T.  The kernels are
// tiny (one element per thread); each reads the finished histogram of the level above and locates the cut bin
// itself with a block scan, so no single-workgroup pass over the candidates is needed.
struct SelGeom { uint32_t base, sh0; };

__device__ inline SelGeom sel_geom(const SelState* st) {
  const uint32_t base = ~st->inv_min, span = st->max - base;
  const int n_bits = span ? 32 - __builtin_clz(span) : 1;
  return SelGeom{base, (uint32_t)(n_bits > 22 ? n_bits - 22 : 0)};
}

// bin of `hist` (kBins counters) that holds 0-based rank `rank`, and the rank inside that bin; all threads of the
// workgroup (kThreads) call it and get the same answer
__device__ inline void find_bin(const uint32_t* __restrict__ hist, uint32_t rank, uint32_t* bin, uint32_t* rank_in) {
  __shared__ uint32_t run[kThreads];
  __shared__ uint32_t s_bin, s_rank;
  constexpr int kPer = kBins / kThreads;
  uint32_t local[kPer], sum = 0;
#pragma unroll
  for (int k = 0; k < kPer; ++k) { local[k] = hist[threadIdx.x * kPer + k]; sum += local[k]; }
  run[threadIdx.x] = sum;
  __syncthreads();
  for (int off = 1; off < kThreads; off <<= 1) {   // inclusive scan
    const uint32_t add = threadIdx.x >= (unsigned)off ? run[threadIdx.x - off] : 0u;
    __syncthreads();
    run[threadIdx.x] += add;
    __syncthreads();
  }
  const uint32_t before = run[threadIdx.x] - sum;
  if (threadIdx.x == 0) { s_bin = kBins - 1; s_rank = 0; }
  __syncthreads();
  if (rank >= before && rank < before + sum) {   // exactly one thread
    uint32_t acc = before;
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
      if (rank < acc + local[k]) { s_bin = threadIdx.x * kPer + k; s_rank = rank - acc; break; }
      acc += local[k];
    }
  }
  __syncthreads();
  *bin = s_bin;
  *rank_in = s_rank;
  __syncthreads();
}

// histogram passes: a few workgroups stride over M, each counts into its own LDS histogram (the totals crowd into
// few bins: global atomics on them would serialise) and adds its non-empty bins to the global one
__global__ __launch_bounds__(kThreads) void select_hist1(const uint32_t* __restrict__ M, int64_t n, SelState* __restrict__ st) {
  if (st->count == 0) return;
  __shared__ uint32_t local[kBins];
  for (int b = threadIdx.x; b < kBins; b += kThreads) local[b] = 0;
  __syncthreads();
  const SelGeom g = sel_geom(st);
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kThreads) {
    const uint32_t v = M[i];
    if (v != kNotFirst) atomicAdd(&local[((v - g.base) >> g.sh0) >> 11], 1u);
  }
  __syncthreads();
  for (int b = threadIdx.x; b < kBins; b += kThreads)
    if (local[b]) atomicAdd(&st->hist1[b], local[b]);
}

__global__ __launch_bounds__(kThreads) void select_hist2(const uint32_t* __restrict__ M, int64_t n, int top_n,
                                                         SelState* __restrict__ st) {
  if (st->count == 0) return;
  __shared__ uint32_t local[kBins];
  for (int b = threadIdx.x; b < kBins; b += kThreads) local[b] = 0;
  const SelGeom g = sel_geom(st);
  uint32_t bin1, rank1;
  find_bin(st->hist1, (uint32_t)min<int64_t>(top_n, st->count) - 1, &bin1, &rank1);   // ends with a barrier
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kThreads) {
    const uint32_t v = M[i];
    if (v != kNotFirst) {
      const uint32_t d = (v - g.base) >> g.sh0;
      if ((d >> 11) == bin1) atomicAdd(&local[d & (kBins - 1)], 1u);
    }
  }
  __syncthreads();
  for (int b = threadIdx.x; b < kBins; b += kThreads)
    if (local[b]) atomicAdd(&st->hist2[b], local[b]);
}

__global__ __launch_bounds__(kThreads) void select_append(const uint32_t* __restrict__ M, int64_t n, int top_n, int cap,
                                                          SelState* __restrict__ st, int32_t* __restrict__ idx_out,
                                                          uint32_t* __restrict__ m_out) {
  if (st->count == 0) return;
  const SelGeom g = sel_geom(st);
  uint32_t bin1, rank1, bin2, rank2;
  find_bin(st->hist1, (uint32_t)min<int64_t>(top_n, st->count) - 1, &bin1, &rank1);
  find_bin(st->hist2, rank1, &bin2, &rank2);
  // the largest M whose reduced value is (bin1, bin2): the T-th smallest itself when sh0 == 0
  const uint64_t cut64 = (uint64_t)g.base + ((((uint64_t)bin1 << 11 | bin2) + 1) << g.sh0) - 1;
  const uint32_t cut = cut64 > 0xFFFFFFFEull ? 0xFFFFFFFEu : (uint32_t)cut64;
  if (blockIdx.x == 0 && threadIdx.x == 0) st->cut = cut;
  const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  const uint32_t v = i < n ? M[i] : kNotFirst;
  const bool take = v != kNotFirst && v <= cut;
  const uint64_t takers = __ballot(take);   // one atomic per wave: its takers get consecutive slots
  if (!takers) return;
  const int lane = threadIdx.x & 63;
  const int leader = __ffsll((unsigned long long)takers) - 1;
  uint32_t first_slot = 0;
  if (lane == leader) first_slot = atomicAdd(&st->selected, (uint32_t)__popcll(takers));
  first_slot = __shfl(first_slot, leader, 64);
  if (take) {
    const uint32_t k = first_slot + (uint32_t)__popcll(takers & ((1ull << lane) - 1ull));
    if (k < (uint32_t)cap) { idx_out[k] = (int32_t)i; m_out[k] = v; }
  }
}

}  // namespace

extern "C" {

// msum_out (device, uint32 [n_cols]) = column sums of the mismatch table written by gk_compat_log_miss
int gk_miss_colsum(gk_ctx* ctx, gk_dptr d_miss8, int64_t ldm, int32_t n_cols, gk_dptr d_msum) {
  gk_bind(ctx);
  GK_REQUIRE(ctx && d_miss8 && d_msum && n_cols > 0 && ldm > 0 && ldm % 64 == 0, "bad mismatch table");
  GK_KERNEL(colsum_u8, dim3((unsigned)n_cols), dim3(kThreads), 0, ctx->stream, gk_ptr<uint8_t>(d_miss8), ldm,
            gk_ptr<uint32_t>(d_msum));
  GK_HIP(hipGetLastError());
  return GK_OK;
}

int gk_bound_step(gk_ctx* ctx, gk_dptr d_miss8, int64_t ldm, int64_t n_rows, gk_dptr d_msum, const int32_t* ids,
                  int32_t n_sets, int32_t c_prev, const int32_t* cols, int32_t n_cols, const uint8_t* first,
                  int32_t top_n, int32_t cap, uint32_t* hdr_out, int32_t* idx_out, uint32_t* m_out) {
  gk_bind(ctx);
  GK_REQUIRE(hdr_out && idx_out && m_out, "null pointer");
  GkBoundCall call;
  int rc = gk_bound_enqueue(ctx, d_miss8, ldm, n_rows, d_msum, ids, n_sets, c_prev, cols, n_cols, first, top_n, cap, call);
  if (rc == GK_OK && gk_fetch_wait(ctx) != hipSuccess) {
    gk_set_error("bound step: waiting for the stream failed");
    rc = GK_ERR_HIP;
  }
  if (rc == GK_OK) gk_bound_collect(ctx, call, hdr_out, idx_out, m_out);
  else gk_release(ctx, call.temps);
  return rc;
}

}  // extern "C"

int gk_bound_enqueue(gk_ctx* ctx, gk_dptr d_miss8, int64_t ldm, int64_t n_rows, gk_dptr d_msum, const int32_t* ids,
                     int32_t n_sets, int32_t c_prev, const int32_t* cols, int32_t n_cols, const uint8_t* first,
                     int32_t top_n, int32_t cap, GkBoundCall& call) {
  gk_bind(ctx);
  GK_REQUIRE(ctx && d_miss8 && d_msum && ids && cols && first, "null pointer");
  GK_REQUIRE(n_sets >= 1 && n_cols >= 1 && c_prev >= 1 && c_prev <= 8 && top_n >= 1 && cap >= 1, "bad bound arguments");
  GK_REQUIRE(ldm >= n_rows && ldm % 64 == 0 && n_rows > 0, "mismatch table stride must be a multiple of 64 rows");
  GK_REQUIRE(n_rows < (int64_t)16000000, "too many reads for 32-bit mismatch totals");
  hipStream_t st = ctx->stream;
  const int64_t n_out = (int64_t)n_sets * n_cols;
  auto take = [&](void** p, size_t bytes) -> hipError_t {
    hipError_t e = gk_pool_malloc(ctx, p, bytes);
    if (e == hipSuccess) call.temps.push_back(*p);
    return e;
  };
  // parameters through the context's pinned ring: [ids | cols | first mask]
  const size_t n_ids = (size_t)n_sets * c_prev;
  const size_t par_bytes = (n_ids + (size_t)n_cols) * sizeof(int32_t) + (size_t)n_out;
  char* d_par = nullptr;
  GK_HIP(take((void**)&d_par, par_bytes));
  {
    std::vector<char> packed(par_bytes);   // one copy instead of three (every runtime call is contended by the host threads)
    memcpy(packed.data(), ids, n_ids * sizeof(int32_t));
    memcpy(packed.data() + n_ids * sizeof(int32_t), cols, (size_t)n_cols * sizeof(int32_t));
    memcpy(packed.data() + (n_ids + n_cols) * sizeof(int32_t), first, (size_t)n_out);
    GK_HIP(gk_send(ctx, d_par, packed.data(), par_bytes));
    if (par_bytes > gk_stage_direct()) GK_HIP(hipStreamSynchronize(st));   // sent straight from `packed`
  }
  const int32_t* d_ids = (const int32_t*)d_par;
  const int32_t* d_cols = d_ids + n_ids;
  const uint8_t* d_first = (const uint8_t*)(d_cols + n_cols);

  const uint8_t* miss = gk_ptr<uint8_t>(d_miss8);
  uint8_t* d_P = nullptr;
  uint32_t* d_psum = nullptr;
  if (c_prev >= 2) {   // previous sets of two or more alleles become one column each
    GK_HIP(take((void**)&d_P, (size_t)n_sets * (size_t)ldm));
    GK_HIP(take((void**)&d_psum, (size_t)n_sets * sizeof(uint32_t)));
    GK_PROF(ctx, "setmin_u8", GK_KERNEL(setmin_u8, dim3((unsigned)n_sets), dim3(kThreads), 0, st, miss, ldm, d_ids,
                                        c_prev, d_P, d_psum));
  }
  const int64_t n16 = ldm / 16;
  const int64_t n_blk = (n16 + kW - 1) / kW;
  const int tiles_t = (n_sets + kBT - 1) / kBT, tiles_a = (n_cols + kBA - 1) / kBA;
  // enough slices of the reads to fill the GPU (256 CUs x 2 workgroups x 4), at least 16 blocks each
  int64_t want = (2048 + (int64_t)tiles_t * tiles_a - 1) / ((int64_t)tiles_t * tiles_a);
  want = std::max<int64_t>(1, std::min<int64_t>(want, std::max<int64_t>(1, n_blk / 16)));
  const int blocks_per_slice = (int)((n_blk + want - 1) / want);
  const int n_slices = (int)((n_blk + blocks_per_slice - 1) / blocks_per_slice);
  uint32_t *d_partial = nullptr, *d_M = nullptr;
  // selection state, indices and totals in ONE block, so that one copy brings the result back
  char* d_sel = nullptr;
  const size_t sel_bytes = sizeof(SelState) + (size_t)cap * (sizeof(int32_t) + sizeof(uint32_t));
  GK_HIP(take((void**)&d_partial, (size_t)n_out * n_slices * sizeof(uint32_t)));
  GK_HIP(take((void**)&d_M, (size_t)n_out * sizeof(uint32_t)));
  GK_HIP(take((void**)&d_sel, sel_bytes));
  SelState* d_state = (SelState*)d_sel;
  int32_t* d_idx = (int32_t*)(d_sel + sizeof(SelState));
  uint32_t* d_mout = (uint32_t*)(d_idx + cap);
  GK_PROF_EXACT(ctx, "minsum_sad",
                GK_KERNEL(minsum_sad, dim3((unsigned)(tiles_t * tiles_a), (unsigned)n_slices), dim3(kThreads), 0, st,
                          c_prev >= 2 ? d_P : miss, ldm, c_prev >= 2 ? (const int32_t*)nullptr : d_ids, n_sets, miss,
                          ldm, d_cols, n_cols, n16, blocks_per_slice, tiles_a, d_partial, (uint32_t*)d_state,
                          (int)(sizeof(SelState) / sizeof(uint32_t))));
  const dim3 per_elem((unsigned)((n_out + kThreads - 1) / kThreads));
  GK_PROF(ctx, "minsum_finish",
          GK_KERNEL(minsum_finish, per_elem, dim3(kThreads), 0, st, d_partial, n_slices, n_out, n_cols, d_psum,
                    c_prev >= 2 ? (const int32_t*)nullptr : d_ids, gk_ptr<uint32_t>(d_msum), d_cols, d_first, d_M,
                    d_state));
  const dim3 strided((unsigned)std::min<int64_t>(kSelBlocks, (n_out + kThreads - 1) / kThreads));
  GK_PROF(ctx, "select_hist1", GK_KERNEL(select_hist1, strided, dim3(kThreads), 0, st, d_M, n_out, d_state));
  GK_PROF(ctx, "select_hist2", GK_KERNEL(select_hist2, strided, dim3(kThreads), 0, st, d_M, n_out, top_n, d_state));
  GK_PROF(ctx, "select_append",
          GK_KERNEL(select_append, per_elem, dim3(kThreads), 0, st, d_M, n_out, top_n, cap, d_state, d_idx, d_mout));
  GK_HIP(hipGetLastError());
  // one copy: the header first, then the entries (the three parts are adjacent)
  call.cap = cap;
  call.state_bytes = sizeof(SelState);
  call.back.resize(sel_bytes);
  GK_HIP(gk_fetch_queue(ctx, call.back.data(), d_sel, sel_bytes));
  return GK_OK;
}

void gk_bound_collect(gk_ctx* ctx, GkBoundCall& call, uint32_t* hdr_out, int32_t* idx_out, uint32_t* m_out) {
  const uint32_t* head = (const uint32_t*)call.back.data();
  hdr_out[0] = head[0];   // candidates
  hdr_out[1] = head[4];   // the cut: the top_n-th smallest M (rounded up to 2^sh0 - 1 when the band is wider than 2^22)
  hdr_out[2] = head[3];   // selected
  hdr_out[3] = 0;
  const uint32_t n_sel = std::min<uint32_t>(hdr_out[2], (uint32_t)call.cap);
  if (n_sel) {
    memcpy(idx_out, call.back.data() + call.state_bytes, (size_t)n_sel * sizeof(int32_t));
    memcpy(m_out, call.back.data() + call.state_bytes + (size_t)call.cap * sizeof(int32_t), (size_t)n_sel * sizeof(uint32_t));
  }
  gk_release(ctx, call.temps);
}

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
// f64-VALU bound.  The 8 strided accumulators of numpy's block loop are spread over a QUAD of lanes,
// two per lane (lane k owns rows == 2k, 2k+1 mod 8: adjacent rows, one ds_read_b128 per operand), and
// every lane carries a TT x TA register tile of outputs.  The recursion stack of a sub-tree of
// <= 512 rows lives in the same four lanes (lane s holds stack slot s), so the whole tree runs
// without dynamically indexed registers; the fraction kernel uses the older 8-lane form of the same
// scheme.  L is column-major [allele][row]: row blocks are staged through LDS with coalesced 512-byte
// wave loads (next block prefetched into registers while the current one is reduced); previous sets
// of two or more alleles are first reduced to one column each (gk_setmax), so nothing of size
// T x R x A (the reference's temporary) is ever formed.  Work is split over workgroups by output tile
// AND by row span: every workgroup owns one sub-tree of <= 1024 rows of the numpy recursion, a second
// small kernel finishes the tree per chunk.
#include <algorithm>

#include "gk_calls.h"
#include "gk_lut.h"

namespace {

constexpr int kThreads = 256;
constexpr int kBlockRows = 128;   // numpy PW_BLOCKSIZE
constexpr int kChunkRows = 8192;  // numpy buffer size for add.reduce
constexpr int kSpanRows = 1024;   // rows of the recursion sub-tree owned by one workgroup
constexpr int kMaxSpans = 16;     // sub-trees per chunk (span sizes are in (512, 1024])
constexpr int kHalfRows = 512;    // a span above this is run as its two child sub-trees (lane stack depth)
constexpr int kMaxSlot = 3;       // deepest stack slot of a leaf program (4 lanes of a quad)
constexpr int kLeafHold = 1 << 8;     // Leaf::n_add flag: park slot 0 (left child of the span is complete)
constexpr int kLeafAddHold = 1 << 9;  // Leaf::n_add flag: slot 0 = parked + slot 0 (right child complete)
constexpr int TT = 4, TA = 4;
constexpr int kTileT = 8 * TT;    // 32 sets per workgroup
constexpr int kTileA = 8 * TA;    // 32 candidate columns per workgroup
constexpr int kPairs = kBlockRows / 2;   // row pairs of a staged leaf
constexpr int kCP = 33;           // LDS columns per row pair (32 + 1 pad)
constexpr int kMaxC = 8;          // alleles per set (copy number) supported
constexpr int kLPer = kTileA * kBlockRows / kThreads;   // 8 staged L values per thread
constexpr int kPPer = kTileT * kBlockRows / kThreads;   // 16 staged P values per thread

// one leaf of the recursion: rows [start, start+len) relative to the span, result pushed to stack
// slot `slot`, then `n_add` times "slot (s-1) += slot s" walking down from `slot`
struct Leaf { int32_t start, len, slot, n_add; };
struct Span { int64_t row0; int32_t leaf_begin, leaf_end; int32_t chunk, pad; };
struct TopOp { int32_t dst, src; };

__device__ inline double vmax(double a, double b) {
  double r;
  asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));   // fmax() would add two canonicalising v_max
  return r;
}

// A table of log-likelihoods is either the float64 values themselves or, in the index form, uint16 dense indices into
// the value table's array (gk_lut: `vals`): the reductions below read an entry through this view -- the same float64
// either way, so the same sums bit for bit.
template <typename TL> struct TableView;
template <> struct TableView<double> {
  const double* base;
  const double* vals;   // unused
  __device__ inline double at(int64_t i) const { return base[i]; }
};
template <> struct TableView<uint16_t> {
  const uint16_t* base;
  const double* vals;
  __device__ inline double at(int64_t i) const { return vals[base[i]]; }
};

// cross-lane move inside a 16-lane row as two v_mov_b32_dpp (no LDS round trip like ds_bpermute)
constexpr int kDppSwap1 = 0xB1;        // quad_perm:[1,0,3,2]  lane ^ 1
constexpr int kDppSwap2 = 0x4E;        // quad_perm:[2,3,0,1]  lane ^ 2
constexpr int kDppHalfMirror = 0x141;  // row_half_mirror      lane -> 7 - lane inside its group of 8
constexpr int kDppShl1 = 0x101;        // row_shl:1            lane reads lane + 1

template <int kCtrl>
__device__ inline double dpp_f64(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  // bound_ctrl: a lane whose source lies outside the row reads 0 (only row_shl's last lane, never used)
  lo = __builtin_amdgcn_update_dpp(0, lo, kCtrl, 0xF, 0xF, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, kCtrl, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}

__device__ inline double group_sum8(double v) {
  // ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) over the 8 lanes of a group; every lane gets the result.
  // After two steps lanes 0-3 hold the left half-sum and lanes 4-7 the right one, so the mirror
  // partner (7 - lane) supplies the other half; a + b == b + a bit for bit.
  v += dpp_f64<kDppSwap1>(v);
  v += dpp_f64<kDppSwap2>(v);
  v += dpp_f64<kDppHalfMirror>(v);
  return v;
}

// cross-lane moves inside a quad
constexpr int kDppQuadNext = 0xF9;     // quad_perm:[1,2,3,3]  lane reads lane + 1 of its quad

// The (max,+) contraction.  Workgroup = 32 previous sets x 32 candidate columns x one row span.
// A quad of lanes owns a 4 x 4 block of outputs; lane k of the quad carries numpy's strided
// accumulators 2k and 2k+1 (rows 8i+2k, 8i+2k+1 of the leaf), so its operands are the two
// adjacent rows of a column = one ds_read_b128, the pair sum r[2k] + r[2k+1] of numpy's
// ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) is an in-lane add and two quad DPP steps finish it.  The
// recursion stack of a sub-tree of <= 512 rows (depth <= 4) lives in the quad too: lane s holds slot s.
// LDS tiles are [row pair][column][parity] with 33 columns per row pair: the b128 reads of a
// 16-lane group then fall on 16 different bank quads (33 = 1 mod 16, column = 4 * group + y).
template <bool kPrev>
__global__ __launch_bounds__(kThreads, 2) void maxsum_chunks(const double* __restrict__ L, int64_t ld,
                                                          const double* __restrict__ Pbase, int64_t ldp,
                                                          const int32_t* __restrict__ pcol, int n_sets,
                                                          const int32_t* __restrict__ cols, int n_cols,
                                                          const Span* __restrict__ spans,
                                                          const Leaf* __restrict__ leaves, int n_tiles,
                                                          int symmetric, double* __restrict__ partial) {
  __shared__ double2 Pt[kPairs * kCP];
  __shared__ double2 Lt[kPairs * kCP];
  __shared__ double hold[(kThreads / 4) * TT * TA];   // parked left-child sums, one quad's 16 outputs each
  __shared__ int32_t p_col[kTileT];
  __shared__ int32_t l_col[kTileA];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int k = lane & 3, g = lane >> 2, gt = g >> 3, ga = g & 7;
  const int tiles_a = (n_cols + kTileA - 1) / kTileA;
  // XCD-aware order: workgroups are dealt round-robin over the 8 XCDs, so give each XCD a
  // contiguous range of (span, tile) pairs with the tile index fastest -- all tiles of a span
  // re-read the same <= 512 rows of L from that XCD's L2 instead of from MALL / HBM.
  const unsigned nb = gridDim.x, b = blockIdx.x;
  const unsigned xcd = b & 7u, kq = b >> 3, q8 = nb >> 3, r8 = nb & 7u;
  const unsigned logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + kq;
  const int span_idx = (int)(logical / (unsigned)n_tiles);
  const int tile = (int)(logical % (unsigned)n_tiles);
  const int tile_t = tile / tiles_a, tile_a = tile % tiles_a;
  const int t0 = tile_t * kTileT, c0 = tile_a * kTileA;
  // symmetric launch (set t IS column t): sum max(L_t, L_a) == sum max(L_a, L_t) term by term, so
  // tiles entirely below the diagonal are left to the host, which mirrors them
  if (symmetric && c0 + kTileA - 1 < t0) return;
  const Span span = spans[span_idx];

  if (tid < kTileT) p_col[tid] = (pcol && t0 + tid < n_sets) ? pcol[t0 + tid] : -1;
  if (tid >= 64 && tid < 64 + kTileA) l_col[tid - 64] = (c0 + tid - 64 < n_cols) ? cols[c0 + tid - 64] : -1;
  __syncthreads();

  const int srow = tid & (kBlockRows - 1);   // staged row of this thread
  const int scol = __builtin_amdgcn_readfirstlane(tid >> 7);   // first staged column (0/1), then +2 per step
  double pre_l[kLPer], pre_p[kPPer];
  // The staged columns of a wave are wave-uniform: their base addresses are scalar, so a prefetch
  // is a scalar add per column and a global_load with a 32-bit lane offset -- no per-lane 64-bit
  // address arithmetic.  Columns past the edge are clamped to a valid one and their results never
  // stored; the previous set's likelihood is ONE column (a column of L for single-allele sets, a
  // column of the row-wise-max buffer otherwise).
  const double* l_base[kLPer];
  const double* p_base[kPPer];
#pragma unroll
  for (int q = 0; q < kLPer; ++q) {
    const int cidx = __builtin_amdgcn_readfirstlane(l_col[scol + 2 * q]);
    l_base[q] = L + (int64_t)(cidx >= 0 ? cidx : 0) * ld + span.row0;
  }
#pragma unroll
  for (int q = 0; q < kPPer; ++q) {
    const int cidx = kPrev ? __builtin_amdgcn_readfirstlane(p_col[scol + 2 * q]) : 0;
    p_base[q] = Pbase + (int64_t)(cidx >= 0 ? cidx : 0) * ldp + span.row0;
  }

  auto prefetch = [&](const Leaf& lf) {
    // rows past the end of the leaf are clamped too, so every load is unconditional
    const uint32_t voff = (uint32_t)(srow < lf.len ? srow : 0) * (uint32_t)sizeof(double);
#pragma unroll
    for (int q = 0; q < kLPer; ++q)
      pre_l[q] = *(const double*)((const char*)(l_base[q] + lf.start) + voff);
#pragma unroll
    for (int q = 0; q < kPPer; ++q)
      pre_p[q] = kPrev ? *(const double*)((const char*)(p_base[q] + lf.start) + voff) : -__builtin_huge_val();
  };

  const int pc = (wid * 2 + gt) * TT;   // first of TT consecutive sets of this quad
  const int lc = ga * TA;               // first of TA consecutive columns of this quad
  double st[TT][TA];
#pragma unroll
  for (int x = 0; x < TT; ++x)
#pragma unroll
    for (int y = 0; y < TA; ++y) st[x][y] = 0.0;
  double* const my_hold = hold + (tid >> 2) * (TT * TA);   // written and read by lane 0 of the quad only
  double* const Pd = reinterpret_cast<double*>(Pt);
  double* const Ld = reinterpret_cast<double*>(Lt);
  const int sdst = ((srow >> 1) * kCP + scol) * 2 + (srow & 1);   // staging slot of column scol

  Leaf lf = leaves[span.leaf_begin];
  prefetch(lf);
  for (int li = span.leaf_begin; li < span.leaf_end; ++li) {
    // ---- publish the prefetched rows, start fetching the next leaf
    __syncthreads();
#pragma unroll
    for (int q = 0; q < kLPer; ++q) Ld[sdst + 4 * q] = pre_l[q];
#pragma unroll
    for (int q = 0; q < kPPer; ++q) Pd[sdst + 4 * q] = pre_p[q];
    __syncthreads();
    const Leaf cur = lf;
    if (li + 1 < span.leaf_end) {
      lf = leaves[li + 1];
      prefetch(lf);
    }
    const int len = cur.len;
    const int n8 = len < 8 ? 0 : len - (len & 7);
    double a[TT][TA];
    if (n8) {
      // ---- numpy pairwise block: accumulators 2k (x of the double2) and 2k+1 (y) in this lane
      double a0[TT][TA], a1[TT][TA];
      const double2* pp = Pt + k * kCP + pc;
      const double2* lp = Lt + k * kCP + lc;
      {
        double2 p[TT], l[TA];
#pragma unroll
        for (int x = 0; x < TT; ++x) p[x] = pp[x];
#pragma unroll
        for (int y = 0; y < TA; ++y) l[y] = lp[y];
#pragma unroll
        for (int x = 0; x < TT; ++x)
#pragma unroll
          for (int y = 0; y < TA; ++y) {
            a0[x][y] = vmax(p[x].x, l[y].x);
            a1[x][y] = vmax(p[x].y, l[y].y);
          }
      }
      const int n_it = n8 >> 3;
      for (int it = 1; it < n_it; ++it) {
        double2 p[TT], l[TA];
#pragma unroll
        for (int x = 0; x < TT; ++x) p[x] = pp[it * 4 * kCP + x];
#pragma unroll
        for (int y = 0; y < TA; ++y) l[y] = lp[it * 4 * kCP + y];
        __builtin_amdgcn_sched_barrier(0);   // all 8 reads in flight before the 64 max / add
        // Software pipeline, order pinned: the add of term i is issued kLag max instructions after
        // the max that feeds it, so no instruction waits on the f64 pipeline latency of its
        // predecessor (the allocator otherwise pairs every max with its add through one register).
        constexpr int kTerms = 2 * TT * TA, kLag = 2;
        double t[kLag + 1];
#pragma unroll
        for (int i = 0; i < kTerms + kLag; ++i) {
          if (i < kTerms) {
            const int x = (i >> 1) / TA, y = (i >> 1) % TA;
            t[i % (kLag + 1)] = (i & 1) ? vmax(p[x].y, l[y].y) : vmax(p[x].x, l[y].x);
          }
          if (i >= kLag) {
            const int jx = ((i - kLag) >> 1) / TA, jy = ((i - kLag) >> 1) % TA;
            if ((i - kLag) & 1) a1[jx][jy] += t[(i - kLag) % (kLag + 1)];
            else a0[jx][jy] += t[(i - kLag) % (kLag + 1)];
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
#pragma unroll
      for (int x = 0; x < TT; ++x)
#pragma unroll
        for (int y = 0; y < TA; ++y) {
          double v = a0[x][y] + a1[x][y];            // r[2k] + r[2k+1]
          v += dpp_f64<kDppSwap1>(v);                // (r0+r1)+(r2+r3) | (r4+r5)+(r6+r7)
          v += dpp_f64<kDppSwap2>(v);                // both halves; a + b == b + a bit for bit
          a[x][y] = v;
        }
    } else {
#pragma unroll
      for (int x = 0; x < TT; ++x)
#pragma unroll
        for (int y = 0; y < TA; ++y) a[x][y] = 0.0;
    }
    for (int r = n8; r < len; ++r) {   // sequential tail (the whole leaf when it is shorter than 8), same in every lane
      const int o = (r >> 1) * kCP * 2 + (r & 1);
#pragma unroll
      for (int x = 0; x < TT; ++x) {
        const double p = Pd[o + (pc + x) * 2];
#pragma unroll
        for (int y = 0; y < TA; ++y) a[x][y] += vmax(p, Ld[o + (lc + y) * 2]);
      }
    }
    // ---- push on the lane stack, then fold finished sub-trees
    {
      const bool mine = (k == cur.slot);
#pragma unroll
      for (int x = 0; x < TT; ++x)
#pragma unroll
        for (int y = 0; y < TA; ++y) st[x][y] = mine ? a[x][y] : st[x][y];
    }
    const int n_fold = cur.n_add & 0xFF;
    for (int f = 0; f < n_fold; ++f) {
      const int s = cur.slot - f;          // slot (s-1) += slot s: lane s-1 reads its right neighbour
      const bool mine = (k == s - 1);
#pragma unroll
      for (int x = 0; x < TT; ++x)
#pragma unroll
        for (int y = 0; y < TA; ++y) {
          const double other = dpp_f64<kDppQuadNext>(st[x][y]);
          st[x][y] = mine ? st[x][y] + other : st[x][y];
        }
    }
    // a span of more than 512 rows is its two children run one after the other on the same four
    // slots: the left child's sum (slot 0 = lane 0 of the quad) is parked, then added in-lane
    if ((cur.n_add & kLeafHold) && k == 0) {
#pragma unroll
      for (int x = 0; x < TT; ++x)
#pragma unroll
        for (int y = 0; y < TA; ++y) my_hold[x * TA + y] = st[x][y];
    }
    if ((cur.n_add & kLeafAddHold) && k == 0) {
#pragma unroll
      for (int x = 0; x < TT; ++x)
#pragma unroll
        for (int y = 0; y < TA; ++y) st[x][y] = my_hold[x * TA + y] + st[x][y];
    }
  }
  if (k == 0) {
#pragma unroll
    for (int x = 0; x < TT; ++x) {
      const int t = t0 + pc + x;
#pragma unroll
      for (int y = 0; y < TA; ++y) {
        const int c = c0 + lc + y;
        if (t < n_sets && c < n_cols) partial[((int64_t)span_idx * n_sets + t) * n_cols + c] = st[x][y];
      }
    }
  }
}

// finish the tree: fold the spans of each chunk with that chunk's program, then accumulate the
// chunk sums sequentially (numpy's outer reduce loop).  A workgroup owns 64 outputs; its four waves
// fold four chunks at a time (the span sums of a chunk are read with all loads in flight), wave 0
// then adds the four chunk sums in chunk order.
constexpr int kCombOut = 64;
constexpr int kCombLanes = kThreads / kCombOut;

__global__ __launch_bounds__(kThreads) void combine_chunks(const double* __restrict__ partial, int64_t n_out,
                                                           int n_chunks, const int32_t* __restrict__ chunk_span0,
                                                           const int32_t* __restrict__ chunk_op0,
                                                           const TopOp* __restrict__ top, double scale_div,
                                                           double* __restrict__ out) {
  // the span sums of a chunk sit in LDS ([span][thread]: conflict-free, and the program's dst / src
  // indices need no private-memory array)
  __shared__ double sp[kMaxSpans][kThreads];
  __shared__ double csum[kCombLanes][kCombOut];
  const int t = threadIdx.x, j = t & (kCombOut - 1);
  const int c = __builtin_amdgcn_readfirstlane(t / kCombOut);
  const int64_t o = (int64_t)blockIdx.x * kCombOut + j;
  const bool live = o < n_out;
  double total = 0.0;
  for (int q0 = 0; q0 < n_chunks; q0 += kCombLanes) {
    const int q = q0 + c;   // wave-uniform
    if (q < n_chunks) {
      const int s0 = chunk_span0[q], n_s = chunk_span0[q + 1] - s0;
      double v[kMaxSpans];
#pragma unroll
      for (int s = 0; s < kMaxSpans; ++s) v[s] = (s < n_s && live) ? partial[(int64_t)(s0 + s) * n_out + o] : 0.0;
#pragma unroll
      for (int s = 0; s < kMaxSpans; ++s)
        if (s < n_s) sp[s][t] = v[s];
      for (int k = chunk_op0[q]; k < chunk_op0[q + 1]; ++k) sp[top[k].dst][t] += sp[top[k].src][t];
      csum[c][j] = sp[0][t];
    }
    __syncthreads();
    if (c == 0) {
      const int n_here = min(kCombLanes, n_chunks - q0);
      for (int i = 0; i < n_here; ++i) total = (q0 + i == 0) ? csum[i][j] : total + csum[i][j];
    }
    __syncthreads();
  }
  if (c == 0 && live) out[o] = scale_div != 0.0 ? total / scale_div : total;
}

// abundance share (typing_mulit_allele.py:575-580): one 8-lane group per allele set, 32 sets per
// workgroup.  The host orders the sets so that neighbours share alleles and hands every tile of 32
// sets the list of its distinct columns; the workgroup stages those columns through LDS in blocks
// of 32 rows (coalesced 256-byte runs) instead of every set reading its own columns from L2.
constexpr int kFracSets = kThreads / 8;   // sets per workgroup
constexpr int kFracRows = 32;             // staged rows per block (4 row-steps of the 8 lanes)
constexpr int kFracLd = kFracRows + 9;    // + up to 7 tail rows of the leaf, odd stride

// kValue: the set's likelihood sum_r max_j L[r, ids[k][j]] (typing_mulit_allele.py:540-542 for ONE set) rides along as
// one more accumulator -- same terms, same tree, hence the bits of maxsum_chunks -- and is stored after the shares.
template <int kC, bool kValue, typename TL>
__global__ __launch_bounds__(kThreads) void fraction_chunks(TableView<TL> L, int64_t ld,
                                                            const int32_t* __restrict__ tile_col_off,
                                                            const int32_t* __restrict__ tile_cols,
                                                            const int32_t* __restrict__ local_idx,
                                                            const int32_t* __restrict__ perm, int n_sets,
                                                            const Span* __restrict__ spans,
                                                            const Leaf* __restrict__ leaves,
                                                            double* __restrict__ partial) {
  extern __shared__ double fbuf[];   // [distinct columns of the tile][kFracLd]
  const int tid = threadIdx.x, lane = tid & 63;
  const int j = lane & 7;
  const int k = blockIdx.x * kFracSets + (tid >> 3);   // position in the host's set order
  const Span span = spans[blockIdx.y];
  const bool live = k < n_sets;
  const int c0 = tile_col_off[blockIdx.x], n_dist = tile_col_off[blockIdx.x + 1] - c0;
  const int32_t* cols = tile_cols + c0;
  int loc[kC];
#pragma unroll
  for (int q = 0; q < kC; ++q) loc[q] = live ? local_idx[k * kC + q] * kFracLd : 0;
  constexpr int kO = kC + (kValue ? 1 : 0);   // accumulators per set: the shares, then the value
  double st[kO], hold[kO];
#pragma unroll
  for (int q = 0; q < kO; ++q) st[q] = hold[q] = 0.0;

  // acc += the row's shares; staged row index r (0..kFracLd)
  auto add_terms = [&](int r, double* acc) {
    double v[kC];
    double best = -__builtin_huge_val();
#pragma unroll
    for (int q = 0; q < kC; ++q) {
      v[q] = fbuf[loc[q] + r];
      best = vmax(best, v[q]);
    }
    int cnt = 0;
#pragma unroll
    for (int q = 0; q < kC; ++q) cnt += v[q] == best ? 1 : 0;
    double share = 1.0;   // 1 / (number of alleles sharing the maximum): exact IEEE quotients, as numpy's bool / int
#pragma unroll
    for (int q = 2; q <= kC; ++q) share = cnt == q ? (1.0 / (double)q) : share;
#pragma unroll
    for (int q = 0; q < kC; ++q) acc[q] += v[q] == best ? share : 0.0;   // 0.0 + x == x: first term exact
    if (kValue) acc[kC] += best;
  };

  for (int li = span.leaf_begin; li < span.leaf_end; ++li) {
    const Leaf cur = leaves[li];
    const int64_t r0 = span.row0 + cur.start;
    const int len = cur.len;
    const int n8 = len < 8 ? 0 : len - (len & 7);   // rows summed by the 8 strided accumulators
    double acc[kO];
#pragma unroll
    for (int q = 0; q < kO; ++q) acc[q] = 0.0;
    const int n_blocks = n8 ? (n8 + kFracRows - 1) / kFracRows : 1;
    for (int sb = 0; sb < n_blocks; ++sb) {
      const int b0 = sb * kFracRows;
      const int rows_in = min(kFracRows, n8 - b0);          // <= 0 when the leaf is all tail
      const bool last = sb + 1 == n_blocks;
      __syncthreads();
      for (int idx = tid; idx < n_dist * kFracRows; idx += kThreads) {
        const int col = idx >> 5, r = idx & (kFracRows - 1);
        if (r < rows_in) fbuf[col * kFracLd + r] = L.at((int64_t)cols[col] * ld + r0 + b0 + r);
      }
      if (last) {   // sequential tail rows n8 .. len (all rows of a leaf shorter than 8)
        for (int idx = tid; idx < n_dist * 8; idx += kThreads) {
          const int col = idx >> 3, r = n8 + (idx & 7);
          if (r < len) fbuf[col * kFracLd + kFracRows + (idx & 7)] = L.at((int64_t)cols[col] * ld + r0 + r);
        }
      }
      __syncthreads();
      for (int r = j; r < rows_in; r += 8) add_terms(r, acc);
      if (last) {
        if (n8) {
#pragma unroll
          for (int q = 0; q < kO; ++q) acc[q] = group_sum8(acc[q]);
        }
        for (int r = n8; r < len; ++r) add_terms(kFracRows + r - n8, acc);   // same in every lane
      }
    }
    {
      const bool mine = (j == cur.slot);
#pragma unroll
      for (int q = 0; q < kO; ++q) st[q] = mine ? acc[q] : st[q];
    }
    const int n_fold = cur.n_add & 0xFF;
    for (int a = 0; a < n_fold; ++a) {
      const int s = cur.slot - a;
      const bool mine = (j == s - 1);
#pragma unroll
      for (int q = 0; q < kO; ++q) {
        const double other = dpp_f64<kDppShl1>(st[q]);
        st[q] = mine ? st[q] + other : st[q];
      }
    }
    if (cur.n_add & kLeafHold) {      // left child of the span complete: park its sum (lane 0)
#pragma unroll
      for (int q = 0; q < kO; ++q) hold[q] = st[q];
    }
    if (cur.n_add & kLeafAddHold) {   // right child complete
#pragma unroll
      for (int q = 0; q < kO; ++q) st[q] = hold[q] + st[q];
    }
  }
  if (j == 0 && live) {
    const int64_t ko = perm[k];
#pragma unroll
    for (int q = 0; q < kO; ++q) partial[((int64_t)blockIdx.y * n_sets + ko) * kO + q] = st[q];
  }
}

// ---- the same sums LEAF BY LEAF (typing_mulit_allele.py:540-542 for one set, 575-580): every column through LDS once.
// fraction_chunks gives a workgroup 32 sets and stages THEIR columns; the ~600 sets a step selects touch every allele of
// the gene, so the table was read 2.85 times through L2 (6.2 GB per configs[1] sample).  Here a workgroup of 1024 threads
// owns ONE leaf of numpy's tree (<= 128 consecutive reads) and ALL sets: the leaf's rows of every column are staged
// through LDS in blocks of 32 rows (the next block's loads are in flight while the current one is summed), each of the
// 128 eight-lane groups carries the accumulators of ~5 sets (lane j = numpy's strided accumulator j), and the leaf's
// sums go to `partial[leaf][set][share..., value]`.  fold_leaves then adds the leaves of a chunk in the order of numpy's
// pairwise recursion and combine_chunks the chunks in sequence -- the same tree as fraction_chunks walks on its lane
// stack, so the same bits.  HBM traffic = the table once.
struct LeafAbs { int64_t row0; int32_t len, chunk; };
constexpr int kLeafThreads = 1024;
constexpr int kLeafGroups = kLeafThreads / 8;        // 128 lane groups = sets in flight per "slot"
constexpr int kLeafSlots = 6;                        // sets per lane group at most: up to 768 sets per launch
constexpr int kStageRows = 32;
#ifndef GK_STAGE_LD
#define GK_STAGE_LD (kStageRows + 8)
#endif
// 40 doubles: a column's 16-bank window (8 lanes x 8 bytes) starts at a multiple of 16 banks, so the two lane groups of an
// LDS pass collide only when their columns' distance is a multiple of 4 (with the odd stride 33 the windows overlapped for
// 15 of 32 distances: 46 % of the LDS cycles were conflicts) -- 2 % of the kernel, profiles/r04_compat_experiments.txt
constexpr int kStageLd = GK_STAGE_LD;
constexpr int kStagePrefetch = 8;                    // staged values a thread carries in registers: <= 256 columns
constexpr int kFoldOut = 64;                         // outputs per workgroup of fold_leaves
constexpr int kMaxChunkLeaves = kChunkRows / 64;     // a leaf of a split node has >= 64 rows

template <int kC, int kS>
__global__ __launch_bounds__(kLeafThreads) void setsum_leaves(const double* __restrict__ L, int64_t ld, int n_cols,
                                                              const int32_t* __restrict__ ids, int n_sets,
                                                              int n_sets_all, const LeafAbs* __restrict__ leaves,
                                                              double* __restrict__ partial) {
  // ids / n_sets: this launch's batch of sets (partial already points at its first set); n_sets_all: sets per leaf row
  extern __shared__ double sbuf[];   // [n_cols][kStageLd]
  constexpr int kO = kC + 1;
  const int tid = threadIdx.x, j = tid & 7, g = tid >> 3;
  int loc[kS][kC];
  double acc[kS][kO];
  // two alleles: a row's shares are 1 / 0, 0 / 1 or 1/2 / 1/2 -- multiples of 1/2, whose sums are exact in float64 in
  // any order -- so the lanes COUNT the rows an allele reaches the maximum on (one compare + add-with-carry each) and the
  // leaf's shares are formed from the counts at the end: n_q - tied / 2, tied = n_0 + n_1 - rows.  The value keeps its tree.
  uint32_t hits[kS][2];
#pragma unroll
  for (int s = 0; s < kS; ++s) {
    const int k = g + kLeafGroups * s;
#pragma unroll
    for (int q = 0; q < kC; ++q) loc[s][q] = (k < n_sets ? ids[k * kC + q] : 0) * kStageLd + j;   // the lane's first row
#pragma unroll
    for (int q = 0; q < kO; ++q) acc[s][q] = 0.0;
    hits[s][0] = hits[s][1] = 0u;
  }
  const LeafAbs leaf = leaves[blockIdx.x];
  const int len = leaf.len;
  const int n8 = len < 8 ? 0 : len - (len & 7);      // rows summed by the 8 strided accumulators
  const double* const base = L + leaf.row0;
  const int n_elem = n_cols * kStageRows;
  // the next block rides in registers (and a column's lane offset fits 32 bits: 31 columns of ld rows)
  const bool carried = n_elem <= kStagePrefetch * kLeafThreads && (uint64_t)ld * 32u * sizeof(double) < (1ull << 32);

  // set s at staged row j + r (r a constant in the unrolled walk of a block: the LDS reads then carry it as their
  // immediate offset and the loop has no address arithmetic at all): shares of the row's maximum (1 / number of alleles
  // that reach it) + the maximum itself
  auto add_terms = [&](int r) {
#pragma unroll
    for (int s = 0; s < kS; ++s) {
      if (kC == 2) {
        const double v0 = sbuf[loc[s][0] + r], v1 = sbuf[loc[s][1] + r];
        acc[s][kC] += vmax(v0, v1);
        hits[s][0] += v0 >= v1 ? 1u : 0u;
        hits[s][1] += v1 >= v0 ? 1u : 0u;
        continue;
      }
      double v[kC];
      double best = -__builtin_huge_val();
#pragma unroll
      for (int q = 0; q < kC; ++q) {
        v[q] = sbuf[loc[s][q] + r];
        best = vmax(best, v[q]);
      }
      int cnt = 0;
#pragma unroll
      for (int q = 0; q < kC; ++q) cnt += v[q] == best ? 1 : 0;
      double share = 1.0;   // exact IEEE quotients, as numpy's bool / int
#pragma unroll
      for (int q = 2; q <= kC; ++q) share = cnt == q ? (1.0 / (double)q) : share;
#pragma unroll
      for (int q = 0; q < kC; ++q) acc[s][q] += v[q] == best ? share : 0.0;   // 0.0 + x == x: first term exact
      acc[s][kC] += best;
    }
  };

  double pre[kStagePrefetch];
  // thread tid stages row (tid & 31) of columns (tid >> 5) + 32 u: one 32-bit lane offset, the rest of the address is
  // wave-uniform (a scalar base per u and block)
  const uint32_t lane_off = ((uint32_t)(tid >> 5) * (uint32_t)ld + (uint32_t)(tid & (kStageRows - 1))) * (uint32_t)sizeof(double);
  const size_t col_step = (size_t)(kLeafThreads / kStageRows) * (size_t)ld * sizeof(double);
  auto fetch = [&](int b0, int rows_in) {
    const char* const block = reinterpret_cast<const char*>(base + b0);
#pragma unroll
    for (int u = 0; u < kStagePrefetch; ++u) {
      const int idx = tid + u * kLeafThreads;
      const int r = idx & (kStageRows - 1);
      pre[u] = (idx < n_elem && r < rows_in) ? *reinterpret_cast<const double*>(block + (size_t)u * col_step + lane_off) : 0.0;
    }
  };
  const int n_stage = (n8 + kStageRows - 1) / kStageRows;
  if (carried && n_stage) fetch(0, min(kStageRows, n8));
  for (int sb = 0; sb < n_stage; ++sb) {
    const int b0 = sb * kStageRows;
    const int rows_in = min(kStageRows, n8 - b0);
    __syncthreads();                       // the readers of the block before this one are done
    if (carried) {
#pragma unroll
      for (int u = 0; u < kStagePrefetch; ++u) {
        const int idx = tid + u * kLeafThreads;
        if (idx < n_elem) sbuf[(idx >> 5) * kStageLd + (idx & (kStageRows - 1))] = pre[u];
      }
    } else {
      for (int idx = tid; idx < n_elem; idx += kLeafThreads) {
        const int col = idx >> 5, r = idx & (kStageRows - 1);
        if (r < rows_in) sbuf[col * kStageLd + r] = base[(int64_t)col * ld + b0 + r];
      }
    }
    __syncthreads();
    if (carried && sb + 1 < n_stage) fetch(b0 + kStageRows, min(kStageRows, n8 - b0 - kStageRows));
    if (kS * kC <= 10 && rows_in == kStageRows) {       // (larger forms do not fit the registers unrolled)
#pragma unroll
      for (int k = 0; k < kStageRows / 8; ++k) {
        add_terms(8 * k);
        // the LDS reads of two rows in flight at a time where the registers allow it, of one row otherwise
        if (kS * kC >= 10 || (k & 1)) __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll 1
      for (int k = 0; k < rows_in / 8; ++k) add_terms(8 * k);      // rows_in is a multiple of 8 (n8 and the blocks are)
    }
  }
  if (n8) {
#pragma unroll
    for (int s = 0; s < kS; ++s) {
#pragma unroll
      for (int q = (kC == 2 ? kC : 0); q < kO; ++q) acc[s][q] = group_sum8(acc[s][q]);
      if (kC == 2) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {      // the counts of the group's 8 lanes (integers: any order), in every lane
          uint32_t h = hits[s][q];
          h += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)h, kDppSwap1, 0xF, 0xF, true);
          h += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)h, kDppSwap2, 0xF, 0xF, true);
          h += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)h, kDppHalfMirror, 0xF, 0xF, true);
          hits[s][q] = h;
        }
      }
    }
  }
  if (len > n8) {                          // sequential tail rows n8 .. len (all rows of a leaf shorter than 8)
    __syncthreads();
    for (int idx = tid; idx < n_cols * 8; idx += kLeafThreads) {
      const int col = idx >> 3, t = idx & 7;
      if (n8 + t < len) sbuf[col * kStageLd + t] = base[(int64_t)col * ld + n8 + t];
    }
    __syncthreads();
    for (int r = 0; r < len - n8; ++r) add_terms(r - j);   // row r, the same in every lane
  }
  if (kC == 2) {
#pragma unroll
    for (int s = 0; s < kS; ++s) {
      const double tied = (double)(hits[s][0] + hits[s][1] - (uint32_t)len);
      acc[s][0] = (double)hits[s][0] - 0.5 * tied;      // exact: integers and halves
      acc[s][1] = (double)hits[s][1] - 0.5 * tied;
    }
  }
  if (j == 0) {
#pragma unroll
    for (int s = 0; s < kS; ++s) {
      const int k = g + kLeafGroups * s;
      if (k < n_sets) {
        double* const dst = partial + ((int64_t)blockIdx.x * n_sets_all + k) * kO;
#pragma unroll
        for (int q = 0; q < kO; ++q) dst[q] = acc[s][q];
      }
    }
  }
}

// the leaves of chunk blockIdx.y, added up in the order of numpy's pairwise recursion (lops: dst += src over the
// chunk's leaves, post-order); a thread owns one output, its leaf sums sit in its own LDS column (all loads in flight).
// The workgroup that finishes LAST for its 64 outputs (a ticket per output group, put back to zero by the taker) adds the
// chunk sums in chunk order -- numpy's outer reduce loop -- into `out` (a launch of its own, combine_chunks, until round 4).
__global__ __launch_bounds__(kFoldOut) void fold_leaves(const double* __restrict__ partial, int64_t n_out,
                                                        const int32_t* __restrict__ chunk_leaf0,
                                                        const int32_t* __restrict__ chunk_lop0,
                                                        const TopOp* __restrict__ lops, double* chunk_sums, int n_chunks,
                                                        uint32_t* __restrict__ tickets, double* __restrict__ out) {
  extern __shared__ double lsum[];   // [leaf of the chunk][kFoldOut]
  const int t = threadIdx.x, q = blockIdx.y;
  const int64_t o = (int64_t)blockIdx.x * kFoldOut + t;
  const int l0 = chunk_leaf0[q], n_l = chunk_leaf0[q + 1] - l0;
  if (o < n_out) {
#pragma unroll 8
    for (int l = 0; l < n_l; ++l) lsum[l * kFoldOut + t] = partial[(int64_t)(l0 + l) * n_out + o];
  }
  for (int k = chunk_lop0[q]; k < chunk_lop0[q + 1]; ++k) lsum[lops[k].dst * kFoldOut + t] += lsum[lops[k].src * kFoldOut + t];
  if (o < n_out) chunk_sums[(int64_t)q * n_out + o] = lsum[t];
  __shared__ uint32_t s_last;
  __threadfence();                       // this chunk's sums are visible before the ticket is taken
  __syncthreads();
  if (t == 0) {
    const uint32_t ticket = atomicAdd(&tickets[blockIdx.x], 1u);
    s_last = ticket == (uint32_t)n_chunks - 1 ? 1u : 0u;
    if (s_last) tickets[blockIdx.x] = 0u;                     // for the next launch on this stream
  }
  __syncthreads();
  if (!s_last || o >= n_out) return;
  __threadfence();
  double total = 0.0;
  for (int c = 0; c < n_chunks; ++c) {
    const double v = __hip_atomic_load(&chunk_sums[(int64_t)c * n_out + o], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    total = c == 0 ? v : total + v;
  }
  out[o] = total;
}

// Column sums log_probs[:, cols].sum(axis=0) (typing_mulit_allele.py:514) with numpy's tree.  An 8-lane group
// owns one column over one span (lane j = numpy's strided accumulator j, stack slot j), so a workgroup of 256 threads
// sums 32 columns.  No LDS staging: every element is read exactly once, straight from HBM -- the 8 lanes of a group
// read 64 contiguous bytes per row-step and all row-steps of a leaf are in flight together.  (gk_maxsum used to run this
// case through the (max,+) kernel with one live set per 32 x 32 tile at two workgroups per CU.)
constexpr int kColsPerGroup = kThreads / 8;

template <typename TL>
__global__ __launch_bounds__(kThreads) void colsum_chunks(TableView<TL> L, int64_t ld,
                                                          const int32_t* __restrict__ cols, int n_cols,
                                                          const Span* __restrict__ spans,
                                                          const Leaf* __restrict__ leaves, double* __restrict__ partial) {
  const int tid = threadIdx.x, j = tid & 7;
  const int c = blockIdx.x * kColsPerGroup + (tid >> 3);
  const bool live = c < n_cols;
  const Span span = spans[blockIdx.y];
  const int64_t col = (int64_t)cols[live ? c : 0] * ld + span.row0;
  double st = 0.0, hold = 0.0;
  for (int li = span.leaf_begin; li < span.leaf_end; ++li) {
    const Leaf cur = leaves[li];
    const int64_t a = col + cur.start;
    const int len = cur.len;
    const int n8 = len < 8 ? 0 : len - (len & 7);
    double acc = 0.0;
    if (n8) {
      double v[kBlockRows / 8];
#pragma unroll
      for (int q = 0; q < kBlockRows / 8; ++q) v[q] = 8 * q < n8 ? L.at(a + 8 * q + j) : 0.0;   // all loads of the leaf in flight
      acc = v[0];
#pragma unroll
      for (int q = 1; q < kBlockRows / 8; ++q)
        if (8 * q < n8) acc += v[q];                     // r[j] += a[i + j], in row order
      acc = group_sum8(acc);                             // ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7))
    }
    for (int r = n8; r < len; ++r) acc += L.at(a + r);     // sequential tail (the whole leaf when shorter than 8)
    st = (j == cur.slot) ? acc : st;
    const int n_fold = cur.n_add & 0xFF;
    for (int f = 0; f < n_fold; ++f) {
      const int s = cur.slot - f;
      const double other = dpp_f64<kDppShl1>(st);
      st = (j == s - 1) ? st + other : st;
    }
    if (cur.n_add & kLeafHold) hold = st;
    if (cur.n_add & kLeafAddHold) st = hold + st;
  }
  if (j == 0 && live) partial[(int64_t)blockIdx.y * n_cols + c] = st;
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

// index form -> float64 form of a table (for the exact (max,+) kernel, which streams float64 operands)
__global__ __launch_bounds__(kThreads) void expand_index(const uint16_t* __restrict__ lidx, int64_t ldi, int64_t n_rows,
                                                         const double* __restrict__ vals, double* __restrict__ L, int64_t ld) {
  const int a = blockIdx.y;
  for (int64_t r = (int64_t)blockIdx.x * kThreads + threadIdx.x; r < n_rows; r += (int64_t)gridDim.x * kThreads)
    L[(int64_t)a * ld + r] = vals[lidx[(int64_t)a * ldi + r]];
}

// ---------------------------------------------------------------------------------------------
// host: numpy pairwise_sum recursion as span / leaf / top programs
struct Program {
  std::vector<Leaf> leaves;
  std::vector<Span> spans;
  std::vector<int32_t> chunk_span0, chunk_op0;
  std::vector<TopOp> top;
  // the same tree leaf by leaf (setsum_leaves / fold_leaves): every leaf with its first row, the leaves of a chunk,
  // and per chunk the additions "leaf dst += leaf src" (chunk-relative) in post-order
  std::vector<LeafAbs> flat;
  std::vector<int32_t> chunk_leaf0, chunk_lop0;
  std::vector<TopOp> lops;
  std::vector<int32_t> unit0, zero0;      // combine_chunks over one sum per chunk: [0, 1, 2, ...] and all zeros
};

// returns the (chunk-relative) leaf that holds the node's sum
int fold_nodes(Program& p, int chunk, int64_t chunk_row0, int start, int n, int first_leaf) {
  if (n <= kBlockRows) {
    p.flat.push_back(LeafAbs{chunk_row0 + start, n, chunk});
    return (int)p.flat.size() - 1 - first_leaf;
  }
  int n2 = n / 2;
  n2 -= n2 % 8;
  const int a = fold_nodes(p, chunk, chunk_row0, start, n2, first_leaf);
  const int b = fold_nodes(p, chunk, chunk_row0, start + n2, n - n2, first_leaf);
  p.lops.push_back(TopOp{a, b});
  return a;
}

// leaves of a span in post-order; a leaf that completes right sub-trees folds them immediately
void span_leaves(int start, int n, int slot, std::vector<Leaf>& out) {
  if (n <= kBlockRows) {
    out.push_back(Leaf{start, n, slot, 0});
    return;
  }
  int n2 = n / 2;
  n2 -= n2 % 8;
  span_leaves(start, n2, slot, out);
  span_leaves(start + n2, n - n2, slot + 1, out);
  out.back().n_add += 1;   // after the right child finished: slot += slot + 1
}

// returns the (chunk-relative) span index that holds the node's sum
int chunk_nodes(Program& p, int chunk, int64_t chunk_row0, int start, int n, int first_span) {
  if (n <= kSpanRows) {
    Span s;
    s.row0 = chunk_row0 + start;
    s.leaf_begin = (int32_t)p.leaves.size();
    if (n <= kHalfRows) {
      span_leaves(0, n, 0, p.leaves);
    } else {   // two children on the same slots: park the left sum, add it to the right one
      int n2 = n / 2;
      n2 -= n2 % 8;
      span_leaves(0, n2, 0, p.leaves);
      p.leaves.back().n_add |= kLeafHold;
      span_leaves(n2, n - n2, 0, p.leaves);
      p.leaves.back().n_add |= kLeafAddHold;
    }
    s.leaf_end = (int32_t)p.leaves.size();
    s.chunk = chunk;
    s.pad = 0;
    p.spans.push_back(s);
    return (int)p.spans.size() - 1 - first_span;
  }
  int n2 = n / 2;
  n2 -= n2 % 8;
  const int a = chunk_nodes(p, chunk, chunk_row0, start, n2, first_span);
  const int b = chunk_nodes(p, chunk, chunk_row0, start + n2, n - n2, first_span);
  p.top.push_back(TopOp{a, b});
  return a;
}

void build_program(int64_t n_rows, Program& p) {
  const int n_chunks = (int)((n_rows + kChunkRows - 1) / kChunkRows);
  for (int q = 0; q < n_chunks; ++q) {
    const int64_t row0 = (int64_t)q * kChunkRows;
    const int n = (int)std::min<int64_t>(kChunkRows, n_rows - row0);
    p.chunk_span0.push_back((int32_t)p.spans.size());
    p.chunk_op0.push_back((int32_t)p.top.size());
    chunk_nodes(p, q, row0, 0, n, (int)p.spans.size());
    p.chunk_leaf0.push_back((int32_t)p.flat.size());
    p.chunk_lop0.push_back((int32_t)p.lops.size());
    fold_nodes(p, q, row0, 0, n, (int)p.flat.size());
    p.unit0.push_back(q);
    p.zero0.push_back(0);
  }
  p.chunk_span0.push_back((int32_t)p.spans.size());
  p.chunk_op0.push_back((int32_t)p.top.size());
  p.chunk_leaf0.push_back((int32_t)p.flat.size());
  p.chunk_lop0.push_back((int32_t)p.lops.size());
  p.unit0.push_back(n_chunks);
  p.zero0.push_back(0);
}

struct DeviceProgram {
  char* base = nullptr;
  Leaf* leaves = nullptr;
  Span* spans = nullptr;
  int32_t *chunk_span0 = nullptr, *chunk_op0 = nullptr;
  TopOp* top = nullptr;
  int32_t *ids = nullptr, *cols = nullptr;
  int n_spans = 0, n_chunks = 0;
  LeafAbs* flat = nullptr;
  int32_t *chunk_leaf0 = nullptr, *chunk_lop0 = nullptr, *unit0 = nullptr, *zero0 = nullptr;
  TopOp* lops = nullptr;
  int n_leaves = 0, max_chunk_leaves = 0;
};

template <typename T>
size_t put(std::vector<char>& buf, const T* src, size_t n) {
  const size_t off = (buf.size() + 15) / 16 * 16;
  buf.resize(off + std::max<size_t>(n, 1) * sizeof(T));
  if (n) memcpy(buf.data() + off, src, n * sizeof(T));
  return off;
}

// The span / leaf / top program depends on the row count only: it is built and uploaded once per
// (context, row count) and kept (a search makes dozens of calls on the same table).  The per-call
// id / column arrays go through the context's pinned staging buffer, so a call costs one truly
// asynchronous copy and no synchronisation before its kernels.
using CachedProgram = gk_ctx::TreeHead;
constexpr size_t kMaxCachedPrograms = 64;

int upload_program(gk_ctx* ctx, int64_t n_rows, const int32_t* ids, size_t n_ids, const int32_t* cols, size_t n_cols,
                   DeviceProgram& d) {
  char* prog = nullptr;
  auto hit = ctx->tree_programs.find(n_rows);
  if (hit != ctx->tree_programs.end()) {
    prog = (char*)hit->second;
  } else {
    Program p;
    build_program(n_rows, p);
    for (size_t q = 0; q + 1 < p.chunk_span0.size(); ++q)
      GK_REQUIRE(p.chunk_span0[q + 1] - p.chunk_span0[q] <= kMaxSpans, "too many spans in a chunk");
    for (const Leaf& lf : p.leaves) GK_REQUIRE(lf.slot <= kMaxSlot, "leaf program deeper than the lane stack");
    std::vector<char> buf;
    CachedProgram head;
    head.o_leaf = put(buf, p.leaves.data(), p.leaves.size());
    head.o_span = put(buf, p.spans.data(), p.spans.size());
    head.o_cs = put(buf, p.chunk_span0.data(), p.chunk_span0.size());
    head.o_co = put(buf, p.chunk_op0.data(), p.chunk_op0.size());
    head.o_top = put(buf, p.top.data(), p.top.size());
    head.o_flat = put(buf, p.flat.data(), p.flat.size());
    head.o_cl = put(buf, p.chunk_leaf0.data(), p.chunk_leaf0.size());
    head.o_clo = put(buf, p.chunk_lop0.data(), p.chunk_lop0.size());
    head.o_lops = put(buf, p.lops.data(), p.lops.size());
    head.o_unit = put(buf, p.unit0.data(), p.unit0.size());
    head.o_zero = put(buf, p.zero0.data(), p.zero0.size());
    head.n_leaves = (int)p.flat.size();
    head.max_chunk_leaves = 1;
    for (size_t q = 0; q + 1 < p.chunk_leaf0.size(); ++q)
      head.max_chunk_leaves = std::max(head.max_chunk_leaves, p.chunk_leaf0[q + 1] - p.chunk_leaf0[q]);
    GK_REQUIRE(head.max_chunk_leaves <= kMaxChunkLeaves, "too many leaves in a chunk");
    head.n_spans = (int)p.spans.size();
    head.n_chunks = (int)p.chunk_span0.size() - 1;
    if (ctx->tree_programs.size() >= kMaxCachedPrograms) {   // every earlier call has synchronised: nothing is in flight
      for (auto& kv : ctx->tree_programs) gk_pool_free(ctx, kv.second);
      ctx->tree_programs.clear();
      ctx->tree_heads.clear();
    }
    GK_HIP(gk_pool_malloc(ctx, (void**)&prog, buf.size()));
    GK_HIP(gk_send(ctx, prog, buf.data(), buf.size()));
    if (buf.size() > gk_stage_direct())          // a large program goes straight from buf, which goes out of scope
      GK_HIP(hipStreamSynchronize(ctx->stream));
    ctx->tree_programs[n_rows] = prog;
    ctx->tree_heads[n_rows] = head;
  }
  const CachedProgram& head = ctx->tree_heads[n_rows];
  d.leaves = (Leaf*)(prog + head.o_leaf);
  d.spans = (Span*)(prog + head.o_span);
  d.chunk_span0 = (int32_t*)(prog + head.o_cs);
  d.chunk_op0 = (int32_t*)(prog + head.o_co);
  d.top = (TopOp*)(prog + head.o_top);
  d.flat = (LeafAbs*)(prog + head.o_flat);
  d.chunk_leaf0 = (int32_t*)(prog + head.o_cl);
  d.chunk_lop0 = (int32_t*)(prog + head.o_clo);
  d.lops = (TopOp*)(prog + head.o_lops);
  d.unit0 = (int32_t*)(prog + head.o_unit);
  d.zero0 = (int32_t*)(prog + head.o_zero);
  d.n_leaves = head.n_leaves;
  d.max_chunk_leaves = head.max_chunk_leaves;
  d.n_spans = head.n_spans;
  d.n_chunks = head.n_chunks;
  // per-call parameters
  const size_t n_par = std::max<size_t>(n_ids + n_cols, 1);
  GK_HIP(gk_pool_malloc(ctx, (void**)&d.base, n_par * sizeof(int32_t)));
  {
    std::vector<int32_t> packed(n_par);   // ids then columns, one copy
    if (n_ids) memcpy(packed.data(), ids, n_ids * sizeof(int32_t));
    if (n_cols) memcpy(packed.data() + n_ids, cols, n_cols * sizeof(int32_t));
    GK_HIP(gk_send(ctx, d.base, packed.data(), n_par * sizeof(int32_t)));
  }
  d.ids = (int32_t*)d.base;
  d.cols = (int32_t*)d.base + n_ids;
  return GK_OK;
}

}  // namespace

extern "C" {

int gk_setmax(gk_ctx* ctx, gk_dptr d_L, int64_t n_rows, int64_t ld, const int32_t* ids, int32_t n_sets, int32_t c,
              gk_dptr d_P);

int gk_maxsum(gk_ctx* ctx, gk_dptr d_L, int64_t n_rows, int64_t ld, const int32_t* ids, int32_t n_sets, int32_t c_prev,
              const int32_t* cols, int32_t n_cols, double* out) {
  gk_bind(ctx);
  GK_REQUIRE(ctx && cols && out && n_rows > 0 && ld >= n_rows && n_cols > 0, "bad maxsum arguments");
  GK_REQUIRE(c_prev >= 0 && c_prev <= kMaxC, "copy number beyond supported set size");
  GK_REQUIRE(n_sets >= 1 && (c_prev == 0 || ids), "missing previous sets");
  // previous sets as single columns: c_prev == 1 -> columns of L itself; c_prev >= 2 -> their
  // row-wise max is materialised once (HBM-bound, typing_mulit_allele.py:569) and used as columns
  hipStream_t st = ctx->stream;
  std::vector<int32_t> pcol_host((size_t)n_sets);
  double* d_P = nullptr;
  // second allele of the search: the sets are the single alleles themselves, so the table is
  // symmetric; run the sets in column order and compute the upper triangle only
  std::vector<int32_t> row_of_set;
  if (c_prev == 1 && n_sets == n_cols && n_sets > kTileT) {
    int32_t max_id = 0;
    for (int j = 0; j < n_cols; ++j) max_id = std::max(max_id, cols[j]);
    std::vector<int32_t> pos((size_t)max_id + 1, -1);
    bool ok = true;
    for (int j = 0; j < n_cols && ok; ++j) {
      ok = cols[j] >= 0 && pos[cols[j]] < 0;
      if (ok) pos[cols[j]] = j;
    }
    row_of_set.resize((size_t)n_sets);
    std::vector<char> seen((size_t)n_cols, 0);
    for (int t = 0; t < n_sets && ok; ++t) {
      ok = ids[t] >= 0 && ids[t] <= max_id && pos[ids[t]] >= 0 && !seen[pos[ids[t]]];
      if (ok) { row_of_set[t] = pos[ids[t]]; seen[row_of_set[t]] = 1; }
    }
    if (!ok) row_of_set.clear();
  }
  const bool symmetric = !row_of_set.empty();
  if (symmetric) {
    for (int t = 0; t < n_sets; ++t) pcol_host[t] = cols[t];
  } else if (c_prev == 1) {
    for (int t = 0; t < n_sets; ++t) pcol_host[t] = ids[t];
  } else if (c_prev >= 2) {
    for (int t = 0; t < n_sets; ++t) pcol_host[t] = t;
    GK_HIP(gk_pool_malloc(ctx, (void**)&d_P, (size_t)n_sets * ld * sizeof(double)));
    int rc0 = gk_setmax(ctx, d_L, n_rows, ld, ids, n_sets, c_prev, gk_addr(d_P));
    if (rc0) return rc0;
  }
  DeviceProgram dp;
  int rc = upload_program(ctx, n_rows, pcol_host.data(), c_prev ? (size_t)n_sets : 0, cols, (size_t)n_cols, dp);
  if (rc) return rc;
  double *d_partial = nullptr, *d_out = nullptr;
  const int64_t n_out = (int64_t)n_sets * n_cols;
  GK_HIP(gk_pool_malloc(ctx, (void**)&d_partial, (size_t)n_out * dp.n_spans * sizeof(double)));
  GK_HIP(gk_pool_malloc(ctx, (void**)&d_out, (size_t)n_out * sizeof(double)));
  const int tiles_t = (n_sets + kTileT - 1) / kTileT, tiles_a = (n_cols + kTileA - 1) / kTileA;
  const dim3 grid((unsigned)(tiles_t * tiles_a * dp.n_spans));
  if (c_prev) {
    GK_PROF_EXACT(ctx, "maxsum_chunks",
            GK_KERNEL(maxsum_chunks<true>, grid, dim3(kThreads), 0, st, gk_ptr<double>(d_L), ld,
                               c_prev >= 2 ? d_P : gk_ptr<double>(d_L), ld, dp.ids, n_sets, dp.cols, n_cols, dp.spans,
                               dp.leaves, tiles_t * tiles_a, symmetric ? 1 : 0, d_partial));
  } else if (n_sets == 1) {
    GK_PROF_EXACT(ctx, "colsum_chunks",
            GK_KERNEL(colsum_chunks<double>, dim3((unsigned)((n_cols + kColsPerGroup - 1) / kColsPerGroup), (unsigned)dp.n_spans),
                      dim3(kThreads), 0, st, TableView<double>{gk_ptr<double>(d_L), nullptr}, ld, dp.cols, n_cols, dp.spans,
                      dp.leaves, d_partial));
  } else {
    GK_PROF_EXACT(ctx, "maxsum_chunks",
            GK_KERNEL(maxsum_chunks<false>, grid, dim3(kThreads), 0, st, gk_ptr<double>(d_L), ld,
                               gk_ptr<double>(d_L), ld, (const int32_t*)nullptr, n_sets, dp.cols, n_cols, dp.spans,
                               dp.leaves, tiles_t * tiles_a, 0, d_partial));
  }
  GK_PROF(ctx, "combine_chunks",
          GK_KERNEL(combine_chunks, dim3((unsigned)((n_out + kCombOut - 1) / kCombOut)), dim3(kThreads), 0, st,
                             d_partial, n_out, dp.n_chunks, dp.chunk_span0, dp.chunk_op0, dp.top, 0.0, d_out));
  GK_HIP(hipGetLastError());
  if (symmetric) {
    std::vector<double> sq((size_t)n_out);
    GK_HIP(gk_fetch(ctx, sq.data(), d_out, (size_t)n_out * sizeof(double)));
    for (int t = 0; t < n_sets; ++t) {
      const int x = row_of_set[t];
      double* dst = out + (size_t)t * n_cols;
      for (int y = 0; y < n_cols; ++y) dst[y] = x <= y ? sq[(size_t)x * n_cols + y] : sq[(size_t)y * n_cols + x];
    }
  } else {
    GK_HIP(gk_fetch(ctx, out, d_out, (size_t)n_out * sizeof(double)));
  }
  gk_pool_free(ctx, d_partial);
  gk_pool_free(ctx, d_out);
  gk_pool_free(ctx, dp.base);
  gk_pool_free(ctx, d_P);
  return GK_OK;
}

int gk_fraction(gk_ctx* ctx, gk_dptr d_L, int64_t n_rows, int64_t ld, const int32_t* ids, int32_t n_sets, int32_t c,
                double* frac_out) {
  gk_bind(ctx);
  GK_REQUIRE(frac_out, "null output");
  GkSumCall call;
  int rc = gk_shares_enqueue(ctx, GkTable{d_L, ld, nullptr}, n_rows, ids, n_sets, c, false, call);
  if (rc == GK_OK && gk_fetch_wait(ctx) != hipSuccess) { gk_set_error("set shares: waiting for the stream failed"); rc = GK_ERR_HIP; }
  if (rc == GK_OK) gk_shares_collect(ctx, call, nullptr, frac_out);
  else gk_release(ctx, call.temps);
  return rc;
}

int gk_setsum(gk_ctx* ctx, gk_dptr d_L, int64_t n_rows, int64_t ld, const int32_t* ids, int32_t n_sets, int32_t c,
              double* value_out, double* frac_out) {
  gk_bind(ctx);
  GK_REQUIRE(value_out && frac_out, "null output");
  GkSumCall call;
  int rc = gk_shares_enqueue(ctx, GkTable{d_L, ld, nullptr}, n_rows, ids, n_sets, c, true, call);
  if (rc == GK_OK && gk_fetch_wait(ctx) != hipSuccess) { gk_set_error("set sums: waiting for the stream failed"); rc = GK_ERR_HIP; }
  if (rc == GK_OK) gk_shares_collect(ctx, call, value_out, frac_out);
  else gk_release(ctx, call.temps);
  return rc;
}

int gk_setmax(gk_ctx* ctx, gk_dptr d_L, int64_t n_rows, int64_t ld, const int32_t* ids, int32_t n_sets, int32_t c,
              gk_dptr d_P) {
  gk_bind(ctx);
  GK_REQUIRE(ctx && ids && n_rows > 0 && ld >= n_rows && n_sets > 0 && c >= 1 && c <= kMaxC, "bad setmax arguments");
  hipStream_t st = ctx->stream;
  int32_t* d_ids = nullptr;
  GK_HIP(gk_pool_malloc(ctx, (void**)&d_ids, (size_t)n_sets * c * sizeof(int32_t)));
  GK_HIP(gk_send(ctx, d_ids, ids, (size_t)n_sets * c * sizeof(int32_t)));
  int64_t want = (n_rows + kThreads - 1) / kThreads;
  unsigned bx = (unsigned)(want < 1024 ? want : 1024);
  GK_PROF(ctx, "setmax_kernel",
          GK_KERNEL(setmax_kernel, dim3(bx, (unsigned)n_sets), dim3(kThreads), 0, st, gk_ptr<double>(d_L), n_rows,
                             ld, d_ids, n_sets, c, gk_ptr<double>(d_P)));
  GK_HIP(hipGetLastError());
  GK_HIP(hipStreamSynchronize(st));
  gk_pool_free(ctx, d_ids);
  return GK_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// the two halves of the set-share / column-sum calls (gk_calls.h)
static int shares_leafwise(gk_ctx* ctx, const GkTable& L, int64_t n_rows, const int32_t* ids, int32_t n_sets, int32_t c,
                           int n_cols, size_t lds, GkSumCall& call) {
  DeviceProgram dp;
  int rc = upload_program(ctx, n_rows, ids, (size_t)n_sets * c, nullptr, 0, dp);
  if (rc) return rc;
  call.temps.push_back(dp.base);
  hipStream_t st = ctx->stream;
  const int per_set = c + 1;
  const int64_t n_out = (int64_t)n_sets * per_set;
  double *d_partial = nullptr, *d_chunk = nullptr, *d_out = nullptr;
  GK_HIP(gk_pool_malloc(ctx, (void**)&d_partial, (size_t)n_out * dp.n_leaves * sizeof(double)));
  call.temps.push_back(d_partial);
  GK_HIP(gk_pool_malloc(ctx, (void**)&d_chunk, (size_t)n_out * dp.n_chunks * sizeof(double)));
  call.temps.push_back(d_chunk);
  // the sums go straight into the pinned result ring (fold_leaves' last workgroups store them there): no copy to queue
  call.back.resize((size_t)n_out);
  GK_HIP(gk_fetch_direct(ctx, call.back.data(), (size_t)n_out * sizeof(double), (void**)&d_out));
  // accumulators per lane group: (c + 1) float64 per set within 128 VGPRs -- 6 sets of 2 alleles, 4 of 3, 3 of 4; a
  // selection beyond 128 x that many sets goes in batches (each batch streams the table once more)
  const int max_slots = c == 2 ? 6 : c == 3 ? 4 : 3;
  const dim3 grid((unsigned)dp.n_leaves);
  for (int set0 = 0; set0 < n_sets; set0 += kLeafGroups * max_slots) {
    const int n_here = std::min(n_sets - set0, kLeafGroups * max_slots);
    const int slots = (n_here + kLeafGroups - 1) / kLeafGroups;
#define GK_LEAF_GO(C, S)                                                                                                  \
  {                                                                                                                       \
    if (lds > 48 * 1024)                                                                                                  \
      GK_HIP(hipFuncSetAttribute((const void*)setsum_leaves<C, S>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
    GK_PROF(ctx, "setsum_leaves",                                                                                           \
            GK_KERNEL((setsum_leaves<C, S>), grid, dim3(kLeafThreads), lds, st, gk_ptr<double>(L.d), L.ld, n_cols,         \
                      dp.ids + (size_t)set0 * C, n_here, n_sets, dp.flat, d_partial + (size_t)set0 * (C + 1)));           \
  }
    if (c == 2) {
      switch (slots) {
        case 1: GK_LEAF_GO(2, 1); break;
        case 2: GK_LEAF_GO(2, 2); break;
        case 3: GK_LEAF_GO(2, 3); break;
        case 4: GK_LEAF_GO(2, 4); break;
        case 5: GK_LEAF_GO(2, 5); break;
        default: GK_LEAF_GO(2, 6); break;
      }
    } else if (c == 3) {
      switch (slots) {
        case 1: GK_LEAF_GO(3, 1); break;
        case 2: GK_LEAF_GO(3, 2); break;
        case 3: GK_LEAF_GO(3, 3); break;
        default: GK_LEAF_GO(3, 4); break;
      }
    } else {
      switch (slots) {
        case 1: GK_LEAF_GO(4, 1); break;
        case 2: GK_LEAF_GO(4, 2); break;
        default: GK_LEAF_GO(4, 3); break;
      }
    }
#undef GK_LEAF_GO
  }
  const size_t fold_lds = (size_t)dp.max_chunk_leaves * kFoldOut * sizeof(double);
  if (fold_lds > 48 * 1024)
    GK_HIP(hipFuncSetAttribute((const void*)fold_leaves, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fold_lds));
  uint32_t* tickets = nullptr;
  const unsigned fold_groups = (unsigned)((n_out + kFoldOut - 1) / kFoldOut);
  { const int trc = gk_ctx_tickets(ctx, fold_groups, &tickets); if (trc) return trc; }
  // the leaves of every chunk in the order of the pairwise recursion, then -- in the workgroup that comes last -- the
  // chunk sums in sequence (numpy's outer reduce loop); the shares are handed back undivided (collect divides)
  GK_PROF(ctx, "fold_leaves",
          GK_KERNEL(fold_leaves, dim3(fold_groups, (unsigned)dp.n_chunks), dim3(kFoldOut), fold_lds, st, d_partial, n_out,
                    dp.chunk_leaf0, dp.chunk_lop0, dp.lops, d_chunk, dp.n_chunks, tickets, d_out));
  GK_HIP(hipGetLastError());
  call.n_rows = n_rows;
  call.n_sets = n_sets;
  call.c = c;
  call.with_value = true;
  call.leafwise = true;
  return GK_OK;
}

int gk_shares_enqueue(gk_ctx* ctx, const GkTable& L, int64_t n_rows, const int32_t* ids, int32_t n_sets, int32_t c,
                      bool with_value, GkSumCall& call) {
  gk_bind(ctx);
  const int64_t ld = L.ld;
  GK_REQUIRE(ctx && L.d && ids && n_rows > 0 && ld >= n_rows && n_sets > 0, "bad fraction arguments");
  GK_REQUIRE(c >= 1 && c <= kMaxC, "copy number beyond supported set size");
  int32_t max_id = 0;
  for (int64_t i = 0; i < (int64_t)n_sets * c; ++i) {
    GK_REQUIRE(ids[i] >= 0, "negative allele id");
    max_id = std::max(max_id, ids[i]);
  }
  // value + shares of a step's selection, leaf by leaf: every column of the table through LDS once (setsum_leaves)
  {
    char form_buf[16];
    const char* const form = gk_test_hook("setsum", form_buf, sizeof(form_buf)) ? form_buf : nullptr;      // tests compare both forms
    const int n_cols = max_id + 1;
    const size_t lds = (size_t)n_cols * kStageLd * sizeof(double);
    // the leaf kernel stages EVERY column below the largest id of the selection: right for a step's few hundred sets,
    // which name most of the gene's alleles -- a handful of sets (exon-first's one-set steps) name a few columns of a
    // gigabyte table, and the tiles read just those
    std::vector<char> named((size_t)n_cols, 0);
    int distinct = 0;
    for (int64_t i = 0; i < (int64_t)n_sets * c; ++i)
      if (!named[ids[i]]) { named[ids[i]] = 1; ++distinct; }
    const bool dense = n_sets >= 64 || distinct * 4 >= n_cols;
    if (with_value && !L.indexed() && c >= 2 && c <= 4 && n_sets <= 4 * kLeafGroups * kLeafSlots && lds <= 158 * 1024 &&
        ((dense && !(form && !strcmp(form, "tiles"))) || (form && !strcmp(form, "leaves"))))
      return shares_leafwise(ctx, L, n_rows, ids, n_sets, c, n_cols, lds, call);
  }
  // Order the sets so that tiles of 32 share columns: the best sets pair a few strong alleles with
  // many partners, so sort by each set's ids taken rarest-first (partners adjacent, hubs shared).
  std::vector<int32_t> freq((size_t)max_id + 1, 0);
  for (int64_t i = 0; i < (int64_t)n_sets * c; ++i) freq[ids[i]]++;
  std::vector<int32_t> key((size_t)n_sets * c), perm((size_t)n_sets);
  for (int k = 0; k < n_sets; ++k) {
    int32_t* kk = key.data() + (size_t)k * c;
    std::copy(ids + (size_t)k * c, ids + (size_t)(k + 1) * c, kk);
    std::sort(kk, kk + c, [&](int32_t x, int32_t y) { return freq[x] != freq[y] ? freq[x] < freq[y] : x < y; });
    perm[k] = k;
  }
  std::sort(perm.begin(), perm.end(), [&](int32_t x, int32_t y) {
    const int32_t *kx = key.data() + (size_t)x * c, *ky = key.data() + (size_t)y * c;
    for (int q = 0; q < c; ++q)
      if (kx[q] != ky[q]) return kx[q] < ky[q];
    return x < y;
  });
  const int n_tiles = (n_sets + kFracSets - 1) / kFracSets;
  std::vector<int32_t> plan;   // [tile_col_off | tile_cols | local_idx | perm]
  std::vector<int32_t> tile_off((size_t)n_tiles + 1, 0), tile_cols, local((size_t)n_sets * c);
  std::vector<int32_t> slot_of((size_t)max_id + 1, -1);
  int max_dist = 1;
  for (int t = 0; t < n_tiles; ++t) {
    const size_t first = tile_cols.size();
    for (int k = t * kFracSets; k < std::min(n_sets, (t + 1) * kFracSets); ++k)
      for (int q = 0; q < c; ++q) {
        const int32_t id = ids[(size_t)perm[k] * c + q];
        if (slot_of[id] < 0) {
          slot_of[id] = (int32_t)(tile_cols.size() - first);
          tile_cols.push_back(id);
        }
        local[(size_t)k * c + q] = slot_of[id];
      }
    for (size_t i = first; i < tile_cols.size(); ++i) slot_of[tile_cols[i]] = -1;
    tile_off[t + 1] = (int32_t)tile_cols.size();
    max_dist = std::max(max_dist, (int)(tile_cols.size() - first));
  }
  plan.insert(plan.end(), tile_off.begin(), tile_off.end());
  const size_t o_cols = plan.size();
  plan.insert(plan.end(), tile_cols.begin(), tile_cols.end());
  const size_t o_local = plan.size();
  plan.insert(plan.end(), local.begin(), local.end());
  const size_t o_perm = plan.size();
  plan.insert(plan.end(), perm.begin(), perm.end());

  DeviceProgram dp;
  int rc = upload_program(ctx, n_rows, plan.data(), plan.size(), nullptr, 0, dp);
  if (rc) return rc;
  call.temps.push_back(dp.base);
  hipStream_t st = ctx->stream;
  double *d_partial = nullptr, *d_out = nullptr;
  const int per_set = c + (with_value ? 1 : 0);
  const int64_t n_out = (int64_t)n_sets * per_set;
  GK_HIP(gk_pool_malloc(ctx, (void**)&d_partial, (size_t)n_out * dp.n_spans * sizeof(double)));
  call.temps.push_back(d_partial);
  call.back.resize((size_t)n_out);
  GK_HIP(gk_fetch_direct(ctx, call.back.data(), (size_t)n_out * sizeof(double), (void**)&d_out));      // combine_chunks stores into the result ring
  const dim3 grid((unsigned)n_tiles, (unsigned)dp.n_spans);
  const size_t lds = (size_t)max_dist * kFracLd * sizeof(double);   // <= 256 columns * 41 * 8 = 84 KB
#define GK_FRAC_GO(C, V, TL, VIEW)                                                                                   \
  {                                                                                                                  \
    if (lds > 48 * 1024)                                                                                             \
      GK_HIP(hipFuncSetAttribute((const void*)fraction_chunks<C, V, TL>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                 (int)lds));                                                                         \
    GK_PROF(ctx, "fraction_chunks",                                                                                      \
            GK_KERNEL((fraction_chunks<C, V, TL>), grid, dim3(kThreads), lds, st, VIEW, ld, dp.ids, dp.ids + o_cols, \
                      dp.ids + o_local, dp.ids + o_perm, n_sets, dp.spans, dp.leaves, d_partial));                   \
  }
#define GK_FRAC_LAUNCH_V(C, V)                                                                          \
  if (L.indexed()) GK_FRAC_GO(C, V, uint16_t, (TableView<uint16_t>{gk_ptr<uint16_t>(L.d), L.vals}))     \
  else GK_FRAC_GO(C, V, double, (TableView<double>{gk_ptr<double>(L.d), nullptr}))
#define GK_FRAC_LAUNCH(C)                 \
  if (with_value) { GK_FRAC_LAUNCH_V(C, true); } \
  else { GK_FRAC_LAUNCH_V(C, false); }
  switch (c) {
    case 1: GK_FRAC_LAUNCH(1); break;
    case 2: GK_FRAC_LAUNCH(2); break;
    case 3: GK_FRAC_LAUNCH(3); break;
    case 4: GK_FRAC_LAUNCH(4); break;
    case 5: GK_FRAC_LAUNCH(5); break;
    case 6: GK_FRAC_LAUNCH(6); break;
    case 7: GK_FRAC_LAUNCH(7); break;
    default: GK_FRAC_LAUNCH(8); break;
  }
#undef GK_FRAC_LAUNCH
#undef GK_FRAC_LAUNCH_V
#undef GK_FRAC_GO
  // with the value riding along the sums are handed back undivided (collect divides the shares by n_rows)
  GK_PROF(ctx, "combine_chunks",
          GK_KERNEL(combine_chunks, dim3((unsigned)((n_out + kCombOut - 1) / kCombOut)), dim3(kThreads), 0, st,
                             d_partial, n_out, dp.n_chunks, dp.chunk_span0, dp.chunk_op0, dp.top,
                             with_value ? 0.0 : (double)n_rows, d_out));
  GK_HIP(hipGetLastError());
  call.n_rows = n_rows;
  call.n_sets = n_sets;
  call.c = c;
  call.with_value = with_value;
  return GK_OK;
}

void gk_shares_collect(gk_ctx* ctx, GkSumCall& call, double* value_out, double* frac_out) {
  const int c = call.c;
  if (!call.with_value) {
    if (frac_out) std::copy(call.back.begin(), call.back.end(), frac_out);
  } else {
    const double rows = (double)call.n_rows;
    const int per_set = c + 1;
    for (int k = 0; k < call.n_sets; ++k) {
      const double* src = call.back.data() + (size_t)k * per_set;
      if (frac_out)
        for (int q = 0; q < c; ++q) frac_out[(size_t)k * c + q] = src[q] / rows;   // the IEEE quotient numpy forms (580)
      if (value_out) value_out[k] = src[c];
    }
  }
  gk_release(ctx, call.temps);
}

int gk_colsum_enqueue(gk_ctx* ctx, const GkTable& L, int64_t n_rows, const int32_t* cols, int32_t n_cols,
                      GkSumCall& call) {
  gk_bind(ctx);
  const int64_t ld = L.ld;
  GK_REQUIRE(ctx && L.d && cols && n_rows > 0 && ld >= n_rows && n_cols > 0, "bad column-sum arguments");
  DeviceProgram dp;
  int rc = upload_program(ctx, n_rows, nullptr, 0, cols, (size_t)n_cols, dp);
  if (rc) return rc;
  call.temps.push_back(dp.base);
  hipStream_t st = ctx->stream;
  double *d_partial = nullptr, *d_out = nullptr;
  GK_HIP(gk_pool_malloc(ctx, (void**)&d_partial, (size_t)n_cols * dp.n_spans * sizeof(double)));
  call.temps.push_back(d_partial);
  call.back.resize((size_t)n_cols);
  GK_HIP(gk_fetch_direct(ctx, call.back.data(), (size_t)n_cols * sizeof(double), (void**)&d_out));      // combine_chunks stores into the result ring
  const dim3 cgrid((unsigned)((n_cols + kColsPerGroup - 1) / kColsPerGroup), (unsigned)dp.n_spans);
  if (L.indexed())
    GK_PROF_EXACT(ctx, "colsum_chunks",
                  GK_KERNEL(colsum_chunks<uint16_t>, cgrid, dim3(kThreads), 0, st,
                            TableView<uint16_t>{gk_ptr<uint16_t>(L.d), L.vals}, ld, dp.cols, n_cols, dp.spans, dp.leaves,
                            d_partial));
  else
    GK_PROF_EXACT(ctx, "colsum_chunks",
                  GK_KERNEL(colsum_chunks<double>, cgrid, dim3(kThreads), 0, st,
                            TableView<double>{gk_ptr<double>(L.d), nullptr}, ld, dp.cols, n_cols, dp.spans, dp.leaves,
                            d_partial));
  GK_PROF(ctx, "combine_chunks",
          GK_KERNEL(combine_chunks, dim3((unsigned)((n_cols + kCombOut - 1) / kCombOut)), dim3(kThreads), 0, st,
                             d_partial, (int64_t)n_cols, dp.n_chunks, dp.chunk_span0, dp.chunk_op0, dp.top, 0.0, d_out));
  GK_HIP(hipGetLastError());
  call.n_rows = n_rows;
  call.n_sets = n_cols;
  return GK_OK;
}

void gk_colsum_collect(gk_ctx* ctx, GkSumCall& call, double* out) {
  if (out) std::copy(call.back.begin(), call.back.end(), out);
  gk_release(ctx, call.temps);
}

int gk_expand_table(gk_ctx* ctx, const GkTable& L, int64_t n_rows, int32_t n_allele, gk_dptr d_L, int64_t ld) {
  gk_bind(ctx);
  GK_REQUIRE(ctx && L.d && L.indexed() && d_L && n_rows > 0 && n_allele > 0 && ld >= n_rows && L.ld >= n_rows,
             "bad table expansion");
  int64_t want = (n_rows + kThreads - 1) / kThreads;
  GK_KERNEL(expand_index, dim3((unsigned)(want < 256 ? want : 256), (unsigned)n_allele), dim3(kThreads), 0, ctx->stream,
            gk_ptr<uint16_t>(L.d), L.ld, n_rows, L.vals, gk_ptr<double>(d_L), ld);
  GK_HIP(hipGetLastError());
  return GK_OK;
}

extern "C" int gk_expand_index(gk_ctx* ctx, gk_lut* lut, gk_dptr d_lidx, int64_t ldi, int64_t n_rows, int32_t n_allele,
                               gk_dptr d_L, int64_t ld) {
  GK_REQUIRE(lut, "null value table");
  return gk_expand_table(ctx, GkTable{d_lidx, ldi, lut->d_vals}, n_rows, n_allele, d_L, ld);
}


// Per-gene read selection, variant error correction and the read x allele compatibility table.
//
//   gk_select_gene      removeMultipleMapped + groupReads      hisat2.py:943-948, kir_typing.py:15-20
//   gk_variant_count /
//   gk_variant_correct  AlleleTyping.errorCorrection            typing_mulit_allele.py:302-338
//   gk_select_nonempty  AlleleTyping.removeEmptyReads           typing_mulit_allele.py:274-281
//   gk_compat(_log)     reads2AlleleProb / read2Onehot / onehot2Prob   typing_mulit_allele.py:287-300, 340-381
//                       (_log: with np.log10 of line 263 applied through the value table)
//
// Compatibility kernel: one wavefront per read pair, lanes = alleles (1-4 allele slots per lane,
// up to 256 alleles per pass).  The pair's variant ordinals are wave-uniform; the bit rows of a chunk of 64
// ordinals are laid down in LDS ([variant][word], per wave) and every lane reads the word that holds its allele's bit,
// turns the bit into a select mask and picks the halves of 0.999 / 0.001, multiplied in the reference's factor order
// (lpv, rpv, lnv, rnv), so the double product is bit-identical to numpy's sequential multiply.reduce.
// (Round 3 tried the lane masks in SGPR pairs through scalar loads -- 3 VALU per factor instead of 4 -- and lost to
// the scalar cache's miss path: DESIGN.md section 8, profiles/r03_compat_variants.txt.)
#include <algorithm>

#include "gk_common.h"
#include "gk_lut.h"

namespace {

constexpr int kThreads = 256;
inline unsigned nblk(int64_t n, int t = kThreads) { return (unsigned)((n + t - 1) / t); }

// Rows grouped by backbone, row order kept (a stable counting sort with one tile per wavefront):
// part_count writes, for every wave of 64 rows, how many selected rows each backbone has (bin-major,
// so ONE exclusive scan yields every (backbone, wave) output offset); part_scatter ranks a row
// among the lanes of its wave with the same backbone.  Replaces one flag + scan + scatter per gene.
template <bool kScatter>
__global__ __launch_bounds__(kThreads) void part_pass(const uint8_t* gene, const uint8_t* nh, int64_t n, int multiple,
                                                      int n_bins, int64_t n_tiles, uint32_t* hist, int32_t* out) {
  const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const int64_t tile = i >> 6;
  int bin = -1;
  if (i < n && (multiple || nh[i] == 1)) {
    bin = gene[i];
    if (bin >= n_bins) bin = -1;
  }
  uint64_t todo = __ballot(bin >= 0);
  while (todo) {
    const int leader = __ffsll((unsigned long long)todo) - 1;
    const int b = __shfl(bin, leader, 64);
    const uint64_t same = __ballot(bin == b);
    if (kScatter) {
      if (bin == b) out[hist[(int64_t)b * n_tiles + tile] + __popcll(same & ((1ull << lane) - 1ull))] = (int32_t)i;
    } else if (lane == leader) {
      hist[(int64_t)b * n_tiles + tile] = (uint32_t)__popcll(same);
    }
    todo &= ~same;
  }
}

__global__ void part_offsets(const uint32_t* hist, const uint32_t* total, int n_bins, int64_t n_tiles, uint32_t* goff) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < n_bins) goff[b] = hist[(int64_t)b * n_tiles];
  if (b == n_bins) goff[b] = *total;
}

// Tally of the surviving ids (AlleleTyping.errorCorrection 302-338 counts them per variant).  Sixteen
// lanes per row, grid-stride: the row's ids (lpv, rpv, lnv, rnv are contiguous in the CSR) are read
// 16 at a time; positives are the ids before the row's mid offset.  Counters of the gene's index
// variants [vbeg, vend) are privatised in LDS (positive / negative halves) -- the ids of one row are
// distinct, so most atomics of a wave instruction hit different counters -- and flushed once per
// workgroup; ordinals outside that range (novel variants) go straight to global atomics.
__global__ __launch_bounds__(kThreads) void count_ids(const int32_t* __restrict__ rows, int64_t n_rows,
                                                      const uint32_t* __restrict__ off,
                                                      const uint32_t* __restrict__ ids,
                                                      const uint8_t* __restrict__ vflag, uint32_t* cnt_pos,
                                                      uint32_t* cnt_neg, int vbeg, int n_local) {
  extern __shared__ uint32_t hist[];   // [2][n_local]
  for (int i = threadIdx.x; i < 2 * n_local; i += kThreads) hist[i] = 0;
  __syncthreads();
  // a group of 16 lanes per row: four rows of a wavefront are in flight at a time (the loads of a row
  // -- row number, offsets, ids, flags -- depend on each other, a list has ~75 ids)
  constexpr int kGroup = 16;
  const int lane = threadIdx.x & (kGroup - 1);
  const int64_t group = ((int64_t)blockIdx.x * kThreads + threadIdx.x) / kGroup;
  const int64_t n_groups = ((int64_t)gridDim.x * kThreads) / kGroup;
  for (int64_t i = group; i < n_rows; i += n_groups) {
    const int64_t row = rows[i];
    const uint32_t b = off[4 * row], mid = off[4 * row + 2], e = off[4 * row + 4];
    for (uint32_t k = b + lane; k < e; k += kGroup) {
      const uint32_t v = ids[k];
      const bool positive = k < mid;
      if (vflag[v] & (positive ? 1 : 2)) continue;
      const uint32_t l = v - (uint32_t)vbeg;
      if (l < (uint32_t)n_local) atomicAdd(&hist[(positive ? 0 : n_local) + l], 1u);
      else atomicAdd(positive ? &cnt_pos[v] : &cnt_neg[v], 1u);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * n_local; i += kThreads) {
    const uint32_t c = hist[i];
    if (c) atomicAdd(i < n_local ? &cnt_pos[vbeg + i] : &cnt_neg[vbeg + i - n_local], c);
  }
}

// thresholds of errorCorrection: P+N < 3 drops both sides, P/(P+N) < 0.2 drops positives,
// N/(P+N) < 0.2 drops negatives.  x/(x+y) < 0.2 in IEEE double <=> 5x < x+y for counts < 2^50.
__global__ __launch_bounds__(kThreads) void apply_correction(const uint32_t* cnt_pos, const uint32_t* cnt_neg,
                                                             int64_t n, uint8_t* vflag) {
  const int64_t v = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (v >= n) return;
  const uint64_t p = cnt_pos[v], q = cnt_neg[v];
  if (p + q == 0) return;
  uint8_t f = vflag[v];
  if (p + q < 3) {
    f |= 3;
  } else {
    if (5 * p < p + q) f |= 1;
    if (5 * q < p + q) f |= 2;
  }
  vflag[v] = f;
}

__global__ __launch_bounds__(kThreads) void flag_nonempty(const int32_t* rows, int64_t n_rows, const uint32_t* off,
                                                          const uint32_t* ids, const uint8_t* vflag, uint32_t* flag,
                                                          uint32_t* zero, int n_zero) {
  const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (zero && i < n_zero) zero[i] = 0;      // the per-gene counters of the launch that follows
  if (i >= n_rows) return;
  const int64_t row = rows[i];
  const uint32_t b = off[4 * row], mid = off[4 * row + 2], e = off[4 * row + 4];
  uint32_t alive = 0;
  for (uint32_t k = b; k < e && !alive; ++k) alive = !(vflag[ids[k]] & (k < mid ? 1 : 2));
  flag[i] = alive;
}

// The same flags for every row of a sample (gk_sample_prepare).  The rows of a gene are every n-th pair of the sample: one
// thread per row of the gene-grouped order took a 64-byte line of the offsets and one of the ids per load (~1 TB/s of
// scattered lines, 1.96 ms per 10 M rows).  Here in PAIR order -- the offsets and the heads of the lists stream -- four
// lanes per pair, four ids at a time until one survives; `alive` is then gathered through the rows.  `maybe[p]` == 0: the
// last tally found every id of pair p dropped already (drop flags only grow) -- in the exon model two pairs in three,
// whose lists would otherwise be walked to their ends (1.74 ms per 10 M pairs against 0.27 ms for the full model).
__global__ __launch_bounds__(kThreads) void flag_pairs(int64_t n_pairs, const uint32_t* __restrict__ off,
                                                       const uint32_t* __restrict__ ids, const uint8_t* __restrict__ vflag,
                                                       const uint8_t* __restrict__ maybe, uint8_t* __restrict__ alive) {
  const int64_t t = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  const int64_t p = t >> 2;
  const uint32_t l = (uint32_t)t & 3u;
  uint32_t b = 0, mid = 0, e = 0;
  if (p < n_pairs && maybe[p]) { b = off[4 * p]; mid = off[4 * p + 2]; e = off[4 * p + 4]; }
  const int shift = (threadIdx.x & 63) & ~3;
  bool any = false;
  for (uint32_t k0 = b; k0 < e && !any; k0 += 4) {      // the same trip count for the four lanes of a pair
    const uint32_t k = k0 + l;
    const bool mine = k < e && !(vflag[ids[k]] & (k < mid ? 1 : 2));
    any = ((__ballot(mine) >> shift) & 0xFull) != 0ull;
  }
  if (p < n_pairs && l == 0) alive[p] = any ? 1 : 0;
}

__global__ __launch_bounds__(kThreads) void gather_pair_flags(const int32_t* __restrict__ rows, int64_t n_rows,
                                                              const uint8_t* __restrict__ alive, uint32_t* __restrict__ flag,
                                                              uint32_t* zero, int n_zero) {
  const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (zero && i < n_zero) zero[i] = 0;      // the per-gene counters of the launch that follows
  if (i < n_rows) flag[i] = alive[rows[i]];
}

// The same tally for ALL genes of a sample in one launch (gk_sample_prepare): `rows` are the rows grouped by
// backbone, workgroup b owns rows [wg_row0[b], wg_row1[b]) of gene wg_gene[b] -- one gene per workgroup, so
// the LDS counters cover that gene's index variants [gene_vbeg[g], gene_vbeg[g + 1]).
__global__ __launch_bounds__(kThreads) void count_ids_genes(const int32_t* __restrict__ rows,
                                                            const int32_t* __restrict__ wg_gene,
                                                            const int64_t* __restrict__ wg_row0,
                                                            const int64_t* __restrict__ wg_row1,
                                                            const int32_t* __restrict__ gene_vbeg, int max_local,
                                                            const uint32_t* __restrict__ off,
                                                            const uint32_t* __restrict__ ids,
                                                            const uint8_t* __restrict__ vflag, uint32_t* cnt_pos,
                                                            uint32_t* cnt_neg, uint8_t* __restrict__ maybe) {
  extern __shared__ uint32_t hist[];   // [2][n_local]
  const int g = wg_gene[blockIdx.x];
  const int vbeg = gene_vbeg[g];
  const int n_local = min(gene_vbeg[g + 1] - vbeg, max_local);
  for (int i = threadIdx.x; i < 2 * n_local; i += kThreads) hist[i] = 0;
  __syncthreads();
  // Sixteen lanes per row.  The loads of a row depend on each other (row number -> list offsets -> ids -> drop flags): as
  // a loop with a test per id the compiler put a wait behind every link and sank the flag loads into the branches that
  // use them (five dependent waits per row: 6.6 us per iteration at configs[2]).  So the walk is straight-line code in
  // which every load is unconditional (clamped indices) and every counter update too (an id that does not count adds to a
  // spare counter of the lane): per row the first 64 ids are requested FIRST, behind them the offsets of the group's next
  // row and the number of the row after that; then the four flags; two waits per row.  A list beyond 64 ids takes the rest
  // 64 at a time.
  constexpr int kGroup = 16, kDeep = 4;
  const int lane = threadIdx.x & (kGroup - 1);
  uint32_t* const spare = hist + 2 * max_local + threadIdx.x;      // a counter nobody reads, one per thread
  const int64_t r0 = wg_row0[blockIdx.x], r1 = wg_row1[blockIdx.x];      // r1 > r0
  constexpr int64_t kStep = kThreads / kGroup;
  int64_t i = r0 + threadIdx.x / kGroup;
  auto row_at = [&](int64_t j) { return rows[j < r1 ? j : r1 - 1]; };
  int32_t row = row_at(i), row1 = row_at(i + kStep);
  uint32_t b = off[4 * (int64_t)row], mid = off[4 * (int64_t)row + 2], e = off[4 * (int64_t)row + 4];
  if (i >= r1) b = mid = e = 0u;
  // the tally of up to 64 ids (k0 + 16 j of this lane) whose flags are at hand
  auto tally = [&](uint32_t k0, const uint32_t (&v)[kDeep], const uint8_t (&f)[kDeep]) {
    bool counted = false;
#pragma unroll
    for (int j = 0; j < kDeep; ++j) {
      const uint32_t k = k0 + kGroup * j;
      const bool positive = k < mid;
      const bool counts = k < e && !(f[j] & (positive ? 1 : 2));
      const uint32_t l = v[j] - (uint32_t)vbeg;
      const bool local = l < (uint32_t)n_local;
      atomicAdd((counts && local) ? &hist[(positive ? 0 : n_local) + l] : spare, (counts && local) ? 1u : 0u);
      if (counts && !local) atomicAdd(positive ? &cnt_pos[v[j]] : &cnt_neg[v[j]], 1u);      // a novel variant
      counted = counted || counts;
    }
    return counted;
  };
  auto ids_of = [&](uint32_t k0, uint32_t (&v)[kDeep]) {
    const uint32_t last = e ? e - 1u : 0u;
#pragma unroll
    for (int j = 0; j < kDeep; ++j) v[j] = ids[k0 + kGroup * j < e ? k0 + kGroup * j : last];
  };
  for (; i < r1; i += kStep) {
    uint32_t v[kDeep];
    uint8_t f[kDeep];
    ids_of(b + lane, v);
    const int64_t next = row1;                                  // arrived: requested an iteration ago
    const uint32_t nb = off[4 * next], nmid = off[4 * next + 2], ne = off[4 * next + 4];
    row1 = row_at(i + 2 * kStep);
#pragma unroll
    for (int j = 0; j < kDeep; ++j) f[j] = vflag[v[j]];
    bool counted = tally(b + lane, v, f);      // an id of this row went into a tally: the row may outlive the correction (flag_pairs)
    for (uint32_t k0 = b + kGroup * kDeep + lane; k0 < e; k0 += kGroup * kDeep) {
      ids_of(k0, v);
#pragma unroll
      for (int j = 0; j < kDeep; ++j) f[j] = vflag[v[j]];
      counted = tally(k0, v, f) || counted;
    }
    if (maybe) {
      const bool any = ((__ballot(counted) >> ((threadIdx.x & 63) & ~(kGroup - 1))) & 0xFFFFull) != 0ull;
      if (lane == 0) maybe[row] = any ? 1 : 0;
    }
    row = (int32_t)next;
    const bool more = i + kStep < r1;
    b = more ? nb : 0u; mid = more ? nmid : 0u; e = more ? ne : 0u;
  }
  __syncthreads();
  for (int i2 = threadIdx.x; i2 < 2 * n_local; i2 += kThreads) {
    const uint32_t c = hist[i2];
    if (c) atomicAdd(i2 < n_local ? &cnt_pos[vbeg + i2] : &cnt_neg[vbeg + i2 - n_local], c);
  }
}

// kept[g] += rows of gene g with flag != 0 (rows grouped by gene: [gene_off[g], gene_off[g + 1])); gridDim.y workgroups
// share a gene (one workgroup per gene spent 70 us waiting for its own loads); `kept` is zero before the launch
constexpr int kFlagSlices = 16;
__global__ __launch_bounds__(kThreads) void count_flags_per_gene(const uint32_t* __restrict__ flag,
                                                                 const int64_t* __restrict__ gene_off,
                                                                 uint32_t* __restrict__ kept) {
  const int g = blockIdx.x;
  const int64_t g0 = gene_off[g], g1 = gene_off[g + 1];
  const int64_t per = (g1 - g0 + gridDim.y - 1) / gridDim.y;
  const int64_t a = g0 + per * blockIdx.y, b = min(g1, a + per);
  uint32_t c = 0;
  for (int64_t i = a + threadIdx.x; i < b; i += kThreads) c += flag[i] != 0;
  __shared__ uint32_t part[kThreads];
  part[threadIdx.x] = c;
  __syncthreads();
  for (int s2 = kThreads / 2; s2 > 0; s2 >>= 1) {
    if (threadIdx.x < s2) part[threadIdx.x] += part[threadIdx.x + s2];
    __syncthreads();
  }
  if (threadIdx.x == 0 && part[0]) atomicAdd(&kept[g], part[0]);
}

constexpr int kMaxSlots = 4;   // alleles per lane and pass: up to 256 alleles per pass
constexpr int kCompatWaves = 8;          // 8 waves share one 16-row tile: 32 waves per CU at 34 KB of LDS per block
constexpr int kCompatThreads = 64 * kCompatWaves;
constexpr int kTileRows = 16;            // rows per output tile (2 per wave)
constexpr int kTileLd = kTileRows + 1;   // padded LDS stride (doubles) of the transposed tile

// bit matrix [variant][words] -> [word][variant]: in the compatibility kernel lane k reads word w of the
// k-th variant of a window, and windows are runs of consecutive ordinals, so the word-major copy turns
// 64 strided row reads into one coalesced 256-byte read per word
__global__ __launch_bounds__(kThreads) void transpose_mask(const uint32_t* mask, int n_span, int words, uint32_t* out,
                                                           uint32_t* zero_word) {
  if (zero_word && blockIdx.x == 0 && threadIdx.x == 0) *zero_word = 0;   // the flag word of the launches that follow
  const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (i >= (int64_t)n_span * words) return;
  const int w = (int)(i / n_span), v = (int)(i % n_span);
  out[i] = mask[(int64_t)v * words + w];
}

// One wavefront per read pair, lanes = alleles (kSlots allele slots per lane; a gene of <= 256
// alleles is one pass, and the last pass of a wider gene only carries the slots it needs).  Per chunk
// of 64 variant ordinals, lane k loads ordinal k, its drop flag and the 2*kSlots bit-row words of
// that variant that this pass needs (windows are runs of consecutive ordinals, so these are coalesced row reads of the
// L2-resident, word-major bit matrix).  The KEPT variants of the chunk are laid down in LDS back to back, in list order
// (a lane's place = the kept lanes below it: v_mbcnt), the rows of the negative ones inverted -- a set bit then means
// "allele and read agree" for both signs -- so the walk over them is a counted loop of straight-line code, four variants
// per round: per factor and slot a lane reads the word holding its allele's bit (a two-address broadcast read),
// sign-extends the bit into a select mask (v_bfe_i32), picks the factor's halves (2 x v_bfi_b32) and multiplies --
// 0.999 / 0.001 in the reference's order.  (Rounds 2 - 4 walked a scalar bit set of the kept variants with a branch per
// variant and, from round 4, a test and two branches per slot for slots nobody carries the variant in: ~19 scalar
// instructions and 4 - 7 taken branches per 12 vector ones.)
// Results of a 16-row tile are transposed through LDS so that the column-major [allele][row]
// output is written as runs instead of one store per (allele, row).
//
// kLog: the tile is mapped through the log10 value table on its way out (typing_mulit_allele.py:263),
// so the table of log-probabilities is the only thing written.  A value whose log10 the host has not
// evaluated yet is inserted into the table and stored as NaN; the host sees the table grow, evaluates
// numpy.log10 for the new values and runs the kernel once more.
// kIdx (with kLog): the table holds, per (allele, read), the DENSE INDEX of the log-likelihood in the value table
// (uint16 [allele][ldm] through `lidx`) instead of the float64 itself -- 2 bytes + the mismatch byte per entry
// instead of 8 + 1; the reductions gather the float64 from the table's value array (gk_search.hip).  An index
// beyond 65534 raises bit 1 of the flag word: the caller then takes the float64 form for this gene.
constexpr uint16_t kNoIndex = 0xFFFFu;   // "log10 not defined yet" in the index table (rewritten by the next pass)

template <bool kLog, int kSlots, bool kMiss, bool kIdx>
__global__ __launch_bounds__(kCompatThreads) void compat_kernel(const int32_t* rows, int64_t n_rows, const uint32_t* off,
                                                                const uint32_t* ids, const uint8_t* vflag, int vbeg, int vend,
                                                                const uint32_t* mask_t, int words, int n_allele, int a_base,
                                                                double* probs, uint8_t* miss_out, uint16_t* nvar_out,
                                                                LutView lut, double empty_p, uint8_t* miss8, int64_t ldm,
                                                                uint32_t* bound_flags, uint16_t* lidx) {
  const int n_span = vend - vbeg;   // mask_t: [words][n_span], see transpose_mask
  constexpr int kPassAlleles = 64 * kSlots;
  constexpr int kPassWords = 2 * kSlots;   // bit-row words covering one pass
  __shared__ double tile[kPassAlleles * kTileLd];
  __shared__ uint32_t wave_rows[kCompatWaves][65 * kPassWords];      // 64 kept rows of a chunk + one nobody reads
  // 0.999 = 0x3FEFF7CED916872B, 0.001 = 0x3F50624DD2F1A9FC
  constexpr int32_t kHi999 = 0x3FEFF7CE, kLo999 = (int32_t)0xD916872B, kHi001 = 0x3F50624D, kLo001 = (int32_t)0xD2F1A9FC;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int w_base = a_base >> 5;   // a_base is a multiple of 256 = 8 words

  int a[kSlots];
  bool live[kSlots];
#pragma unroll
  for (int s = 0; s < kSlots; ++s) {
    a[s] = a_base + lane + 64 * s;
    live[s] = a[s] < n_allele;
  }
  const int n_pass = min(kPassAlleles, n_allele - a_base);
  const int64_t n_tiles = (n_rows + kTileRows - 1) / kTileRows;
  // Tiles to workgroups, XCD by XCD: a 128-byte line of the mismatch table holds the bytes of EIGHT consecutive tiles
  // (16 rows each).  Workgroup b runs on XCD b % 8 (round-robin dispatch), each XCD has an L2 of its own, and a line
  // whose pieces are written through different L2s leaves each of them as a masked partial write (a read-modify-write
  // at the memory side: 1.7 x the kernel's algorithmic traffic in round 4).  So the tiles of a line go to workgroups of
  // ONE XCD, whose write-back L2 puts the line together: group g = tiles 8 g .. 8 g + 7 belongs to XCD g % 8, and the
  // workgroups of an XCD share out the tiles of its groups one by one.  The grid is a multiple of 8 workgroups.
  const int64_t n_groups = (n_tiles + 7) / 8;
  const int xcd = blockIdx.x & 7;
  const int64_t my_groups = n_groups > xcd ? (n_groups - xcd + 7) / 8 : 0;
  const int64_t wg_per_xcd = gridDim.x >> 3;
  for (int64_t u = blockIdx.x >> 3; u < 8 * my_groups; u += wg_per_xcd) {
    const int64_t tile_i = 8 * (xcd + 8 * (u >> 3)) + (u & 7);
    if (tile_i >= n_tiles) continue;
    const int64_t row0 = tile_i * kTileRows;
    // the list offsets of the wave's rows of this tile, requested together (row number -> offsets -> ids -> bit rows is a
    // chain of four loads per row; the second row's first two links ride on the first row's)
    constexpr int kRowsPerWave = kTileRows / kCompatWaves;
    uint32_t row_b[kRowsPerWave], row_mid[kRowsPerWave], row_e[kRowsPerWave];
#pragma unroll
    for (int q = 0; q < kRowsPerWave; ++q) {
      const int64_t i = row0 + wid * kRowsPerWave + q;
      const int64_t row = rows[i < n_rows ? i : n_rows - 1];
      row_b[q] = off[4 * row]; row_mid[q] = off[4 * row + 2]; row_e[q] = off[4 * row + 4];
    }
    uint32_t first_id[kRowsPerWave];      // a lane's id of each row's first chunk: the third link, for all rows at once
#pragma unroll
    for (int q = 0; q < kRowsPerWave; ++q) {
      const uint32_t k = row_b[q] + (uint32_t)lane;
      first_id[q] = ids[k < row_e[q] ? k : (row_e[q] ? row_e[q] - 1u : 0u)];
    }
#pragma unroll
    for (int q = 0; q < kRowsPerWave; ++q) {
      const int rt = wid * kRowsPerWave + q;   // row inside the tile
      const int64_t i = row0 + rt;
      if (i >= n_rows) break;                                   // wave-uniform
      const uint32_t b = __builtin_amdgcn_readfirstlane(row_b[q]);
      const uint32_t mid = __builtin_amdgcn_readfirstlane(row_mid[q]);
      const uint32_t e = __builtin_amdgcn_readfirstlane(row_e[q]);
      double p[kSlots];
      uint32_t miss[kSlots];
      uint32_t nvar = 0;
#pragma unroll
      for (int s = 0; s < kSlots; ++s) { p[s] = 1.0; miss[s] = 0; }
      // this lane's word of the t-th kept variant and slot s: wave's base + the lane's half + t * kPassWords + 2 s
      const uint32_t* const my_words = &wave_rows[wid][lane >> 5];
      const int my_bit = lane & 31;
      for (uint32_t base = b; base < e; base += 64) {
        const uint32_t k = base + lane;
        // Straight-line loads (no branch per word: the compiler then waits for the first word before it asks for the
        // second, and keeps the uniform guards in spilled scalars): a lane past the end of the list repeats the last id
        // and keeps nothing; a variant outside the gene's span (novel: no allele carries it) reads row 0 and takes zeros;
        // only the LAST word of a pass can lie beyond the gene's words (kSlots is the number of slots the pass needs).
        const bool in = k < e;
        const uint32_t v = base == b ? first_id[q] : ids[in ? k : e - 1u];
        const uint32_t local = v - (uint32_t)vbeg;
        const bool indexed = local < (uint32_t)n_span;
        const uint32_t at = indexed ? local : 0u;
        const uint8_t dropped = vflag[v];
        const uint32_t flip = k >= mid ? 0xFFFFFFFFu : 0u;     // "the allele lacks it" = agreement with a negative id
        uint32_t mrow[kPassWords];
#pragma unroll
        for (int w = 0; w < kPassWords; ++w) {
          const bool has = w + 1 < kPassWords || w_base + w < words;                     // uniform
          const uint32_t* const col = mask_t + (int64_t)(has ? w_base + w : 0) * n_span;     // uniform
          const uint32_t x = col[at];
          mrow[w] = ((indexed && has) ? x : 0u) ^ flip;
        }
        const bool my_keep = in && !(dropped & (k < mid ? 1 : 2));
        // the kept variants of the chunk back to back in LDS, in list order (positives, then negatives)
        const uint64_t kept = __ballot(my_keep);
        const int n_kept = __builtin_popcountll(kept);
        nvar += (uint32_t)n_kept;
        {      // every lane stores (a lane that keeps nothing into the row behind the 64: the loads above are then not
               // sunk below the keep test, where they would wait for the drop flag first)
          const uint32_t place = __builtin_amdgcn_mbcnt_hi((uint32_t)(kept >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)kept, 0u));
          uint32_t* const dst = &wave_rows[wid][(my_keep ? place : 64u) * kPassWords];
#pragma unroll
          for (int w = 0; w < kPassWords; ++w) dst[w] = mrow[w];
        }
        __builtin_amdgcn_wave_barrier();   // LDS operations of a wave are executed in order
        // one factor per slot: bit -> select mask -> the factor's two halves -> multiply (1.0 * f == f)
        auto apply = [&](const uint32_t (&w)[kSlots]) {
#pragma unroll
          for (int s = 0; s < kSlots; ++s) {
            const int32_t m = __builtin_amdgcn_sbfe((int32_t)w[s], my_bit, 1);   // -1: allele and read agree
            p[s] *= __hiloint2double((m & kHi999) | (~m & kHi001), (m & kLo999) | (~m & kLo001));
            if (kMiss) miss[s] += (uint32_t)(m + 1);
          }
        };
        int t = 0;
        for (; t + 4 <= n_kept; t += 4) {      // four variants per round: their words requested together, then 16 x kSlots VALU
          uint32_t w4[4][kSlots];
          const uint32_t* const at = my_words + t * kPassWords;
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int s = 0; s < kSlots; ++s) w4[j][s] = at[j * kPassWords + 2 * s];
#pragma unroll
          for (int j = 0; j < 4; ++j) apply(w4[j]);
        }
        for (; t < n_kept; ++t) {
          uint32_t w1[kSlots];
#pragma unroll
          for (int s = 0; s < kSlots; ++s) w1[s] = my_words[t * kPassWords + 2 * s];
          apply(w1);
        }
        __builtin_amdgcn_wave_barrier();   // the next chunk overwrites the rows
      }
#pragma unroll
      for (int s = 0; s < kSlots; ++s) {
        if (live[s]) {
          // a read without any kept variant: 1.0, or 0.999 for every allele when such reads stay in the
          // model (no_empty=False, typing_mulit_allele.py:372-374)
          tile[(lane + 64 * s) * kTileLd + rt] = nvar ? p[s] : empty_p;
          if (kMiss && miss_out) miss_out[(int64_t)a[s] * n_rows + i] = (uint8_t)min(miss[s], 255u);
        }
      }
      if (kMiss && nvar_out && lane == 0 && a_base == 0) nvar_out[i] = (uint16_t)min(nvar, 65535u);
      // the mismatch count read back from the log-likelihood (m = floor(-L / 3 + 1/4), gk_miss_of_log) is exact only
      // for rows of fewer than ~5000 factors: a longer row sends the gene to the exact search (wide records and
      // windows beyond 256 variants can produce such rows)
      if (kLog && miss8 && nvar >= 4096u && lane == 0) atomicOr(bound_flags, 1u);
    }
    __syncthreads();
    if (kIdx) {
      // index form: (dense index, mismatch count) of every product from the value table; four rows of a quad packed
      // into one 8-byte store of indices and one 4-byte store of counts; rows past the end hold index 0xFFFF / count 0
      const int n_r = (int)min<int64_t>(kTileRows, n_rows - row0);
      uint64_t key0 = kLutEmptyKey, key1 = kLutEmptyKey;   // the two most recent values of this thread's read
      uint64_t inf0 = kLutNoInfo, inf1 = kLutNoInfo;
      for (int idx = tid; idx < n_pass * kTileRows; idx += kCompatThreads) {
        const int al = idx / kTileRows, r = idx % kTileRows;
        const bool in = r < n_r;
        uint32_t id16 = kNoIndex, m = 0;
        if (in) {
          const uint64_t key = (uint64_t)__double_as_longlong(tile[al * kTileLd + r]);
          if (key != key0) {
            uint64_t info;
            if (key == key1) {
              info = inf1;
            } else {
              info = gk_lut_info(lut, key);
              if (info == kLutNoInfo) { gk_lut_insert(lut, key); atomicOr(bound_flags, 4u); }   // no log10 yet: this table is written again
            }
            key1 = key0; inf1 = inf0;
            key0 = key; inf0 = info;
          }
          if (inf0 != kLutNoInfo) {
            const uint32_t dense = (uint32_t)inf0;
            m = (uint32_t)(inf0 >> 32) & 0xFFu;
            if (dense >= kNoIndex) atomicOr(bound_flags, 2u);       // the value table outgrew 16-bit indices
            else id16 = dense;
            if (m == 255u) atomicOr(bound_flags, 1u);               // a count >= 100 / a product that left the normal range
          }
        }
        uint32_t x = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)m, 0xF9, 0xF, 0xF, true);      // lane + 1 of the quad
        uint32_t packed = m | (x << 8);
        x = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0xF9, 0xF, 0xF, true);               // lane + 2
        packed |= x << 16;
        x = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0xF9, 0xF, 0xF, true);               // lane + 3
        packed |= x << 24;
        const uint32_t i1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)id16, 0xF9, 0xF, 0xF, true);
        const uint32_t i2 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)i1, 0xF9, 0xF, 0xF, true);
        const uint32_t i3 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)i2, 0xF9, 0xF, 0xF, true);
        if ((r & 3) == 0) {
          const int64_t at = (int64_t)(a_base + al) * ldm + row0 + r;
          *reinterpret_cast<uint32_t*>(miss8 + at) = packed;
          *reinterpret_cast<uint2*>(lidx + at) = make_uint2(id16 | (i1 << 16), i2 | (i3 << 16));
        }
      }
    } else
    if (probs) {
      const int n_r = (int)min<int64_t>(kTileRows, n_rows - row0);
      if (kLog) {
        // Pass 1 of the way out: every product of the tile is replaced, in place, by its log10 from the value table.
        // A thread keeps to ONE read (its alleles share a handful of values: most lookups end in the two registers).
        uint64_t key0 = kLutEmptyKey, key1 = kLutEmptyKey;   // the two most recent values of this thread's read
        double val0 = 0.0, val1 = 0.0;
        uint32_t raise = 0;      // flag bits this thread wants raised: ONE atomic per wave below, not one per entry (the
                                 // first sample of a run meets nothing but new products: 150 M atomics on one word)
        // a thread keeps to one read and takes CONSECUTIVE alleles of it (neighbours in the index differ in a few variants
        // and mostly share their product: the two registers answer; alleles 32 apart, as the stride of the tile's
        // threads would give, rarely do)
        constexpr int kRowThreads = kCompatThreads / kTileRows;
        const int per_thread = (n_pass + kRowThreads - 1) / kRowThreads;
        const int r = tid % kTileRows, al0 = (tid / kTileRows) * per_thread;
        for (int al = al0; al < min(al0 + per_thread, n_pass); ++al) {
          if (r >= n_r) break;
          double* const cell = &tile[al * kTileLd + r];
          const uint64_t key = (uint64_t)__double_as_longlong(*cell);
          if (key != key0) {
            double val;
            if (key == key1) {
              val = val1;
            } else {
              bool found;
              val = gk_lut_lookup(lut, key, &found);
              if (!found) {
                // no log10 yet.  With a flag word the PRODUCT itself is stored in the entry's place -- a strictly positive
                // double, which no log10 of a probability is -- and bit 2 is raised: patch_pending puts the log10 there
                // once the host has defined it (bit 3: the product is +0.0, which cannot mark itself: write the table again)
                gk_lut_insert(lut, key);
                if (bound_flags) {
                  const bool marks = (int64_t)key > 0;
                  raise |= marks ? 4u : 12u;
                  if (marks) val = __longlong_as_double((long long)key);
                }
              }
            }
            key1 = key0; val1 = val0;
            key0 = key; val0 = val;
          }
          *cell = val0;
        }
        if (bound_flags) {
          const uint32_t pend = __ballot((raise & 4u) != 0) ? 4u : 0u, zero = __ballot((raise & 8u) != 0) ? 8u : 0u;
          if ((pend | zero) && lane == 0) atomicOr(bound_flags, pend | zero);
        }
        __syncthreads();
      }
      // Pass 2: a thread takes FOUR consecutive reads of one allele -- 32 bytes of the column-major table (four stores off
      // one address) and, on the product path, their four mismatch bytes as one word.  The mismatch count of
      // (read, allele) is read back from the log-likelihood: -L = 3 m + 0.000434 (n - m), so m = floor(-L / 3 + 1/4)
      // for any list shorter than ~5000 ids.  u8 table [allele][ldm]; rows past the end hold 0 (they add nothing to any
      // |a - b| sum).  A count >= 100 (the product is about to leave the normal range / underflow, L = -inf) raises the
      // flag that sends the gene to the exact search; NaN (log10 not defined yet) is rewritten by the next pass.
      constexpr int kQuads = kTileRows / 4;
      for (int it = tid; it < n_pass * kQuads; it += kCompatThreads) {
        const int al = it / kQuads, r0 = 4 * (it % kQuads);
        const double* const cell = &tile[al * kTileLd + r0];
        double v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = cell[j];
        if (r0 < n_r) {
          double* const dst = probs + (int64_t)(a_base + al) * n_rows + row0 + r0;
          if (r0 + 4 <= n_r) {
#pragma unroll
            for (int j = 0; j < 4; ++j) dst[j] = v[j];
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (r0 + j < n_r) dst[j] = v[j];
          }
        }
        if (kLog && miss8) {
          uint32_t packed = 0;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (r0 + j < n_r) {
              const double t = __builtin_fma(v[j], -1.0 / 3.0, 0.25);
              uint32_t m;
              if (t < 100.0) m = (uint32_t)(int)t;
              else { m = 255u; if (v[j] == v[j]) atomicOr(bound_flags, 1u); }
              packed |= m << (8 * j);
            }
          }
          *reinterpret_cast<uint32_t*>(miss8 + (int64_t)(a_base + al) * ldm + row0 + r0) = packed;
        }
      }
    }
    if (kLog && miss8 && tile_i == n_tiles - 1) {
      // rows of the mismatch table past the last tile, up to its stride (a multiple of 64 rows): zero, they add nothing
      // to any |a - b| sum of the bound
      const int64_t pad0 = n_tiles * kTileRows;
      const int n_pad = (int)((ldm - pad0) / 4);
      for (int it = tid; it < n_pass * n_pad; it += kCompatThreads) {
        const int al = it / n_pad, w = it % n_pad;
        *reinterpret_cast<uint32_t*>(miss8 + (int64_t)(a_base + al) * ldm + pad0 + 4 * w) = 0u;
      }
    }
    __syncthreads();
  }
}

// Entries of a log-likelihood table that still hold their PRODUCT (a strictly positive double: the compatibility kernel
// met it before the host had evaluated its log10, typing_mulit_allele.py:263) get the log10 from the value table, and
// their byte of the mismatch table is set from it; an entry whose value is still undefined stays and raises bit 2 again.
// One pass over the table (8 bytes per entry read, the few patched ones written) instead of the kernel that made it.
__global__ __launch_bounds__(kThreads) void patch_pending(double* __restrict__ L, int64_t n_rows, int n_allele,
                                                          LutView lut, uint8_t* __restrict__ miss8, int64_t ldm,
                                                          uint32_t* __restrict__ flags) {
  const int64_t n = n_rows * (int64_t)n_allele;
  const int64_t stride = (int64_t)gridDim.x * kThreads;
  uint64_t key0 = kLutEmptyKey;
  double val0 = 0.0;
  bool ok0 = false;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += stride) {
    const double v = L[i];
    if (!(v > 0.0)) continue;
    const uint64_t key = (uint64_t)__double_as_longlong(v);
    if (key != key0) {
      key0 = key;
      val0 = gk_lut_lookup(lut, key, &ok0);
    }
    if (!ok0) { atomicOr(flags, 4u); continue; }
    L[i] = val0;
    const uint32_t m = gk_miss_of_log(val0);
    if (m == 255u && val0 == val0) atomicOr(flags, 1u);
    const int64_t a = i / n_rows, r = i - a * n_rows;
    miss8[a * ldm + r] = (uint8_t)m;
  }
}

template <bool kLog>
int launch_compat(gk_ctx* ctx, gk_tab* tab, gk_dptr d_rows, int64_t n_rows, gk_dptr d_vflag, int vbeg, int vend,
                  gk_dptr d_mask, int words, int n_allele, double* out, uint8_t* miss, uint16_t* nvar, LutView view,
                  int keep_empty, uint8_t* miss8 = nullptr, int64_t ldm = 0, uint32_t* bound_flags = nullptr,
                  uint16_t* lidx = nullptr) {
  const double empty_p = keep_empty ? 0.999 : 1.0;
  int64_t want = ((n_rows + kTileRows - 1) / kTileRows + 7) / 8 * 8;      // a multiple of 8: see the tile order in the kernel
  const dim3 grid((unsigned)(want < 2048 ? (want < 8 ? 8 : want) : 2048)), block(kCompatThreads);
  const int64_t n_mask = (int64_t)(vend - vbeg) * words;
  uint32_t* mask_t = nullptr;
  GK_HIP(gk_pool_malloc(ctx, (void**)&mask_t, (size_t)std::max<int64_t>(n_mask, 1) * sizeof(uint32_t)));
  // the flag word starts at zero: cleared by the transposition when there is one (a fill of its own otherwise)
  if (n_mask > 0)
    GK_KERNEL(transpose_mask, dim3((unsigned)((n_mask + kThreads - 1) / kThreads)), dim3(kThreads), 0, ctx->stream,
              gk_ptr<uint32_t>(d_mask), vend - vbeg, words, mask_t, bound_flags);
  else if (bound_flags)
    GK_HIP(hipMemsetAsync(bound_flags, 0, sizeof(uint32_t), ctx->stream));
  for (int a_base = 0; a_base < n_allele; a_base += 64 * kMaxSlots) {
    const int slots = std::min(kMaxSlots, (n_allele - a_base + 63) / 64);
#define GK_COMPAT_GO(S, IDX)                                                                                        \
  GK_KERNEL((compat_kernel<kLog, S, !kLog, IDX>), grid, block, 0, ctx->stream, gk_ptr<int32_t>(d_rows), n_rows,     \
            tab->d_off, tab->d_ids, gk_ptr<uint8_t>(d_vflag), vbeg, vend, mask_t, words, n_allele, a_base, out,     \
            miss, nvar, view, empty_p, miss8, ldm, bound_flags, lidx)
#define GK_COMPAT_LAUNCH(S)                              \
  GK_PROF(ctx, "compat_kernel", {                        \
    if (kLog && lidx) GK_COMPAT_GO(S, (kLog && true));   \
    else GK_COMPAT_GO(S, false);                         \
  })
    switch (slots) {
      case 1: GK_COMPAT_LAUNCH(1); break;
      case 2: GK_COMPAT_LAUNCH(2); break;
      case 3: GK_COMPAT_LAUNCH(3); break;
      default: GK_COMPAT_LAUNCH(4); break;
    }
#undef GK_COMPAT_LAUNCH
#undef GK_COMPAT_GO
  }
  gk_pool_free(ctx, mask_t);   // stream-ordered reuse: the next user of the block runs after these launches
  GK_HIP(hipGetLastError());
  return GK_OK;
}

}  // namespace

extern "C" {

static int build_partition(gk_ctx* ctx, gk_tab* tab, int multiple) {
  gk_tab::GenePartition& part = tab->part[multiple ? 1 : 0];
  const int64_t n = tab->n_valid;
  const int n_bins = tab->idx ? tab->idx->n_gene : 256;
  const int64_t n_tiles = (n + 63) / 64;
  const size_t n_hist = (size_t)n_bins * n_tiles;
  uint32_t *hist = nullptr, *goff = nullptr;
  GK_HIP(gk_pool_malloc(ctx, (void**)&hist, (n_hist + 1) * sizeof(uint32_t)));
  GK_HIP(gk_pool_malloc(ctx, (void**)&goff, (size_t)(n_bins + 1) * sizeof(uint32_t)));
  GK_HIP(gk_pool_malloc(ctx, (void**)&part.d_rows, (size_t)(n + 1) * sizeof(int32_t)));
  part.owner = ctx;
  GK_HIP(hipMemsetAsync(hist, 0, n_hist * sizeof(uint32_t), ctx->stream));
  GK_PROF(ctx, "part_pass", GK_KERNEL(part_pass<false>, dim3(nblk(n)), dim3(kThreads), 0, ctx->stream,
                                               tab->d_pair_gene, tab->d_pair_nh, n, multiple, n_bins, n_tiles, hist,
                                               (int32_t*)nullptr));
  int rc = gk_scan_u32(ctx, hist, (int64_t)n_hist, hist + n_hist);
  if (rc) return rc;
  GK_PROF(ctx, "part_pass", GK_KERNEL(part_pass<true>, dim3(nblk(n)), dim3(kThreads), 0, ctx->stream,
                                               tab->d_pair_gene, tab->d_pair_nh, n, multiple, n_bins, n_tiles, hist,
                                               part.d_rows));
  GK_KERNEL(part_offsets, dim3((unsigned)(n_bins / 256 + 1)), dim3(256), 0, ctx->stream, hist, hist + n_hist,
                     n_bins, n_tiles, goff);
  GK_HIP(hipGetLastError());
  std::vector<uint32_t> host((size_t)n_bins + 1);
  GK_HIP(gk_fetch(ctx, host.data(), goff, host.size() * sizeof(uint32_t)));
  part.gene_off.assign(host.begin(), host.end());
  gk_pool_free(ctx, hist);
  gk_pool_free(ctx, goff);
  return GK_OK;
}

int gk_select_gene(gk_ctx* ctx, gk_tab* tab, int gene, int multiple, gk_dptr d_rows_out, int64_t* n_out) {
  gk_bind(ctx);
  GK_REQUIRE(ctx && tab && n_out, "null pointer");
  *n_out = 0;
  if (tab->n_valid == 0 || gene < 0) return GK_OK;
  gk_tab::GenePartition& part = tab->part[multiple ? 1 : 0];
  {
    std::lock_guard<std::mutex> lock(tab->part_mutex);
    if (part.gene_off.empty()) {
      int rc = build_partition(ctx, tab, multiple);
      if (rc) { part.gene_off.clear(); return rc; }
    }
  }
  if ((size_t)gene + 1 >= part.gene_off.size()) return GK_OK;
  const int64_t first = part.gene_off[gene], count = part.gene_off[gene + 1] - first;
  if (count > 0) {
    GK_REQUIRE(d_rows_out, "null output");
    GK_HIP(hipMemcpyAsync(gk_ptr<int32_t>(d_rows_out), part.d_rows + first, (size_t)count * sizeof(int32_t),
                          hipMemcpyDeviceToDevice, ctx->stream));
  }
  *n_out = count;
  return GK_OK;
}

int gk_select_nonempty(gk_ctx* ctx, gk_tab* tab, gk_dptr d_rows, int64_t n_rows, gk_dptr d_vflag, gk_dptr d_rows_out,
                       int64_t* n_out) {
  gk_bind(ctx);
  GK_REQUIRE(ctx && tab && n_out, "null pointer");
  if (n_rows == 0) { *n_out = 0; return GK_OK; }
  uint32_t* flag = nullptr;
  GK_HIP(gk_pool_malloc(ctx, (void**)&flag, (size_t)n_rows * sizeof(uint32_t)));
  GK_PROF(ctx, "flag_nonempty", GK_KERNEL(flag_nonempty, dim3(nblk(n_rows)), dim3(kThreads), 0, ctx->stream, gk_ptr<int32_t>(d_rows),
                     n_rows, tab->d_off, tab->d_ids, gk_ptr<uint8_t>(d_vflag), flag, (uint32_t*)nullptr, 0));
  int rc = gk_compact(ctx, flag, gk_ptr<int32_t>(d_rows), n_rows, gk_ptr<int32_t>(d_rows_out), n_out);
  gk_pool_free(ctx,flag);
  return rc;
}

int gk_variant_count_range(gk_ctx* ctx, gk_tab* tab, gk_dptr d_rows, int64_t n_rows, gk_dptr d_vflag, gk_dptr d_cnt,
                            int32_t vbeg, int32_t vend);

int gk_variant_count(gk_ctx* ctx, gk_tab* tab, gk_dptr d_rows, int64_t n_rows, gk_dptr d_vflag, gk_dptr d_cnt) {
  gk_bind(ctx);
  return gk_variant_count_range(ctx, tab, d_rows, n_rows, d_vflag, d_cnt, 0, 0);
}

int gk_variant_count_range(gk_ctx* ctx, gk_tab* tab, gk_dptr d_rows, int64_t n_rows, gk_dptr d_vflag, gk_dptr d_cnt,
                            int32_t vbeg, int32_t vend) {
  gk_bind(ctx);
  GK_REQUIRE(ctx && tab && vend >= vbeg, "bad arguments");
  int n_local = vend - vbeg;
  if ((size_t)n_local * 8 > 60 * 1024) n_local = 60 * 1024 / 8;   // LDS budget per workgroup
  const int64_t nv = (int64_t)tab->n_var + tab->n_novel;
  uint32_t* cnt = gk_ptr<uint32_t>(d_cnt);
  GK_HIP(hipMemsetAsync(cnt, 0, (size_t)(2 * nv) * sizeof(uint32_t), ctx->stream));
  if (n_rows) {
    unsigned blocks = nblk(16 * n_rows);   // 16 lanes per row
    if (blocks > 1024) blocks = 1024;      // grid-stride: many rows per workgroup before the LDS flush (measured optimum)
    GK_PROF(ctx, "count_ids",
            GK_KERNEL(count_ids, dim3(blocks), dim3(kThreads), (size_t)n_local * 8, ctx->stream,
                               gk_ptr<int32_t>(d_rows), n_rows, tab->d_off, tab->d_ids, gk_ptr<uint8_t>(d_vflag), cnt,
                               cnt + nv, vbeg, n_local));
  }
  GK_HIP(hipGetLastError());
  return GK_OK;
}

int gk_variant_correct(gk_ctx* ctx, gk_tab* tab, gk_dptr d_cnt, gk_dptr d_vflag) {
  gk_bind(ctx);
  GK_REQUIRE(ctx && tab, "null pointer");
  const int64_t nv = (int64_t)tab->n_var + tab->n_novel;
  uint32_t* cnt = gk_ptr<uint32_t>(d_cnt);
  if (nv)
    GK_PROF(ctx, "apply_correction", GK_KERNEL(apply_correction, dim3(nblk(nv)), dim3(kThreads), 0, ctx->stream, cnt, cnt + nv, nv,
                       gk_ptr<uint8_t>(d_vflag)));
  GK_HIP(hipGetLastError());
  return GK_OK;
}

namespace {
// gene >= 0: only the variants of that backbone count -- index ordinals [vbeg, vend) and novel variants whose key
// names it (tallies shared by all genes of a sample, gk_sample_prepare)
__global__ __launch_bounds__(kThreads) void flag_surviving(const uint32_t* cnt_pos, const uint32_t* cnt_neg,
                                                           const uint8_t* vflag, int64_t n, uint32_t* flag, int gene,
                                                           int vbeg, int vend, int n_index,
                                                           const uint64_t* __restrict__ novel_key) {
  const int64_t v = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (v >= n) return;
  if (gene >= 0) {
    const bool mine = v < n_index ? (v >= vbeg && v < vend) : (int)(novel_key[v - n_index] >> GK_KEY_REF_SHIFT) == gene;
    if (!mine) { flag[v] = 0; return; }
  }
  const uint8_t f = vflag[v];
  flag[v] = ((!(f & 1) && cnt_pos[v]) || (!(f & 2) && cnt_neg[v])) ? 1u : 0u;
}
__global__ __launch_bounds__(kThreads) void gather_surviving(const int32_t* ord, int64_t n, const uint32_t* cnt_pos,
                                                             const uint32_t* cnt_neg, const uint8_t* vflag,
                                                             uint32_t* out /*[2][n]*/) {
  const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (i >= n) return;
  const int32_t v = ord[i];
  const uint8_t f = vflag[v];
  out[i] = (f & 1) ? 0u : cnt_pos[v];
  out[n + i] = (f & 2) ? 0u : cnt_neg[v];
}
// the same with the number of survivors still on the device (the compaction has not been waited for): entries
// [0, min(*d_n, cap)) of out[2][cap]
__global__ __launch_bounds__(kThreads) void gather_surviving_queued(const int32_t* ord, const uint32_t* d_n, int64_t cap,
                                                                    const uint32_t* cnt_pos, const uint32_t* cnt_neg,
                                                                    const uint8_t* vflag, uint32_t* out /*[2][cap]*/) {
  const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (i >= cap || i >= (int64_t)*d_n) return;
  const int32_t v = ord[i];
  const uint8_t f = vflag[v];
  out[i] = (f & 1) ? 0u : cnt_pos[v];
  out[cap + i] = (f & 2) ? 0u : cnt_neg[v];
}
}  // namespace

/* Variants whose tally survives the drop flags: ordinals + (positive, negative) counts, compacted on
 * the device (feeds isHomozygous, typing_mulit_allele.py:807-857).  Host arrays must hold max_out. */
static int variant_surviving(gk_ctx* ctx, gk_tab* tab, gk_dptr d_cnt, gk_dptr d_vflag, int64_t max_out, int32_t* ord_out,
                             uint32_t* pos_out, uint32_t* neg_out, int64_t* n_out, int gene, int vbeg, int vend) {
  gk_bind(ctx);
  GK_REQUIRE(ctx && tab && ord_out && pos_out && neg_out && n_out, "null pointer");
  const int64_t nv = (int64_t)tab->n_var + tab->n_novel;
  *n_out = 0;
  if (!nv) return GK_OK;
  uint32_t* cnt = gk_ptr<uint32_t>(d_cnt);
  uint32_t *flag = nullptr, *vals = nullptr;
  int32_t* ord = nullptr;
  GK_HIP(gk_pool_malloc(ctx, (void**)&flag, (size_t)nv * sizeof(uint32_t)));
  GK_HIP(gk_pool_malloc(ctx, (void**)&ord, (size_t)nv * sizeof(int32_t)));
  GK_KERNEL(flag_surviving, dim3(nblk(nv)), dim3(kThreads), 0, ctx->stream, cnt, cnt + nv,
                     gk_ptr<uint8_t>(d_vflag), nv, flag, gene, vbeg, vend, tab->n_var, tab->d_novel_key);
  int64_t n = 0;
  int rc = gk_compact(ctx, flag, nullptr, nv, ord, &n);
  if (rc == GK_OK && n > max_out) {
    gk_set_error("surviving-variant buffer too small (%lld > %lld)", (long long)n, (long long)max_out);
    rc = GK_ERR_CAPACITY;
  }
  if (rc == GK_OK && n) {
    GK_HIP(gk_pool_malloc(ctx, (void**)&vals, (size_t)(2 * n) * sizeof(uint32_t)));
    GK_KERNEL(gather_surviving, dim3(nblk(n)), dim3(kThreads), 0, ctx->stream, ord, n, cnt, cnt + nv,
                       gk_ptr<uint8_t>(d_vflag), vals);
    GK_HIP(gk_fetch_queue(ctx, ord_out, ord, (size_t)n * sizeof(int32_t)));
    GK_HIP(gk_fetch_queue(ctx, pos_out, vals, (size_t)n * sizeof(uint32_t)));
    GK_HIP(gk_fetch_queue(ctx, neg_out, vals + n, (size_t)n * sizeof(uint32_t)));
    GK_HIP(gk_fetch_wait(ctx));
    gk_pool_free(ctx, vals);
  }
  gk_pool_free(ctx, flag);
  gk_pool_free(ctx, ord);
  if (rc == GK_OK) *n_out = n;
  return rc;
}

int gk_variant_surviving(gk_ctx* ctx, gk_tab* tab, gk_dptr d_cnt, gk_dptr d_vflag, int64_t max_out, int32_t* ord_out,
                         uint32_t* pos_out, uint32_t* neg_out, int64_t* n_out) {
  return variant_surviving(ctx, tab, d_cnt, d_vflag, max_out, ord_out, pos_out, neg_out, n_out, -1, 0, 0);
}

/* The same restricted to one backbone: its index variants [vbeg, vend) and the novel variants on it (for tallies
 * that cover every gene of the sample, gk_sample_prepare). */
int gk_variant_surviving_gene(gk_ctx* ctx, gk_tab* tab, gk_dptr d_cnt, gk_dptr d_vflag, int32_t gene, int32_t vbeg,
                              int32_t vend, int64_t max_out, int32_t* ord_out, uint32_t* pos_out, uint32_t* neg_out,
                              int64_t* n_out) {
  GK_REQUIRE(gene >= 0 && vend >= vbeg, "bad gene range");
  return variant_surviving(ctx, tab, d_cnt, d_vflag, max_out, ord_out, pos_out, neg_out, n_out, gene, vbeg, vend);
}

/* errorCorrection + removeEmptyReads (typing_mulit_allele.py:302-338, 274-281) for EVERY gene of a sample at once.
 * The variants of different backbones are disjoint, so one tally over all rows (grouped by backbone, NH == 1 unless
 * `multiple`), one pass of the thresholds and one compaction give what the per-gene calls give -- in ~8 launches and
 * one wait instead of six launches and a wait per gene.
 *   d_vflag  uint8  [n_var + n_novel]      zeroed here, then bit0 / bit1 = dropped from positive / negative lists
 *   d_cnt    uint32 [2][n_var + n_novel]   tallies of the uncorrected lists (masked by d_vflag = those of the corrected)
 *   d_rows   int32  [n_valid]              rows with a surviving id, grouped by backbone in row order
 *   gene_off_out int64 [n_gene + 1]        rows of gene g = d_rows[gene_off_out[g] .. gene_off_out[g + 1]) */
namespace {
struct Survivors {        // what gk_variant_surviving would fetch, asked for together with the preamble
  int64_t max_out = 0;
  int32_t* ord = nullptr;
  uint32_t *pos = nullptr, *neg = nullptr;
  int64_t* n_out = nullptr;
  uint64_t* novel_key = nullptr;      // [n_novel], optional
};
}  // namespace

static int sample_prepare(gk_ctx* ctx, gk_tab* tab, int32_t multiple, gk_dptr d_vflag, gk_dptr d_cnt, gk_dptr d_rows,
                          int64_t* gene_off_out, const Survivors* surv, bool keep_vflag = false, int rounds = 1);

int gk_sample_prepare(gk_ctx* ctx, gk_tab* tab, int32_t multiple, gk_dptr d_vflag, gk_dptr d_cnt, gk_dptr d_rows,
                      int64_t* gene_off_out) {
  return sample_prepare(ctx, tab, multiple, d_vflag, d_cnt, d_rows, gene_off_out, nullptr);
}

/* gk_sample_prepare + gk_variant_surviving (all backbones) + the sample's novel keys in ONE call with ONE wait after
 * the grouping by backbone is known: the preamble of a sample cost six waits, each of them behind whatever long kernel
 * of another sample holds the GPU at that moment (4 - 6 ms instead of 1.5 next to a search).  ord / pos / neg_out hold
 * max_out entries (max_out >= the sample's variants: index + novel; GK_ERR_CAPACITY otherwise), novel_key_out the
 * tabulation's novel variants (may be NULL). */
int gk_sample_prepare_all(gk_ctx* ctx, gk_tab* tab, int32_t multiple, gk_dptr d_vflag, gk_dptr d_cnt, gk_dptr d_rows,
                          int64_t* gene_off_out, int64_t max_out, int32_t* ord_out, uint32_t* pos_out, uint32_t* neg_out,
                          int64_t* n_out, uint64_t* novel_key_out) {
  GK_REQUIRE(ord_out && pos_out && neg_out && n_out && max_out >= 0, "null pointer");
  Survivors s;
  s.max_out = max_out; s.ord = ord_out; s.pos = pos_out; s.neg = neg_out; s.n_out = n_out; s.novel_key = novel_key_out;
  return sample_prepare(ctx, tab, multiple, d_vflag, d_cnt, d_rows, gene_off_out, &s);
}

/* The preamble of the EXON model of every gene (AlleleTypingExonFirst, typing_mulit_allele.py:640-664): d_vflag comes in
 * holding 3 for every variant outside the exons (removeIntronVariant 703-714: their ids are dropped from every list) and
 * the error correction runs TWICE on what is left (644-645, then once more inside the base class, 664), then the rows
 * without a surviving id are removed and the surviving tallies fetched -- as gk_sample_prepare_all does for the full model. */
int gk_sample_prepare_exon(gk_ctx* ctx, gk_tab* tab, int32_t multiple, gk_dptr d_vflag, gk_dptr d_cnt, gk_dptr d_rows,
                           int64_t* gene_off_out, int64_t max_out, int32_t* ord_out, uint32_t* pos_out, uint32_t* neg_out,
                           int64_t* n_out) {
  GK_REQUIRE(ord_out && pos_out && neg_out && n_out && max_out >= 0, "null pointer");
  Survivors s;
  s.max_out = max_out; s.ord = ord_out; s.pos = pos_out; s.neg = neg_out; s.n_out = n_out; s.novel_key = nullptr;
  return sample_prepare(ctx, tab, multiple, d_vflag, d_cnt, d_rows, gene_off_out, &s, true, 2);
}

static int sample_prepare(gk_ctx* ctx, gk_tab* tab, int32_t multiple, gk_dptr d_vflag, gk_dptr d_cnt, gk_dptr d_rows,
                          int64_t* gene_off_out, const Survivors* surv, bool keep_vflag, int rounds) {
  gk_bind(ctx);
  GK_REQUIRE(ctx && tab && tab->idx && d_vflag && d_cnt && d_rows && gene_off_out, "null pointer");
  if (surv) *surv->n_out = 0;
  const int n_gene = tab->idx->n_gene;
  const int64_t nv = (int64_t)tab->n_var + tab->n_novel;
  hipStream_t st = ctx->stream;
  for (int g = 0; g <= n_gene; ++g) gene_off_out[g] = 0;
  if (!keep_vflag) GK_HIP(hipMemsetAsync(gk_ptr<void>(d_vflag), 0, (size_t)std::max<int64_t>(nv, 1), st));
  GK_HIP(hipMemsetAsync(gk_ptr<void>(d_cnt), 0, (size_t)std::max<int64_t>(2 * nv, 1) * sizeof(uint32_t), st));
  auto novel_only = [&]() -> int {          // nothing to tally: no variant survives; the novel keys may still be wanted
    if (surv && surv->novel_key && tab->n_novel > 0)
      GK_HIP(gk_fetch(ctx, surv->novel_key, tab->d_novel_key, (size_t)tab->n_novel * sizeof(uint64_t)));
    return GK_OK;
  };
  if (tab->n_valid == 0) return novel_only();
  gk_tab::GenePartition& part = tab->part[multiple ? 1 : 0];
  {
    std::lock_guard<std::mutex> lock(tab->part_mutex);
    if (part.gene_off.empty()) {
      int rc = build_partition(ctx, tab, multiple);
      if (rc) { part.gene_off.clear(); return rc; }
    }
  }
  GK_REQUIRE((int)part.gene_off.size() >= n_gene + 1, "partition does not cover the genes");
  const int64_t n_rows = part.gene_off[n_gene];
  if (n_rows == 0) return novel_only();
  // workgroups: a share of ~2048 per gene in proportion to its rows, never two genes in one workgroup
  std::vector<int32_t> wg_gene;
  std::vector<int64_t> wg_row0, wg_row1, goff(part.gene_off.begin(), part.gene_off.begin() + n_gene + 1);
  const int64_t per_wg = std::max<int64_t>(64, (n_rows + 2047) / 2048);
  int max_span = 0;
  for (int g = 0; g < n_gene; ++g) {
    max_span = std::max(max_span, tab->idx->gene_vbeg[g + 1] - tab->idx->gene_vbeg[g]);
    for (int64_t r = goff[g]; r < goff[g + 1]; r += per_wg) {
      wg_gene.push_back(g);
      wg_row0.push_back(r);
      wg_row1.push_back(std::min(goff[g + 1], r + per_wg));
    }
  }
  const int max_local = std::min(max_span, 60 * 1024 / 8);   // LDS budget per workgroup
  const size_t n_wg = wg_gene.size();
  char* d_tab = nullptr;
  const size_t o_row0 = (n_wg * sizeof(int32_t) + 15) / 16 * 16, o_row1 = o_row0 + n_wg * sizeof(int64_t),
               o_goff = o_row1 + n_wg * sizeof(int64_t), tab_bytes = o_goff + (size_t)(n_gene + 1) * sizeof(int64_t);
  {
    std::vector<char> packed(tab_bytes);
    memcpy(packed.data(), wg_gene.data(), n_wg * sizeof(int32_t));
    memcpy(packed.data() + o_row0, wg_row0.data(), n_wg * sizeof(int64_t));
    memcpy(packed.data() + o_row1, wg_row1.data(), n_wg * sizeof(int64_t));
    memcpy(packed.data() + o_goff, goff.data(), (size_t)(n_gene + 1) * sizeof(int64_t));
    GK_HIP(gk_pool_malloc(ctx, (void**)&d_tab, tab_bytes));
    GK_HIP(gk_send(ctx, d_tab, packed.data(), tab_bytes));
  }
  uint32_t* cnt = gk_ptr<uint32_t>(d_cnt);
  uint8_t* vflag = gk_ptr<uint8_t>(d_vflag);
  // per pair: did the last tally count any of its ids / does one survive the correction (flag_pairs)
  uint8_t *maybe = nullptr, *alive = nullptr;
  GK_HIP(gk_pool_malloc(ctx, (void**)&maybe, (size_t)tab->n_valid));
  GK_HIP(gk_pool_malloc(ctx, (void**)&alive, (size_t)tab->n_valid));
  GK_HIP(hipMemsetAsync(maybe, 0, (size_t)tab->n_valid, st));      // pairs outside the partition: never asked for
  for (int round = 0; round < rounds; ++round) {      // the exon model corrects its lists twice (typing_mulit_allele.py:644-645, 664)
    if (round) GK_HIP(hipMemsetAsync(cnt, 0, (size_t)(2 * nv) * sizeof(uint32_t), st));
    GK_PROF(ctx, "count_ids_genes",
            GK_KERNEL(count_ids_genes, dim3((unsigned)n_wg), dim3(kThreads), (size_t)max_local * 8 + kThreads * sizeof(uint32_t), st, part.d_rows,
                      (const int32_t*)d_tab, (const int64_t*)(d_tab + o_row0), (const int64_t*)(d_tab + o_row1),
                      tab->idx->d_gene_vbeg, max_local, tab->d_off, tab->d_ids, vflag, cnt, cnt + nv,
                      round == rounds - 1 ? maybe : (uint8_t*)nullptr));
    GK_PROF(ctx, "apply_correction", GK_KERNEL(apply_correction, dim3(nblk(nv)), dim3(kThreads), 0, st, cnt, cnt + nv, nv, vflag));
  }
  // rows with a surviving id, compacted in place of the grouping (stable: the groups stay contiguous and ordered)
  uint32_t *flag = nullptr, *kept = nullptr;
  GK_HIP(gk_pool_malloc(ctx, (void**)&flag, (size_t)n_rows * sizeof(uint32_t)));
  GK_HIP(gk_pool_malloc(ctx, (void**)&kept, (size_t)n_gene * sizeof(uint32_t)));
  GK_PROF(ctx, "flag_pairs", GK_KERNEL(flag_pairs, dim3(nblk(4 * (int64_t)tab->n_valid)), dim3(kThreads), 0, st, (int64_t)tab->n_valid,
                                   tab->d_off, tab->d_ids, vflag, maybe, alive));
  GK_PROF(ctx, "gather_pair_flags", GK_KERNEL(gather_pair_flags, dim3(nblk(std::max<int64_t>(n_rows, n_gene))), dim3(kThreads), 0, st,
                                          part.d_rows, n_rows, alive, flag, kept, n_gene));
  GK_PROF(ctx, "count_flags_per_gene", GK_KERNEL(count_flags_per_gene, dim3((unsigned)n_gene, kFlagSlices), dim3(kThreads), 0, st, flag,
                                      (const int64_t*)(d_tab + o_goff), kept));
  GK_HIP(hipGetLastError());
  // everything below is queued, then ONE wait: the compaction of the rows, the rows kept per gene, and -- when asked for --
  // the surviving tallies of every variant and the novel keys
  std::vector<void*> temps{flag, kept, d_tab, alive, maybe};
  auto done = [&](int code) {
    for (void* t : temps) gk_pool_free(ctx, t);
    return code;
  };
  uint32_t* d_total = nullptr;
  int rc = gk_compact_enqueue(ctx, flag, part.d_rows, n_rows, gk_ptr<int32_t>(d_rows), &d_total, temps);
  if (rc) return done(rc);
  uint32_t total = 0, n_surv = 0;
  std::vector<uint32_t> host((size_t)n_gene);
  bool queued = gk_fetch_queue(ctx, &total, d_total, sizeof(uint32_t)) == hipSuccess &&
                gk_fetch_queue(ctx, host.data(), kept, (size_t)n_gene * sizeof(uint32_t)) == hipSuccess;
  if (queued && surv && nv > 0) {
    const int64_t cap = std::min<int64_t>(nv, surv->max_out);
    uint32_t *sflag = nullptr, *vals = nullptr, *d_n = nullptr;
    int32_t* ord = nullptr;
    auto take = [&](void** p, size_t bytes) {
      if (gk_pool_malloc(ctx, p, bytes) != hipSuccess) return false;
      temps.push_back(*p);
      return true;
    };
    if (!take((void**)&sflag, (size_t)nv * sizeof(uint32_t)) || !take((void**)&ord, (size_t)nv * sizeof(int32_t)) ||
        !take((void**)&vals, (size_t)std::max<int64_t>(2 * cap, 1) * sizeof(uint32_t))) {
      gk_fetch_cancel(ctx);
      gk_set_error("out of device memory for the surviving tallies");
      return done(GK_ERR_HIP);
    }
    GK_KERNEL(flag_surviving, dim3(nblk(nv)), dim3(kThreads), 0, st, cnt, cnt + nv, vflag, nv, sflag, -1, 0, 0, tab->n_var,
              tab->d_novel_key);
    rc = gk_compact_enqueue(ctx, sflag, nullptr, nv, ord, &d_n, temps);
    if (rc) { gk_fetch_cancel(ctx); return done(rc); }
    if (cap > 0) {
      GK_KERNEL(gather_surviving_queued, dim3(nblk(cap)), dim3(kThreads), 0, st, ord, d_n, cap, cnt, cnt + nv, vflag, vals);
      queued = gk_fetch_queue(ctx, surv->ord, ord, (size_t)cap * sizeof(int32_t)) == hipSuccess &&
               gk_fetch_queue(ctx, surv->pos, vals, (size_t)cap * sizeof(uint32_t)) == hipSuccess &&
               gk_fetch_queue(ctx, surv->neg, vals + cap, (size_t)cap * sizeof(uint32_t)) == hipSuccess;
    }
    queued = queued && gk_fetch_queue(ctx, &n_surv, d_n, sizeof(uint32_t)) == hipSuccess;
    if (queued && surv->novel_key && tab->n_novel > 0)
      queued = gk_fetch_queue(ctx, surv->novel_key, tab->d_novel_key, (size_t)tab->n_novel * sizeof(uint64_t)) == hipSuccess;
  }
  if (!queued || gk_fetch_wait(ctx) != hipSuccess) {
    gk_fetch_cancel(ctx);
    gk_set_error("sample preamble: %s", hipGetErrorString(hipGetLastError()));
    return done(GK_ERR_HIP);
  }
  int64_t run = 0;
  for (int g = 0; g < n_gene; ++g) { gene_off_out[g] = run; run += host[g]; }
  gene_off_out[n_gene] = run;
  if (run != (int64_t)total) { gk_set_error("non-empty rows per gene do not add up"); return done(GK_ERR_ASSERT); }
  if (surv) {
    if ((int64_t)n_surv > surv->max_out) {
      gk_set_error("surviving-variant buffer too small (%lld > %lld)", (long long)n_surv, (long long)surv->max_out);
      return done(GK_ERR_CAPACITY);
    }
    *surv->n_out = n_surv;
  }
  return done(GK_OK);
}

int gk_compat(gk_ctx* ctx, gk_tab* tab, gk_dptr d_rows, int64_t n_rows, gk_dptr d_vflag, int32_t vbeg, int32_t vend,
              gk_dptr d_mask, int32_t words, int32_t n_allele, int32_t keep_empty, gk_dptr d_probs, gk_dptr d_miss,
              gk_dptr d_nvar) {
  gk_bind(ctx);
  GK_REQUIRE(ctx && tab, "null pointer");
  GK_REQUIRE(words >= 1 && n_allele >= 0 && n_allele <= words * 32 && vend >= vbeg, "bad mask geometry");
  if (n_rows == 0 || n_allele == 0) return GK_OK;
  return launch_compat<false>(ctx, tab, d_rows, n_rows, d_vflag, vbeg, vend, d_mask, words, n_allele,
                              gk_ptr<double>(d_probs), gk_ptr<uint8_t>(d_miss), gk_ptr<uint16_t>(d_nvar), LutView{}, keep_empty);
}

int gk_compat_log(gk_ctx* ctx, gk_tab* tab, gk_dptr d_rows, int64_t n_rows, gk_dptr d_vflag, int32_t vbeg,
                  int32_t vend, gk_dptr d_mask, int32_t words, int32_t n_allele, int32_t keep_empty, gk_lut* lut,
                  gk_dptr d_log) {
  gk_bind(ctx);
  GK_REQUIRE(ctx && tab && lut, "null pointer");   // the table may belong to another context of the same device
  GK_REQUIRE(words >= 1 && n_allele >= 0 && n_allele <= words * 32 && vend >= vbeg, "bad mask geometry");
  if (n_rows == 0 || n_allele == 0) return GK_OK;
  GK_REQUIRE(d_log, "null output");
  return launch_compat<true>(ctx, tab, d_rows, n_rows, d_vflag, vbeg, vend, d_mask, words, n_allele,
                             gk_ptr<double>(d_log), nullptr, nullptr, gk_lut_view(lut), keep_empty);
}

int gk_compat_log_miss(gk_ctx* ctx, gk_tab* tab, gk_dptr d_rows, int64_t n_rows, gk_dptr d_vflag, int32_t vbeg,
                       int32_t vend, gk_dptr d_mask, int32_t words, int32_t n_allele, int32_t keep_empty, gk_lut* lut,
                       gk_dptr d_log, gk_dptr d_miss8, int64_t ldm, gk_dptr d_flags) {
  gk_bind(ctx);
  GK_REQUIRE(ctx && tab && lut, "null pointer");
  GK_REQUIRE(words >= 1 && n_allele >= 0 && n_allele <= words * 32 && vend >= vbeg, "bad mask geometry");
  if (n_rows == 0 || n_allele == 0) return GK_OK;
  GK_REQUIRE(d_log && d_miss8 && d_flags, "null output");
  GK_REQUIRE(ldm >= n_rows && ldm % 64 == 0, "mismatch table stride must be a multiple of 64 rows");
  // rows past the end of every column are zero (the workgroup of the last tile writes them), and so is the flag word
  // before the kernel raises bits in it (launch_compat)
  return launch_compat<true>(ctx, tab, d_rows, n_rows, d_vflag, vbeg, vend, d_mask, words, n_allele,
                             gk_ptr<double>(d_log), nullptr, nullptr, gk_lut_view(lut), keep_empty,
                             gk_ptr<uint8_t>(d_miss8), ldm, gk_ptr<uint32_t>(d_flags));
}

/* After gk_compat_log_miss raised bit 2 of *d_flags (and not bit 3) and the value table has been resolved: the entries
 * that hold their product instead of its log10 are patched in place, with their mismatch bytes (d_miss8 then needs its
 * column sums again: gk_miss_colsum).  *d_flags is cleared first; bit 2 comes back if some value is still undefined,
 * bit 0 as for gk_compat_log_miss (typing_mulit_allele.py:263). */
int gk_compat_patch(gk_ctx* ctx, gk_lut* lut, gk_dptr d_log, int64_t n_rows, int32_t n_allele, gk_dptr d_miss8, int64_t ldm,
                    gk_dptr d_flags) {
  gk_bind(ctx);
  GK_REQUIRE(ctx && lut && d_log && d_miss8 && d_flags && n_rows >= 0 && n_allele >= 0 && ldm >= n_rows, "bad patch arguments");
  GK_HIP(hipMemsetAsync(gk_ptr<void>(d_flags), 0, sizeof(uint32_t), ctx->stream));
  if (n_rows == 0 || n_allele == 0) return GK_OK;
  const int64_t n = n_rows * (int64_t)n_allele;
  const int64_t want = (n + kThreads - 1) / kThreads;
  GK_PROF(ctx, "patch_pending",
          GK_KERNEL(patch_pending, dim3((unsigned)(want < 4096 ? want : 4096)), dim3(kThreads), 0, ctx->stream,
                    gk_ptr<double>(d_log), n_rows, n_allele, gk_lut_view(lut), gk_ptr<uint8_t>(d_miss8), ldm,
                    gk_ptr<uint32_t>(d_flags)));
  GK_HIP(hipGetLastError());
  return GK_OK;
}

/* The index form of gk_compat_log_miss: d_lidx uint16 [n_allele][ldm] receives, per (allele, read), the dense index of
 * the log-likelihood in the value table (gk_lut: value = vals[index]; 0xFFFF while the log10 of a product is not defined
 * yet -- resolve and call again, as for gk_compat_log), d_miss8 the mismatch counts.  *d_flags: bit 0 as for
 * gk_compat_log_miss, bit 1 = the value table holds more than 65535 values (use the float64 form for this gene). */
int gk_compat_index(gk_ctx* ctx, gk_tab* tab, gk_dptr d_rows, int64_t n_rows, gk_dptr d_vflag, int32_t vbeg, int32_t vend,
                    gk_dptr d_mask, int32_t words, int32_t n_allele, int32_t keep_empty, gk_lut* lut, gk_dptr d_lidx,
                    gk_dptr d_miss8, int64_t ldm, gk_dptr d_flags) {
  gk_bind(ctx);
  GK_REQUIRE(ctx && tab && lut, "null pointer");
  GK_REQUIRE(words >= 1 && n_allele >= 0 && n_allele <= words * 32 && vend >= vbeg, "bad mask geometry");
  if (n_rows == 0 || n_allele == 0) return GK_OK;
  GK_REQUIRE(d_lidx && d_miss8 && d_flags, "null output");
  GK_REQUIRE(ldm >= n_rows && ldm % 64 == 0, "table stride must be a multiple of 64 rows");
  return launch_compat<true>(ctx, tab, d_rows, n_rows, d_vflag, vbeg, vend, d_mask, words, n_allele, nullptr, nullptr,
                             nullptr, gk_lut_view(lut), keep_empty, gk_ptr<uint8_t>(d_miss8), ldm,
                             gk_ptr<uint32_t>(d_flags), gk_ptr<uint16_t>(d_lidx));
}

}  // extern "C"

// EM strategy ("report" / "em"): typing_em.py:68-188.
//
//   candidate sets   getCandidateAllelePerRead 68-87 + getMostFreqAllele 90-104 as bit-set algebra:
//                    mate set  = AND of the positive variants' allele rows, minus OR of the negative rows
//                                (empty when the mate has no positive variant);
//                    pair set  = L & R when that is non-empty (alleles named twice), else L | R.
//                    em_sets_groups: a GROUP of G = 4 / 8 / 16 lanes per pair, lane w = word w of both mate sets.  The
//                    pair's ids (its four lists lie back to back in the CSR) are read G at a time, coalesced, and handed
//                    round the group; the gene's bit rows [variant][word] are staged in LDS once per workgroup, so an id
//                    costs one LDS read and three bit operations per lane; "named twice?" is a ballot over the group.
//   distinct sets    the EM only needs the DISTINCT sets and their multiplicities.  The set is still in registers when
//                    its 64-bit hash is formed (a sum of per-word mixes over the group); the workgroup counts hashes in a
//                    small LDS table and adds its totals to the gene's table in HBM when it is done (one atomic per
//                    distinct hash and workgroup, not per pair).  em_sets_verify then compares every pair's words with the
//                    words of the first pair of its hash: equal hashes of different sets raise a flag and the sets are
//                    hashed again with another seed -- the classes are exact, never "probably" so.
//   gk_em_run / gk_sample_em   hisatEMnp 107-188 on the distinct sets with multiplicities.  The reference builds a dense
//                    0/1 read x allele float matrix and sweeps it 3-4 times per iteration; here a workgroup per gene runs
//                    the whole SQUAREM loop on two sparse forms of the sets (members of a set / sets of an allele), 16
//                    lanes per set or allele, the abundances in LDS.  Sums are evaluated in a fixed order, so results are
//                    run-to-run deterministic; they agree with numpy's row-sequential sums to rounding (tolerance 1e-5
//                    relative per BASELINE.json north_star).
#include <algorithm>
#include <type_traits>
#include <vector>

#include "gk_common.h"

namespace {

constexpr int kMaxWords = 16;    // up to 512 alleles per gene
constexpr int kMaxAllele = kMaxWords * 32;

// ------------------------------------------------------------------------------------------------ candidate sets
constexpr int kSetThreads = 1024;        // 16 waves: one workgroup per CU (the bit rows of a gene fill most of its LDS)
constexpr int kLocalSlots = 2048;        // the workgroup's own table of (hash, pairs, first pair)
constexpr int kLocalProbes = 8;
constexpr uint32_t kTableProbes = 1u << 12;
constexpr size_t kLocalBytes = (size_t)kLocalSlots * (8 + 4 + 4);
constexpr size_t kMaskLdsMax = 100 * 1024;    // bit rows of a gene staged in LDS up to this size, read from HBM / L2 beyond

struct EmSetsJob {     // one gene of a launch (blockIdx.y)
  const int32_t* rows;
  int64_t n_rows;
  const uint32_t* mask;            // [n_span][words]
  int32_t vbeg, n_span, words, mask_in_lds;
  uint32_t* sets;                  // [n_rows][words]
  unsigned long long* tag;         // the gene's table of hashes (nullptr: sets only), 0 = free
  uint32_t* cnt;
  uint32_t* row;                   // a pair of the hash (the one whose workgroup claimed the slot): what the others are compared with
  uint32_t slot_mask, seed;
  uint32_t* flags;                 // bit 0: table full; bit 1: two different sets with one hash
  uint32_t* out_sets;              // emit: distinct sets, multiplicities, their number
  uint32_t* out_count;
  uint32_t* n_out;
  uint32_t n_blocks;               // workgroups of the launch that work on this gene
  uint32_t max_out;
};
static_assert(sizeof(EmSetsJob) % 8 == 0, "job table layout");

__device__ inline uint64_t mix_word(uint32_t w, int lane, uint32_t seed) {
  uint64_t x = (((uint64_t)(uint32_t)(lane + 1 + (int)seed * 17) << 32) | w) * 0x9E3779B97F4A7C15ull;
  x ^= x >> 32;
  x *= 0xD6E8FEB86659FD93ull;
  x ^= x >> 29;
  return x;
}
__device__ inline uint64_t finish_hash(uint64_t s) {
  s ^= s >> 31;
  s *= 0xBF58476D1CE4E5B9ull;
  s ^= s >> 29;
  return s | 1ull;      // 0 marks a free slot
}

template <int G>
__device__ inline uint64_t group_sum(uint64_t v) {
  uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
#pragma unroll
  for (int d = G / 2; d >= 1; d >>= 1) {
    const uint32_t lo2 = __shfl_xor(lo, d, G), hi2 = __shfl_xor(hi, d, G);
    const uint64_t s = (((uint64_t)hi << 32) | lo) + (((uint64_t)hi2 << 32) | lo2);
    lo = (uint32_t)s;
    hi = (uint32_t)(s >> 32);
  }
  return ((uint64_t)hi << 32) | lo;
}

// n pairs of hash h, the first of them at position first_row, into the gene's table
__device__ inline void table_add(const EmSetsJob& J, uint64_t h, uint32_t n, uint32_t first_row) {
  uint32_t s = (uint32_t)h & J.slot_mask;
  const uint32_t limit = min(J.slot_mask, kTableProbes);
  for (uint32_t p = 0; p <= limit; ++p) {
    unsigned long long old = __hip_atomic_load(&J.tag[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old == 0ull) old = atomicCAS(&J.tag[s], 0ull, (unsigned long long)h);
    if (old == 0ull || old == h) {
      atomicAdd(&J.cnt[s], n);
      if (old == 0ull) J.row[s] = first_row;      // whoever claims the slot names the pair the others are compared with
      return;
    }
    s = (s + 1) & J.slot_mask;
  }
  atomicOr(J.flags, 1u);
}

struct LocalTable {
  unsigned long long* tag;
  uint32_t* cnt;
  uint32_t* row;
  __device__ void clear(int tid) {
    for (int e = tid; e < kLocalSlots; e += kSetThreads) { tag[e] = 0ull; cnt[e] = 0u; row[e] = 0xFFFFFFFFu; }
  }
  __device__ void add(const EmSetsJob& J, uint64_t h, uint32_t i) {
    uint32_t s = (uint32_t)(h >> 40) & (kLocalSlots - 1);
    for (int p = 0; p < kLocalProbes; ++p) {
      unsigned long long old = tag[s];
      if (old == 0ull) old = atomicCAS(&tag[s], 0ull, (unsigned long long)h);
      if (old == 0ull || old == h) {
        atomicAdd(&cnt[s], 1u);
        if (old == 0ull) row[s] = i;
        return;
      }
      s = (s + 1) & (kLocalSlots - 1);
    }
    table_add(J, h, 1u, i);      // the workgroup's table is crowded: straight to the gene's
  }
  __device__ void flush(const EmSetsJob& J, int tid) {
    for (int e = tid; e < kLocalSlots; e += kSetThreads)
      if (tag[e] != 0ull) table_add(J, tag[e], cnt[e], row[e]);
  }
};

// the rows of job J that this workgroup takes: their candidate sets written, their hashes counted.  A group of G lanes
// per pair, TWO words of both mate sets per lane (a 64-bit LDS read per id and lane: the LDS pipe -- the hand-round of the
// ids and the row reads -- is what bounds this kernel, and a wave carries twice the pairs this way).  `M`: the bit rows,
// `stride` words per variant (even in LDS, so that a lane's pair of words is one aligned 8-byte read).
template <int G, bool kLds>
__device__ inline void sets_of_rows(const EmSetsJob& J, const uint32_t* __restrict__ off, const uint32_t* __restrict__ ids,
                                    const uint32_t* M, int stride, LocalTable& lt) {
  constexpr int kGroups = kSetThreads / G;
  const int tid = threadIdx.x, l = tid & (G - 1), grp = tid / G, wl = tid & 63;
  const int words = J.words;
  const uint32_t vbeg = (uint32_t)J.vbeg, n_span = (uint32_t)J.n_span;
  const bool lo_lane = 2 * l < words, hi_lane = 2 * l + 1 < words;
  const bool hashing = J.tag != nullptr;
  // The loads of a pair depend on each other (position -> list offsets -> ids) and the kernel runs at four waves per SIMD
  // (the bit rows take most of the LDS): the offsets of the group's NEXT pair are requested before this pair's lists are
  // walked, and a lane takes FOUR ids per request (4 G ids of a list in flight at once instead of G).
  auto offsets_of = [&](int64_t i, uint32_t (&o)[5]) {
    o[0] = o[1] = o[2] = o[3] = o[4] = 0u;
    if (i < J.n_rows) {      // list order in the CSR: lpv, rpv, lnv, rnv -- back to back
      const uint32_t* p = off + 4 * (int64_t)J.rows[i];
      const uint4 q = *reinterpret_cast<const uint4*>(p);
      o[0] = q.x; o[1] = q.y; o[2] = q.z; o[3] = q.w; o[4] = p[4];
    }
  };
  const int64_t step = (int64_t)J.n_blocks * kGroups;
  uint32_t nxt[5];
  offsets_of((int64_t)blockIdx.x * kGroups + grp, nxt);
  for (int64_t base = (int64_t)blockIdx.x * kGroups; base < J.n_rows; base += step) {
    const int64_t i = base + grp;
    const bool live = i < J.n_rows;
    const uint32_t o0 = nxt[0], o1 = nxt[1], o2 = nxt[2], o3 = nxt[3], o4 = nxt[4];
    offsets_of(i + step, nxt);
    // a mate without positives names nobody; words past the gene's last one stay zero
    uint2 side0 = make_uint2((lo_lane && o1 > o0) ? 0xFFFFFFFFu : 0u, (hi_lane && o1 > o0) ? 0xFFFFFFFFu : 0u);
    uint2 side1 = make_uint2((lo_lane && o2 > o1) ? 0xFFFFFFFFu : 0u, (hi_lane && o2 > o1) ? 0xFFFFFFFFu : 0u);
    // one list at a time (its sign and its mate are then fixed): the group reads 4 G ids of the list and hands them round.
    // With the bit rows in LDS a step is branch-free: two rows stand behind the gene's own -- row n_span all zeros, row
    // n_span + 1 all ones -- every id outside the gene (a novel variant, the padding past a list's end) lands on the zero
    // row through ONE v_min_u32, and the padding of a POSITIVE list (spelled 0xFFFFFFFF: bit 31 set, which no ordinal has)
    // steps on to the row of ones (AND with ones: nothing happens): no test per step, and the lanes of a group whose
    // list has ended run the padded steps with the others
    // (a break per step cost ~27 scalar instructions of exec-mask bookkeeping per id: 1.08e9 per configs[2] launch).
    auto walk = [&](uint32_t begin, uint32_t end, uint2& side, bool negative) {
      for (uint32_t k0 = begin; k0 < end; k0 += 4 * G) {
        uint32_t idv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const uint32_t k = k0 + 4u * (uint32_t)l + (uint32_t)q;
          idv[q] = k < end ? ids[k] : 0xFFFFFFFFu;      // past the end of the list (ordinals stay below 2^31)
        }
#pragma unroll
        for (int j = 0; j < 4 * G; ++j) {
          const uint32_t v = (uint32_t)__shfl(idv[j & 3], j >> 2, G);
          const uint32_t rel = v - vbeg;
          uint2 m = make_uint2(0u, 0u);                      // a variant outside the index (novel) has no allele
          if (kLds) {
            uint32_t at = min(rel, n_span);                 // outside the gene (rel >= n_span): the zero row
            if (!negative) at += v >> 31;                   // ... but the padding of a positive list: the row of ones behind it
            m = *reinterpret_cast<const uint2*>(M + at * (uint32_t)stride + 2u * (uint32_t)l);
          } else {
            if (k0 + (uint32_t)j >= end) break;            // uniform within the group
            if (rel < n_span && lo_lane) {
              const uint32_t* at = M + rel * (uint32_t)stride + 2u * (uint32_t)l;
              m.x = at[0];
              m.y = hi_lane ? at[1] : 0u;
            }
          }
          // positives intersect (a novel one leaves nobody); negatives of the index subtract (a novel one: nothing)
          if (negative) { side.x &= ~m.x; side.y &= ~m.y; }
          else { side.x &= m.x; side.y &= m.y; }
        }
      }
    };
    walk(o0, o1, side0, false);
    walk(o1, o2, side1, false);
    walk(o2, o3, side0, true);
    walk(o3, o4, side1, true);
    const uint2 twice = make_uint2(side0.x & side1.x, side0.y & side1.y);
    const unsigned long long named = __ballot((twice.x | twice.y) != 0u);
    const bool both = ((named >> (wl & ~(G - 1))) & ((1ull << G) - 1ull)) != 0ull;
    const uint2 w = both ? twice : make_uint2(side0.x | side1.x, side0.y | side1.y);
    if (live && lo_lane) J.sets[i * words + 2 * l] = w.x;
    if (live && hi_lane) J.sets[i * words + 2 * l + 1] = w.y;
    if (hashing) {
      const uint64_t mine = (lo_lane ? mix_word(w.x, 2 * l, J.seed) : 0ull) + (hi_lane ? mix_word(w.y, 2 * l + 1, J.seed) : 0ull);
      const uint64_t h = finish_hash(group_sum<G>(mine));
      if (live && l == 0) lt.add(J, h, (uint32_t)i);
    }
  }
}

template <int G>
__global__ __launch_bounds__(kSetThreads) void em_sets_groups(const EmSetsJob* __restrict__ jobs, const uint32_t* __restrict__ off,
                                                              const uint32_t* __restrict__ ids) {
  extern __shared__ __align__(16) unsigned char em_lds[];
  const EmSetsJob J = jobs[blockIdx.y];
  if (blockIdx.x >= J.n_blocks) return;
  LocalTable lt{(unsigned long long*)em_lds, (uint32_t*)(em_lds + (size_t)kLocalSlots * 8),
                (uint32_t*)(em_lds + (size_t)kLocalSlots * 12)};
  uint32_t* lmask = (uint32_t*)(em_lds + kLocalBytes);
  const int tid = threadIdx.x;
  const int stride = (J.words + 1) & ~1;      // rows of an even number of words in LDS
  if (J.tag) lt.clear(tid);
  if (J.mask_in_lds) {
    for (int e = tid; e < J.n_span * stride; e += kSetThreads) {
      const int v = e / stride, w = e - v * stride;
      lmask[e] = w < J.words ? J.mask[v * J.words + w] : 0u;
    }
    // the two rows behind the gene's own: all zeros (anything outside the gene), all ones (padding of a positive list);
    // a lane whose words lie beyond the row reads on into these or the following bytes: its sets are zero from the start
    // and stay so under AND and AND-NOT alike
    for (int e = tid; e < 2 * stride + 2 * kMaxWords; e += kSetThreads)
      lmask[J.n_span * stride + e] = (e >= stride && e < 2 * stride) ? 0xFFFFFFFFu : 0u;
  }
  __syncthreads();
  if (J.mask_in_lds) sets_of_rows<G, true>(J, off, ids, lmask, stride, lt);      // two copies of the loop: LDS reads / global reads
  else sets_of_rows<G, false>(J, off, ids, J.mask, J.words, lt);
  __syncthreads();
  if (J.tag) lt.flush(J, tid);
}

// the hashes of sets that are in memory already (gk_em_distinct, or a second seed)
template <int G>
__global__ __launch_bounds__(kSetThreads) void em_sets_hash(const EmSetsJob* __restrict__ jobs) {
  extern __shared__ __align__(16) unsigned char em_lds[];
  const EmSetsJob J = jobs[blockIdx.y];
  if (blockIdx.x >= J.n_blocks) return;
  LocalTable lt{(unsigned long long*)em_lds, (uint32_t*)(em_lds + (size_t)kLocalSlots * 8),
                (uint32_t*)(em_lds + (size_t)kLocalSlots * 12)};
  constexpr int kGroups = kSetThreads / G;
  const int tid = threadIdx.x, l = tid & (G - 1), grp = tid / G;
  lt.clear(tid);
  __syncthreads();
  for (int64_t base = (int64_t)blockIdx.x * kGroups; base < J.n_rows; base += (int64_t)J.n_blocks * kGroups) {
    const int64_t i = base + grp;
    const bool live = i < J.n_rows, word_lane = l < J.words;
    const uint32_t w = live && word_lane ? J.sets[i * J.words + l] : 0u;
    const uint64_t h = finish_hash(group_sum<G>(word_lane ? mix_word(w, l, J.seed) : 0ull));
    if (live && l == 0) lt.add(J, h, (uint32_t)i);
  }
  __syncthreads();
  lt.flush(J, tid);
}

// every pair's words against the words of the first pair of its hash
template <int G>
__global__ __launch_bounds__(kSetThreads) void em_sets_verify(const EmSetsJob* __restrict__ jobs) {
  const EmSetsJob J = jobs[blockIdx.y];
  if (blockIdx.x >= J.n_blocks) return;
  constexpr int kGroups = kSetThreads / G;
  const int tid = threadIdx.x, l = tid & (G - 1), grp = tid / G;
  const uint32_t limit = min(J.slot_mask, kTableProbes);
  for (int64_t base = (int64_t)blockIdx.x * kGroups; base < J.n_rows; base += (int64_t)J.n_blocks * kGroups) {
    const int64_t i = base + grp;
    const bool live = i < J.n_rows, word_lane = l < J.words;
    const uint32_t w = live && word_lane ? J.sets[i * J.words + l] : 0u;
    const uint64_t h = finish_hash(group_sum<G>(word_lane ? mix_word(w, l, J.seed) : 0ull));
    uint32_t rep = 0xFFFFFFFFu;
    if (live && l == 0) {
      uint32_t s = (uint32_t)h & J.slot_mask;
      for (uint32_t p = 0; p <= limit; ++p) {
        const unsigned long long t = J.tag[s];
        if (t == h) { rep = J.row[s]; break; }
        if (t == 0ull) break;
        s = (s + 1) & J.slot_mask;
      }
    }
    rep = __shfl(rep, 0, G);
    if (live && word_lane && rep != 0xFFFFFFFFu && rep != (uint32_t)i && J.sets[(int64_t)rep * J.words + l] != w)
      atomicOr(J.flags, 2u);
  }
}

__global__ __launch_bounds__(256) void em_sets_emit(const EmSetsJob* __restrict__ jobs) {
  const EmSetsJob J = jobs[blockIdx.y];
  const uint32_t s = blockIdx.x * 256u + threadIdx.x;
  if (s > J.slot_mask || J.tag[s] == 0ull) return;
  const uint32_t k = atomicAdd(J.n_out, 1u);
  if (k >= J.max_out) return;
  const uint32_t rep = J.row[s];
  for (int w = 0; w < J.words; ++w) J.out_sets[(int64_t)k * J.words + w] = J.sets[(int64_t)rep * J.words + w];
  J.out_count[k] = J.cnt[s];
}

// ------------------------------------------------------------------------------------------------ SQUAREM
constexpr int kEmThreads = 1024;
constexpr int kEmScaleLds = 8192;        // sets whose 1 / (sum of member abundances) live in LDS; more go through HBM / L2

// one gene of em_kernel_genes: its distinct sets in two sparse forms
struct EmGene {
  int64_t w_off;          // weight / scale [n_sets]
  int64_t so_off;         // set_off [n_sets + 1]: members of set u = members[mem_off + set_off[u] .. set_off[u + 1])
  int64_t mem_off;        // members: allele numbers (uint16), and al_sets: set numbers (uint32) -- both nnz entries
  int64_t ao_off;         // al_off [n_allele + 1]: sets of allele a = al_sets[mem_off + al_off[a] .. al_off[a + 1])
  int64_t prob_off;
  int32_t n_sets, n_allele;
};

struct EmArrays {
  const double* weight;
  double* scale;
  const uint32_t* set_off;
  const uint16_t* members;
  const uint32_t* al_off;
  const uint32_t* al_sets;
};

struct EmLds {
  double p[kMaxAllele], p1[kMaxAllele], p2[kMaxAllele], p3[kMaxAllele];
  double scalar[4];
  int flag;
};

__device__ inline double lanes16_sum(double v) {
#pragma unroll
  for (int d = 8; d >= 1; d >>= 1) v += __shfl_xor(v, d, 16);
  return v;
}
__device__ inline double wave_sum(double v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
  return v;
}

// next(p): q[a] = sum_u w_u * p[a] / (sum_{b in u} p[b]) over the sets that contain a, then normalised
__device__ void em_step(const EmGene& g, const EmArrays& A, double* scale, const double* in, double* out, double* scalar) {
  const int tid = threadIdx.x, l = tid & 15, grp = tid >> 4;
  constexpr int kGroups = kEmThreads / 16;
  const uint32_t* set_off = A.set_off + g.so_off;
  const uint16_t* members = A.members + g.mem_off;
  const uint32_t* al_off = A.al_off + g.ao_off;
  const uint32_t* al_sets = A.al_sets + g.mem_off;
  const double* weight = A.weight + g.w_off;
  for (int u0 = 0; u0 < g.n_sets; u0 += kGroups) {
    const int u = u0 + grp;
    double t = 0.0;
    if (u < g.n_sets)
      for (uint32_t k = set_off[u] + l, e = set_off[u + 1]; k < e; k += 16) t += in[members[k]];
    t = lanes16_sum(t);
    if (u < g.n_sets && l == 0) scale[u] = t != 0.0 ? weight[u] / t : 0.0;
  }
  __syncthreads();
  for (int a0 = 0; a0 < g.n_allele; a0 += kGroups) {
    const int a = a0 + grp;
    double s = 0.0;
    if (a < g.n_allele)
      for (uint32_t k = al_off[a] + l, e = al_off[a + 1]; k < e; k += 16) s += scale[al_sets[k]];
    s = lanes16_sum(s);
    if (a < g.n_allele && l == 0) out[a] = in[a] * s;
  }
  __syncthreads();
  if (tid < 64) {
    double t = 0.0;
    for (int a = tid; a < g.n_allele; a += 64) t += out[a];
    t = wave_sum(t);
    if (tid == 0) scalar[0] = t;
  }
  __syncthreads();
  const double tot = scalar[0];
  for (int a = tid; a < g.n_allele; a += kEmThreads) out[a] = out[a] / tot;
  __syncthreads();
}

__device__ void em_solve(EmLds& sh, const EmGene& g, const EmArrays& A, double* scale, int iter_max, double diff_threshold,
                         double* prob_out, int* iters_out) {
  const int tid = threadIdx.x;
  const int n_allele = g.n_allele;
  for (int a = tid; a < n_allele; a += kEmThreads) sh.p3[a] = 1.0;
  __syncthreads();
  em_step(g, A, scale, sh.p3, sh.p, sh.scalar);
  int iters = 0;
  for (iters = 0; iters < iter_max; ++iters) {
    em_step(g, A, scale, sh.p, sh.p1, sh.scalar);
    em_step(g, A, scale, sh.p1, sh.p2, sh.scalar);
    if (tid < 64) {
      double rs = 0.0, vs = 0.0;
      for (int a = tid; a < n_allele; a += 64) {
        const double r = sh.p1[a] - sh.p[a];
        const double v = sh.p2[a] - sh.p1[a] - r;
        rs += r * r;
        vs += v * v;
      }
      rs = wave_sum(rs);
      vs = wave_sum(vs);
      if (tid == 0) { sh.scalar[1] = rs; sh.scalar[2] = vs; }
    }
    __syncthreads();
    const double rs = sh.scalar[1], vs = sh.scalar[2];
    if (vs > 0.0) {
      const double gs = -sqrt(rs / vs);
      for (int a = tid; a < n_allele; a += kEmThreads) {
        const double r = sh.p1[a] - sh.p[a];
        const double v = sh.p2[a] - sh.p1[a] - r;
        const double x = sh.p[a] - r * gs * 2 + v * (gs * gs);
        sh.p3[a] = x > 0.0 ? x : 0.0;
      }
      __syncthreads();
      em_step(g, A, scale, sh.p3, sh.p1, sh.scalar);
    }
    if (tid < 64) {
      double d = 0.0;
      for (int a = tid; a < n_allele; a += 64) d += fabs(sh.p[a] - sh.p1[a]);
      d = wave_sum(d);
      if (tid == 0) sh.flag = d <= diff_threshold;
    }
    __syncthreads();
    if (sh.flag) break;
    for (int a = tid; a < n_allele; a += kEmThreads) sh.p[a] = sh.p1[a];
    __syncthreads();
  }
  for (int a = tid; a < n_allele; a += kEmThreads) prob_out[a] = sh.p[a];
  if (tid == 0) *iters_out = iters;
}

// the EM of every gene of a sample in ONE launch: workgroup g solves gene g (gk_sample_em; gk_em_run: one gene)
__global__ __launch_bounds__(kEmThreads) void em_kernel_genes(const EmGene* __restrict__ genes, EmArrays A, int iter_max,
                                                              double diff_threshold, double* prob_out, int* iters_out) {
  extern __shared__ __align__(16) unsigned char em_lds[];
  EmLds& sh = *reinterpret_cast<EmLds*>(em_lds);
  const EmGene g = genes[blockIdx.x];
  double* scale = g.n_sets <= kEmScaleLds ? reinterpret_cast<double*>(em_lds + sizeof(EmLds)) : A.scale + g.w_off;
  em_solve(sh, g, A, scale, iter_max, diff_threshold, prob_out + g.prob_off, iters_out + blockIdx.x);
}

// ------------------------------------------------------------------------------------------------ host side
int lanes_per_pair(int words) { return words <= 4 ? 4 : words <= 8 ? 8 : 16; }      // one word per lane (hash / verify)
int lanes_per_pair2(int words) { return words <= 4 ? 2 : words <= 8 ? 4 : 8; }      // two words per lane (candidate sets)

uint32_t blocks_for(int64_t n_rows, int G) {
  const int64_t per_block = (int64_t)(kSetThreads / G) * 16;      // at least 16 rounds of rows per workgroup
  return (uint32_t)std::max<int64_t>(1, std::min<int64_t>(256, (n_rows + per_block - 1) / per_block));
}

template <typename Launch>
void by_group(int G, Launch&& go) {
  if (G == 4) go(std::integral_constant<int, 4>());
  else if (G == 8) go(std::integral_constant<int, 8>());
  else go(std::integral_constant<int, 16>());
}

struct Geometry { int G = 4; uint32_t blocks = 1, slots = 0; size_t lds = kLocalBytes; };

Geometry geometry_of(const std::vector<EmSetsJob>& jobs) {
  Geometry g;
  int words = 1;
  for (const EmSetsJob& j : jobs) {
    words = std::max(words, j.words);
    g.blocks = std::max(g.blocks, j.n_blocks);
    g.slots = std::max(g.slots, j.slot_mask);
    if (j.mask_in_lds)      // the gene's rows, the two rows behind them, room for the lanes that read past a row's end
      g.lds = std::max(g.lds, kLocalBytes + ((size_t)(j.n_span + 2) * (size_t)((j.words + 1) & ~1) + 2 * kMaxWords) * sizeof(uint32_t));
  }
  g.G = lanes_per_pair(words);
  return g;
}

// candidate sets (+ hashes) of the listed jobs: one launch; `d_jobs` holds them on the device
int launch_sets(gk_ctx* ctx, gk_tab* tab, const std::vector<EmSetsJob>& jobs, const EmSetsJob* d_jobs) {
  const Geometry geo = geometry_of(jobs);
  int rc = GK_OK;
  auto go = [&](auto g) {
    constexpr int G = decltype(g)::value;
    if (geo.lds > 48 * 1024 && hipFuncSetAttribute((const void*)em_sets_groups<G>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                   (int)geo.lds) != hipSuccess) {
      gk_set_error("candidate sets: %zu bytes of LDS refused", geo.lds);
      rc = GK_ERR_HIP;
      return;
    }
    // the workgroups of a gene were counted for groups of geo.G lanes: with half as many lanes per pair a workgroup
    // takes twice the pairs per round -- the same grid then makes eight rounds instead of sixteen
    GK_PROF(ctx, "em_sets_groups", GK_KERNEL(em_sets_groups<G>, dim3(geo.blocks, (unsigned)jobs.size()), dim3(kSetThreads),
                                             geo.lds, ctx->stream, d_jobs, tab->d_off, tab->d_ids));
  };
  const int G2 = geo.G / 2;      // two words per lane
  if (G2 == 2) go(std::integral_constant<int, 2>());
  else if (G2 == 4) go(std::integral_constant<int, 4>());
  else go(std::integral_constant<int, 8>());
  return rc;
}

void launch_hash(gk_ctx* ctx, const std::vector<EmSetsJob>& jobs, const EmSetsJob* d_jobs) {
  const Geometry geo = geometry_of(jobs);
  by_group(geo.G, [&](auto g) {
    constexpr int G = decltype(g)::value;
    GK_PROF(ctx, "em_sets_hash", GK_KERNEL(em_sets_hash<G>, dim3(geo.blocks, (unsigned)jobs.size()), dim3(kSetThreads),
                                           kLocalBytes, ctx->stream, d_jobs));
  });
}

void launch_verify_emit(gk_ctx* ctx, const std::vector<EmSetsJob>& jobs, const EmSetsJob* d_jobs) {
  const Geometry geo = geometry_of(jobs);
  by_group(geo.G, [&](auto g) {
    constexpr int G = decltype(g)::value;
    GK_PROF(ctx, "em_sets_verify", GK_KERNEL(em_sets_verify<G>, dim3(geo.blocks, (unsigned)jobs.size()), dim3(kSetThreads), 0,
                                             ctx->stream, d_jobs));
  });
  GK_PROF(ctx, "em_sets_emit", GK_KERNEL(em_sets_emit, dim3(geo.slots / 256 + 1, (unsigned)jobs.size()), dim3(256), 0,
                                         ctx->stream, d_jobs));
}

uint32_t slots_for(int64_t n_distinct_max) {
  uint32_t log2 = 10;
  while ((1ull << log2) < (uint64_t)n_distinct_max * 2 && log2 < 28) ++log2;   // load factor <= 0.5 even if all sets differ
  return 1u << log2;
}

// the sparse forms of sorted, non-empty distinct sets for em_kernel_genes, appended to the sample's arrays
struct EmHost {
  std::vector<EmGene> genes;
  std::vector<double> weight;
  std::vector<uint32_t> set_off, al_off, al_sets;
  std::vector<uint16_t> members;
  int max_sets = 0;
  void add(const uint32_t* sets, const double* w, int n_sets, int words, int n_allele, int64_t prob_off) {
    EmGene g{(int64_t)weight.size(), (int64_t)set_off.size(), (int64_t)members.size(), (int64_t)al_off.size(), prob_off, n_sets,
             n_allele};
    std::vector<uint32_t> per_allele((size_t)n_allele + 1, 0);
    const size_t m0 = members.size();
    for (int u = 0; u < n_sets; ++u) {
      set_off.push_back((uint32_t)(members.size() - m0));
      for (int q = 0; q < words; ++q) {
        uint32_t bits = sets[(size_t)u * words + q];
        while (bits) {
          const int a = q * 32 + __builtin_ctz(bits);
          bits &= bits - 1;
          if (a >= n_allele) continue;
          members.push_back((uint16_t)a);
          per_allele[(size_t)a + 1]++;
        }
      }
      weight.push_back(w[u]);
    }
    set_off.push_back((uint32_t)(members.size() - m0));
    for (int a = 0; a < n_allele; ++a) per_allele[(size_t)a + 1] += per_allele[a];
    al_off.insert(al_off.end(), per_allele.begin(), per_allele.end());
    al_sets.resize(members.size());
    std::vector<uint32_t> at(per_allele.begin(), per_allele.end() - 1);
    for (int u = 0; u < n_sets; ++u)
      for (uint32_t k = set_off[(size_t)g.so_off + u]; k < set_off[(size_t)g.so_off + u + 1]; ++k)
        al_sets[m0 + at[members[m0 + k]]++] = (uint32_t)u;       // ascending set numbers per allele
    genes.push_back(g);
    max_sets = std::max(max_sets, n_sets);
  }
};

// every gene's SQUAREM loop in one launch; probs [total alleles] and iters [genes] on the host when it returns
int run_em(gk_ctx* ctx, EmHost& h, int64_t n_prob, int iter_max, double diff_threshold, std::vector<double>& probs,
           std::vector<int>& iters) {
  hipStream_t st = ctx->stream;
  std::vector<void*> temps;
  auto take = [&](void** p, size_t bytes) -> hipError_t {
    hipError_t e = gk_pool_malloc(ctx, p, bytes ? bytes : 16);
    if (e == hipSuccess) temps.push_back(*p);
    return e;
  };
  auto done = [&](int rc) { for (void* p : temps) gk_pool_free(ctx, p); return rc; };
  EmGene* d_genes = nullptr;
  double *d_w = nullptr, *d_scale = nullptr, *d_prob = nullptr;
  uint32_t *d_so = nullptr, *d_ao = nullptr, *d_as = nullptr;
  uint16_t* d_mem = nullptr;
  int* d_it = nullptr;
  if (take((void**)&d_genes, h.genes.size() * sizeof(EmGene)) != hipSuccess ||
      take((void**)&d_w, h.weight.size() * sizeof(double)) != hipSuccess ||
      take((void**)&d_scale, h.weight.size() * sizeof(double)) != hipSuccess ||
      take((void**)&d_so, h.set_off.size() * sizeof(uint32_t)) != hipSuccess ||
      take((void**)&d_mem, h.members.size() * sizeof(uint16_t)) != hipSuccess ||
      take((void**)&d_ao, h.al_off.size() * sizeof(uint32_t)) != hipSuccess ||
      take((void**)&d_as, h.al_sets.size() * sizeof(uint32_t)) != hipSuccess ||
      take((void**)&d_prob, (size_t)std::max<int64_t>(n_prob, 1) * sizeof(double)) != hipSuccess ||
      take((void**)&d_it, h.genes.size() * sizeof(int)) != hipSuccess) {
    gk_set_error("out of device memory for the EM of a sample");
    return done(GK_ERR_HIP);
  }
  // the sources of these copies live until the stream is waited for below
  hipMemcpyAsync(d_genes, h.genes.data(), h.genes.size() * sizeof(EmGene), hipMemcpyHostToDevice, st);
  hipMemcpyAsync(d_w, h.weight.data(), h.weight.size() * sizeof(double), hipMemcpyHostToDevice, st);
  hipMemcpyAsync(d_so, h.set_off.data(), h.set_off.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st);
  if (!h.members.empty()) {
    hipMemcpyAsync(d_mem, h.members.data(), h.members.size() * sizeof(uint16_t), hipMemcpyHostToDevice, st);
    hipMemcpyAsync(d_as, h.al_sets.data(), h.al_sets.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st);
  }
  hipMemcpyAsync(d_ao, h.al_off.data(), h.al_off.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st);
  hipMemsetAsync(d_prob, 0, (size_t)std::max<int64_t>(n_prob, 1) * sizeof(double), st);
  const size_t lds = sizeof(EmLds) + (size_t)std::min(h.max_sets, kEmScaleLds) * sizeof(double);
  if (lds > 48 * 1024 &&
      hipFuncSetAttribute((const void*)em_kernel_genes, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
    gk_set_error("EM: %zu bytes of LDS refused", lds);
    return done(GK_ERR_HIP);
  }
  const EmArrays A{d_w, d_scale, d_so, d_mem, d_ao, d_as};
  GK_PROF(ctx, "em_kernel_genes", GK_KERNEL(em_kernel_genes, dim3((unsigned)h.genes.size()), dim3(kEmThreads), lds, st, d_genes, A,
                                            iter_max, diff_threshold, d_prob, d_it));
  iters.assign(h.genes.size(), 0);
  probs.assign((size_t)std::max<int64_t>(n_prob, 1), 0.0);
  hipMemcpyAsync(probs.data(), d_prob, (size_t)n_prob * sizeof(double), hipMemcpyDeviceToHost, st);
  hipMemcpyAsync(iters.data(), d_it, h.genes.size() * sizeof(int), hipMemcpyDeviceToHost, st);
  if (hipGetLastError() != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
    gk_set_error("EM: %s", hipGetErrorString(hipGetLastError()));
    return done(GK_ERR_HIP);
  }
  return done(GK_OK);
}

}  // namespace

extern "C" {

int gk_em_sets(gk_ctx* ctx, gk_tab* tab, gk_dptr d_rows, int64_t n_rows, int32_t vbeg, int32_t vend, gk_dptr d_mask,
               int32_t words, gk_dptr d_sets_out) {
  gk_bind(ctx);
  GK_REQUIRE(ctx && tab, "null pointer");
  GK_REQUIRE(words >= 1 && words <= kMaxWords, "more than 512 alleles per gene are not supported by the EM kernel");
  GK_REQUIRE(n_rows >= 0 && n_rows < (1ll << 31) && vend >= vbeg, "bad candidate-set geometry");
  if (!n_rows) return GK_OK;
  const size_t mask_bytes = (size_t)(vend - vbeg) * words * sizeof(uint32_t);
  std::vector<EmSetsJob> jobs(1);
  EmSetsJob& j = jobs[0];
  memset(&j, 0, sizeof(j));
  j.rows = gk_ptr<int32_t>(d_rows); j.n_rows = n_rows; j.mask = gk_ptr<uint32_t>(d_mask);
  j.vbeg = vbeg; j.n_span = vend - vbeg; j.words = words; j.mask_in_lds = mask_bytes <= kMaskLdsMax ? 1 : 0;
  j.sets = gk_ptr<uint32_t>(d_sets_out);
  j.n_blocks = blocks_for(n_rows, lanes_per_pair(words));
  EmSetsJob* d_jobs = nullptr;
  GK_HIP(gk_pool_malloc(ctx, (void**)&d_jobs, sizeof(EmSetsJob)));
  const hipError_t e = gk_send(ctx, d_jobs, jobs.data(), sizeof(EmSetsJob));
  const int rc = e == hipSuccess ? launch_sets(ctx, tab, jobs, d_jobs) : GK_ERR_HIP;
  gk_pool_free(ctx, d_jobs);
  if (rc) return rc;
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
  const uint32_t n_slots = slots_for(n_rows);
  char* block = nullptr;      // [tag | cnt | row | n_out, flags | job]
  const size_t o_cnt = (size_t)n_slots * 8, o_row = o_cnt + (size_t)n_slots * 4, o_n = o_row + (size_t)n_slots * 4, o_job = o_n + 16;
  uint32_t *d_out_sets = nullptr, *d_out_count = nullptr;
  GK_HIP(gk_pool_malloc(ctx, (void**)&block, o_job + sizeof(EmSetsJob)));
  GK_HIP(gk_pool_malloc(ctx, (void**)&d_out_sets, (size_t)max_out * words * sizeof(uint32_t)));
  GK_HIP(gk_pool_malloc(ctx, (void**)&d_out_count, (size_t)max_out * sizeof(uint32_t)));
  auto done = [&](int rc) { gk_pool_free(ctx, block); gk_pool_free(ctx, d_out_sets); gk_pool_free(ctx, d_out_count); return rc; };
  std::vector<EmSetsJob> jobs(1);
  uint32_t back[2] = {0, 0};      // distinct sets, flags
  for (uint32_t seed = 0; seed < 4; ++seed) {
    EmSetsJob& j = jobs[0];
    memset(&j, 0, sizeof(j));
    j.n_rows = n_rows; j.words = words; j.sets = gk_ptr<uint32_t>(d_sets);
    j.tag = (unsigned long long*)block; j.cnt = (uint32_t*)(block + o_cnt); j.row = (uint32_t*)(block + o_row);
    j.slot_mask = n_slots - 1; j.seed = seed; j.n_out = (uint32_t*)(block + o_n); j.flags = j.n_out + 1;
    j.n_blocks = blocks_for(n_rows, lanes_per_pair(words));
    j.out_sets = d_out_sets; j.out_count = d_out_count; j.max_out = (uint32_t)max_out;
    if (hipMemsetAsync(block, 0, o_row, st) != hipSuccess || hipMemsetAsync(block + o_row, 0xFF, (size_t)n_slots * 4, st) != hipSuccess ||
        hipMemsetAsync(block + o_n, 0, 16, st) != hipSuccess) return done(GK_ERR_HIP);
    EmSetsJob* d_jobs = (EmSetsJob*)(block + o_job);
    if (gk_send(ctx, d_jobs, jobs.data(), sizeof(EmSetsJob)) != hipSuccess) return done(GK_ERR_HIP);
    launch_hash(ctx, jobs, d_jobs);
    launch_verify_emit(ctx, jobs, d_jobs);
    if (hipGetLastError() != hipSuccess || gk_fetch(ctx, back, block + o_n, sizeof(back)) != hipSuccess) {
      gk_set_error("distinct candidate sets: %s", hipGetErrorString(hipGetLastError()));
      return done(GK_ERR_HIP);
    }
    if (!(back[1] & 2u)) break;       // no two different sets under one hash
    if (seed == 3) { gk_set_error("candidate sets: hash collisions under four seeds"); return done(GK_ERR_ASSERT); }
  }
  const uint32_t n = back[0];
  int rc = GK_OK;
  if ((back[1] & 1u) || n > (uint32_t)max_out) {
    gk_set_error("%u distinct candidate sets exceed the output capacity %d", n, max_out);
    rc = GK_ERR_CAPACITY;
  } else if (n) {
    if (gk_fetch_queue(ctx, sets_out, d_out_sets, (size_t)n * words * sizeof(uint32_t)) != hipSuccess ||
        gk_fetch_queue(ctx, count_out, d_out_count, (size_t)n * sizeof(uint32_t)) != hipSuccess ||
        gk_fetch_wait(ctx) != hipSuccess) {
      gk_fetch_cancel(ctx);
      return done(GK_ERR_HIP);
    }
  }
  *n_out = (int32_t)n;
  return done(rc);
}

int gk_em_run(gk_ctx* ctx, const uint32_t* sets, const double* weight, int32_t n_sets, int32_t words, int32_t n_allele,
              int32_t iter_max, double diff_threshold, double* prob_out, int32_t* iters_out) {
  gk_bind(ctx);
  GK_REQUIRE(ctx && sets && weight && prob_out && iters_out, "null pointer");
  GK_REQUIRE(n_sets > 0 && words >= 1 && words <= kMaxWords && n_allele >= 1 && n_allele <= words * 32,
             "bad EM geometry");
  EmHost h;
  h.add(sets, weight, n_sets, words, n_allele, 0);
  std::vector<double> probs;
  std::vector<int> iters;
  const int rc = run_em(ctx, h, n_allele, iter_max, diff_threshold, probs, iters);
  if (rc) return rc;
  std::copy(probs.begin(), probs.begin() + n_allele, prob_out);
  *iters_out = iters[0];
  return GK_OK;
}

/* The EM strategy for ALL genes of a sample in one call on the calling thread and the context's one stream
 * (kir_typing.py:163-195 is the reference's gene loop, typing_em.py:68-188 the work per gene): the candidate sets of every
 * gene in ONE launch (hashes counted on the way), one launch that verifies the classes, one that emits the distinct sets
 * (one wait for the counts, one for the sets), the host half -- the ascending order numpy.unique gives the sets, the reads
 * naming each allele, the empty set dropped, the two sparse forms -- in C++, and the SQUAREM loops of all genes in ONE
 * launch (a workgroup per gene).  jobs[i]: the gene's rows (NH == 1 pairs), variant span, bit rows; prob_out / count_out:
 * n_allele entries per job, one after the other; per job the distinct sets and the SQUAREM steps come back.
 * GK_ERR_CAPACITY when a gene has more than 2^18 distinct sets (the caller takes the per-gene calls). */
int gk_sample_em(gk_ctx* ctx, gk_tab* tab, gk_em_job* jobs, int32_t n_jobs, int32_t iter_max, double diff_threshold,
                 double* prob_out, int64_t* count_out) {
  gk_bind(ctx);
  GK_REQUIRE(ctx && tab && jobs && prob_out && count_out && n_jobs >= 0, "null pointer");
  hipStream_t st = ctx->stream;
  struct Work {
    uint32_t cap = 0, n_slots = 0;
    uint32_t *d_sets = nullptr, *d_out_sets = nullptr, *d_out_count = nullptr;
    uint32_t back[2] = {0, 0};      // distinct sets, flags
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
               j.n_rows < (1ll << 31) && j.vend >= j.vbeg, "bad EM job");
    prob_off[i] = out_off;
    for (int a = 0; a < j.n_allele; ++a) { prob_out[out_off + a] = 0.0; count_out[out_off + a] = 0; }
    out_off += j.n_allele;
  }
  // ---- phase 1: candidate sets + distinct sets of every gene: three launches
  std::vector<int> live;
  int max_words = 1;
  for (int i = 0; i < n_jobs; ++i)
    if (jobs[i].n_rows && jobs[i].n_allele) { live.push_back(i); max_words = std::max(max_words, (int)jobs[i].words); }
  const int G = lanes_per_pair(max_words);
  std::vector<EmSetsJob> sj(live.size());
  // one block for the tables of all genes: [tag | cnt] of every gene zeroed, [row] all ones, [n_out, flags] per gene zeroed
  size_t tag_bytes = 0, row_bytes = 0;
  for (int i : live) {
    Work& w = work[i];
    w.cap = (uint32_t)std::min<int64_t>(jobs[i].n_rows, 1ll << 18);
    w.n_slots = slots_for(w.cap);
    tag_bytes += (size_t)w.n_slots * 12;
    row_bytes += (size_t)w.n_slots * 4;
  }
  char* table = nullptr;
  EmSetsJob* d_sj = nullptr;
  const size_t o_row = tag_bytes, o_small = tag_bytes + row_bytes, table_bytes = o_small + live.size() * 16;
  if (!live.empty()) {
    if (take((void**)&table, table_bytes) != hipSuccess || take((void**)&d_sj, live.size() * sizeof(EmSetsJob)) != hipSuccess) {
      gk_set_error("out of device memory for the candidate sets of a sample");
      return done(GK_ERR_HIP);
    }
    for (int i : live) {
      gk_em_job& j = jobs[i];
      Work& w = work[i];
      if (take((void**)&w.d_sets, (size_t)j.n_rows * j.words * sizeof(uint32_t)) != hipSuccess ||
          take((void**)&w.d_out_sets, (size_t)w.cap * j.words * sizeof(uint32_t)) != hipSuccess ||
          take((void**)&w.d_out_count, (size_t)w.cap * sizeof(uint32_t)) != hipSuccess) {
        gk_set_error("out of device memory for the candidate sets of a gene");
        return done(GK_ERR_HIP);
      }
    }
  }
  for (uint32_t seed = 0; seed < 4 && !live.empty(); ++seed) {
    size_t at_tag = 0, at_row = o_row;
    for (size_t q = 0; q < live.size(); ++q) {
      const int i = live[q];
      gk_em_job& j = jobs[i];
      Work& w = work[i];
      EmSetsJob& s = sj[q];
      memset(&s, 0, sizeof(s));
      const size_t mask_bytes = (size_t)(j.vend - j.vbeg) * j.words * sizeof(uint32_t);
      s.rows = gk_ptr<int32_t>(j.d_rows); s.n_rows = j.n_rows; s.mask = gk_ptr<uint32_t>(j.d_mask);
      s.vbeg = j.vbeg; s.n_span = j.vend - j.vbeg; s.words = j.words; s.mask_in_lds = mask_bytes <= kMaskLdsMax ? 1 : 0;
      s.sets = w.d_sets;
      s.tag = (unsigned long long*)(table + at_tag); s.cnt = (uint32_t*)(table + at_tag + (size_t)w.n_slots * 8);
      s.row = (uint32_t*)(table + at_row);
      s.slot_mask = w.n_slots - 1; s.seed = seed;
      s.n_out = (uint32_t*)(table + o_small + q * 16); s.flags = s.n_out + 1;
      s.n_blocks = blocks_for(j.n_rows, G);
      s.out_sets = w.d_out_sets; s.out_count = w.d_out_count; s.max_out = w.cap;
      at_tag += (size_t)w.n_slots * 12;
      at_row += (size_t)w.n_slots * 4;
    }
    if (hipMemsetAsync(table, 0, tag_bytes, st) != hipSuccess || hipMemsetAsync(table + o_row, 0xFF, row_bytes, st) != hipSuccess ||
        hipMemsetAsync(table + o_small, 0, live.size() * 16, st) != hipSuccess ||
        gk_send(ctx, d_sj, sj.data(), sj.size() * sizeof(EmSetsJob)) != hipSuccess) {
      gk_set_error("sample EM: %s", hipGetErrorString(hipGetLastError()));
      return done(GK_ERR_HIP);
    }
    if (seed == 0) {
      const int rc = launch_sets(ctx, tab, sj, d_sj);
      if (rc) return done(rc);
    } else {
      launch_hash(ctx, sj, d_sj);      // the sets stand; only their hashes are taken again
    }
    launch_verify_emit(ctx, sj, d_sj);
    for (size_t q = 0; q < live.size(); ++q)
      if (gk_fetch_queue(ctx, work[live[q]].back, sj[q].n_out, 8) != hipSuccess) { gk_fetch_cancel(ctx); return done(GK_ERR_HIP); }
    if (hipGetLastError() != hipSuccess || gk_fetch_wait(ctx) != hipSuccess) {
      gk_fetch_cancel(ctx);
      gk_set_error("sample EM: %s", hipGetErrorString(hipGetLastError()));
      return done(GK_ERR_HIP);
    }
    bool collided = false;
    for (int i : live) collided = collided || (work[i].back[1] & 2u);
    if (!collided) break;
    if (seed == 3) { gk_set_error("candidate sets: hash collisions under four seeds"); return done(GK_ERR_ASSERT); }
  }
  for (int i : live) {
    Work& w = work[i];
    const uint32_t n = w.back[0];
    if ((w.back[1] & 1u) || n > w.cap) {
      gk_set_error("%u distinct candidate sets exceed the capacity %u of the one-call EM", n, w.cap);
      return done(GK_ERR_CAPACITY);
    }
    if (!n) continue;
    w.sets.resize((size_t)n * jobs[i].words);
    w.mult.resize(n);
    if (gk_fetch_queue(ctx, w.sets.data(), w.d_out_sets, w.sets.size() * sizeof(uint32_t)) != hipSuccess ||
        gk_fetch_queue(ctx, w.mult.data(), w.d_out_count, w.mult.size() * sizeof(uint32_t)) != hipSuccess) {
      gk_fetch_cancel(ctx);
      return done(GK_ERR_HIP);
    }
  }
  if (gk_fetch_wait(ctx) != hipSuccess) { gk_fetch_cancel(ctx); return done(GK_ERR_HIP); }
  // ---- phase 2 (host): numpy.unique's order, the reads naming each allele, the empty set dropped, the sparse forms
  EmHost h;
  std::vector<int> gene_job;
  for (int i : live) {
    Work& w = work[i];
    gk_em_job& j = jobs[i];
    const uint32_t n = w.back[0];
    j.n_distinct = (int32_t)n;
    if (!n) continue;
    const int words = j.words;
    w.order.resize(n);
    for (uint32_t u = 0; u < n; ++u) w.order[u] = u;
    std::sort(w.order.begin(), w.order.end(), [&](int64_t x, int64_t y) {
      const uint32_t *a = w.sets.data() + (size_t)x * words, *b = w.sets.data() + (size_t)y * words;
      for (int q = 0; q < words; ++q)
        if (a[q] != b[q]) return a[q] < b[q];
      return false;
    });
    std::vector<uint32_t> sorted_sets;
    std::vector<double> sorted_w;
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
      sorted_sets.insert(sorted_sets.end(), row, row + words);
      sorted_w.push_back((double)w.mult[(size_t)u]);
    }
    if (!sorted_w.empty()) {
      h.add(sorted_sets.data(), sorted_w.data(), (int)sorted_w.size(), words, j.n_allele, prob_off[i]);
      gene_job.push_back(i);
    }
  }
  if (h.genes.empty()) return done(GK_OK);
  // ---- phase 3: every gene's SQUAREM loop in one launch
  std::vector<double> probs;
  std::vector<int> iters;
  const int rc = run_em(ctx, h, out_off, iter_max, diff_threshold, probs, iters);
  if (rc) return done(rc);
  for (size_t k = 0; k < h.genes.size(); ++k) {
    const int i = gene_job[k];
    jobs[i].iterations = iters[k];
    for (int a = 0; a < jobs[i].n_allele; ++a) prob_out[prob_off[i] + a] = probs[(size_t)(prob_off[i] + a)];
  }
  return done(GK_OK);
}

}  // extern "C"

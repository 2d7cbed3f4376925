// Read -> variant hit tabulation on the device.
//
// Replaces, for packed records, the per-pair Python loop of the reference:
//   filterRead            hisat2.py:541-578   (flag & 2, NM present and <= 4, both mates)
//   recordToRawVariant    hisat2.py:279-515   (CIGAR x mismatch co-walk -> match/single/ins/del events)
//   findVariantId         hisat2.py:581-606   (exact (pos,ref,typ,val) lookup, novel ids in first-seen order)
//   getVariantsBoundary   hisat2.py:692-713   (bisect window [single@left 'A', single@right 'T'))
//   getPNFromVariantList  hisat2.py:716-800   (positives, negatives, novel-indel drop, N exclusion, deletion edge rule)
//   extractVariant        hisat2.py:803-844   (pair assembly, NH, backbone)
//
// Data layout (HBM): mates are 128-byte records, two per pair.  A workgroup stages its 256 records
// (32 KB contiguous) into LDS with coalesced 16-byte loads; every lane then walks its own record out
// of LDS (stride 33 dwords: conflict-free, and the CIGAR / mismatch cursors index it dynamically
// without private memory).  The sorted variant key table (<= ~0.5 MB) stays L2 resident and is
// entered through a 16-bp bucket table, so a lookup is two bucket reads plus a <= 3-step bisection
// instead of a 16-step one.  Outputs are one CSR (uint32 offsets, uint32 ordinals) in the factor
// order lpv, rpv, lnv, rnv.
//
// Two passes (count, emit) around one exclusive scan; novel variants are deduplicated in a device
// hash table keyed by the packed variant key, ranked by first appearance through a bitmap over
// (mate, event) sequence numbers, so that the numbering equals the reference's sequential counter.
#include <algorithm>
#include <vector>

#include "gk_common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kMaxEv = GK_MAX_EVENTS;
constexpr uint64_t kEmpty = ~0ull;

constexpr int kMateWords = sizeof(gk_mate) / 4;   // 32
constexpr int kCigWord = 3;                         // uint16 cig[] starts at byte 12
constexpr int kMmWord = kCigWord + GK_MAX_CIG / 2;  // 10
constexpr int kInsWord = kMmWord + GK_MAX_MM;       // 26
static_assert(sizeof(gk_mate) == 128 && kInsWord + GK_MAX_INS == kMateWords, "gk_mate layout");
constexpr int kRecLd = kMateWords + 1;   // LDS stride of a staged record (dwords)
constexpr int kEvLd = kMaxEv + 1;        // LDS stride of a lane's event words (odd: conflict-free)
constexpr int kMaskWords = 8;            // kept-candidate bits saved per mate by pass 1: windows of up to 256 variants

// event word: what the negative filter needs to know about one non-match event of the mate
constexpr uint32_t kEvNovel = 1u << 31;   // not an index variant; low 24 bits = position
constexpr uint32_t kEvIsN = 1u << 30;     // substitution to 'N'
constexpr uint32_t kEvOrdMask = (1u << 26) - 1;
constexpr uint32_t kEvSlotMask = (1u << 30) - 1;   // saved form of a novel event: kEvNovel | slot of the novel table (at most 2^30 slots)

// a staged record, read out of LDS
struct MateView {
  static constexpr int kCapCig = GK_MAX_CIG, kCapMm = GK_MAX_MM, kCapIns = GK_MAX_INS, kCapEv = kMaxEv;
  const uint32_t* w;
  __device__ bool spilled() const { return n_cig() == GK_SPILLED; }   // the pair is in the wide array
  __device__ uint32_t pos0() const { return w[0]; }
  __device__ uint32_t flag() const { return w[1] & 0xFFFFu; }
  __device__ uint32_t ref() const { return (w[1] >> 16) & 0xFFu; }
  __device__ uint32_t nm() const { return w[2] & 0xFFu; }
  __device__ uint32_t n_cig() const { return (w[2] >> 8) & 0xFFu; }
  __device__ uint32_t n_mm() const { return (w[2] >> 16) & 0xFFu; }
  __device__ uint32_t cig(int i) const {  // uint16 array starting at byte 12
    const uint32_t word = w[kCigWord + (i >> 1)];
    return (i & 1) ? (word >> 16) : (word & 0xFFFFu);
  }
  __device__ uint32_t mm_off(int i) const { return w[kMmWord + i] & 0xFFFFu; }
  __device__ uint32_t mm_base(int i) const { return (w[kMmWord + i] >> 16) & 0xFFu; }
  __device__ uint32_t ins(int i) const { return w[kInsWord + i]; }
  __device__ bool passes() const { return (flag() & 2u) && nm() != GK_NM_ABSENT && nm() <= 4u; }
};

// a record whose first kHeadWords words are staged in LDS and whose rest is read where it lies in global memory: the
// header, all 14 CIGAR operations and the first 6 mismatches are in the head -- a 150-base read that passes the NM <= 4
// filter rarely has more events -- so pass 1 stages 64 bytes per mate instead of 128 (8 waves per SIMD fit)
constexpr int kHeadWords = 16;
constexpr int kHeadLd = kHeadWords + 1;
struct HeadView {
  static constexpr int kCapCig = GK_MAX_CIG, kCapMm = GK_MAX_MM, kCapIns = GK_MAX_INS, kCapEv = kMaxEv;
  const uint32_t* w;      // LDS: words [0, kHeadWords)
  const uint32_t* g;      // the whole record in global memory
  __device__ uint32_t word(int i) const { return i < kHeadWords ? w[i] : g[i]; }
  __device__ bool spilled() const { return n_cig() == GK_SPILLED; }
  __device__ uint32_t pos0() const { return w[0]; }
  __device__ uint32_t flag() const { return w[1] & 0xFFFFu; }
  __device__ uint32_t ref() const { return (w[1] >> 16) & 0xFFu; }
  __device__ uint32_t nm() const { return w[2] & 0xFFu; }
  __device__ uint32_t n_cig() const { return (w[2] >> 8) & 0xFFu; }
  __device__ uint32_t n_mm() const { return (w[2] >> 16) & 0xFFu; }
  __device__ uint32_t cig(int i) const {
    const uint32_t x = w[kCigWord + (i >> 1)];      // words 3 .. 9: always in the head
    return (i & 1) ? (x >> 16) : (x & 0xFFFFu);
  }
  __device__ uint32_t mm_off(int i) const { return word(kMmWord + i) & 0xFFFFu; }
  __device__ uint32_t mm_base(int i) const { return (word(kMmWord + i) >> 16) & 0xFFu; }
  __device__ uint32_t ins(int i) const { return g[kInsWord + i]; }
  __device__ bool passes() const { return (flag() & 2u) && nm() != GK_NM_ABSENT && nm() <= 4u; }
};
static_assert(kCigWord + GK_MAX_CIG / 2 <= kHeadWords, "the CIGAR lies in the staged head");

// a record of the wide format (gk_mate_wide), read where it lies in global memory: such pairs are rare
struct WideView {
  static constexpr int kCapCig = GK_WIDE_CIG, kCapMm = GK_WIDE_MM, kCapIns = GK_WIDE_INS, kCapEv = GK_WIDE_EVENTS;
  const gk_mate_wide* p;
  __device__ uint32_t pos0() const { return p->pos0; }
  __device__ uint32_t flag() const { return p->flag; }
  __device__ uint32_t ref() const { return p->ref; }
  __device__ uint32_t nm() const { return p->nm; }
  __device__ uint32_t n_cig() const { return p->n_cig; }
  __device__ uint32_t n_mm() const { return p->n_mm; }
  __device__ uint32_t cig(int i) const { return p->cig[i]; }
  __device__ uint32_t mm_off(int i) const { return p->mm[i] >> 8; }
  __device__ uint32_t mm_base(int i) const { return p->mm[i] & 0xFFu; }
  __device__ uint32_t ins(int i) const { return p->ins[i]; }
  __device__ bool passes() const { return (flag() & 2u) && nm() != GK_NM_ABSENT && nm() <= 4u; }
};
static_assert(sizeof(gk_mate_wide) == 2048, "gk_mate_wide layout");

// the heads (first 64 bytes) of the workgroup's records [m0, m0 + 256) -> LDS: 4 lanes per record, 16 bytes each
__device__ inline void stage_heads(const gk_mate* mates, int64_t m0, int64_t n_mates, uint32_t* rec) {
  const int n_rec = (int)min<int64_t>(kThreads, n_mates - m0);
#pragma unroll
  for (int k = 0; k < kHeadWords / 4; ++k) {
    const int idx = threadIdx.x + kThreads * k;        // (record, quarter of its head)
    const int rcd = idx >> 2, part = idx & 3;
    if (rcd < n_rec) {
      const uint4 v = reinterpret_cast<const uint4*>(mates + m0 + rcd)[part];
      uint32_t* d = rec + rcd * kHeadLd + part * 4;
      d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
  }
  __syncthreads();
}

// the workgroup's records [m0, m0 + 256) -> LDS, coalesced
__device__ inline void stage_mates(const gk_mate* mates, int64_t m0, int64_t n_mates, uint32_t* rec) {
  const uint4* src = reinterpret_cast<const uint4*>(mates + m0);
  const int n_vec = (int)min<int64_t>(kThreads, n_mates - m0) * (kMateWords / 4);
#pragma unroll
  for (int k = 0; k < kMateWords / 4; ++k) {
    const int idx = threadIdx.x + kThreads * k;
    if (idx < n_vec) {
      const uint4 v = src[idx];
      uint32_t* d = rec + (idx >> 3) * kRecLd + (idx & 7) * 4;
      d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
  }
  __syncthreads();
}

struct IndexView {
  const uint64_t* key;
  const int32_t* bucket;      // per gene: lower bound of (gene, 16 * b) for b = 0 .. n_bucket(gene) - 1
  const int32_t* gene_boff;   // [n_gene + 1] first bucket of each gene
  int n_var, n_gene;
  // optional pileup correction of mismatches (hisat2.errorCorrection 609-654): corr[(gene_pos0[ref] + pos) * 5
  // + code(read base)] = the base to use instead (0 = keep), codes A C G T N; null = no correction
  const uint8_t* corr;
  const int64_t* gene_pos0;   // [n_gene + 1]
  const uint32_t* del_bits;   // bit v = index variant v is a deletion (two zero words of padding at the end)
  const int32_t* lb_a;        // per position: first ordinal with key >= (gene, pos, single, 'A')
  const int32_t* lb_t;        // ... >= (gene, pos, single, 'T')
  const int32_t* gene_pbase;  // [n_gene + 1] first table entry of every gene
  const int32_t* snp_ord;     // per position, 4 entries: ordinal of the substitution to A / C / G / T there, -1 = none
};

// first ordinal whose key is >= k, where k = (ref, pos, ...)
__device__ inline int lower_bound_key(const IndexView& ix, uint32_t ref, uint32_t pos, uint64_t k) {
  int lo = 0, hi = ix.n_var;
  if ((int)ref < ix.n_gene) {
    const int b0 = ix.gene_boff[ref], last = ix.gene_boff[ref + 1] - b0 - 1;
    const int b = min((int)(pos >> 4), last);
    lo = ix.bucket[b0 + b];
    hi = ix.bucket[b0 + min(b + 1, last)];
  }
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (ix.key[mid] < k) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// first ordinal whose key is >= (ref, pos, single, 'A') ('T' with `t`): the bounds of a variant window
// (getVariantsBoundary, hisat2.py:692-713), read from the per-position tables; positions past a gene's last variant
// give the gene's end
__device__ inline int lower_bound_single(const IndexView& ix, uint32_t ref, uint32_t pos, bool t) {
  if ((int)ref >= ix.n_gene)
    return lower_bound_key(ix, ref, pos, gk_make_key(ref, pos, GK_TYP_SINGLE, t ? 'T' : 'A'));
  const int base = ix.gene_pbase[ref], n_pos = ix.gene_pbase[ref + 1] - base;
  const int p = min((int)pos, n_pos - 1);      // the last entry of a gene is its end ordinal for both tables
  return (t ? ix.lb_t : ix.lb_a)[base + p];
}

// lower bound of a substitution key: the table entry of its position, then at most a few steps over the
// substitutions listed there (they share a cache line)
__device__ inline int lower_bound_snp(const IndexView& ix, uint32_t ref, uint32_t pos, uint64_t k) {
  int i = lower_bound_single(ix, ref, pos, false);
  while (i < ix.n_var && ix.key[i] < k) ++i;
  return i;
}


__device__ inline uint32_t hash64(uint64_t k) {
  k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33;
  return (uint32_t)k;
}

struct NovelTable {
  uint64_t* keys;   // kEmpty when free
  uint64_t* seq;    // min (mate << 16 | event) over insertions: the first appearance in read order
  uint32_t* rank;   // first-appearance rank (filled by rank kernel)
  uint32_t mask;
  int* flags;       // bit 3 is raised when a key finds no slot within kNovelProbes: the host retries with a larger table
};
constexpr uint32_t kNovelProbes = 2048;   // far beyond any chain at the load factor the host accepts (0.5)

// returns the slot of the key; when the table is (nearly) full the search is cut short and reported
__device__ inline uint32_t novel_insert(const NovelTable& t, uint64_t key, uint64_t seq) {
  uint32_t s = hash64(key) & t.mask;
  for (uint32_t probe = 0; probe <= min(t.mask, kNovelProbes); ++probe) {
    unsigned long long prev = atomicCAS((unsigned long long*)&t.keys[s], (unsigned long long)kEmpty,
                                        (unsigned long long)key);
    if (prev == kEmpty || prev == key) {
      atomicMin((unsigned long long*)&t.seq[s], (unsigned long long)seq);
      return s;
    }
    s = (s + 1) & t.mask;
  }
  atomicOr(t.flags, 8);
  return 0;
}

__device__ inline uint32_t novel_rank(const NovelTable& t, uint64_t key) {
  uint32_t s = hash64(key) & t.mask;
  for (uint32_t probe = 0; probe <= t.mask; ++probe) {
    uint64_t k = t.keys[s];
    if (k == key) return t.rank[s];
    if (k == kEmpty) break;
    s = (s + 1) & t.mask;
  }
  return 0xFFFFFFFFu;
}

// One mate: the CIGAR x mismatch co-walk (recordToRawVariant), with every non-match event resolved
// against the index as it is produced (findVariantId); only an event word per event is kept (LDS).
// The match segments matter solely through the left edge (pos0) and the right edge.
struct Walked {
  int n;                 // events
  bool clipped, overflow, drop, bad_window;
  uint32_t right;        // right edge of the variant window
  uint32_t any_n;        // some event is a substitution to 'N'
  int lo, hi;            // ordinals of the window [lo, hi)
};

// Where a walk keeps its event words.  EvPtr: a row of words the negative filter reads back (LDS in tab_emit, global for
// the wide format), and optionally the saved form of pass 1 beside it.  EvRow (tab_count): ONE word per event -- the saved
// form (ordinal, or kEvNovel | slot of the novel table) with the N flag on top -- the first four in registers, the rest in
// the mate's overflow row; a row of 22 words per mate and array used to be written a few words at a time, which cost a
// full line of HBM traffic (and its read-modify-write) per mate and array.
constexpr int kEvRegs = 4;
constexpr int kEvMore = kMaxEv - kEvRegs;
struct EvPtr {
  uint32_t* evw;
  uint32_t* out;
  __device__ void put(int n, uint32_t word, uint32_t saved, bool) const {
    evw[n] = word;
    if (out) out[n] = saved;
  }
};
struct EvView {      // read side of a row of event words in the filter's own form
  const uint32_t* evw;
  __device__ uint32_t get(int e) const { return evw[e]; }
  __device__ uint32_t n_pos(uint32_t w, const IndexView& ix) const {
    return (w & kEvNovel) ? (w & 0xFFFFFFu) : gk_key_pos(ix.key[w & kEvOrdMask]);
  }
};
struct EvRow {
  uint32_t r[kEvRegs];
  uint32_t* more;        // the mate's overflow row: events kEvRegs .. kMaxEv - 1
  const uint64_t* novel_keys;
  __device__ void put(int n, uint32_t, uint32_t saved, bool is_n) {
    const uint32_t w = saved | (is_n ? kEvIsN : 0u);
    if (n == 0) r[0] = w; else if (n == 1) r[1] = w; else if (n == 2) r[2] = w; else if (n == 3) r[3] = w;
    else more[n - kEvRegs] = w;
  }
  __device__ uint32_t get(int e) const {   // a novel event carries bit 31 in either form: far outside any window
    return e == 0 ? r[0] : e == 1 ? r[1] : e == 2 ? r[2] : e == 3 ? r[3] : more[e - kEvRegs];
  }
  __device__ uint32_t n_pos(uint32_t w, const IndexView& ix) const {      // position of an event that reads N
    // the slot's key was stored by an atomicCAS of this very launch (at L2): a plain load may be served by a line the
    // CU's L1 took when the slot was still empty, so the key is read at agent scope
    return (w & kEvNovel) ? gk_key_pos(__hip_atomic_load(&novel_keys[w & kEvSlotMask], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
                          : gk_key_pos(ix.key[w & kEvOrdMask]);
  }
};

// ev_out (pass 1 only, may be null): the positive list of the mate as it will be emitted -- the ordinal of a
// known variant, kEvNovel | slot of the novel table otherwise -- so that pass 2 need not walk again.
// sequence numbers of novel variants are mate << 16 | event (first appearance = smallest), whatever the record format;
// the stride is the event capacity of the widest record format present in the sample.
template <bool kEmit, typename View, typename Ev>
__device__ inline void walk_mate(const View& r, const IndexView& ix, const NovelTable& nt, int64_t m,
                                 Ev& ev, uint32_t* ids, uint32_t o_pos, uint32_t o_pos_end, Walked& wk) {
  wk.n = 0; wk.clipped = false; wk.overflow = false; wk.drop = false; wk.any_n = 0; wk.bad_window = false;
  wk.lo = wk.hi = 0; wk.right = 0;
  const int n_cig = min((int)r.n_cig(), View::kCapCig);
  // a soft-clipped read yields no variants at all and registers no novel ones (hisat2.py:681-684: returned before findVariantId)
  for (int c = 0; c < n_cig; ++c) wk.clipped |= (r.cig(c) & 15u) == GK_CIG_S;
  if (wk.clipped) return;
  bool last_is_event = false, last_novel = false;
  uint32_t last_pos = 0, last_len = 0;
  const uint32_t ref = r.ref(), pos0 = r.pos0();
  uint32_t cur = pos0;
  int mi = 0, ii = 0;
  const int n_mm = min((int)r.n_mm(), View::kCapMm);

  auto event = [&](uint32_t pos, uint32_t len, uint32_t typ, uint32_t val) {
    last_is_event = true;
    if (wk.n >= View::kCapEv) { wk.overflow = true; return; }
    if (ix.corr && typ == GK_TYP_SINGLE && (int)ref < ix.n_gene) {
      const int code = val == 'A' ? 0 : val == 'C' ? 1 : val == 'G' ? 2 : val == 'T' ? 3 : val == 'N' ? 4 : -1;
      const int64_t at = ix.gene_pos0[ref] + pos;
      if (code >= 0 && at < ix.gene_pos0[ref + 1]) {
        const uint8_t c = ix.corr[at * 5 + code];
        if (c) val = c;
      }
    }
    const uint64_t k = gk_make_key(ref, pos, typ, val);
    int i;
    bool known;
    const int base_code = val == 'A' ? 0 : val == 'C' ? 1 : val == 'G' ? 2 : val == 'T' ? 3 : -1;
    if (typ == GK_TYP_SINGLE && base_code >= 0 && (int)ref < ix.n_gene) {
      // a substitution to a plain base: its ordinal (or "not in the index") is ONE table read, not a bound look-up, a
      // short scan over the keys of the position and a key comparison -- three to four dependent L2 round trips
      const int tbase = ix.gene_pbase[ref], n_pos = ix.gene_pbase[ref + 1] - tbase;
      i = (int)pos < n_pos - 1 ? ix.snp_ord[(int64_t)(tbase + (int)pos) * 4 + base_code] : -1;
      known = i >= 0;
    } else {
      i = typ == GK_TYP_SINGLE ? lower_bound_snp(ix, ref, pos, k) : lower_bound_key(ix, ref, pos, k);
      known = i < ix.n_var && ix.key[i] == k;
    }
    const bool is_n = typ == GK_TYP_SINGLE && val == 'N';
    if (is_n) wk.any_n = 1;
    const uint32_t word = (known ? (uint32_t)i : (kEvNovel | (pos & 0xFFFFFFu))) | (is_n ? kEvIsN : 0u);
    if (!known && typ != GK_TYP_SINGLE) wk.drop = true;   // novel insertion / deletion: mate yields ([], [])
    uint32_t saved = (uint32_t)i;
    if (kEmit) {
      if (o_pos + wk.n < o_pos_end)
        ids[o_pos + wk.n] = known ? (uint32_t)i : (uint32_t)ix.n_var + novel_rank(nt, k);
    } else {
      if (!known) saved = kEvNovel | novel_insert(nt, k, ((uint64_t)m << 16) | (uint64_t)wk.n);
    }
    ev.put(wk.n, word, saved, is_n);
    last_pos = pos; last_len = len; last_novel = !known;
    wk.n++;
  };

  for (int c = 0; c < n_cig; ++c) {
    const uint32_t cg = r.cig(c);
    const uint32_t op = cg & 15u, len = cg >> 4;
    if (op == GK_CIG_M) {
      const uint32_t end = cur + len;
      uint32_t seg = cur;
      while (mi < n_mm && pos0 + r.mm_off(mi) < end) {
        const uint32_t p = pos0 + r.mm_off(mi);
        event(p, 1u, GK_TYP_SINGLE, r.mm_base(mi));
        seg = p + 1;
        ++mi;
      }
      if (seg < end) last_is_event = false;  // trailing match segment
      cur = end;
    } else if (op == GK_CIG_I) {
      event(cur, len, GK_TYP_INS, ii < View::kCapIns ? r.ins(ii) : 0u);
      ++ii;
    } else if (op == GK_CIG_D) {
      event(cur, len, GK_TYP_DEL, len);
      cur += len;
    }
  }
  // index records carry length 0, novel ones their walker length
  wk.right = (last_is_event && wk.n > 0) ? last_pos + (last_novel ? last_len : 0u) : cur;
  // getVariantsBoundary: [single 'A' at the left edge, single 'T' at the right edge)
  wk.lo = lower_bound_single(ix, ref, pos0, false);
  wk.hi = lower_bound_single(ix, ref, wk.right, true);
  wk.bad_window = wk.lo > wk.hi;
}

template <typename Ev>
__device__ inline bool negative_kept(uint64_t k, int i, const IndexView& ix, const Ev& ev, int n_ev,
                                     uint32_t any_n, uint32_t right) {
  const uint32_t typ = gk_key_typ(k), pos = gk_key_pos(k), val = gk_key_val(k);
  for (int e = 0; e < n_ev; ++e)
    if ((ev.get(e) & ~kEvIsN) == (uint32_t)i) return false;   // a positive of this mate
  if (any_n && typ == GK_TYP_SINGLE && (val == 'A' || val == 'C' || val == 'G' || val == 'T')) {
    for (int e = 0; e < n_ev; ++e) {
      const uint32_t w = ev.get(e);
      if (!(w & kEvIsN)) continue;
      if (ev.n_pos(w, ix) == pos) return false;             // the read says N here
    }
  }
  if (typ == GK_TYP_DEL && pos + val + 10u >= right) return false;
  return true;
}

// The negative lists of the 64 mates of a wavefront, enumerated TOGETHER: the windows [lo, hi) of the
// lanes are laid end to end (wave prefix sum) and the wave walks that sequence 64 candidates at a
// time, every lane taking one candidate of whichever mate owns it -- window lengths differ a lot
// between mates (0 .. ~100 variants), so a loop per lane would run every lane for the longest one.
// A candidate's owner is found by bisecting the prefix sums; the owner's event words, right edge and
// N flag are read from LDS.  kEmit: kept candidates are written to the owner's negative list in
// ascending ordinal order (rank inside the wave step + the owner's running count); otherwise they are
// only counted.  Returns the calling lane's own count.
struct WaveNeg {            // per wave, in LDS
  uint32_t incl[64];        // inclusive prefix sums of the window lengths
  uint32_t lo[64], right[64], n_ev[64], any_n[64];
  uint32_t kept[64];        // running count of kept candidates per owner lane
  uint32_t out0[64];        // kEmit: first slot of the owner's negative list
  uint32_t mask[64][kMaskWords];   // pass 1: bit b of an owner's words = candidate lo + b is kept
};

template <bool kEmit>
__device__ inline uint32_t cooperative_negatives(WaveNeg& wv, const IndexView& ix, const uint32_t* evs_wave,
                                                 uint32_t my_len, uint32_t my_lo, uint32_t my_right, uint32_t my_n,
                                                 uint32_t my_any_n, uint32_t my_out0, uint32_t* ids) {
  const int lane = threadIdx.x & 63;
  uint32_t incl = my_len;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t o = __shfl_up(incl, d, 64);
    if (lane >= d) incl += o;
  }
  wv.incl[lane] = incl; wv.lo[lane] = my_lo; wv.right[lane] = my_right; wv.n_ev[lane] = my_n;
  wv.any_n[lane] = my_any_n; wv.kept[lane] = 0; wv.out0[lane] = my_out0;
  if (!kEmit) {
#pragma unroll
    for (int w = 0; w < kMaskWords; ++w) wv.mask[lane][w] = 0;
  }
  const uint32_t total = __shfl(incl, 63, 64);
  __builtin_amdgcn_wave_barrier();
  for (uint32_t j = 0; j < total; j += 64) {
    const uint32_t c = j + lane;
    const bool active = c < total;
    // owner = first lane whose inclusive sum exceeds c
    int owner = 0;
    if (active) {
#pragma unroll
      for (int step = 32; step; step >>= 1)
        if (wv.incl[owner + step - 1] <= c) owner += step;
    }
    const uint32_t start = owner ? wv.incl[owner - 1] : 0u;
    bool keep = false;
    uint32_t i = 0;
    if (active) {
      i = wv.lo[owner] + (c - start);
      keep = negative_kept(ix.key[i], (int)i, ix, EvView{evs_wave + owner * kEvLd}, (int)wv.n_ev[owner],
                           wv.any_n[owner], wv.right[owner]);
    }
    if (!kEmit && keep && c - start < 32u * kMaskWords) atomicOr(&wv.mask[owner][(c - start) >> 5], 1u << ((c - start) & 31u));
    // candidates of one owner are a run of consecutive lanes
    const uint64_t kept_mask = __ballot(keep);
    const uint32_t first_lane = start > j ? start - j : 0u;          // first lane of my owner's run in this step
    const uint32_t end = wv.incl[owner] - j;                          // one past its last lane (may exceed 64)
    const uint64_t run = (end >= 64 ? ~0ull : ((1ull << end) - 1ull)) & ~((1ull << first_lane) - 1ull);
    const uint32_t before = wv.kept[owner];                           // read by every lane of the run ...
    __builtin_amdgcn_wave_barrier();
    if (kEmit && keep) ids[wv.out0[owner] + before + (uint32_t)__popcll(kept_mask & run & ((1ull << lane) - 1ull))] = i;
    if (active && (uint32_t)lane == first_lane) wv.kept[owner] = before + (uint32_t)__popcll(kept_mask & run);   // ... written by its first
    __builtin_amdgcn_wave_barrier();
  }
  return wv.kept[lane];
}

// Pass 1's form of the negative filter, 32 candidates at a time: a mate's window [lo, lo + len) starts as all
// ones; its own positives are cleared by ordinal (a handful of events), and only the DELETIONS of the window -- found
// through the index's static deletion bitmap -- are looked at one by one for the right-edge rule.  Substitutions to N
// (which exclude the four bases at their position) and windows beyond the saved 256 bits are rare and take the
// candidate-by-candidate loop.  The kept bits of the first 128 candidates land in `first` (registers), the rest in the
// mate's overflow row `more`; returns the kept count.
constexpr int kMaskRegs = 4;     // kept bits of the first 128 candidates ride in registers (one dense 16-byte store per mate)
__device__ inline uint32_t window_negatives(const IndexView& ix, const EvRow& ev, int n_ev, uint32_t any_n,
                                            uint32_t right, uint32_t lo, uint32_t len, uint32_t (&first)[kMaskRegs],
                                            uint32_t* more /* words kMaskRegs .. kMaskWords - 1 of the mate */) {
  auto keep_word = [&](uint32_t w, uint32_t word) {
    if (w == 0) first[0] = word; else if (w == 1) first[1] = word; else if (w == 2) first[2] = word;
    else if (w == 3) first[3] = word; else more[w - kMaskRegs] = word;
  };
  // the mate's event words come from `ev` (four registers + its overflow row); a word of kept bits is built in a register
  // and kept there (or stored once, beyond the fourth)
  if (len == 0) return 0;
  if (any_n || len > 32u * kMaskWords) {
    uint32_t kept = 0, cur = 0;
    for (uint32_t c = 0; c < len; ++c) {
      const uint32_t i = lo + c;
      if (negative_kept(ix.key[i], (int)i, ix, ev, n_ev, any_n, right)) {
        ++kept;
        cur |= 1u << (c & 31u);
      }
      if ((c & 31u) == 31u || c + 1 == len) {
        if (c < 32u * kMaskWords) keep_word(c >> 5, cur);
        cur = 0;
      }
    }
    return kept;
  }
  // the first events relative to the window, in registers (novel events carry bit 31: far outside any window)
  uint32_t rel4[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) rel4[q] = q < n_ev ? (ev.r[q] & ~kEvIsN) - lo : 0xFFFFFFFFu;
  const uint32_t n_words = (len + 31u) >> 5;
  uint32_t kept = 0;
  for (uint32_t w = 0; w < n_words; ++w) {
    const uint32_t left = len - 32u * w;
    uint32_t word = left >= 32u ? ~0u : ((1u << left) - 1u);
#pragma unroll
    for (int q = 0; q < 4; ++q) {                          // a positive of this mate is not a negative
      const uint32_t rel = rel4[q] - 32u * w;
      if (rel < 32u) word &= ~(1u << rel);
    }
    for (int e = 4; e < n_ev; ++e) {
      const uint32_t rel = (ev.more[e - kEvRegs] & ~kEvIsN) - lo - 32u * w;
      if (rel < 32u) word &= ~(1u << rel);
    }
    // deletions reaching within 10 bases of the right edge
    const uint32_t first = lo + 32u * w, sh = first & 31u;
    const uint32_t a = ix.del_bits[first >> 5], b = ix.del_bits[(first >> 5) + 1];
    uint32_t dels = (sh ? (a >> sh) | (b << (32u - sh)) : a) & word;
    while (dels) {
      const int bit = __ffs(dels) - 1;
      dels &= dels - 1;
      const uint64_t k = ix.key[first + (uint32_t)bit];
      if (gk_key_pos(k) + gk_key_val(k) + 10u >= right) word &= ~(1u << bit);
    }
    kept += (uint32_t)__popc(word);
    keep_word(w, word);
  }
  return kept;
}

// pass 1: validity, counts, novel registration.  One lane per mate; mates of a pair sit in
// adjacent lanes so the pair verdict is one lane shuffle.
__global__ __launch_bounds__(kThreads) void tab_count(const gk_mate* mates, int64_t n_mates, IndexView ix,
                                                      NovelTable nt, uint32_t* cnt /*[4*n_pairs+1]*/,
                                                      uint32_t* valid /*[n_pairs]*/, int* err_flags,
                                                      uint4* ev_save /*[n_mates]: events 0 .. 3*/,
                                                      uint32_t* ev_more /*[n_mates][kEvMore]: the rest*/,
                                                      uint32_t* lo_save /*[n_mates]*/,
                                                      uint4* mask_save /*[n_mates]: kept bits of candidates 0 .. 127*/,
                                                      uint32_t* mask_more /*[n_mates][kMaskWords - kMaskRegs]*/) {
  // LDS holds the heads of the staged records only (17 KB per workgroup); a mate's first four event words and the kept
  // bits of the first 128 candidates of its window ride in registers and leave as ONE dense 16-byte store each (a wave
  // writes 1 KB of full lines); what goes beyond (rare) lives in the mate's overflow rows
  __shared__ uint32_t rec[kThreads * kHeadLd];
  const int64_t m0 = (int64_t)blockIdx.x * kThreads;
  stage_heads(mates, m0, n_mates, rec);
  const int64_t m = m0 + threadIdx.x;
  const bool in = m < n_mates;
  const HeadView r{rec + threadIdx.x * kHeadLd, reinterpret_cast<const uint32_t*>(mates + (in ? m : 0))};
  EvRow ev{{0u, 0u, 0u, 0u}, ev_more + (in ? m : 0) * kEvMore, nt.keys};
  const bool ok = in && r.passes() && !r.spilled();   // wide pairs: tab_count_wide writes their counts afterwards
  const bool ok_other = __shfl_xor((int)ok, 1, 64) != 0;
  const bool pair_ok = ok && ok_other;
  const int64_t pair = m >> 1;
  const int side = (int)(m & 1);
  uint32_t n_pos = 0;
  Walked wk;
  wk.n = 0; wk.lo = wk.hi = 0; wk.right = 0; wk.any_n = 0;
  bool enumerate = false;
  if (pair_ok) {
    walk_mate<false>(r, ix, nt, m, ev, nullptr, 0, 0, wk);
    if (wk.overflow) atomicOr(err_flags, 2);
    if (wk.clipped) {
    } else if (wk.bad_window) {
      atomicOr(err_flags, 1);
    } else if (!wk.drop) {
      n_pos = wk.n;
      enumerate = true;
    }
  }
  // the kept bits of the window: what pass 2 needs to write the negative list without enumerating the window again
  uint32_t kept_bits[kMaskRegs] = {0u, 0u, 0u, 0u};
  const uint32_t n_neg = enumerate ? window_negatives(ix, ev, wk.n, wk.any_n, wk.right, (uint32_t)wk.lo,
                                                      (uint32_t)(wk.hi - wk.lo), kept_bits,
                                                      mask_more + (in ? m : 0) * (kMaskWords - kMaskRegs))
                                   : 0u;
  if (!in) return;
  cnt[4 * pair + side] = n_pos;
  cnt[4 * pair + 2 + side] = n_neg;
  if (side == 0) valid[pair] = pair_ok ? 1u : 0u;
  if (enumerate && (uint32_t)(wk.hi - wk.lo) > 32u * kMaskWords)
    atomicOr(err_flags, 4);   // does not fit the saved bits: the host takes the two-walk path
  // dense stores, every mate (pass 2 only reads the rows of mates that have lists)
  ev_save[m] = make_uint4(ev.r[0], ev.r[1], ev.r[2], ev.r[3]);
  mask_save[m] = make_uint4(kept_bits[0], kept_bits[1], kept_bits[2], kept_bits[3]);
  lo_save[m] = (uint32_t)wk.lo;
}

// pass 2 without a second walk: the lists are written from what pass 1 saved -- the positive list as
// ordinals / novel-table slots, the negative list as the window's first ordinal + the kept bits.
// The lists of the 32 pairs of a wavefront are one contiguous run of `ids` (offsets are in pair order), but a lane
// writing its own lists touches 64 different cache lines per store.  So the lanes assemble the run in LDS (scattered
// LDS writes are cheap) and the wave then copies it out in 256-byte rows; a run longer than the staging area (not
// seen on 150-bp reads: ~70 ids per pair) is written directly.
constexpr int kExpandThreads = 128;
constexpr int kExpandWaves = kExpandThreads / 64;
constexpr uint32_t kExpandCap = 4096;   // ids staged per wavefront (16 KB)

__global__ __launch_bounds__(kExpandThreads) void tab_expand(int64_t n_mates, int n_var, const uint32_t* rank,
                                                             const uint32_t* off, const uint32_t* valid,
                                                             const uint4* ev_save, const uint32_t* ev_more,
                                                             const uint32_t* lo_save, const uint4* mask_save,
                                                             const uint32_t* mask_more, uint32_t* ids) {
  __shared__ uint32_t stage[kExpandWaves][kExpandCap];
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t m0 = ((int64_t)blockIdx.x * kExpandWaves + wid) * 64;   // first mate of the wave (an even number: a pair's left mate)
  if (m0 >= n_mates) return;                                             // wave-uniform
  const int64_t m = m0 + lane;
  const int64_t pair0 = m0 >> 1, pair_end = min<int64_t>((m0 + 64) >> 1, n_mates >> 1);
  const uint32_t run0 = off[4 * pair0], run1 = off[4 * pair_end];      // wave-uniform: the run of the wave's pairs
  const bool staged = run1 - run0 <= kExpandCap;
  uint32_t* const mine = &stage[wid][0];
  auto put = [&](uint32_t slot, uint32_t v) {                          // list slot (a global offset) <- v
    if (staged) mine[slot - run0] = v;                                  // (wave-uniform choice; ds_write, in order with the reads below)
    else ids[slot] = v;
  };
  {
    // Everything a mate needs is requested at once, for every lane (a lane past the end repeats the last mate and writes
    // nothing): its pair's flag and five offsets, the saved event words, window start and kept bits; then the ranks of the
    // novel events among its first four -- two waits.  (With a test around each load the compiler waited for the flag,
    // then the offsets, then the events, then for every rank on its own: eight dependent waits per wave at ten waves per
    // CU.)  The empty asm statements keep the loads from being sunk into the branches that use their results.
    const int64_t mc = m < n_mates ? m : n_mates - 1;
    const int64_t pair = mc >> 1;
    const int side = (int)(mc & 1);
    const uint32_t flag = valid[pair];
    const uint4 o4 = *reinterpret_cast<const uint4*>(off + 4 * pair);
    const uint32_t o5 = off[4 * pair + 4];
    const uint4 e4 = ev_save[mc];
    const uint32_t lo = lo_save[mc];
    const uint4 m4 = mask_save[mc];
    asm volatile("" ::"v"(flag), "v"(o4.x), "v"(o5), "v"(e4.x), "v"(lo), "v"(m4.x));
    const uint32_t first[kEvRegs] = {e4.x, e4.y, e4.z, e4.w};
    uint32_t first_rank[kEvRegs];
#pragma unroll
    for (int e = 0; e < kEvRegs; ++e) first_rank[e] = rank[(first[e] & kEvNovel) ? (first[e] & kEvSlotMask) : 0u];
    asm volatile("" ::"v"(first_rank[0]), "v"(first_rank[1]), "v"(first_rank[2]), "v"(first_rank[3]));
    if (m < n_mates && flag == 1u) {      // 2 = a pair of the wide format: pass 1 saved nothing for it, tab_emit_wide writes its lists
      const uint32_t o_pos = side ? o4.y : o4.x, n_pos = (side ? o4.z : o4.y) - o_pos;
      const uint32_t o_neg = side ? o4.w : o4.z, n_neg = (side ? o5 : o4.w) - o_neg;
      for (uint32_t e = 0; e < n_pos; ++e) {
        if (e < (uint32_t)kEvRegs) {
          const uint32_t w = first[e] & ~kEvIsN;
          put(o_pos + e, (w & kEvNovel) ? (uint32_t)n_var + first_rank[e] : w);
        } else {
          const uint32_t w = ev_more[m * kEvMore + e - kEvRegs] & ~kEvIsN;
          put(o_pos + e, (w & kEvNovel) ? (uint32_t)n_var + rank[w & kEvSlotMask] : w);
        }
      }
      if (n_neg) {
        const uint32_t firstw[kMaskRegs] = {m4.x, m4.y, m4.z, m4.w};
        uint32_t j = 0;
        for (int w = 0; w < kMaskWords && j < n_neg; ++w) {
          uint32_t bits = w < kMaskRegs ? firstw[w] : mask_more[m * (kMaskWords - kMaskRegs) + w - kMaskRegs];
          while (bits) {
            const int b = __ffs(bits) - 1;
            bits &= bits - 1;
            put(o_neg + j++, lo + 32u * (uint32_t)w + (uint32_t)b);
          }
        }
      }
    }
  }
  if (!staged) return;
  __builtin_amdgcn_wave_barrier();   // LDS operations of a wave complete in order: the copy below sees the lists
  for (uint32_t i = lane; i < run1 - run0; i += 64) ids[run0 + i] = mine[i];
}

// Novel ranking by first appearance (hisat2.py:597-602: ids nv0, nv1, ... in the order the walk meets them): a novel
// variant's first appearance is (mate, event).  Every mate owns one word of event bits (a gk_mate has at most 22
// events); a mate of the wide format (up to 384 events) owns kWideWords words in a second, small array, found through
// the ascending list of wide pairs.  rank = first appearances in earlier mates (a scan over per-mate popcounts) +
// earlier first appearances inside the mate.  No dense (mate x event capacity) numbering: the sample size is not
// coupled to the widest record format present.
constexpr int kWideWords = (GK_WIDE_EVENTS + 31) / 32;

__device__ inline int64_t wide_index(const int64_t* spill_pair, int64_t n_spill, int64_t pair) {
  int64_t lo = 0, hi = n_spill;       // first k with spill_pair[k] >= pair
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (spill_pair[mid] < pair) lo = mid + 1; else hi = mid;
  }
  return lo < n_spill && spill_pair[lo] == pair ? lo : -1;
}

__global__ void novel_mark(NovelTable nt, uint32_t* mate_bits, uint32_t* wide_bits, const int64_t* spill_pair,
                           int64_t n_spill) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s > nt.mask) return;
  if (nt.keys[s] == kEmpty) return;
  const uint64_t q = nt.seq[s];
  const int64_t m = (int64_t)(q >> 16);
  const uint32_t e = (uint32_t)(q & 0xFFFFu);
  const int64_t w = n_spill ? wide_index(spill_pair, n_spill, m >> 1) : -1;
  if (w < 0) atomicOr(&mate_bits[m], 1u << (e & 31));      // a gk_mate: e < 22
  else atomicOr(&wide_bits[(2 * w + (m & 1)) * kWideWords + (e >> 5)], 1u << (e & 31));
}
// cnt[m] = first appearances in mate m
__global__ void novel_count(const uint32_t* mate_bits, uint32_t* cnt, int64_t n_mates) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_mates) cnt[i] = __popc(mate_bits[i]);
}
__global__ void novel_count_wide(const uint32_t* wide_bits, const int64_t* spill_pair, int64_t n_spill, uint32_t* cnt) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= 2 * n_spill) return;
  uint32_t c = 0;
  for (int w = 0; w < kWideWords; ++w) c += __popc(wide_bits[t * kWideWords + w]);
  cnt[2 * spill_pair[t >> 1] + (t & 1)] = c;
}
__global__ void novel_assign(NovelTable nt, const uint32_t* mate_bits, const uint32_t* wide_bits,
                             const int64_t* spill_pair, int64_t n_spill, const uint32_t* prefix, uint64_t* novel_key) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s > nt.mask) return;
  const uint64_t k = nt.keys[s];
  if (k == kEmpty) return;
  const uint64_t q = nt.seq[s];
  const int64_t m = (int64_t)(q >> 16);
  const uint32_t e = (uint32_t)(q & 0xFFFFu);
  const int64_t w = n_spill ? wide_index(spill_pair, n_spill, m >> 1) : -1;
  uint32_t rank = prefix[m];
  if (w < 0) {
    rank += __popc(mate_bits[m] & ((1u << (e & 31)) - 1u));
  } else {
    const uint32_t* bits = wide_bits + (2 * w + (m & 1)) * kWideWords;
    for (uint32_t j = 0; j < (e >> 5); ++j) rank += __popc(bits[j]);
    rank += __popc(bits[e >> 5] & ((1u << (e & 31)) - 1u));
  }
  nt.rank[s] = rank;
  novel_key[rank] = k;
}

// pass 2: emit ordinals at the scanned offsets
__global__ __launch_bounds__(kThreads) void tab_emit(const gk_mate* mates, int64_t n_mates, IndexView ix,
                                                     NovelTable nt, const uint32_t* off, const uint32_t* valid,
                                                     uint32_t* ids) {
  __shared__ uint32_t rec[kThreads * kRecLd];
  __shared__ uint32_t evs[kThreads * kEvLd];
  __shared__ WaveNeg wneg[kThreads / 64];
  const int64_t m0 = (int64_t)blockIdx.x * kThreads;
  stage_mates(mates, m0, n_mates, rec);
  const int64_t m = m0 + threadIdx.x;
  const int64_t pair = m >> 1;
  const int side = (int)(m & 1);
  uint32_t o_pos = 0, o_pos_end = 0, o_neg = 0, o_neg_end = 0;
  const MateView r{rec + threadIdx.x * kRecLd};
  if (m < n_mates && valid[pair] && !r.spilled()) {   // wide pairs: tab_emit_wide writes their lists
    o_pos = off[4 * pair + side]; o_pos_end = off[4 * pair + side + 1];
    o_neg = off[4 * pair + 2 + side]; o_neg_end = off[4 * pair + 2 + side + 1];
  }
  uint32_t* evw = evs + threadIdx.x * kEvLd;
  Walked wk;
  wk.n = 0; wk.lo = wk.hi = 0; wk.right = 0; wk.any_n = 0;
  EvPtr ev{evw, nullptr};
  if (o_pos != o_pos_end || o_neg != o_neg_end) walk_mate<true>(r, ix, nt, m, ev, ids, o_pos, o_pos_end, wk);
  const int wid = threadIdx.x >> 6;
  // a mate with an empty negative list has nothing to enumerate (its window may still be non-empty)
  cooperative_negatives<true>(wneg[wid], ix, evs + wid * 64 * kEvLd, o_neg != o_neg_end ? (uint32_t)(wk.hi - wk.lo) : 0u,
                              (uint32_t)wk.lo, wk.right, (uint32_t)wk.n, wk.any_n, o_neg, ids);
}

// The pairs of the wide format, one lane per mate (mates of a pair in adjacent lanes), records and event words in
// global memory, windows enumerated candidate by candidate: the same walk and the same rules as above, without the
// staging -- there are a handful of such pairs in a sample, if any.
__global__ __launch_bounds__(64) void tab_count_wide(const gk_mate_wide* wide, const int64_t* spill_pair, int64_t n_spill,
                                                     IndexView ix, NovelTable nt, uint32_t* cnt,
                                                     uint32_t* valid, int* err_flags, uint32_t* evw_all) {
  const int64_t t = (int64_t)blockIdx.x * 64 + threadIdx.x;
  const bool in = t < 2 * n_spill;
  const WideView r{wide + (in ? t : 0)};
  const bool ok = in && r.passes();
  const bool ok_other = __shfl_xor((int)ok, 1, 64) != 0;
  const bool pair_ok = ok && ok_other;
  if (!in) return;
  const int64_t pair = spill_pair[t >> 1];
  const int side = (int)(t & 1);
  uint32_t* evw = evw_all + t * GK_WIDE_EVENTS;
  uint32_t n_pos = 0, n_neg = 0;
  if (pair_ok) {
    Walked wk;
    EvPtr ev{evw, nullptr};
    walk_mate<false>(r, ix, nt, 2 * pair + side, ev, nullptr, 0, 0, wk);
    if (wk.overflow) atomicOr(err_flags, 2);
    if (wk.clipped) {
    } else if (wk.bad_window) {
      atomicOr(err_flags, 1);
    } else if (!wk.drop) {
      n_pos = (uint32_t)wk.n;
      for (int i = wk.lo; i < wk.hi; ++i)
        n_neg += negative_kept(ix.key[i], i, ix, EvView{evw}, wk.n, wk.any_n, wk.right) ? 1u : 0u;
    }
  }
  cnt[4 * pair + side] = n_pos;
  cnt[4 * pair + 2 + side] = n_neg;
  if (side == 0) valid[pair] = pair_ok ? 2u : 0u;   // 2: a valid pair whose lists tab_emit_wide writes (tab_expand skips it)
}

__global__ __launch_bounds__(64) void tab_emit_wide(const gk_mate_wide* wide, const int64_t* spill_pair, int64_t n_spill,
                                                    IndexView ix, NovelTable nt, const uint32_t* off,
                                                    const uint32_t* valid, uint32_t* ids, uint32_t* evw_all) {
  const int64_t t = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (t >= 2 * n_spill) return;
  const int64_t pair = spill_pair[t >> 1];
  const int side = (int)(t & 1);
  if (!valid[pair]) return;
  const uint32_t o_pos = off[4 * pair + side], o_pos_end = off[4 * pair + side + 1];
  const uint32_t o_neg = off[4 * pair + 2 + side], o_neg_end = off[4 * pair + 2 + side + 1];
  if (o_pos == o_pos_end && o_neg == o_neg_end) return;
  const WideView r{wide + t};
  uint32_t* evw = evw_all + t * GK_WIDE_EVENTS;
  Walked wk;
  EvPtr ev{evw, nullptr};
  walk_mate<true>(r, ix, nt, 2 * pair + side, ev, ids, o_pos, o_pos_end, wk);
  uint32_t at = o_neg;
  for (int i = wk.lo; i < wk.hi && at < o_neg_end; ++i)
    if (negative_kept(ix.key[i], i, ix, EvView{evw}, wk.n, wk.any_n, wk.right)) ids[at++] = (uint32_t)i;
}

__global__ __launch_bounds__(kThreads) void gather_pairs(const gk_mate* mates, const int32_t* pair_src, int64_t n_valid,
                                                         const uint32_t* off_in, uint32_t* off_out, uint8_t* gene,
                                                         uint8_t* nh, uint32_t total) {
  const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (i > n_valid) return;
  if (i == n_valid) {
    off_out[4 * n_valid] = total;
    return;
  }
  const int64_t p = pair_src[i];
#pragma unroll
  for (int k = 0; k < 4; ++k) off_out[4 * i + k] = off_in[4 * p + k];
  const uint32_t w1 = reinterpret_cast<const uint32_t*>(mates + 2 * p)[1];
  gene[i] = (uint8_t)((w1 >> 16) & 0xFFu);
  nh[i] = (uint8_t)(w1 >> 24);
}

// ---- the records of a sample in a compact form (gk_mates_compact / gk_mates_expand): per mate only the words it uses
// -- the 12-byte header, its CIGAR operations, its mismatches, its inserted-string ids -- behind a table of word offsets.
// A 150-base mate with one CIGAR operation and three mismatches takes 7 words + its offset instead of 32.
__device__ inline int mate_used_words(const uint32_t* w, int* n_cw, int* n_mm, int* n_ins) {
  const uint32_t h = w[2];
  const int n_cig = (int)((h >> 8) & 0xFFu);
  const bool spilled = n_cig == GK_SPILLED;       // header + ins[0] = the pair's place in the wide array
  *n_cw = spilled ? 0 : (min(n_cig, GK_MAX_CIG) + 1) / 2;
  *n_mm = spilled ? 0 : min((int)((h >> 16) & 0xFFu), GK_MAX_MM);
  *n_ins = spilled ? 1 : min((int)(h >> 24), GK_MAX_INS);
  return 3 + *n_cw + *n_mm + *n_ins;
}

__global__ __launch_bounds__(kThreads) void mates_count_words(const gk_mate* mates, int64_t n, uint32_t* cnt) {
  const int64_t m = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (m >= n) return;
  int a, b, c;
  cnt[m] = (uint32_t)mate_used_words(reinterpret_cast<const uint32_t*>(mates + m), &a, &b, &c);
}

__global__ __launch_bounds__(kThreads) void mates_pack_words(const gk_mate* mates, int64_t n, const uint32_t* off,
                                                             uint32_t* off_out, uint32_t* words, uint32_t total) {
  const int64_t m = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (m > n) return;
  if (m == n) { off_out[n] = total; return; }
  const uint32_t* w = reinterpret_cast<const uint32_t*>(mates + m);
  int n_cw, n_mm, n_ins;
  mate_used_words(w, &n_cw, &n_mm, &n_ins);
  uint32_t* d = words + off[m];
  off_out[m] = off[m];
  d[0] = w[0]; d[1] = w[1]; d[2] = w[2];
  d += 3;
  for (int i = 0; i < n_cw; ++i) d[i] = w[kCigWord + i];
  d += n_cw;
  for (int i = 0; i < n_mm; ++i) d[i] = w[kMmWord + i];
  d += n_mm;
  for (int i = 0; i < n_ins; ++i) d[i] = w[kInsWord + i];
}

__global__ __launch_bounds__(kThreads) void mates_unpack_words(const uint32_t* off, const uint32_t* words, int64_t n,
                                                               gk_mate* mates) {
  const int64_t m = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (m >= n) return;
  const uint32_t* s = words + off[m];
  uint32_t rec[kMateWords];
#pragma unroll
  for (int i = 0; i < kMateWords; ++i) rec[i] = 0u;
  rec[0] = s[0]; rec[1] = s[1]; rec[2] = s[2];
  int n_cw, n_mm, n_ins;
  mate_used_words(rec, &n_cw, &n_mm, &n_ins);
  uint32_t* d = reinterpret_cast<uint32_t*>(mates + m);
  uint4* d4 = reinterpret_cast<uint4*>(d);
#pragma unroll
  for (int i = 0; i < kMateWords / 4; ++i) d4[i] = make_uint4(0u, 0u, 0u, 0u);
  d[0] = s[0]; d[1] = s[1]; d[2] = s[2];
  s += 3;
  for (int i = 0; i < n_cw; ++i) d[kCigWord + i] = s[i];
  s += n_cw;
  for (int i = 0; i < n_mm; ++i) d[kMmWord + i] = s[i];
  s += n_mm;
  for (int i = 0; i < n_ins; ++i) d[kInsWord + i] = s[i];
}

inline unsigned nblk(int64_t n, int t = kThreads) { return (unsigned)((n + t - 1) / t); }

}  // namespace

extern "C" {

int gk_index_create(gk_ctx* ctx, const uint64_t* key, int32_t n_var, const int32_t* gene_vbeg, int32_t n_gene,
                    gk_index** out) {
  gk_bind(ctx);
  GK_REQUIRE(ctx && out && gene_vbeg && n_gene > 0 && n_gene < 255 && n_var >= 0 && n_var < (1 << 26),
             "bad index arguments");
  GK_REQUIRE(gene_vbeg[0] == 0 && gene_vbeg[n_gene] == n_var, "gene_vbeg must cover the key table");
  for (int i = 1; i < n_var; ++i) GK_REQUIRE(key[i - 1] < key[i], "index keys must be strictly increasing");
  gk_index* idx = new gk_index();
  idx->ctx = ctx; idx->n_var = n_var; idx->n_gene = n_gene;
  idx->gene_vbeg.assign(gene_vbeg, gene_vbeg + n_gene + 1);
  // 16-bp bucket table per gene: bucket b holds the first ordinal at or after (gene, 16 * b); one
  // extra bucket past the last variant position closes the last interval
  std::vector<int32_t> boff((size_t)n_gene + 1, 0), bucket;
  for (int g = 0; g < n_gene; ++g) {
    const uint64_t* first = key + gene_vbeg[g];
    const uint64_t* last = key + gene_vbeg[g + 1];
    for (const uint64_t* k = first; k < last; ++k)
      GK_REQUIRE((uint32_t)(*k >> GK_KEY_REF_SHIFT) == (uint32_t)g, "gene_vbeg does not match the key table");
    const uint32_t max_pos = first < last ? gk_key_pos(*(last - 1)) : 0u;
    const int nb = (int)(max_pos >> 4) + 2;
    for (int b = 0; b < nb; ++b)
      bucket.push_back((int32_t)(std::lower_bound(key, key + n_var, gk_make_key((uint32_t)g, (uint32_t)b << 4, 0, 0)) - key));
    boff[g + 1] = (int32_t)bucket.size();
  }
  GK_HIP(gk_pool_malloc(ctx, (void**)&idx->d_key, (size_t)(n_var + 1) * sizeof(uint64_t)));
  GK_HIP(gk_pool_malloc(ctx, (void**)&idx->d_gene_vbeg, (size_t)(n_gene + 1) * sizeof(int32_t)));
  GK_HIP(gk_pool_malloc(ctx, (void**)&idx->d_bucket, bucket.size() * sizeof(int32_t)));
  GK_HIP(gk_pool_malloc(ctx, (void**)&idx->d_gene_boff, boff.size() * sizeof(int32_t)));
  GK_HIP(hipMemcpyAsync(idx->d_key, key, (size_t)n_var * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
  GK_HIP(hipMemcpyAsync(idx->d_gene_vbeg, gene_vbeg, (size_t)(n_gene + 1) * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
  GK_HIP(hipMemcpyAsync(idx->d_bucket, bucket.data(), bucket.size() * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
  GK_HIP(hipMemcpyAsync(idx->d_gene_boff, boff.data(), boff.size() * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
  // per-position bounds: gene g has (last variant position + 2) entries, the last one = the gene's end ordinal
  std::vector<int32_t> pbase((size_t)n_gene + 1, 0), lb_a, lb_t, snp_ord;
  for (int g = 0; g < n_gene; ++g) {
    const int32_t v0 = gene_vbeg[g], v1 = gene_vbeg[g + 1];
    const uint32_t n_pos = v1 > v0 ? gk_key_pos(key[v1 - 1]) + 2u : 1u;
    pbase[g + 1] = pbase[g] + (int32_t)n_pos;
    int32_t ia = v0, it = v0;
    for (uint32_t p = 0; p + 1 < n_pos; ++p) {
      const uint64_t ka = gk_make_key((uint32_t)g, p, GK_TYP_SINGLE, 'A'), kt = gk_make_key((uint32_t)g, p, GK_TYP_SINGLE, 'T');
      while (ia < v1 && key[ia] < ka) ++ia;
      while (it < v1 && key[it] < kt) ++it;
      lb_a.push_back(ia);
      lb_t.push_back(it);
      for (const char base : {'A', 'C', 'G', 'T'}) {     // the substitutions listed at p lie between the two bounds
        const uint64_t kb = gk_make_key((uint32_t)g, p, GK_TYP_SINGLE, (uint32_t)base);
        int32_t at = -1;
        for (int32_t v = ia; v < v1 && key[v] <= kb; ++v)
          if (key[v] == kb) at = v;
        snp_ord.push_back(at);
      }
    }
    lb_a.push_back(v1);
    lb_t.push_back(v1);
    for (int q = 0; q < 4; ++q) snp_ord.push_back(-1);
  }
  GK_HIP(gk_pool_malloc(ctx, (void**)&idx->d_snp_ord, snp_ord.size() * sizeof(int32_t)));
  GK_HIP(hipMemcpyAsync(idx->d_snp_ord, snp_ord.data(), snp_ord.size() * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
  GK_HIP(gk_pool_malloc(ctx, (void**)&idx->d_lb_a, lb_a.size() * sizeof(int32_t)));
  GK_HIP(gk_pool_malloc(ctx, (void**)&idx->d_lb_t, lb_t.size() * sizeof(int32_t)));
  GK_HIP(gk_pool_malloc(ctx, (void**)&idx->d_gene_pbase, pbase.size() * sizeof(int32_t)));
  GK_HIP(hipMemcpyAsync(idx->d_lb_a, lb_a.data(), lb_a.size() * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
  GK_HIP(hipMemcpyAsync(idx->d_lb_t, lb_t.data(), lb_t.size() * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
  GK_HIP(hipMemcpyAsync(idx->d_gene_pbase, pbase.data(), pbase.size() * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
  std::vector<uint32_t> del_bits((size_t)(n_var + 31) / 32 + 2, 0u);
  for (int32_t v = 0; v < n_var; ++v)
    if (gk_key_typ(key[v]) == GK_TYP_DEL) del_bits[(size_t)v >> 5] |= 1u << (v & 31);
  GK_HIP(gk_pool_malloc(ctx, (void**)&idx->d_del_bits, del_bits.size() * sizeof(uint32_t)));
  GK_HIP(hipMemcpyAsync(idx->d_del_bits, del_bits.data(), del_bits.size() * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
  GK_HIP(hipStreamSynchronize(ctx->stream));
  *out = idx;
  return GK_OK;
}

int gk_index_destroy(gk_index* idx) {
  gk_bind(idx ? idx->ctx : nullptr);
  if (!idx) return GK_OK;
  gk_ctx* ctx = idx->ctx;
  gk_pool_free(ctx,idx->d_key);
  gk_pool_free(ctx,idx->d_gene_vbeg);
  gk_pool_free(ctx,idx->d_bucket);
  gk_pool_free(ctx,idx->d_gene_boff);
  gk_pool_free(ctx,idx->d_del_bits);
  gk_pool_free(ctx,idx->d_lb_a);
  gk_pool_free(ctx,idx->d_lb_t);
  gk_pool_free(ctx,idx->d_gene_pbase);
  gk_pool_free(ctx,idx->d_snp_ord);
  delete idx;
  return GK_OK;
}

int gk_tabulate(gk_ctx* ctx, gk_index* idx, gk_dptr d_mates_p, int64_t n_pairs, gk_tab** out) {
  return gk_tabulate_corrected(ctx, idx, d_mates_p, n_pairs, 0, 0, out);
}

int gk_tabulate_corrected(gk_ctx* ctx, gk_index* idx, gk_dptr d_mates_p, int64_t n_pairs, gk_dptr d_corr,
                          gk_dptr d_gene_pos0, gk_tab** out) {
  return gk_tabulate_spilled(ctx, idx, d_mates_p, n_pairs, d_corr, d_gene_pos0, nullptr, nullptr, 0, out);
}

static int tabulate_with_table(gk_ctx* ctx, gk_index* idx, gk_dptr d_mates_p, int64_t n_pairs, gk_dptr d_corr,
                               gk_dptr d_gene_pos0, const gk_mate_wide* wide, const int64_t* spill_pair,
                               int64_t n_spill, uint32_t log2cap, bool* table_too_small, gk_tab** out);

int gk_tabulate_spilled(gk_ctx* ctx, gk_index* idx, gk_dptr d_mates_p, int64_t n_pairs, gk_dptr d_corr,
                        gk_dptr d_gene_pos0, const gk_mate_wide* wide, const int64_t* spill_pair, int64_t n_spill,
                        gk_tab** out) {
  // Hash table of the novel variants: a slot per mate to begin with, 2^22 at most (distinct novel variants are few -- read errors
  // repeat, positions are finite -- while the worst case, every event of every mate novel, would need 44 slots per
  // mate: 2^24 slots, i.e. 200 MB to clear and three passes over them per sample).  A sample that fills half of it
  // is tabulated again with a table eight times the size.
  uint32_t log2cap = 16;
  while ((1ull << log2cap) < (uint64_t)(2 * std::max<int64_t>(n_pairs, 0)) && log2cap < 22) ++log2cap;   // at most 4 M slots to begin with
  log2cap = (uint32_t)std::min<long>(30, std::max<long>(4, gk_test_hook_value("novel_log2cap", (long)log2cap)));   // tests: force the retries
  uint32_t log2max = 16;
  while ((1ull << log2max) < (uint64_t)(2 * std::max<int64_t>(n_pairs, 0)) * GK_WIDE_EVENTS * 2 && log2max < 30) ++log2max;   // a slot number is 30 bits of an event word
  for (;;) {
    bool too_small = false;
    const int rc = tabulate_with_table(ctx, idx, d_mates_p, n_pairs, d_corr, d_gene_pos0, wide, spill_pair, n_spill, log2cap,
                                       &too_small, out);
    if (!too_small) return rc;
    if (log2cap >= log2max) return rc;
    log2cap = std::min(log2cap + 3, log2max);
  }
}

static int tabulate_with_table(gk_ctx* ctx, gk_index* idx, gk_dptr d_mates_p, int64_t n_pairs, gk_dptr d_corr,
                               gk_dptr d_gene_pos0, const gk_mate_wide* wide, const int64_t* spill_pair,
                               int64_t n_spill, uint32_t log2cap, bool* table_too_small, gk_tab** out) {
  gk_bind(ctx);
  GK_REQUIRE(ctx && idx && out && n_pairs >= 0, "bad tabulate arguments");
  GK_REQUIRE(n_spill >= 0 && n_spill <= n_pairs && (n_spill == 0 || (wide && spill_pair)), "bad wide-pair arguments");
  for (int64_t k = 0; k < n_spill; ++k)
    GK_REQUIRE(spill_pair[k] >= 0 && spill_pair[k] < n_pairs && (k == 0 || spill_pair[k - 1] < spill_pair[k]),
               "wide pairs must name pairs of the sample, ascending");
  GK_REQUIRE((d_corr == 0) == (d_gene_pos0 == 0), "correction table and position offsets come together");
  GK_REQUIRE(n_pairs < (1ll << 26), "more than 2^26 pairs per call");
  const gk_mate* mates = gk_ptr<const gk_mate>(d_mates_p);
  const int64_t n_mates = 2 * n_pairs;
  hipStream_t st = ctx->stream;
  gk_tab* tab = new gk_tab();
  tab->ctx = ctx; tab->idx = idx; tab->n_pairs = n_pairs; tab->n_var = idx->n_var;

  // novel hash table of 2^log2cap slots, load factor <= 0.5 (checked below)
  NovelTable nt;
  const size_t cap = 1ull << log2cap;
  nt.mask = (uint32_t)(cap - 1);
  GK_HIP(gk_pool_malloc(ctx, (void**)&nt.keys, cap * sizeof(uint64_t)));
  GK_HIP(gk_pool_malloc(ctx, (void**)&nt.seq, cap * sizeof(uint64_t)));
  GK_HIP(gk_pool_malloc(ctx, (void**)&nt.rank, cap * sizeof(uint32_t)));
  GK_HIP(hipMemsetAsync(nt.keys, 0xFF, cap * sizeof(uint64_t), st));
  GK_HIP(hipMemsetAsync(nt.seq, 0xFF, cap * sizeof(uint64_t), st));

  uint32_t *cnt = nullptr, *valid = nullptr, *lo_save = nullptr, *ev_more = nullptr, *mask_more = nullptr;
  uint4 *ev_save = nullptr, *mask_save = nullptr;
  int* d_err = nullptr;
  // what pass 1 saves for pass 2: per mate 16 dense bytes of event words, 16 of kept bits, the window's first ordinal, and
  // overflow rows that a mate with more than four events / a window beyond 128 candidates writes
  GK_HIP(gk_pool_malloc(ctx, (void**)&ev_save, (size_t)(n_mates + 1) * sizeof(uint4)));
  GK_HIP(gk_pool_malloc(ctx, (void**)&ev_more, (size_t)(n_mates + 1) * kEvMore * sizeof(uint32_t)));
  GK_HIP(gk_pool_malloc(ctx, (void**)&lo_save, (size_t)(n_mates + 1) * sizeof(uint32_t)));
  GK_HIP(gk_pool_malloc(ctx, (void**)&mask_save, (size_t)(n_mates + 1) * sizeof(uint4)));
  GK_HIP(gk_pool_malloc(ctx, (void**)&mask_more, (size_t)(n_mates + 1) * (kMaskWords - kMaskRegs) * sizeof(uint32_t)));
  GK_HIP(gk_pool_malloc(ctx, (void**)&cnt, (size_t)(4 * n_pairs + 2) * sizeof(uint32_t)));
  GK_HIP(gk_pool_malloc(ctx, (void**)&valid, (size_t)(n_pairs + 1) * sizeof(uint32_t)));
  GK_HIP(gk_pool_malloc(ctx, (void**)&d_err, sizeof(int)));
  GK_HIP(hipMemsetAsync(d_err, 0, sizeof(int), st));
  nt.flags = d_err;

  const IndexView ix{idx->d_key, idx->d_bucket, idx->d_gene_boff, idx->n_var, idx->n_gene,
                     gk_ptr<uint8_t>(d_corr), gk_ptr<int64_t>(d_gene_pos0), idx->d_del_bits, idx->d_lb_a, idx->d_lb_t,
                     idx->d_gene_pbase, idx->d_snp_ord};
  if (n_mates) {
    GK_PROF(ctx, "tab_count", GK_KERNEL(tab_count, dim3(nblk(n_mates)), dim3(kThreads), 0, st, mates, n_mates, ix,
                       nt, cnt, valid, d_err, ev_save, ev_more, lo_save, mask_save, mask_more));
  }
  int64_t* d_spill_pair = nullptr;
  uint32_t* wide_ev = nullptr;
  if (n_spill) {   // the wide pairs overwrite the zeros the kernel above left for them
    GK_HIP(gk_pool_malloc(ctx, (void**)&tab->d_wide, (size_t)(2 * n_spill) * sizeof(gk_mate_wide)));
    GK_HIP(gk_pool_malloc(ctx, (void**)&d_spill_pair, (size_t)n_spill * sizeof(int64_t)));
    GK_HIP(gk_pool_malloc(ctx, (void**)&wide_ev, (size_t)(2 * n_spill) * GK_WIDE_EVENTS * sizeof(uint32_t)));
    GK_HIP(hipMemcpyAsync(tab->d_wide, wide, (size_t)(2 * n_spill) * sizeof(gk_mate_wide), hipMemcpyHostToDevice, st));
    GK_HIP(hipMemcpyAsync(d_spill_pair, spill_pair, (size_t)n_spill * sizeof(int64_t), hipMemcpyHostToDevice, st));
    GK_HIP(hipStreamSynchronize(st));   // the caller's arrays are free again
    tab->n_spill = n_spill;
    GK_KERNEL(tab_count_wide, dim3(nblk(2 * n_spill, 64)), dim3(64), 0, st, tab->d_wide, d_spill_pair, n_spill, ix, nt,
              cnt, valid, d_err, wide_ev);
  }
  // offsets over input pairs (invalid pairs contribute zeros)
  int rc = gk_scan_u32(ctx, cnt, 4 * n_pairs, cnt + 4 * n_pairs);
  if (rc) return rc;

  // novel ranks: event bits per mate (bitmap) + per wide mate (wide_bits), per-mate counts scanned into `prefix`
  const int64_t n_words = n_mates + 1;
  uint32_t *bitmap = nullptr, *prefix = nullptr, *wide_bits = nullptr;
  GK_HIP(gk_pool_malloc(ctx, (void**)&bitmap, (size_t)n_words * sizeof(uint32_t)));
  GK_HIP(gk_pool_malloc(ctx, (void**)&prefix, (size_t)(n_words + 1) * sizeof(uint32_t)));
  GK_HIP(gk_pool_malloc(ctx, (void**)&wide_bits, (size_t)(2 * n_spill + 1) * kWideWords * sizeof(uint32_t)));
  GK_HIP(hipMemsetAsync(bitmap, 0, (size_t)n_words * sizeof(uint32_t), st));
  GK_HIP(hipMemsetAsync(wide_bits, 0, (size_t)(2 * n_spill + 1) * kWideWords * sizeof(uint32_t), st));
  GK_PROF(ctx, "novel_mark", GK_KERNEL(novel_mark, dim3(nblk((int64_t)cap)), dim3(kThreads), 0, st, nt, bitmap, wide_bits,
                                     d_spill_pair, n_spill));
  GK_PROF(ctx, "novel_count", GK_KERNEL(novel_count, dim3(nblk(n_words)), dim3(kThreads), 0, st, bitmap, prefix, n_words));
  if (n_spill)
    GK_KERNEL(novel_count_wide, dim3(nblk(2 * n_spill)), dim3(kThreads), 0, st, wide_bits, d_spill_pair, n_spill, prefix);
  rc = gk_scan_u32(ctx, prefix, n_words, prefix + n_words);
  if (rc) return rc;

  uint32_t totals[2] = {0, 0};
  int err = 0;
  GK_HIP(gk_fetch_queue(ctx, &totals[0], cnt + 4 * n_pairs, sizeof(uint32_t)));
  GK_HIP(gk_fetch_queue(ctx, &totals[1], prefix + n_words, sizeof(uint32_t)));
  GK_HIP(gk_fetch_queue(ctx, &err, d_err, sizeof(int)));
  GK_HIP(gk_fetch_wait(ctx));
  tab->n_ids = totals[0];
  tab->n_novel = (int32_t)totals[1];
  tab->err_flags = err & 3;   // bit 2 (a window beyond the saved bits) only selects the pass-2 kernel
  if ((err & 8) || (uint64_t)tab->n_novel * 2 > cap) {   // (a full table stops taking keys: the count is then a lower bound)
    gk_set_error("novel variant table overflow (%d novel variants in %llu slots)", tab->n_novel, (unsigned long long)cap);
    *table_too_small = true;
    GK_HIP(hipStreamSynchronize(st));
    gk_pool_free(ctx,cnt); gk_pool_free(ctx,valid); gk_pool_free(ctx,d_err); gk_pool_free(ctx,bitmap); gk_pool_free(ctx,prefix); gk_pool_free(ctx,wide_bits);
    gk_pool_free(ctx,ev_save); gk_pool_free(ctx,lo_save); gk_pool_free(ctx,mask_save); gk_pool_free(ctx,ev_more); gk_pool_free(ctx,mask_more);
    gk_pool_free(ctx,d_spill_pair); gk_pool_free(ctx,wide_ev);
    gk_pool_free(ctx,nt.keys); gk_pool_free(ctx,nt.seq); gk_pool_free(ctx,nt.rank);
    gk_tab_destroy(tab);
    return GK_ERR_CAPACITY;
  }

  GK_HIP(gk_pool_malloc(ctx, (void**)&tab->d_novel_key, (size_t)(tab->n_novel + 1) * sizeof(uint64_t)));
  GK_PROF(ctx, "novel_assign", GK_KERNEL(novel_assign, dim3(nblk((int64_t)cap)), dim3(kThreads), 0, st, nt, bitmap, wide_bits,
                     d_spill_pair, n_spill, prefix, tab->d_novel_key));

  GK_HIP(gk_pool_malloc(ctx, (void**)&tab->d_ids, (size_t)(tab->n_ids + 1) * sizeof(uint32_t)));
  if (n_mates) {
    // pass 2: from what pass 1 saved; the second walk only when some window did not fit the saved bits
    const bool two_walks = gk_test_hook("two_walks");   // tests: the second walk for every sample
    if ((err & 4) || two_walks) {
      GK_PROF(ctx, "tab_emit", GK_KERNEL(tab_emit, dim3(nblk(n_mates)), dim3(kThreads), 0, st, mates, n_mates, ix, nt,
                         cnt, valid, tab->d_ids));
    } else {
      GK_PROF(ctx, "tab_expand", GK_KERNEL(tab_expand, dim3(nblk(n_mates, kExpandThreads)), dim3(kExpandThreads), 0, st, n_mates, idx->n_var,
                         nt.rank, cnt, valid, ev_save, ev_more, lo_save, mask_save, mask_more, tab->d_ids));
    }
    // the pairs of the wide format, AFTER the kernel above (tab_expand copies a wavefront's whole run of lists out of LDS,
    // the slots of a wide pair included; tab_emit leaves them alone): their lists overwrite whatever lies there
    if (n_spill)
      GK_KERNEL(tab_emit_wide, dim3(nblk(2 * n_spill, 64)), dim3(64), 0, st, tab->d_wide, d_spill_pair, n_spill, ix, nt,
                cnt, valid, tab->d_ids, wide_ev);
  }
  // compact valid pairs (order preserving)
  GK_HIP(gk_pool_malloc(ctx, (void**)&tab->d_pair_src, (size_t)(n_pairs + 1) * sizeof(int32_t)));
  rc = gk_compact(ctx, valid, nullptr, n_pairs, tab->d_pair_src, &tab->n_valid);
  if (rc) return rc;
  GK_HIP(gk_pool_malloc(ctx, (void**)&tab->d_off, (size_t)(4 * tab->n_valid + 1) * sizeof(uint32_t)));
  GK_HIP(gk_pool_malloc(ctx, (void**)&tab->d_pair_gene, (size_t)tab->n_valid + 1));
  GK_HIP(gk_pool_malloc(ctx, (void**)&tab->d_pair_nh, (size_t)tab->n_valid + 1));
  GK_PROF(ctx, "gather_pairs", GK_KERNEL(gather_pairs, dim3(nblk(tab->n_valid + 1)), dim3(kThreads), 0, st, mates, tab->d_pair_src,
                     tab->n_valid, cnt, tab->d_off, tab->d_pair_gene, tab->d_pair_nh, (uint32_t)tab->n_ids));
  GK_HIP(hipGetLastError());
  GK_HIP(hipStreamSynchronize(st));
  gk_pool_free(ctx,cnt); gk_pool_free(ctx,valid); gk_pool_free(ctx,d_err); gk_pool_free(ctx,bitmap); gk_pool_free(ctx,prefix); gk_pool_free(ctx,wide_bits);
  gk_pool_free(ctx,ev_save); gk_pool_free(ctx,lo_save); gk_pool_free(ctx,mask_save); gk_pool_free(ctx,ev_more); gk_pool_free(ctx,mask_more);
  gk_pool_free(ctx,d_spill_pair); gk_pool_free(ctx,wide_ev);
  gk_pool_free(ctx,nt.keys); gk_pool_free(ctx,nt.seq); gk_pool_free(ctx,nt.rank);
  if (err & 2) {
    gk_set_error("a filter-passing mate carries more variant events than its record format allows (%d, wide %d)", kMaxEv,
                 GK_WIDE_EVENTS);
    gk_tab_destroy(tab);
    return GK_ERR_CAPACITY;
  }
  *out = tab;
  return GK_OK;
}

int gk_tab_from_csr(gk_ctx* ctx, int32_t n_var_total, int64_t n_valid, const uint32_t* off, const uint32_t* ids,
                    const uint8_t* pair_gene, const uint8_t* pair_nh, gk_tab** out) {
  gk_bind(ctx);
  GK_REQUIRE(ctx && out && off && pair_gene && pair_nh && n_valid >= 0 && n_var_total >= 0, "bad CSR arguments");
  const int64_t n_ids = off[4 * n_valid];
  for (int64_t i = 0; i < 4 * n_valid; ++i) GK_REQUIRE(off[i] <= off[i + 1], "CSR offsets must be non-decreasing");
  for (int64_t i = 0; i < n_ids; ++i) GK_REQUIRE(ids[i] < (uint32_t)n_var_total, "variant ordinal out of range");
  gk_tab* tab = new gk_tab();
  tab->ctx = ctx; tab->n_pairs = n_valid; tab->n_valid = n_valid; tab->n_ids = n_ids;
  tab->n_var = n_var_total; tab->n_novel = 0;
  hipStream_t st = ctx->stream;
  GK_HIP(gk_pool_malloc(ctx, (void**)&tab->d_off, (size_t)(4 * n_valid + 1) * sizeof(uint32_t)));
  GK_HIP(gk_pool_malloc(ctx, (void**)&tab->d_ids, (size_t)(n_ids + 1) * sizeof(uint32_t)));
  GK_HIP(gk_pool_malloc(ctx, (void**)&tab->d_pair_gene, (size_t)n_valid + 1));
  GK_HIP(gk_pool_malloc(ctx, (void**)&tab->d_pair_nh, (size_t)n_valid + 1));
  GK_HIP(hipMemcpyAsync(tab->d_off, off, (size_t)(4 * n_valid + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, st));
  if (n_ids) GK_HIP(hipMemcpyAsync(tab->d_ids, ids, (size_t)n_ids * sizeof(uint32_t), hipMemcpyHostToDevice, st));
  if (n_valid) {
    GK_HIP(hipMemcpyAsync(tab->d_pair_gene, pair_gene, (size_t)n_valid, hipMemcpyHostToDevice, st));
    GK_HIP(hipMemcpyAsync(tab->d_pair_nh, pair_nh, (size_t)n_valid, hipMemcpyHostToDevice, st));
  }
  GK_HIP(hipStreamSynchronize(st));
  *out = tab;
  return GK_OK;
}

int gk_tab_get_info(gk_tab* tab, gk_tab_info* info) {
  gk_bind(tab ? tab->ctx : nullptr);
  GK_REQUIRE(tab && info, "null pointer");
  info->n_pairs = tab->n_pairs; info->n_valid = tab->n_valid; info->n_ids = tab->n_ids;
  info->n_novel = tab->n_novel; info->err_flags = tab->err_flags;
  info->d_pair_src = gk_addr(tab->d_pair_src); info->d_off = gk_addr(tab->d_off); info->d_ids = gk_addr(tab->d_ids);
  info->d_pair_gene = gk_addr(tab->d_pair_gene); info->d_pair_nh = gk_addr(tab->d_pair_nh);
  info->d_novel_key = gk_addr(tab->d_novel_key);
  return GK_OK;
}

int gk_tab_destroy(gk_tab* tab) {
  gk_bind(tab ? tab->ctx : nullptr);
  if (!tab) return GK_OK;
  gk_ctx* ctx = tab->ctx;
  gk_pool_free(ctx,tab->d_pair_src); gk_pool_free(ctx,tab->d_off); gk_pool_free(ctx,tab->d_ids);
  gk_pool_free(ctx,tab->d_pair_gene); gk_pool_free(ctx,tab->d_pair_nh); gk_pool_free(ctx,tab->d_novel_key);
  gk_pool_free(ctx,tab->d_wide);
  for (auto& part : tab->part)
    if (part.d_rows) gk_pool_free(part.owner, part.d_rows);
  delete tab;
  return GK_OK;
}

}  // extern "C"

extern "C" {

/* The packed records of a sample in a compact form for the time between its depth and its typing (--cn-cohort types a
 * sample only after the pooled fit of the whole cohort, main.py:572-589; hisat2.py:228-276 is what the records hold):
 * uint32 offsets [n_mates + 1] followed by the words the mates use.  *d_compact_out is a block of the context's pool
 * (gk_free), *bytes_out its size.  gk_mates_expand writes the 128-byte records back (unused parts zero). */
int gk_mates_compact(gk_ctx* ctx, gk_dptr d_mates, int64_t n_mates, gk_dptr* d_compact_out, int64_t* bytes_out) {
  gk_bind(ctx);
  GK_REQUIRE(ctx && d_compact_out && bytes_out && n_mates >= 0 && (n_mates == 0 || d_mates), "bad compaction arguments");
  GK_REQUIRE(n_mates < (1ll << 27), "more than 2^26 pairs per call");
  hipStream_t st = ctx->stream;
  const gk_mate* mates = gk_ptr<const gk_mate>(d_mates);
  uint32_t* cnt = nullptr;
  GK_HIP(gk_pool_malloc(ctx, (void**)&cnt, (size_t)(n_mates + 2) * sizeof(uint32_t)));
  if (n_mates) GK_KERNEL(mates_count_words, dim3(nblk(n_mates)), dim3(kThreads), 0, st, mates, n_mates, cnt);
  int rc = gk_scan_u32(ctx, cnt, n_mates, cnt + n_mates);
  if (rc) { gk_pool_free(ctx, cnt); return rc; }
  uint32_t total = 0;
  GK_HIP(gk_fetch(ctx, &total, cnt + n_mates, sizeof(uint32_t)));
  const size_t bytes = ((size_t)(n_mates + 1) + (size_t)total) * sizeof(uint32_t);
  uint32_t* out = nullptr;
  GK_HIP(gk_pool_malloc(ctx, (void**)&out, bytes));
  GK_KERNEL(mates_pack_words, dim3(nblk(n_mates + 1)), dim3(kThreads), 0, st, mates, n_mates, cnt, out, out + n_mates + 1, total);
  GK_HIP(hipGetLastError());
  gk_pool_free(ctx, cnt);      // stream-ordered reuse
  *d_compact_out = gk_addr(out);
  *bytes_out = (int64_t)bytes;
  return GK_OK;
}

int gk_mates_expand(gk_ctx* ctx, gk_dptr d_compact, int64_t n_mates, gk_dptr d_mates_out) {
  gk_bind(ctx);
  GK_REQUIRE(ctx && n_mates >= 0 && (n_mates == 0 || (d_compact && d_mates_out)), "bad expansion arguments");
  if (!n_mates) return GK_OK;
  const uint32_t* off = gk_ptr<const uint32_t>(d_compact);
  GK_KERNEL(mates_unpack_words, dim3(nblk(n_mates)), dim3(kThreads), 0, ctx->stream, off, off + n_mates + 1, n_mates,
            gk_ptr<gk_mate>(d_mates_out));
  GK_HIP(hipGetLastError());
  return GK_OK;
}

}  // extern "C"

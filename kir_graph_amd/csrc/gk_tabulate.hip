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
// Data layout (HBM): mates are 64-byte records, two per pair, read once per pass with 16-byte
// vector loads; the sorted variant key table (<= ~0.5 MB) stays L2/MALL resident; outputs are one
// CSR (uint32 offsets, uint32 ordinals) in the factor order lpv, rpv, lnv, rnv.
//
// Two passes (count, emit) around one exclusive scan; novel variants are deduplicated in a device
// hash table keyed by the packed variant key, ranked by first appearance through a bitmap over
// (mate, event) sequence numbers, so that the numbering equals the reference's sequential counter.
#include "gk_common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kMaxEv = GK_MAX_EVENTS;
constexpr uint64_t kEmpty = ~0ull;

struct Events {
  int n;
  uint64_t key[kMaxEv];   // position / type / value of every non-match event (private memory)
  uint32_t last_len;      // walker length of the last event: 1 single, k insertion / deletion
  bool clipped;
  bool overflow;
  bool last_is_event;     // last walk element is events[n-1]; otherwise a match ending at ref_end
  uint32_t ref_end;
};

__device__ inline uint4 load16(const void* p) { return *reinterpret_cast<const uint4*>(p); }

constexpr int kMateWords = sizeof(gk_mate) / 4;   // 32
constexpr int kCigWord = 3;                         // uint16 cig[] starts at byte 12
constexpr int kMmWord = kCigWord + GK_MAX_CIG / 2;  // 10
constexpr int kInsWord = kMmWord + GK_MAX_MM;       // 26
static_assert(sizeof(gk_mate) == 128 && kInsWord + GK_MAX_INS == kMateWords, "gk_mate layout");

struct MateRegs {
  uint32_t w[kMateWords];
  __device__ uint32_t pos0() const { return w[0]; }
  __device__ uint32_t flag() const { return w[1] & 0xFFFFu; }
  __device__ uint32_t ref() const { return (w[1] >> 16) & 0xFFu; }
  __device__ uint32_t nh() const { return w[1] >> 24; }
  __device__ uint32_t nm() const { return w[2] & 0xFFu; }
  __device__ uint32_t n_cig() const { return (w[2] >> 8) & 0xFFu; }
  __device__ uint32_t n_mm() const { return (w[2] >> 16) & 0xFFu; }
  __device__ uint32_t n_ins() const { return w[2] >> 24; }
  __device__ uint32_t cig(int i) const {  // uint16 array starting at byte 12
    uint32_t word = w[kCigWord + (i >> 1)];
    return (i & 1) ? (word >> 16) : (word & 0xFFFFu);
  }
  __device__ uint32_t mm_off(int i) const { return w[kMmWord + i] & 0xFFFFu; }
  __device__ uint32_t mm_base(int i) const { return (w[kMmWord + i] >> 16) & 0xFFu; }
  __device__ uint32_t ins(int i) const { return w[kInsWord + i]; }
};

__device__ inline void load_mate(const gk_mate* mates, int64_t m, MateRegs& r) {
  const uint4* p = reinterpret_cast<const uint4*>(mates + m);
#pragma unroll
  for (int k = 0; k < kMateWords / 4; ++k) {
    uint4 v = p[k];
    r.w[4 * k + 0] = v.x; r.w[4 * k + 1] = v.y; r.w[4 * k + 2] = v.z; r.w[4 * k + 3] = v.w;
  }
}

__device__ inline bool mate_passes(const MateRegs& r) {
  return (r.flag() & 2u) && r.nm() != GK_NM_ABSENT && r.nm() <= 4u;
}

__device__ inline void push_event(Events& ev, uint32_t pos, uint32_t len, uint64_t key) {
  (void)pos;
  if (ev.n < kMaxEv) {
    ev.key[ev.n] = key;
    ev.last_len = len;
    ev.n++;
  } else {
    ev.overflow = true;
  }
  ev.last_is_event = true;
}

// CIGAR x mismatch co-walk (recordToRawVariant): only non-match events are stored; the match
// segments matter solely through the left edge (pos0) and the right edge (ref_end / last event).
__device__ inline void walk(const MateRegs& r, Events& ev) {
  ev.n = 0; ev.clipped = false; ev.overflow = false; ev.last_is_event = false;
  uint32_t cur = r.pos0();
  const uint32_t ref = r.ref();
  int mi = 0, ii = 0;
  const int n_mm = min((int)r.n_mm(), GK_MAX_MM), n_cig = min((int)r.n_cig(), GK_MAX_CIG);
  for (int c = 0; c < n_cig; ++c) {
    const uint32_t cg = r.cig(c);
    const uint32_t op = cg & 15u, len = cg >> 4;
    if (op == GK_CIG_S) {
      ev.clipped = true;
    } else if (op == GK_CIG_M) {
      const uint32_t end = cur + len;
      uint32_t seg = cur;
      while (mi < n_mm && r.pos0() + r.mm_off(mi) < end) {
        const uint32_t p = r.pos0() + r.mm_off(mi);
        push_event(ev, p, 1u, gk_make_key(ref, p, GK_TYP_SINGLE, r.mm_base(mi)));
        seg = p + 1;
        ++mi;
      }
      if (seg < end) ev.last_is_event = false;  // trailing match segment
      cur = end;
    } else if (op == GK_CIG_I) {
      const uint32_t sid = ii < GK_MAX_INS ? r.ins(ii) : 0u;
      push_event(ev, cur, len, gk_make_key(ref, cur, GK_TYP_INS, sid));
      ++ii;
    } else if (op == GK_CIG_D) {
      push_event(ev, cur, len, gk_make_key(ref, cur, GK_TYP_DEL, len));
      cur += len;
    }
  }
  ev.ref_end = cur;
}

__device__ inline int lower_bound_key(const uint64_t* key, int n, uint64_t k) {
  int lo = 0, hi = n;
  while (lo < hi) {
    int mid = (lo + hi) >> 1;
    if (key[mid] < k) lo = mid + 1; else hi = mid;
  }
  return lo;
}

__device__ inline uint32_t hash64(uint64_t k) {
  k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33;
  return (uint32_t)k;
}

struct NovelTable {
  uint64_t* keys;   // kEmpty when free
  uint32_t* seq;    // min (mate*4 + event) over insertions
  uint32_t* rank;   // first-appearance rank (filled by rank kernel)
  uint32_t mask;
};

__device__ inline void novel_insert(const NovelTable& t, uint64_t key, uint32_t seq) {
  uint32_t s = hash64(key) & t.mask;
  for (uint32_t probe = 0; probe <= t.mask; ++probe) {
    unsigned long long prev = atomicCAS((unsigned long long*)&t.keys[s], (unsigned long long)kEmpty,
                                        (unsigned long long)key);
    if (prev == kEmpty || prev == key) {
      atomicMin(&t.seq[s], seq);
      return;
    }
    s = (s + 1) & t.mask;
  }
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

struct Resolved {
  int32_t ord[kMaxEv];   // >= 0 index ordinal, -1 novel
  bool drop;             // novel insertion / deletion present -> mate yields ([], [])
  uint32_t right;
  int lo, hi;
  bool bad_window;
};

__device__ inline void resolve(const Events& ev, const MateRegs& r, const uint64_t* key, int n_var, Resolved& rs) {
  rs.drop = false;
  for (int e = 0; e < ev.n; ++e) {
    int i = lower_bound_key(key, n_var, ev.key[e]);
    if (i < n_var && key[i] == ev.key[e]) {
      rs.ord[e] = i;
    } else {
      rs.ord[e] = -1;
      if (gk_key_typ(ev.key[e]) != GK_TYP_SINGLE) rs.drop = true;
    }
  }
  if (ev.last_is_event && ev.n > 0) {
    const int e = ev.n - 1;
    rs.right = gk_key_pos(ev.key[e]) + (rs.ord[e] >= 0 ? 0u : ev.last_len);  // index records carry length 0
  } else {
    rs.right = ev.ref_end;
  }
  const uint32_t ref = r.ref();
  rs.lo = lower_bound_key(key, n_var, gk_make_key(ref, r.pos0(), GK_TYP_SINGLE, 'A'));
  rs.hi = lower_bound_key(key, n_var, gk_make_key(ref, rs.right, GK_TYP_SINGLE, 'T'));
  rs.bad_window = rs.lo > rs.hi;
}

__device__ inline bool negative_kept(uint64_t k, int i, const Events& ev, const Resolved& rs) {
  const uint32_t typ = gk_key_typ(k), pos = gk_key_pos(k), val = gk_key_val(k);
  for (int e = 0; e < ev.n; ++e) {
    if (rs.ord[e] == i) return false;
    if (gk_key_typ(ev.key[e]) == GK_TYP_SINGLE && gk_key_val(ev.key[e]) == 'N' && typ == GK_TYP_SINGLE &&
        pos == gk_key_pos(ev.key[e]) && (val == 'A' || val == 'C' || val == 'G' || val == 'T'))
      return false;
  }
  if (typ == GK_TYP_DEL && pos + val + 10u >= rs.right) return false;
  return true;
}

// pass 1: validity, counts, novel registration.  One thread per mate; mates of a pair sit in
// adjacent lanes so the pair verdict is one lane shuffle.
__global__ __launch_bounds__(kThreads) void tab_count(const gk_mate* mates, int64_t n_mates, const uint64_t* key,
                                                      int n_var, NovelTable nt, uint32_t* cnt /*[4*n_pairs+1]*/,
                                                      uint32_t* valid /*[n_pairs]*/, int* err_flags) {
  const int64_t m = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  const bool in = m < n_mates;
  MateRegs r;
  if (in) load_mate(mates, m, r); else { for (int k = 0; k < kMateWords; ++k) r.w[k] = 0; }
  const bool ok = in && mate_passes(r);
  const bool ok_other = __shfl_xor((int)ok, 1, 64) != 0;
  const bool pair_ok = ok && ok_other;
  if (!in) return;
  const int64_t pair = m >> 1;
  const int side = (int)(m & 1);
  uint32_t n_pos = 0, n_neg = 0;
  if (pair_ok) {
    Events ev;
    walk(r, ev);
    if (ev.overflow) atomicOr(err_flags, 2);
    if (!ev.clipped) {
      Resolved rs;
      resolve(ev, r, key, n_var, rs);
      for (int e = 0; e < ev.n; ++e)
        if (rs.ord[e] < 0) novel_insert(nt, ev.key[e], (uint32_t)(m * kMaxEv + e));
      if (rs.bad_window) {
        atomicOr(err_flags, 1);
      } else if (!rs.drop) {
        n_pos = ev.n;
        for (int i = rs.lo; i < rs.hi; ++i) n_neg += negative_kept(key[i], i, ev, rs) ? 1u : 0u;
      }
    }
  }
  cnt[4 * pair + side] = n_pos;
  cnt[4 * pair + 2 + side] = n_neg;
  if (side == 0) valid[pair] = pair_ok ? 1u : 0u;
}

// novel ranking: mark first-appearance sequence numbers, prefix-popcount, assign ranks
__global__ void novel_mark(NovelTable nt, uint32_t* bitmap) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s > nt.mask) return;
  if (nt.keys[s] != kEmpty) atomicOr(&bitmap[nt.seq[s] >> 5], 1u << (nt.seq[s] & 31));
}
__global__ void bitmap_popc(const uint32_t* bitmap, uint32_t* cnt, int64_t n_words) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_words) cnt[i] = __popc(bitmap[i]);
}
__global__ void novel_assign(NovelTable nt, const uint32_t* bitmap, const uint32_t* prefix, uint64_t* novel_key) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s > nt.mask) return;
  const uint64_t k = nt.keys[s];
  if (k == kEmpty) return;
  const uint32_t q = nt.seq[s];
  const uint32_t rank = prefix[q >> 5] + __popc(bitmap[q >> 5] & ((1u << (q & 31)) - 1u));
  nt.rank[s] = rank;
  novel_key[rank] = k;
}

// pass 2: emit ordinals at the scanned offsets
__global__ __launch_bounds__(kThreads) void tab_emit(const gk_mate* mates, int64_t n_mates, const uint64_t* key,
                                                     int n_var, NovelTable nt, const uint32_t* off,
                                                     const uint32_t* valid, uint32_t* ids) {
  const int64_t m = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (m >= n_mates) return;
  const int64_t pair = m >> 1;
  const int side = (int)(m & 1);
  if (!valid[pair]) return;
  const uint32_t o_pos = off[4 * pair + side], o_pos_end = off[4 * pair + side + 1];
  const uint32_t o_neg = off[4 * pair + 2 + side], o_neg_end = off[4 * pair + 2 + side + 1];
  if (o_pos == o_pos_end && o_neg == o_neg_end) return;
  MateRegs r;
  load_mate(mates, m, r);
  Events ev;
  walk(r, ev);
  Resolved rs;
  resolve(ev, r, key, n_var, rs);
  for (int e = 0; e < ev.n && o_pos + e < o_pos_end; ++e)
    ids[o_pos + e] = rs.ord[e] >= 0 ? (uint32_t)rs.ord[e] : (uint32_t)n_var + novel_rank(nt, ev.key[e]);
  uint32_t w = o_neg;
  for (int i = rs.lo; i < rs.hi && w < o_neg_end; ++i)
    if (negative_kept(key[i], i, ev, rs)) ids[w++] = (uint32_t)i;
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

inline unsigned nblk(int64_t n, int t = kThreads) { return (unsigned)((n + t - 1) / t); }

}  // namespace

extern "C" {

int gk_index_create(gk_ctx* ctx, const uint64_t* key, int32_t n_var, const int32_t* gene_vbeg, int32_t n_gene,
                    gk_index** out) {
  GK_REQUIRE(ctx && out && gene_vbeg && n_gene > 0 && n_gene < 255 && n_var >= 0, "bad index arguments");
  for (int i = 1; i < n_var; ++i) GK_REQUIRE(key[i - 1] < key[i], "index keys must be strictly increasing");
  gk_index* idx = new gk_index();
  idx->ctx = ctx; idx->n_var = n_var; idx->n_gene = n_gene;
  idx->gene_vbeg.assign(gene_vbeg, gene_vbeg + n_gene + 1);
  GK_HIP(gk_pool_malloc(ctx, (void**)&idx->d_key, (size_t)(n_var + 1) * sizeof(uint64_t)));
  GK_HIP(gk_pool_malloc(ctx, (void**)&idx->d_gene_vbeg, (size_t)(n_gene + 1) * sizeof(int32_t)));
  GK_HIP(hipMemcpyAsync(idx->d_key, key, (size_t)n_var * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
  GK_HIP(hipMemcpyAsync(idx->d_gene_vbeg, gene_vbeg, (size_t)(n_gene + 1) * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
  GK_HIP(hipStreamSynchronize(ctx->stream));
  *out = idx;
  return GK_OK;
}

int gk_index_destroy(gk_index* idx) {
  if (!idx) return GK_OK;
  gk_ctx* ctx = idx->ctx;
  gk_pool_free(ctx,idx->d_key);
  gk_pool_free(ctx,idx->d_gene_vbeg);
  delete idx;
  return GK_OK;
}

int gk_tabulate(gk_ctx* ctx, gk_index* idx, gk_dptr d_mates_p, int64_t n_pairs, gk_tab** out) {
  GK_REQUIRE(ctx && idx && out && n_pairs >= 0, "bad tabulate arguments");
  GK_REQUIRE(n_pairs < (1ll << 26), "more than 2^26 pairs per call");
  const gk_mate* mates = gk_ptr<const gk_mate>(d_mates_p);
  const int64_t n_mates = 2 * n_pairs;
  hipStream_t st = ctx->stream;
  gk_tab* tab = new gk_tab();
  tab->ctx = ctx; tab->idx = idx; tab->n_pairs = n_pairs; tab->n_var = idx->n_var;

  // novel hash table: at most kMaxEv novel events per mate, load factor <= 0.5, >= 2^16 slots
  uint32_t log2cap = 16;
  while ((1ull << log2cap) < (uint64_t)n_mates * kMaxEv * 2 && log2cap < 30) ++log2cap;
  // most events are known variants or repeats: start small and rely on the bound only up to 2^24
  if (log2cap > 24) log2cap = 24;
  NovelTable nt;
  const size_t cap = 1ull << log2cap;
  nt.mask = (uint32_t)(cap - 1);
  GK_HIP(gk_pool_malloc(ctx, (void**)&nt.keys, cap * sizeof(uint64_t)));
  GK_HIP(gk_pool_malloc(ctx, (void**)&nt.seq, cap * sizeof(uint32_t)));
  GK_HIP(gk_pool_malloc(ctx, (void**)&nt.rank, cap * sizeof(uint32_t)));
  GK_HIP(hipMemsetAsync(nt.keys, 0xFF, cap * sizeof(uint64_t), st));
  GK_HIP(hipMemsetAsync(nt.seq, 0xFF, cap * sizeof(uint32_t), st));

  uint32_t *cnt = nullptr, *valid = nullptr;
  int* d_err = nullptr;
  GK_HIP(gk_pool_malloc(ctx, (void**)&cnt, (size_t)(4 * n_pairs + 2) * sizeof(uint32_t)));
  GK_HIP(gk_pool_malloc(ctx, (void**)&valid, (size_t)(n_pairs + 1) * sizeof(uint32_t)));
  GK_HIP(gk_pool_malloc(ctx, (void**)&d_err, sizeof(int)));
  GK_HIP(hipMemsetAsync(d_err, 0, sizeof(int), st));

  if (n_mates) {
    GK_PROF(ctx, GK_K_TAB_COUNT, hipLaunchKernelGGL(tab_count, dim3(nblk(n_mates)), dim3(kThreads), 0, st, mates, n_mates, idx->d_key,
                       idx->n_var, nt, cnt, valid, d_err));
  }
  // offsets over input pairs (invalid pairs contribute zeros)
  int rc = gk_scan_u32(ctx, cnt, 4 * n_pairs, cnt + 4 * n_pairs);
  if (rc) return rc;

  // novel ranks
  const int64_t n_seq = n_mates * kMaxEv;
  const int64_t n_words = (n_seq + 31) / 32 + 1;
  uint32_t *bitmap = nullptr, *prefix = nullptr;
  GK_HIP(gk_pool_malloc(ctx, (void**)&bitmap, (size_t)n_words * sizeof(uint32_t)));
  GK_HIP(gk_pool_malloc(ctx, (void**)&prefix, (size_t)(n_words + 1) * sizeof(uint32_t)));
  GK_HIP(hipMemsetAsync(bitmap, 0, (size_t)n_words * sizeof(uint32_t), st));
  GK_PROF(ctx, GK_K_NOVEL, hipLaunchKernelGGL(novel_mark, dim3(nblk((int64_t)cap)), dim3(kThreads), 0, st, nt, bitmap));
  GK_PROF(ctx, GK_K_NOVEL, hipLaunchKernelGGL(bitmap_popc, dim3(nblk(n_words)), dim3(kThreads), 0, st, bitmap, prefix, n_words));
  rc = gk_scan_u32(ctx, prefix, n_words, prefix + n_words);
  if (rc) return rc;

  uint32_t totals[2] = {0, 0};
  int err = 0;
  GK_HIP(hipMemcpyAsync(&totals[0], cnt + 4 * n_pairs, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  GK_HIP(hipMemcpyAsync(&totals[1], prefix + n_words, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  GK_HIP(hipMemcpyAsync(&err, d_err, sizeof(int), hipMemcpyDeviceToHost, st));
  GK_HIP(hipStreamSynchronize(st));
  tab->n_ids = totals[0];
  tab->n_novel = (int32_t)totals[1];
  tab->err_flags = err;
  if ((uint64_t)tab->n_novel * 2 > cap) {
    gk_set_error("novel variant table overflow (%d novel variants)", tab->n_novel);
    return GK_ERR_CAPACITY;
  }

  GK_HIP(gk_pool_malloc(ctx, (void**)&tab->d_novel_key, (size_t)(tab->n_novel + 1) * sizeof(uint64_t)));
  GK_PROF(ctx, GK_K_NOVEL, hipLaunchKernelGGL(novel_assign, dim3(nblk((int64_t)cap)), dim3(kThreads), 0, st, nt, bitmap, prefix,
                     tab->d_novel_key));

  GK_HIP(gk_pool_malloc(ctx, (void**)&tab->d_ids, (size_t)(tab->n_ids + 1) * sizeof(uint32_t)));
  if (n_mates) {
    GK_PROF(ctx, GK_K_TAB_EMIT, hipLaunchKernelGGL(tab_emit, dim3(nblk(n_mates)), dim3(kThreads), 0, st, mates, n_mates, idx->d_key, idx->n_var,
                       nt, cnt, valid, tab->d_ids));
  }
  // compact valid pairs (order preserving)
  GK_HIP(gk_pool_malloc(ctx, (void**)&tab->d_pair_src, (size_t)(n_pairs + 1) * sizeof(int32_t)));
  rc = gk_compact(ctx, valid, nullptr, n_pairs, tab->d_pair_src, &tab->n_valid);
  if (rc) return rc;
  GK_HIP(gk_pool_malloc(ctx, (void**)&tab->d_off, (size_t)(4 * tab->n_valid + 1) * sizeof(uint32_t)));
  GK_HIP(gk_pool_malloc(ctx, (void**)&tab->d_pair_gene, (size_t)tab->n_valid + 1));
  GK_HIP(gk_pool_malloc(ctx, (void**)&tab->d_pair_nh, (size_t)tab->n_valid + 1));
  GK_PROF(ctx, GK_K_SELECT, hipLaunchKernelGGL(gather_pairs, dim3(nblk(tab->n_valid + 1)), dim3(kThreads), 0, st, mates, tab->d_pair_src,
                     tab->n_valid, cnt, tab->d_off, tab->d_pair_gene, tab->d_pair_nh, (uint32_t)tab->n_ids));
  GK_HIP(hipGetLastError());
  GK_HIP(hipStreamSynchronize(st));
  gk_pool_free(ctx,cnt); gk_pool_free(ctx,valid); gk_pool_free(ctx,d_err); gk_pool_free(ctx,bitmap); gk_pool_free(ctx,prefix);
  gk_pool_free(ctx,nt.keys); gk_pool_free(ctx,nt.seq); gk_pool_free(ctx,nt.rank);
  if (err & 2) {
    gk_set_error("a filter-passing mate carries more than %d variant events", kMaxEv);
    gk_tab_destroy(tab);
    return GK_ERR_CAPACITY;
  }
  *out = tab;
  return GK_OK;
}

int gk_tab_from_csr(gk_ctx* ctx, int32_t n_var_total, int64_t n_valid, const uint32_t* off, const uint32_t* ids,
                    const uint8_t* pair_gene, const uint8_t* pair_nh, gk_tab** out) {
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
  GK_REQUIRE(tab && info, "null pointer");
  info->n_pairs = tab->n_pairs; info->n_valid = tab->n_valid; info->n_ids = tab->n_ids;
  info->n_novel = tab->n_novel; info->err_flags = tab->err_flags;
  info->d_pair_src = gk_addr(tab->d_pair_src); info->d_off = gk_addr(tab->d_off); info->d_ids = gk_addr(tab->d_ids);
  info->d_pair_gene = gk_addr(tab->d_pair_gene); info->d_pair_nh = gk_addr(tab->d_pair_nh);
  info->d_novel_key = gk_addr(tab->d_novel_key);
  return GK_OK;
}

int gk_tab_destroy(gk_tab* tab) {
  if (!tab) return GK_OK;
  gk_ctx* ctx = tab->ctx;
  gk_pool_free(ctx,tab->d_pair_src); gk_pool_free(ctx,tab->d_off); gk_pool_free(ctx,tab->d_ids);
  gk_pool_free(ctx,tab->d_pair_gene); gk_pool_free(ctx,tab->d_pair_nh); gk_pool_free(ctx,tab->d_novel_key);
  delete tab;
  return GK_OK;
}

}  // extern "C"

// Internal helpers shared by the HIP translation units of libgraphkir_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "graphkir_hip.h"
#include "gk_env.h"

void gk_set_error(const char* fmt, ...);

#define GK_HIP(call)                                                                   \
  do {                                                                                 \
    hipError_t e_ = (call);                                                            \
    if (e_ != hipSuccess) {                                                            \
      gk_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
      return GK_ERR_HIP;                                                               \
    }                                                                                  \
  } while (0)

#define GK_REQUIRE(cond, msg)                        \
  do {                                               \
    if (!(cond)) {                                   \
      gk_set_error("%s (%s:%d)", msg, __FILE__, __LINE__); \
      return GK_ERR_ARG;                             \
    }                                                \
  } while (0)

struct gk_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  // small reusable device scratch (scan partials, counters)
  void* scratch = nullptr;
  size_t scratch_bytes = 0;
  // pinned host staging: a ring for host -> device parameters (gk_send) and one for device -> host results
  // (gk_fetch_*).  Copies to / from pageable memory go through a staging path inside the runtime that the host threads
  // of a process share; with a dozen gene threads that path was the throughput limit of one process.  Both rings are
  // addressed by byte counters that only grow (offset = counter % size); space comes back when a MARK -- an event
  // recorded on the stream, gk_fetch_mark -- is known to have passed, or when the stream has been drained.
  struct Ring { void* base = nullptr; size_t bytes = 0; uint64_t head = 0, tail = 0; };
  Ring send_ring, fetch_ring;
  struct Pending { void* dst; size_t off, bytes; uint64_t end; };
  std::deque<Pending> fetches;
  struct Mark { hipEvent_t ev; uint64_t id, send_head, fetch_head; };
  std::deque<Mark> marks;
  uint64_t mark_next = 1, mark_done = 0;      // ids handed out / the newest mark known to have passed
  std::vector<hipEvent_t> mark_pool;
  // reduction-tree programs of gk_search.hip, one device block per row count (they depend on nothing else)
  std::map<int64_t, void*> tree_programs;
  struct TreeHead {
    size_t o_leaf, o_span, o_cs, o_co, o_top, o_flat, o_cl, o_clo, o_lops, o_unit, o_zero;
    int n_spans, n_chunks, n_leaves, max_chunk_leaves;
  };
  std::map<int64_t, TreeHead> tree_heads;
  // tickets of kernels whose last workgroup continues the work (gk_ctx_tickets): zero between launches -- the
  // workgroup that takes the last ticket of a group puts the counter back
  uint32_t* tickets = nullptr;
  size_t n_tickets = 0;
  // caching allocator state (gk_pool_*)
  std::mutex pool_mutex;   // frees may come from another host thread (Python GC)
  std::multimap<size_t, void*> pool_free;
  std::unordered_map<void*, size_t> pool_live;
  size_t pool_cached_bytes = 0;
  // optional per-kernel timing with HIP events on `stream` (bench.py roofline leg)
  bool prof_on = false;
  struct ProfSpan { int id; hipEvent_t a, b; };
  std::vector<ProfSpan> prof_spans;
  std::vector<hipEvent_t> prof_pool;
};

// Per-kernel timing is kept under the kernel's own name (template arguments dropped): GK_PROF(ctx, "name", launch).  A
// name gets its id on first use (gk_prof_register, thread-safe); at most GK_PROF_MAX names.
constexpr int GK_PROF_MAX = 128;
int gk_prof_register(const char* name);
// Per-kernel timing.  GK_PROF brackets a launch with two events recorded on the stream (cheap; under
// multi-stream load the span also counts the time the kernel waits for CUs that other streams are using).
// GK_PROF_EXACT hands the events to hipExtLaunchKernelGGL, which binds them to the kernel's own begin
// and end on the GPU -- what rocprofv3 reports; such dispatches carry a profiling signal and cost some
// cross-stream overlap, so only the kernel the roofline is reported for is timed this way.
void gk_prof_begin(gk_ctx* ctx, int id, int exact);
void gk_prof_end(gk_ctx* ctx);
hipEvent_t gk_prof_start_event();   // events of the exact span opened on this thread, else nullptr
hipEvent_t gk_prof_stop_event();
#define GK_PROF(ctx, name, launch)                          \
  do {                                                      \
    static const int gk_prof_id_ = gk_prof_register(name);  \
    gk_prof_begin((ctx), gk_prof_id_, 0);                   \
    launch;                                                 \
    gk_prof_end((ctx));                                     \
  } while (0)
#define GK_PROF_EXACT(ctx, name, launch)                    \
  do {                                                      \
    static const int gk_prof_id_ = gk_prof_register(name);  \
    gk_prof_begin((ctx), gk_prof_id_, 1);                   \
    launch;                                                 \
    gk_prof_end((ctx));                                     \
  } while (0)
#define GK_KERNEL(kernel, grid, block, lds, stream, ...) \
  hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, gk_prof_start_event(), gk_prof_stop_event(), 0, __VA_ARGS__)

int gk_ctx_scratch(gk_ctx* ctx, size_t bytes, void** out);
// `n` counters that are zero whenever no kernel of this context's stream is using them (see gk_ctx::tickets)
int gk_ctx_tickets(gk_ctx* ctx, size_t n, uint32_t** out);

// host -> device through the context's pinned ring: queued on the stream, `src` may be reused at once -- unless the
// transfer is larger than gk_stage_direct() bytes: that one is queued straight from `src`, which must then stay valid
// until the stream has passed it
hipError_t gk_send(gk_ctx* ctx, void* dst_dev, const void* src, size_t bytes);
size_t gk_stage_direct();
// device -> host through the pinned ring: queue any number of copies, then wait once (stream synchronise) and have
// them delivered to their destinations; gk_fetch = queue + wait.  `dst` must stay valid until the delivery.
hipError_t gk_fetch_queue(gk_ctx* ctx, void* dst, const void* src_dev, size_t bytes);
hipError_t gk_fetch_wait(gk_ctx* ctx);
// a slot of the pinned result ring that a kernel writes itself (no copy): see gk_runtime.hip
hipError_t gk_fetch_direct(gk_ctx* ctx, void* dst, size_t bytes, void** dev_out);
// A mark is a point of the stream: gk_fetch_wait_mark returns when the stream has passed it, with every copy queued
// before it delivered -- work queued after the mark keeps running, so a caller that drives several independent
// sequences on one stream (the genes of a sample, gk_sample_search) waits for one without draining the others.
hipError_t gk_fetch_mark(gk_ctx* ctx, uint64_t* mark);
hipError_t gk_fetch_wait_mark(gk_ctx* ctx, uint64_t mark);
// Drain the stream and DROP the copies still queued (their destinations are going away: an error path).
void gk_fetch_cancel(gk_ctx* ctx);
static inline hipError_t gk_fetch(gk_ctx* ctx, void* dst, const void* src_dev, size_t bytes) {
  hipError_t e = gk_fetch_queue(ctx, dst, src_dev, bytes);
  return e != hipSuccess ? e : gk_fetch_wait(ctx);
}

// Make the context's GPU the current HIP device of the calling host thread.  Every entry point of
// the C ABI starts with it: contexts are driven from pool threads (gene workers, sample prefetch)
// that never chose a device themselves, and on a multi-GPU node the default device is not theirs.
static inline void gk_bind(gk_ctx* ctx) {
  if (ctx) (void)hipSetDevice(ctx->device);   // per-thread state in the runtime: cheap, and never stale
}

// Stream-ordered caching allocator: freed blocks are kept per size class and handed out again
// without hipMalloc / hipFree (both synchronise the device).  Safe because every kernel and copy
// of a context runs on its single stream.
hipError_t gk_pool_malloc(gk_ctx* ctx, void** out, size_t bytes);
void gk_pool_free(gk_ctx* ctx, void* p);

struct gk_index {
  gk_ctx* ctx = nullptr;
  uint64_t* d_key = nullptr;
  int32_t* d_gene_vbeg = nullptr;
  int32_t* d_bucket = nullptr;      // 16-bp position buckets into d_key, all genes back to back
  int32_t* d_gene_boff = nullptr;   // [n_gene + 1] first bucket of each gene
  uint32_t* d_del_bits = nullptr;   // bit v = index variant v is a deletion ((n_var + 31) / 32 + 2 words, zero padded)
  // per backbone position p (gene g: entries d_gene_pbase[g] .. d_gene_pbase[g + 1]): first ordinal whose key is
  // >= (g, p, single, 'A') / (g, p, single, 'T') -- the two bounds of a variant window and the entry point of a
  // substitution look-up, one read instead of bucket + bisection
  int32_t* d_lb_a = nullptr;
  int32_t* d_lb_t = nullptr;
  int32_t* d_gene_pbase = nullptr;
  int32_t* d_snp_ord = nullptr;     // per position and base (A, C, G, T): ordinal of that substitution, -1 when the index has none
  int32_t n_var = 0, n_gene = 0;
  std::vector<int32_t> gene_vbeg;
};

struct gk_tab {
  gk_ctx* ctx = nullptr;
  gk_index* idx = nullptr;
  int32_t n_var = 0;   // index variants (ordinals >= n_var are novel)
  int64_t n_pairs = 0, n_valid = 0, n_ids = 0;
  int32_t n_novel = 0, err_flags = 0;
  int32_t* d_pair_src = nullptr;
  uint32_t* d_off = nullptr;
  uint32_t* d_ids = nullptr;
  uint8_t* d_pair_gene = nullptr;
  uint8_t* d_pair_nh = nullptr;
  uint64_t* d_novel_key = nullptr;
  gk_mate_wide* d_wide = nullptr;   // pairs in the wide format (2 records each): gk_depth reads their CIGARs
  int64_t n_spill = 0;
  // rows grouped by backbone in row order, built once per (multiple) flavour on first use:
  // [0] reads mapped to one backbone only, [1] every read
  struct GenePartition {
    gk_ctx* owner = nullptr;          // context whose pool holds d_rows
    int32_t* d_rows = nullptr;
    std::vector<int64_t> gene_off;    // [n_bins + 1]
  } part[2];
  std::mutex part_mutex;
};

template <typename T>
static inline T* gk_ptr(gk_dptr p) {
  return reinterpret_cast<T*>(static_cast<uintptr_t>(p));
}
static inline gk_dptr gk_addr(const void* p) { return static_cast<gk_dptr>(reinterpret_cast<uintptr_t>(p)); }

// ---- device scan / compaction primitives (gk_scan.hip)
// exclusive scan of uint32 in place; total written to *d_total (device) if non-null
int gk_scan_u32(gk_ctx* ctx, uint32_t* d_data, int64_t n, uint32_t* d_total);
// stable compaction: out[k] = values[i] (or i when values == nullptr) for the k-th i with flag[i] != 0
int gk_compact_enqueue(gk_ctx* ctx, const uint32_t* d_flag, const int32_t* d_values, int64_t n, int32_t* d_out,
                       uint32_t** d_total_out, std::vector<void*>& temps);
int gk_compact(gk_ctx* ctx, const uint32_t* d_flag, const int32_t* d_values, int64_t n, int32_t* d_out,
               int64_t* n_out);

// ---- packed key helpers (host + device)
#define GK_KEY_REF_SHIFT 56
#define GK_KEY_POS_SHIFT 32
#define GK_KEY_TYP_SHIFT 30
#define GK_KEY_VAL_MASK 0x3FFFFFFFull
#define GK_TYP_INS 0ull
#define GK_TYP_SINGLE 1ull
#define GK_TYP_DEL 2ull
__host__ __device__ static inline uint64_t gk_make_key(uint32_t ref, uint32_t pos, uint64_t typ, uint32_t val) {
  return ((uint64_t)ref << GK_KEY_REF_SHIFT) | ((uint64_t)(pos & 0xFFFFFFu) << GK_KEY_POS_SHIFT) |
         (typ << GK_KEY_TYP_SHIFT) | ((uint64_t)val & GK_KEY_VAL_MASK);
}
__host__ __device__ static inline uint32_t gk_key_pos(uint64_t k) { return (uint32_t)(k >> GK_KEY_POS_SHIFT) & 0xFFFFFFu; }
__host__ __device__ static inline uint32_t gk_key_typ(uint64_t k) { return (uint32_t)(k >> GK_KEY_TYP_SHIFT) & 3u; }
__host__ __device__ static inline uint32_t gk_key_val(uint64_t k) { return (uint32_t)(k & GK_KEY_VAL_MASK); }

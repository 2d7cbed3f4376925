// Runtime plumbing of the C ABI: context, memory, timing, errors.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>

#include <atomic>
#include <mutex>
#include <cstring>
#include <string>

#include "gk_common.h"

static thread_local char g_err[512] = "";

// every context of the process (their pools are flushed together when the device runs out of memory) and the bytes their
// pools hold idle
static std::mutex g_ctx_mutex;
static std::vector<gk_ctx*>& all_contexts() {
  static std::vector<gk_ctx*> v;
  return v;
}
static std::atomic<size_t> g_pool_cached{0};

// large blocks live in one pool per device (see gk_pool_malloc)
constexpr size_t kBigBlock = (size_t)32 << 20;
struct BigBlock { void* p; hipEvent_t ev; gk_ctx* by; };
struct DevicePool {
  std::mutex m;
  std::multimap<size_t, BigBlock> idle;
  std::vector<hipEvent_t> events;
  size_t cached = 0;
};
static DevicePool& big_pool(int device) {
  static DevicePool pools[64];
  return pools[device & 63];
}


void gk_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" {

int gk_abi_version(void) { return GK_ABI_VERSION; }
const char* gk_last_error(void) { return g_err; }

int gk_device_count(int* n) {
  GK_REQUIRE(n, "null pointer");
  int c = 0;
  hipError_t e = hipGetDeviceCount(&c);
  if (e != hipSuccess) {
    *n = 0;
    gk_set_error("hipGetDeviceCount: %s", hipGetErrorString(e));
    return GK_ERR_NO_DEVICE;
  }
  *n = c;
  return GK_OK;
}

int gk_ctx_create(int device, gk_ctx** out) { return gk_ctx_create_priority(device, 0, out); }

int gk_ctx_create_priority(int device, int urgent, gk_ctx** out) {
  GK_REQUIRE(out, "null pointer");
  int c = 0;
  if (hipGetDeviceCount(&c) != hipSuccess || c <= 0) {
    gk_set_error("no HIP device visible: the typing path has no CPU fallback");
    return GK_ERR_NO_DEVICE;
  }
  GK_REQUIRE(device >= 0 && device < c, "device ordinal out of range");
  GK_HIP(hipSetDevice(device));
  {
    // How a host thread waits for the GPU (GK_WAIT_POLICY = spin | yield | block; default: the runtime's choice).
    // Set once per process and device, before the first stream exists; an error (flags already fixed) is ignored.
    static std::once_flag once;
    std::call_once(once, [] {
      const char* e = getenv("GK_WAIT_POLICY");
      if (!e) return;
      const unsigned flags = !strcmp(e, "spin") ? hipDeviceScheduleSpin : !strcmp(e, "yield") ? hipDeviceScheduleYield
                             : !strcmp(e, "block") ? hipDeviceScheduleBlockingSync : hipDeviceScheduleAuto;
      (void)hipSetDeviceFlags(flags);
      (void)hipGetLastError();
    });
  }
  gk_ctx* ctx = new gk_ctx();
  ctx->device = device;
  if (urgent) {
    int least = 0, greatest = 0;           // numerically lower = more urgent
    GK_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
    GK_HIP(hipStreamCreateWithPriority(&ctx->stream, hipStreamNonBlocking, greatest));
  } else {
    GK_HIP(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
  }
  GK_HIP(hipEventCreate(&ctx->ev0));
  GK_HIP(hipEventCreate(&ctx->ev1));
  {
    std::lock_guard<std::mutex> lock(g_ctx_mutex);
    all_contexts().push_back(ctx);
  }
  *out = ctx;
  return GK_OK;
}

int gk_ctx_destroy(gk_ctx* ctx) {
  gk_bind(ctx);
  if (!ctx) return GK_OK;
  hipSetDevice(ctx->device);
  hipStreamSynchronize(ctx->stream);
  {
    std::lock_guard<std::mutex> lock(g_ctx_mutex);
    auto& v = all_contexts();
    v.erase(std::remove(v.begin(), v.end(), ctx), v.end());
  }
  g_pool_cached -= ctx->pool_cached_bytes;
  {
    DevicePool& dp = big_pool(ctx->device);      // its idle large blocks stay for the others (the stream has drained)
    std::lock_guard<std::mutex> lock(dp.m);
    for (auto& kv : dp.idle)
      if (kv.second.by == ctx) kv.second.by = nullptr;
  }
  for (auto& kv : ctx->pool_free) hipFree(kv.second);
  for (auto& kv : ctx->pool_live) hipFree(kv.first);
  if (ctx->scratch) hipFree(ctx->scratch);
  if (ctx->tickets) hipFree(ctx->tickets);
  if (ctx->send_ring.base) hipHostFree(ctx->send_ring.base);
  if (ctx->fetch_ring.base) hipHostFree(ctx->fetch_ring.base);
  for (const auto& m : ctx->marks) hipEventDestroy(m.ev);
  for (hipEvent_t ev : ctx->mark_pool) hipEventDestroy(ev);
  hipEventDestroy(ctx->ev0);
  hipEventDestroy(ctx->ev1);
  hipStreamDestroy(ctx->stream);
  delete ctx;
  return GK_OK;
}

int gk_sync(gk_ctx* ctx) {
  gk_bind(ctx);
  GK_REQUIRE(ctx, "null context");
  GK_HIP(hipStreamSynchronize(ctx->stream));
  return GK_OK;
}

int gk_malloc(gk_ctx* ctx, size_t bytes, gk_dptr* out) {
  gk_bind(ctx);
  GK_REQUIRE(ctx && out, "null pointer");
  GK_HIP(hipSetDevice(ctx->device));
  void* p = nullptr;
  GK_HIP(gk_pool_malloc(ctx, &p, bytes ? bytes : 16));
  *out = gk_addr(p);
  return GK_OK;
}

int gk_free(gk_ctx* ctx, gk_dptr p) {
  gk_bind(ctx);
  GK_REQUIRE(ctx, "null context");
  if (!p) return GK_OK;
  gk_pool_free(ctx, gk_ptr<void>(p));
  return GK_OK;
}

int gk_memset(gk_ctx* ctx, gk_dptr p, int value, size_t bytes) {
  gk_bind(ctx);
  GK_REQUIRE(ctx, "null context");
  if (!bytes) return GK_OK;
  GK_HIP(hipMemsetAsync(gk_ptr<void>(p), value, bytes, ctx->stream));
  return GK_OK;
}

int gk_h2d(gk_ctx* ctx, gk_dptr dst, const void* src, size_t bytes) {
  gk_bind(ctx);
  GK_REQUIRE(ctx, "null context");
  if (!bytes) return GK_OK;
  GK_HIP(gk_send(ctx, gk_ptr<void>(dst), src, bytes));
  GK_HIP(hipStreamSynchronize(ctx->stream));
  return GK_OK;
}

int gk_d2h(gk_ctx* ctx, void* dst, gk_dptr src, size_t bytes) {
  gk_bind(ctx);
  GK_REQUIRE(ctx, "null context");
  if (!bytes) return GK_OK;
  GK_HIP(gk_fetch(ctx, dst, gk_ptr<void>(src), bytes));
  return GK_OK;
}

int gk_d2d(gk_ctx* ctx, gk_dptr dst, gk_dptr src, size_t bytes) {
  gk_bind(ctx);
  GK_REQUIRE(ctx, "null context");
  if (!bytes) return GK_OK;
  GK_HIP(hipMemcpyAsync(gk_ptr<void>(dst), gk_ptr<void>(src), bytes, hipMemcpyDeviceToDevice, ctx->stream));
  return GK_OK;
}

int gk_timer_start(gk_ctx* ctx) {
  gk_bind(ctx);
  GK_REQUIRE(ctx, "null context");
  GK_HIP(hipEventRecord(ctx->ev0, ctx->stream));
  return GK_OK;
}

int gk_timer_stop_ms(gk_ctx* ctx, float* ms) {
  gk_bind(ctx);
  GK_REQUIRE(ctx && ms, "null pointer");
  GK_HIP(hipEventRecord(ctx->ev1, ctx->stream));
  GK_HIP(hipEventSynchronize(ctx->ev1));
  GK_HIP(hipEventElapsedTime(ms, ctx->ev0, ctx->ev1));
  return GK_OK;
}

}  // extern "C"

// ---- pinned staging (see gk_ctx)
constexpr size_t kStageDirect = (size_t)4 << 20;   // larger transfers go straight to / from the caller's memory
size_t gk_stage_direct() { return kStageDirect; }


// ---- the two rings and the marks that give their space back
namespace {

bool wait_blocks() {        // GK_WAIT_POLICY=block: events that put the waiting thread to sleep
  static const bool v = [] { const char* e = getenv("GK_WAIT_POLICY"); return e && !strcmp(e, "block"); }();
  return v;
}

void deliver_upto(gk_ctx* ctx, uint64_t fetch_head) {
  while (!ctx->fetches.empty() && ctx->fetches.front().end <= fetch_head) {
    const auto& f = ctx->fetches.front();
    memcpy(f.dst, (const char*)ctx->fetch_ring.base + f.off, f.bytes);
    ctx->fetches.pop_front();
  }
  ctx->fetch_ring.tail = std::max(ctx->fetch_ring.tail, fetch_head);
}

// the oldest `n` marks are known to have passed
void complete_marks(gk_ctx* ctx, size_t n) {
  for (size_t k = 0; k < n && !ctx->marks.empty(); ++k) {
    const gk_ctx::Mark m = ctx->marks.front();
    ctx->marks.pop_front();
    ctx->send_ring.tail = std::max(ctx->send_ring.tail, m.send_head);
    deliver_upto(ctx, m.fetch_head);
    ctx->mark_done = m.id;
    ctx->mark_pool.push_back(m.ev);
  }
}

hipError_t drain(gk_ctx* ctx, bool deliver) {
  hipError_t e = hipStreamSynchronize(ctx->stream);
  if (deliver) deliver_upto(ctx, ctx->fetch_ring.head);
  ctx->fetches.clear();
  ctx->fetch_ring.tail = ctx->fetch_ring.head;
  ctx->send_ring.tail = ctx->send_ring.head;
  for (const auto& m : ctx->marks) ctx->mark_pool.push_back(m.ev);
  ctx->marks.clear();
  ctx->mark_done = ctx->mark_next - 1;
  return e;
}

// `need` contiguous bytes of a ring: waits for the oldest mark (or the whole stream) while the ring is full
hipError_t ring_take(gk_ctx* ctx, gk_ctx::Ring& r, size_t need, size_t min_bytes, size_t* off) {
  for (;;) {
    if (r.bytes >= need) {
      uint64_t head = r.head;
      const size_t phys = (size_t)(head % r.bytes);
      if (phys + need > r.bytes) head += r.bytes - phys;          // does not fit before the end: start over at 0
      if (head + need - r.tail <= r.bytes) {
        *off = (size_t)(head % r.bytes);
        r.head = head + need;
        return hipSuccess;
      }
    }
    if (r.head != r.tail) {                 // something is in flight: let the oldest part of it pass
      hipError_t e;
      if (ctx->marks.empty()) {
        e = drain(ctx, true);
      } else {
        e = hipEventSynchronize(ctx->marks.front().ev);
        complete_marks(ctx, 1);
      }
      if (e != hipSuccess) return e;
      continue;
    }
    // empty and too small: a larger area (nothing refers to the old one)
    if (r.base) hipHostFree(r.base);
    r.base = nullptr;
    r.bytes = std::max<size_t>(need * 4, min_bytes);
    r.head = r.tail = 0;
    hipError_t e = hipHostMalloc(&r.base, r.bytes, hipHostMallocDefault);
    if (e != hipSuccess) { r.bytes = 0; return e; }
  }
}

}  // namespace

hipError_t gk_send(gk_ctx* ctx, void* dst_dev, const void* src, size_t bytes) {
  if (!bytes) return hipSuccess;
  if (bytes > kStageDirect) return hipMemcpyAsync(dst_dev, src, bytes, hipMemcpyHostToDevice, ctx->stream);
  size_t off = 0;
  hipError_t e = ring_take(ctx, ctx->send_ring, (bytes + 63) / 64 * 64, (size_t)8 << 20, &off);
  if (e != hipSuccess) return e;
  char* slot = (char*)ctx->send_ring.base + off;
  memcpy(slot, src, bytes);
  return hipMemcpyAsync(dst_dev, slot, bytes, hipMemcpyHostToDevice, ctx->stream);
}

hipError_t gk_fetch_queue(gk_ctx* ctx, void* dst, const void* src_dev, size_t bytes) {
  if (!bytes) return hipSuccess;
  if (bytes > kStageDirect) return hipMemcpyAsync(dst, src_dev, bytes, hipMemcpyDeviceToHost, ctx->stream);
  size_t off = 0;
  hipError_t e = ring_take(ctx, ctx->fetch_ring, (bytes + 63) / 64 * 64, (size_t)8 << 20, &off);
  if (e != hipSuccess) return e;
  ctx->fetches.push_back({dst, off, bytes, ctx->fetch_ring.head});
  return hipMemcpyAsync((char*)ctx->fetch_ring.base + off, src_dev, bytes, hipMemcpyDeviceToHost, ctx->stream);
}

// A result that a kernel writes ITSELF into the pinned ring (device-visible host memory): no copy is queued -- a copy per
// fetch was ~85 runtime calls and as many 5 us copy kernels per configs[1] sample.  `*dev_out` is where the kernel
// (queued on this context's stream by the caller, after this call) stores `bytes` bytes; they are delivered to `dst`
// like a queued copy when the stream has passed the next mark / wait.  Only for results up to gk_stage_direct() bytes
// that ONE kernel writes once (state that kernels update with atomics stays in device memory and is copied).
hipError_t gk_fetch_direct(gk_ctx* ctx, void* dst, size_t bytes, void** dev_out) {
  *dev_out = nullptr;
  if (!bytes || bytes > kStageDirect) return hipErrorInvalidValue;
  size_t off = 0;
  hipError_t e = ring_take(ctx, ctx->fetch_ring, (bytes + 63) / 64 * 64, (size_t)8 << 20, &off);
  if (e != hipSuccess) return e;
  ctx->fetches.push_back({dst, off, bytes, ctx->fetch_ring.head});
  *dev_out = (char*)ctx->fetch_ring.base + off;
  return hipSuccess;
}

hipError_t gk_fetch_wait(gk_ctx* ctx) { return drain(ctx, true); }

void gk_fetch_cancel(gk_ctx* ctx) { (void)drain(ctx, false); }

hipError_t gk_fetch_mark(gk_ctx* ctx, uint64_t* mark) {
  hipEvent_t ev = nullptr;
  if (!ctx->mark_pool.empty()) {
    ev = ctx->mark_pool.back();
    ctx->mark_pool.pop_back();
  } else {
    hipError_t e = hipEventCreateWithFlags(&ev, hipEventDisableTiming | (wait_blocks() ? hipEventBlockingSync : 0));
    if (e != hipSuccess) return e;
  }
  hipError_t e = hipEventRecord(ev, ctx->stream);
  if (e != hipSuccess) { ctx->mark_pool.push_back(ev); return e; }
  ctx->marks.push_back({ev, ctx->mark_next, ctx->send_ring.head, ctx->fetch_ring.head});
  *mark = ctx->mark_next++;
  return hipSuccess;
}

hipError_t gk_fetch_wait_mark(gk_ctx* ctx, uint64_t mark) {
  if (mark <= ctx->mark_done) return hipSuccess;          // passed already (a wait for a later mark, or a drain)
  size_t n = 0;
  while (n < ctx->marks.size() && ctx->marks[n].id <= mark) ++n;
  if (n == 0) return hipSuccess;
  hipError_t e = hipEventSynchronize(ctx->marks[n - 1].ev);   // one stream: the earlier marks have passed too
  if (e != hipSuccess) return e;
  complete_marks(ctx, n);
  return hipSuccess;
}

static size_t pool_class(size_t bytes) {
  if (bytes < 256) return 256;
  if (bytes <= (1u << 20)) {   // powers of two up to 1 MiB
    size_t c = 256;
    while (c < bytes) c <<= 1;
    return c;
  }
  // then multiples of 1 MiB up to 64 MiB, beyond that of 1/16 of the size's power of two (<= 6 % over): the tables of a
  // sample are sized by its read counts, which differ from sample to sample -- coarse classes let the next sample's
  // tables land in this sample's blocks instead of in fresh hipMalloc's (a 1 GB allocation costs milliseconds)
  size_t step = 1u << 20;
  if (bytes > ((size_t)64 << 20)) {
    size_t p2 = (size_t)1 << 26;
    while ((p2 << 1) <= bytes) p2 <<= 1;
    step = p2 >> 4;
  }
  return (bytes + step - 1) / step * step;
}

// ---- large blocks: ONE pool per device, shared by the contexts of the process.  The lanes of a process type samples of
// different sizes: a pool per context kept, idle, a set of gigabyte tables per lane that no other lane could use (five
// lanes of 20 M-read samples typed exon-first: more than the card).  A large block goes back with an EVENT recorded on the
// stream that last used it; a context that takes a block another context freed makes its stream wait for that event
// (hipStreamWaitEvent: a dependency on the GPU, no host wait), so reuse stays stream-ordered across streams.
// idle large blocks a process may keep per device (GK_POOL_CACHE_GB; default: no limit -- they are reused by whichever
// context asks next, and all of them go back when the device runs out): beyond it the largest idle blocks are released
static size_t pool_cache_limit() {
  static const size_t v = [] {
    const char* e = getenv("GK_POOL_CACHE_GB");
    if (!e) return (size_t)-1;
    return (size_t)(std::max(atof(e), 0.0) * (double)(1ull << 30));
  }();
  return v;
}

// the idle blocks of one context's own (small-block) pool back to the device; the caller holds no pool lock
static size_t flush_pool(gk_ctx* c) {
  std::lock_guard<std::mutex> lock(c->pool_mutex);
  if (c->pool_free.empty()) return 0;
  hipStreamSynchronize(c->stream);      // a cached block may still be read by work queued before it was freed
  const size_t bytes = c->pool_cached_bytes;
  for (auto& kv : c->pool_free) hipFree(kv.second);
  c->pool_free.clear();
  c->pool_cached_bytes = 0;
  g_pool_cached -= bytes;
  return bytes;
}

// idle large blocks of a device back to it, the largest first, until at most `keep` bytes stay
static size_t release_big(int device, size_t keep) {
  DevicePool& dp = big_pool(device);
  std::lock_guard<std::mutex> lock(dp.m);
  size_t given = 0;
  while (!dp.idle.empty() && dp.cached > keep) {
    auto it = std::prev(dp.idle.end());
    hipEventSynchronize(it->second.ev);
    hipFree(it->second.p);
    dp.events.push_back(it->second.ev);
    dp.cached -= it->first;
    g_pool_cached -= it->first;
    given += it->first;
    dp.idle.erase(it);
  }
  return given;
}

hipError_t gk_pool_malloc(gk_ctx* ctx, void** out, size_t bytes) {
  const size_t cls = pool_class(bytes);
  static const bool trace = gk_trace("pool");      // dev: every large pool miss and what it costs
  const bool big = cls >= kBigBlock;
  if (big) {
    DevicePool& dp = big_pool(ctx->device);
    std::lock_guard<std::mutex> lock(dp.m);
    // the smallest idle block that holds the request, if it is not more than a quarter too large; one this context
    // freed itself if there is one among the first few (no dependency between streams then)
    auto it = dp.idle.lower_bound(cls);
    auto pick = dp.idle.end();
    for (int tries = 0; it != dp.idle.end() && it->first <= cls + cls / 4 && tries < 8; ++it, ++tries) {
      if (pick == dp.idle.end()) pick = it;
      if (it->second.by == ctx) { pick = it; break; }
    }
    if (pick != dp.idle.end()) {
      const BigBlock blk = pick->second;
      const size_t have = pick->first;
      dp.idle.erase(pick);
      dp.cached -= have;
      g_pool_cached -= have;
      if (blk.by != ctx) (void)hipStreamWaitEvent(ctx->stream, blk.ev, 0);
      dp.events.push_back(blk.ev);
      *out = blk.p;
      std::lock_guard<std::mutex> mine(ctx->pool_mutex);
      ctx->pool_live[*out] = have;
      return hipSuccess;
    }
  } else {
    std::lock_guard<std::mutex> lock(ctx->pool_mutex);
    auto it = ctx->pool_free.lower_bound(cls);
    if (it != ctx->pool_free.end() && it->first <= cls + cls / 4) {
      const size_t have = it->first;
      *out = it->second;
      ctx->pool_free.erase(it);
      ctx->pool_cached_bytes -= have;
      g_pool_cached -= have;
      ctx->pool_live[*out] = have;
      return hipSuccess;
    }
  }
  const auto t0 = std::chrono::steady_clock::now();
  hipError_t e = hipMalloc(out, cls);
  if (trace && big)
    fprintf(stderr, "[gk_pool] ctx %p: hipMalloc of %zu MB took %.0f us (%zu MB idle in all pools)\n", (void*)ctx, cls >> 20,
            std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count(), g_pool_cached.load() >> 20);
  if (e != hipSuccess) {
    // out of device memory: every idle block of the process on this device goes back, then once more
    (void)hipGetLastError();
    size_t given = release_big(ctx->device, 0);
    {
      std::lock_guard<std::mutex> lock(g_ctx_mutex);
      for (gk_ctx* c : all_contexts())
        if (c->device == ctx->device) given += flush_pool(c);
    }
    if (trace) fprintf(stderr, "[gk_pool] ctx %p: hipMalloc of %zu MB FAILED, %zu MB of idle blocks given back\n", (void*)ctx,
                       cls >> 20, given >> 20);
    e = hipMalloc(out, cls);
  }
  if (e == hipSuccess) {
    std::lock_guard<std::mutex> lock(ctx->pool_mutex);
    ctx->pool_live[*out] = cls;
  }
  return e;
}

void gk_pool_free(gk_ctx* ctx, void* p) {
  if (!p) return;
  size_t cls = 0;
  {
    std::lock_guard<std::mutex> lock(ctx->pool_mutex);
    auto it = ctx->pool_live.find(p);
    if (it == ctx->pool_live.end()) {   // not ours
      hipFree(p);
      return;
    }
    cls = it->second;
    ctx->pool_live.erase(it);
    if (cls < kBigBlock) {
      ctx->pool_free.emplace(cls, p);
      ctx->pool_cached_bytes += cls;
      g_pool_cached += cls;
      return;
    }
  }
  (void)hipSetDevice(ctx->device);      // frees may come from a thread that never chose a device (Python's collector)
  DevicePool& dp = big_pool(ctx->device);
  bool over = false;
  {
    std::lock_guard<std::mutex> lock(dp.m);
    hipEvent_t ev = nullptr;
    if (!dp.events.empty()) { ev = dp.events.back(); dp.events.pop_back(); }
    else if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) ev = nullptr;
    if (!ev || hipEventRecord(ev, ctx->stream) != hipSuccess) {      // cannot mark the point of the stream: give it back
      if (ev) dp.events.push_back(ev);
      (void)hipGetLastError();
      hipStreamSynchronize(ctx->stream);
      hipFree(p);
      return;
    }
    dp.idle.emplace(cls, BigBlock{p, ev, ctx});
    dp.cached += cls;
    g_pool_cached += cls;
    over = dp.cached > pool_cache_limit();
  }
  if (over) release_big(ctx->device, pool_cache_limit());
}

extern "C" int gk_device_memory(gk_ctx* ctx, int64_t* free_bytes, int64_t* total_bytes, int64_t* pool_cached_bytes) {
  gk_bind(ctx);
  GK_REQUIRE(ctx && free_bytes && total_bytes, "null pointer");
  size_t f = 0, t = 0;
  GK_HIP(hipMemGetInfo(&f, &t));
  *free_bytes = (int64_t)f;
  *total_bytes = (int64_t)t;
  if (pool_cached_bytes) *pool_cached_bytes = (int64_t)g_pool_cached.load();
  return GK_OK;
}

static thread_local gk_ctx::ProfSpan* t_span = nullptr;   // span whose events the next GK_KERNEL launch takes

hipEvent_t gk_prof_start_event() { return t_span ? t_span->a : nullptr; }
hipEvent_t gk_prof_stop_event() { return t_span ? t_span->b : nullptr; }

// exact != 0: the events are bound to the kernel itself (GK_KERNEL hands them to
// hipExtLaunchKernelGGL); otherwise they are recorded on the stream before and after the launch.
void gk_prof_begin(gk_ctx* ctx, int id, int exact) {
  t_span = nullptr;
  if (!ctx->prof_on) return;
  gk_ctx::ProfSpan sp;
  sp.id = id;
  for (hipEvent_t* e : {&sp.a, &sp.b}) {
    if (!ctx->prof_pool.empty()) {
      *e = ctx->prof_pool.back();
      ctx->prof_pool.pop_back();
    } else {
      hipEventCreate(e);
    }
  }
  ctx->prof_spans.push_back(sp);
  if (exact) t_span = &ctx->prof_spans.back();
  else hipEventRecord(sp.a, ctx->stream);
}

void gk_prof_end(gk_ctx* ctx) {
  if (t_span) { t_span = nullptr; return; }
  if (!ctx->prof_on || ctx->prof_spans.empty()) return;
  hipEventRecord(ctx->prof_spans.back().b, ctx->stream);
}

// names of the timed kernels, in the order they were first launched
static std::mutex g_prof_names_mutex;
static std::vector<std::string>& prof_names() {
  static std::vector<std::string> v;
  return v;
}

int gk_prof_register(const char* name) {
  std::lock_guard<std::mutex> lock(g_prof_names_mutex);
  auto& v = prof_names();
  for (size_t i = 0; i < v.size(); ++i)
    if (v[i] == name) return (int)i;
  if ((int)v.size() >= GK_PROF_MAX) return GK_PROF_MAX - 1;     // cannot happen with the kernels this library has
  v.emplace_back(name);
  return (int)v.size() - 1;
}

extern "C" int gk_prof_enable(gk_ctx* ctx, int on) {
  gk_bind(ctx);
  GK_REQUIRE(ctx, "null context");
  ctx->prof_on = on != 0;
  return GK_OK;
}

// the capacity of the two arrays gk_prof_collect fills (names register as kernels are first launched)
extern "C" int gk_prof_kernel_count(void) { return GK_PROF_MAX; }
extern "C" const char* gk_prof_kernel_name(int id) {
  static thread_local std::string name;
  std::lock_guard<std::mutex> lock(g_prof_names_mutex);
  name = id >= 0 && id < (int)prof_names().size() ? prof_names()[(size_t)id] : "";
  return name.c_str();
}

// launches[id], total_ms[id] for id < gk_prof_kernel_count(); clears the recorded spans
extern "C" int gk_prof_collect(gk_ctx* ctx, int64_t* launches, double* total_ms) {
  gk_bind(ctx);
  GK_REQUIRE(ctx && launches && total_ms, "null pointer");
  GK_HIP(hipStreamSynchronize(ctx->stream));
  for (int i = 0; i < GK_PROF_MAX; ++i) { launches[i] = 0; total_ms[i] = 0.0; }
  for (auto& sp : ctx->prof_spans) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, sp.a, sp.b) == hipSuccess) {
      launches[sp.id] += 1;
      total_ms[sp.id] += ms;
    }
    ctx->prof_pool.push_back(sp.a);
    ctx->prof_pool.push_back(sp.b);
  }
  ctx->prof_spans.clear();
  (void)hipGetLastError();   // a span whose kernel never ran leaves an error behind; it is not the caller's
  return GK_OK;
}

int gk_ctx_tickets(gk_ctx* ctx, size_t n, uint32_t** out) {
  gk_bind(ctx);
  if (n > ctx->n_tickets) {
    GK_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->tickets) GK_HIP(hipFree(ctx->tickets));
    const size_t want = std::max<size_t>(4096, n * 2);
    GK_HIP(hipMalloc((void**)&ctx->tickets, want * sizeof(uint32_t)));
    GK_HIP(hipMemsetAsync(ctx->tickets, 0, want * sizeof(uint32_t), ctx->stream));
    ctx->n_tickets = want;
  }
  *out = ctx->tickets;
  return GK_OK;
}

int gk_ctx_scratch(gk_ctx* ctx, size_t bytes, void** out) {
  gk_bind(ctx);
  if (bytes > ctx->scratch_bytes) {
    GK_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->scratch) GK_HIP(hipFree(ctx->scratch));
    size_t want = bytes < (1u << 20) ? (1u << 20) : bytes * 2;
    GK_HIP(hipMalloc(&ctx->scratch, want));
    ctx->scratch_bytes = want;
  }
  *out = ctx->scratch;
  return GK_OK;
}

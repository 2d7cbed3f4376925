// Runtime plumbing of the C ABI: context, memory, timing, errors.
#include <algorithm>

#include <mutex>
#include <cstring>

#include "gk_common.h"

static thread_local char g_err[512] = "";

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

int gk_ctx_create(int device, gk_ctx** out) {
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
  GK_HIP(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
  GK_HIP(hipEventCreate(&ctx->ev0));
  GK_HIP(hipEventCreate(&ctx->ev1));
  *out = ctx;
  return GK_OK;
}

int gk_ctx_destroy(gk_ctx* ctx) {
  gk_bind(ctx);
  if (!ctx) return GK_OK;
  hipSetDevice(ctx->device);
  hipStreamSynchronize(ctx->stream);
  for (auto& kv : ctx->pool_free) hipFree(kv.second);
  for (auto& kv : ctx->pool_live) hipFree(kv.first);
  if (ctx->scratch) hipFree(ctx->scratch);
  if (ctx->pinned) hipHostFree(ctx->pinned);
  if (ctx->bounce) hipHostFree(ctx->bounce);
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

hipError_t gk_send(gk_ctx* ctx, void* dst_dev, const void* src, size_t bytes) {
  if (!bytes) return hipSuccess;
  if (bytes > kStageDirect) return hipMemcpyAsync(dst_dev, src, bytes, hipMemcpyHostToDevice, ctx->stream);
  const size_t need = (bytes + 63) / 64 * 64;
  if (ctx->pinned_bytes < need || ctx->pinned_head + need > ctx->pinned_bytes) {
    // the ring is full (or too small): every copy queued from it has left once the stream has drained
    hipError_t e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) return e;
    if (ctx->pinned_bytes < need * 4) {
      if (ctx->pinned) hipHostFree(ctx->pinned);
      ctx->pinned = nullptr;
      ctx->pinned_bytes = std::max<size_t>(need * 4, (size_t)1 << 20);
      e = hipHostMalloc(&ctx->pinned, ctx->pinned_bytes, hipHostMallocDefault);
      if (e != hipSuccess) { ctx->pinned_bytes = 0; return e; }
    }
    ctx->pinned_head = 0;
  }
  char* slot = (char*)ctx->pinned + ctx->pinned_head;
  ctx->pinned_head += need;
  memcpy(slot, src, bytes);
  return hipMemcpyAsync(dst_dev, slot, bytes, hipMemcpyHostToDevice, ctx->stream);
}

hipError_t gk_fetch_queue(gk_ctx* ctx, void* dst, const void* src_dev, size_t bytes) {
  if (!bytes) return hipSuccess;
  if (bytes > kStageDirect) return hipMemcpyAsync(dst, src_dev, bytes, hipMemcpyDeviceToHost, ctx->stream);
  const size_t need = (bytes + 63) / 64 * 64;
  if (ctx->bounce_head + need > ctx->bounce_bytes) {
    hipError_t e = gk_fetch_wait(ctx);          // deliver what is queued, then the area is free
    if (e != hipSuccess) return e;
    if (ctx->bounce_bytes < need) {
      if (ctx->bounce) hipHostFree(ctx->bounce);
      ctx->bounce = nullptr;
      ctx->bounce_bytes = std::max<size_t>(need * 2, (size_t)1 << 20);
      e = hipHostMalloc(&ctx->bounce, ctx->bounce_bytes, hipHostMallocDefault);
      if (e != hipSuccess) { ctx->bounce_bytes = 0; return e; }
    }
  }
  const size_t off = ctx->bounce_head;
  ctx->bounce_head += need;
  ctx->fetches.push_back({dst, off, bytes});
  return hipMemcpyAsync((char*)ctx->bounce + off, src_dev, bytes, hipMemcpyDeviceToHost, ctx->stream);
}

hipError_t gk_fetch_wait(gk_ctx* ctx) {
  hipError_t e = hipStreamSynchronize(ctx->stream);
  for (const auto& f : ctx->fetches) memcpy(f.dst, (const char*)ctx->bounce + f.off, f.bytes);
  ctx->fetches.clear();
  ctx->bounce_head = 0;
  return e;
}

static size_t pool_class(size_t bytes) {
  if (bytes < 256) return 256;
  if (bytes <= (1u << 20)) {   // powers of two up to 1 MiB
    size_t c = 256;
    while (c < bytes) c <<= 1;
    return c;
  }
  const size_t step = 1u << 20;  // then multiples of 1 MiB
  return (bytes + step - 1) / step * step;
}

hipError_t gk_pool_malloc(gk_ctx* ctx, void** out, size_t bytes) {
  std::lock_guard<std::mutex> lock(ctx->pool_mutex);
  const size_t cls = pool_class(bytes);
  auto it = ctx->pool_free.find(cls);
  if (it != ctx->pool_free.end()) {
    *out = it->second;
    ctx->pool_free.erase(it);
    ctx->pool_cached_bytes -= cls;
    ctx->pool_live[*out] = cls;
    return hipSuccess;
  }
  hipError_t e = hipMalloc(out, cls);
  if (e != hipSuccess && ctx->pool_cached_bytes) {   // give the cache back and retry once
    hipStreamSynchronize(ctx->stream);
    for (auto& kv : ctx->pool_free) hipFree(kv.second);
    ctx->pool_free.clear();
    ctx->pool_cached_bytes = 0;
    e = hipMalloc(out, cls);
  }
  if (e == hipSuccess) ctx->pool_live[*out] = cls;
  return e;
}

void gk_pool_free(gk_ctx* ctx, void* p) {
  if (!p) return;
  std::lock_guard<std::mutex> lock(ctx->pool_mutex);
  auto it = ctx->pool_live.find(p);
  if (it == ctx->pool_live.end()) {   // not ours
    hipFree(p);
    return;
  }
  const size_t cls = it->second;
  ctx->pool_live.erase(it);
  ctx->pool_free.emplace(cls, p);
  ctx->pool_cached_bytes += cls;
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

static const char* kKernelNames[GK_K_N] = {
    "tab_count", "tab_emit", "scan", "novel_rank", "count_ids", "select", "compat_kernel", "lut_collect",
    "lut_apply", "maxsum_chunks", "combine_chunks", "fraction_chunks", "setmax_kernel", "em_sets_kernel", "em_kernel",
    "setmin_u8", "minsum_sad", "select_cut"};

extern "C" int gk_prof_enable(gk_ctx* ctx, int on) {
  gk_bind(ctx);
  GK_REQUIRE(ctx, "null context");
  ctx->prof_on = on != 0;
  return GK_OK;
}

extern "C" int gk_prof_kernel_count(void) { return GK_K_N; }
extern "C" const char* gk_prof_kernel_name(int id) { return id >= 0 && id < GK_K_N ? kKernelNames[id] : ""; }

// launches[id], total_ms[id] for id < GK_K_N; clears the recorded spans
extern "C" int gk_prof_collect(gk_ctx* ctx, int64_t* launches, double* total_ms) {
  gk_bind(ctx);
  GK_REQUIRE(ctx && launches && total_ms, "null pointer");
  GK_HIP(hipStreamSynchronize(ctx->stream));
  for (int i = 0; i < GK_K_N; ++i) { launches[i] = 0; total_ms[i] = 0.0; }
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

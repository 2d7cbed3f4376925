// Runtime plumbing of the C ABI: context, memory, timing, errors.
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
  gk_ctx* ctx = new gk_ctx();
  ctx->device = device;
  GK_HIP(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
  GK_HIP(hipEventCreate(&ctx->ev0));
  GK_HIP(hipEventCreate(&ctx->ev1));
  *out = ctx;
  return GK_OK;
}

int gk_ctx_destroy(gk_ctx* ctx) {
  if (!ctx) return GK_OK;
  hipSetDevice(ctx->device);
  hipStreamSynchronize(ctx->stream);
  if (ctx->scratch) hipFree(ctx->scratch);
  if (ctx->pinned) hipHostFree(ctx->pinned);
  hipEventDestroy(ctx->ev0);
  hipEventDestroy(ctx->ev1);
  hipStreamDestroy(ctx->stream);
  delete ctx;
  return GK_OK;
}

int gk_sync(gk_ctx* ctx) {
  GK_REQUIRE(ctx, "null context");
  GK_HIP(hipStreamSynchronize(ctx->stream));
  return GK_OK;
}

int gk_malloc(gk_ctx* ctx, size_t bytes, gk_dptr* out) {
  GK_REQUIRE(ctx && out, "null pointer");
  GK_HIP(hipSetDevice(ctx->device));
  void* p = nullptr;
  GK_HIP(hipMalloc(&p, bytes ? bytes : 16));
  *out = gk_addr(p);
  return GK_OK;
}

int gk_free(gk_ctx* ctx, gk_dptr p) {
  GK_REQUIRE(ctx, "null context");
  if (!p) return GK_OK;
  GK_HIP(hipStreamSynchronize(ctx->stream));
  GK_HIP(hipFree(gk_ptr<void>(p)));
  return GK_OK;
}

int gk_memset(gk_ctx* ctx, gk_dptr p, int value, size_t bytes) {
  GK_REQUIRE(ctx, "null context");
  if (!bytes) return GK_OK;
  GK_HIP(hipMemsetAsync(gk_ptr<void>(p), value, bytes, ctx->stream));
  return GK_OK;
}

int gk_h2d(gk_ctx* ctx, gk_dptr dst, const void* src, size_t bytes) {
  GK_REQUIRE(ctx, "null context");
  if (!bytes) return GK_OK;
  GK_HIP(hipMemcpyAsync(gk_ptr<void>(dst), src, bytes, hipMemcpyHostToDevice, ctx->stream));
  GK_HIP(hipStreamSynchronize(ctx->stream));
  return GK_OK;
}

int gk_d2h(gk_ctx* ctx, void* dst, gk_dptr src, size_t bytes) {
  GK_REQUIRE(ctx, "null context");
  if (!bytes) return GK_OK;
  GK_HIP(hipMemcpyAsync(dst, gk_ptr<void>(src), bytes, hipMemcpyDeviceToHost, ctx->stream));
  GK_HIP(hipStreamSynchronize(ctx->stream));
  return GK_OK;
}

int gk_d2d(gk_ctx* ctx, gk_dptr dst, gk_dptr src, size_t bytes) {
  GK_REQUIRE(ctx, "null context");
  if (!bytes) return GK_OK;
  GK_HIP(hipMemcpyAsync(gk_ptr<void>(dst), gk_ptr<void>(src), bytes, hipMemcpyDeviceToDevice, ctx->stream));
  return GK_OK;
}

int gk_timer_start(gk_ctx* ctx) {
  GK_REQUIRE(ctx, "null context");
  GK_HIP(hipEventRecord(ctx->ev0, ctx->stream));
  return GK_OK;
}

int gk_timer_stop_ms(gk_ctx* ctx, float* ms) {
  GK_REQUIRE(ctx && ms, "null pointer");
  GK_HIP(hipEventRecord(ctx->ev1, ctx->stream));
  GK_HIP(hipEventSynchronize(ctx->ev1));
  GK_HIP(hipEventElapsedTime(ms, ctx->ev0, ctx->ev1));
  return GK_OK;
}

}  // extern "C"

int gk_ctx_scratch(gk_ctx* ctx, size_t bytes, void** out) {
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

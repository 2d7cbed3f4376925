// Cohort collective on RCCL (one rank per GPU, xGMI inside a node).
//
// The only exchange of the path is the pooling of gene depths for `--cn-cohort`
// (graphkir/kir_cn.py:61, 167-177: the reference concatenates the depths of ALL samples before ONE fit;
// main.py:572-589).  With the samples sharded over ranks that is one all-gather of a few hundred bytes
// per rank: latency bound, so the payload is staged through one small device buffer per communicator.
// bench.py uses the same communicator for its barrier and its max-over-ranks time.
//
// librccl.so is opened on first use (dlopen): single-process runs never load it, and the library has no
// link-time dependency on it.  The ncclUniqueId travels between the ranks through the host-side
// rendezvous of kir_graph_amd/comm.py (a directory of small files; one node).
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstring>
#include <cstdlib>
#include "gk_common.h"

namespace {

struct Rccl {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

Rccl g_rccl;
std::mutex g_rccl_mutex;

int load_rccl() {
  std::lock_guard<std::mutex> lock(g_rccl_mutex);
  if (g_rccl.handle) return GK_OK;
  void* h = nullptr;
  for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
    h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
    if (h) break;
  }
  if (!h) {
    gk_set_error("librccl.so not found: %s", dlerror());
    return GK_ERR_HIP;
  }
#define GK_SYM(field, sym)                                        \
  g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(h, sym)); \
  if (!g_rccl.field) {                                            \
    gk_set_error("librccl.so lacks %s", sym);                     \
    dlclose(h);                                                   \
    return GK_ERR_HIP;                                            \
  }
  GK_SYM(GetUniqueId, "ncclGetUniqueId")
  GK_SYM(CommInitRank, "ncclCommInitRank")
  GK_SYM(CommDestroy, "ncclCommDestroy")
  GK_SYM(AllGather, "ncclAllGather")
  GK_SYM(AllReduce, "ncclAllReduce")
  GK_SYM(GetErrorString, "ncclGetErrorString")
#undef GK_SYM
  g_rccl.handle = h;
  return GK_OK;
}

#define GK_NCCL(call)                                                                      \
  do {                                                                                     \
    ncclResult_t r_ = (call);                                                              \
    if (r_ != ncclSuccess) {                                                               \
      gk_set_error("%s failed: %s (%s:%d)", #call, g_rccl.GetErrorString(r_), __FILE__, __LINE__); \
      return GK_ERR_HIP;                                                                   \
    }                                                                                      \
  } while (0)

}  // namespace

struct gk_comm {
  gk_ctx* ctx = nullptr;
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1;
  double* d_buf = nullptr;   // [send | recv] staging in HBM
  size_t buf_doubles = 0;
};

static int comm_stage(gk_comm* c, size_t doubles) {
  if (doubles <= c->buf_doubles) return GK_OK;
  if (c->d_buf) GK_HIP(hipFree(c->d_buf));
  c->buf_doubles = std::max<size_t>(doubles * 2, 1024);
  GK_HIP(hipMalloc((void**)&c->d_buf, c->buf_doubles * sizeof(double)));
  return GK_OK;
}

extern "C" {

int gk_comm_unique_id(void* id_out, size_t capacity) {
  GK_REQUIRE(id_out && capacity >= sizeof(ncclUniqueId), "unique id buffer too small (128 bytes)");
  int rc = load_rccl();
  if (rc) return rc;
  ncclUniqueId id;
  GK_NCCL(g_rccl.GetUniqueId(&id));
  memcpy(id_out, &id, sizeof(id));
  return GK_OK;
}

int gk_comm_create(gk_ctx* ctx, const void* id, size_t id_bytes, int32_t rank, int32_t world, gk_comm** out) {
  gk_bind(ctx);
  GK_REQUIRE(ctx && id && out && id_bytes == sizeof(ncclUniqueId), "bad communicator arguments");
  GK_REQUIRE(world >= 1 && rank >= 0 && rank < world, "rank outside the world");
  int rc = load_rccl();
  if (rc) return rc;
  ncclUniqueId uid;
  memcpy(&uid, id, sizeof(uid));
  gk_comm* c = new gk_comm();
  c->ctx = ctx;
  c->rank = rank;
  c->world = world;
  ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, uid, rank);
  if (r != ncclSuccess) {
    gk_set_error("ncclCommInitRank failed: %s", g_rccl.GetErrorString(r));
    delete c;
    return GK_ERR_HIP;
  }
  *out = c;
  return GK_OK;
}

int gk_comm_destroy(gk_comm* c) {
  if (!c) return GK_OK;
  gk_bind(c->ctx);
  hipStreamSynchronize(c->ctx->stream);
  if (c->comm) g_rccl.CommDestroy(c->comm);
  if (c->d_buf) hipFree(c->d_buf);
  delete c;
  return GK_OK;
}

// recv[r * n .. (r + 1) * n) = send of rank r, on every rank (host buffers)
int gk_allgather_f64(gk_comm* c, const double* send, double* recv, int64_t n) {
  GK_REQUIRE(c && send && recv && n > 0, "bad all-gather arguments");
  gk_bind(c->ctx);
  hipStream_t st = c->ctx->stream;
  const size_t total = (size_t)n * (size_t)(c->world + 1);
  int rc = comm_stage(c, total);
  if (rc) return rc;
  double *d_send = c->d_buf, *d_recv = c->d_buf + n;
  GK_HIP(hipMemcpyAsync(d_send, send, (size_t)n * sizeof(double), hipMemcpyHostToDevice, st));
  GK_NCCL(g_rccl.AllGather(d_send, d_recv, (size_t)n, ncclDouble, c->comm, st));
  GK_HIP(hipMemcpyAsync(recv, d_recv, (size_t)n * c->world * sizeof(double), hipMemcpyDeviceToHost, st));
  GK_HIP(hipStreamSynchronize(st));
  return GK_OK;
}

// element-wise maximum over the ranks, in place (host buffer)
int gk_allreduce_max_f64(gk_comm* c, double* inout, int64_t n) {
  GK_REQUIRE(c && inout && n > 0, "bad all-reduce arguments");
  gk_bind(c->ctx);
  hipStream_t st = c->ctx->stream;
  int rc = comm_stage(c, (size_t)n);
  if (rc) return rc;
  GK_HIP(hipMemcpyAsync(c->d_buf, inout, (size_t)n * sizeof(double), hipMemcpyHostToDevice, st));
  GK_NCCL(g_rccl.AllReduce(c->d_buf, c->d_buf, (size_t)n, ncclDouble, ncclMax, c->comm, st));
  GK_HIP(hipMemcpyAsync(inout, c->d_buf, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, st));
  GK_HIP(hipStreamSynchronize(st));
  return GK_OK;
}

// every rank has reached this point (and the work queued on the context's stream is done)
int gk_comm_barrier(gk_comm* c) {
  double one = 1.0;
  return gk_allreduce_max_f64(c, &one, 1);
}

// pinned host memory (records of a sample on their way to HBM) and a copy that does not wait
int gk_host_alloc(size_t bytes, void** out) {
  GK_REQUIRE(out, "null pointer");
  GK_HIP(hipHostMalloc(out, bytes ? bytes : 16, hipHostMallocDefault));
  return GK_OK;
}

int gk_host_free(void* p) {
  if (p) GK_HIP(hipHostFree(p));
  return GK_OK;
}

int gk_h2d_async(gk_ctx* ctx, gk_dptr dst, const void* src, size_t bytes) {
  gk_bind(ctx);
  GK_REQUIRE(ctx, "null context");
  if (!bytes) return GK_OK;
  // A bulk copy goes out in pieces of 8 MB: the DMA engine that serves host-to-device
  // copies also carries the small parameter blocks of the samples being typed, and it arbitrates between queues at
  // command boundaries -- behind ONE 700 MB command (a configs[2] sample) every search stage of the other lanes waited
  // for the whole transfer.
  constexpr size_t chunk = (size_t)8 << 20;
  if (!chunk || bytes <= chunk) {
    GK_HIP(hipMemcpyAsync(gk_ptr<void>(dst), src, bytes, hipMemcpyHostToDevice, ctx->stream));
    return GK_OK;
  }
  for (size_t at = 0; at < bytes; at += chunk) {
    const size_t n = std::min(chunk, bytes - at);
    GK_HIP(hipMemcpyAsync(gk_ptr<char>(dst) + at, (const char*)src + at, n, hipMemcpyHostToDevice, ctx->stream));
  }
  return GK_OK;
}

}  // extern "C"

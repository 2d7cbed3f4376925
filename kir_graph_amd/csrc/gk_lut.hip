// log10 through a value table (typing_mulit_allele.py:263  log_probs = np.log10(probs)).
//
// The read x allele probabilities take few distinct bit patterns (products of 0.999 / 0.001 in a
// handful of orders).  The device collects the distinct patterns in a hash set, the host evaluates
// numpy.log10 on them -- so the result bits are the reference's on the same machine, whatever
// libm / SVML numpy dispatches to -- and the device maps every element through the table.
#include "gk_lut.h"

namespace {

constexpr int kThreads = 256;
constexpr uint64_t kEmptyKey = kLutEmptyKey;

__device__ inline uint32_t hash64(uint64_t k) { return gk_hash64(k); }

__global__ __launch_bounds__(kThreads) void lut_collect(const uint64_t* vals, int64_t n, uint64_t* keys,
                                                        uint32_t* slot_idx, uint64_t* list, uint32_t* count,
                                                        uint32_t mask) {
  const int64_t stride = (int64_t)gridDim.x * kThreads;
  uint64_t last = kEmptyKey;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += stride) {
    const uint64_t k = vals[i];
    if (k == last) continue;   // runs of equal values are common along a column
    last = k;
    uint32_t s = hash64(k) & mask;
    for (uint32_t probe = 0; probe <= mask; ++probe) {
      const uint64_t cur = keys[s];
      if (cur == k) break;
      if (cur == kEmptyKey) {
        const unsigned long long prev = atomicCAS((unsigned long long*)&keys[s], (unsigned long long)kEmptyKey,
                                                  (unsigned long long)k);
        if (prev == kEmptyKey) {
          const uint32_t idx = atomicAdd(count, 1u);
          slot_idx[s] = idx;
          if (idx <= mask) list[idx] = k;
          break;
        }
        if (prev == k) break;
      }
      s = (s + 1) & mask;
    }
  }
}

__global__ __launch_bounds__(kThreads) void lut_apply(const uint64_t* in, double* out, int64_t n, const uint64_t* keys,
                                                      const uint32_t* slot_idx, const double* table, uint32_t mask) {
  const int64_t stride = (int64_t)gridDim.x * kThreads;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += stride) {
    const uint64_t k = in[i];
    uint32_t s = hash64(k) & mask;
    double v = __longlong_as_double(0x7FF8000000000000ll);
    for (uint32_t probe = 0; probe <= mask; ++probe) {
      const uint64_t cur = keys[s];
      if (cur == k) { v = table[slot_idx[s]]; break; }
      if (cur == kEmptyKey) break;
      s = (s + 1) & mask;
    }
    out[i] = v;
  }
}

// values [first, first + count) of the dense list have just been defined: store each next to its key
__global__ __launch_bounds__(kThreads) void lut_publish(const uint64_t* list, const double* vals, uint32_t first,
                                                        uint32_t count, const uint64_t* keys, double* slot_val,
                                                        uint64_t* slot_info, uint32_t mask) {
  const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
  if (i >= count) return;
  const uint64_t k = list[first + i];
  uint32_t s = hash64(k) & mask;
  for (uint32_t probe = 0; probe <= mask; ++probe) {
    const uint64_t cur = keys[s];
    if (cur == k) {
      const double v = vals[first + i];
      slot_val[s] = v;
      slot_info[s] = (uint64_t)(first + i) | ((uint64_t)gk_miss_of_log(v) << 32);
      return;
    }
    if (cur == kEmptyKey) return;
    s = (s + 1) & mask;
  }
}

__global__ void fill_keys(uint64_t* keys, uint64_t n) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) keys[i] = kEmptyKey;
}

}  // namespace

extern "C" {

int gk_lut_create(gk_ctx* ctx, int32_t log2_capacity, gk_lut** out) {
  gk_bind(ctx);
  GK_REQUIRE(ctx && out && log2_capacity >= 10 && log2_capacity <= 26, "bad table capacity");
  gk_lut* l = new gk_lut();
  l->ctx = ctx; l->log2cap = (uint32_t)log2_capacity;
  const size_t cap = 1ull << log2_capacity;
  GK_HIP(hipMalloc((void**)&l->d_keys, cap * sizeof(uint64_t)));
  GK_HIP(hipMalloc((void**)&l->d_slot_idx, cap * sizeof(uint32_t)));
  GK_HIP(hipMalloc((void**)&l->d_list, cap * sizeof(uint64_t)));
  GK_HIP(hipMalloc((void**)&l->d_vals, cap * sizeof(double)));
  GK_HIP(hipMalloc((void**)&l->d_slot_val, cap * sizeof(double)));
  GK_HIP(hipMalloc((void**)&l->d_slot_info, cap * sizeof(uint64_t)));
  GK_HIP(hipMemsetAsync(l->d_slot_info, 0xFF, cap * sizeof(uint64_t), ctx->stream));   // kLutNoInfo
  GK_HIP(hipMalloc((void**)&l->d_count, sizeof(uint32_t)));
  GK_HIP(hipMemsetAsync(l->d_count, 0, sizeof(uint32_t), ctx->stream));
  GK_KERNEL(fill_keys, dim3((unsigned)((cap + 255) / 256)), dim3(256), 0, ctx->stream, l->d_keys, (uint64_t)cap);
  // unwritten list entries are recognisable: a table shared by several streams may be read while a
  // kernel of another stream has claimed an index but not stored its key yet
  GK_KERNEL(fill_keys, dim3((unsigned)((cap + 255) / 256)), dim3(256), 0, ctx->stream, l->d_list, (uint64_t)cap);
  GK_KERNEL(fill_keys, dim3((unsigned)((cap + 255) / 256)), dim3(256), 0, ctx->stream, (uint64_t*)l->d_slot_val,
            (uint64_t)cap);   // "not defined yet"
  GK_HIP(hipGetLastError());
  GK_HIP(hipStreamSynchronize(ctx->stream));   // kernels of other contexts may use the table right away
  *out = l;
  return GK_OK;
}

int gk_lut_destroy(gk_lut* l) {
  gk_bind(l ? l->ctx : nullptr);
  if (!l) return GK_OK;
  hipStreamSynchronize(l->ctx->stream);
  hipFree(l->d_keys); hipFree(l->d_slot_idx); hipFree(l->d_list); hipFree(l->d_vals); hipFree(l->d_slot_val); hipFree(l->d_slot_info);
  hipFree(l->d_count);
  delete l;
  return GK_OK;
}

int gk_lut_collect(gk_lut* l, gk_dptr d_vals, int64_t n) {
  gk_bind(l ? l->ctx : nullptr);
  GK_REQUIRE(l, "null table");
  if (n <= 0) return GK_OK;
  int64_t want = (n + kThreads - 1) / kThreads;
  unsigned blocks = (unsigned)(want < 4096 ? want : 4096);
  gk_ctx* ctx = l->ctx;
  GK_PROF(ctx, "lut_collect", GK_KERNEL(lut_collect, dim3(blocks), dim3(kThreads), 0, l->ctx->stream, gk_ptr<uint64_t>(d_vals), n,
                     l->d_keys, l->d_slot_idx, l->d_list, l->d_count, (uint32_t)((1ull << l->log2cap) - 1)));
  GK_HIP(hipGetLastError());
  return GK_OK;
}

// entries claimed so far / defined so far; `drain`: when the table grew, let every stream of the device finish first, so
// that every claimed entry is stored (the resolver then never meets a claimed-but-unstored key)
static int lut_pending(gk_lut* l, bool drain, int32_t* n_total, int32_t* n_known) {
  gk_bind(l ? l->ctx : nullptr);
  GK_REQUIRE(l && n_total && n_known, "null pointer");
  uint32_t c = 0;
  GK_HIP(gk_fetch(l->ctx, &c, l->d_count, sizeof(uint32_t)));
  if (drain && (int64_t)c > (int64_t)l->n_known) {
    GK_HIP(hipDeviceSynchronize());
    GK_HIP(gk_fetch(l->ctx, &c, l->d_count, sizeof(uint32_t)));
  }
  if ((uint64_t)c * 2 > (1ull << l->log2cap)) {
    gk_set_error("probability value table overflow (%u distinct values)", c);
    return GK_ERR_CAPACITY;
  }
  *n_total = (int32_t)c;
  *n_known = l->n_known;
  return GK_OK;
}

int gk_lut_pending(gk_lut* l, int32_t* n_total, int32_t* n_known) { return lut_pending(l, true, n_total, n_known); }

int gk_lut_export(gk_lut* l, int32_t first, int32_t count, double* keys_out) {
  gk_bind(l ? l->ctx : nullptr);
  GK_REQUIRE(l && keys_out && first >= 0 && count >= 0, "bad arguments");
  if (!count) return GK_OK;
  GK_HIP(gk_fetch(l->ctx, keys_out, l->d_list + first, (size_t)count * sizeof(uint64_t)));
  return GK_OK;
}

int gk_lut_define(gk_lut* l, int32_t first, int32_t count, const double* log_vals) {
  gk_bind(l ? l->ctx : nullptr);
  GK_REQUIRE(l && log_vals && first == l->n_known && count >= 0, "values must be defined in order");
  if (!count) return GK_OK;
  GK_HIP(gk_send(l->ctx, l->d_vals + first, log_vals, (size_t)count * sizeof(double)));
  GK_KERNEL(lut_publish, dim3((unsigned)((count + kThreads - 1) / kThreads)), dim3(kThreads), 0, l->ctx->stream, l->d_list,
            l->d_vals, (uint32_t)first, (uint32_t)count, l->d_keys, l->d_slot_val, l->d_slot_info,
            (uint32_t)((1ull << l->log2cap) - 1));
  GK_HIP(hipGetLastError());
  GK_HIP(hipStreamSynchronize(l->ctx->stream));
  l->n_known = first + count;
  return GK_OK;
}

/* Evaluate log10 (through the caller's function: numpy.log10 in the Python binding, so that the bits are the
 * reference's on the machine at hand) for the values first seen since the last call, define them.  One resolver at a
 * time.  The caller's own kernels must have completed.  Kernels of other streams may still be inserting: an entry that is
 * claimed but not stored yet ends the batch (n_undefined > 0) -- whoever launched that kernel resolves it. */
static int lut_resolve(gk_lut* l, gk_log10_fn log10_fn, bool drain, int32_t* n_new_out, int32_t* n_known_out,
                       int32_t* n_undefined_out) {
  GK_REQUIRE(l && log10_fn, "null pointer");
  std::lock_guard<std::mutex> lock(l->resolve_mutex);
  int32_t tot = 0, known = 0;
  int rc = lut_pending(l, drain, &tot, &known);
  if (rc) return rc;
  int32_t fresh = tot - known;
  if (fresh > 0) {
    std::vector<double> keys((size_t)fresh), vals((size_t)fresh);
    rc = gk_lut_export(l, known, fresh, keys.data());
    if (rc) return rc;
    for (int32_t i = 0; i < fresh; ++i) {
      uint64_t bits;
      memcpy(&bits, &keys[i], 8);
      if (bits == kLutEmptyKey) { fresh = i; break; }   // claimed, not stored yet
    }
    if (fresh > 0) {
      GK_REQUIRE(log10_fn(keys.data(), fresh, vals.data()) == 0, "host log10 failed");
      rc = gk_lut_define(l, known, fresh, vals.data());
      if (rc) return rc;
    }
  } else {
    fresh = 0;
  }
  l->n_undefined = tot - l->n_known;
  if (n_new_out) *n_new_out = fresh;
  if (n_known_out) *n_known_out = l->n_known;
  if (n_undefined_out) *n_undefined_out = l->n_undefined;
  return GK_OK;
}

int gk_lut_resolve(gk_lut* l, gk_log10_fn log10_fn, int32_t* n_new_out, int32_t* n_known_out, int32_t* n_undefined_out) {
  return lut_resolve(l, log10_fn, true, n_new_out, n_known_out, n_undefined_out);
}

/* gk_lut_resolve for a caller whose OWN kernels have completed and who does not want the other streams of the device
 * drained: defines the stored entries up to the first one that a kernel still running elsewhere has claimed but not
 * stored (n_undefined counts what is left; whoever launched that kernel resolves it, or this caller asks again). */
int gk_lut_resolve_stored(gk_lut* l, gk_log10_fn log10_fn, int32_t* n_new_out, int32_t* n_known_out, int32_t* n_undefined_out) {
  return lut_resolve(l, log10_fn, false, n_new_out, n_known_out, n_undefined_out);
}

int gk_lut_known(gk_lut* l, int32_t* n_known) {
  GK_REQUIRE(l && n_known, "null pointer");
  std::lock_guard<std::mutex> lock(l->resolve_mutex);
  *n_known = l->n_known;
  return GK_OK;
}

int gk_lut_apply(gk_lut* l, gk_dptr d_in, gk_dptr d_out, int64_t n) {
  gk_bind(l ? l->ctx : nullptr);
  GK_REQUIRE(l, "null table");
  if (n <= 0) return GK_OK;
  int64_t want = (n + kThreads - 1) / kThreads;
  unsigned blocks = (unsigned)(want < 4096 ? want : 4096);
  gk_ctx* ctx = l->ctx;
  GK_PROF(ctx, "lut_apply", GK_KERNEL(lut_apply, dim3(blocks), dim3(kThreads), 0, l->ctx->stream, gk_ptr<uint64_t>(d_in),
                     gk_ptr<double>(d_out), n, l->d_keys, l->d_slot_idx, l->d_vals,
                     (uint32_t)((1ull << l->log2cap) - 1)));
  GK_HIP(hipGetLastError());
  return GK_OK;
}

}  // extern "C"

// Device value table shared by gk_lut.hip (collect / apply passes) and the compatibility kernel
// (direct lookup): probability bit pattern -> numpy.log10 of it, evaluated on the host.
#pragma once
#include "gk_common.h"

struct gk_lut {
  gk_ctx* ctx = nullptr;
  uint32_t log2cap = 0;
  uint64_t* d_keys = nullptr;      // slot -> bit pattern (kLutEmptyKey when free)
  uint32_t* d_slot_idx = nullptr;  // slot -> dense index
  uint64_t* d_list = nullptr;      // dense index -> bit pattern (insertion order)
  double* d_vals = nullptr;        // dense index -> log10 (defined for index < n_known)
  double* d_slot_val = nullptr;    // slot -> log10 once defined (kLutEmptyKey's bit pattern until then): lets the
                                   // fused lookup load key and value side by side instead of key -> index -> value
  uint64_t* d_slot_info = nullptr; // slot -> dense index (bits 0-31) | mismatch count of the value, capped at 255
                                   // (bits 32-39) once defined, kLutNoInfo until then: what the index form of the
                                   // table (u16 / u32 per entry instead of the float64) stores
  uint32_t* d_count = nullptr;     // number of dense entries
  int32_t n_known = 0;             // entries with a defined value
  int32_t n_undefined = 0;         // entries the last gk_lut_resolve saw claimed but not stored yet
  std::mutex resolve_mutex;        // one resolver at a time (host threads of a process share the table)
};

// a NaN payload no product of 0.999 / 0.001 can produce
constexpr uint64_t kLutEmptyKey = 0x7FF8DEADBEEF0001ull;
constexpr uint64_t kLutNoInfo = ~0ull;

// mismatch count of a (read, allele) entry read back from its log-likelihood: -L = 3 m + 0.000434 (n - m), so
// m = floor(-L / 3 + 1/4) for any list shorter than ~5000 ids; 255 when the count is >= 100 or L is not finite
// (the product is about to leave the normal range / underflowed: such a gene takes the exact search)
__host__ __device__ inline uint32_t gk_miss_of_log(double v) {
  const double t = v * (-1.0 / 3.0) + 0.25;
  return (t < 100.0) ? (uint32_t)(int)t : 255u;     // NaN / inf compare false
}

struct LutView {
  uint64_t* keys;
  uint32_t* slot_idx;
  uint64_t* list;
  uint32_t* count;
  const double* vals;
  const double* slot_val;
  const uint64_t* slot_info;
  uint32_t mask;
  uint32_t n_known;
};

static inline LutView gk_lut_view(const gk_lut* l) {
  return LutView{l->d_keys, l->d_slot_idx, l->d_list, l->d_count, l->d_vals, l->d_slot_val, l->d_slot_info,
                 (uint32_t)((1ull << l->log2cap) - 1), (uint32_t)l->n_known};
}

// slot of a probability's bit pattern: the low mantissa word of a product of 0.999 / 0.001 factors is as good as random
// already, so one 32-bit multiply mixes it (two 64-bit multiplies -- eight quarter-rate instructions -- per lookup were
// a measurable part of the compatibility kernel's epilogue)
__device__ inline uint32_t gk_hash64(uint64_t k) {
  const uint32_t lo = (uint32_t)k, hi = (uint32_t)(k >> 32);
  uint32_t h = lo ^ (hi << 11) ^ (hi >> 7);
  h ^= h >> 15;
  h *= 0x2C1B3C6Du;
  h ^= h >> 12;
  return h;
}

// insert `k` if absent (open addressing, first writer assigns the dense index)
__device__ inline void gk_lut_insert(const LutView& t, uint64_t k) {
  uint32_t s = gk_hash64(k) & t.mask;
  for (uint32_t probe = 0; probe <= t.mask; ++probe) {
    const uint64_t cur = t.keys[s];
    if (cur == k) return;
    if (cur == kLutEmptyKey) {
      const unsigned long long prev =
          atomicCAS((unsigned long long*)&t.keys[s], (unsigned long long)kLutEmptyKey, (unsigned long long)k);
      if (prev == kLutEmptyKey) {
        const uint32_t idx = atomicAdd(t.count, 1u);
        t.slot_idx[s] = idx;
        if (idx <= t.mask) t.list[idx] = k;
        return;
      }
      if (prev == k) return;
    }
    s = (s + 1) & t.mask;
  }
}

// value of `k` if its log is already defined; otherwise *found = false.  Key and value of a slot are
// loaded together (one memory latency per probe); a value published after this kernel's launch may be
// seen or not -- either is right, an unseen one is simply asked for again.
__device__ inline double gk_lut_lookup(const LutView& t, uint64_t k, bool* found) {
  uint32_t s = gk_hash64(k) & t.mask;
  for (uint32_t probe = 0; probe <= t.mask; ++probe) {
    const uint64_t cur = t.keys[s];
    const double v = t.slot_val[s];
    if (cur == k) {
      if ((uint64_t)__double_as_longlong(v) != kLutEmptyKey) {
        *found = true;
        return v;
      }
      break;
    }
    if (cur == kLutEmptyKey) break;
    s = (s + 1) & t.mask;
  }
  *found = false;
  return __longlong_as_double(0x7FF8000000000000ll);
}

// (dense index | mismatch count << 32) of `k` if its log is already defined; kLutNoInfo otherwise
__device__ inline uint64_t gk_lut_info(const LutView& t, uint64_t k) {
  uint32_t s = gk_hash64(k) & t.mask;
  for (uint32_t probe = 0; probe <= t.mask; ++probe) {
    const uint64_t cur = t.keys[s];
    const uint64_t info = t.slot_info[s];
    if (cur == k) return info;
    if (cur == kLutEmptyKey) break;
    s = (s + 1) & t.mask;
  }
  return kLutNoInfo;
}


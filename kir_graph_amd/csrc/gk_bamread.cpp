// Host-side ingest: BGZF / BAM -> name-collated SAM text (no GPU work, no samtools).
//
// Native counterpart of hisat2.readBam (hisat2.py:103-110), which shells out to
// `samtools sort -n bam -O SAM` and splits its stdout into lines: the file is inflated with zlib,
// the alignment records are ordered the way a query-name sort orders them (natural order of the read
// names -- digit runs compare as numbers --, READ1 before READ2, ties in input order) and rendered
// as SAM text lines, which then flow through the same packer as a `.sam` input (gk_sampack.cpp).
// Format: SAM/BAM specification v1 sections 4.1 (BGZF) and 4.2 (BAM records).
//
// samtools is not part of this image, so the rendering and the order are checked against BAM files
// written by the test suite's own encoder, not against samtools output ("parity unpinned").
#include <dlfcn.h>
#include <sys/stat.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <cctype>
#include <climits>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "gk_env.h"
#include "gk_ingest.h"
#include "graphkir_hip.h"

void gk_set_error(const char* fmt, ...);

// the inflated stream is hundreds of megabytes that the inflating threads are about to overwrite
// Blocks of hundreds of megabytes (the inflated stream, the file, the sort's records) are handed back to a pool
// when a file is closed and taken from it when the next one is opened: a fresh block of that size costs a page
// fault per 4 KiB when it is first written (measured on the GPU box: 0.6 s of system time per 2 M-read sample,
// a quarter of the ingest's CPU time), a recycled one is already mapped.  6 GB of idle
// blocks at most; a request takes the smallest idle block that holds it and is not more than twice its size.
namespace {
struct BlockPool {
  std::mutex lock;
  std::vector<std::pair<void*, size_t>> idle;
  size_t idle_bytes = 0;
  static size_t limit() { return (size_t)6 << 30; }      // idle inflate buffers kept for the next file
  void* take(size_t n, size_t& cap) {
    {
      std::lock_guard<std::mutex> g(lock);
      size_t best = idle.size();
      for (size_t i = 0; i < idle.size(); ++i)
        if (idle[i].second >= n && idle[i].second <= 2 * n + (1u << 20) && (best == idle.size() || idle[i].second < idle[best].second)) best = i;
      if (best != idle.size()) {
        void* p = idle[best].first;
        cap = idle[best].second;
        idle_bytes -= cap;
        idle.erase(idle.begin() + (ptrdiff_t)best);
        return p;
      }
    }
    cap = n + n / 16;   // room for the next file to be a little larger
    return malloc(cap ? cap : 1);
  }
  void give(void* p, size_t cap) {
    if (!p) return;
    {
      std::lock_guard<std::mutex> g(lock);
      if (cap >= (1u << 20) && idle_bytes + cap <= limit()) {
        idle.emplace_back(p, cap);
        idle_bytes += cap;
        return;
      }
    }
    free(p);
  }
};
BlockPool& block_pool() { static BlockPool* pool = new BlockPool(); return *pool; }   // never destroyed: threads may outlive main
}  // namespace

// array of trivially copyable items in a pooled block; new items are not initialised
template <typename T>
class Pooled {
 public:
  Pooled() = default;
  explicit Pooled(size_t n) { resize(n); }
  Pooled(const Pooled&) = delete;
  Pooled& operator=(const Pooled&) = delete;
  ~Pooled() { release(); }
  T* data() { return p_; }
  const T* data() const { return p_; }
  T* begin() { return p_; }
  T* end() { return p_ + size_; }
  size_t size() const { return size_; }
  bool empty() const { return size_ == 0; }
  T& operator[](size_t i) { return p_[i]; }
  const T& operator[](size_t i) const { return p_[i]; }
  void clear() { size_ = 0; }
  bool reserve(size_t n) {
    if (n <= cap_) return true;
    size_t cap_bytes = 0;
    T* q = (T*)block_pool().take(n * sizeof(T), cap_bytes);
    if (!q) return false;
    if (size_) memcpy((void*)q, (const void*)p_, size_ * sizeof(T));
    block_pool().give(p_, cap_ * sizeof(T) + slack_);
    p_ = q;
    cap_ = cap_bytes / sizeof(T);
    slack_ = cap_bytes - cap_ * sizeof(T);
    return true;
  }
  void resize(size_t n) {
    if (!reserve(n)) throw std::bad_alloc();
    size_ = n;
  }
  void append(const T* from, size_t n) {
    if (size_ + n > cap_ && !reserve(std::max(size_ + n, 2 * cap_))) throw std::bad_alloc();
    memcpy((void*)(p_ + size_), (const void*)from, n * sizeof(T));
    size_ += n;
  }
  void swap(Pooled& o) { std::swap(p_, o.p_); std::swap(size_, o.size_); std::swap(cap_, o.cap_); std::swap(slack_, o.slack_); }
  void release() {
    block_pool().give(p_, cap_ * sizeof(T) + slack_);
    p_ = nullptr;
    size_ = cap_ = slack_ = 0;
  }

 private:
  T* p_ = nullptr;
  size_t size_ = 0, cap_ = 0, slack_ = 0;
};
using Bytes = Pooled<uint8_t>;

struct gk_bam {
  Bytes data;                           // inflated BAM stream
  std::string header;                   // SAM header text ('@' lines)
  std::vector<std::string> ref_names;
  struct Rec { uint64_t off; uint32_t size; };
  Pooled<Rec> recs;                     // in output order after sorting
  bool name_sorted = false;             // recs are in query-name order
  bool collated = false;                // ... and so are the records in `data`, back to back (no header in front)
  size_t next = 0;                      // next record to render
  std::string staged;                   // rendered lines not handed out yet
  size_t staged_off = 0;
};

namespace {

inline uint32_t rd32(const uint8_t* p) { return (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24; }
inline int32_t rds32(const uint8_t* p) { return (int32_t)rd32(p); }
inline uint16_t rd16(const uint8_t* p) { return (uint16_t)(p[0] | p[1] << 8); }

// Every gzip member of the BGZF file, inflated back to back.
bool inflate_members(const Bytes& in, Bytes& out) {
  z_stream zs;
  memset(&zs, 0, sizeof(zs));
  if (inflateInit2(&zs, 15 + 16) != Z_OK) return false;
  out.clear();
  if (!out.reserve(in.size() * 4)) return false;
  std::vector<uint8_t> buf(1 << 20);
  zs.next_in = const_cast<Bytef*>(in.data());
  zs.avail_in = 0;
  size_t consumed = 0;
  bool ok = true;
  while (consumed < in.size() || zs.avail_in) {
    if (zs.avail_in == 0) {
      const size_t take = std::min<size_t>(in.size() - consumed, 1u << 30);
      zs.next_in = const_cast<Bytef*>(in.data() + consumed);
      zs.avail_in = (uInt)take;
      consumed += take;
    }
    zs.next_out = buf.data();
    zs.avail_out = (uInt)buf.size();
    const int rc = inflate(&zs, Z_NO_FLUSH);
    out.append(buf.data(), buf.size() - zs.avail_out);
    if (rc == Z_STREAM_END) {
      if (zs.avail_in == 0 && consumed >= in.size()) break;
      if (inflateReset(&zs) != Z_OK) { ok = false; break; }   // next member
    } else if (rc != Z_OK) {
      ok = false;
      break;
    }
  }
  inflateEnd(&zs);
  return ok;
}

// libdeflate inflates BGZF blocks 2-3 times faster than zlib.  The image ships its runtime library without
// the header, so the three entry points used are bound by name at first use (signatures of libdeflate.h,
// stable since 1.0); without the library, or under the test hook no_libdeflate (GK_TEST_HOOKS), zlib does the work.
struct FastInflate {
  void* (*alloc)() = nullptr;
  int (*run)(void*, const void*, size_t, void*, size_t, size_t*) = nullptr;   // 0 = LIBDEFLATE_SUCCESS
  void (*release)(void*) = nullptr;
};

const FastInflate* fast_inflate() {
  static const FastInflate* found = []() -> const FastInflate* {
    if (gk_test_hook("no_libdeflate")) return nullptr;
    void* lib = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
    if (!lib) return nullptr;
    static FastInflate f;
    f.alloc = (void* (*)())dlsym(lib, "libdeflate_alloc_decompressor");
    f.run = (int (*)(void*, const void*, size_t, void*, size_t, size_t*))dlsym(lib, "libdeflate_deflate_decompress");
    f.release = (void (*)(void*))dlsym(lib, "libdeflate_free_decompressor");
    return (f.alloc && f.run && f.release) ? &f : nullptr;
  }();
  return found;
}

// BGZF members carry their own compressed size (extra field 'BC') and end with the uncompressed size,
// so the file splits into independent blocks that inflate in parallel, each straight into its place.
// Returns false when the stream is not made of such blocks (the caller then inflates sequentially).
bool inflate_bgzf_parallel(const Bytes& in, Bytes& out, int n_threads) {
  struct Block { size_t in_off, in_len, out_off, out_len; };
  std::vector<Block> blocks;
  size_t o = 0, total = 0;
  while (o < in.size()) {
    if (o + 18 > in.size()) return false;
    const uint8_t* h = in.data() + o;
    if (h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || !(h[3] & 4)) return false;
    const size_t xlen = rd16(h + 10);
    if (o + 12 + xlen > in.size()) return false;
    size_t bsize = 0;
    for (size_t x = 0; x + 4 <= xlen;) {
      const uint8_t* sub = h + 12 + x;
      const size_t slen = rd16(sub + 2);
      if (sub[0] == 'B' && sub[1] == 'C' && slen == 2 && x + 6 <= xlen) bsize = (size_t)rd16(sub + 4) + 1;
      x += 4 + slen;
    }
    if (!bsize || bsize < 12 + xlen + 8 || o + bsize > in.size()) return false;
    const size_t isize = rd32(h + bsize - 4);
    blocks.push_back({o + 12 + xlen, bsize - 12 - xlen - 8, total, isize});
    total += isize;
    o += bsize;
  }
  out.resize(total);
  std::vector<char> bad((size_t)std::max(n_threads, 1), 0);
  const FastInflate* fast = fast_inflate();
  auto work = [&](int t, size_t a, size_t b) {
    if (fast) {
      void* d = fast->alloc();
      if (!d) { bad[(size_t)t] = 1; return; }
      for (size_t i = a; i < b; ++i) {
        const Block& bl = blocks[i];
        if (!bl.out_len) continue;
        size_t made = 0;
        if (fast->run(d, in.data() + bl.in_off, bl.in_len, out.data() + bl.out_off, bl.out_len, &made) != 0 || made != bl.out_len) {
          bad[(size_t)t] = 1;
          break;
        }
      }
      fast->release(d);
      return;
    }
    z_stream zs;
    memset(&zs, 0, sizeof(zs));
    if (inflateInit2(&zs, -15) != Z_OK) { bad[(size_t)t] = 1; return; }
    for (size_t i = a; i < b; ++i) {
      const Block& bl = blocks[i];
      if (!bl.out_len) continue;
      inflateReset(&zs);
      zs.next_in = const_cast<Bytef*>(in.data() + bl.in_off);
      zs.avail_in = (uInt)bl.in_len;
      zs.next_out = out.data() + bl.out_off;
      zs.avail_out = (uInt)bl.out_len;
      if (inflate(&zs, Z_FINISH) != Z_STREAM_END || zs.avail_out != 0) { bad[(size_t)t] = 1; break; }
    }
    inflateEnd(&zs);
  };
  const size_t n = blocks.size();
  n_threads = (int)std::min<size_t>((size_t)std::max(n_threads, 1), std::max<size_t>(n / 8, 1));
  if (n_threads <= 1) {
    work(0, 0, n);
  } else {
    std::vector<std::thread> pool;
    for (int t = 0; t < n_threads; ++t) pool.emplace_back(work, t, n * t / n_threads, n * (t + 1) / n_threads);
    for (auto& th : pool) th.join();
  }
  for (char b : bad) if (b) return false;
  return true;
}

int ingest_threads() { return gk_ingest_threads(); }

// Query-name order: characters compare by code, except that where both names have a digit the two
// digit runs compare as numbers (leading zeros skipped, more digits = larger); equal numbers written
// with a different count of leading zeros order the shorter spelling last.
int name_order(const char* a0, const char* b0) {
  const unsigned char *a = (const unsigned char*)a0, *b = (const unsigned char*)b0;
  const unsigned char *pa = a, *pb = b;
  while (*pa && *pb) {
    if (isdigit(*pa) && isdigit(*pb)) {
      while (*pa == '0') ++pa;
      while (*pb == '0') ++pb;
      const unsigned char *ea = pa, *eb = pb;
      while (isdigit(*ea)) ++ea;
      while (isdigit(*eb)) ++eb;
      const long la = ea - pa, lb = eb - pb;
      if (la != lb) return la > lb ? 1 : -1;
      for (long i = 0; i < la; ++i)
        if (pa[i] != pb[i]) return (int)pa[i] - (int)pb[i];
      pa = ea;
      pb = eb;
      if (pa - a != pb - b) return (pa - a) < (pb - b) ? 1 : -1;
    } else {
      if (*pa != *pb) return (int)*pa - (int)*pb;
      ++pa;
      ++pb;
    }
  }
  return *pa ? 1 : *pb ? -1 : 0;
}

// Sort key of a query name: a byte string whose memcmp order is name_order's.  Characters stand for
// themselves; a digit run becomes '0' + its count of significant digits, those digits, then 255 - its count
// of leading zeros (more zeros order first); the name ends with a 0 byte.  Only the first kNameKey bytes
// are kept: keys that agree over the shorter of the two kept lengths decide nothing (the names themselves
// are compared then), unless both are whole -- then the names are equal.
constexpr int kNameKey = 16;
struct SortRec {
  uint64_t k0, k1;   // the key bytes as two big-endian numbers: their order is the bytes' order
  uint64_t off;
  uint32_t size;
  uint8_t kept;      // key bytes that are meaningful
  uint8_t whole;     // the key covers the whole name (terminator included)
  uint8_t mate;      // FLAG & 0xC0: READ1 before READ2 among equal names
};

void name_key(const char* name, SortRec& r) {
  const unsigned char* p = (const unsigned char*)name;
  uint8_t key[kNameKey];
  int n = 0;
  bool room = true;
  auto put = [&](unsigned v) { if (n < kNameKey) key[n++] = (uint8_t)v; else room = false; };
  while (*p && room) {
    if (isdigit(*p)) {
      const unsigned char* z = p;
      while (*p == '0') ++p;
      const unsigned zeros = (unsigned)(p - z);
      const unsigned char* e = p;
      while (isdigit(*e)) ++e;
      if (e - p > 9 || zeros > 200) { room = false; break; }   // not expressible: the key stops before the run
      put('0' + (unsigned)(e - p));
      for (; p < e; ++p) put(*p);
      put(255u - zeros);
    } else {
      put(*p++);
    }
  }
  if (room) put(0);
  r.kept = (uint8_t)n;
  r.whole = room ? 1 : 0;
  for (int i = n; i < kNameKey; ++i) key[i] = 0;
  uint64_t a, b;
  memcpy(&a, key, 8);
  memcpy(&b, key + 8, 8);
  r.k0 = __builtin_bswap64(a);
  r.k1 = __builtin_bswap64(b);
}

void append_int(std::string& s, long long v) {
  char tmp[24];
  char* e = tmp + sizeof(tmp);
  char* p = e;
  unsigned long long u = v < 0 ? 0ull - (unsigned long long)v : (unsigned long long)v;
  do { *--p = (char)('0' + u % 10); u /= 10; } while (u);
  if (v < 0) *--p = '-';
  s.append(p, (size_t)(e - p));
}

// one optional field "TAG:TYPE:VALUE"; returns the bytes consumed or 0 on a malformed field
size_t render_tag(const uint8_t* p, size_t left, std::string& s) {
  if (left < 3) return 0;
  s.push_back((char)p[0]); s.push_back((char)p[1]); s.push_back(':');
  const char type = (char)p[2];
  const uint8_t* v = p + 3;
  left -= 3;
  auto need = [&](size_t n) { return left >= n; };
  switch (type) {
    case 'A': if (!need(1)) return 0; s += "A:"; s.push_back((char)v[0]); return 4;
    case 'c': if (!need(1)) return 0; s += "i:"; append_int(s, (int8_t)v[0]); return 4;
    case 'C': if (!need(1)) return 0; s += "i:"; append_int(s, v[0]); return 4;
    case 's': if (!need(2)) return 0; s += "i:"; append_int(s, (int16_t)rd16(v)); return 5;
    case 'S': if (!need(2)) return 0; s += "i:"; append_int(s, rd16(v)); return 5;
    case 'i': if (!need(4)) return 0; s += "i:"; append_int(s, rds32(v)); return 7;
    case 'I': if (!need(4)) return 0; s += "i:"; append_int(s, rd32(v)); return 7;
    case 'f': {
      if (!need(4)) return 0;
      float f; const uint32_t u = rd32(v); memcpy(&f, &u, 4);
      char tmp[32]; const int n = snprintf(tmp, sizeof(tmp), "%g", f);
      s += "f:"; s.append(tmp, (size_t)n);
      return 7;
    }
    case 'Z': case 'H': {
      const void* z = memchr(v, 0, left);
      if (!z) return 0;
      const size_t n = (const uint8_t*)z - v;
      s.push_back(type); s.push_back(':'); s.append((const char*)v, n);
      return 3 + n + 1;
    }
    case 'B': {
      if (!need(5)) return 0;
      const char sub = (char)v[0];
      const uint32_t count = rd32(v + 1);
      const size_t w = (sub == 'c' || sub == 'C') ? 1 : (sub == 's' || sub == 'S') ? 2 : (sub == 'i' || sub == 'I' || sub == 'f') ? 4 : 0;
      if (!w || !need(5 + (size_t)count * w)) return 0;
      s += "B:"; s.push_back(sub);
      const uint8_t* e = v + 5;
      for (uint32_t i = 0; i < count; ++i, e += w) {
        s.push_back(',');
        switch (sub) {
          case 'c': append_int(s, (int8_t)e[0]); break;
          case 'C': append_int(s, e[0]); break;
          case 's': append_int(s, (int16_t)rd16(e)); break;
          case 'S': append_int(s, rd16(e)); break;
          case 'i': append_int(s, rds32(e)); break;
          case 'I': append_int(s, rd32(e)); break;
          default: {
            float f; const uint32_t u = rd32(e); memcpy(&f, &u, 4);
            char tmp[32]; const int n = snprintf(tmp, sizeof(tmp), "%g", f);
            s.append(tmp, (size_t)n);
          }
        }
      }
      return 3 + 5 + (size_t)count * w;
    }
    default: return 0;
  }
}

// SAM text of one alignment record (without the trailing newline)
bool render(const gk_bam& b, const gk_bam::Rec& r, std::string& s) {
  const uint8_t* p = b.data.data() + r.off;   // first byte after block_size
  if (r.size < 32) return false;
  const int32_t ref_id = rds32(p), pos = rds32(p + 4);
  const uint32_t l_name = p[8], mapq = p[9], n_cig = rd16(p + 12), flag = rd16(p + 14);
  const uint32_t l_seq = rd32(p + 16);
  const int32_t next_ref = rds32(p + 20), next_pos = rds32(p + 24), tlen = rds32(p + 28);
  const size_t fixed = 32, need = fixed + l_name + 4ull * n_cig + (l_seq + 1) / 2 + l_seq;
  if (need > r.size || l_name == 0) return false;
  const uint8_t* name = p + fixed;
  const uint8_t* cig = name + l_name;
  const uint8_t* seq = cig + 4ull * n_cig;
  const uint8_t* qual = seq + (l_seq + 1) / 2;
  const uint8_t* tags = qual + l_seq;
  const uint8_t* end = p + r.size;
  auto ref_name = [&](int32_t id) -> const char* {
    return (id >= 0 && (size_t)id < b.ref_names.size()) ? b.ref_names[(size_t)id].c_str() : "*";
  };
  s.clear();
  s.append((const char*)name, strnlen((const char*)name, l_name));
  s.push_back('\t'); append_int(s, flag);
  s.push_back('\t'); s += ref_name(ref_id);
  s.push_back('\t'); append_int(s, (long long)pos + 1);
  s.push_back('\t'); append_int(s, mapq);
  s.push_back('\t');
  if (n_cig == 0) {
    s.push_back('*');
  } else {
    for (uint32_t i = 0; i < n_cig; ++i) {
      const uint32_t c = rd32(cig + 4ull * i), op = c & 15u;
      append_int(s, c >> 4);
      s.push_back(op < 9 ? "MIDNSHP=X"[op] : '?');
    }
  }
  s.push_back('\t');
  if (next_ref < 0) s.push_back('*');
  else if (next_ref == ref_id) s.push_back('=');
  else s += ref_name(next_ref);
  s.push_back('\t'); append_int(s, (long long)next_pos + 1);
  s.push_back('\t'); append_int(s, tlen);
  s.push_back('\t');
  if (l_seq == 0) {
    s.push_back('*');
  } else {
    static const char kBase[] = "=ACMGRSVTWYHKDBN";
    const size_t at = s.size();
    s.resize(at + l_seq);
    char* d = &s[at];
    for (uint32_t i = 0; i + 1 < l_seq; i += 2) {
      const uint8_t b2 = seq[i >> 1];
      d[i] = kBase[b2 >> 4];
      d[i + 1] = kBase[b2 & 15u];
    }
    if (l_seq & 1u) d[l_seq - 1] = kBase[seq[(l_seq - 1) >> 1] >> 4];
  }
  s.push_back('\t');
  if (l_seq == 0 || qual[0] == 0xFF) {
    s.push_back('*');
  } else {
    const size_t at = s.size();
    s.resize(at + l_seq);
    char* d = &s[at];
    for (uint32_t i = 0; i < l_seq; ++i) d[i] = (char)(qual[i] + 33);
  }
  while (tags < end) {
    s.push_back('\t');
    const size_t used = render_tag(tags, (size_t)(end - tags), s);
    if (!used) return false;
    tags += used;
  }
  return true;
}

}  // namespace

extern "C" {

static int bam_open_impl(const char* path, int32_t name_sorted, gk_bam** out) {
  if (!path || !out) { gk_set_error("null argument"); return GK_ERR_ARG; }
  FILE* f = fopen(path, "rb");
  if (!f) { gk_set_error("cannot open %s", path); return GK_ERR_ARG; }
  GkPhaseClock clock("bam_open");
  Bytes raw;
  {
    struct stat st;
    size_t have = 0;
    raw.resize(fstat(fileno(f), &st) == 0 && st.st_size > 0 ? (size_t)st.st_size : (size_t)1 << 22);
    for (;;) {   // whole file, whatever fstat said
      if (have == raw.size()) {          // the usual end: the block fstat sized is full -- look before growing (and copying) it
        const int c = fgetc(f);
        if (c == EOF) break;
        raw.resize(raw.size() * 2);
        raw[have++] = (uint8_t)c;
      }
      const size_t n = fread(raw.data() + have, 1, raw.size() - have, f);
      if (!n) break;
      have += n;
    }
    raw.resize(have);
    fclose(f);
  }
  clock.lap("read");
  gk_bam* b = new gk_bam();
  if (!inflate_bgzf_parallel(raw, b->data, ingest_threads()) && !inflate_members(raw, b->data)) {
    delete b;
    gk_set_error("%s is not a BGZF / gzip stream or is truncated", path);
    return GK_ERR_ARG;
  }
  raw.release();
  clock.lap("inflate");
  const Bytes& d = b->data;
  auto bad = [&](const char* what) {
    gk_set_error("%s: malformed BAM (%s)", path, what);
    delete b;
    return GK_ERR_ARG;
  };
  if (d.size() < 12 || memcmp(d.data(), "BAM\1", 4) != 0) return bad("magic");
  size_t o = 4;
  const uint32_t l_text = rd32(d.data() + o); o += 4;
  if (o + l_text + 4 > d.size()) return bad("header text");
  b->header.assign((const char*)d.data() + o, strnlen((const char*)d.data() + o, l_text));
  o += l_text;
  const uint32_t n_ref = rd32(d.data() + o); o += 4;
  for (uint32_t i = 0; i < n_ref; ++i) {
    if (o + 4 > d.size()) return bad("reference list");
    const uint32_t l_name = rd32(d.data() + o); o += 4;
    if (o + l_name + 4 > d.size() || l_name == 0) return bad("reference name");
    b->ref_names.emplace_back((const char*)d.data() + o, strnlen((const char*)d.data() + o, l_name));
    o += l_name + 4;   // name, l_ref
  }
  // Record index.  A record starts where the one before it ends, so the walk over the stream is a pointer chase;
  // it is cut into segments that are chased at once.  Inside a segment the start of the first record is not
  // known: a thread picks the first offset that looks like a record and whose chain of block sizes stays valid,
  // and chases from there.  Its chain is only taken over once the true chain -- chased record by record from the
  // previous segment's end -- lands on one of its offsets: from that offset on the two chains are the same chain.
  // A guess that never meets the true chain costs time (that segment is chased serially), never correctness.
  const size_t n_bytes = d.size();
  const uint8_t* const base0 = d.data();
  const int64_t n_ref_i = (int64_t)n_ref;
  auto valid_at = [&](size_t s, uint32_t& size) {   // s: offset of a block_size field
    if (s + 4 > n_bytes) return false;
    size = rd32(base0 + s);
    if (size < 32 || s + 4 + (size_t)size > n_bytes) return false;
    // every walker below (text rendering, binary packing, pileup, name collation) trusts these fields: the
    // variable-length parts must lie inside the block and the query name must end in NUL inside its field
    const uint8_t* r = base0 + s + 4;
    const uint64_t l_name = r[8], n_cig = rd16(r + 12), l_seq = rd32(r + 16);
    const uint64_t need = 32 + l_name + 4 * n_cig + (l_seq + 1) / 2 + l_seq;
    return l_name >= 1 && need <= size && r[32 + l_name - 1] == 0;
  };
  auto plausible = [&](size_t s) {                  // stricter: only used to pick where a guessed chain starts
    uint32_t size;
    if (!valid_at(s, size) || size > (1u << 24)) return false;
    const uint8_t* r = base0 + s + 4;
    const int64_t ref_id = rds32(r), pos = rds32(r + 4), next_ref = rds32(r + 20), next_pos = rds32(r + 24);
    if (ref_id < -1 || ref_id >= n_ref_i || next_ref < -1 || next_ref >= n_ref_i || pos < -1 || next_pos < -1) return false;
    const uint32_t l_name = r[8];
    for (uint32_t i = 0; i + 1 < l_name; ++i)
      if (r[32 + i] < '!' || r[32 + i] > '~') return false;
    return true;
  };
  struct Segment {
    size_t lo = 0, hi = 0;
    std::vector<gk_bam::Rec> chain;   // records chased from the guessed start, in order
    size_t chain_end = 0;             // offset after the last of them (where the chain stopped or left the segment)
    std::vector<gk_bam::Rec> serial;  // records of the true chain before it met the guessed one (usually none)
    size_t take_from = 0, take_n = 0; // the part of `chain` from where the two met
    size_t out_at = 0;
  };
  const size_t span = n_bytes - o;
  size_t n_seg = std::min<size_t>((size_t)ingest_threads() * 4, std::max<size_t>(span >> 20, 1));
  n_seg = std::max<size_t>(1, (size_t)gk_test_hook_value("bam_segments", (long)n_seg));   // tests: many small segments
  std::vector<Segment> seg(n_seg);
  for (size_t t = 0; t < n_seg; ++t) { seg[t].lo = o + span * t / n_seg; seg[t].hi = o + span * (t + 1) / n_seg; }
  auto chase = [&](Segment& sg, bool known_start) {
    size_t c = sg.lo;
    const size_t give_up = std::min(sg.hi, sg.lo + ((size_t)1 << 20));
    while (c < give_up) {
      if (!known_start && !plausible(c)) { ++c; continue; }
      sg.chain.clear();
      size_t s = c;
      bool broke = false;
      while (s < sg.hi) {
        __builtin_prefetch(base0 + s + 768);
        __builtin_prefetch(base0 + s + 1536);
        uint32_t size;
        if (!valid_at(s, size)) { broke = true; break; }
        sg.chain.push_back({(uint64_t)s + 4, size});
        s += 4 + (size_t)size;
      }
      sg.chain_end = s;
      // a chain that breaks early was a wrong guess (or a damaged file: the true chain will say so); one that
      // runs for a while is kept even if it breaks later, the merge takes it no further than the break
      if (known_start || !broke || sg.chain.size() >= 64) return;
      ++c;
    }
    sg.chain.clear();
    sg.chain_end = sg.lo;
  };
  {
    std::atomic<size_t> next_seg{0};
    auto run = [&] {
      for (size_t t; (t = next_seg.fetch_add(1)) < n_seg;) {
        seg[t].chain.reserve((seg[t].hi - seg[t].lo) / 256 + 16);
        chase(seg[t], t == 0);
      }
    };
    const int n_thr = (int)std::min<size_t>((size_t)ingest_threads(), n_seg);
    if (n_thr <= 1) {
      run();
    } else {
      std::vector<std::thread> pool;
      for (int t = 0; t < n_thr; ++t) pool.emplace_back(run);
      for (auto& th : pool) th.join();
    }
  }
  size_t n_rec = 0;
  for (size_t t = 0; t < n_seg; ++t) {   // the true chain, segment by segment
    Segment& sg = seg[t];
    auto it = sg.chain.begin();
    while (o < sg.hi && o + 4 <= n_bytes) {
      it = std::lower_bound(it, sg.chain.end(), (uint64_t)o + 4, [](const gk_bam::Rec& r, uint64_t v) { return r.off < v; });
      if (it != sg.chain.end() && it->off == (uint64_t)o + 4) {   // met: the rest of the guessed chain is the true one
        sg.take_from = (size_t)(it - sg.chain.begin());
        sg.take_n = sg.chain.size() - sg.take_from;
        o = sg.chain_end;
        if (o < sg.hi) return bad("alignment block");             // the chain stopped inside the segment: at a bad record
        break;
      }
      uint32_t size;
      if (!valid_at(o, size)) return bad("alignment block");
      sg.serial.push_back({(uint64_t)o + 4, size});               // not met yet: one record of the true chain at a time
      o += 4 + (size_t)size;
    }
    sg.out_at = n_rec;
    n_rec += sg.serial.size() + sg.take_n;
  }
  b->recs.resize(n_rec);
  {
    std::atomic<size_t> next_seg{0};
    auto run = [&] {
      for (size_t t; (t = next_seg.fetch_add(1)) < n_seg;) {
        const Segment& sg = seg[t];
        std::copy(sg.serial.begin(), sg.serial.end(), b->recs.begin() + (ptrdiff_t)sg.out_at);
        std::copy(sg.chain.begin() + (ptrdiff_t)sg.take_from, sg.chain.begin() + (ptrdiff_t)(sg.take_from + sg.take_n),
                  b->recs.begin() + (ptrdiff_t)(sg.out_at + sg.serial.size()));
      }
    };
    const int n_thr = (int)std::min<size_t>((size_t)ingest_threads(), n_seg);
    std::vector<std::thread> pool;
    for (int t = 1; t < n_thr; ++t) pool.emplace_back(run);
    run();
    for (auto& th : pool) th.join();
  }
  if (o != d.size()) return bad("trailing bytes");
  b->name_sorted = name_sorted != 0;
  clock.lap("index");
  if (name_sorted) {
    const uint8_t* base = d.data();
    // A strict total order (name, then READ1 before READ2, then the place in the file), so that any sort gives the
    // order a stable sort by name and mate gives.  The keys usually decide without touching the records.
    // Read names of one run share a long prefix (instrument, run, flow cell, lane): it decides nothing and would
    // fill the 16-byte keys, sending every comparison to the records.  The keys (and the full comparison) start
    // after the prefix common to ALL names, cut back so that it does not end inside or right after a digit run
    // (a run compares as a number: it must be seen whole).
    size_t lcp = 0;
    if (!b->recs.empty()) {
      const char* first = (const char*)base + b->recs[0].off + 32;
      const size_t n_all = b->recs.size();
      const int n_lcp_thr = (int)std::max<size_t>(1, std::min<size_t>((size_t)ingest_threads(), n_all / 4096));
      std::vector<size_t> part_lcp((size_t)n_lcp_thr, strlen(first));
      std::vector<std::thread> pool;
      auto scan = [&](int t) {
        size_t keep = part_lcp[(size_t)t];
        for (size_t k = n_all * (size_t)t / (size_t)n_lcp_thr; k < n_all * (size_t)(t + 1) / (size_t)n_lcp_thr && keep; ++k) {
          const char* name = (const char*)base + b->recs[k].off + 32;
          size_t i = 0;
          while (i < keep && name[i] == first[i]) ++i;   // stops at the name's NUL at the latest: first[i] != 0 for i < keep
          keep = i;
        }
        part_lcp[(size_t)t] = keep;
      };
      for (int t = 1; t < n_lcp_thr; ++t) pool.emplace_back(scan, t);
      scan(0);
      for (auto& th : pool) th.join();
      lcp = *std::min_element(part_lcp.begin(), part_lcp.end());
      while (lcp > 0 && isdigit((unsigned char)first[lcp - 1])) --lcp;
    }
    auto before = [base, lcp](const SortRec& x, const SortRec& y) {
      if (x.whole & y.whole) {          // unused key bytes are zero and no name byte is: whole keys compare as they are
        if (x.k0 != y.k0) return x.k0 < y.k0;
        if (x.k1 != y.k1) return x.k1 < y.k1;
      } else {                          // the first min(kept) bytes only
        const unsigned m = std::min(x.kept, y.kept);
        const unsigned s0 = m >= 8 ? 0 : 8 * (8 - m), s1 = m >= 16 ? 0 : 8 * (16 - m);
        const uint64_t a0 = m ? x.k0 >> s0 : 0, b0 = m ? y.k0 >> s0 : 0;
        if (a0 != b0) return a0 < b0;
        const uint64_t a1 = m > 8 ? x.k1 >> s1 : 0, b1 = m > 8 ? y.k1 >> s1 : 0;
        if (a1 != b1) return a1 < b1;
        const int t = name_order((const char*)base + x.off + 32 + lcp, (const char*)base + y.off + 32 + lcp);
        if (t) return t < 0;
      }
      if (x.mate != y.mate) return x.mate < y.mate;   // READ1 (0x40) before READ2 (0x80)
      return x.off < y.off;
    };
    // Sample sort over the ingest threads: keyed records, splitters from a sample, one scatter into buckets whose
    // key ranges follow one another, then every bucket sorted on its own (a bucket fits a core's cache).
    const size_t n = b->recs.size();
    Pooled<SortRec> recs(n), sorted(n);
    const int n_thr = (int)std::max<size_t>(1, std::min<size_t>((size_t)ingest_threads(), n / 4096));
    auto part = [&](int t) { return n * (size_t)t / (size_t)n_thr; };
    auto on_all = [&](const std::function<void(int)>& work) {
      if (n_thr == 1) { work(0); return; }
      std::vector<std::thread> pool;
      for (int t = 1; t < n_thr; ++t) pool.emplace_back(work, t);
      work(0);
      for (auto& th : pool) th.join();
    };
    on_all([&](int t) {
      for (size_t k = part(t); k < part(t + 1); ++k) {
        SortRec& r = recs[k];
        r.off = b->recs[k].off;
        r.size = b->recs[k].size;
        r.mate = (uint8_t)(rd16(base + r.off + 14) & 0xC0u);
        name_key((const char*)base + r.off + 32 + lcp, r);
      }
    });
    const size_t n_bucket = n < 16384 ? 1 : std::min<size_t>(256, std::max<size_t>(4, n / 16384));
    if (n_bucket == 1) {
      std::sort(recs.begin(), recs.end(), before);
      sorted.swap(recs);
    } else {
      std::vector<SortRec> sample;
      const size_t n_sample = n_bucket * 32;
      for (size_t i = 0; i < n_sample; ++i) sample.push_back(recs[(n - 1) * i / (n_sample - 1)]);
      std::sort(sample.begin(), sample.end(), before);
      std::vector<SortRec> splitter;     // bucket q holds the records x with splitter[q-1] <= x < splitter[q]
      for (size_t q = 1; q < n_bucket; ++q) splitter.push_back(sample[q * n_sample / n_bucket]);
      std::vector<uint8_t> bucket_of(n);
      std::vector<std::vector<size_t>> count((size_t)n_thr, std::vector<size_t>(n_bucket, 0));
      on_all([&](int t) {
        std::vector<size_t>& c = count[(size_t)t];
        for (size_t k = part(t); k < part(t + 1); ++k) {
          const size_t q = (size_t)(std::upper_bound(splitter.begin(), splitter.end(), recs[k], before) - splitter.begin());
          bucket_of[k] = (uint8_t)q;
          ++c[q];
        }
      });
      std::vector<size_t> bucket_at(n_bucket + 1, 0);
      for (size_t q = 0; q < n_bucket; ++q) {
        size_t at = bucket_at[q];
        for (int t = 0; t < n_thr; ++t) { const size_t c = count[(size_t)t][q]; count[(size_t)t][q] = at; at += c; }
        bucket_at[q + 1] = at;
      }
      on_all([&](int t) {
        std::vector<size_t>& at = count[(size_t)t];
        for (size_t k = part(t); k < part(t + 1); ++k) sorted[at[bucket_of[k]]++] = recs[k];
      });
      // A bucket: byte-wise radix passes over the first eight key bytes (constant bytes skipped; `recs`, free since the
      // scatter, is the other buffer), then the runs of equal first halves -- the two mates of a name at least -- by the
      // full comparison.  Valid when every key that does not cover its whole name keeps at least eight bytes: two keys
      // that differ in their first halves are then ordered by them (a whole key differs from any other key within its
      // own length).  Otherwise, and for small buckets, the comparison sort.
      auto sort_bucket = [&](SortRec* a, SortRec* tmp, size_t len) {
        bool radix = len >= 512;
        for (size_t i = 0; i < len && radix; ++i) radix = a[i].whole || a[i].kept >= 8;
        if (!radix) { std::sort(a, a + len, before); return; }
        SortRec *src = a, *dst = tmp;
        for (int byte = 0; byte < 8; ++byte) {
          const int sh = 8 * byte;
          size_t hist[256] = {0};
          for (size_t i = 0; i < len; ++i) ++hist[(src[i].k0 >> sh) & 255u];
          bool constant = false;
          size_t at = 0;
          for (int v = 0; v < 256; ++v) { const size_t c = hist[v]; constant = constant || c == len; hist[v] = at; at += c; }
          if (constant) continue;
          for (size_t i = 0; i < len; ++i) dst[hist[(src[i].k0 >> sh) & 255u]++] = src[i];
          std::swap(src, dst);
        }
        if (src != a) memcpy((void*)a, (const void*)src, len * sizeof(SortRec));
        for (size_t i = 0; i < len;) {
          size_t j = i + 1;
          while (j < len && a[j].k0 == a[i].k0) ++j;
          if (j - i > 1) std::sort(a + i, a + j, before);
          i = j;
        }
      };
      std::atomic<size_t> next_bucket{0};
      on_all([&](int) {
        for (size_t q; (q = next_bucket.fetch_add(1)) < n_bucket;)
          sort_bucket(sorted.data() + bucket_at[q], recs.data() + bucket_at[q], bucket_at[q + 1] - bucket_at[q]);
      });
    }
    on_all([&](int t) {
      for (size_t k = part(t); k < part(t + 1); ++k) b->recs[k] = {sorted[k].off, sorted[k].size};
    });
    clock.lap("name sort");
    // (Not done: gathering the records themselves into name order, back to back -- 9 % less CPU for the ingest on a small
    // host, 7 % more on the GPU boxes, where the second 300 MB block costs first-touch page faults: round 3.)
    constexpr bool collate = false;
    if (collate && n > 0) {
      std::vector<uint64_t> first((size_t)n_thr + 1, 0);      // bytes of the records before part(t)
      on_all([&](int t) {
        uint64_t sum = 0;
        for (size_t k = part(t); k < part(t + 1); ++k) sum += b->recs[k].size;
        first[(size_t)t + 1] = sum;
      });
      for (int t = 0; t < n_thr; ++t) first[(size_t)t + 1] += first[(size_t)t];
      Bytes ordered;
      ordered.resize((size_t)first[(size_t)n_thr]);
      uint8_t* const dst = ordered.data();
      on_all([&](int t) {
        uint64_t at = first[(size_t)t];
        const size_t e = part(t + 1);
        for (size_t k = part(t); k < e; ++k) {
          if (k + 24 < e) {               // the source of a copy two dozen records ahead: head, middle and tail lines
            const uint8_t* q = base + b->recs[k + 24].off;
            const uint32_t sz = b->recs[k + 24].size;
            for (uint32_t o = 0; o < sz; o += 64) __builtin_prefetch(q + o);
          }
          const gk_bam::Rec r = b->recs[k];
          memcpy(dst + at, base + r.off, r.size);
          b->recs[k] = {at, r.size};
          at += r.size;
        }
      });
      b->data.swap(ordered);            // the inflated stream goes back to the pool with `ordered`
      b->collated = true;
      clock.lap("collate");
    }
  } else {
    clock.lap("name sort");
  }
  *out = b;
  return GK_OK;
}

// The records, in output order, straight into a packer: same pairing, checks and records as feeding
// the rendered text to gk_packer_feed, without rendering or re-parsing it.  Only what the packer reads
// is converted (CIGAR and SEQ to text; NM / MD / Zs / NH from the optional fields).
static int bam_pack_impl(gk_bam* b, gk_packer* pk) {
  if (!b || !pk) { gk_set_error("null handle"); return GK_ERR_ARG; }
  const uint8_t* base = b->data.data();
  auto ref_name = [&](int32_t id) -> std::string_view {
    return (id >= 0 && (size_t)id < b->ref_names.size()) ? std::string_view(b->ref_names[(size_t)id]) : std::string_view("*");
  };
  auto key = [&](int64_t i, GkAlnKey& k) {
    const uint8_t* p = base + b->recs[(size_t)i].off;
    const int32_t ref_id = rds32(p), next_ref = rds32(p + 20);
    k.name = std::string_view((const char*)p + 32, strnlen((const char*)p + 32, p[8]));
    k.ref = ref_name(ref_id);
    k.flag = rd16(p + 14);
    k.pos = (long)rds32(p + 4) + 1;
    k.next_pos = (long)rds32(p + 24) + 1;
    k.mate_same_ref = next_ref >= 0 && next_ref == ref_id;
  };
  std::vector<int> gene_of_ref(b->ref_names.size());   // reference id of the file -> backbone ordinal of the index
  for (size_t i = 0; i < gene_of_ref.size(); ++i) gene_of_ref[i] = gk_packer_gene_of(pk, b->ref_names[i]);
  auto soon = [&](int64_t i, bool head_only) {   // in name order the records are scattered over the inflated stream
    const gk_bam::Rec& rec = b->recs[(size_t)i];
    const uint8_t* p = base + rec.off;
    if (head_only || rec.size <= 320) {
      for (uint32_t o = 0; o < (head_only ? 128u : rec.size); o += 64) __builtin_prefetch(p + o);
      return;
    }
    // a long record: its head (fixed fields, name, CIGAR, first bases) and its tail (the optional fields); the base
    // qualities in between are never read, and a core has only so many line fills in flight
    __builtin_prefetch(p);
    __builtin_prefetch(p + 64);
    __builtin_prefetch(p + 128);
    for (uint32_t o = rec.size - 192; o < rec.size; o += 64) __builtin_prefetch(p + o);
  };
  auto full = [&](int64_t i, GkAlnRecord& r) {
    const gk_bam::Rec& rec = b->recs[(size_t)i];
    const uint8_t* p = base + rec.off;
    const uint32_t l_name = p[8], n_cig = rd16(p + 12), l_seq = rd32(p + 16);
    const int32_t ref_id = rds32(p);
    r.ref = ref_name(ref_id);
    r.gene = (ref_id >= 0 && (size_t)ref_id < gene_of_ref.size()) ? gene_of_ref[(size_t)ref_id] : gk_packer_gene_of(pk, r.ref);
    r.flag = rd16(p + 14);
    r.pos = (long)rds32(p + 4) + 1;
    const uint8_t* cig = p + 32 + l_name;
    const uint8_t* seq = cig + 4ull * n_cig;
    const uint8_t* tags = seq + (l_seq + 1) / 2 + l_seq;
    const uint8_t* end = p + rec.size;
    bool plain = n_cig >= 1 && l_seq >= 1;   // every op one of M I D N S: CIGAR and SEQ go over as they are
    for (uint32_t c = 0; c < n_cig && plain; ++c) plain = (cig[4ull * c] & 15u) <= 4u;
    if (plain) {
      r.bam_cigar = cig; r.n_bam_cigar = n_cig;
      r.bam_seq = seq; r.l_bam_seq = l_seq;
      r.cigar = r.seq = std::string_view();
    } else {                                 // the rare rest as the text a SAM line would hold
      r.bam_cigar = r.bam_seq = nullptr;
      r.cigar_text.clear();
      if (n_cig == 0) r.cigar_text.push_back('*');
      for (uint32_t c = 0; c < n_cig; ++c) {
        const uint32_t v = rd32(cig + 4ull * c);
        append_int(r.cigar_text, v >> 4);
        r.cigar_text.push_back((v & 15u) < 9 ? "MIDNSHP=X"[v & 15u] : '?');
      }
      static const char kBase[] = "=ACMGRSVTWYHKDBN";
      r.seq_text.resize(l_seq ? l_seq : 1);
      if (!l_seq) r.seq_text[0] = '*';
      for (uint32_t q = 0; q < l_seq; ++q) r.seq_text[q] = kBase[(seq[q >> 1] >> ((~q & 1u) << 2)) & 15u];
      r.cigar = r.cigar_text;
      r.seq = r.seq_text;
    }
    r.has_nm = r.has_md = r.has_zs = false;
    r.nh = 1;
    bool has_nh = false;
    while (tags + 3 <= end) {
      const char t0 = (char)tags[0], t1 = (char)tags[1], type = (char)tags[2];
      const uint8_t* v = tags + 3;
      size_t used = 0;
      long ival = 0;
      bool is_int = true;
      const size_t avail = (size_t)(end - v);   // a value is only read once it is known to lie inside the record
      switch (type) {
        case 'A': used = 1; is_int = false; break;
        case 'c': if (avail < 1) return; used = 1; ival = (int8_t)v[0]; break;
        case 'C': if (avail < 1) return; used = 1; ival = v[0]; break;
        case 's': if (avail < 2) return; used = 2; ival = (int16_t)rd16(v); break;
        case 'S': if (avail < 2) return; used = 2; ival = rd16(v); break;
        case 'i': if (avail < 4) return; used = 4; ival = rds32(v); break;
        case 'I': if (avail < 4) return; used = 4; ival = (long)rd32(v); break;
        case 'f': used = 4; is_int = false; break;
        case 'Z': case 'H': {
          const void* z = memchr(v, 0, avail);
          if (!z) return;
          const size_t n = (const uint8_t*)z - v;
          if (type == 'Z' && t0 == 'M' && t1 == 'D' && !r.has_md) { r.has_md = true; r.md = std::string_view((const char*)v, n); }
          if (type == 'Z' && t0 == 'Z' && t1 == 's' && !r.has_zs) { r.has_zs = true; r.zs = std::string_view((const char*)v, n); }
          used = n + 1; is_int = false;
          break;
        }
        case 'B': {
          if (avail < 5) return;
          const char sub = (char)v[0];
          const size_t w = (sub == 'c' || sub == 'C') ? 1 : (sub == 's' || sub == 'S') ? 2 : 4;
          used = 5 + (size_t)rd32(v + 1) * w; is_int = false;
          break;
        }
        default: return;
      }
      if (v + used > end) return;
      if (is_int && t0 == 'N' && t1 == 'M') { r.has_nm = true; r.nm = ival; }            // the last NM counts (hisat2.py:564-567)
      if (is_int && t0 == 'N' && t1 == 'H' && !has_nh && ival >= 0) { has_nh = true; r.nh = ival; }   // getNH: the first
      tags = v + used;
    }
  };
  // collated records are read front to back: the hardware prefetcher does what `soon` is for
  if (b->name_sorted && b->collated) return gk_packer_feed_records(pk, (int64_t)b->recs.size(), true, key, full, nullptr);
  if (b->name_sorted) return gk_packer_feed_records(pk, (int64_t)b->recs.size(), true, key, full, soon);
  return gk_packer_feed_records(pk, (int64_t)b->recs.size(), false, key, full);
}

int gk_bam_close(gk_bam* b) {
  delete b;
  return GK_OK;
}

int gk_bam_info(gk_bam* b, int64_t* n_records, int64_t* header_bytes, int32_t* n_ref) {
  if (!b) { gk_set_error("null handle"); return GK_ERR_ARG; }
  if (n_records) *n_records = (int64_t)b->recs.size();
  if (header_bytes) *header_bytes = (int64_t)b->header.size();
  if (n_ref) *n_ref = (int32_t)b->ref_names.size();
  return GK_OK;
}

int gk_bam_header(gk_bam* b, char* text_out, int64_t capacity) {
  if (!b || !text_out || capacity < (int64_t)b->header.size()) { gk_set_error("header buffer too small"); return GK_ERR_ARG; }
  memcpy(text_out, b->header.data(), b->header.size());
  return GK_OK;
}

// Whole lines ('\n' terminated) in output order until the buffer is full; *n_written == 0 at the end.
// Records are rendered in batches, each batch split over the ingest threads.
static int bam_next_impl(gk_bam* b, char* text_out, int64_t capacity, int64_t* n_written) {
  if (!b || !text_out || !n_written || capacity < 1) { gk_set_error("bad arguments"); return GK_ERR_ARG; }
  int64_t w = 0;
  while (true) {
    if (b->staged_off >= b->staged.size()) {
      b->staged.clear();
      b->staged_off = 0;
      if (b->next >= b->recs.size()) break;
      const size_t first = b->next, last = std::min(b->recs.size(), first + (size_t)65536);
      const int n_thr = (int)std::min<size_t>((size_t)ingest_threads(), (last - first + 1023) / 1024);
      std::vector<std::string> part((size_t)std::max(n_thr, 1));
      std::vector<long long> bad((size_t)std::max(n_thr, 1), -1);
      auto work = [&](int t, size_t a, size_t e) {
        std::string line;
        part[(size_t)t].reserve((e - a) * 480);
        for (size_t i = a; i < e; ++i) {
          if (!render(*b, b->recs[i], line)) { bad[(size_t)t] = (long long)i; return; }
          part[(size_t)t] += line;
          part[(size_t)t].push_back('\n');
        }
      };
      const size_t n = last - first;
      if (n_thr <= 1) {
        work(0, first, last);
      } else {
        std::vector<std::thread> pool;
        for (int t = 0; t < n_thr; ++t) pool.emplace_back(work, t, first + n * t / n_thr, first + n * (t + 1) / n_thr);
        for (auto& th : pool) th.join();
      }
      for (long long i : bad)
        if (i >= 0) { gk_set_error("malformed alignment record %lld", i); return GK_ERR_ARG; }
      for (auto& p : part) b->staged += p;
      b->next = last;
    }
    // hand out whole lines
    const size_t avail = b->staged.size() - b->staged_off;
    size_t take = std::min<size_t>(avail, (size_t)(capacity - w));
    if (take < avail) {   // cut at the last newline that fits
      const size_t nl = b->staged.rfind('\n', b->staged_off + take - 1);
      take = (nl == std::string::npos || nl < b->staged_off) ? 0 : nl + 1 - b->staged_off;
    }
    if (take == 0) {
      if (w == 0) { gk_set_error("a line does not fit the buffer of %lld bytes", (long long)capacity); return GK_ERR_CAPACITY; }
      break;
    }
    memcpy(text_out + w, b->staged.data() + b->staged_off, take);
    w += (int64_t)take;
    b->staged_off += take;
    if (w == capacity) break;
  }
  *n_written = w;
  return GK_OK;
}

}  // extern "C"

// The entry points that allocate with the size of the input: running out of memory is an error code, not an abort
// (an exception must not cross the C boundary).
template <typename Call>
static int guarded(const char* what, const Call& call) {
  try {
    return call();
  } catch (const std::bad_alloc&) {
    gk_set_error("%s: out of host memory", what);
    return GK_ERR_CAPACITY;
  } catch (const std::exception& e) {
    gk_set_error("%s: %s", what, e.what());
    return GK_ERR_ARG;
  }
}

extern "C" int gk_bam_open(const char* path, int32_t name_sorted, gk_bam** out) {
  return guarded("gk_bam_open", [&] { return bam_open_impl(path, name_sorted, out); });
}
extern "C" int gk_bam_pack(gk_bam* b, gk_packer* pk) {
  return guarded("gk_bam_pack", [&] { return bam_pack_impl(b, pk); });
}
extern "C" int gk_bam_next(gk_bam* b, char* text_out, int64_t capacity, int64_t* n_written) {
  return guarded("gk_bam_next", [&] { return bam_next_impl(b, text_out, capacity, n_written); });
}

// ---------------------------------------------------------------------------------------------
// BAM writer: SAM text (header lines + alignment lines) -> BGZF-compressed BAM, the native form of
// utils.samtobam / hisat2.saveReadsToBam (`samtools sort` of the rewritten SAM, hisat2.py:869-901).
// Records are optionally ordered by (reference, position) like `samtools sort` (stable; unmapped
// last); BGZF blocks are deflated in parallel.  No .bai index is written.
namespace {

struct SamHeaderRefs {
  std::vector<std::string> names;
  std::vector<uint32_t> lengths;
};

void put32(std::string& s, uint32_t v) { char b[4] = {(char)v, (char)(v >> 8), (char)(v >> 16), (char)(v >> 24)}; s.append(b, 4); }
void put16(std::string& s, uint32_t v) { char b[2] = {(char)v, (char)(v >> 8)}; s.append(b, 2); }

// UCSC binning scheme (SAM specification section 5.3) for the 0-based half-open interval [beg, end)
uint32_t reg2bin(int64_t beg, int64_t end) {
  --end;
  if (beg >> 14 == end >> 14) return (uint32_t)(((1 << 15) - 1) / 7 + (beg >> 14));
  if (beg >> 17 == end >> 17) return (uint32_t)(((1 << 12) - 1) / 7 + (beg >> 17));
  if (beg >> 20 == end >> 20) return (uint32_t)(((1 << 9) - 1) / 7 + (beg >> 20));
  if (beg >> 23 == end >> 23) return (uint32_t)(((1 << 6) - 1) / 7 + (beg >> 23));
  if (beg >> 26 == end >> 26) return (uint32_t)(((1 << 3) - 1) / 7 + (beg >> 26));
  return 0;
}

bool parse_ll(std::string_view s, long long& v) {
  if (s.empty()) return false;
  size_t i = 0;
  bool neg = false;
  if (s[0] == '-' || s[0] == '+') { neg = s[0] == '-'; i = 1; }
  if (i >= s.size()) return false;
  long long x = 0;
  for (; i < s.size(); ++i) {
    if (!isdigit((unsigned char)s[i])) return false;
    if (x > (LLONG_MAX - 9) / 10) return false;     // no field of an alignment line is that large
    x = x * 10 + (s[i] - '0');
  }
  v = neg ? -x : x;
  return true;
}

bool encode_tag(std::string_view f, std::string& out) {
  if (f.size() < 5 || f[2] != ':' || f[4] != ':') return false;
  out.append(f.data(), 2);
  const char type = f[3];
  std::string_view val = f.substr(5);
  long long v;
  switch (type) {
    case 'i':
      if (!parse_ll(val, v)) return false;
      if (v >= -128 && v <= 127) { out.push_back('c'); out.push_back((char)v); }
      else if (v >= 0 && v <= 255) { out.push_back('C'); out.push_back((char)v); }
      else if (v >= -32768 && v <= 32767) { out.push_back('s'); put16(out, (uint32_t)v); }
      else if (v >= 0 && v <= 65535) { out.push_back('S'); put16(out, (uint32_t)v); }
      else if (v >= -2147483648LL && v <= 2147483647LL) { out.push_back('i'); put32(out, (uint32_t)v); }
      else if (v >= 0 && v <= 4294967295LL) { out.push_back('I'); put32(out, (uint32_t)v); }
      else return false;
      return true;
    case 'A':
      if (val.size() != 1) return false;
      out.push_back('A'); out.push_back(val[0]);
      return true;
    case 'f': {
      const float fl = strtof(std::string(val).c_str(), nullptr);
      uint32_t u; memcpy(&u, &fl, 4);
      out.push_back('f'); put32(out, u);
      return true;
    }
    case 'Z': case 'H':
      out.push_back(type); out.append(val); out.push_back('\0');
      return true;
    case 'B': {
      if (val.empty()) return false;
      const char sub = val[0];
      std::vector<std::string_view> items;
      size_t a = 1;
      while (a < val.size()) {
        if (val[a] != ',') return false;
        const size_t c = val.find(',', a + 1);
        items.push_back(val.substr(a + 1, c == std::string_view::npos ? std::string_view::npos : c - a - 1));
        a = c == std::string_view::npos ? val.size() : c;
      }
      out.push_back('B'); out.push_back(sub); put32(out, (uint32_t)items.size());
      for (auto it : items) {
        if (sub == 'f') { const float fl = strtof(std::string(it).c_str(), nullptr); uint32_t u; memcpy(&u, &fl, 4); put32(out, u); continue; }
        if (!parse_ll(it, v)) return false;
        if (sub == 'c' || sub == 'C') out.push_back((char)v);
        else if (sub == 's' || sub == 'S') put16(out, (uint32_t)v);
        else if (sub == 'i' || sub == 'I') put32(out, (uint32_t)v);
        else return false;
      }
      return true;
    }
    default: return false;
  }
}

// one alignment line -> BAM record (with its leading block_size); ref_id / pos returned for sorting
bool encode_record(std::string_view line, const std::vector<std::string>& ref_names, std::string& out, int32_t& ref_id,
                   int32_t& pos0) {
  std::vector<std::string_view> f;
  size_t a = 0;
  for (;;) {
    const size_t t = line.find('\t', a);
    f.push_back(line.substr(a, t == std::string_view::npos ? std::string_view::npos : t - a));
    if (t == std::string_view::npos) break;
    a = t + 1;
  }
  if (f.size() < 11) return false;
  long long flag, pos, mapq, pnext, tlen;
  if (!parse_ll(f[1], flag) || !parse_ll(f[3], pos) || !parse_ll(f[4], mapq) || !parse_ll(f[7], pnext) || !parse_ll(f[8], tlen))
    return false;
  auto find_ref = [&](std::string_view n) -> int32_t {
    for (size_t i = 0; i < ref_names.size(); ++i) if (ref_names[i] == n) return (int32_t)i;
    return -1;
  };
  ref_id = f[2] == "*" ? -1 : find_ref(f[2]);
  const int32_t next_id = f[6] == "=" ? ref_id : (f[6] == "*" ? -1 : find_ref(f[6]));
  pos0 = (int32_t)(pos - 1);                      // 64-bit arithmetic first: POS / PNEXT of a damaged line may be INT32_MIN
  std::vector<uint32_t> cig;
  int64_t ref_len = 0;
  if (f[5] != "*") {
    long long n = 0;
    bool have = false;
    for (char c : f[5]) {
      if (isdigit((unsigned char)c)) {
        n = n * 10 + (c - '0');
        have = true;
        if (n > 0xFFFFFFF) return false;           // a BAM op length has 28 bits (and n must not overflow)
        continue;
      }
      const char* ops = "MIDNSHP=X";
      const char* at = strchr(ops, c);
      if (!at || !have) return false;
      const uint32_t op = (uint32_t)(at - ops);
      cig.push_back((uint32_t)n << 4 | op);
      if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) ref_len += n;
      n = 0; have = false;
    }
  }
  const std::string_view seq = f[9], qual = f[10];
  const uint32_t l_seq = seq == "*" ? 0u : (uint32_t)seq.size();
  std::string body;
  put32(body, (uint32_t)ref_id);
  put32(body, (uint32_t)pos0);
  body.push_back((char)(f[0].size() + 1));
  body.push_back((char)mapq);
  put16(body, reg2bin(pos0, pos0 + (ref_len ? ref_len : 1)));
  put16(body, (uint32_t)cig.size());
  put16(body, (uint32_t)flag);
  put32(body, l_seq);
  put32(body, (uint32_t)next_id);
  put32(body, (uint32_t)(int32_t)(pnext - 1));
  put32(body, (uint32_t)(int32_t)tlen);
  body.append(f[0]); body.push_back('\0');
  for (uint32_t c : cig) put32(body, c);
  static const char kBase[] = "=ACMGRSVTWYHKDBN";
  for (uint32_t i = 0; i < l_seq; i += 2) {
    auto code = [&](char c) -> uint32_t { const char* at = strchr(kBase, toupper((unsigned char)c)); return at ? (uint32_t)(at - kBase) : 15u; };
    body.push_back((char)(code(seq[i]) << 4 | (i + 1 < l_seq ? code(seq[i + 1]) : 0u)));
  }
  if (qual == "*" || qual.size() != l_seq) body.append(l_seq, (char)0xFF);
  else for (char c : qual) body.push_back((char)(c - 33));
  for (size_t i = 11; i < f.size(); ++i)
    if (!encode_tag(f[i], body)) return false;
  put32(out, (uint32_t)body.size());
  out += body;
  return true;
}

bool deflate_block(const uint8_t* data, size_t n, std::string& out) {
  z_stream zs;
  memset(&zs, 0, sizeof(zs));
  if (deflateInit2(&zs, 6, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) return false;
  std::vector<uint8_t> buf(compressBound((uLong)n) + 64);
  zs.next_in = const_cast<Bytef*>(data);
  zs.avail_in = (uInt)n;
  zs.next_out = buf.data();
  zs.avail_out = (uInt)buf.size();
  const int rc = deflate(&zs, Z_FINISH);
  const size_t clen = buf.size() - zs.avail_out;
  deflateEnd(&zs);
  if (rc != Z_STREAM_END || clen + 26 > 65536) return false;
  static const uint8_t head[16] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0};
  out.append((const char*)head, 16);
  put16(out, (uint32_t)(clen + 25));   // BSIZE = total block size - 1
  out.append((const char*)buf.data(), clen);
  put32(out, (uint32_t)crc32(crc32(0L, Z_NULL, 0), data, (uInt)n));
  put32(out, (uint32_t)n);
  return true;
}

}  // namespace

namespace {

// lines of a text: '@' lines go to the header (and its @SQ records to refs), the others to `lines`
void split_sam_text(std::string_view text, std::string& header, SamHeaderRefs& refs, std::vector<std::string_view>* lines) {
  for (size_t a = 0; a < text.size();) {
    size_t nl = text.find('\n', a);
    if (nl == std::string_view::npos) nl = text.size();
    std::string_view line = text.substr(a, nl - a);
    if (!line.empty() && line.back() == '\r') line.remove_suffix(1);
    a = nl + 1;
    if (line.empty()) continue;
    if (line[0] == '@') {
      header.append(line); header.push_back('\n');
      if (line.substr(0, 3) == "@SQ") {
        std::string name; long long len = 0;
        for (size_t p = 0; p < line.size();) {
          size_t t = line.find('\t', p);
          std::string_view fld = line.substr(p, t == std::string_view::npos ? std::string_view::npos : t - p);
          if (fld.substr(0, 3) == "SN:") name = std::string(fld.substr(3));
          if (fld.substr(0, 3) == "LN:") parse_ll(fld.substr(3), len);
          if (t == std::string_view::npos) break;
          p = t + 1;
        }
        refs.names.push_back(name); refs.lengths.push_back((uint32_t)len);
      }
    } else if (lines) {
      lines->push_back(line);
    }
  }
}

int write_bam_lines(const char* path, const std::string& header, const SamHeaderRefs& refs,
                    const std::vector<std::string_view>& lines, int32_t coordinate_sort);

}  // namespace

extern "C" int gk_bam_write(const char* path, const char* sam_text, int64_t n_bytes, int32_t coordinate_sort) {
  if (!path || (!sam_text && n_bytes)) { gk_set_error("null argument"); return GK_ERR_ARG; }
  std::string header;
  SamHeaderRefs refs;
  std::vector<std::string_view> lines;
  split_sam_text(std::string_view(sam_text, (size_t)n_bytes), header, refs, &lines);
  return write_bam_lines(path, header, refs, lines, coordinate_sort);
}

// The selected lines of a SAM text (0-based line numbers, in the given order) under the given header: the
// .bam / .no_multi.bam rewrites of the filter-passing pairs (hisat2.py:869-901) without building their text.
extern "C" int gk_bam_write_lines(const char* path, const char* header_text, int64_t n_header, const char* sam_text,
                                  int64_t n_bytes, const int64_t* line_idx, int64_t n_lines, int32_t coordinate_sort) {
  if (!path || (!sam_text && n_bytes) || (!line_idx && n_lines) || (!header_text && n_header)) {
    gk_set_error("null argument");
    return GK_ERR_ARG;
  }
  std::string header;
  SamHeaderRefs refs;
  split_sam_text(std::string_view(header_text, (size_t)n_header), header, refs, nullptr);
  const std::string_view text(sam_text, (size_t)n_bytes);
  const std::vector<int64_t> starts = gk_line_starts(text);
  std::vector<std::string_view> lines((size_t)n_lines);
  for (int64_t i = 0; i < n_lines; ++i) {
    if (line_idx[i] < 0 || (size_t)line_idx[i] + 1 >= starts.size()) { gk_set_error("line number out of range"); return GK_ERR_ARG; }
    lines[(size_t)i] = gk_line_at(text, starts, line_idx[i]);
  }
  return write_bam_lines(path, header, refs, lines, coordinate_sort);
}

namespace {

int write_bam_lines(const char* path, const std::string& header, const SamHeaderRefs& refs,
                    const std::vector<std::string_view>& lines, int32_t coordinate_sort) {
  struct Enc { std::string rec; int32_t ref_id, pos0; };
  std::vector<Enc> enc(lines.size());
  std::vector<char> bad((size_t)std::max(ingest_threads(), 1), 0);
  {
    const size_t n = lines.size();
    const int n_thr = (int)std::min<size_t>((size_t)ingest_threads(), std::max<size_t>(n / 1024, 1));
    auto work = [&](int t, size_t a, size_t b) {
      for (size_t i = a; i < b; ++i)
        if (!encode_record(lines[i], refs.names, enc[i].rec, enc[i].ref_id, enc[i].pos0)) { bad[(size_t)t] = 1; return; }
    };
    if (n_thr <= 1) work(0, 0, n);
    else {
      std::vector<std::thread> pool;
      for (int t = 0; t < n_thr; ++t) pool.emplace_back(work, t, n * t / n_thr, n * (t + 1) / n_thr);
      for (auto& th : pool) th.join();
    }
  }
  for (char b : bad) if (b) { gk_set_error("malformed SAM line"); return GK_ERR_ARG; }
  std::vector<uint32_t> order(enc.size());
  for (size_t i = 0; i < order.size(); ++i) order[i] = (uint32_t)i;
  if (coordinate_sort)
    std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) {
      const uint32_t rx = (uint32_t)enc[x].ref_id, ry = (uint32_t)enc[y].ref_id;   // -1 (unmapped) sorts last
      if (rx != ry) return rx < ry;
      return enc[x].pos0 < enc[y].pos0;
    });
  std::string raw("BAM\1", 4);
  put32(raw, (uint32_t)header.size());
  raw += header;
  put32(raw, (uint32_t)refs.names.size());
  for (size_t i = 0; i < refs.names.size(); ++i) {
    put32(raw, (uint32_t)refs.names[i].size() + 1);
    raw += refs.names[i]; raw.push_back('\0');
    put32(raw, refs.lengths[i]);
  }
  for (uint32_t i : order) raw += enc[i].rec;
  // BGZF blocks of <= 0xff00 input bytes, deflated in parallel, written in order, then the EOF block
  const size_t kBlock = 0xff00, n_blocks = (raw.size() + kBlock - 1) / kBlock;
  std::vector<std::string> comp(n_blocks + 1);
  std::vector<char> bad2(n_blocks + 1, 0);
  {
    const int n_thr = (int)std::min<size_t>((size_t)ingest_threads(), std::max<size_t>(n_blocks / 4, 1));
    auto work = [&](size_t a, size_t b) {
      for (size_t i = a; i < b; ++i) {
        const size_t off = i * kBlock, len = std::min(kBlock, raw.size() - off);
        if (!deflate_block((const uint8_t*)raw.data() + off, len, comp[i])) bad2[i] = 1;
      }
    };
    if (n_thr <= 1) work(0, n_blocks);
    else {
      std::vector<std::thread> pool;
      for (int t = 0; t < n_thr; ++t) pool.emplace_back(work, n_blocks * t / n_thr, n_blocks * (t + 1) / n_thr);
      for (auto& th : pool) th.join();
    }
  }
  if (!deflate_block((const uint8_t*)"", 0, comp[n_blocks])) bad2[n_blocks] = 1;
  for (char b : bad2) if (b) { gk_set_error("deflate failed"); return GK_ERR_ARG; }
  FILE* f = fopen(path, "wb");
  if (!f) { gk_set_error("cannot write %s", path); return GK_ERR_ARG; }
  bool ok = true;
  for (auto& c : comp) ok = ok && fwrite(c.data(), 1, c.size(), f) == c.size();
  ok = fclose(f) == 0 && ok;
  if (!ok) { gk_set_error("short write to %s", path); return GK_ERR_ARG; }
  if (!coordinate_sort) return GK_OK;

  // ---- {path}.bai (SAM specification, section 5.2): what `samtools index` adds in utils.samtobam.
  // Virtual offset of a byte of the BAM stream = file offset of its BGZF block << 16 | offset in the block.
  std::vector<uint64_t> block_at(n_blocks + 1, 0);
  for (size_t i = 0; i < n_blocks; ++i) block_at[i + 1] = block_at[i] + comp[i].size();
  auto voffset = [&](size_t raw_off) { return block_at[raw_off / kBlock] << 16 | (uint64_t)(raw_off % kBlock); };
  struct Chunk { uint64_t beg, end; };
  struct RefIndex {
    std::map<uint32_t, std::vector<Chunk>> bins;
    std::vector<uint64_t> linear;
    uint64_t first = 0, last = 0, n_mapped = 0, n_unmapped = 0;
    bool any = false;
  };
  std::vector<RefIndex> index(refs.names.size());
  uint64_t n_no_coor = 0;
  size_t at = raw.size();
  for (uint32_t i : order) at -= enc[i].rec.size();   // first record = end of the header part
  for (uint32_t i : order) {
    const std::string& rec = enc[i].rec;
    const uint64_t v0 = voffset(at), v1 = voffset(at + rec.size());
    at += rec.size();
    const uint8_t* p = (const uint8_t*)rec.data() + 4;
    const int32_t ref_id = rds32(p), pos = rds32(p + 4);
    if (ref_id < 0 || (size_t)ref_id >= index.size() || pos < 0) { ++n_no_coor; continue; }
    const uint32_t l_name = p[8], bin = rd16(p + 10), n_cig = rd16(p + 12), flag = rd16(p + 14);
    int64_t span = 0;
    for (uint32_t c = 0; c < n_cig; ++c) {
      const uint32_t v = rd32(p + 32 + l_name + 4ull * c), op = v & 15u;
      if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) span += v >> 4;
    }
    const int64_t end = pos + (span > 0 ? span : 1);
    RefIndex& ri = index[(size_t)ref_id];
    auto& chunks = ri.bins[bin];
    if (!chunks.empty() && chunks.back().end == v0) chunks.back().end = v1;   // adjacent records of one bin: one chunk
    else chunks.push_back({v0, v1});
    const size_t w0 = (size_t)(pos >> 14), w1 = (size_t)((end - 1) >> 14);
    if (ri.linear.size() <= w1) ri.linear.resize(w1 + 1, 0);
    for (size_t w = w0; w <= w1; ++w)
      if (!ri.linear[w]) ri.linear[w] = v0;
    if (!ri.any) { ri.first = v0; ri.any = true; }
    ri.last = v1;
    if (flag & 4u) ++ri.n_unmapped; else ++ri.n_mapped;
  }
  std::string bai("BAI\1", 4);
  auto put64 = [&](uint64_t v) { put32(bai, (uint32_t)v); put32(bai, (uint32_t)(v >> 32)); };
  put32(bai, (uint32_t)index.size());
  for (RefIndex& ri : index) {
    put32(bai, (uint32_t)(ri.bins.size() + (ri.any ? 1 : 0)));
    for (auto& kv : ri.bins) {
      put32(bai, kv.first);
      put32(bai, (uint32_t)kv.second.size());
      for (const Chunk& c : kv.second) { put64(c.beg); put64(c.end); }
    }
    if (ri.any) {   // the metadata pseudo-bin htslib writes: file range of the reference, mapped / unmapped counts
      put32(bai, 37450u);
      put32(bai, 2u);
      put64(ri.first); put64(ri.last); put64(ri.n_mapped); put64(ri.n_unmapped);
    }
    for (size_t w = 1; w < ri.linear.size(); ++w)
      if (!ri.linear[w]) ri.linear[w] = ri.linear[w - 1];   // windows nothing starts in point at the previous one
    put32(bai, (uint32_t)ri.linear.size());
    for (uint64_t v : ri.linear) put64(v);
  }
  put64(n_no_coor);
  const std::string bai_path = std::string(path) + ".bai";
  FILE* fi = fopen(bai_path.c_str(), "wb");
  if (!fi) { gk_set_error("cannot write %s", bai_path.c_str()); return GK_ERR_ARG; }
  ok = fwrite(bai.data(), 1, bai.size(), fi) == bai.size();
  ok = fclose(fi) == 0 && ok;
  if (!ok) { gk_set_error("short write to %s", bai_path.c_str()); return GK_ERR_ARG; }
  return GK_OK;
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// Base counts per reference position: the native form of pileup.getPileupBaseRatio (pileup.py:57-81),
// which parses `samtools mpileup -a bam`.  The model of mpileup's defaults, from its manual page:
// records flagged UNMAP / SECONDARY / QCFAIL / DUP are skipped, so are paired reads that are not in a
// proper pair ("anomalous"); bases of quality < 13 are not counted; where the two mates of a pair
// overlap, equal bases count once (the later mate's quality becomes 0), unequal ones keep only the
// better base at 0.8 of its quality; a deleted position counts as '*'; no BAQ (no reference given).
// Not modelled: the 8000-read depth cap.  Counts are [position][A, C, G, T, N, *]; gene_off[g] is
// the first position of reference g (header order) in the concatenated position space.
extern "C" int gk_bam_pileup(gk_bam* b, const int64_t* gene_off, int32_t n_gene, uint32_t* counts_out) {
  if (!b || !gene_off || !counts_out || n_gene <= 0) { gk_set_error("bad pileup arguments"); return GK_ERR_ARG; }
  const int64_t total = gene_off[n_gene];
  memset(counts_out, 0, (size_t)total * 6 * sizeof(uint32_t));
  const uint8_t* base = b->data.data();
  struct Cov { int32_t pos; uint8_t code; uint8_t qual; bool is_del; };
  auto code_of = [](uint8_t nib) -> uint8_t {   // BAM nibble -> A C G T N
    switch (nib) { case 1: return 0; case 2: return 1; case 4: return 2; case 8: return 3; default: return 4; }
  };
  auto usable = [&](const uint8_t* p) {
    const uint32_t flag = rd16(p + 14);
    if (flag & (4u | 256u | 512u | 1024u)) return false;
    if ((flag & 1u) && !(flag & 2u)) return false;
    return rds32(p) >= 0 && rds32(p) < n_gene;
  };
  // Only positions inside the reference are kept (nothing else is ever counted, and a mate's overlap with its partner
  // only matters where both are counted): a damaged record -- an I / S run longer than the read, a D or N run of 2^28
  // bases -- then costs neither a read outside its block nor memory for positions that do not exist.
  auto cover = [&](const uint8_t* p, std::vector<Cov>& out) {
    out.clear();
    const uint32_t l_name = p[8], n_cig = rd16(p + 12), l_seq = rd32(p + 16);
    const uint8_t* cig = p + 32 + l_name;
    const uint8_t* seq = cig + 4ull * n_cig;
    const uint8_t* qual = seq + (l_seq + 1) / 2;
    const int64_t ref_len = std::min<int64_t>(gene_off[rds32(p) + 1] - gene_off[rds32(p)], INT32_MAX);
    int64_t pos = rds32(p + 4);
    uint32_t ri = 0;                                  // bases of the read consumed, never more than l_seq
    for (uint32_t c = 0; c < n_cig; ++c) {
      const uint32_t v = rd32(cig + 4ull * c), op = v & 15u, len = v >> 4;
      if (op == 0 || op == 7 || op == 8) {
        for (uint32_t k = 0; k < len && ri < l_seq; ++k, ++ri, ++pos)
          if (pos >= 0 && pos < ref_len)
            out.push_back({(int32_t)pos, code_of((seq[ri >> 1] >> ((~ri & 1u) << 2)) & 15u), qual[ri] == 0xFF ? (uint8_t)255 : qual[ri], false});
      } else if (op == 2) {
        const uint8_t q = ri ? (qual[ri - 1] == 0xFF ? (uint8_t)255 : qual[ri - 1]) : (uint8_t)255;
        const int64_t stop = std::min<int64_t>(pos + (int64_t)len, ref_len);
        for (int64_t at = std::max<int64_t>(pos, 0); at < stop; ++at) out.push_back({(int32_t)at, 5, q, true});
        pos += (int64_t)len;
      } else if (op == 1 || op == 4) {
        ri = (uint32_t)std::min<uint64_t>((uint64_t)ri + len, l_seq);
      } else if (op == 3) {
        pos += (int64_t)len;
      }
    }
  };
  // mates of proper pairs by name
  std::unordered_map<std::string, int64_t> first_of;
  std::vector<int64_t> mate((size_t)b->recs.size(), -1);
  for (size_t i = 0; i < b->recs.size(); ++i) {
    const uint8_t* p = base + b->recs[i].off;
    if (!usable(p) || !(rd16(p + 14) & 1u)) continue;
    std::string name((const char*)p + 32, strnlen((const char*)p + 32, p[8]));
    auto it = first_of.find(name);
    if (it == first_of.end()) first_of.emplace(std::move(name), (int64_t)i);
    else { mate[i] = it->second; mate[(size_t)it->second] = (int64_t)i; first_of.erase(it); }
  }
  std::vector<Cov> mine, other;
  for (size_t i = 0; i < b->recs.size(); ++i) {
    const uint8_t* p = base + b->recs[i].off;
    if (!usable(p)) continue;
    cover(p, mine);
    if (mate[i] >= 0) {
      const uint8_t* q = base + b->recs[(size_t)mate[i]].off;
      if (rds32(q) == rds32(p)) {
        cover(q, other);
        // "first" = the mate that starts earlier (file order on a tie)
        const bool i_first = rds32(p + 4) < rds32(q + 4) || (rds32(p + 4) == rds32(q + 4) && (int64_t)i < mate[i]);
        size_t a = 0, c = 0;
        while (a < mine.size() && c < other.size()) {
          if (mine[a].pos < other[c].pos) { ++a; continue; }
          if (mine[a].pos > other[c].pos) { ++c; continue; }
          Cov& x = mine[a];
          const Cov& y = other[c];
          if (!x.is_del && !y.is_del) {
            if (x.code == y.code) {
              x.qual = i_first ? (uint8_t)std::min<int>(200, (int)x.qual + (int)y.qual) : (uint8_t)0;
            } else {
              const bool x_wins = i_first ? x.qual >= y.qual : x.qual > y.qual;
              x.qual = x_wins ? (uint8_t)(0.8 * x.qual) : (uint8_t)0;
            }
          }
          ++a; ++c;
        }
      }
    }
    const int64_t off = gene_off[rds32(p)], len = gene_off[rds32(p) + 1] - off;
    for (const Cov& x : mine)
      if (x.qual >= 13 && x.pos >= 0 && x.pos < len) counts_out[(off + x.pos) * 6 + x.code] += 1;
  }
  return GK_OK;
}

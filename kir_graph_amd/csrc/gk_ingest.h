// Internal interface between the alignment readers and the packer (gk_sampack.cpp): what the packer
// needs to know about one alignment record, whatever it was read from.
#pragma once
#include <cstdint>
#include <functional>
#include <memory>
#include <string>
#include <string_view>
#include <algorithm>
#include <cstring>
#include <vector>

#include <chrono>
#include <cstdio>
#include <cstdlib>

#include <thread>

#include "gk_env.h"

struct gk_packer;

// allocator whose resize() leaves new elements uninitialised: buffers of hundreds of megabytes that the
// worker threads are about to overwrite (and first-touch in parallel)
template <typename T>
struct GkRawInit : std::allocator<T> {
  template <typename U> struct rebind { using other = GkRawInit<U>; };
  template <typename U> void construct(U* p) noexcept { ::new ((void*)p) U; }
  template <typename U, typename... A> void construct(U* p, A&&... a) { ::new ((void*)p) U(std::forward<A>(a)...); }
};

// threads of the native readers / packers / writers: GK_PACK_THREADS (default 8), at most the hardware's
inline int gk_ingest_threads() {
  const char* e = getenv("GK_PACK_THREADS");
  long n = e ? atol(e) : 8;
  const long hw = (long)std::thread::hardware_concurrency();
  if (hw > 0) n = std::min(n, hw);
  return (int)std::max<long>(1, std::min<long>(n, 64));
}

// GK_TRACE=ingest: phase times of the host ingest on stderr (development aid)
struct GkPhaseClock {
  const char* what;
  bool on;
  std::chrono::steady_clock::time_point t;
  explicit GkPhaseClock(const char* w) : what(w), on(gk_trace("ingest")), t(std::chrono::steady_clock::now()) {}
  void lap(const char* phase) {
    if (!on) return;
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "[ingest] %s %s %.1f ms\n", what, phase, std::chrono::duration<double, std::milli>(now - t).count());
    t = now;
  }
};

struct GkAlnKey {            // fields of readPair's pairing rule (hisat2.py:248-258)
  std::string_view name, ref;
  long flag = 0, pos = 0, next_pos = 0;   // positions 1-based like the SAM text
  bool mate_same_ref = false;             // RNEXT is "="
};

struct GkAlnRecord {         // fields of filterRead / getNH / recordToRawVariant
  std::string_view ref, cigar, seq, md, zs;
  std::string cigar_text, seq_text;       // storage when the source is not text (views above point here)
  // A BAM source may hand over CIGAR and SEQ as they lie in the record instead of as text (bam_cigar != nullptr;
  // `cigar` / `seq` are then unset): n_bam_cigar little-endian words `len << 4 | op`, every op one of M I D N S
  // (0..4), at least one of them, and l_bam_seq >= 1 bases of 4 bits ("=ACMGRSVTWYHKDBN", high nibble first).
  // Records outside that shape (other ops, no CIGAR, no SEQ) come as text: their SAM spelling has quirks
  // under the reference's regular expression that only the text walk reproduces.
  const uint8_t* bam_cigar = nullptr;
  const uint8_t* bam_seq = nullptr;
  uint32_t n_bam_cigar = 0, l_bam_seq = 0;
  long flag = 0, pos = 0, nm = 0, nh = 1;
  bool has_nm = false, has_md = false, has_zs = false;
  // backbone ordinal of `ref` when the source knows it (BAM: reference ids map to it once per file), -1 when
  // it is no backbone of the index; kGeneByName: look `ref` up by name
  static constexpr int kGeneByName = -2;
  int gene = kGeneByName;
};

// backbone ordinal of a reference name in the packer's index, -1 when it is none
int gk_packer_gene_of(const gk_packer* pk, std::string_view ref);

// Pair n records in stream order like readPair and pack the emitted pairs (decoding on several
// threads).  key(i, k) and full(i, r) fill the fields of record i; both must be thread-safe and the
// views must stay valid until the call returns.  Line numbers reported for errors / pairs are
// first_line + i.  names_contiguous: records of one name are adjacent (a name-collated stream), which
// lets the pairing run on several threads.  soon(i, head_only), when given, announces that key(i) (head_only) or full(i)
// is about to be called (a source whose records lie scattered in memory can prefetch them).  Returns GK_OK or the packer's error code.
int gk_packer_feed_records(gk_packer* pk, int64_t n, bool names_contiguous,
                           const std::function<void(int64_t, GkAlnKey&)>& key,
                           const std::function<void(int64_t, GkAlnRecord&)>& full,
                           const std::function<void(int64_t, bool)>& soon = nullptr);

// start offset of every line of a text (a final line without '\n' counts), plus the end as sentinel
inline std::vector<int64_t> gk_line_starts(std::string_view text) {
  std::vector<int64_t> starts;
  const char* base = text.data();
  size_t a = 0;
  while (a < text.size()) {
    starts.push_back((int64_t)a);
    const void* nl = memchr(base + a, '\n', text.size() - a);
    if (!nl) { a = text.size(); break; }
    a = (size_t)((const char*)nl - base) + 1;
  }
  starts.push_back((int64_t)text.size() + (text.empty() || text.back() == '\n' ? 0 : 1));
  return starts;
}

// line i without its line terminator ("\n" or "\r\n")
inline std::string_view gk_line_at(std::string_view text, const std::vector<int64_t>& starts, int64_t i) {
  if (i < 0 || (size_t)i + 1 >= starts.size()) return std::string_view();
  size_t a = (size_t)starts[(size_t)i], b = (size_t)starts[(size_t)i + 1];
  if (b > a) --b;                                   // the '\n' (or the sentinel's virtual one)
  b = std::min(b, text.size());
  std::string_view line = text.substr(a, b - a);
  if (!line.empty() && line.back() == '\r') line.remove_suffix(1);
  return line;
}


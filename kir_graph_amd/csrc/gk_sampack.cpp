// Host-side ingest: name-collated SAM text -> packed gk_mate records (no GPU work here).
//
// Native counterpart of kir_graph_amd/packed.py::packPairs + hisat2.pairLines, i.e. of the text side
// of the reference: readPair (hisat2.py:228-276), the field reads of filterRead (551-569), getNH
// (95-100) and the CIGAR / MD / Zs co-walk CHECKS of recordToRawVariant (279-515: asserts at 416-417,
// 461, 512-514; NotImplementedError for N and unknown ops at 499-502).  The variant walk itself runs on
// the device from the records produced here.  Errors are reported with the reference's exception kind
// and the 0-based index of the offending input line; the Python wrapper re-raises them.
//
// A chunk is processed in three steps: (1) the mate pairing, sequential and cheap (it decides the
// emission order, which numbers the novel variants downstream); (2) the record decoding of the
// emitted pairs, on several threads (GK_PACK_THREADS, default 8), each pair independent; (3) a
// sequential merge in emission order that interns the inserted strings -- same ids as a one-by-one
// walk -- and stops at the first pair the reference would have raised on.
#include <algorithm>
#include <atomic>
#include <cctype>
#include <climits>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <new>
#include <string>
#include <string_view>
#include <thread>
#include <unordered_map>
#include <vector>

#include "gk_ingest.h"
#include "graphkir_hip.h"

void gk_set_error(const char* fmt, ...);

namespace {

using sv = std::string_view;

struct LineRef { int64_t begin, len, index; };   // byte range in the stream and line number

// a record waiting for its mate: a view into the chunk being fed, turned into an owned copy only if the
// mate has not arrived by the end of the chunk
struct Pending {
  sv view;
  std::string owned;
  int64_t index;
  int flag;
};

struct MdTok { int kind; long num; char ch; };   // kind 0 = number, 1 = character

}  // namespace

inline uint64_t next_packer_serial() {
  static std::atomic<uint64_t> n{0};
  return ++n;
}

struct gk_packer {
  const uint64_t serial = next_packer_serial();   // tells packers apart even when one reuses another's address
  std::vector<std::string> genes;
  std::unordered_map<std::string, int> gene_id;
  std::unordered_map<std::string, uint32_t> ins_id;
  std::vector<std::string> ins_strings;
  uint32_t n_index_ins = 0;
  std::unordered_map<std::string, Pending> waiting;
  std::vector<gk_mate, GkRawInit<gk_mate>> own;   // the records, unless the caller gave a place for them
  gk_mate* ext = nullptr;            // caller's buffer (gk_packer_set_output) of ext_cap records
  int64_t ext_cap = 0;
  size_t n_mates = 0;                // records made so far (2 per pair)
  std::vector<gk_mate_wide> wide;    // pairs that do not fit gk_mate (2 records per pair) ...
  std::vector<int64_t> spill_pair;   // ... and which pairs they are, ascending
  gk_mate* mates() { return ext ? ext : own.data(); }
  bool resize_mates(size_t n) {      // new records are NOT initialised: the decoder writes every byte of them
    if (ext) {
      if ((int64_t)n > ext_cap) return false;
    } else {
      own.resize(n);
    }
    n_mates = n;
    return true;
  }
  std::vector<int64_t, GkRawInit<int64_t>> pair_lines;   // 2 per pair: line index of left (later) and right (earlier) record (written by the decoding threads: not zero-filled first)
  struct Job { sv left, right; std::string right_owned; int64_t left_idx, right_idx; };
  std::vector<Job> jobs;             // pairs emitted by the pairing pass of the current chunk
  std::string carry;                 // partial last line of the previous chunk
  int64_t n_lines = 0, n_reads = 0, n_pairs = 0, n_strange = 0;
  int err_kind = 0;                  // 0 none, 1 AssertionError, 2 NotImplementedError, 3 capacity, 4 ValueError
  int64_t err_line = -1;
  std::string err_msg;
};

namespace {

sv strip(sv s) {
  size_t a = 0, b = s.size();
  while (a < b && isspace((unsigned char)s[a])) ++a;
  while (b > a && isspace((unsigned char)s[b - 1])) --b;
  return s.substr(a, b - a);
}

void split_tabs(sv s, std::vector<sv>& out) {
  out.clear();
  size_t a = 0;
  for (;;) {
    size_t t = s.find('\t', a);
    if (t == sv::npos) { out.push_back(s.substr(a)); return; }
    out.push_back(s.substr(a, t - a));
    a = t + 1;
  }
}

// x * 10 + digit that stops growing instead of overflowing: a run of twenty digits in a damaged line is a number no
// check downstream accepts either way (Python's int() would hold it; what follows only compares it)
inline long push_digit(long x, char c) {
  constexpr long kTop = (LONG_MAX - 9) / 10;
  return x > kTop ? x : x * 10 + (c - '0');
}

bool to_long(sv s, long& v) {   // Python int(): optional sign, digits, surrounding blanks
  s = strip(s);
  if (s.empty()) return false;
  size_t i = 0;
  bool neg = false;
  if (s[0] == '+' || s[0] == '-') { neg = s[0] == '-'; i = 1; }
  if (i >= s.size()) return false;
  long x = 0;
  for (; i < s.size(); ++i) {
    if (!isdigit((unsigned char)s[i])) return false;
    x = push_digit(x, s[i]);
  }
  v = neg ? -x : x;
  return true;
}

struct Fail { int kind; std::string msg; };

// What a walk leaves behind: the M / I / D ops, the MD mismatches and the inserted strings.  The first
// GK_MAX_CIG ops and GK_MAX_MM mismatches are kept (what a record can hold); the counts run on so that the
// capacity verdict comes after the walk's own checks, like a walk into growing lists.
struct Walked {
  uint32_t cig[GK_WIDE_CIG];                    // len << 4 | GK_CIG_* (for a clipped mate: its whole CIGAR, S ops included)
  uint32_t mm[GK_WIDE_MM];                      // ref_off << 8 | read base
  size_t n_ops = 0, n_mm = 0, n_indel = 0;      // ops walked (S excluded), mismatches, I + D ops
  bool long_op = false, far_mm = false;         // an op longer than 4095 / a mismatch beyond reference offset 65535: not a gk_mate
  bool huge = false;                            // an op of 2^28 or more / a mismatch at offset 2^24 or beyond: not even a wide one
  std::vector<std::string> ins;                 // inserted strings in I-op order (interned at merge time)
  bool clipped = false;
  void reset() { n_ops = n_mm = n_indel = 0; long_op = far_mm = huge = clipped = false; ins.clear(); }
  void op(int kind, long n) {
    if (n_ops < GK_WIDE_CIG) cig[n_ops] = (uint32_t)(((unsigned long)n << 4) | (unsigned)kind);
    long_op = long_op || n > 4095;
    huge = huge || n >= (1l << 28);
    ++n_ops;
  }
  void mismatch(long ref_off, unsigned char base) {
    if (n_mm < GK_WIDE_MM) mm[n_mm] = (uint32_t)(((unsigned long)ref_off << 8) | base);
    far_mm = far_mm || ref_off > 0xFFFF;
    huge = huge || ref_off >= (1l << 24);
    ++n_mm;
  }
};

bool is_acgt(const MdTok& t) { return t.kind == 1 && (t.ch == 'A' || t.ch == 'C' || t.ch == 'G' || t.ch == 'T'); }

// CIGAR and SEQ of a SAM line.  The ops are what re.findall(r"(\d+)(\w)") yields: digit runs followed by one
// word character; anything else is skipped.
struct TextSource {
  sv seq;
  std::vector<std::pair<char, long>>& ops;
  TextSource(sv cigar, sv seq_, std::vector<std::pair<char, long>>& scratch) : seq(seq_), ops(scratch) {
    ops.clear();
    for (size_t i = 0; i < cigar.size();) {
      if (!isdigit((unsigned char)cigar[i])) { ++i; continue; }
      long n = 0;
      size_t j = i;
      while (j < cigar.size() && isdigit((unsigned char)cigar[j])) n = push_digit(n, cigar[j++]);
      if (j >= cigar.size() || !(isalnum((unsigned char)cigar[j]) || cigar[j] == '_')) {
        // regex backtracking: with no word character after the digits, "\d+" gives its last digit to "\w"
        if (j - i < 2) { i = j; continue; }
        ops.push_back({cigar[j - 1], n / 10});
        i = j;
      } else {
        ops.push_back({cigar[j], n});
        i = j + 1;
      }
    }
  }
  size_t n_ops() const { return ops.size(); }
  char op(size_t i, long& n) const { n = ops[i].second; return ops[i].first; }
  long seq_size() const { return (long)seq.size(); }
  unsigned char base(long i) const { return (unsigned char)seq[(size_t)i]; }
  std::string bases(long a, long n) const { return std::string(seq.substr((size_t)a, (size_t)n)); }
};

// CIGAR and SEQ as they lie in a BAM record (see GkAlnRecord): no text is made of them
struct BamSource {
  const uint8_t *cig, *seq4;
  uint32_t n_cig, l_seq;
  static constexpr const char* kBase = "=ACMGRSVTWYHKDBN";
  size_t n_ops() const { return n_cig; }
  char op(size_t i, long& n) const {
    const uint8_t* p = cig + 4 * i;
    const uint32_t v = (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24;
    n = (long)(v >> 4);
    return "MIDNS"[v & 15u];
  }
  long seq_size() const { return (long)l_seq; }
  unsigned char base(long i) const { return (unsigned char)kBase[(seq4[i >> 1] >> ((~i & 1) << 2)) & 15u]; }
  std::string bases(long a, long n) const {
    std::string s((size_t)n, ' ');
    for (long i = 0; i < n; ++i) s[(size_t)i] = (char)base(a + i);
    return s;
  }
};

// CIGAR / MD / Zs co-walk with the reference's consumption checks (see packed.py::_walkText)
struct ZsEntry { long gap; char kind; };

// what one decoding thread reuses from mate to mate (no allocation per mate, no thread-local look-ups)
struct Scratch {
  std::vector<MdTok> md;
  std::vector<ZsEntry> zs;
  Walked w[2];                       // left and right mate of the pair being decoded
  int side = 0;                      // the one being walked
  std::vector<std::pair<char, long>> text_ops;
  std::string seen_ref;              // neighbouring records are mostly of one backbone: the last name looked up
  int seen_id = -1;
  GkAlnRecord pr[2];                 // their text buffers keep their capacity from pair to pair
};

template <typename Source>
bool walk_alignment(const Source& src, bool has_md, sv md_s, bool has_zs, sv zs_s, Scratch& sc, Fail& f) {
  std::vector<MdTok>& md = sc.md;
  std::vector<ZsEntry>& zs = sc.zs;
  Walked& w = sc.w[sc.side];
  md.clear();
  if (has_md) {
    for (size_t i = 0; i < md_s.size();) {
      if (isdigit((unsigned char)md_s[i])) {
        long x = 0;
        while (i < md_s.size() && isdigit((unsigned char)md_s[i])) x = push_digit(x, md_s[i++]);
        md.push_back({0, x, 0});
      } else {
        md.push_back({1, 0, md_s[i++]});
      }
    }
  }
  zs.clear();
  if (has_zs && !zs_s.empty()) {
    size_t a = 0;
    for (;;) {
      size_t c = zs_s.find(',', a);
      sv ent = zs_s.substr(a, c == sv::npos ? sv::npos : c - a);
      size_t p1 = ent.find('|');
      size_t p2 = p1 == sv::npos ? sv::npos : ent.find('|', p1 + 1);
      long gap = 0;
      if (p1 == sv::npos || p2 == sv::npos || !to_long(ent.substr(0, p1), gap)) {
        f = {4, "malformed Zs entry"};
        return false;
      }
      sv kind = ent.substr(p1 + 1, p2 - p1 - 1);
      zs.push_back({gap, kind.size() == 1 ? kind[0] : '?'});
      if (c == sv::npos) break;
      a = c + 1;
    }
  }
  const long seq_size = src.seq_size();
  long ref = 0, ri = 0, owed = 0, zpos = 0;
  size_t mi = 0, zi = 0;
  auto take_zs = [&](char kind) {
    if (zi < zs.size() && zs[zi].kind == kind && ri + owed == zpos + zs[zi].gap) {
      zpos += zs[zi].gap + (kind == 'S' ? 1 : 0);
      ++zi;
    }
  };
  auto md_is_zero = [&](size_t i) { return i < md.size() && md[i].kind == 0 && md[i].num == 0; };
  const size_t n_src = src.n_ops();
  for (size_t k = 0; k < n_src; ++k) {
    long n;
    const char op = src.op(k, n);
    if (md_is_zero(mi)) ++mi;
    if (op == 'M') {
      w.op(GK_CIG_M, n);
      long done = 0;
      for (;;) {
        if (owed <= done && mi < md.size() && md[mi].kind == 0) { owed += md[mi].num; ++mi; }
        if (owed >= n) { owed -= n; break; }
        if (ri + owed >= seq_size || mi >= md.size()) { f = {1, "MD / SEQ exhausted inside an M op"}; return false; }
        const unsigned char base = src.base(ri + owed);
        if (md_is_zero(mi)) ++mi;
        if (mi >= md.size()) { f = {1, "MD exhausted inside an M op"}; return false; }
        if (!is_acgt(md[mi])) { f = {1, "MD mismatch token is not a base"}; return false; }
        if ((unsigned char)md[mi].ch == base) { f = {1, "MD reference base equals the read base"}; return false; }
        ++mi;
        take_zs('S');
        w.mismatch(ref + owed, base);
        owed += 1;
        done = owed;
        if (owed == n) { owed = 0; break; }
      }
      ref += n;
      ri += n;
    } else if (op == 'I') {
      w.op(GK_CIG_I, n);
      ++w.n_indel;
      take_zs('I');
      const long at = std::min<long>(ri, seq_size);
      w.ins.emplace_back(src.bases(at, std::max<long>(0, std::min<long>(n, seq_size - ri))));
      ri += n;
    } else if (op == 'D') {
      w.op(GK_CIG_D, n);
      ++w.n_indel;
      if (mi >= md.size() || !(md[mi].kind == 1 && md[mi].ch == '^')) { f = {1, "MD has no deletion at a D op"}; return false; }
      ++mi;
      while (mi < md.size() && is_acgt(md[mi])) ++mi;
      take_zs('D');
      ref += n;
    } else if (op == 'S') {
      w.clipped = true;
      zpos += n;
      ri += n;
    } else if (op == 'N') {
      f = {2, "Cannot typing with splicing"};
      return false;
    } else {
      f = {2, "unsupported CIGAR operation"};
      return false;
    }
  }
  if (md_is_zero(mi)) ++mi;
  if (zi != zs.size()) { f = {1, "Zs entries do not line up with the alignment"}; return false; }
  if (mi != md.size()) { f = {1, "MD not fully consumed"}; return false; }
  if (ri != seq_size) { f = {1, "CIGAR does not cover the read"}; return false; }
  return true;
}

// SAM text line -> record fields (views into the line)
bool parse_record(sv line, GkAlnRecord& p, Fail& f) {
  std::vector<sv> cols;
  split_tabs(strip(line), cols);
  if (cols.size() < 11 || !to_long(cols[1], p.flag) || !to_long(cols[3], p.pos)) {
    f = {4, "malformed SAM record"};
    return false;
  }
  p.ref = cols[2]; p.cigar = cols[5]; p.seq = cols[9];
  for (size_t i = 11; i < cols.size(); ++i) {
    sv c = cols[i];
    if (c.substr(0, 2) == "NM") {
      long v;
      if (!to_long(c.size() >= 5 ? c.substr(5) : sv(), v)) { f = {4, "malformed NM tag"}; return false; }
      p.has_nm = true; p.nm = v;
    } else if (!p.has_md && c.substr(0, 2) == "MD") {
      p.has_md = true; p.md = c.size() >= 5 ? c.substr(5) : sv();
    } else if (!p.has_zs && c.substr(0, 2) == "Zs") {
      p.has_zs = true; p.zs = c.size() >= 5 ? c.substr(5) : sv();
    }
  }
  // getNH: first "NH:i:<digits>" anywhere in the line
  p.nh = 1;
  size_t at = line.find("NH:i:");
  while (at != sv::npos) {
    size_t d = at + 5;
    if (d < line.size() && isdigit((unsigned char)line[d])) {
      long x = 0;
      while (d < line.size() && isdigit((unsigned char)line[d])) x = push_digit(x, line[d++]);
      p.nh = x;
      break;
    }
    at = line.find("NH:i:", at + 1);
  }
  return true;
}

bool passes(const GkAlnRecord& p) { return (p.flag & 2) && p.has_nm && p.nm <= 4; }

bool fail(gk_packer* pk, const Fail& f, int64_t line_index) {
  pk->err_kind = f.kind;
  pk->err_line = line_index;
  pk->err_msg = f.msg;
  return false;
}

// one emitted pair: left = the later line, right = the earlier one (readPair yields (line, next_line)).
// Pure function of the two lines and the gene table: safe to run for many pairs at once.
struct Outcome {
  std::vector<std::string> ins[2];   // inserted strings met by the walk of each mate, in order
  bool store_ins[2] = {false, false};   // ... and whether the record keeps their ids (not for clipped mates)
  bool spilled = false;              // the pair does not fit two gk_mate records: it is in `wide`
  size_t spill_slot = 0;             // ... which the decoding thread filed under this number of its own list
  Fail fail{0, ""};
  int64_t fail_line = -1;
};

struct Decoded : Outcome {
  gk_mate* rec = nullptr;            // the pair's two records, in their final place
  gk_mate_wide wide[2];              // ... and in the wide format when `spilled`
  Scratch sc;
};

void decode_records(const gk_packer* pk, const GkAlnRecord* pr, const int64_t* idx, Decoded& out);

void decode_pair(const gk_packer* pk, sv left, int64_t left_idx, sv right, int64_t right_idx, Decoded& out) {
  GkAlnRecord pr[2];
  Fail f{0, ""};
  sv lines[2] = {left, right};
  int64_t idx[2] = {left_idx, right_idx};
  memset(out.rec, 0, 2 * sizeof(gk_mate));
  for (int s = 0; s < 2; ++s)
    if (!parse_record(lines[s], pr[s], f)) { out.fail = f; out.fail_line = idx[s]; return; }
  decode_records(pk, pr, idx, out);
}

// the pair (left, right) as record fields -> two gk_mate records + the strings to intern
void decode_records(const gk_packer* pk, const GkAlnRecord* pr, const int64_t* idx, Decoded& out) {
  Fail f{0, ""};
  auto failed = [&](const Fail& why, int64_t line) { out.fail = why; out.fail_line = line; };
  memset(out.rec, 0, 2 * sizeof(gk_mate));
  const bool both = passes(pr[0]) && passes(pr[1]);
  for (int s = 0; s < 2; ++s) {
    const GkAlnRecord& p = pr[s];
    gk_mate& r = out.rec[s];
    Scratch& sc = out.sc;
    int gene = p.gene;
    if (gene == GkAlnRecord::kGeneByName) {
      if (sc.seen_id < 0 || sv(sc.seen_ref) != p.ref) {
        auto g = pk->gene_id.find(std::string(p.ref));
        if (g == pk->gene_id.end()) return failed({4, "reference is not a backbone of the index"}, idx[s]);
        sc.seen_ref.assign(p.ref); sc.seen_id = (int)g->second;
      }
      gene = sc.seen_id;
    }
    if (gene < 0) return failed({4, "reference is not a backbone of the index"}, idx[s]);
    const int seen_id = gene;
    r.pos0 = (uint32_t)(p.pos - 1);
    r.flag = (uint16_t)(p.flag & 0xFFFF);
    r.ref = (uint8_t)seen_id;
    r.nh = (uint8_t)std::min<long>(p.nh, 255);
    r.nm = p.has_nm ? (uint8_t)std::min<long>(std::max<long>(p.nm, 0), 254) : (uint8_t)GK_NM_ABSENT;
    if (!both) continue;
    sc.side = s;
    Walked& w = sc.w[s];
    w.reset();
    const bool binary = p.bam_cigar != nullptr;
    const bool walked =
        binary ? walk_alignment(BamSource{p.bam_cigar, p.bam_seq, p.n_bam_cigar, p.l_bam_seq}, p.has_md, p.md, p.has_zs, p.zs, sc, f)
               : walk_alignment(TextSource(p.cigar, p.seq, sc.text_ops), p.has_md, p.md, p.has_zs, p.zs, sc, f);
    out.ins[s].swap(w.ins);          // strings met before a failure are interned too, like a one-by-one walk
    w.ins.clear();                   // (swapped, not moved: both lists keep their storage from pair to pair)
    if (!walked) return failed(f, idx[s]);
    if (w.clipped) {   // its strings are still interned at merge time, the record keeps none of them
      // keep the CIGAR (S ops included) for read depth when it fits, else only the clip marker
      w.n_ops = 0;
      w.long_op = w.huge = false;
      auto add = [&](char op, long n) {
        w.op(op == 'S' ? GK_CIG_S : op == 'M' ? GK_CIG_M : op == 'I' ? GK_CIG_I : GK_CIG_D, n);
      };
      if (binary) {
        const BamSource src{p.bam_cigar, p.bam_seq, p.n_bam_cigar, p.l_bam_seq};
        for (size_t i = 0; i < src.n_ops(); ++i) {
          long n;
          const char op = src.op(i, n);
          add(op, n);
        }
      } else {
        sv cg = p.cigar;
        for (size_t i = 0; i < cg.size();) {
          if (!isdigit((unsigned char)cg[i])) { ++i; continue; }
          long n = 0;
          while (i < cg.size() && isdigit((unsigned char)cg[i])) n = push_digit(n, cg[i++]);
          if (i >= cg.size()) break;
          add(cg[i++], n);
        }
      }
      const bool narrow_ok = w.n_ops <= GK_MAX_CIG && !w.long_op;
      const bool wide_ok = w.n_ops <= GK_WIDE_CIG && !w.huge;
      if (!narrow_ok && wide_ok) out.spilled = true;                                        // its M runs count for the depth
      if (!narrow_ok && !wide_ok) { w.n_ops = 1; w.cig[0] = (uint32_t)GK_CIG_S; }           // {S, 0}: in neither format
      if (narrow_ok || !wide_ok) {
        r.n_cig = (uint8_t)w.n_ops;
        for (size_t i = 0; i < w.n_ops; ++i) r.cig[i] = (uint16_t)w.cig[i];
      }
      w.n_mm = 0;
      continue;
    }
    const size_t n_ins = out.ins[s].size();
    const bool fits = w.n_ops <= GK_MAX_CIG && w.n_mm <= GK_MAX_MM && n_ins <= GK_MAX_INS &&
                      w.n_mm + w.n_indel <= GK_MAX_EVENTS && !w.long_op && !w.far_mm;
    out.store_ins[s] = true;
    if (!fits) {
      // the second format (gk_mate_wide) takes the pair; beyond that one too the mate cannot be represented
      const bool wide_fits = w.n_ops <= GK_WIDE_CIG && w.n_mm <= GK_WIDE_MM && n_ins <= GK_WIDE_INS &&
                             w.n_mm + w.n_indel <= GK_WIDE_EVENTS && !w.huge;
      if (!wide_fits) return failed({3, "record does not fit gk_mate_wide"}, idx[s]);
      out.spilled = true;
      continue;
    }
    r.n_cig = (uint8_t)w.n_ops; r.n_mm = (uint8_t)w.n_mm; r.n_ins = (uint8_t)n_ins;
    for (size_t i = 0; i < w.n_ops; ++i) r.cig[i] = (uint16_t)w.cig[i];
    for (size_t i = 0; i < w.n_mm; ++i) { r.mm[i].ref_off = (uint16_t)(w.mm[i] >> 8); r.mm[i].base = (uint8_t)(w.mm[i] & 0xFFu); }
  }
  if (!out.spilled) return;
  // both mates move to the wide array; their gk_mate records keep the header and point there (the index of the
  // pair in that array is known once the pairs of all threads are in order: decode_all fills ins[0])
  for (int s = 0; s < 2; ++s) {
    gk_mate& r = out.rec[s];
    const Walked& w = out.sc.w[s];
    gk_mate_wide& x = out.wide[s];
    memset(&x, 0, sizeof(x));
    x.pos0 = r.pos0; x.flag = r.flag; x.ref = r.ref; x.nh = r.nh; x.nm = r.nm;
    x.n_cig = (uint16_t)w.n_ops;
    x.n_mm = (uint16_t)w.n_mm;
    x.n_ins = (uint16_t)(out.store_ins[s] ? out.ins[s].size() : 0);
    for (size_t i = 0; i < w.n_ops; ++i) x.cig[i] = w.cig[i];
    for (size_t i = 0; i < w.n_mm; ++i) x.mm[i] = w.mm[i];
    gk_mate header;
    memset(&header, 0, sizeof(header));
    header.pos0 = r.pos0; header.flag = r.flag; header.ref = r.ref; header.nh = r.nh; header.nm = r.nm;
    header.n_cig = GK_SPILLED;
    r = header;
  }
}

int pack_threads() { return gk_ingest_threads(); }

// step (3): in emission order, intern the inserted strings, stop at the first failed pair
template <typename Work>
void on_threads(size_t n, const Work& work) {
  const int n_thr = (int)std::min<size_t>((size_t)pack_threads(), (n + 255) / 256);
  if (n_thr <= 1) {
    work(0, 0, n);
    return;
  }
  std::vector<std::thread> pool;
  for (int t = 0; t < n_thr; ++t) pool.emplace_back(work, t, n * t / n_thr, n * (t + 1) / n_thr);
  for (auto& th : pool) th.join();
}

// Steps (2) and (3) for n emitted pairs: decode(i, out) fills the records of pair i; lines(i, s) is the
// stream index of its left (s = 0) / right (s = 1) record.  The records are written straight into their
// final place by the decoding threads.  What depends on the order of the pairs -- ids of inserted
// strings (first seen, first numbered) and "stop at the first failing pair" -- is settled afterwards
// in one pass over the few pairs that met an inserted string or failed.
struct Special {
  size_t pair;
  Outcome d;
};

template <typename DecodeOne, typename LineOf>
bool decode_all(gk_packer* pk, size_t n, const DecodeOne& decode, const LineOf& line_of) {
  if (!n) return true;
  const size_t first = pk->n_mates / 2;
  if (!pk->resize_mates(2 * (first + n)))
    return fail(pk, Fail{3, "more records than the output buffer of gk_packer_set_output holds"}, -1);
  gk_mate* const mates = pk->mates();
  pk->pair_lines.resize(2 * (first + n));
  std::vector<std::vector<Special>> special((size_t)pack_threads() + 1);
  struct Spill { size_t pair; gk_mate_wide w[2]; };
  std::vector<std::vector<Spill>> spills((size_t)pack_threads() + 1);   // per thread, ascending pairs
  on_threads(n, [&](int t, size_t a, size_t b) {
    std::unique_ptr<Decoded> holder(new Decoded());   // two wide records inside: not for the stack
    Decoded& d = *holder;
    for (size_t i = a; i < b; ++i) {
      d.ins[0].clear(); d.ins[1].clear();
      d.store_ins[0] = d.store_ins[1] = false;
      d.spilled = false;
      d.fail.kind = 0;
      d.fail.msg.clear();
      d.fail_line = -1;
      d.rec = mates + 2 * (first + i);
      decode(i, d);
      pk->pair_lines[2 * (first + i)] = line_of(i, 0);
      pk->pair_lines[2 * (first + i) + 1] = line_of(i, 1);
      if (d.fail.kind) d.spilled = false;
      if (d.spilled) {
        d.spill_slot = spills[(size_t)t].size();
        spills[(size_t)t].push_back(Spill{i, {d.wide[0], d.wide[1]}});
      }
      if (!d.fail.kind && d.ins[0].empty() && d.ins[1].empty()) continue;
      // A string that is in the table already (an index string, or one met in an earlier feed) has its final id:
      // the table is not written to while the threads decode.  Only pairs with a string met for the first time,
      // or with a failure, wait for the ordered pass.
      bool settled = !d.fail.kind;
      for (int s = 0; s < 2 && settled; ++s)
        for (size_t q = 0; q < d.ins[s].size(); ++q) {
          auto it = pk->ins_id.find(d.ins[s][q]);
          if (it == pk->ins_id.end()) { settled = false; break; }
          if (!d.store_ins[s]) continue;
          if (d.spilled) {
            if (q < GK_WIDE_INS) spills[(size_t)t][d.spill_slot].w[s].ins[q] = it->second;
          } else if (q < GK_MAX_INS) {
            d.rec[s].ins[q] = it->second;
          }
        }
      if (!settled) special[(size_t)t].push_back(Special{i, static_cast<const Outcome&>(d)});
    }
  });
  // the wide pairs of all threads, in pair order, join the packer's wide array; their gk_mate records learn where
  auto file_spills = [&](size_t n_kept) {
    for (auto& list : spills)
      for (Spill& sp : list) {
        if (sp.pair >= n_kept) continue;
        const uint32_t at = (uint32_t)(pk->wide.size() / 2);
        pk->wide.push_back(sp.w[0]);
        pk->wide.push_back(sp.w[1]);
        pk->spill_pair.push_back((int64_t)(first + sp.pair));
        mates[2 * (first + sp.pair)].ins[0] = at;
        mates[2 * (first + sp.pair) + 1].ins[0] = at;
      }
  };
  for (size_t t = 0; t < special.size(); ++t) {   // threads own ascending ranges: this walks the pairs in order
    for (Special& sp : special[t]) {
      gk_mate* dst = mates + 2 * (first + sp.pair);
      for (int s = 0; s < 2; ++s) {
        for (size_t q = 0; q < sp.d.ins[s].size(); ++q) {
          auto it = pk->ins_id.find(sp.d.ins[s][q]);
          uint32_t id;
          if (it == pk->ins_id.end()) {
            id = (uint32_t)pk->ins_strings.size();
            pk->ins_id.emplace(sp.d.ins[s][q], id);
            pk->ins_strings.push_back(sp.d.ins[s][q]);
          } else {
            id = it->second;
          }
          if (!sp.d.store_ins[s]) continue;
          if (sp.d.spilled) {
            if (q < GK_WIDE_INS) spills[t][sp.d.spill_slot].w[s].ins[q] = id;
          } else if (q < GK_MAX_INS) {
            dst[s].ins[q] = id;
          }
        }
      }
      if (sp.d.fail.kind) {   // the pairs before it stay, like a one-by-one walk
        file_spills(sp.pair);
        pk->resize_mates(2 * (first + sp.pair));
        pk->pair_lines.resize(2 * (first + sp.pair));
        return fail(pk, sp.d.fail, sp.d.fail_line);
      }
    }
  }
  file_spills(n);
  return true;
}

// the pairs queued by the pairing pass of the text reader
bool run_jobs(gk_packer* pk) {
  const bool ok = decode_all(
      pk, pk->jobs.size(),
      [&](size_t i, Decoded& out) {
        const gk_packer::Job& j = pk->jobs[i];
        decode_pair(pk, j.left, j.left_idx, j.right_owned.empty() ? j.right : sv(j.right_owned), j.right_idx, out);
      },
      [&](size_t i, int s) { return s ? pk->jobs[i].right_idx : pk->jobs[i].left_idx; });
  pk->jobs.clear();
  return ok;
}

bool feed_line(gk_packer* pk, sv line, int64_t index) {
  if (line.empty() || line[0] == '@' || line.substr(0, 15) == "[bam_sort_core]") return true;
  // the first 8 tab fields (readPair: qname, flag, ref, pos, _, _, rnext, pnext)
  sv f[8];
  size_t a = 0;
  for (int k = 0; k < 8; ++k) {
    size_t t = line.find('\t', a);
    if (t == sv::npos) {
      if (k < 7) return fail(pk, {4, "SAM line has fewer than 8 fields"}, index);
      f[k] = line.substr(a);
      a = line.size();
    } else {
      f[k] = line.substr(a, t - a);
      a = t + 1;
    }
  }
  if (f[6] != "=") return true;
  pk->n_reads += 1;
  long flag;
  if (!to_long(f[1], flag)) return fail(pk, {4, "malformed FLAG"}, index);
  const char sec = (flag & 256) ? '1' : '0';
  static thread_local std::string kbuf;
  auto key = [&](sv pos) -> std::string& {
    kbuf.clear();
    kbuf.append(f[0]); kbuf.push_back('\t'); kbuf.append(f[2]); kbuf.push_back('\t'); kbuf.append(pos); kbuf.push_back('\t'); kbuf.push_back(sec);
    return kbuf;
  };
  auto it = pk->waiting.find(key(f[7]));
  if (it == pk->waiting.end()) {
    pk->waiting[key(f[3])] = Pending{line, std::string(), index, (int)flag};
    return true;
  }
  if (((it->second.flag | flag) & 192) != 192) {   // READ1 and READ2 must both be present
    pk->n_strange += 1;
    return true;
  }
  Pending mate = std::move(it->second);
  pk->waiting.erase(it);
  pk->n_pairs += 1;
  pk->jobs.push_back({line, mate.view, std::move(mate.owned), index, mate.index});
  return true;
}

}  // namespace

// Records of another source (BAM) through the same pairing rule, decoder and merge as SAM text.
int gk_packer_gene_of(const gk_packer* pk, std::string_view ref) {
  auto g = pk->gene_id.find(std::string(ref));
  return g == pk->gene_id.end() ? -1 : (int)g->second;
}

int gk_packer_feed_records(gk_packer* pk, int64_t n, bool names_contiguous,
                           const std::function<void(int64_t, GkAlnKey&)>& key,
                           const std::function<void(int64_t, GkAlnRecord&)>& full,
                           const std::function<void(int64_t, bool)>& soon) {
  if (!pk || n < 0) { gk_set_error("bad packer arguments"); return GK_ERR_ARG; }
  if (pk->err_kind) return GK_ERR_ASSERT;
  if (!pk->waiting.empty() || !pk->carry.empty()) { gk_set_error("text and record input cannot be mixed"); return GK_ERR_ARG; }
  const int64_t base = pk->n_lines;
  pk->n_lines += n;
  GkPhaseClock clock("feed_records");
  // (1) pairing in stream order (hisat2.py:248-270).  Only records with the same name can pair, so
  // when equal names are contiguous (a name-collated stream) the stream is cut at name changes and
  // the pieces are paired independently; their emission lists, concatenated, are the sequential one.
  struct Wait { int64_t index; long flag; };
  struct Piece { std::vector<int64_t, GkRawInit<int64_t>> pairs; int64_t n_reads = 0, n_strange = 0; };
  auto pair_range = [&](int64_t a, int64_t b, Piece& out) {
    out.pairs.reserve((size_t)(b - a));          // a pair takes two records: the list never grows past this (no reallocation)
    std::unordered_map<std::string, Wait> waiting;
    std::string kbuf;
    GkAlnKey k;
    auto make = [&](long pos) -> std::string& {
      kbuf.assign(k.name); kbuf.push_back('\t'); kbuf.append(k.ref); kbuf.push_back('\t');
      kbuf.append(std::to_string(pos)); kbuf.push_back('\t'); kbuf.push_back((k.flag & 256) ? '1' : '0');
      return kbuf;
    };
    auto general = [&](int64_t i) {   // k holds record i
      if (!k.mate_same_ref) return;
      out.n_reads += 1;
      auto it = waiting.find(make(k.next_pos));
      if (it == waiting.end()) {
        waiting[make(k.pos)] = Wait{i, k.flag};
        return;
      }
      if (((it->second.flag | k.flag) & 192) != 192) { out.n_strange += 1; return; }
      out.pairs.push_back(i);                    // left = the later record
      out.pairs.push_back(it->second.index);     // right = the earlier one
      waiting.erase(it);
    };
    if (!names_contiguous) {
      for (int64_t i = a; i < b; ++i) { key(i, k); general(i); }
      return;
    }
    // Only records of one name can pair and they are adjacent: the table is per name, and the usual
    // group -- exactly two records -- is settled by comparing the two keys directly.
    // Every record is keyed once: the key that ended a group starts the next.
    GkAlnKey first, second, third;
    auto single = [&](const GkAlnKey& only) { if (only.mate_same_ref) out.n_reads += 1; };   // waits for a mate that never comes
    int64_t i = a;
    if (i < b) key(i, first);
    while (i < b) {
      if (soon && i + 32 < b) soon(i + 32, true);
      if (i + 1 == b) { single(first); break; }
      key(i + 1, second);
      if (second.name != first.name) { single(first); first = second; i += 1; continue; }
      bool more = false;
      if (i + 2 < b) { key(i + 2, third); more = third.name == first.name; }
      if (!more) {
        if (first.mate_same_ref) out.n_reads += 1;
        if (second.mate_same_ref) out.n_reads += 1;
        // the second record finds the waiting first one iff (ref, its mate position, secondary flag) agree
        if (first.mate_same_ref && second.mate_same_ref && second.ref == first.ref && second.next_pos == first.pos &&
            ((second.flag ^ first.flag) & 256) == 0) {
          if (((first.flag | second.flag) & 192) != 192) {
            out.n_strange += 1;
          } else {
            out.pairs.push_back(i + 1);
            out.pairs.push_back(i);
          }
        }
        i += 2;
        if (i < b) first = third;
        continue;
      }
      int64_t e = i + 3;   // three or more records of one name: the general rule over the group
      bool ended = false;
      while (e < b) {
        key(e, second);
        if (second.name != first.name) { ended = true; break; }
        ++e;
      }
      waiting.clear();
      for (int64_t j = i; j < e; ++j) { key(j, k); general(j); }
      i = e;
      if (ended) first = second;
    }
  };
  std::vector<int64_t> cuts{0};
  if (names_contiguous) {
    const int want = (int)std::min<int64_t>(pack_threads(), std::max<int64_t>(n / 4096, 1));
    for (int t = 1; t < want; ++t) {
      int64_t c = std::max(cuts.back(), n * t / want);
      GkAlnKey prev, cur;
      while (c > cuts.back() && c < n) {
        key(c - 1, prev); key(c, cur);
        if (prev.name != cur.name) break;
        ++c;
      }
      if (c > cuts.back() && c < n) cuts.push_back(c);
    }
  }
  cuts.push_back(n);
  std::vector<Piece> pieces(cuts.size() - 1);
  if (pieces.size() == 1) {
    pair_range(0, n, pieces[0]);
  } else {
    std::vector<std::thread> pool;
    for (size_t t = 0; t < pieces.size(); ++t) pool.emplace_back([&, t] { pair_range(cuts[t], cuts[t + 1], pieces[t]); });
    for (auto& th : pool) th.join();
  }
  // left (later) and right (earlier) record of every emitted pair: the lists of the pieces one after the other, each
  // copied by a thread of its own into a block that nobody zero-fills first (8 MB per million records, first touched here)
  std::vector<int64_t, GkRawInit<int64_t>> pairs;
  {
    std::vector<size_t> at(pieces.size() + 1, 0);
    for (size_t t = 0; t < pieces.size(); ++t) {
      const Piece& pc = pieces[t];
      at[t + 1] = at[t] + pc.pairs.size();
      pk->n_reads += pc.n_reads;
      pk->n_strange += pc.n_strange;
      pk->n_pairs += (int64_t)pc.pairs.size() / 2;
    }
    auto place = [&](size_t t) {
      if (!pieces[t].pairs.empty()) memcpy(pairs.data() + at[t], pieces[t].pairs.data(), pieces[t].pairs.size() * sizeof(int64_t));
    };
    if (pieces.size() == 1) {
      pairs.swap(pieces[0].pairs);                // one piece: its list is the list
    } else {
      pairs.resize(at.back());
      std::vector<std::thread> pool;
      if (pairs.size() >= (size_t)1 << 16)
        for (size_t t = 1; t < pieces.size(); ++t) pool.emplace_back(place, t);
      for (size_t t = pool.empty() ? 1 : pieces.size(); t < pieces.size(); ++t) place(t);
      place(0);
      for (auto& th : pool) th.join();
    }
  }
  clock.lap("pairing");
  constexpr size_t kAhead = 6;
  // (2) decode on threads, (3) ordered pass over the pairs with inserted strings
  const bool merged = decode_all(
      pk, pairs.size() / 2,
      [&](size_t i, Decoded& out) {
        GkAlnRecord* pr = out.sc.pr;
        if (soon && 2 * (i + kAhead) + 1 < pairs.size()) {   // the records are far apart in memory: ask for them early
          soon(pairs[2 * (i + kAhead)], false);
          soon(pairs[2 * (i + kAhead) + 1], false);
        }
        const int64_t idx[2] = {base + pairs[2 * i], base + pairs[2 * i + 1]};
        full(pairs[2 * i], pr[0]);
        full(pairs[2 * i + 1], pr[1]);
        decode_records(pk, pr, idx, out);
      },
      [&](size_t i, int s) { return base + pairs[2 * i + s]; });
  clock.lap("decode + merge");
  if (!merged) {
    gk_set_error("alignment record %lld: %s", (long long)pk->err_line, pk->err_msg.c_str());
    return GK_ERR_ASSERT;
  }
  return GK_OK;
}

// helpers of gk_mates_compact_size / gk_mates_compact_host (below)
namespace {
inline int mate_words(const uint32_t* w, int* n_cw, int* n_mm, int* n_ins) {
  const uint32_t h = w[2];
  const int n_cig = (int)((h >> 8) & 0xFFu);
  const bool spilled = n_cig == GK_SPILLED;       // header + ins[0] = the pair's place in the wide array
  *n_cw = spilled ? 0 : (std::min(n_cig, GK_MAX_CIG) + 1) / 2;
  *n_mm = spilled ? 0 : std::min((int)((h >> 16) & 0xFFu), GK_MAX_MM);
  *n_ins = spilled ? 1 : std::min((int)(h >> 24), GK_MAX_INS);
  return 3 + *n_cw + *n_mm + *n_ins;
}
constexpr int kCigWord = 3, kMmWord = kCigWord + GK_MAX_CIG / 2, kInsWord = kMmWord + GK_MAX_MM;
static_assert(sizeof(gk_mate) == 128 && kInsWord + GK_MAX_INS == 32, "gk_mate layout");

template <typename F>
void over_ranges(int64_t n, int n_threads, F&& f) {
  n_threads = (int)std::max<int64_t>(1, std::min<int64_t>(n_threads, n / 65536 + 1));
  if (n_threads == 1) { f(0, (int64_t)0, n); return; }
  std::vector<std::thread> pool;
  for (int t = 0; t < n_threads; ++t)
    pool.emplace_back([&, t] { f(t, n * t / n_threads, n * (t + 1) / n_threads); });
  for (auto& th : pool) th.join();
}
}  // namespace

extern "C" {

int gk_packer_create(const char* const* gene_names, int32_t n_genes, const char* const* ins_strings, int32_t n_ins,
                     gk_packer** out) {
  if (!out || n_genes < 0 || n_ins < 0) { gk_set_error("bad packer arguments"); return GK_ERR_ARG; }
  gk_packer* pk = new gk_packer();
  for (int i = 0; i < n_genes; ++i) { pk->genes.emplace_back(gene_names[i]); pk->gene_id[gene_names[i]] = i; }
  for (int i = 0; i < n_ins; ++i) { pk->ins_strings.emplace_back(ins_strings[i]); pk->ins_id[ins_strings[i]] = (uint32_t)i; }
  pk->n_index_ins = (uint32_t)n_ins;
  *out = pk;
  return GK_OK;
}

int gk_packer_destroy(gk_packer* pk) {
  delete pk;
  return GK_OK;
}

// Feed a chunk of SAM text (lines may straddle chunks; pass final != 0 with the last chunk).
// Returns GK_OK, or GK_ERR_ASSERT / GK_ERR_ARG with the details available from gk_packer_error.
static int packer_feed_impl(gk_packer* pk, const char* text, size_t n_bytes, int32_t final) {
  if (!pk || (!text && n_bytes)) { gk_set_error("bad packer arguments"); return GK_ERR_ARG; }
  if (pk->err_kind) return GK_ERR_ASSERT;
  std::string owned;
  sv data;
  if (!pk->carry.empty()) {
    owned = std::move(pk->carry);
    pk->carry.clear();
    owned.append(text, n_bytes);
    data = owned;
  } else {
    data = sv(text, n_bytes);
  }
  size_t a = 0;
  bool ok = true;
  while (a < data.size() && ok) {
    size_t nl = data.find('\n', a);
    if (nl == sv::npos) {
      if (!final) { pk->carry.assign(data.substr(a)); break; }
      nl = data.size();
    }
    sv line = data.substr(a, nl - a);
    if (!line.empty() && line.back() == '\r') line.remove_suffix(1);
    const int64_t index = pk->n_lines++;
    ok = feed_line(pk, line, index);    // pairing only: emitted pairs are queued
    a = nl + 1;
  }
  // decode the queued pairs (their left lines point into `data`, so before this call returns).  A
  // pairing error comes from a later line than every queued pair: the queued pairs still decide
  // whether an earlier one fails first.
  const Fail pairing{pk->err_kind, pk->err_msg};
  const int64_t pairing_line = pk->err_line;
  if (!ok) { pk->err_kind = 0; pk->err_line = -1; pk->err_msg.clear(); }
  bool decoded = run_jobs(pk);
  for (auto& kv : pk->waiting)      // records still waiting outlive this chunk: keep their text
    if (kv.second.owned.empty()) { kv.second.owned.assign(kv.second.view); kv.second.view = sv(); }
  if (decoded && !ok) fail(pk, pairing, pairing_line);
  if (!decoded || !ok) {
    gk_set_error("SAM line %lld: %s", (long long)pk->err_line, pk->err_msg.c_str());
    return GK_ERR_ASSERT;
  }
  return GK_OK;
}

int gk_packer_feed(gk_packer* pk, const char* text, size_t n_bytes, int32_t final) {
  try {
    return packer_feed_impl(pk, text, n_bytes, final);
  } catch (const std::bad_alloc&) {      // an exception must not cross the C boundary
    gk_set_error("gk_packer_feed: out of host memory");
    return GK_ERR_CAPACITY;
  }
}

int gk_packer_counts(gk_packer* pk, int64_t* n_lines, int64_t* n_reads, int64_t* n_pairs, int64_t* n_strange,
                     int64_t* n_strings) {
  if (!pk) return GK_ERR_ARG;
  if (n_lines) *n_lines = pk->n_lines;
  if (n_reads) *n_reads = pk->n_reads;
  if (n_pairs) *n_pairs = (int64_t)pk->n_mates / 2;
  if (n_strange) *n_strange = pk->n_strange;
  if (n_strings) *n_strings = (int64_t)pk->ins_strings.size();
  return GK_OK;
}

// kind: 0 none, 1 AssertionError, 2 NotImplementedError, 3 record capacity, 4 ValueError
int gk_packer_error(gk_packer* pk, int32_t* kind, int64_t* line_index) {
  if (!pk) return GK_ERR_ARG;
  if (kind) *kind = pk->err_kind;
  if (line_index) *line_index = pk->err_line;
  return GK_OK;
}

// Records go straight into the caller's buffer (e.g. pinned memory, so the upload can start from where the
// decoder wrote) instead of the packer's own storage.  Call before the first feed; capacity in records (2 per
// pair; a BAM file of n alignment records yields at most n).  A feed that would overflow it fails with kind 3.
int gk_packer_set_output(gk_packer* pk, gk_mate* mates_out, int64_t capacity) {
  if (!pk || !mates_out || capacity < 0) { gk_set_error("bad packer arguments"); return GK_ERR_ARG; }
  if (pk->n_mates) { gk_set_error("gk_packer_set_output after records were made"); return GK_ERR_ARG; }
  pk->ext = mates_out;
  pk->ext_cap = capacity;
  return GK_OK;
}

// Copy out the records (2 per pair; nothing to copy when they were written to the buffer of
// gk_packer_set_output, mates_out may then be that buffer or null) and the line indices (left, right) of every pair.
int gk_packer_records(gk_packer* pk, gk_mate* mates_out, int64_t* pair_lines_out) {
  if (!pk) return GK_ERR_ARG;
  if (mates_out && pk->n_mates && mates_out != pk->mates()) {   // first touch of the caller's pages: worth spreading over the threads
    const size_t bytes = pk->n_mates * sizeof(gk_mate), piece = 1u << 22;
    const char* from = (const char*)pk->mates();
    on_threads((bytes + piece - 1) / piece, [&](int, size_t a, size_t b) {
      const size_t lo = a * piece, hi = std::min(bytes, b * piece);
      if (hi > lo) memcpy((char*)mates_out + lo, from + lo, hi - lo);
    });
  }
  if (pair_lines_out && !pk->pair_lines.empty())
    memcpy(pair_lines_out, pk->pair_lines.data(), pk->pair_lines.size() * sizeof(int64_t));
  return GK_OK;
}

int gk_packer_spilled(gk_packer* pk, int64_t* n_spilled_pairs) {
  if (!pk || !n_spilled_pairs) return GK_ERR_ARG;
  *n_spilled_pairs = (int64_t)pk->spill_pair.size();
  return GK_OK;
}

int gk_packer_spill_records(gk_packer* pk, gk_mate_wide* wide_out, int64_t* pair_index_out) {
  if (!pk) return GK_ERR_ARG;
  if (wide_out && !pk->wide.empty()) memcpy(wide_out, pk->wide.data(), pk->wide.size() * sizeof(gk_mate_wide));
  if (pair_index_out && !pk->spill_pair.empty()) memcpy(pair_index_out, pk->spill_pair.data(), pk->spill_pair.size() * sizeof(int64_t));
  return GK_OK;
}

// ---- the compact form of packed records, made on the host (the same layout gk_mates_compact makes on the device): what
// goes over PCIe is ~30 bytes per mate instead of 128, gk_mates_expand writes the 128-byte records in HBM.

/* words the mates take in compact form (offsets not counted) */
int gk_mates_compact_size(const gk_mate* mates, int64_t n_mates, int32_t n_threads, int64_t* n_words_out) {
  if (!n_words_out || n_mates < 0 || (n_mates && !mates)) { gk_set_error("bad compaction arguments"); return GK_ERR_ARG; }
  std::vector<int64_t> part((size_t)std::max(1, n_threads) + 1, 0);
  over_ranges(n_mates, std::max(1, n_threads), [&](int t, int64_t a, int64_t b) {
    int64_t sum = 0;
    int x, y, z;
    for (int64_t m = a; m < b; ++m) sum += mate_words(reinterpret_cast<const uint32_t*>(mates + m), &x, &y, &z);
    part[(size_t)t] = sum;
  });
  int64_t total = 0;
  for (int64_t v : part) total += v;
  *n_words_out = total;
  return GK_OK;
}

/* out: uint32 [n_mates + 1] word offsets followed by the words (capacity_words = all of it); the layout of
 * gk_mates_compact, so gk_mates_expand reads it once it is in HBM */
int gk_mates_compact_host(const gk_mate* mates, int64_t n_mates, int32_t n_threads, uint32_t* out, int64_t capacity_words) {
  if (n_mates < 0 || (n_mates && !mates) || !out || capacity_words < n_mates + 1) { gk_set_error("bad compaction arguments"); return GK_ERR_ARG; }
  if (n_mates >= (1ll << 27)) { gk_set_error("more than 2^26 pairs per call"); return GK_ERR_ARG; }
  n_threads = std::max(1, n_threads);
  std::vector<int64_t> first((size_t)n_threads + 1, 0);
  int used = 1;
  over_ranges(n_mates, n_threads, [&](int t, int64_t a, int64_t b) {
    int64_t sum = 0;
    int x, y, z;
    for (int64_t m = a; m < b; ++m) sum += mate_words(reinterpret_cast<const uint32_t*>(mates + m), &x, &y, &z);
    first[(size_t)t + 1] = sum;
  });
  for (int t = 0; t < n_threads; ++t) first[(size_t)t + 1] += first[(size_t)t];
  const int64_t total = first[(size_t)n_threads];
  if (n_mates + 1 + total > capacity_words || total >= (1ll << 32)) { gk_set_error("compact records do not fit the buffer"); return GK_ERR_CAPACITY; }
  uint32_t* const off = out;
  uint32_t* const words = out + n_mates + 1;
  // the ranges over_ranges hands out are the ones of the counting pass (same n, same thread count)
  std::atomic<int> next{0};
  (void)used;
  over_ranges(n_mates, n_threads, [&](int t, int64_t a, int64_t b) {
    int64_t at = first[(size_t)t];
    for (int64_t m = a; m < b; ++m) {
      const uint32_t* w = reinterpret_cast<const uint32_t*>(mates + m);
      int n_cw, n_mm, n_ins;
      const int n = mate_words(w, &n_cw, &n_mm, &n_ins);
      off[m] = (uint32_t)at;
      uint32_t* d = words + at;
      d[0] = w[0]; d[1] = w[1]; d[2] = w[2];
      d += 3;
      for (int i = 0; i < n_cw; ++i) d[i] = w[kCigWord + i];
      d += n_cw;
      for (int i = 0; i < n_mm; ++i) d[i] = w[kMmWord + i];
      d += n_mm;
      for (int i = 0; i < n_ins; ++i) d[i] = w[kInsWord + i];
      at += n;
    }
  });
  (void)next;
  off[n_mates] = (uint32_t)total;
  return GK_OK;
}

// The i-th interned inserted string (index strings first, then novel ones in first-seen order).
const char* gk_packer_string(gk_packer* pk, int64_t i) {
  if (!pk || i < 0 || i >= (int64_t)pk->ins_strings.size()) return "";
  return pk->ins_strings[(size_t)i].c_str();
}

}  // extern "C"

// The "reads" array of a .variant.json (hisat2.py:847-856: json.dump of dataclasses.asdict(PairRead)),
// written natively: for a million pairs the file is ~1.5 GB, mostly the SAM text of the pairs, and the
// Python encoders need a minute for it.  Output is what json.dump writes -- separators ", " and ": ",
// ensure_ascii escapes, key order of the PairRead fields -- so files made either way are identical.
#include <cstdint>
#include <cstdio>
#include <string>
#include <thread>
#include <vector>

#include "gk_ingest.h"
#include "graphkir_hip.h"

void gk_set_error(const char* fmt, ...);

namespace {

// json.encoder.encode_basestring_ascii on UTF-8 input
void put_json_string(std::string& out, std::string_view s) {
  static const char* hex = "0123456789abcdef";
  auto u_escape = [&](uint32_t cp) {
    out += "\\u";
    out.push_back(hex[(cp >> 12) & 15]); out.push_back(hex[(cp >> 8) & 15]);
    out.push_back(hex[(cp >> 4) & 15]); out.push_back(hex[cp & 15]);
  };
  out.push_back('"');
  for (size_t i = 0; i < s.size();) {
    const unsigned char c = (unsigned char)s[i];
    if (c >= 0x20 && c <= 0x7E && c != '"' && c != '\\') { out.push_back((char)c); ++i; continue; }
    if (c < 0x80) {
      switch (c) {
        case '"': out += "\\\""; break;
        case '\\': out += "\\\\"; break;
        case '\n': out += "\\n"; break;
        case '\r': out += "\\r"; break;
        case '\t': out += "\\t"; break;
        case '\b': out += "\\b"; break;
        case '\f': out += "\\f"; break;
        default: u_escape(c);
      }
      ++i;
      continue;
    }
    // multi-byte UTF-8 sequence -> code point -> \uXXXX (a surrogate pair above the BMP)
    int extra = (c >> 5) == 6 ? 1 : (c >> 4) == 14 ? 2 : (c >> 3) == 30 ? 3 : 0;
    uint32_t cp = extra == 1 ? (c & 0x1Fu) : extra == 2 ? (c & 0x0Fu) : extra == 3 ? (c & 0x07u) : c;
    size_t j = i + 1;
    for (int k = 0; k < extra && j < s.size(); ++k, ++j) cp = (cp << 6) | ((unsigned char)s[j] & 0x3Fu);
    if (cp >= 0x10000) {
      cp -= 0x10000;
      u_escape(0xD800 + (cp >> 10));
      u_escape(0xDC00 + (cp & 0x3FF));
    } else {
      u_escape(cp);
    }
    i = j;
  }
  out.push_back('"');
}

}  // namespace

// Appends `[{...}, {...}, ...]` to the file at `path`: one object per row i < n_rows with
//   l_sam / r_sam = lines pair_lines[2 * src[i]] and pair_lines[2 * src[i] + 1] of `sam_text`,
//   multiple = nh[i], backbone = genes[gene_of[i]],
//   lpv / rpv / lnv / rnv = names[ids[...]] over the four CSR segments of the row (file keys in the
//   order lpv, lnv, rpv, rnv of the dataclass).
extern "C" int gk_json_write_reads(const char* path, const char* sam_text, int64_t n_bytes, const int64_t* pair_lines,
                                   int64_t n_pairs, const int64_t* src, int64_t n_rows, const uint32_t* off,
                                   const uint32_t* ids, const char* const* names, int64_t n_names,
                                   const char* const* genes, int32_t n_genes, const uint8_t* gene_of,
                                   const uint8_t* nh) {
  if (!path || (!sam_text && n_bytes) || (n_rows && (!pair_lines || !src || !off || !names || !genes || !gene_of || !nh))) {
    gk_set_error("null argument");
    return GK_ERR_ARG;
  }
  const std::string_view text(sam_text, (size_t)n_bytes);
  const std::vector<int64_t> starts = gk_line_starts(text);
  std::vector<std::string> name_json((size_t)n_names), gene_json((size_t)n_genes);
  for (int64_t i = 0; i < n_names; ++i) put_json_string(name_json[(size_t)i], names[i]);
  for (int32_t g = 0; g < n_genes; ++g) put_json_string(gene_json[(size_t)g], genes[g]);
  FILE* f = fopen(path, "ab");
  if (!f) { gk_set_error("cannot append to %s", path); return GK_ERR_ARG; }
  bool ok = fputc('[', f) != EOF;
  const int n_thr = gk_ingest_threads();
  const int64_t kBatch = 32768;   // rows formatted per round (a few tens of MB of text)
  std::vector<std::string> piece((size_t)n_thr);
  std::vector<char> bad((size_t)n_thr, 0);
  for (int64_t r0 = 0; r0 < n_rows && ok; r0 += kBatch) {
    const int64_t r1 = std::min(n_rows, r0 + kBatch);
    auto work = [&](int t) {
      std::string& out = piece[(size_t)t];
      out.clear();
      const int64_t a = r0 + (r1 - r0) * t / n_thr, b = r0 + (r1 - r0) * (t + 1) / n_thr;
      auto list = [&](const char* key, uint32_t lo, uint32_t hi) {
        out += key;
        for (uint32_t k = lo; k < hi; ++k) {
          if (k > lo) out += ", ";
          if (ids[k] >= (uint64_t)n_names) { bad[(size_t)t] = 1; return; }
          out += name_json[ids[k]];
        }
        out.push_back(']');
      };
      for (int64_t i = a; i < b; ++i) {
        if (src[i] < 0 || src[i] >= n_pairs || gene_of[i] >= n_genes) { bad[(size_t)t] = 1; return; }
        if (i > 0) out += ", ";
        out += "{\"l_sam\": ";
        put_json_string(out, gk_line_at(text, starts, pair_lines[2 * src[i]]));
        out += ", \"r_sam\": ";
        put_json_string(out, gk_line_at(text, starts, pair_lines[2 * src[i] + 1]));
        out += ", \"multiple\": ";
        out += std::to_string((unsigned)nh[i]);
        out += ", \"backbone\": ";
        out += gene_json[gene_of[i]];
        const uint32_t* o = off + 4 * i;   // CSR segments: lpv, rpv, lnv, rnv
        list(", \"lpv\": [", o[0], o[1]);
        list(", \"lnv\": [", o[2], o[3]);
        list(", \"rpv\": [", o[1], o[2]);
        list(", \"rnv\": [", o[3], o[4]);
        out.push_back('}');
      }
    };
    std::vector<std::thread> pool;
    for (int t = 0; t < n_thr; ++t) pool.emplace_back(work, t);
    for (auto& th : pool) th.join();
    for (int t = 0; t < n_thr && ok; ++t) {
      if (bad[(size_t)t]) { fclose(f); gk_set_error("row refers to a missing line, gene or variant name"); return GK_ERR_ARG; }
      ok = fwrite(piece[(size_t)t].data(), 1, piece[(size_t)t].size(), f) == piece[(size_t)t].size();
    }
  }
  ok = ok && fputc(']', f) != EOF;
  ok = fclose(f) == 0 && ok;
  if (!ok) { gk_set_error("short write to %s", path); return GK_ERR_ARG; }
  return GK_OK;
}


// `samtools depth -aa` text (samtools_utils.py:9-22 reads it back): one line "gene\tpos\tdepth" per position of
// every backbone, positions 1-based, no header.  gene_off[g] .. gene_off[g + 1] are gene g's positions in `depth`.
extern "C" int gk_depth_write_tsv(const char* path, const char* const* genes, const int64_t* gene_off, int32_t n_genes,
                                  const uint32_t* depth) {
  if (!path || (n_genes && (!genes || !gene_off || !depth))) {
    gk_set_error("null argument");
    return GK_ERR_ARG;
  }
  FILE* f = fopen(path, "wb");
  if (!f) {
    gk_set_error("cannot write %s", path);
    return GK_ERR_ARG;
  }
  std::string buf;
  buf.reserve(1 << 20);
  auto put_uint = [&buf](uint64_t v) {
    char tmp[24];
    int n = 0;
    do { tmp[n++] = (char)('0' + v % 10); v /= 10; } while (v);
    while (n) buf.push_back(tmp[--n]);
  };
  bool ok = true;
  for (int32_t g = 0; g < n_genes && ok; ++g) {
    const std::string name = genes[g];
    for (int64_t i = gene_off[g]; i < gene_off[g + 1]; ++i) {
      buf += name;
      buf.push_back('\t');
      put_uint((uint64_t)(i - gene_off[g] + 1));
      buf.push_back('\t');
      put_uint(depth[i]);
      buf.push_back('\n');
      if (buf.size() > (1 << 20) - 256) {
        ok = fwrite(buf.data(), 1, buf.size(), f) == buf.size();
        buf.clear();
        if (!ok) break;
      }
    }
  }
  if (ok && !buf.empty()) ok = fwrite(buf.data(), 1, buf.size(), f) == buf.size();
  ok = (fclose(f) == 0) && ok;
  if (!ok) {
    gk_set_error("short write to %s", path);
    return GK_ERR_ARG;
  }
  return GK_OK;
}

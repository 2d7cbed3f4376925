// The two list-valued environment variables of the library (everything else that reads the environment is a setting a
// user of the command line may want: see README.md):
//   GK_TRACE       development traces on stderr, comma-separated: pool (large pool misses), search (host / wait split of
//                  gk_sample_search), ingest (phases of the host ingest), bench (host timeline of bench.py)
//   GK_TEST_HOOKS  switches the TESTS use to force rarely taken paths, comma-separated `name` or `name=value`:
//                  two_walks (tabulation: second walk instead of the saved words), novel_log2cap=N (size of the first
//                  novel-variant table), bam_segments=N (segments of the BAM record index), setsum=tiles|leaves,
//                  no_libdeflate (zlib for BGZF); read by the Python side only: bam_reader=samtools, ingest_ahead=N
#pragma once
#include <cstdlib>
#include <cstring>

// is `name` listed in the variable `var`?  With `value`: the text after `name=` (up to the next comma) goes there.
inline bool gk_env_list_has(const char* var, const char* name, char* value = nullptr, size_t value_cap = 0) {
  const char* e = getenv(var);
  if (!e) return false;
  const size_t n = strlen(name);
  for (const char* p = e; *p;) {
    const char* end = strchr(p, ',');
    const size_t len = end ? (size_t)(end - p) : strlen(p);
    if (len >= n && !strncmp(p, name, n) && (len == n || p[n] == '=')) {
      if (value && value_cap) {
        const size_t vlen = len > n ? len - n - 1 : 0;
        const size_t take = vlen < value_cap - 1 ? vlen : value_cap - 1;
        memcpy(value, p + n + (len > n ? 1 : 0), take);
        value[take] = 0;
      }
      return true;
    }
    if (!end) break;
    p = end + 1;
  }
  return false;
}
inline bool gk_trace(const char* what) { return gk_env_list_has("GK_TRACE", what); }
inline bool gk_test_hook(const char* name, char* value = nullptr, size_t cap = 0) {
  return gk_env_list_has("GK_TEST_HOOKS", name, value, cap);
}
inline long gk_test_hook_value(const char* name, long otherwise) {
  char buf[32];
  return gk_test_hook(name, buf, sizeof(buf)) && buf[0] ? atol(buf) : otherwise;
}

/*
 * graphkir_hip.h -- C ABI of the MI355X (gfx950) Graph-KIR allele-typing hot path.
 *
 * The reference (linnil1/KIR_graph) is pure Python and has no FFI; its plug-in
 * surface for this path is the Python API listed in SURVEY.md section 8(b).
 * This header is what a ctypes binding on the reference side would bind; every
 * entry point names the reference code it replaces.  Conventions:
 *   - plain C, opaque handles, caller-provided host buffers, no torch types;
 *   - every function returns 0 on success, a negative code on error, and
 *     gk_last_error() gives the message (thread-local);
 *   - device buffers are addressed by opaque 64-bit device pointers obtained
 *     from gk_malloc (so a host language only needs integers);
 *   - nothing here falls back to the CPU: without a HIP device every compute
 *     call fails with GK_ERR_NO_DEVICE.
 */
#ifndef GRAPHKIR_HIP_H
#define GRAPHKIR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GK_ABI_VERSION 1

#define GK_OK 0
#define GK_ERR_NO_DEVICE (-1)
#define GK_ERR_HIP (-2)
#define GK_ERR_ARG (-3)
#define GK_ERR_ASSERT (-4)   /* a reference `assert` would have fired (e.g. window lo > hi) */
#define GK_ERR_CAPACITY (-5)

/* ---- packed alignment record: 128 bytes per mate, 2 consecutive mates = 1 pair.
 * mate 0 of a pair is the record emitted as `left` by readPair (hisat2.py:270),
 * i.e. the LATER line of the name-collated stream; mate 1 the earlier one.
 * It carries exactly the fields recordToRawVariant / filterRead / getNH read
 * from the SAM text (hisat2.py:342-350, 551-569, 95-100). */
#define GK_MAX_CIG 14
#define GK_MAX_MM 16
#define GK_MAX_INS 6
#define GK_MAX_EVENTS 22 /* mismatches + I ops + D ops of one mate (NM excludes graph variants, so NM <= 4
                            does not bound them) */
#define GK_CIG_M 0
#define GK_CIG_I 1
#define GK_CIG_D 2
#define GK_CIG_S 4
#define GK_NM_ABSENT 255

typedef struct gk_mm {
  uint16_t ref_off; /* reference offset of the mismatch from pos0 */
  uint8_t base;     /* read base (ASCII) */
  uint8_t rsv;
} gk_mm;

typedef struct gk_mate {
  uint32_t pos0;  /* 0-based leftmost reference position (POS-1) */
  uint16_t flag;  /* SAM FLAG */
  uint8_t ref;    /* backbone ordinal in sorted-name order */
  uint8_t nh;     /* NH:i (1 when absent), saturated at 255 */
  uint8_t nm;     /* NM:i saturated at 254, GK_NM_ABSENT when the tag is missing */
  uint8_t n_cig;  /* CIGAR ops stored (0 when filterRead already rejects the mate) */
  uint8_t n_mm;   /* MD mismatches stored */
  uint8_t n_ins;  /* inserted strings stored */
  uint16_t cig[GK_MAX_CIG]; /* len << 4 | op */
  gk_mm mm[GK_MAX_MM];
  uint32_t ins[GK_MAX_INS]; /* string-table id of the k-th I op */
} gk_mate;

/* A mate that does not fit gk_mate (more ops, mismatches, inserted strings or events than it holds, an op longer
 * than 4095, a mismatch beyond reference offset 65535) is kept in a second, rarely used format.  Both mates of
 * such a pair go there: their gk_mate records keep the header fields, carry n_cig == GK_SPILLED and, in ins[0],
 * the index of the pair in the wide array (records 2 * index and 2 * index + 1).  The device walks wide pairs with
 * the same code as the others, one workgroup per pair, events in global memory. */
#define GK_SPILLED 0xFF
#define GK_WIDE_CIG 128
#define GK_WIDE_MM 256
#define GK_WIDE_INS 120
#define GK_WIDE_EVENTS 384 /* mismatches + I ops + D ops of one wide mate */
typedef struct gk_mate_wide {
  uint32_t pos0;
  uint16_t flag;
  uint8_t ref;
  uint8_t nh;
  uint8_t nm;
  uint8_t rsv0;
  uint16_t n_cig;
  uint16_t n_mm;
  uint16_t n_ins;
  uint32_t cig[GK_WIDE_CIG]; /* len << 4 | op, lengths up to 2^28 - 1 */
  uint32_t mm[GK_WIDE_MM];   /* ref_off << 8 | read base (ASCII), offsets up to 2^24 - 1 */
  uint32_t ins[GK_WIDE_INS]; /* string-table id of the k-th I op */
  uint8_t rsv1[16];
} gk_mate_wide; /* 2048 bytes */

typedef struct gk_ctx gk_ctx;
typedef struct gk_index gk_index;
typedef struct gk_tab gk_tab;
typedef struct gk_lut gk_lut;
typedef uint64_t gk_dptr; /* device address */

/* ---- runtime ------------------------------------------------------------ */
int gk_abi_version(void);
const char* gk_last_error(void);
int gk_device_count(int* n);
int gk_ctx_create(int device, gk_ctx** out);
/* The same with a stream priority: urgent != 0 asks for the device's highest stream priority -- for the short,
 * latency-bound work of a sample (copy + tabulation of the next sample, the preamble of a gene loop), whose small
 * kernels otherwise queue behind the workgroups of another sample's long kernels.  No reference counterpart (the
 * reference is single-threaded numpy); the hot path it serves is hisat2.py:551-620 + kir_typing.py:103-132. */
int gk_ctx_create_priority(int device, int urgent, gk_ctx** out);
int gk_ctx_destroy(gk_ctx* ctx);
int gk_sync(gk_ctx* ctx);
/* device memory as the runtime reports it (free, total) and the bytes the pools of this process's contexts hold idle;
 * cohort.SampleTyper admits samples by it (the reference holds one sample at a time, main.py:171-220) */
int gk_device_memory(gk_ctx* ctx, int64_t* free_bytes, int64_t* total_bytes, int64_t* pool_cached_bytes);
int gk_malloc(gk_ctx* ctx, size_t bytes, gk_dptr* out);
int gk_free(gk_ctx* ctx, gk_dptr p);
int gk_memset(gk_ctx* ctx, gk_dptr p, int value, size_t bytes);
int gk_h2d(gk_ctx* ctx, gk_dptr dst, const void* src, size_t bytes);
int gk_d2h(gk_ctx* ctx, void* dst, gk_dptr src, size_t bytes);
int gk_d2d(gk_ctx* ctx, gk_dptr dst, gk_dptr src, size_t bytes);
/* HIP-event timing on the context stream (bench.py roofline leg). */
int gk_timer_start(gk_ctx* ctx);
int gk_timer_stop_ms(gk_ctx* ctx, float* ms);
/* Per-kernel HIP-event spans on the context stream: enable, run, then collect launches / total ms
 * per kernel id (names via gk_prof_kernel_name). */
int gk_prof_enable(gk_ctx* ctx, int on);
int gk_prof_kernel_count(void);
const char* gk_prof_kernel_name(int id);
int gk_prof_collect(gk_ctx* ctx, int64_t* launches, double* total_ms);

/* ---- index: replaces getVariants() (hisat2.py:183-203) as a device table.
 * key[v] = ref:8 | pos:24 | type:2 | val:30 sorted ascending (msa2hisat.py:48-53). */
int gk_index_create(gk_ctx* ctx, const uint64_t* key, int32_t n_var, const int32_t* gene_vbeg,
                    int32_t n_gene, gk_index** out);
int gk_index_destroy(gk_index* idx);

/* ---- tabulation: replaces filterRead + extractVariant (hisat2.py:541-578, 803-844).
 * d_mates: 2*n_pairs gk_mate records resident in HBM.  On return the handle owns:
 *   n_valid pairs that pass filterRead on both mates, in input order;
 *   CSR lists per valid pair in the order lpv, rpv, lnv, rnv (variant ordinals:
 *   < n_var = index ordinal, >= n_var = n_var + rank of the novel variant in
 *   first-appearance order, i.e. id "nv{novel_base + rank}");
 *   the novel variants' keys in that order. */
typedef struct gk_tab_info {
  int64_t n_pairs;
  int64_t n_valid;
  int64_t n_ids;
  int32_t n_novel;
  int32_t err_flags;    /* bit0: window lo > hi somewhere (reference assert, hisat2.py:744) */
  gk_dptr d_pair_src;   /* int32 [n_valid] input pair index */
  gk_dptr d_off;        /* uint32 [4*n_valid+1] */
  gk_dptr d_ids;        /* uint32 [n_ids] */
  gk_dptr d_pair_gene;  /* uint8 [n_valid] backbone ordinal */
  gk_dptr d_pair_nh;    /* uint8 [n_valid] */
  gk_dptr d_novel_key;  /* uint64 [n_novel] */
} gk_tab_info;
int gk_tabulate(gk_ctx* ctx, gk_index* idx, gk_dptr d_mates, int64_t n_pairs, gk_tab** out);
/* Same with the optional pileup correction of mismatches (hisat2.errorCorrection 609-654, applied in
 * recordToVariants 684-685 before the id lookup): d_corr uint8 [total positions][5] holds, per reference
 * position and read base (A, C, G, T, N), the base to use instead (0 = keep the read's);
 * d_gene_pos0 int64 [n_gene + 1] = first position of every backbone in that table. */
int gk_tabulate_corrected(gk_ctx* ctx, gk_index* idx, gk_dptr d_mates, int64_t n_pairs, gk_dptr d_corr,
                          gk_dptr d_gene_pos0, gk_tab** out);
/* Same for a sample with pairs in the wide format (gk_mate_wide above): wide = host array of 2 * n_spill records,
 * spill_pair = the pairs they stand for, ascending (as gk_packer_spill_records hands them out).  The tabulation
 * keeps a device copy of the wide records (gk_depth reads their CIGARs).  n_spill == 0: gk_tabulate_corrected. */
int gk_tabulate_spilled(gk_ctx* ctx, gk_index* idx, gk_dptr d_mates, int64_t n_pairs, gk_dptr d_corr,
                        gk_dptr d_gene_pos0, const gk_mate_wide* wide, const int64_t* spill_pair, int64_t n_spill,
                        gk_tab** out);
/* Same handle from host CSR lists (the `.variant.json` hand-off of hisat2.py:847-866 loaded by
 * loadReadsAndVariantsData): off[4*n_valid+1] in list order lpv, rpv, lnv, rnv; ordinals < n_var_total. */
int gk_tab_from_csr(gk_ctx* ctx, int32_t n_var_total, int64_t n_valid, const uint32_t* off, const uint32_t* ids,
                    const uint8_t* pair_gene, const uint8_t* pair_nh, gk_tab** out);
int gk_tab_get_info(gk_tab* tab, gk_tab_info* info);
int gk_tab_destroy(gk_tab* tab);

/* ---- ordered selection: stable compaction of [0,n) where flag != 0.
 * Used for removeMultipleMapped + groupReads (hisat2.py:943-948, kir_typing.py:15-20)
 * and removeEmptyReads (typing_mulit_allele.py:274-281). */
int gk_select_gene(gk_ctx* ctx, gk_tab* tab, int gene, int multiple, gk_dptr d_rows_out, int64_t* n_out);
int gk_select_nonempty(gk_ctx* ctx, gk_tab* tab, gk_dptr d_rows, int64_t n_rows, gk_dptr d_vflag,
                       gk_dptr d_rows_out, int64_t* n_out);

/* ---- variant error correction: AlleleTyping.errorCorrection (typing_mulit_allele.py:302-338).
 * d_vflag uint8 [n_var + n_novel]: bit0 = dropped from positive lists, bit1 = from negative lists.
 * gk_variant_count tallies the surviving ids of the given rows into d_cnt (uint32 [2][n_var+n_novel],
 * positives then negatives); gk_variant_correct applies the <3 / <20 % thresholds to d_vflag. */
int gk_variant_count(gk_ctx* ctx, gk_tab* tab, gk_dptr d_rows, int64_t n_rows, gk_dptr d_vflag,
                     gk_dptr d_cnt);
/* Same tally with a hint: the rows' index variants lie in [vbeg, vend) (one gene), so their counters
 * are privatised in LDS and flushed once per workgroup. */
int gk_variant_count_range(gk_ctx* ctx, gk_tab* tab, gk_dptr d_rows, int64_t n_rows, gk_dptr d_vflag,
                           gk_dptr d_cnt, int32_t vbeg, int32_t vend);
int gk_variant_correct(gk_ctx* ctx, gk_tab* tab, gk_dptr d_cnt, gk_dptr d_vflag);
/* Ordinals and (positive, negative) tallies of the variants that survive d_vflag, compacted on the
 * device: input of isHomozygous (typing_mulit_allele.py:807-857).  Host arrays hold max_out entries. */
int gk_variant_surviving(gk_ctx* ctx, gk_tab* tab, gk_dptr d_cnt, gk_dptr d_vflag, int64_t max_out,
                         int32_t* ord_out, uint32_t* pos_out, uint32_t* neg_out, int64_t* n_out);

/* The same restricted to one backbone: index ordinals [vbeg, vend) and the novel variants on backbone `gene`. */
int gk_variant_surviving_gene(gk_ctx* ctx, gk_tab* tab, gk_dptr d_cnt, gk_dptr d_vflag, int32_t gene, int32_t vbeg,
                              int32_t vend, int64_t max_out, int32_t* ord_out, uint32_t* pos_out, uint32_t* neg_out,
                              int64_t* n_out);
/* errorCorrection + removeEmptyReads (302-338, 274-281) for EVERY gene of a sample at once (variants of different
 * backbones are disjoint, so one tally, one pass of the thresholds and one stable compaction give what the per-gene
 * calls give).  d_vflag uint8 [n_var + n_novel] and d_cnt uint32 [2][n_var + n_novel] are zeroed and filled here;
 * d_rows int32 [n_valid] receives the rows with a surviving id grouped by backbone in row order (NH == 1 only
 * unless `multiple`), gene_off_out int64 [n_gene + 1] their bounds.  Needs a tabulation made by gk_tabulate. */
int gk_sample_prepare(gk_ctx* ctx, gk_tab* tab, int32_t multiple, gk_dptr d_vflag, gk_dptr d_cnt, gk_dptr d_rows,
                      int64_t* gene_off_out);
/* gk_sample_prepare + gk_variant_surviving (every backbone) + the tabulation's novel keys in one call and two waits
 * instead of six: the preamble of the gene loop (kir_typing.py:80-101 groups the reads, typing_mulit_allele.py:302-338,
 * 274-281 and 807-834 per gene).  ord / pos / neg_out hold max_out entries (>= index + novel variants of the sample;
 * GK_ERR_CAPACITY otherwise); novel_key_out [n_novel] may be NULL. */
int gk_sample_prepare_all(gk_ctx* ctx, gk_tab* tab, int32_t multiple, gk_dptr d_vflag, gk_dptr d_cnt, gk_dptr d_rows,
                          int64_t* gene_off_out, int64_t max_out, int32_t* ord_out, uint32_t* pos_out, uint32_t* neg_out,
                          int64_t* n_out, uint64_t* novel_key_out);
/* The same preamble for the EXON model of every gene (AlleleTypingExonFirst.__init__, typing_mulit_allele.py:640-664):
 * d_vflag comes in holding 3 for every variant outside the exons (removeIntronVariant 703-714) and is corrected TWICE
 * (644-645 and again inside the base class, 664); rows without a surviving id are removed, the surviving tallies come back
 * as for gk_sample_prepare_all. */
int gk_sample_prepare_exon(gk_ctx* ctx, gk_tab* tab, int32_t multiple, gk_dptr d_vflag, gk_dptr d_cnt, gk_dptr d_rows,
                           int64_t* gene_off_out, int64_t max_out, int32_t* ord_out, uint32_t* pos_out, uint32_t* neg_out,
                           int64_t* n_out);

/* ---- compatibility: reads2AlleleProb (typing_mulit_allele.py:340-381).
 * d_mask uint32 [vend-vbeg][words]: allele bit rows of the gene's index variants.
 * Outputs are column-major [allele][row] with leading dimension n_rows:
 *   d_probs double (ordered product lpv, rpv, lnv, rnv of 0.999 / 0.001), may be 0;
 *   d_miss uint8 (#mismatching ids, saturated 255), may be 0;  d_nvar uint16 [n_rows], may be 0.
 * keep_empty != 0: a row without any kept id scores 0.999 for every allele (no_empty=False, 372-374;
 * the caller then passes all rows instead of gk_select_nonempty's); otherwise such a row scores 1.0. */
int gk_compat(gk_ctx* ctx, gk_tab* tab, gk_dptr d_rows, int64_t n_rows, gk_dptr d_vflag,
              int32_t vbeg, int32_t vend, gk_dptr d_mask, int32_t words, int32_t n_allele, int32_t keep_empty,
              gk_dptr d_probs, gk_dptr d_miss, gk_dptr d_nvar);

/* ---- log10 through a value table: log_probs = np.log10(probs) (typing_mulit_allele.py:263).
 * The distinct probability bit patterns are collected on the device, the HOST evaluates
 * numpy.log10 on them (so the bits are the reference's on this machine) and the table is applied
 * on the device. */
int gk_lut_create(gk_ctx* ctx, int32_t log2_capacity, gk_lut** out);
int gk_lut_destroy(gk_lut* lut);
int gk_lut_collect(gk_lut* lut, gk_dptr d_vals, int64_t n);              /* insert, async */
int gk_lut_pending(gk_lut* lut, int32_t* n_total, int32_t* n_known);     /* syncs */
int gk_lut_export(gk_lut* lut, int32_t first, int32_t count, double* keys_out);
int gk_lut_define(gk_lut* lut, int32_t first, int32_t count, const double* log_vals);
int gk_lut_apply(gk_lut* lut, gk_dptr d_in, gk_dptr d_out, int64_t n);
/* pending + export + host log10 + define in one call, serialised inside the library (typing_mulit_allele.py:263:
 * np.log10 -- the binding hands numpy.log10 in, so the bits are numpy's on the machine at hand).  Outputs may be NULL:
 * values newly defined, values defined in all, entries claimed by kernels still running elsewhere. */
typedef int (*gk_log10_fn)(const double* values, int64_t n, double* out);
int gk_lut_resolve(gk_lut* lut, gk_log10_fn log10_fn, int32_t* n_new, int32_t* n_known, int32_t* n_undefined);
/* The same without draining the device: for a caller whose own kernels have completed (it waited for them) while other
 * streams keep running; entries a running kernel has claimed but not stored yet stay undefined (n_undefined), the values
 * this caller's kernels stored before them are defined (typing_mulit_allele.py:263). */
int gk_lut_resolve_stored(gk_lut* lut, gk_log10_fn log10_fn, int32_t* n_new, int32_t* n_known, int32_t* n_undefined);
int gk_lut_known(gk_lut* lut, int32_t* n_known);
/* reads2AlleleProb and np.log10 in one pass (typing_mulit_allele.py:257-263): gk_compat's product,
 * mapped through the value table as it is written; d_log is column-major double [allele][row].
 * A product whose log10 is not defined yet is inserted into the table and stored as NaN: when
 * gk_lut_pending reports n_total > n_known afterwards, export/define the new values and call again. */
int gk_compat_log(gk_ctx* ctx, gk_tab* tab, gk_dptr d_rows, int64_t n_rows, gk_dptr d_vflag,
                  int32_t vbeg, int32_t vend, gk_dptr d_mask, int32_t words, int32_t n_allele, int32_t keep_empty,
                  gk_lut* lut, gk_dptr d_log);

/* ---- likelihood search: AlleleTyping.addCandidate (typing_mulit_allele.py:478-598).
 * L is column-major double [n_allele][ld] (ld >= n_rows).  Reductions over rows follow
 * numpy's add.reduce tree exactly (8192-row chunks, pairwise blocks of 128, 8 strided
 * accumulators) so that ranks and ties are the reference's.
 *   gk_maxsum: out[t*n_cols + j] = sum_r max(L[r, cols[j]], max_k L[r, ids[t*c_prev + k]])
 *              (c_prev == 0: plain column sums, line 514; else lines 540-542)
 *   gk_fraction: frac[k*c + j] = (sum_r [L[r,ids[k,j]] == max_j'] / #argmax) / n_rows (575-580)
 *   gk_setmax:  P[t][r] = max_k L[r, ids[t*c + k]]  (allele_prob, line 569), column-major [t][ld] */
int gk_maxsum(gk_ctx* ctx, gk_dptr d_L, int64_t n_rows, int64_t ld, const int32_t* ids, int32_t n_sets,
              int32_t c_prev, const int32_t* cols, int32_t n_cols, double* out);
int gk_fraction(gk_ctx* ctx, gk_dptr d_L, int64_t n_rows, int64_t ld, const int32_t* ids,
                int32_t n_sets, int32_t c, double* frac_out);
int gk_setmax(gk_ctx* ctx, gk_dptr d_L, int64_t n_rows, int64_t ld, const int32_t* ids, int32_t n_sets,
              int32_t c, gk_dptr d_P);

/* ---- integer bound of a search step (typing_mulit_allele.py:534-567; SURVEY.md section 8 "integer reformulation").
 * gk_compat_log_miss = gk_compat_log that also writes the mismatch counts miss[r, a] as a u8 table
 *   d_miss8 [n_allele][ldm] (ldm a multiple of 64 >= n_rows, rows past the end zero), derived from the
 *   log-likelihoods; *d_flags (uint32) gets bit 0 when some count is >= 100 (products near underflow: the caller
 *   must then use gk_maxsum for this gene); bits 2 and 3: see gk_compat_patch.
 * gk_miss_colsum: d_msum uint32 [n_cols] = column sums of that table.
 * gk_bound_step: M[t, j] = sum_r min(miss[r, cols[j]], min_k miss[r, ids[t*c_prev + k]]) for every candidate,
 *   restricted to first[t*n_cols + j] != 0 (first occurrences of an allele multiset, uniqueAllele 456-476);
 *   returns the candidates with M <= M_T, the top_n-th smallest (hdr_out = {candidates, M_T, selected, 0};
 *   idx_out / m_out hold min(selected, cap) entries in no particular order: flat index t*n_cols + j and M).
 *   Two sets with different M are ordered by M in float64 as well, so the reference's top_n-by-value (567)
 *   lies inside the selection.
 * gk_setsum: exact float64 value (540-542 for one set: sum_r max_j L[r, ids[k,j]], numpy's tree) and the
 *   abundance shares (575-580) of the given sets in one pass. */
int gk_compat_log_miss(gk_ctx* ctx, gk_tab* tab, gk_dptr d_rows, int64_t n_rows, gk_dptr d_vflag,
                       int32_t vbeg, int32_t vend, gk_dptr d_mask, int32_t words, int32_t n_allele,
                       int32_t keep_empty, gk_lut* lut, gk_dptr d_log, gk_dptr d_miss8, int64_t ldm, gk_dptr d_flags);
int gk_miss_colsum(gk_ctx* ctx, gk_dptr d_miss8, int64_t ldm, int32_t n_cols, gk_dptr d_msum);
/* Bits 2 and 3 of *d_flags after gk_compat_log_miss: bit 2 = some product had no log10 in the value table yet; its entry
 * of d_log holds the PRODUCT itself (strictly positive, which no log10 of a probability is) until gk_compat_patch, after
 * the table has been resolved (gk_lut_resolve*), puts the log10 and the mismatch byte there -- one pass over this gene's
 * table instead of the kernel that made it, the tables of the other genes are not concerned.  Bit 3 = such a product was
 * +0.0 and could not mark itself (NaN stored): write the table again.  gk_compat_patch clears *d_flags first; bit 2
 * comes back when some value is still undefined, bit 0 as above (typing_mulit_allele.py:263, 340-381). */
int gk_compat_patch(gk_ctx* ctx, gk_lut* lut, gk_dptr d_log, int64_t n_rows, int32_t n_allele, gk_dptr d_miss8,
                    int64_t ldm, gk_dptr d_flags);
/* The index form of gk_compat_log_miss (typing_mulit_allele.py:263, 340-381): d_lidx uint16 [n_allele][ldm] = dense index
 * of every log-likelihood in the value table (0xFFFF while a product's log10 is undefined: resolve and call again);
 * *d_flags bit 1 = the table holds more than 65535 values (use gk_compat_log_miss).  gk_expand_index writes the float64
 * form d_L double [n_allele][ld] of such a table. */
int gk_compat_index(gk_ctx* ctx, gk_tab* tab, gk_dptr d_rows, int64_t n_rows, gk_dptr d_vflag, int32_t vbeg,
                    int32_t vend, gk_dptr d_mask, int32_t words, int32_t n_allele, int32_t keep_empty, gk_lut* lut,
                    gk_dptr d_lidx, gk_dptr d_miss8, int64_t ldm, gk_dptr d_flags);
int gk_expand_index(gk_ctx* ctx, gk_lut* lut, gk_dptr d_lidx, int64_t ldi, int64_t n_rows, int32_t n_allele, gk_dptr d_L,
                    int64_t ld);
int gk_bound_step(gk_ctx* ctx, gk_dptr d_miss8, int64_t ldm, int64_t n_rows, gk_dptr d_msum, const int32_t* ids,
                  int32_t n_sets, int32_t c_prev, const int32_t* cols, int32_t n_cols, const uint8_t* first,
                  int32_t top_n, int32_t cap, uint32_t* hdr_out, int32_t* idx_out, uint32_t* m_out);
int gk_setsum(gk_ctx* ctx, gk_dptr d_L, int64_t n_rows, int64_t ld, const int32_t* ids, int32_t n_sets, int32_t c,
              double* value_out, double* frac_out);

/* ---- the whole greedy search of one gene, host half included: AlleleTyping.addCandidate called n_steps times
 * (typing_mulit_allele.py:478-598 with 156-214, 456-476), in the calling thread, without the host language's
 * interpreter.  Every step: first occurrences of the candidate multisets, gk_bound_step + gk_setsum (when
 * d_miss8 / d_msum are given) or gk_maxsum + gk_fraction, the top_n cut and the stable three-key ranking.
 * `argsort` stands for numpy.argsort wherever the reference calls it (its order among equal values is part of
 * the result): ascending order of `values` into `order_out`, 0 on success.  colsum_in (may be NULL) = the
 * per-allele column sums when the caller already has them.  Results per step (0-based; step s holds sets of
 * s + 1 alleles): rows, then value[rows], value_sum_indv[rows][n], allele ids[rows][n], fraction[rows][n]. */
typedef int (*gk_argsort_fn)(const double* values, int64_t n, int64_t* order_out);
typedef struct gk_search gk_search;
int gk_search_run(gk_ctx* ctx, gk_dptr d_L, int64_t n_rows, int64_t ld, int32_t n_allele, gk_dptr d_miss8,
                  int64_t ldm, gk_dptr d_msum, const int32_t* cols, int32_t n_cols, int32_t n_steps, int32_t top_n,
                  gk_argsort_fn argsort, const double* colsum_in, gk_search** out);
/* ---- the searches of ALL genes of one sample in one call (kir_typing.py:103-132 is the reference's gene loop; per gene
 * typing_mulit_allele.py:340-381 for the table and 478-598 for the search).  The genes advance in lock-step on the
 * calling thread and the context's ONE stream: compatibility tables of every gene, one wait; column sums of every gene,
 * one wait; then step 2, 3, ... of every gene that has one -- bound, wait, exact sums of the selections, wait, ranking.
 * A sample costs about ten waits whatever its number of genes, so one host thread feeds the GPU.  more_ctx (may be
 * NULL): further contexts of the calling thread on the same GPU -- every gene queues its work on one of them, so the
 * kernels of different genes overlap; a wait then covers all of them.
 * A job names the gene's rows (after error correction / empty-read removal: gk_sample_prepare), its variant span and
 * bit rows, and the tables the caller allocated: d_L double [n_allele][n_rows] -- or d_lidx for the index form
 * (2 bytes per entry instead of 8: the sums gather the float64 from the value table) --; optionally d_miss8 u8
 * [n_allele][ldm], d_msum uint32 [n_allele], d_flags uint32 [1] for the integer bound (required with d_lidx).  n_steps = copy-number steps to run (1 for a
 * gene typed as homozygous).  Outputs per job: bound_ok (the integer bound served the gene), passes (how often its table
 * was written: > 1 when the value table met new products), and out[i] = its search (gk_search_*; NULL without rows). */
typedef struct gk_gene_job {
  gk_dptr d_rows;
  int64_t n_rows;
  gk_dptr d_mask;
  gk_dptr d_L;
  gk_dptr d_miss8;
  int64_t ldm;
  gk_dptr d_msum;
  gk_dptr d_flags;
  gk_dptr d_lidx; /* uint16 [n_allele][ldm]: the INDEX form of the table (gk_compat_index) -- given with d_L == 0 */
  int32_t vbeg, vend;
  int32_t words, n_allele;
  int32_t n_steps, top_n;
  int32_t bound_ok, passes; /* out */
  int32_t indexed, patches; /* out: d_lidx holds the table (0: the value table outgrew 16-bit indices, the call worked on a float64 table of its own); how often the table was patched (gk_compat_patch) */
  /* Exon-first (typing_mulit_allele.py:740-746: for every exon candidate, addCandidate(alleles of its k-th group) per copy):
   * table_of >= 0: this job is a SEARCH on the table that job `table_of` writes (that job may have n_steps == 0: table and
   * column sums only); -1: the job writes its own table.  n_step_cols > 0: step k of the search offers the alleles
   * step_cols[step_cols_off[k] .. step_cols_off[k + 1]) (host arrays; the last list serves any further step) instead of
   * every allele.  Both need the pipelined form (float64 + mismatch tables, one stream). */
  int32_t table_of, n_step_cols;
  const int32_t* step_cols;
  const int32_t* step_cols_off;
} gk_gene_job;
int gk_sample_search(gk_ctx* ctx, gk_ctx** more_ctx, int32_t n_more, gk_tab* tab, gk_dptr d_vflag, gk_lut* lut,
                     gk_gene_job* jobs, int32_t n_jobs, gk_argsort_fn argsort, gk_log10_fn log10_fn, gk_search** out);
int gk_search_steps(gk_search* s, int32_t* n_steps);
int gk_search_info(gk_search* s, int32_t step, int32_t* n, int64_t* rows, int32_t* bounded);
int gk_search_copy(gk_search* s, int32_t step, double* value, double* sum_indv, int32_t* ids, double* frac);
/* every step of many searches in one call (the candidate searches of exon-first, typing_mulit_allele.py:740-797, are
 * adopted by the hundred): totals[3] = steps, rows, cells (rows x set size) over the listed searches; with the arrays
 * given, meta = [steps of search 0 .. n - 1 | per step (set size, rows, bounded)] and value [rows], sum_indv / frac / ids
 * [cells] hold the rows of the steps back to back in (search, step) order (gk_search_copy's columns) */
int gk_search_export(gk_search* const* s, int32_t n, int64_t* totals, int64_t* meta, double* value, double* sum_indv,
                     double* frac, int32_t* ids);
int gk_search_colsum(gk_search* s, double* out);
/* launch geometries of the run, 7 int64 per device call: kind (0 gk_maxsum, 1 gk_bound_step, 2 gk_setsum /
 * gk_fraction) and the arguments of the roofline model (kir_graph_amd/roofmodel.py) */
int gk_search_log(gk_search* s, int64_t* out, int64_t capacity, int64_t* n_out);
int gk_search_destroy(gk_search* s);
/* isHomozygous (typing_mulit_allele.py:835-857) on flat observations (position, label code, negative?, count):
 * *homozygous = 0 when some position shows a second allele above 0.1 and 1 / (2 cn) of its kept counts. */
int gk_site_verdict(const int64_t* pos, const int64_t* code, const uint8_t* negative, const int64_t* count, int64_t n,
                    int32_t cn, int32_t* homozygous);
/* the same verdict straight from the surviving tallies (gk_variant_surviving*): ordinals into `keys` (index keys
 * followed by the sample's novel keys) with their positive / negative counts; label_of_insert[id] = the label code of an
 * inserted string (a one-base insertion prints like a substitution of that base) */
int gk_site_verdict_tallies(const uint64_t* keys, int64_t n_keys, const int64_t* label_of_insert, int64_t n_insert,
                            const int32_t* ordinal, const uint32_t* positive, const uint32_t* negative, int64_t n,
                            int32_t cn, int32_t* homozygous);
/* The same verdict for every gene of a sample in one call (the gene loop of kir_typing.py:103-132 asks it once per gene):
 * the tallies are grouped by gene, group g = entries [bounds[g], bounds[g + 1]) with copy number cn[g];
 * homozygous[g] = 0 for cn[g] <= 1 (isHomozygous is not asked then, typing_mulit_allele.py:386). */
int gk_site_verdict_genes(const uint64_t* keys, int64_t n_keys, const int64_t* label_of_insert, int64_t n_insert,
                          const int32_t* ordinal, const uint32_t* positive, const uint32_t* negative,
                          const int64_t* bounds, int32_t n_groups, const int32_t* cn, int32_t* homozygous);

/* ---- EM strategy: typing_em.py:68-188.
 * gk_em_sets: per-row candidate-allele bit sets (getCandidateAllelePerRead + getMostFreqAllele).
 * gk_em_run:  SQUAREM EM on weighted distinct sets (hisatEMnp 107-188), one workgroup. */
int gk_em_sets(gk_ctx* ctx, gk_tab* tab, gk_dptr d_rows, int64_t n_rows, int32_t vbeg, int32_t vend,
               gk_dptr d_mask, int32_t words, gk_dptr d_sets_out /* uint32 [n_rows][words] */);
/* distinct rows of d_sets (uint32 [n_rows][words], as written by gk_em_sets) with their multiplicities,
 * in no particular order; GK_ERR_CAPACITY when there are more than max_out (n_out then holds the count). */
int gk_em_distinct(gk_ctx* ctx, gk_dptr d_sets, int64_t n_rows, int32_t words, int32_t max_out,
                   uint32_t* sets_out, uint32_t* count_out, int32_t* n_out);
int gk_em_run(gk_ctx* ctx, const uint32_t* sets, const double* weight, int32_t n_sets, int32_t words,
              int32_t n_allele, int32_t iter_max, double diff_threshold, double* prob_out,
              int32_t* iters_out);
/* The EM strategy for ALL genes of a sample in one call on one host thread and one stream (kir_typing.py:163-195 is the
 * reference's gene loop; typing_em.py:68-188 per gene): candidate sets and distinct sets of every gene queued together,
 * the host half (numpy.unique's ascending order of the sets, the reads naming each allele, the empty set dropped) in the
 * library, the SQUAREM loops of all genes in ONE launch -- a workgroup per gene.  A job names a gene's rows (the NH == 1
 * pairs, gk_select_gene), its variant span and bit rows; prob_out / count_out hold n_allele entries per job, one job after
 * the other.  Out per job: the number of distinct candidate sets and the SQUAREM steps taken (< iter_max: stopped on
 * diff_threshold).  GK_ERR_CAPACITY when a gene has more than 2^18 distinct sets (use the per-gene calls). */
typedef struct gk_em_job {
  gk_dptr d_rows;
  int64_t n_rows;
  gk_dptr d_mask;
  int32_t vbeg, vend;
  int32_t words, n_allele;
  int32_t n_distinct, iterations; /* out */
} gk_em_job;
int gk_sample_em(gk_ctx* ctx, gk_tab* tab, gk_em_job* jobs, int32_t n_jobs, int32_t iter_max, double diff_threshold,
                 double* prob_out, int64_t* count_out);

/* ---- host ingest (no GPU): name-collated SAM text -> gk_mate records.
 * Native form of readPair (hisat2.py:228-276), of the field reads of filterRead / getNH (551-569,
 * 95-100) and of the CIGAR / MD / Zs consistency checks of recordToRawVariant (279-515).  Text is fed
 * in chunks; on a reference `assert` / NotImplementedError the feed stops and gk_packer_error tells
 * kind (1 AssertionError, 2 NotImplementedError, 3 record capacity, 4 ValueError) and input line. */
typedef struct gk_packer gk_packer;
int gk_packer_create(const char* const* gene_names, int32_t n_genes, const char* const* ins_strings,
                     int32_t n_ins, gk_packer** out);
int gk_packer_destroy(gk_packer* pk);
int gk_packer_feed(gk_packer* pk, const char* text, size_t n_bytes, int32_t final);
int gk_packer_counts(gk_packer* pk, int64_t* n_lines, int64_t* n_reads, int64_t* n_pairs, int64_t* n_strange,
                     int64_t* n_strings);
int gk_packer_error(gk_packer* pk, int32_t* kind, int64_t* line_index);
/* records are written straight into mates_out (capacity in records, 2 per pair) instead of the packer's own
 * storage -- e.g. gk_host_alloc memory, so the upload starts from where the decoder wrote; before the first feed */
int gk_packer_set_output(gk_packer* pk, gk_mate* mates_out, int64_t capacity);
int gk_packer_records(gk_packer* pk, gk_mate* mates_out, int64_t* pair_lines_out);
/* pairs kept in the wide format: their count, then their records (2 per pair) and their pair indices (ascending) */
int gk_packer_spilled(gk_packer* pk, int64_t* n_spilled_pairs);
int gk_packer_spill_records(gk_packer* pk, gk_mate_wide* wide_out, int64_t* pair_index_out);
const char* gk_packer_string(gk_packer* pk, int64_t i);

/* ---- host ingest (no GPU, no samtools): BGZF / BAM -> SAM text lines, replacing the
 * `samtools sort -n bam -O SAM` of readBam (hisat2.py:103-110).  name_sorted != 0 orders the records
 * by query name (digit runs as numbers), READ1 before READ2, ties in file order; 0 keeps file order.
 * gk_bam_next fills the buffer with whole '\n'-terminated lines; *n_written == 0 marks the end. */
typedef struct gk_bam gk_bam;
int gk_bam_open(const char* path, int32_t name_sorted, gk_bam** out);
int gk_bam_close(gk_bam* bam);
int gk_bam_info(gk_bam* bam, int64_t* n_records, int64_t* header_bytes, int32_t* n_ref);
int gk_bam_header(gk_bam* bam, char* text_out, int64_t capacity);
int gk_bam_next(gk_bam* bam, char* text_out, int64_t capacity, int64_t* n_written);
/* every record, in output order, straight into a packer (created with gk_packer_create): the same pairs
 * and gk_mate records as feeding the rendered text to gk_packer_feed, without producing the text. */
int gk_bam_pack(gk_bam* bam, struct gk_packer* packer);
/* SAM text (header lines + alignment lines) -> BGZF-compressed BAM at `path`: the native form of
 * saveReadsToBam / samtobam (hisat2.py:869-901, `samtools sort`).  coordinate_sort != 0 orders the
 * records by (reference, position), stable, unmapped last, and writes the index `{path}.bai` next to the
 * file (`samtools index`, utils.samtobam). */
int gk_bam_write(const char* path, const char* sam_text, int64_t n_bytes, int32_t coordinate_sort);
/* the same for selected lines (0-based line numbers of `sam_text`, in the given order) under the '@'
 * lines of `header_text`: the .bam / .no_multi.bam rewrites of the filter-passing pairs
 * (hisat2.saveReadsToBam 880-901) without assembling their text first. */
int gk_bam_write_lines(const char* path, const char* header_text, int64_t n_header, const char* sam_text,
                       int64_t n_bytes, const int64_t* line_idx, int64_t n_lines, int32_t coordinate_sort);
/* Appends the "reads" array of a .variant.json to the file at `path`, as json.dump would write
 * [dataclasses.asdict(PairRead), ...] (hisat2.writeReadsAndVariantsData 847-856): row i < n_rows has
 * l_sam / r_sam = lines pair_lines[2 src[i]] / pair_lines[2 src[i] + 1] of `sam_text`, multiple = nh[i],
 * backbone = genes[gene_of[i]], and the names of ids over the row's four CSR segments off[4i .. 4i+4]
 * (lpv, rpv, lnv, rnv). */
int gk_json_write_reads(const char* path, const char* sam_text, int64_t n_bytes, const int64_t* pair_lines,
                        int64_t n_pairs, const int64_t* src, int64_t n_rows, const uint32_t* off,
                        const uint32_t* ids, const char* const* names, int64_t n_names,
                        const char* const* genes, int32_t n_genes, const uint8_t* gene_of, const uint8_t* nh);
/* base counts per reference position, replacing pileup.getPileupBaseRatio (pileup.py:57-81: parse of
 * `samtools mpileup -a`; the defaults modelled are listed in csrc/gk_bamread.cpp).  gene_off[g] = first
 * position of reference g (header order) in the concatenated position space, gene_off[n_gene] = total;
 * counts_out uint32 [total][6] = A, C, G, T, N, '*'. */
int gk_bam_pileup(gk_bam* bam, const int64_t* gene_off, int32_t n_gene, uint32_t* counts_out);

/* ---- read depth: replaces `samtools depth -aa {name}.no_multi.bam` (samtools_utils.py:9-14).
 * Depth of every backbone position from the M runs of the filter-passing pairs of a tabulation made
 * by gk_tabulate (NH == 1 only unless `multiple`).  gene_off[g] = start of backbone g in the
 * concatenated position space, gene_off[n_gene] = total; depth_out holds gene_off[n_gene] values. */
int gk_depth(gk_ctx* ctx, gk_tab* tab, gk_dptr d_mates, int32_t multiple, const int64_t* gene_off,
             int32_t n_gene, uint32_t* depth_out);

/* the `samtools depth -aa` text of that table ("gene\tpos\tdepth", positions 1-based, no header), as
 * samtools_utils.readSamtoolsDepth / kir_cn.predictSamplesCN read it (samtools_utils.py:17-22) */
int gk_depth_write_tsv(const char* path, const char* const* genes, const int64_t* gene_off, int32_t n_genes,
                       const uint32_t* depth);

/* ---- copy-number model (LCND / "CNgroup"): cn_model.py:124-204.
 * gk_cn_fit:    loglik_out[j] = sum_x log(max_n N(x; bases[j]*n, dev[n]) * space + 1e-9) * density[x]
 *               for n = first_cn .. first_cn + n_cn - 1 (CNgroup.fit 153-164, calcCNGroupProb 179-204)
 * gk_cn_assign: cn_of_bin_out[x] = argmax_n of the same table at `base` (assignCN 171-177). */
int gk_cn_fit(gk_ctx* ctx, const double* x, const double* density, int32_t bins, const double* bases,
              int32_t n_bases, const double* dev, int32_t n_cn, int32_t first_cn, double space,
              double* loglik_out);
int gk_cn_assign(gk_ctx* ctx, const double* x, int32_t bins, double base, const double* dev, int32_t n_cn,
                 int32_t first_cn, double space, int32_t* cn_of_bin_out);

/* ---- cohort collective (one rank per GPU, RCCL over xGMI): the pooling of the gene depths of all samples
 * before the single copy-number fit of `--cn-cohort` (kir_cn.py:61, 167-177; main.py:572-589) when the
 * samples are sharded over ranks.  librccl.so is opened on first use.  One rank calls gk_comm_unique_id and
 * hands the 128 bytes to the others (host-side rendezvous, kir_graph_amd/comm.py); all call gk_comm_create.
 *   gk_allgather_f64:      recv[r*n .. (r+1)*n) = send of rank r, host buffers, on every rank
 *   gk_allreduce_max_f64:  element-wise maximum over the ranks, in place (bench.py: max-over-ranks time)
 *   gk_comm_barrier:       all ranks arrived and the context's stream has drained */
typedef struct gk_comm gk_comm;
int gk_comm_unique_id(void* id_out, size_t capacity);
int gk_comm_create(gk_ctx* ctx, const void* id, size_t id_bytes, int32_t rank, int32_t world, gk_comm** out);
int gk_comm_destroy(gk_comm* comm);
int gk_allgather_f64(gk_comm* comm, const double* send, double* recv, int64_t n);
int gk_allreduce_max_f64(gk_comm* comm, double* inout, int64_t n);
int gk_comm_barrier(gk_comm* comm);

/* ---- the packed records of a sample in a compact form, for the time between its depth and its typing: with --cn-cohort a
 * sample is typed only after the pooled copy-number fit of the whole cohort (main.py:572-589, kir_cn.py:167-186), and what
 * waits in HBM meanwhile is this -- per mate the words it uses (header, CIGAR operations, mismatches, inserted-string
 * ids: hisat2.py:228-276 is what a record holds) behind uint32 word offsets [n_mates + 1], ~30 bytes per mate instead of
 * 128 -- not the tabulation (~1 GB per 5 M reads).  *d_compact_out: a block of the context's pool (gk_free), *bytes_out
 * its size.  gk_mates_expand writes the records back (unused parts zero); gk_tabulate on them gives the same lists. */
int gk_mates_compact(gk_ctx* ctx, gk_dptr d_mates, int64_t n_mates, gk_dptr* d_compact_out, int64_t* bytes_out);
int gk_mates_expand(gk_ctx* ctx, gk_dptr d_compact, int64_t n_mates, gk_dptr d_mates_out);
/* The same compact form made on the HOST (no device needed): what crosses PCIe for a sample is then ~30 bytes per mate
 * instead of 128, and gk_mates_expand writes the 128-byte records where the tabulation reads them.  gk_mates_compact_size:
 * the words the mates take (offsets not counted); gk_mates_compact_host: `out` = uint32 [n_mates + 1 + words], the layout
 * of gk_mates_compact.  n_threads: host threads for the two passes over the records. */
int gk_mates_compact_size(const gk_mate* mates, int64_t n_mates, int32_t n_threads, int64_t* n_words_out);
int gk_mates_compact_host(const gk_mate* mates, int64_t n_mates, int32_t n_threads, uint32_t* out, int64_t capacity_words);

/* ---- pinned host memory for the packed records of a sample (the packer's output on its way to HBM) and a
 * host-to-device copy that is only queued on the context's stream (gk_h2d waits for it). */
int gk_host_alloc(size_t bytes, void** out);
int gk_host_free(void* p);
int gk_h2d_async(gk_ctx* ctx, gk_dptr dst, const void* src, size_t bytes);

#ifdef __cplusplus
}
#endif
#endif /* GRAPHKIR_HIP_H */

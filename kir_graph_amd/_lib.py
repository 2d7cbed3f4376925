"""
ctypes binding of ``libgraphkir_hip.so`` (C ABI: ``include/graphkir_hip.h``).

There is deliberately no fallback: if the library is missing, or no HIP device
is visible when a context is created, the typing path raises.
"""
from __future__ import annotations

import ctypes as C
import threading
import os
from pathlib import Path

import numpy as np

LIB_NAME = "libgraphkir_hip.so"
_lib = None


class GkError(RuntimeError):
    """Error reported by the native library."""


class GkNoDevice(GkError):
    """No HIP device: the typing path cannot run (there is no CPU fallback)."""


# gk_mate (include/graphkir_hip.h): 128 bytes, mm is an array of 16 {u16 ref_off, u8 base, u8 rsv}
_MM = np.dtype([("ref_off", "<u2"), ("base", "u1"), ("rsv", "u1")])
MATE_DTYPE = np.dtype([
    ("pos0", "<u4"), ("flag", "<u2"), ("ref", "u1"), ("nh", "u1"),
    ("nm", "u1"), ("n_cig", "u1"), ("n_mm", "u1"), ("n_ins", "u1"),
    ("cig", "<u2", (14,)), ("mm", _MM, (16,)), ("ins", "<u4", (6,)),
])
assert MATE_DTYPE.itemsize == 128

# gk_mate_wide: 2048 bytes, the rarely used second format for pairs that do not fit gk_mate
MATE_WIDE_DTYPE = np.dtype([
    ("pos0", "<u4"), ("flag", "<u2"), ("ref", "u1"), ("nh", "u1"), ("nm", "u1"), ("rsv0", "u1"),
    ("n_cig", "<u2"), ("n_mm", "<u2"), ("n_ins", "<u2"),
    ("cig", "<u4", (128,)), ("mm", "<u4", (256,)), ("ins", "<u4", (120,)), ("rsv1", "u1", (16,)),
])
assert MATE_WIDE_DTYPE.itemsize == 2048
SPILLED = 0xFF

CIG_M, CIG_I, CIG_D, CIG_S = 0, 1, 2, 4
NM_ABSENT = 255
MAX_CIG, MAX_MM, MAX_INS, MAX_EV = 14, 16, 6, 22


class GeneJob(C.Structure):
    """``gk_gene_job`` (include/graphkir_hip.h): one gene of ``gk_sample_search``."""
    _fields_ = [
        ("d_rows", C.c_uint64), ("n_rows", C.c_int64), ("d_mask", C.c_uint64), ("d_L", C.c_uint64),
        ("d_miss8", C.c_uint64), ("ldm", C.c_int64), ("d_msum", C.c_uint64), ("d_flags", C.c_uint64),
        ("d_lidx", C.c_uint64),
        ("vbeg", C.c_int32), ("vend", C.c_int32), ("words", C.c_int32), ("n_allele", C.c_int32),
        ("n_steps", C.c_int32), ("top_n", C.c_int32), ("bound_ok", C.c_int32), ("passes", C.c_int32),
        ("indexed", C.c_int32), ("patches", C.c_int32),
        ("table_of", C.c_int32), ("n_step_cols", C.c_int32), ("step_cols", C.c_void_p), ("step_cols_off", C.c_void_p),
    ]


class EmJob(C.Structure):
    """``gk_em_job`` (include/graphkir_hip.h): one gene of ``gk_sample_em``."""
    _fields_ = [
        ("d_rows", C.c_uint64), ("n_rows", C.c_int64), ("d_mask", C.c_uint64),
        ("vbeg", C.c_int32), ("vend", C.c_int32), ("words", C.c_int32), ("n_allele", C.c_int32),
        ("n_distinct", C.c_int32), ("iterations", C.c_int32),
    ]


class TabInfo(C.Structure):
    _fields_ = [
        ("n_pairs", C.c_int64), ("n_valid", C.c_int64), ("n_ids", C.c_int64),
        ("n_novel", C.c_int32), ("err_flags", C.c_int32),
        ("d_pair_src", C.c_uint64), ("d_off", C.c_uint64), ("d_ids", C.c_uint64),
        ("d_pair_gene", C.c_uint64), ("d_pair_nh", C.c_uint64), ("d_novel_key", C.c_uint64),
    ]


def libPath() -> Path:
    return Path(__file__).resolve().parent / LIB_NAME


_SIGS = {
    "gk_abi_version": (C.c_int, []),
    "gk_last_error": (C.c_char_p, []),
    "gk_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "gk_ctx_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "gk_ctx_create_priority": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "gk_ctx_destroy": (C.c_int, [C.c_void_p]),
    "gk_sync": (C.c_int, [C.c_void_p]),
    "gk_device_memory": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "gk_malloc": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_uint64)]),
    "gk_free": (C.c_int, [C.c_void_p, C.c_uint64]),
    "gk_memset": (C.c_int, [C.c_void_p, C.c_uint64, C.c_int, C.c_size_t]),
    "gk_h2d": (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p, C.c_size_t]),
    "gk_d2h": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_size_t]),
    "gk_d2d": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_size_t]),
    "gk_timer_start": (C.c_int, [C.c_void_p]),
    "gk_timer_stop_ms": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    "gk_prof_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "gk_prof_kernel_count": (C.c_int, []),
    "gk_prof_kernel_name": (C.c_char_p, [C.c_int]),
    "gk_prof_collect": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "gk_index_create": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.POINTER(C.c_void_p)]),
    "gk_index_destroy": (C.c_int, [C.c_void_p]),
    "gk_tabulate": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int64, C.POINTER(C.c_void_p)]),
    "gk_tabulate_spilled": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int64, C.c_uint64, C.c_uint64,
                                     C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_void_p)]),
    "gk_tabulate_corrected": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int64, C.c_uint64, C.c_uint64,
                                        C.POINTER(C.c_void_p)]),
    "gk_tab_from_csr": (C.c_int, [C.c_void_p, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.POINTER(C.c_void_p)]),
    "gk_tab_get_info": (C.c_int, [C.c_void_p, C.POINTER(TabInfo)]),
    "gk_tab_destroy": (C.c_int, [C.c_void_p]),
    "gk_select_gene": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_uint64, C.POINTER(C.c_int64)]),
    "gk_select_nonempty": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int64, C.c_uint64, C.c_uint64,
                                     C.POINTER(C.c_int64)]),
    "gk_variant_count": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int64, C.c_uint64, C.c_uint64]),
    "gk_variant_count_range": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int64, C.c_uint64, C.c_uint64,
                                         C.c_int32, C.c_int32]),
    "gk_variant_surviving": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_int64, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.POINTER(C.c_int64)]),
    "gk_variant_surviving_gene": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_int32, C.c_int32, C.c_int32,
                                            C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64)]),
    "gk_sample_prepare": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p]),
    "gk_sample_prepare_all": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p,
                                       C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64), C.c_void_p]),
    "gk_variant_correct": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64]),
    "gk_compat": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int64, C.c_uint64, C.c_int32, C.c_int32,
                            C.c_uint64, C.c_int32, C.c_int32, C.c_int32, C.c_uint64, C.c_uint64, C.c_uint64]),
    "gk_lut_create": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_void_p)]),
    "gk_lut_destroy": (C.c_int, [C.c_void_p]),
    "gk_lut_collect": (C.c_int, [C.c_void_p, C.c_uint64, C.c_int64]),
    "gk_lut_pending": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "gk_lut_export": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "gk_lut_define": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "gk_lut_apply": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_int64]),
    "gk_bam_open": (C.c_int, [C.c_char_p, C.c_int32, C.POINTER(C.c_void_p)]),
    "gk_bam_close": (C.c_int, [C.c_void_p]),
    "gk_bam_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int32)]),
    "gk_bam_header": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int64]),
    "gk_bam_pileup": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "gk_bam_write": (C.c_int, [C.c_char_p, C.c_char_p, C.c_int64, C.c_int32]),
    "gk_bam_write_lines": (C.c_int, [C.c_char_p, C.c_char_p, C.c_int64, C.c_char_p, C.c_int64, C.c_void_p, C.c_int64,
                                     C.c_int32]),
    "gk_json_write_reads": (C.c_int, [C.c_char_p, C.c_char_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int32,
                                      C.c_void_p, C.c_void_p]),
    "gk_bam_pack": (C.c_int, [C.c_void_p, C.c_void_p]),
    "gk_bam_next": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]),
    "gk_compat_log": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int64, C.c_uint64, C.c_int32, C.c_int32,
                                C.c_uint64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_uint64]),
    "gk_maxsum": (C.c_int, [C.c_void_p, C.c_uint64, C.c_int64, C.c_int64, C.c_void_p, C.c_int32, C.c_int32,
                            C.c_void_p, C.c_int32, C.c_void_p]),
    "gk_fraction": (C.c_int, [C.c_void_p, C.c_uint64, C.c_int64, C.c_int64, C.c_void_p, C.c_int32, C.c_int32,
                              C.c_void_p]),
    "gk_setmax": (C.c_int, [C.c_void_p, C.c_uint64, C.c_int64, C.c_int64, C.c_void_p, C.c_int32, C.c_int32,
                            C.c_uint64]),
    "gk_packer_create": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.POINTER(C.c_void_p)]),
    "gk_packer_destroy": (C.c_int, [C.c_void_p]),
    "gk_packer_feed": (C.c_int, [C.c_void_p, C.c_char_p, C.c_size_t, C.c_int32]),
    "gk_packer_counts": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                   C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "gk_packer_error": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int64)]),
    "gk_packer_records": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "gk_packer_set_output": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]),
    "gk_packer_spilled": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
    "gk_packer_spill_records": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "gk_packer_string": (C.c_char_p, [C.c_void_p, C.c_int64]),
    "gk_depth": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]),
    "gk_depth_write_tsv": (C.c_int, [C.c_char_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "gk_cn_fit": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p,
                            C.c_int32, C.c_int32, C.c_double, C.c_void_p]),
    "gk_cn_assign": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_double, C.c_void_p, C.c_int32, C.c_int32,
                               C.c_double, C.c_void_p]),
    "gk_em_sets": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int64, C.c_int32, C.c_int32, C.c_uint64,
                             C.c_int32, C.c_uint64]),
    "gk_em_distinct": (C.c_int, [C.c_void_p, C.c_uint64, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                 C.POINTER(C.c_int32)]),
    "gk_compat_log_miss": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int64, C.c_uint64, C.c_int32, C.c_int32,
                                     C.c_uint64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_uint64, C.c_uint64,
                                     C.c_int64, C.c_uint64]),
    "gk_miss_colsum": (C.c_int, [C.c_void_p, C.c_uint64, C.c_int64, C.c_int32, C.c_uint64]),
    "gk_compat_index": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int64, C.c_uint64, C.c_int32, C.c_int32,
                                  C.c_uint64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_uint64, C.c_uint64,
                                  C.c_int64, C.c_uint64]),
    "gk_expand_index": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int64, C.c_int64, C.c_int32, C.c_uint64,
                                  C.c_int64]),
    "gk_bound_step": (C.c_int, [C.c_void_p, C.c_uint64, C.c_int64, C.c_int64, C.c_uint64, C.c_void_p, C.c_int32,
                                C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p,
                                C.c_void_p, C.c_void_p]),
    "gk_setsum": (C.c_int, [C.c_void_p, C.c_uint64, C.c_int64, C.c_int64, C.c_void_p, C.c_int32, C.c_int32,
                            C.c_void_p, C.c_void_p]),
    "gk_search_run": (C.c_int, [C.c_void_p, C.c_uint64, C.c_int64, C.c_int64, C.c_int32, C.c_uint64, C.c_int64,
                                C.c_uint64, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                C.POINTER(C.c_void_p)]),
    "gk_sample_search": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p,
                                   C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gk_lut_resolve": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                 C.POINTER(C.c_int32)]),
    "gk_sample_prepare_exon": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p,
                                         C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64)]),
    "gk_sample_em": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_double, C.c_void_p, C.c_void_p]),
    "gk_mates_compact": (C.c_int, [C.c_void_p, C.c_uint64, C.c_int64, C.POINTER(C.c_uint64), C.POINTER(C.c_int64)]),
    "gk_mates_expand": (C.c_int, [C.c_void_p, C.c_uint64, C.c_int64, C.c_uint64]),
    "gk_mates_compact_size": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.POINTER(C.c_int64)]),
    "gk_mates_compact_host": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_int64]),
    "gk_compat_patch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int64, C.c_int32, C.c_uint64, C.c_int64, C.c_uint64]),
    "gk_lut_resolve_stored": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                        C.POINTER(C.c_int32)]),
    "gk_lut_known": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32)]),
    "gk_search_steps": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32)]),
    "gk_search_info": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int64),
                                 C.POINTER(C.c_int32)]),
    "gk_search_copy": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gk_search_export": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gk_search_colsum": (C.c_int, [C.c_void_p, C.c_void_p]),
    "gk_search_log": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]),
    "gk_search_destroy": (C.c_int, [C.c_void_p]),
    "gk_site_verdict_tallies": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_int64, C.c_int32, C.POINTER(C.c_int32)]),
    "gk_site_verdict_genes": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "gk_site_verdict": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32,
                                  C.POINTER(C.c_int32)]),
    "gk_comm_unique_id": (C.c_int, [C.c_void_p, C.c_size_t]),
    "gk_comm_create": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]),
    "gk_comm_destroy": (C.c_int, [C.c_void_p]),
    "gk_allgather_f64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]),
    "gk_allreduce_max_f64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]),
    "gk_comm_barrier": (C.c_int, [C.c_void_p]),
    "gk_host_alloc": (C.c_int, [C.c_size_t, C.POINTER(C.c_void_p)]),
    "gk_host_free": (C.c_int, [C.c_void_p]),
    "gk_h2d_async": (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p, C.c_size_t]),
    "gk_em_run": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                            C.c_double, C.c_void_p, C.POINTER(C.c_int32)]),
}

EXPORTED = sorted(_SIGS)

# numpy.argsort as the callback of gk_search_run (the reference's own sort: its order among equal values is part
# of the result).  ctypes takes the interpreter lock for the few microseconds of the call.
ARGSORT_FN = C.CFUNCTYPE(C.c_int, C.POINTER(C.c_double), C.c_int64, C.POINTER(C.c_int64))


def _numpy_argsort(values, n, order_out):
    try:
        v = np.ctypeslib.as_array(values, shape=(n,))
        np.ctypeslib.as_array(order_out, shape=(n,))[:] = np.argsort(v)
        return 0
    except Exception:      # never let an exception cross the C frame
        return 1


NUMPY_ARGSORT = ARGSORT_FN(_numpy_argsort)

# numpy.log10 as the callback of gk_lut_resolve / gk_sample_search (typing_mulit_allele.py:263): the bits of the
# log-likelihoods are numpy's on the machine at hand
LOG10_FN = C.CFUNCTYPE(C.c_int, C.POINTER(C.c_double), C.c_int64, C.POINTER(C.c_double))


def _numpy_log10(values, n, out):
    try:
        v = np.ctypeslib.as_array(values, shape=(n,))
        with np.errstate(divide="ignore"):
            np.ctypeslib.as_array(out, shape=(n,))[:] = np.log10(v)
        return 0
    except Exception:      # never let an exception cross the C frame
        return 1


NUMPY_LOG10 = LOG10_FN(_numpy_log10)


def lib():
    """Load the native library once; raise if it has not been built."""
    global _lib
    if _lib is None:
        path = libPath()
        if not path.exists():
            raise GkError(
                f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). The typing path has no CPU fallback.")
        handle = C.CDLL(str(path))
        for name, (res, args) in _SIGS.items():
            fn = getattr(handle, name)
            fn.restype, fn.argtypes = res, args
        if handle.gk_abi_version() != 1:
            raise GkError("libgraphkir_hip.so ABI version mismatch")
        _lib = handle
    return _lib


def check(rc: int) -> None:
    if rc == 0:
        return
    msg = lib().gk_last_error().decode(errors="replace")
    if rc == -1:
        raise GkNoDevice(msg)
    if rc == -4:
        raise AssertionError(msg)
    raise GkError(f"[{rc}] {msg}")


def _np_ptr(a: np.ndarray) -> C.c_void_p:
    return C.c_void_p(a.ctypes.data)


class DeviceBuffer:
    """A typed allocation in HBM."""

    def __init__(self, dev: "Device", shape, dtype):
        self.dev = dev
        self.shape = tuple(int(s) for s in (shape if isinstance(shape, (tuple, list)) else (shape,)))
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize
        p = C.c_uint64()
        check(lib().gk_malloc(dev.ctx, max(self.nbytes, 16), C.byref(p)))
        self.ptr = p.value

    @property
    def size(self) -> int:
        return int(np.prod(self.shape, dtype=np.int64))

    def upload(self, a: np.ndarray) -> "DeviceBuffer":
        a = np.ascontiguousarray(a, dtype=self.dtype)
        assert a.nbytes == self.nbytes, (a.shape, self.shape)
        if a.nbytes:
            check(lib().gk_h2d(self.dev.ctx, self.ptr, _np_ptr(a), a.nbytes))
        return self

    def download(self, count: int | None = None, offset: int = 0) -> np.ndarray:
        n = self.size - offset if count is None else count
        out = np.empty(n, dtype=self.dtype)
        if n:
            check(lib().gk_d2h(self.dev.ctx, _np_ptr(out), self.ptr + offset * self.dtype.itemsize, out.nbytes))
        return out if count is not None or len(self.shape) == 1 else out.reshape(self.shape)

    def zero(self) -> "DeviceBuffer":
        check(lib().gk_memset(self.dev.ctx, self.ptr, 0, self.nbytes))
        return self

    def free(self) -> None:
        if self.ptr:
            lib().gk_free(self.dev.ctx, self.ptr)
            self.ptr = 0

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class DeviceSlice(DeviceBuffer):
    """Part of another buffer (not owned: freeing it does nothing; the parent must outlive it)."""

    def __init__(self, parent: DeviceBuffer, offset_items: int, count: int, dev: "Device | None" = None):
        self.dev = dev or parent.dev
        self.shape = (int(count),)
        self.dtype = parent.dtype
        self.nbytes = int(count) * self.dtype.itemsize
        self.ptr = parent.ptr + int(offset_items) * self.dtype.itemsize
        self._parent = parent

    def free(self) -> None:
        self.ptr = 0


class Device:
    """One HIP device context (one process drives one GPU)."""

    def __init__(self, ordinal: int | None = None, urgent: bool = False):
        """``urgent``: a stream of the device's highest priority (short, latency-bound work next to long kernels)."""
        if ordinal is None:
            ordinal = int(os.environ.get("LOCAL_RANK", "0"))
            n = deviceCount()
            ordinal = ordinal % n if n else 0
        ctx = C.c_void_p()
        check(lib().gk_ctx_create_priority(ordinal, int(urgent), C.byref(ctx)))
        self.ctx = ctx
        self.ordinal = ordinal
        self._urgent: "Device | None" = None
        self.call_log: list[tuple] | None = None   # set to [] to record launch geometry (bench roofline)
        self._workers: dict[int, "Device"] = {}
        self._workers_lock = threading.Lock()
        Device.instances.append(self)

    instances: list["Device"] = []

    def worker(self, k: int, urgent: bool = False) -> "Device":
        """k-th extra context (own stream + allocator) on the same GPU, for per-gene host threads (``urgent`` counts
        when the context is made: see ``Device``)."""
        with self._workers_lock:        # made on first use: the lanes of a process use a few of the slots they own
            w = self._workers.get(k)
            if w is None:
                w = self._workers[k] = Device(self.ordinal, urgent=urgent)
            return w

    def urgent(self) -> "Device":
        """This context's high-priority sibling (made on first use): for the short preamble of a sample, whose small
        kernels and waits would otherwise sit behind the long kernels of the samples typed next to it."""
        if self._urgent is None:
            self._urgent = Device(self.ordinal, urgent=True)
        return self._urgent

    def alloc(self, shape, dtype) -> DeviceBuffer:
        return DeviceBuffer(self, shape, dtype)

    def put(self, a: np.ndarray) -> DeviceBuffer:
        a = np.ascontiguousarray(a)
        return DeviceBuffer(self, a.shape, a.dtype).upload(a)

    def view(self, ptr: int, count: int, dtype) -> np.ndarray:
        """Download ``count`` items of ``dtype`` from a raw device address (large ones into pinned memory)."""
        out = pinnedEmpty(count, dtype) if count * np.dtype(dtype).itemsize >= (8 << 20) else np.empty(count, dtype=dtype)
        if count:
            check(lib().gk_d2h(self.ctx, _np_ptr(out), ptr, out.nbytes))
        return out

    def sync(self) -> None:
        check(lib().gk_sync(self.ctx))

    def memory(self) -> tuple[int, int, int]:
        """(free, total) bytes of the device as the runtime reports them, bytes this process's pools hold idle."""
        f, t, c = C.c_int64(), C.c_int64(), C.c_int64()
        check(lib().gk_device_memory(self.ctx, C.byref(f), C.byref(t), C.byref(c)))
        return int(f.value), int(t.value), int(c.value)

    def timerStart(self) -> None:
        check(lib().gk_timer_start(self.ctx))

    def timerStopMs(self) -> float:
        ms = C.c_float()
        check(lib().gk_timer_stop_ms(self.ctx, C.byref(ms)))
        return float(ms.value)

    def profEnable(self, on: bool = True) -> None:
        check(lib().gk_prof_enable(self.ctx, int(on)))

    def profCollect(self) -> dict[str, tuple[int, float]]:
        """{kernel name: (launches, total ms)} of the spans recorded since the last call."""
        n = lib().gk_prof_kernel_count()
        launches = np.zeros(n, dtype=np.int64)
        total = np.zeros(n, dtype=np.float64)
        check(lib().gk_prof_collect(self.ctx, launches.ctypes.data, total.ctypes.data))
        return {lib().gk_prof_kernel_name(i).decode(): (int(launches[i]), float(total[i]))
                for i in range(n) if launches[i]}

    def close(self) -> None:
        if self.ctx:
            lib().gk_ctx_destroy(self.ctx)
            self.ctx = None


_pinned_ok: bool | None = None
# Pinning and unpinning hundreds of megabytes costs tens of milliseconds each way and holds the runtime's lock
# meanwhile (measured: 48 ms + 44 ms for a sample's id array), so blocks go back to a pool instead of to the
# runtime: a few size classes per order of magnitude, 4 GB of idle blocks at most.
_pool_lock = threading.Lock()
_pool: dict[int, list[int]] = {}
_pool_idle = 0


def _pinnedClass(nbytes: int) -> int:
    """Size class of a request: the next multiple of an eighth of its power of two (at least 1 MiB)."""
    step = max(1 << 20, 1 << max(0, nbytes.bit_length() - 4))
    return (nbytes + step - 1) // step * step


def _pinnedTake(nbytes: int) -> tuple[int, int]:
    """(address, size class) of a pinned block of at least ``nbytes``; address 0 when the runtime has none."""
    global _pool_idle
    cls = _pinnedClass(nbytes)
    with _pool_lock:
        free = _pool.get(cls)
        if free:
            _pool_idle -= cls
            return free.pop(), cls
    p = C.c_void_p()
    if lib().gk_host_alloc(cls, C.byref(p)) != 0 or not p.value:
        return 0, cls
    return p.value, cls


def _pinnedGive(address: int, cls: int) -> None:
    global _pool_idle
    limit = 4 << 30
    with _pool_lock:
        if _pool_idle + cls <= limit:
            _pool.setdefault(cls, []).append(address)
            _pool_idle += cls
            return
    lib().gk_host_free(C.c_void_p(address))


def pinnedEmpty(count: int, dtype) -> np.ndarray:
    """``np.empty(count, dtype)`` in pinned host memory when a GPU is there (the packed records of a sample on
    their way to HBM: the copy then runs at PCIe speed instead of through the runtime's pageable staging);
    plain memory otherwise.  The block goes back to the pool when the array is garbage collected."""
    global _pinned_ok
    dtype = np.dtype(dtype)
    nbytes = int(count) * dtype.itemsize
    if _pinned_ok is None:
        _pinned_ok = deviceCount() > 0
    if not _pinned_ok or nbytes < (1 << 20):
        return np.empty(count, dtype=dtype)
    address, cls = _pinnedTake(nbytes)
    if not address:
        return np.empty(count, dtype=dtype)
    import weakref
    raw = (C.c_uint8 * nbytes).from_address(address)
    arr = np.frombuffer(raw, dtype=dtype, count=count)
    weakref.finalize(raw, _pinnedGive, address, cls)     # arr keeps `raw` alive through its base
    return arr


def deviceCount() -> int:
    n = C.c_int()
    rc = lib().gk_device_count(C.byref(n))
    return int(n.value) if rc == 0 else 0

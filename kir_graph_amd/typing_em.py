"""
EM ("report") strategy on the GPU -- drop-in for ``graphkir/typing_em.py``.

``hisat2TypingPerGene`` (191-215) becomes: candidate allele bit sets per read on the device
(``gk_em_sets`` = getCandidateAllelePerRead 68-87 + getMostFreqAllele 90-104), distinct sets with
multiplicities, SQUAREM EM in one workgroup (``gk_em_run`` = hisatEMnp 107-188).

Deviation (documented in DESIGN.md): ties in abundance are ordered by allele name here; the
reference orders them by the iteration order of a Python ``set`` (typing_em.py:213), which changes
from process to process.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from ._lib import DeviceBuffer, check, lib
from .engine import Tabulation


@dataclass
class Hisat2AlleleResult:
    """Abundance record of one allele (typing_em.py:21-28)."""

    allele: str
    count: int
    prob: float
    cn: int = 0


def candidateSets(tab: Tabulation, rows: DeviceBuffer, n_rows: int, vbeg: int, vend: int, mask: DeviceBuffer,
                  words: int) -> np.ndarray:
    """uint32 [n_rows][words] candidate-allele bit sets of the given rows."""
    out = tab.dev.alloc((max(n_rows, 1), words), np.uint32)
    check(lib().gk_em_sets(tab.dev.ctx, tab.handle, rows.ptr, n_rows, vbeg, vend, mask.ptr, words, out.ptr))
    sets = out.download()[:n_rows]
    out.free()
    return sets


def candidateSetsDistinct(tab: Tabulation, rows: DeviceBuffer, n_rows: int, vbeg: int, vend: int,
                          mask: DeviceBuffer, words: int) -> tuple[np.ndarray, np.ndarray]:
    """Distinct candidate-allele bit sets of the rows (ascending, like ``np.unique(axis=0)``) and their
    multiplicities; sets and grouping stay on the device, only the distinct ones come back."""
    buf = tab.dev.alloc((max(n_rows, 1), words), np.uint32)
    check(lib().gk_em_sets(tab.dev.ctx, tab.handle, rows.ptr, n_rows, vbeg, vend, mask.ptr, words, buf.ptr))
    cap = 1 << 14
    while True:
        sets = np.empty((cap, words), dtype=np.uint32)
        count = np.empty(cap, dtype=np.uint32)
        n = C.c_int32()
        rc = lib().gk_em_distinct(tab.dev.ctx, buf.ptr, n_rows, words, cap, sets.ctypes.data, count.ctypes.data, C.byref(n))
        if rc == -5 and cap < max(n_rows, 1):
            cap = min(cap * 16, max(n_rows, 1))
            continue
        check(rc)
        break
    buf.free()
    sets, count = sets[:n.value], count[:n.value].astype(np.int64)
    order = np.lexsort(sets.T[::-1]) if len(sets) else np.zeros(0, dtype=np.int64)   # rows ascending, word 0 first
    return np.ascontiguousarray(sets[order]), count[order]


_MIX = (np.arange(1, 65, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)) | np.uint64(1)


def distinctSets(sets: np.ndarray) -> tuple[np.ndarray, np.ndarray]:
    """Distinct rows (ascending, like ``np.unique(axis=0)``) and their multiplicities.

    Rows are grouped through a 64-bit mix of their words; the grouping is then verified against the
    rows themselves, and the exact (slow) path is taken in the unlikely case of a collision."""
    if len(sets) == 0:
        return sets.reshape(0, sets.shape[1]), np.zeros(0, dtype=np.int64)
    with np.errstate(over="ignore"):
        h = (sets.astype(np.uint64) * _MIX[:sets.shape[1]]).sum(axis=1, dtype=np.uint64)
        h ^= h >> np.uint64(29)
    _, first, inverse, counts = np.unique(h, return_index=True, return_inverse=True, return_counts=True)
    reps = sets[first]
    if not np.array_equal(reps[inverse], sets):
        return np.unique(sets, axis=0, return_counts=True)
    uniq, back = np.unique(reps, axis=0, return_inverse=True)
    weight = np.bincount(back.ravel(), weights=counts, minlength=len(uniq)).astype(np.int64)
    return uniq, weight


def hisatEMdevice(tab: Tabulation, sets, n_allele: int, iter_max: int = 300,
                  diff_threshold: float = 0.0001) -> tuple[np.ndarray, np.ndarray, int]:
    """Abundance per allele column, read count per allele, iterations used.

    ``sets``: the per-read bit sets, or the pair (distinct sets, multiplicities)."""
    uniq, weight = sets if isinstance(sets, tuple) else distinctSets(sets)
    words = uniq.shape[1]
    bits = np.unpackbits(uniq.view(np.uint8), axis=1, bitorder="little")[:, :n_allele]
    count = (bits.astype(np.int64) * weight[:, None]).sum(axis=0)      # reads naming each allele
    keep = uniq.any(axis=1)
    uniq, weight = np.ascontiguousarray(uniq[keep]), np.ascontiguousarray(weight[keep].astype(np.float64))
    prob = np.zeros(n_allele, dtype=np.float64)
    iters = C.c_int32(0)
    if len(uniq):
        check(lib().gk_em_run(tab.dev.ctx, uniq.ctypes.data, weight.ctypes.data, len(uniq), words, n_allele,
                              iter_max, diff_threshold, prob.ctypes.data, C.byref(iters)))
    return prob, count, int(iters.value)


def hisat2TypingPerGene(tab: Tabulation, rows: DeviceBuffer, n_rows: int, vbeg: int, vend: int, mask: DeviceBuffer,
                        words: int, alleles: list[str], info: dict | None = None) -> list[Hisat2AlleleResult]:
    """Per-gene EM report.  Raises like the reference when no read names any allele.  ``info`` (optional)
    receives ``iterations`` (SQUAREM steps until the 1e-4 stop, at most 300) and ``distinct_sets``."""
    sets = candidateSetsDistinct(tab, rows, n_rows, vbeg, vend, mask, words)
    prob, count, iters = hisatEMdevice(tab, sets, len(alleles))
    if info is not None:
        info["iterations"], info["distinct_sets"] = iters, len(sets[0])
    named = np.nonzero(count)[0]
    return [Hisat2AlleleResult(allele=alleles[a], count=int(count[a]), prob=float(prob[a])) for a in named]

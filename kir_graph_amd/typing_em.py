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


def hisatEMdevice(tab: Tabulation, sets: np.ndarray, n_allele: int, iter_max: int = 300,
                  diff_threshold: float = 0.0001) -> tuple[np.ndarray, np.ndarray, int]:
    """Abundance per allele column, read count per allele, iterations used."""
    words = sets.shape[1]
    uniq, weight = np.unique(sets, axis=0, return_counts=True)
    bits = np.unpackbits(sets.view(np.uint8), axis=1, bitorder="little")[:, :n_allele]
    count = bits.sum(axis=0).astype(np.int64)
    keep = uniq.any(axis=1)
    uniq, weight = np.ascontiguousarray(uniq[keep]), np.ascontiguousarray(weight[keep].astype(np.float64))
    prob = np.zeros(n_allele, dtype=np.float64)
    iters = C.c_int32(0)
    if len(uniq):
        check(lib().gk_em_run(tab.dev.ctx, uniq.ctypes.data, weight.ctypes.data, len(uniq), words, n_allele,
                              iter_max, diff_threshold, prob.ctypes.data, C.byref(iters)))
    return prob, count, int(iters.value)


def hisat2TypingPerGene(tab: Tabulation, rows: DeviceBuffer, n_rows: int, vbeg: int, vend: int, mask: DeviceBuffer,
                        words: int, alleles: list[str]) -> list[Hisat2AlleleResult]:
    """Per-gene EM report.  Raises like the reference when no read names any allele."""
    sets = candidateSets(tab, rows, n_rows, vbeg, vend, mask, words)
    prob, count, _ = hisatEMdevice(tab, sets, len(alleles))
    named = np.nonzero(count)[0]
    return [Hisat2AlleleResult(allele=alleles[a], count=int(count[a]), prob=float(prob[a])) for a in named]

"""
Pileup base ratios and the SNP correction derived from them -- drop-in for ``graphkir/pileup.py``
(``getPileupBaseRatio`` 57-81) and for ``hisat2.errorCorrection`` (609-654).

The reference parses ``samtools mpileup -a``; here the base counts come from the native BAM reader
(``gk_bam_pileup``, which documents the mpileup defaults it models) and the correction is turned
into a per-position table that the tabulation kernels apply to every mismatch before the variant
lookup.  Off by default in the CLI, like in the reference (``main.py:149``).  samtools is not in this
image, so the counts are checked against the suite's own restatement of the same rules
(``oracle/pileup.py``), not against mpileup output: parity unpinned.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._lib import check, lib
from .index import GkIndex

BASES = "ACGTN*"
PileupCount = dict[tuple[str, int], dict[str, float]]


def pileupCounts(bam_file: str, index: GkIndex) -> tuple[np.ndarray, np.ndarray]:
    """(counts uint32 [total][6] in the index's backbone order, first position of every backbone);
    backbone lengths are the ``@SQ LN`` of the BAM header."""
    if not bam_file.endswith(".bam"):
        raise ValueError("pileup needs a BAM file (write SAM text with packed.writeBam first)")
    h = C.c_void_p()
    check(lib().gk_bam_open(bam_file.encode(), 0, C.byref(h)))
    try:
        n = C.c_int64()
        check(lib().gk_bam_info(h, None, C.byref(n), None))
        buf = C.create_string_buffer(max(int(n.value), 1))
        check(lib().gk_bam_header(h, buf, n.value))
        header = buf.raw[:n.value].decode()
        sq = [dict(f.split(":", 1) for f in line.split("\t")[1:] if ":" in f)
              for line in header.split("\n") if line.startswith("@SQ")]
        refs = [d["SN"] for d in sq]
        lengths = {d["SN"]: int(d["LN"]) for d in sq}
        ref_len = np.array([lengths[r] for r in refs], dtype=np.int64)
        ref_off = np.concatenate([[0], np.cumsum(ref_len)]).astype(np.int64)
        raw = np.zeros((int(ref_off[-1]), 6), dtype=np.uint32)
        check(lib().gk_bam_pileup(h, ref_off.ctypes.data, len(refs), raw.ctypes.data))
    finally:
        lib().gk_bam_close(h)
    gene_len = np.array([lengths.get(g, 0) for g in index.genes], dtype=np.int64)
    pos0 = np.concatenate([[0], np.cumsum(gene_len)]).astype(np.int64)
    counts = np.zeros((int(pos0[-1]), 6), dtype=np.uint32)
    for k, r in enumerate(refs):
        g = index.gene_id.get(r)
        if g is not None:
            counts[pos0[g]:pos0[g + 1]] = raw[ref_off[k]:ref_off[k + 1]]
    return counts, pos0


def ratiosOf(counts: np.ndarray, pos0: np.ndarray, genes: list[str]) -> PileupCount:
    """The reference's dictionary form: ``{(ref, pos): {"A": 0.2, "C": 0.8, "all": 30}}`` (depth > 0 only)."""
    stat: PileupCount = {}
    total = counts.sum(axis=1)
    for at in np.flatnonzero(total):
        g = int(np.searchsorted(pos0, at, side="right") - 1)
        s = int(total[at])
        entry = {BASES[j]: int(c) / s for j, c in enumerate(counts[at]) if c}
        entry["all"] = s
        stat[(genes[g], int(at - pos0[g]))] = entry
    return stat


def getPileupBaseRatio(bam_file: str, index: GkIndex) -> PileupCount:
    counts, pos0 = pileupCounts(bam_file, index)
    return ratiosOf(counts, pos0, index.genes)


def correctionTable(counts: np.ndarray) -> np.ndarray:
    """uint8 [positions][5]: for a mismatch whose read base is A, C, G, T or N, the base that
    ``hisat2.errorCorrection`` (609-654) puts in its place (0 = the read's base stays):
    depth >= 20 and the read base's share <= 0.2 -> the base with a share >= 0.8 if there is one, else N."""
    total = counts.sum(axis=1).astype(np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        ratio = counts / total[:, None]                      # the reference's v / s in float64
    deep = total >= 20
    top = ratio.argmax(axis=1)
    has_major = np.nan_to_num(ratio.max(axis=1)) >= 0.8
    replacement = np.where(has_major, np.frombuffer(BASES.encode(), dtype=np.uint8)[top], ord("N")).astype(np.uint8)
    table = np.zeros((len(counts), 5), dtype=np.uint8)
    for j in range(5):
        minority = deep & ~(np.nan_to_num(ratio[:, j]) > 0.2) & (replacement != ord(BASES[j]))
        table[minority, j] = replacement[minority]
    return table


def correctionFromRatios(pileup: PileupCount, index: GkIndex) -> tuple[np.ndarray, np.ndarray]:
    """(correction table uint8 [positions][5], first position of every backbone) from the reference's own
    dictionary form ``{(ref, pos): {"A": 0.2, "C": 0.8, "all": 30}}`` (``getPileupBaseRatio`` 57-81): what
    ``extractVariant(pairs, variants, pileup=...)`` hands to ``hisat2.errorCorrection`` (609-654) per mismatch.
    A backbone's table reaches to its last listed position; mismatches beyond have no entry and stay."""
    n_pos = np.zeros(len(index.genes), dtype=np.int64)
    for (ref, pos) in pileup:
        g = index.gene_id.get(ref)
        if g is not None and pos >= 0:
            n_pos[g] = max(n_pos[g], int(pos) + 1)
    pos0 = np.concatenate([[0], np.cumsum(n_pos)]).astype(np.int64)
    table = np.zeros((int(pos0[-1]), 5), dtype=np.uint8)
    for (ref, pos), p in pileup.items():
        g = index.gene_id.get(ref)
        if g is None or pos < 0 or not p or p["all"] < 20:
            continue
        bases = [(b, r) for b, r in p.items() if b != "all"]
        if any(r >= 0.8 for _, r in bases):
            rep = str(max(bases, key=lambda t: t[1])[0])
        else:
            rep = "N"
        if len(rep) != 1:
            raise ValueError(f"pileup entry {ref}:{pos} names {rep!r}: a base is one character")
        for j, b in enumerate("ACGTN"):
            if not p.get(b, 0) > 0.2 and rep != b:
                table[pos0[g] + int(pos), j] = ord(rep)
    return table, pos0

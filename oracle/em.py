"""
Oracle: HISAT-genotype style per-read candidate alleles + SQUAREM EM abundances.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Restates, from
``/root/reference/graphkir``:

* ``readCandidates``   <- typing_em.getCandidateAllelePerRead 68-87,
                          getMostFreqAllele 90-104, preprocessHisatReads 37-65
* ``squaremEM``        <- typing_em.hisatEMnp 107-188
* ``geneReport``       <- typing_em.hisat2TypingPerGene 191-215
* ``callByAbundance``  <- kir_typing.TypingWithReport.typingPerGene 163-195
"""
from __future__ import annotations

from collections import Counter, defaultdict
from itertools import chain

import numpy as np


def mateCandidates(pos_sets: list[list[str]], neg_sets: list[list[str]]) -> list[str]:
    """Intersection of the positive variants' allele sets minus every negative one (68-87)."""
    if not pos_sets:
        return []
    cand = set(pos_sets[0])
    for s in pos_sets[1:]:
        cand &= set(s)
    for s in neg_sets:
        cand -= set(s)
    return list(cand)


def mostFrequent(cands: list[str]) -> list[str]:
    """Alleles named most often by the two mates (90-104)."""
    cnt = Counter(cands)
    if not cnt:
        return []
    top = max(cnt.values())
    return [a for a, c in cnt.items() if c == top]


def readCandidates(reads: list[dict], alleles_of: dict[str, list[str]]) -> list[list[str]]:
    out = []
    for r in reads:
        lp = [alleles_of[v] for v in r["lpv"]]
        ln = [alleles_of[v] for v in r["lnv"]]
        rp = [alleles_of[v] for v in r["rpv"]]
        rn = [alleles_of[v] for v in r["rnv"]]
        out.append(mostFrequent(mateCandidates(lp, ln) + mateCandidates(rp, rn)))
    return out


def squaremEM(allele_per_read: list[list[str]], iter_max: int = 300,
              diff_threshold: float = 0.0001) -> tuple[dict[str, float], int]:
    """EM over a dense 0/1 read x allele matrix with SQUAREM acceleration (107-188)."""
    names = sorted(set(chain.from_iterable(allele_per_read)))
    col = {a: i for i, a in enumerate(names)}
    S = np.zeros((len(allele_per_read), len(names)))
    for i, al in enumerate(allele_per_read):
        for a in al:
            S[i, col[a]] = 1
    unit = np.ones(len(names))

    def step(p: np.ndarray) -> np.ndarray:
        w = p * S
        tot = w.sum(axis=1)[:, None]
        w = np.divide(w, tot, out=np.zeros(w.shape), where=tot != 0)
        w /= unit
        w = w.sum(axis=0)
        return w / w.sum()

    p = step(np.ones(len(names)))
    iters = 0
    for iters in range(iter_max):
        p1 = step(p)
        p2 = step(p1)
        r = p1 - p
        v = p2 - p1 - r
        rs, vs = (r ** 2).sum(), (v ** 2).sum()
        if vs > 0.0:
            g = -np.sqrt(rs / vs)
            p3 = np.maximum(p - r * g * 2 + v * g ** 2, 0)
            p1 = step(p3)
        if np.abs(p - p1).sum() <= diff_threshold:
            break
        p = p1
    return dict(zip(names, p)), iters


def geneReport(reads: list[dict], alleles_of: dict[str, list[str]]) -> list[dict]:
    """Per-gene abundance/count table (191-215)."""
    per_read = readCandidates(reads, alleles_of)
    prob, _ = squaremEM(per_read)
    count = Counter(chain.from_iterable(per_read))
    return [{"allele": a, "count": count[a], "prob": prob[a], "cn": 0}
            for a in prob.keys() | count.keys()]


def callByAbundance(report: list[dict], cn: int) -> list[str]:
    """Greedy CN allocation by abundance (kir_typing.py:176-195)."""
    report.sort(key=lambda e: -e["prob"])
    unit = 1 / cn
    called = []
    for e in report:
        k = max(1, round(e["prob"] / unit))
        called.extend([e["allele"]] * min(cn, k))
        e["cn"] = k
        cn -= k
        if cn <= 0:
            break
    return called


class ReportTyper:
    """TypingWithReport (kir_typing.py:153-204) on an in-memory tabulation."""

    def __init__(self, data: dict):
        reads = [r for r in data["reads"] if r["multiple"] == 1]
        self.alleles_of = {v.id: v.allele for v in data["variants"]}
        self.gene_reads: dict[str, list[dict]] = defaultdict(list)
        for r in reads:
            self.gene_reads[r["backbone"]].append(r)
        self.results: dict[str, list[dict]] = {}

    def typingPerGene(self, gene: str, cn: int) -> tuple[list[str], int]:
        report = geneReport(self.gene_reads[gene], self.alleles_of)
        called = callByAbundance(report, cn)
        self.results[gene] = report
        return called, len(self.gene_reads[gene])

    def typing(self, gene_cn: dict[str, int], min_reads_num: int = 100) -> tuple[list[str], list[str]]:
        calls, warn = [], []
        for gene, cn in gene_cn.items():
            if not cn:
                continue
            alleles, n = self.typingPerGene(gene, cn)
            calls.extend(alleles)
            if n < min_reads_num:
                warn.append(gene)
        return calls, warn

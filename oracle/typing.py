"""
Oracle: positive/negative variant lists -> allele calls (likelihood search).

TEST INFRASTRUCTURE (see oracle/__init__.py).  Restates, from
``/root/reference/graphkir``:

* ``errorCorrect`` / ``dropEmpty``  <- typing_mulit_allele.py:302-338, 274-281
* ``readProbs``                     <- reads2AlleleProb 340-381, read2Onehot 287-292,
                                       onehot2Prob 294-300
* ``firstStep`` / ``nextStep``      <- addCandidate 478-598, uniqueAllele 456-476
* ``rankRows``                      <- rankScore 202-214, argSortRow 197-199
* ``isHomozygous`` / ``homoResult`` <- 807-857, 423-454
* ``GeneModel`` / ``ExonFirstModel``<- AlleleTyping 217-410, AlleleTypingExonFirst 622-797
* ``selectBest`` / ``topRank``      <- TypingResult 63-103, 173-194
* ``SampleTyper``                   <- kir_typing.py:15-150, 207-228

The array expressions of the search are kept in NumPy on purpose: NumPy's
reduction tree, ``log10`` and ``argsort`` ARE the reference arithmetic
(SURVEY.md section 8c "third-party arithmetic").  Reads are plain dicts with
keys lpv/lnv/rpv/rnv/backbone/multiple.
"""
from __future__ import annotations

import copy
from collections import defaultdict
from dataclasses import dataclass, field
from itertools import chain

import numpy as np

from kir_graph_amd.msa2hisat import Variant

HIT, MISS = 0.999, 0.001


# ---------------------------------------------------------------- read clean-up
def errorCorrect(reads: list[dict]) -> list[dict]:
    """Drop low-support variant ids from the lists, in place (302-338)."""
    n_pos: dict[str, int] = defaultdict(int)
    n_neg: dict[str, int] = defaultdict(int)
    seen: dict[str, None] = {}
    for r in reads:
        for v in r["lpv"] + r["rpv"]:
            n_pos[v] += 1
            seen[v] = None
        for v in r["lnv"] + r["rnv"]:
            n_neg[v] += 1
            seen[v] = None
    bad_pos, bad_neg = set(), set()
    for v in seen:
        p, n = n_pos[v], n_neg[v]
        if p + n < 3:
            bad_pos.add(v)
            bad_neg.add(v)
            continue
        if p / (p + n) < 0.2:
            bad_pos.add(v)
        if n / (p + n) < 0.2:
            bad_neg.add(v)
    for r in reads:
        r["lpv"] = [v for v in r["lpv"] if v not in bad_pos]
        r["rpv"] = [v for v in r["rpv"] if v not in bad_pos]
        r["lnv"] = [v for v in r["lnv"] if v not in bad_neg]
        r["rnv"] = [v for v in r["rnv"] if v not in bad_neg]
    return reads


def dropEmpty(reads: list[dict]) -> list[dict]:
    """Pairs without any id left carry no information (274-281)."""
    return [r for r in reads if r["lpv"] or r["lnv"] or r["rpv"] or r["rnv"]]


def alleleNames(variants: list[Variant]) -> set[str]:
    return set(chain.from_iterable(v.allele for v in variants))


def missTable(reads: list[dict], vmap: dict[str, Variant], col: dict[str, int]
              ) -> tuple[np.ndarray, np.ndarray]:
    """Integer hit table: miss[r, a] and nvar[r]."""
    A = len(col)
    miss = np.zeros((len(reads), A), dtype=np.int64)
    nvar = np.zeros(len(reads), dtype=np.int64)
    for i, r in enumerate(reads):
        for key, positive in (("lpv", True), ("rpv", True), ("lnv", False), ("rnv", False)):
            for vid in r[key]:
                has = np.zeros(A, dtype=bool)
                for a in vmap[vid].allele:
                    has[col[a]] = True
                miss[i] += (~has) if positive else has
                nvar[i] += 1
    return miss, nvar


def readProbs(reads: list[dict], vmap: dict[str, Variant], col: dict[str, int],
              no_empty: bool = True) -> np.ndarray:
    """P(read | allele) as the ordered product lpv, rpv, lnv, rnv of 0.999 / 0.001 (340-381)."""
    A = len(col)
    rows = []
    for r in reads:
        factors = []
        for key, positive in (("lpv", True), ("rpv", True), ("lnv", False), ("rnv", False)):
            for vid in r[key]:
                has = np.zeros(A, dtype=bool)
                for a in vmap[vid].allele:
                    has[col[a]] = True
                if not positive:
                    has = np.logical_not(has)
                f = np.ones(A) * MISS
                f[has] = HIT
                factors.append(f)
        if not factors and not no_empty:
            factors = [np.ones(A) * HIT]
        rows.append(np.stack(factors).prod(axis=0))
    if not rows:
        return np.array([])
    return np.stack(rows)


# ---------------------------------------------------------------- search result
@dataclass
class Result:
    """One copy-number step of the search (TypingResult 27-58)."""

    n: int
    value: np.ndarray
    value_sum_indv: np.ndarray
    allele_id: np.ndarray
    allele_name: list[list[str]]
    allele_prob: np.ndarray
    fraction: np.ndarray
    fraction_uniq: np.ndarray
    allele_name_group: list[list[list[str]]] = field(default_factory=list)

    def failed(self) -> bool:
        return not len(self.value)


def emptyResult(n: int) -> Result:
    e = np.array([])
    return Result(n, e, e, e, [], e, e, e)


def rankRows(value: np.ndarray, sum_indv: np.ndarray, fraction: np.ndarray) -> list[int]:
    """Stable order by (-value, -sum of per-allele sums, abundance unevenness) (202-214)."""
    uneven = np.abs(fraction - fraction.mean(axis=1, keepdims=True)).sum(axis=1)
    keys = np.array([-value, -sum_indv.sum(axis=1), uneven]).T
    return sorted(range(len(keys)), key=lambda i: tuple(keys[i]))


def reorder(res: Result, keep: int = -1) -> Result:
    """sortByScoreAndEveness 156-171."""
    if keep == -1:
        keep = res.value.shape[0]
    idx = rankRows(res.value, res.value_sum_indv, res.fraction)
    return Result(
        n=res.n,
        value=res.value[idx][:keep],
        value_sum_indv=res.value_sum_indv[idx][:keep],
        allele_id=res.allele_id[idx][:keep],
        allele_name=[res.allele_name[i] for i in idx][:keep],
        allele_prob=res.allele_prob[:, idx][:, :keep],
        fraction=res.fraction[idx][:keep],
        fraction_uniq=res.fraction_uniq[idx][:keep],
    )


def topRank(res: Result, threshold: float = 0.9) -> list[int]:
    """Rank 0 plus every rank whose value*threshold >= best (173-184)."""
    assert not res.failed()
    best = res.value[0]
    return [0] + [i for i, v in enumerate(res.value) if i and v * threshold >= best]


def selectBest(res: Result) -> list[str]:
    """First rank whose every abundance >= 0.5/n, else rank 0; fail -> ["fail"]*n (63-103)."""
    if res.failed():
        return ["fail"] * res.n
    floor = (1 / res.n) / 2
    ok = [i for i in range(len(res.fraction)) if all(f >= floor for f in res.fraction[i])]
    best = (ok or [0])[0]
    assert len(res.allele_name[best]) == res.n
    return res.allele_name[best]


def selectAllPossible(res: Result, threshold: float = 0.9) -> list[tuple[float, list[str]]]:
    if res.failed():
        return []
    return [(res.value[i], res.allele_name[i]) for i in topRank(res, threshold)]


def firstMask(ids: np.ndarray) -> np.ndarray:
    """True for the first occurrence of each allele multiset (uniqueAllele 456-476)."""
    seen = set()
    keep = []
    for row in ids:
        k = tuple(sorted(row))
        keep.append(k not in seen)
        seen.add(k)
    return np.array(keep)


def firstStep(L: np.ndarray, cols: np.ndarray, top_n: int, names: dict[int, str]) -> Result:
    """CN = 1: column sums, best top_n (512-532)."""
    score = L[:, cols].sum(axis=0)
    ids = cols[:, None]
    top = np.argsort(score)[::-1][:top_n]
    top_ids = ids[top]
    return Result(
        n=1,
        value=score[top],
        value_sum_indv=score[top][:, None],
        allele_id=top_ids,
        allele_name=[[names[i] for i in row] for row in top_ids],
        allele_prob=L[:, top_ids.flatten()],
        fraction=np.ones(top_ids.shape),
        fraction_uniq=np.ones(top_ids.shape),
    )


def nextStep(L: np.ndarray, prev: Result, cols: np.ndarray, top_n: int,
             names: dict[int, str], t_chunk: int = 0, k_chunk: int = 0) -> Result:
    """CN = k: extend every kept set by every candidate allele (534-598)."""
    P = prev.allele_prob       # R x T
    prev_ids = prev.allele_id  # T x (k-1)
    T, R = P.shape[1], L.shape[0]
    sub = L[:, cols]
    if t_chunk and T > t_chunk:
        parts = []
        for s in range(0, T, t_chunk):
            blk = np.maximum(sub, P.T[s:s + t_chunk, :, None]).sum(axis=1)
            parts.append(blk)
        score = np.concatenate(parts, axis=0).flatten()
    else:
        score = np.maximum(sub, P.T[:, :, None]).sum(axis=1).flatten()
    ids = np.hstack([np.repeat(prev_ids, len(cols), axis=0),
                     np.tile(cols, len(prev_ids))[:, None]])
    first = firstMask(ids)
    ids, score = ids[first], score[first]
    top = np.argsort(score)[::-1][:max(top_n, score.shape[0] // 5)]
    top_ids = ids[top]
    K = len(top_ids)
    if k_chunk and K > k_chunk:
        best_parts, sum_parts, frac_parts = [], [], []
        for s in range(0, K, k_chunk):
            g = L[:, top_ids[s:s + k_chunk]]
            b = g.max(axis=2)
            eq = np.equal(g, b[:, :, None])
            best_parts.append(b)
            sum_parts.append(g.sum(axis=0))
            frac_parts.append((eq / eq.sum(axis=2)[:, :, None]).sum(axis=0) / R)
        best = np.concatenate(best_parts, axis=1)
        sum_indv = np.concatenate(sum_parts, axis=0)
        frac = np.concatenate(frac_parts, axis=0)
    else:
        gathered = L[:, top_ids]               # R x K x k
        best = gathered.max(axis=2)            # R x K
        sum_indv = gathered.sum(axis=0)
        owns = np.equal(gathered, best[:, :, None])
        frac = (owns / owns.sum(axis=2)[:, :, None]).sum(axis=0) / R
    res = Result(
        n=prev.n + 1,
        value=score[top],
        value_sum_indv=sum_indv,
        allele_id=top_ids,
        allele_name=[[names[i] for i in row] for row in top_ids],
        allele_prob=best,
        fraction=frac,
        fraction_uniq=np.ones(frac.shape),
    )
    return reorder(res, keep=top_n)


def homoResult(res1: Result, cn: int) -> Result:
    """Replicate the CN=1 result cn times (createHomoResult 423-454)."""
    if cn <= 1:
        raise ValueError(f"CN should be > 1, got {cn}")
    m = len(res1.value)
    return Result(
        n=cn,
        value=res1.value * cn,
        value_sum_indv=np.repeat(res1.value_sum_indv, cn, axis=1),
        allele_id=np.repeat(res1.allele_id, cn, axis=1),
        allele_name=[[row[0]] * cn for row in res1.allele_name],
        allele_prob=res1.allele_prob,
        fraction=np.ones((m, cn)) / cn,
        fraction_uniq=np.ones((m, cn)) / cn,
    )


def isHomozygous(reads: list[dict], vmap: dict[str, Variant], cn: int) -> bool:
    """No position with convincing bi-allelic support => homozygous (807-857)."""
    if cn <= 1:
        return False
    site: dict[int, dict[str, int]] = defaultdict(lambda: defaultdict(int))
    for r in reads:
        for vid in chain(r["lpv"], r["rpv"]):
            v = vmap[vid]
            if v.typ != "deletion":
                site[v.pos][str(v.val)] += 1
        for vid in chain(r["lnv"], r["rnv"]):
            v = vmap[vid]
            if v.typ != "deletion":
                site[v.pos][f"*{v.val}"] += 1
    hits = 0
    for obs in site.values():
        if len(obs) <= 1 or all("*" in k for k in obs):
            continue
        counts = [c for c in sorted(obs.values(), reverse=True) if c > 3]
        total = sum(counts)
        if total < 20:
            continue
        major = [c / total for c in counts if c / total > 0.1]
        if len(major) == 1:
            continue
        if major[1] > 1 / (cn * 2):
            hits += 1
    return hits == 0


def forcedHetero(gene: str) -> bool:
    return "2DL1S1" in gene or "2DL5" in gene


# ---------------------------------------------------------------- models
class GeneModel:
    """Full-variant likelihood model of one gene (AlleleTyping 217-410)."""

    def __init__(self, reads: list[dict], variants: list[Variant], force_homo=None,
                 top_n: int = 300, no_empty: bool = True, variant_correction: bool = True):
        self.top_n = top_n
        self.force_homo = force_homo
        self.variants = {str(v.id): v for v in variants}
        self.id_to_allele = dict(enumerate(sorted(alleleNames(variants))))
        self.allele_to_id = {a: i for i, a in self.id_to_allele.items()}
        if variant_correction:
            reads = errorCorrect(reads)
        if no_empty:
            reads = dropEmpty(reads)
        self.reads = reads
        self.probs = readProbs(reads, self.variants, self.allele_to_id, no_empty)
        with np.errstate(divide="ignore"):
            self.log_probs = np.log10(self.probs)
        self.result: list[Result] = []

    def readsNum(self) -> int:
        return len(self.probs)

    def addCandidate(self, candidates: list[str] | None = None) -> Result:
        if not self.probs.shape[0]:
            self.result.append(emptyResult(len(self.result) + 1))
            return self.result[-1]
        if candidates is None:
            cols = np.arange(self.log_probs.shape[1])
        else:
            cols = np.array([self.allele_to_id.get(a) for a in candidates])
        if not self.result:
            res = firstStep(self.log_probs, cols, self.top_n, self.id_to_allele)
        else:
            res = nextStep(self.log_probs, self.result[-1], cols, self.top_n, self.id_to_allele,
                           t_chunk=self._chunk("t", len(cols)), k_chunk=self._chunk("k", self.result[-1].n + 1))
        self.result.append(res)
        return res

    def _chunk(self, kind: str, width: int) -> int:
        """Bound temporaries to ~1 GiB; 0 = evaluate in one piece like the reference."""
        R = max(1, self.log_probs.shape[0])
        per = R * max(1, width) * 8
        n = int((1 << 30) // per)
        return max(1, n) if n < 100000 else 0

    def typing(self, cn: int) -> Result:
        if cn < 1:
            raise ValueError(f"CN should be >= 1, got {cn}")
        homo = isHomozygous(self.reads, self.variants, cn) if self.force_homo is None else self.force_homo
        self.result = []
        if homo:
            self.addCandidate()
            if cn > 1:
                self.result.append(homoResult(self.result[0], cn))
        else:
            for _ in range(cn):
                self.addCandidate()
        return self.result[-1]


def exonGroups(variants: list[Variant]) -> dict[str, list[str]]:
    """Alleles with identical exon-variant sets share one group "a|b|c" (649-656, 689-700)."""
    exon_vars = [v for v in variants if v.in_exon]
    per_allele: dict[str, list[str]] = defaultdict(list)
    for v in exon_vars:
        for a in v.allele:
            per_allele[a].append(str(v.id))
    by_set: dict[tuple, list[str]] = defaultdict(list)
    for a, ids in per_allele.items():
        by_set[tuple(sorted(set(ids)))].append(a)
    rest = alleleNames(variants) - alleleNames(exon_vars)
    if rest:
        by_set[tuple()] = sorted(rest)
    return {"|".join(m): m for m in by_set.values()}


class ExonFirstModel(GeneModel):
    """Exon variants pick candidate groups, the full model refines them (622-797)."""

    def __init__(self, reads: list[dict], variants: list[Variant], top_n: int = 300,
                 exon_only: bool = False, candidate_set_threshold: float = 1.0,
                 variant_correction: bool = True, force_homo=None):
        exon_ids = {v.id for v in variants if v.in_exon}
        exon_reads = copy.deepcopy(reads)
        for r in exon_reads:
            for k in ("lpv", "lnv", "rpv", "rnv"):
                r[k] = [v for v in r[k] if v in exon_ids]
        if variant_correction:
            exon_reads = errorCorrect(exon_reads)
        exon_reads = dropEmpty(exon_reads)
        self.allele_group = exonGroups(variants)
        to_group = {a: g for g, members in self.allele_group.items() for a in members}
        grouped = copy.deepcopy(variants)
        for v in grouped:
            v.allele = list(set(filter(None, [to_group.get(a, "") for a in v.allele])))
        super().__init__(exon_reads, grouped, force_homo=force_homo, top_n=top_n)
        self.candidate_set_threshold = candidate_set_threshold
        self.full_model = None if exon_only else GeneModel(
            reads, variants, force_homo=force_homo, top_n=top_n // 5,
            variant_correction=variant_correction)

    def typing(self, cn: int) -> Result:
        res = super().typing(cn)
        res.allele_name_group = [[self.allele_group[a] for a in row] for row in res.allele_name]
        if self.full_model is None:
            return res
        assert cn == res.n
        if not res.value.shape[0]:
            return self.full_model.typing(cn)
        finals = []
        for i in topRank(res, self.candidate_set_threshold):
            model = copy.deepcopy(self.full_model)
            for cand in res.allele_name_group[i]:
                model.addCandidate(cand)
            self.result.extend(model.result)
            finals.append(model.result[-1])
        merged = Result(
            n=finals[0].n,
            value=np.concatenate([f.value for f in finals]),
            value_sum_indv=np.concatenate([f.value_sum_indv for f in finals]),
            allele_id=np.concatenate([f.allele_id for f in finals]),
            allele_name=list(chain.from_iterable(f.allele_name for f in finals)),
            allele_prob=np.concatenate([f.allele_prob for f in finals], axis=1),
            fraction=np.concatenate([f.fraction for f in finals]),
            fraction_uniq=np.concatenate([f.fraction for f in finals]),
        )
        merged = reorder(merged)
        self.result.append(merged)
        return merged


# ---------------------------------------------------------------- per-sample driver
class SampleTyper:
    """TypingWithPosNegAllele (kir_typing.py:77-150) on an in-memory tabulation."""

    def __init__(self, data: dict, top_n: int = 300, multiple: bool = False,
                 exon_first: bool = False, exon_only: bool = False,
                 exon_candidate_threshold: float = 0.9, variant_correction: bool = False):
        reads = data["reads"]
        if not multiple:
            reads = [r for r in reads if r["multiple"] == 1]
        self.gene_reads: dict[str, list[dict]] = defaultdict(list)
        for r in reads:
            self.gene_reads[r["backbone"]].append(r)
        self.gene_variants: dict[str, list[Variant]] = defaultdict(list)
        for v in data["variants"]:
            self.gene_variants[v.ref].append(v)
        self.top_n, self.exon_first, self.exon_only = top_n, exon_first, exon_only
        self.threshold, self.variant_correction = exon_candidate_threshold, variant_correction
        self.results: dict[str, list[Result]] = {}

    def typingPerGene(self, gene: str, cn: int) -> tuple[list[str], int]:
        force = False if forcedHetero(gene) else None
        if not self.exon_first and not self.exon_only:
            model: GeneModel = GeneModel(self.gene_reads[gene], self.gene_variants[gene], force_homo=force,
                                         top_n=self.top_n, variant_correction=self.variant_correction)
        else:
            model = ExonFirstModel(self.gene_reads[gene], self.gene_variants[gene], force_homo=force,
                                   top_n=self.top_n, exon_only=self.exon_only,
                                   candidate_set_threshold=self.threshold)
        res = model.typing(cn)
        self.results[gene] = model.result
        short = gene.split("*")[0]
        return [a if a != "fail" else f"{short}*" for a in selectBest(res)], model.readsNum()

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

    def allPossible(self) -> list[dict]:
        rows = []
        for gene, steps in self.results.items():
            for rank, (value, alleles) in enumerate(selectAllPossible(steps[-1], 0.9)):
                row = {"gene": gene, "rank": rank, "value": value}
                for i, a in enumerate(alleles):
                    row[str(i + 1)] = a
                rows.append(row)
        return rows


def makeTyper(method: str, data: dict, **kwargs) -> SampleTyper:
    """selectKirTypingModel 207-228 for the likelihood strategies."""
    if method == "full":
        return SampleTyper(data, **kwargs)
    if method.startswith("exonfirst"):
        parts = method.split("_")
        th = float(method[len("exonfirst_"):]) if len(parts) == 2 else 0.0
        return SampleTyper(data, exon_first=True, exon_candidate_threshold=th, **kwargs)
    raise NotImplementedError

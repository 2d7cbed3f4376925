"""
Multi-allele likelihood typing on the GPU -- drop-in for
``graphkir/typing_mulit_allele.py`` (the file name keeps the reference's spelling).

Same classes and attributes as the reference (``TypingResult``, ``AlleleTyping``,
``AlleleTypingExonFirst``, ``isHetrozygous``, ``isHomozygous``; attributes
``probs / log_probs / reads / variants / allele_to_id / id_to_allele / result /
top_n`` that ``novel_discover.py:58-64,273-274`` reads), but:

* reads x alleles tables live in HBM (``engine.DeviceModel``); ``probs`` /
  ``log_probs`` / ``allele_prob`` are fetched only when somebody asks;
* the T x R x A temporary of ``addCandidate`` (540-542) never exists: the
  device forms ``max`` and the read reduction in one pass with numpy's
  summation tree, the host only ranks T x A scores with the same numpy calls
  as the reference (``argsort``), so ties break identically;
* the per-set statistics (569-580) are computed without the R x K x CN gathers.

Host work per copy-number step is O(T x A) numpy, never O(reads).
"""
from __future__ import annotations

from collections import defaultdict
from dataclasses import dataclass, field
from itertools import chain
from typing import Iterable, Optional

import threading

import numpy as np

from ._lib import Device, DeviceBuffer
from .engine import DeviceModel, LogTable, Tabulation
from .index import buildMask
from .msa2hisat import Variant
from .utils import logger

# search steps served by the integer bound / steps it handed back to the exact kernels (diagnostics)
SEARCH_STATS = {"bounded": 0, "redone_exactly": 0}
_default_logs: dict[int, LogTable] = {}
_default_logs_lock = threading.Lock()


def nativeSearch() -> bool:
    """The steps of a gene's search run inside the library (``gk_search_run``); ``addCandidate`` remains as the
    reference's API for a step with given candidates."""
    return True


_GROUP_CACHE_LOCK = threading.Lock()      # guards the per-gene exon-group caches (AlleleTypingExonFirst)


def sharedLogTable(dev: Device) -> LogTable:
    """One log10 value table per GPU, shared by all its contexts (values recur across genes, samples
    and host threads, so every value is evaluated by numpy once per process)."""
    with _default_logs_lock:
        t = _default_logs.get(dev.ordinal)
        if t is None:
            t = _default_logs[dev.ordinal] = LogTable(dev, private_context=True)
    return t


@dataclass
class TypingResult:
    """Result of one copy-number step (typing_mulit_allele.py:27-58)."""

    n: int
    value: np.ndarray
    value_sum_indv: np.ndarray
    allele_id: np.ndarray
    allele_name: list[list[str]]
    allele_prob: "np.ndarray | LazyAlleleProb"
    fraction: np.ndarray
    fraction_uniq: np.ndarray
    allele_name_group: list[list[list[str]]] = field(default_factory=list)

    def isFail(self) -> bool:
        return not len(self.value)

    def selectBest(self, filter_fraction: bool = True, filter_minor: bool = False) -> list[str]:
        ids: Iterable[int] = range(len(self.fraction))
        if filter_fraction and len(self.fraction):
            floor = (1 / self.n) / 2
            ids = np.flatnonzero((np.asarray(self.fraction) >= floor).all(axis=1)).tolist()
        if filter_minor:
            ids = [i for i in ids
                   if np.abs(self.value_sum_indv[i]).min() / np.abs(self.value_sum_indv[i]).max() > 0.8]
        best = (list(ids) or [0])[0]
        if not self.isFail():
            logger.debug(f"[Allele] Select best rank: {best}")
            assert len(self.allele_name[best]) == self.n
            return self.allele_name[best]
        logger.warning("[Allele] No candidates found. Return fail")
        return ["fail"] * self.n

    def print(self, num: int = 100, top_threshold: float = 0.9) -> None:
        if self.isFail():
            print("Fail Alleles:", ["fail"] * self.n)   # the reference prints this to stdout (line 125)
            return
        if not logger.isEnabledFor(10):
            return
        lines = [f"Allele_num =  {self.n}"]
        for k, rank in enumerate(self.topRank(top_threshold)):
            if k > num:
                break
            lines.append(f"Rank {rank} probility {self.value[rank]} sum {self.value_sum_indv[rank].sum()}")
            for i in range(self.n):
                lines.append(f"  id {self.allele_id[rank][i]:3}   name {self.allele_name[rank][i]:20s}"
                             f"   fraction {self.fraction[rank][i]:.5f}   sum {self.value_sum_indv[rank][i]:8.3f}")
        logger.debug("[Allele] " + "\n".join(lines))

    def setNameGroup(self, allele_group_mapping: dict[str, list[str]]) -> None:
        """``allele_name_group[i][j]`` = the member alleles of the j-th group of row i (166-169).  Rows are built when they
        are read: exon-first reads the few rows that reach its threshold, the table has up to top_n."""
        self.allele_name_group = _GroupRows(self.allele_name, allele_group_mapping)

    def sortByScoreAndEveness(self, preserve_topn: int = -1) -> "TypingResult":
        if preserve_topn == -1:
            preserve_topn = self.value.shape[0]
        order = rankScore(self.value, self.value_sum_indv, self.fraction)[:preserve_topn]
        return TypingResult(
            n=self.n, value=self.value[order], value_sum_indv=self.value_sum_indv[order],
            allele_id=self.allele_id[order], allele_name=[self.allele_name[i] for i in order],
            allele_prob=takeColumns(self.allele_prob, order),
            fraction=self.fraction[order], fraction_uniq=self.fraction_uniq[order])

    def topRank(self, threshold: float = 0.9) -> Iterable[int]:
        assert not self.isFail()
        best = self.value[0]
        return [0] + [int(i) for i in np.nonzero(self.value[1:] * threshold >= best)[0] + 1]

    def selectAllPossible(self, threshold: float = 0.9) -> list[tuple[float, list[str]]]:
        if self.isFail():
            return []
        return [(self.value[r], self.allele_name[r]) for r in self.topRank(threshold)]


class _GroupRows:
    """``[[mapping[a] for a in row] for row in names]`` as a sequence whose rows are built on access."""

    def __init__(self, names, mapping: dict[str, list[str]]):
        self._names, self._mapping = names, mapping

    def __len__(self) -> int:
        return len(self._names)

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[k] for k in range(*i.indices(len(self)))]
        return [self._mapping[a] for a in self._names[i]]

    def __iter__(self):
        return (self[i] for i in range(len(self)))

    def __eq__(self, other) -> bool:
        return list(self) == list(other)


class SearchSteps:
    """The steps of many native searches on one table (``AlleleTyping.adoptSearches``) as a sequence of ``TypingResult`` in
    (search, step) order; a step is built from the packed columns when it is read."""

    def __init__(self, typing, steps_of: np.ndarray, meta: np.ndarray, value, sum_indv, frac, ids):
        self._typing = typing
        self.steps_of = steps_of                                   # steps per search
        self.set_size, self.rows, self.bounded = meta[:, 0], meta[:, 1], meta[:, 2]
        self._value, self._sum_indv, self._frac, self._ids = value, sum_indv, frac, ids
        self.row0 = np.concatenate([[0], np.cumsum(self.rows)])
        self.cell0 = np.concatenate([[0], np.cumsum(self.rows * self.set_size)])
        self.first_step = np.concatenate([[0], np.cumsum(steps_of)])          # of search q: its steps are first_step[q] ..
        self.step_in_search = np.arange(len(self.rows)) - np.repeat(self.first_step[:-1], steps_of)
        self._made: dict[int, TypingResult] = {}

    def __len__(self) -> int:
        return len(self.rows)

    def __getitem__(self, j):
        if isinstance(j, slice):
            return [self[k] for k in range(*j.indices(len(self)))]
        if j < 0:
            j += len(self)
        got = self._made.get(j)
        if got is None:
            t = self._typing
            k, c = int(self.rows[j]), int(self.set_size[j])
            r0, c0 = int(self.row0[j]), int(self.cell0[j])
            ids = self._ids[c0:c0 + k * c].reshape(k, c)
            frac = self._frac[c0:c0 + k * c].reshape(k, c)
            got = self._made[j] = TypingResult(
                n=c, value=self._value[r0:r0 + k], value_sum_indv=self._sum_indv[c0:c0 + k * c].reshape(k, c), allele_id=ids,
                allele_name=LazyNames(ids, t.id_to_allele), allele_prob=LazyAlleleProb([(t._model, ids)]),
                fraction=frac if self.step_in_search[j] else np.ones(ids.shape), fraction_uniq=np.ones(ids.shape))
        return got

    def __iter__(self):
        return (self[j] for j in range(len(self)))

    def lastSteps(self) -> "TypingResult":
        """The last steps of all searches as ONE result, rows in search order (what ``mergeCandidates`` concatenates)."""
        last = self.first_step[1:] - 1
        c = int(self.set_size[last[0]])
        assert np.all(self.set_size[last] == c)
        k = self.rows[last]
        within = np.arange(int(k.sum())) - np.repeat(np.cumsum(k) - k, k)
        rows = np.repeat(self.row0[last], k) + within
        cells = (np.repeat(self.cell0[last], k) + within * c)[:, None] + np.arange(c)[None, :]
        ids, t = self._ids[cells], self._typing
        return TypingResult(n=c, value=self._value[rows], value_sum_indv=self._sum_indv[cells], allele_id=ids,
                            allele_name=LazyNames(ids, t.id_to_allele), allele_prob=LazyAlleleProb([(t._model, ids)]),
                            fraction=self._frac[cells], fraction_uniq=self._frac[cells])


class StepList:
    """``result`` of a gene typed exon-first: the exon model's steps, the steps of every candidate search (built when
    read), the merged final result -- a sequence like the reference's list (typing_mulit_allele.py:776-796)."""

    def __init__(self, head: list, steps: SearchSteps, tail: list):
        self._head, self._steps, self._tail = list(head), steps, list(tail)

    def __len__(self) -> int:
        return len(self._head) + len(self._steps) + len(self._tail)

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[k] for k in range(*i.indices(len(self)))]
        n = len(self)
        if i < 0:
            i += n
        if not 0 <= i < n:
            raise IndexError(i)
        if i < len(self._head):
            return self._head[i]
        i -= len(self._head)
        if i < len(self._steps):
            return self._steps[i]
        return self._tail[i - len(self._steps)]

    def __iter__(self):
        return (self[i] for i in range(len(self)))

    def __bool__(self) -> bool:
        return len(self) > 0


class LazyAlleleProb:
    """``allele_prob`` (reads x sets) evaluated on the device only when it is read."""

    def __init__(self, parts: list[tuple[DeviceModel, np.ndarray]]):
        self.parts = parts  # [(model, ids [k x c])] concatenated along sets

    @property
    def shape(self) -> tuple[int, int]:
        return (self.parts[0][0].n_rows, sum(len(ids) for _, ids in self.parts))

    def __array__(self, dtype=None, copy=None):
        cols = [m.setmax(ids) for m, ids in self.parts if len(ids)]
        out = np.concatenate(cols, axis=1) if cols else np.zeros((0, 0))
        return out.astype(dtype) if dtype is not None else out


def takeColumns(prob, order: np.ndarray):
    if not isinstance(prob, LazyAlleleProb):
        return prob[:, order]
    bounds = np.cumsum([0] + [len(ids) for _, ids in prob.parts])
    parts = []
    # keep the requested order: one part per run of consecutive picks from the same source
    for o in order:
        k = int(np.searchsorted(bounds, o, side="right") - 1)
        m, ids = prob.parts[k]
        row = ids[o - bounds[k]][None, :]
        if parts and parts[-1][0] is m:
            parts[-1] = (m, np.concatenate([parts[-1][1], row]))
        else:
            parts.append((m, row))
    return LazyAlleleProb(parts or [(prob.parts[0][0], prob.parts[0][1][:0])])


def rankScore(value: np.ndarray, value_sum_indv: np.ndarray, fraction: np.ndarray) -> np.ndarray:
    """Stable order by (-value, -sum of per-allele sums, abundance unevenness) (202-214).

    ``np.lexsort`` is a stable multi-key sort, i.e. the same order as Python's ``sorted`` on the
    key tuples that the reference uses.
    """
    uneven = np.abs(fraction - fraction.mean(axis=1, keepdims=True)).sum(axis=1)
    return np.lexsort((uneven, -value_sum_indv.sum(axis=1), -value))


def firstOccurrence(ids: np.ndarray, n_allele: int) -> np.ndarray:
    """Mask of the first occurrence of every allele multiset (uniqueAllele 456-476), vectorised."""
    ids = np.asarray(ids, dtype=np.int64)
    bits = max(1, int(n_allele - 1).bit_length())
    if ids.shape[1] == 2:
        # two alleles per set: order them with min / max and test duplicates by hashing
        import pandas as pd
        key = (np.minimum(ids[:, 0], ids[:, 1]) << bits) | np.maximum(ids[:, 0], ids[:, 1])
        return ~pd.Series(key).duplicated(keep="first").to_numpy()
    srt = np.sort(ids, axis=1)
    if bits * ids.shape[1] > 62:
        seen, keep = set(), np.zeros(len(ids), dtype=bool)
        for i, row in enumerate(map(tuple, srt)):
            keep[i] = row not in seen
            seen.add(row)
        return keep
    key = np.zeros(len(ids), dtype=np.int64)
    for j in range(ids.shape[1]):
        key = (key << bits) | srt[:, j]
    import pandas as pd
    return ~pd.Series(key).duplicated(keep="first").to_numpy()


def firstOfSets(prev_ids: np.ndarray, cols: np.ndarray, n_allele: int) -> np.ndarray:
    """First-occurrence mask over the candidates ``prev_ids[t] + [cols[a]]`` in (t-major, a-minor) order.

    Same result as ``firstOccurrence`` on the stacked id table (uniqueAllele 456-476), without hashing
    T x A keys: candidate (t, a) repeats an earlier one exactly when swapping the new allele with a
    member e of the previous set gives a previous set that sits earlier in the list -- the multiset
    ``prev[t] - {e} + {a}`` at a position < t, with e itself among the offered alleles -- or when
    ``prev[t]`` already occurred earlier.  One- and two-allele previous sets are looked up in
    position tables; larger ones fall back to the hashed table."""
    prev_ids = np.asarray(prev_ids, dtype=np.int64)
    cols = np.asarray(cols, dtype=np.int64)
    T, k = prev_ids.shape
    A = len(cols)
    if k > 2 or n_allele > 2048 or len(np.unique(cols)) != A:
        ids = np.hstack([np.repeat(prev_ids, A, axis=0), np.tile(cols, T)[:, None]])
        return firstOccurrence(ids, n_allele)
    never = np.int64(T)
    offered = np.zeros(n_allele, dtype=bool)
    offered[cols] = True
    rank = np.arange(T, dtype=np.int64)
    if k == 1:
        p = prev_ids[:, 0]
        pos = np.full(n_allele, never)
        np.minimum.at(pos, p[::-1], rank[::-1])              # first position of every single-allele set
        first = (pos[cols][None, :] >= rank[:, None]) | ~offered[p][:, None]
        first &= (pos[p] == rank)[:, None]                   # prev[t] itself must be a first occurrence
        return first.ravel()
    lo, hi = prev_ids.min(axis=1), prev_ids.max(axis=1)
    pos = np.full(n_allele * n_allele, never)
    np.minimum.at(pos, (lo * n_allele + hi)[::-1], rank[::-1])
    pos = pos.reshape(n_allele, n_allele)
    pos = np.minimum(pos, pos.T)            # symmetric: pos[x, y] = first position of the pair {x, y}
    by_col = pos[:, cols]                   # [allele][offered column]

    def earlier(keep: np.ndarray, gone: np.ndarray) -> np.ndarray:
        """candidate repeats (keep[t], a) + gone[t] found at an earlier position"""
        return (by_col[keep] < rank[:, None]) & offered[gone][:, None]

    first = ~(earlier(hi, lo) | earlier(lo, hi))
    first &= (pos[lo, hi] == rank)[:, None]
    return first.ravel()


class LazyNames:
    """``allele_name`` (list of name lists) built from the id table only for the rows that are read."""

    def __init__(self, ids: np.ndarray, id_to_allele: dict[int, str]):
        self._ids, self._map = ids, id_to_allele

    def __len__(self) -> int:
        return len(self._ids)

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(len(self)))]
        return [self._map[int(x)] for x in self._ids[i]]

    def __iter__(self):
        return (self[i] for i in range(len(self)))

    def __eq__(self, other) -> bool:
        return list(self) == list(other)

    def __repr__(self) -> str:
        return f"LazyNames({len(self)} sets)"


# ---------------------------------------------------------------- device read set
class ReadSet:
    """Ordered reads of one gene on the device: row list + variant drop flags."""

    def __init__(self, tab: Tabulation, rows: DeviceBuffer, n_rows: int, vflag: DeviceBuffer | None = None):
        self.tab, self.rows, self.n_rows = tab, rows, n_rows
        self.vflag = vflag if vflag is not None else tab.dev.alloc(max(tab.n_var_total, 1), np.uint8).zero()

    @classmethod
    def fromLists(cls, dev: Device, reads, variants: list[Variant]) -> tuple["ReadSet", dict[str, int]]:
        """Upload host ``PairRead`` lists (drop-in constructor path)."""
        import ctypes as C
        from . import _lib
        from ._lib import check, lib
        ordinal = {str(v.id): i for i, v in enumerate(variants)}
        n = len(reads)
        off = np.zeros(4 * n + 1, dtype=np.uint32)
        flat: list[int] = []
        for i, r in enumerate(reads):
            for k, lst in enumerate((r.lpv, r.rpv, r.lnv, r.rnv)):
                flat.extend(ordinal[v] for v in lst)
                off[4 * i + k + 1] = len(flat)
        ids = np.array(flat, dtype=np.uint32)
        zeros = np.zeros(max(n, 1), dtype=np.uint8)
        ones = np.ones(max(n, 1), dtype=np.uint8)
        tab = Tabulation.__new__(Tabulation)
        tab.dev, tab.dindex, tab.mates, tab.n_pairs = dev, None, None, n
        h = C.c_void_p()
        check(lib().gk_tab_from_csr(dev.ctx, max(len(variants), 1), n, off.ctypes.data,
                                    ids.ctypes.data if len(ids) else None, zeros.ctypes.data, ones.ctypes.data,
                                    C.byref(h)))
        tab.handle = h
        info = _lib.TabInfo()
        check(lib().gk_tab_get_info(h, C.byref(info)))
        tab.info, tab.n_valid, tab.n_ids = info, n, int(info.n_ids)
        tab.n_novel, tab.novel_base, tab._novel_keys = 0, 0, None
        tab._id_names = [str(v.id) for v in variants] or ["-"]
        tab._variant_src = list(variants)
        rows = dev.put(np.arange(max(n, 1), dtype=np.int32))
        return cls(tab, rows, n), ordinal

    def copyFlags(self) -> DeviceBuffer:
        from ._lib import check, lib
        out = self.tab.dev.alloc(self.vflag.shape, np.uint8)
        check(lib().gk_d2d(self.tab.dev.ctx, out.ptr, self.vflag.ptr, self.vflag.nbytes))
        return out

    def hostReads(self, template=None) -> list:
        """Surviving reads with surviving ids as ``PairRead`` objects (API parity)."""
        from .hisat2 import PairRead
        tab = self.tab
        rows = self.rows.download(self.n_rows) if self.n_rows else np.zeros(0, np.int32)
        off, ids = tab.offsets().astype(np.int64), tab.ids()
        flag = self.vflag.download()
        names = np.array(tab.idNames(), dtype=object)
        out = []
        for r in rows:
            o = off[4 * r:4 * r + 5]
            lists = []
            for k, bit in ((0, 1), (1, 1), (2, 2), (3, 2)):
                seg = ids[o[k]:o[k + 1]]
                lists.append(list(names[seg[(flag[seg] & bit) == 0]]))
            base = template[int(r)] if template is not None else None
            out.append(PairRead(l_sam=getattr(base, "l_sam", ""), r_sam=getattr(base, "r_sam", ""),
                                multiple=getattr(base, "multiple", 1), backbone=getattr(base, "backbone", ""),
                                lpv=lists[0], rpv=lists[1], lnv=lists[2], rnv=lists[3]))
        return out


class AlleleTyping:
    """Likelihood model and greedy multi-allele search of one gene (217-619).

    Ordinal convention: the read lists hold variant ordinals of the owning ``Tabulation``;
    ``variants[:n_span]`` are the variants with ordinals ``[vbeg, vbeg + n_span)`` (for a list-built
    model that is simply every variant, ``vbeg = 0``); ordinals outside carry no allele (novel).
    """

    def __init__(self, reads, variants: list[Variant], force_homo: bool | None = None, top_n: int = 300,
                 no_empty: bool = True, variant_correction: bool = True, *, device: Device | None = None,
                 logs: LogTable | None = None, _vbeg: int = 0, _n_span: int | None = None,
                 _mask: DeviceBuffer | None = None, _alleles: list[str] | None = None, _defer_log: bool = False,
                 _novel=None, _prepared: tuple | None = None, _defer_launch: bool = False):
        """``_prepared`` = (rows with a surviving id, their count, shared drop flags, shared tallies, (gene, vbeg, vend)):
        error correction and removal of empty reads already done for the whole sample (``Tabulation.prepared``).
        ``_defer_launch``: the tables are allocated, not written -- ``gk_sample_search`` writes them and runs the
        search together with the sample's other genes (``kir_typing.TypingWithPosNegAllele``)."""
        self.top_n = top_n
        self._no_empty = no_empty
        self.force_homo = force_homo
        self._variant_list = variants
        self._novel_provider = _novel    # callable -> novel variants of the gene (built on demand)
        self._variants_map: dict[str, Variant] | None = None
        names = _alleles if _alleles is not None else sorted(self.collectAlleleNames(variants))
        self.id_to_allele: dict[int, str] = dict(enumerate(names))
        self.allele_to_id: dict[str, int] = {a: i for i, a in self.id_to_allele.items()}
        self.result: list[TypingResult] = []

        if isinstance(reads, ReadSet):
            rs = reads
            self._template = None
        else:
            device = device or Device()
            self._template = list(reads)
            rs, _ = ReadSet.fromLists(device, self._template, variants)
        self._dev = rs.tab.dev
        self._logs = logs or sharedLogTable(self._dev)
        tab = rs.tab
        n_span = len(variants) if _n_span is None else _n_span
        self._span = (_vbeg, _vbeg + n_span)
        self._tally = None
        self._tally_gene = None       # (gene, vbeg, vend) when the tallies cover every gene of the sample
        self._surviving = None
        if _prepared is not None:
            assert variant_correction and no_empty
            rows, n_rows, _, self._tally, self._tally_gene = _prepared[:5]   # rs.vflag IS the shared, corrected one
            self._surviving = _prepared[5] if len(_prepared) > 5 else None      # this gene's surviving tallies, if fetched already
        else:
            if variant_correction:
                self._tally = tab.errorCorrection(rs.rows, rs.n_rows, rs.vflag, span=self._span, keep=True)
            if no_empty:
                rows, n_rows = tab.selectNonEmpty(rs.rows, rs.n_rows, rs.vflag)
            else:   # reads without information stay and score 0.999 for every allele (372-374)
                rows, n_rows = rs.rows, rs.n_rows
        self._readset = ReadSet(tab, rows, n_rows, rs.vflag)
        n_allele = len(names)
        words = max(1, (n_allele + 31) // 32)
        if _mask is None:
            _mask = self._dev.put(buildMask(variants[:n_span], names))
        import os
        self._model = DeviceModel(tab, rows, n_rows, rs.vflag, _vbeg, _vbeg + n_span, _mask, words, n_allele,
                                  self._logs, keep_empty=not no_empty, launch=not _defer_launch,
                                  indexed=_defer_launch and os.environ.get("GK_INDEX_TABLE", "0") == "1")
        self._colsum_all: np.ndarray | None = None
        self._pair_table: np.ndarray | None = None    # scores of all allele pairs (second step), when formed
        self._reads_cache = None
        if not _defer_log and not _defer_launch:
            self.finish()
        if n_rows == 0:
            logger.warning("[Allele] Error: Empty reads for typing (or Maybe read depth is too low)")

    def finish(self) -> None:
        """Make sure the log-likelihood table is final (every product had its log10 in the value table);
        cheap when it already is.  Called before the table is read -- not right after its launch, so
        that the host can do other work (the zygosity test) while the compatibility kernel runs."""
        self._model.finishLog()

    # ---- reference attribute surface (lazy)
    @property
    def variants(self) -> dict[str, Variant]:
        """variant id -> Variant (index variants of the gene + the sample's novel ones)."""
        if self._variants_map is None:
            extra = self._novel_provider() if self._novel_provider is not None else []
            self._variants_map = {str(v.id): v for v in chain(self._variant_list, extra)}
        return self._variants_map

    @property
    def probs(self) -> np.ndarray:
        return self._model.hostProbs()

    @property
    def log_probs(self) -> np.ndarray:
        self.finish()
        return self._model.hostLogProbs()

    @property
    def reads(self) -> list:
        if self._reads_cache is None:
            self._reads_cache = self._readset.hostReads(self._template)
        return self._reads_cache

    def getReadsNum(self) -> int:
        return self._model.n_rows

    @staticmethod
    def collectAlleleNames(variants: list[Variant]) -> set[str]:
        return set(chain.from_iterable(v.allele for v in variants))

    def mapAlleleIDs(self, list_ids: np.ndarray) -> list[list[str]]:
        return [[self.id_to_allele[int(i)] for i in ids] for ids in list_ids]

    def fork(self) -> "AlleleTyping":
        """Fresh search state on the same device tables (stands in for ``copy.deepcopy``, line 742)."""
        other = object.__new__(type(self))
        other.__dict__.update(self.__dict__)
        other.result = []
        return other

    # ---- search
    def typing(self, cn: int) -> TypingResult:
        if cn < 1:
            raise ValueError(f"CN should be >= 1, got {cn}")
        homo = self._isHomozygous(cn) if self.force_homo is None else self.force_homo
        self.finish()
        self.result = []
        if homo:
            self.addCandidate()
            self.addHomoResultForCn(cn)
        else:
            if self._model.n_rows:
                self._searchNatively(cn)
            else:
                for _ in range(cn):
                    self.addCandidate()
        self.result[-1].print()
        return self.result[-1]

    def _searchNatively(self, cn: int) -> None:
        """All ``cn`` steps of ``addCandidate()`` in one library call (``gk_search_run``): the same device
        kernels, the host half of every step in C++ on this thread instead of numpy under the interpreter
        lock; ``numpy.argsort`` is handed in as a callback wherever its tie order is part of the result."""
        import ctypes as C
        from . import _lib
        from ._lib import check, lib
        m = self._model
        cols = np.arange(m.n_allele, dtype=np.int32)
        bound = m.boundOk
        colsum = None if self._colsum_all is None else np.ascontiguousarray(self._colsum_all, dtype=np.float64)
        h = C.c_void_p()
        check(lib().gk_search_run(m.dev.ctx, m.L.ptr, m.n_rows, m.n_rows, m.n_allele,
                                  m.miss8.ptr if bound else 0, m.ldm, m.msum.ptr if bound else 0,
                                  cols.ctypes.data, len(cols), cn, self.top_n, _lib.NUMPY_ARGSORT,
                                  None if colsum is None else colsum.ctypes.data, C.byref(h)))
        try:
            self._adoptSearch(h, cn)
        finally:
            lib().gk_search_destroy(h)

    def _adoptSearch(self, h, n_steps: int) -> None:
        """Results of a native search (``gk_search`` handle) -> ``self.result`` (one TypingResult per step)."""
        import ctypes as C
        from ._lib import check, lib
        m = self._model
        if self._colsum_all is None:
            self._colsum_all = np.empty(m.n_allele, dtype=np.float64)
            check(lib().gk_search_colsum(h, self._colsum_all.ctypes.data))
        for step in range(n_steps):
            n, rows, bounded = C.c_int32(), C.c_int64(), C.c_int32()
            check(lib().gk_search_info(h, step, C.byref(n), C.byref(rows), C.byref(bounded)))
            k, c = int(rows.value), int(n.value)
            value = np.empty(k, dtype=np.float64)
            sum_indv, frac = np.empty((k, c), dtype=np.float64), np.empty((k, c), dtype=np.float64)
            ids32 = np.empty((k, c), dtype=np.int32)
            check(lib().gk_search_copy(h, step, value.ctypes.data, sum_indv.ctypes.data, ids32.ctypes.data,
                                       frac.ctypes.data))
            ids = ids32.astype(np.int64)
            if step:
                SEARCH_STATS["bounded" if bounded.value else "redone_exactly"] += 1
            self.result.append(TypingResult(
                n=c, value=value, value_sum_indv=sum_indv, allele_id=ids,
                allele_name=LazyNames(ids, self.id_to_allele), allele_prob=LazyAlleleProb([(m, ids)]),
                fraction=frac if step else np.ones(ids.shape), fraction_uniq=np.ones(ids.shape)))
        if m.dev.call_log is not None:
            self._logLaunches(h)

    def adoptSearches(self, handles: list, keep_log: bool = True) -> "SearchSteps":
        """Every step of MANY native searches on this model's table (the candidate searches of exon-first: hundreds per
        sample) through ONE library call (``gk_search_export``); the steps become ``TypingResult`` objects when they are
        read (``SearchSteps``)."""
        import ctypes as C
        from ._lib import check, lib
        m = self._model
        n = len(handles)
        arr = (C.c_void_p * max(n, 1))(*handles)
        totals = np.zeros(3, dtype=np.int64)
        check(lib().gk_search_export(arr, n, totals.ctypes.data, None, None, None, None, None))
        n_steps, n_rows, n_cells = (int(x) for x in totals)
        meta = np.zeros(n + 3 * n_steps, dtype=np.int64)
        value = np.empty(n_rows, dtype=np.float64)
        sum_indv, frac = np.empty(n_cells, dtype=np.float64), np.empty(n_cells, dtype=np.float64)
        ids = np.empty(n_cells, dtype=np.int32)
        check(lib().gk_search_export(arr, n, totals.ctypes.data, meta.ctypes.data, value.ctypes.data, sum_indv.ctypes.data,
                                     frac.ctypes.data, ids.ctypes.data))
        steps = SearchSteps(self, meta[:n], meta[n:].reshape(-1, 3), value, sum_indv, frac, ids.astype(np.int64))
        later = steps.step_in_search > 0
        SEARCH_STATS["bounded"] += int(np.count_nonzero(later & (steps.bounded != 0)))
        SEARCH_STATS["redone_exactly"] += int(np.count_nonzero(later & (steps.bounded == 0)))
        if keep_log and m.dev.call_log is not None:
            for h in handles:
                self._logLaunches(C.c_void_p(h))
        return steps

    def _logLaunches(self, h) -> None:
        """Launch geometries of a native search into the device's call log (bench roofline)."""
        import ctypes as C
        from ._lib import check, lib
        m = self._model
        n_log = C.c_int64()
        check(lib().gk_search_log(h, None, 0, C.byref(n_log)))
        raw = np.empty(n_log.value, dtype=np.int64)
        check(lib().gk_search_log(h, raw.ctypes.data, len(raw), C.byref(n_log)))
        for kind, a, b, c_, d, e, f in raw.reshape(-1, 7).tolist():
            if kind == 0:      # column sums are a launch of their own kernel (no previous sets, one "set")
                m.dev.call_log.append(("colsum_chunks" if b == 1 and c_ == 0 else "maxsum_chunks", a, b, c_, d, e, bool(f)))
            elif kind == 1:
                m.dev.call_log.append(("minsum_sad", a, b, c_, d, False))
            else:
                m.dev.call_log.append(("setsum_leaves" if kind == 3 else "fraction_chunks", a, b, c_, d))

    def geneJob(self, cn: int, verdict: bool | None = None):
        """(``_lib.GeneJob`` for ``gk_sample_search``, homozygous?) of a model built with ``_defer_launch``: the
        zygosity decision of ``typing`` (383-410) is taken here, the table and the search run in the library.
        ``verdict``: what ``_isHomozygous(cn)`` says, when the caller asked for all genes at once."""
        from ._lib import GeneJob
        if cn < 1:
            raise ValueError(f"CN should be >= 1, got {cn}")
        if self.force_homo is not None:
            homo = self.force_homo
        else:
            homo = self._isHomozygous(cn) if verdict is None else verdict
        m = self._model
        vbeg, vend, mask, words = m._geom
        job = GeneJob(d_rows=m.rows.ptr, n_rows=m.n_rows, d_mask=mask.ptr, d_L=m._L.ptr if m._L else 0,
                      d_miss8=m.miss8.ptr if m.miss8 else 0, ldm=m.ldm, d_msum=m.msum.ptr if m.msum else 0,
                      d_flags=m._bound_flags.ptr if m._bound_flags else 0, d_lidx=m.lidx.ptr if m.lidx else 0,
                      vbeg=vbeg, vend=vend, words=words, n_allele=m.n_allele, n_steps=1 if homo else cn,
                      top_n=self.top_n, bound_ok=0, passes=0, indexed=0, patches=0, table_of=-1, n_step_cols=0,
                      step_cols=None, step_cols_off=None)
        return job, homo

    def adoptJob(self, job, handle, cn: int, homo: bool) -> TypingResult:
        """What ``typing(cn)`` leaves behind, from the gene's part of ``gk_sample_search``."""
        m = self._model
        m._bound_ok = bool(job.bound_ok)
        m._indexed = bool(job.indexed)
        m._known_at_launch = -1
        if m.dev.call_log is not None:
            per_row = m.tab.n_ids / max(m.tab.n_valid, 1)
            for _ in range(max(1, int(job.passes))):
                m.dev.call_log.append(("compat_kernel", m.n_rows, m.n_allele, per_row * m.n_rows, 2 if m._indexed else 8))
        self.result = []
        self._adoptSearch(handle, 1 if homo else cn)
        if homo:
            self.addHomoResultForCn(cn)
        self.result[-1].print()
        return self.result[-1]

    def adoptTable(self, job, handle) -> None:
        """A table-only job of ``gk_sample_search`` (n_steps == 0): the model's table is final and its column sums are
        known; the searches on it were other jobs of the call."""
        import ctypes as C
        from ._lib import check, lib
        m = self._model
        m._bound_ok = bool(job.bound_ok)
        m._indexed = False
        m._known_at_launch = -1
        if m.dev.call_log is not None:
            per_row = m.tab.n_ids / max(m.tab.n_valid, 1)
            for _ in range(max(1, int(job.passes))):
                m.dev.call_log.append(("compat_kernel", m.n_rows, m.n_allele, per_row * m.n_rows, 8))
            self._logLaunches(handle)       # the column sums of the whole table: this job's launch, nobody else logs it
        self._colsum_all = np.empty(m.n_allele, dtype=np.float64)
        check(lib().gk_search_colsum(handle, self._colsum_all.ctypes.data))
        self.result = []

    def addHomoResultForCn(self, cn: int) -> None:
        if cn > 1:
            self.result.append(self.createHomoResult(self.result[0], cn))

    @staticmethod
    def createHomoResult(cn1_result: TypingResult, cn: int) -> TypingResult:
        if cn <= 1:
            raise ValueError(f"CN should be > 1, got {cn}")
        if cn1_result.isFail():
            # documented deviation: the reference raises numpy.AxisError here
            e = np.array([])
            return TypingResult(cn, e, e, e, [], e, e, e)
        m = len(cn1_result.value)
        return TypingResult(
            n=cn, value=cn1_result.value * cn,
            value_sum_indv=np.repeat(cn1_result.value_sum_indv, cn, axis=1),
            allele_id=np.repeat(cn1_result.allele_id, cn, axis=1),
            allele_name=[[row[0]] * cn for row in cn1_result.allele_name],
            allele_prob=cn1_result.allele_prob,
            fraction=np.ones((m, cn)) / cn, fraction_uniq=np.ones((m, cn)) / cn)

    @staticmethod
    def uniqueAllele(data: np.ndarray) -> np.ndarray:
        return firstOccurrence(np.asarray(data), int(np.max(data)) + 1 if np.size(data) else 1)

    def _colsums(self) -> np.ndarray:
        if self._colsum_all is None:
            self.finish()
            self._colsum_all = self._model.colsum(np.arange(self._model.n_allele))
        return self._colsum_all

    def addCandidate(self, candidate_allele: Optional[list[str]] = None) -> TypingResult:
        """One copy-number step (478-598).  The step is written as a generator that yields its device
        requests (``_candidateSteps``); here they are served one by one, ``runLockstep`` serves the
        same requests of many independent searches with one launch each."""
        steps = self._candidateSteps(candidate_allele)
        try:
            request = next(steps)
            while True:
                request = steps.send(self._serve([request])[0])
        except StopIteration as done:
            return done.value

    def _serve(self, requests: list[tuple]) -> list[np.ndarray]:
        """Answer device requests of one kind -- ("maxsum", prev_ids, cols) or ("fraction", ids) -- that
        all refer to this model's table, with ONE launch: the sets are stacked (and the columns united),
        every requester gets its own block back.  Each value depends on its set, its column and the
        read order only, so the blocks hold the same bits as separate launches would."""
        m = self._model
        kind = requests[0][0]
        assert all(r[0] == kind for r in requests)
        if kind == "fraction":
            ids = [np.asarray(r[1], dtype=np.int64) for r in requests]
            if len(ids) == 1:
                return [m.fraction(ids[0])]
            uniq, back = np.unique(np.concatenate(ids), axis=0, return_inverse=True)   # searches share sets
            out = m.fraction(uniq)[back.ravel()]
            cuts = np.cumsum([len(x) for x in ids])[:-1]
            return np.split(out, cuts)
        prev = [np.asarray(r[1], dtype=np.int64) for r in requests]
        cols = [np.asarray(r[2], dtype=np.int64) for r in requests]
        if len(requests) == 1:
            return [m.maxsum(prev[0], cols[0])]
        union = np.unique(np.concatenate(cols))
        uniq, back = np.unique(np.concatenate(prev), axis=0, return_inverse=True)       # searches share sets
        out = m.maxsum(uniq, union)
        back = back.ravel()
        blocks, row = [], 0
        for p_, c_ in zip(prev, cols):
            blocks.append(out[back[row:row + len(p_)]][:, np.searchsorted(union, c_)])
            row += len(p_)
        return blocks

    @staticmethod
    def runLockstep(searches: list[tuple["AlleleTyping", list]]) -> None:
        """Run ``model.addCandidate(c)`` for every c of every (model, [c, ...]) entry, where all models
        are forks of one table: searches advance together, and the device requests they make at the
        same point are served by one launch (``_serve``).  Same results as running them one by one."""
        if not searches:
            return
        if len(searches) > 64:       # bound the stacked launches (their partial sums scale with the sets)
            for i in range(0, len(searches), 64):
                AlleleTyping.runLockstep(searches[i:i + 64])
            return
        owner = searches[0][0]
        n_round = max(len(c) for _, c in searches)
        for k in range(n_round):
            # searches in lock-step share launches of the exact kernels; the integer bound is per search
            active = [(model, model._candidateSteps(cands[k], allow_bound=False))
                      for model, cands in searches if k < len(cands)]
            pending = []
            for model, gen in active:
                try:
                    pending.append((gen, next(gen)))
                except StopIteration:
                    pass
            while pending:
                kinds = sorted({req[0] for _, req in pending})
                nxt = []
                for kind in kinds:
                    group = [(g, r) for g, r in pending if r[0] == kind]
                    answers = owner._serve([r for _, r in group])
                    for (g, _), ans in zip(group, answers):
                        try:
                            nxt.append((g, g.send(ans)))
                        except StopIteration:
                            pass
                pending = nxt

    def _boundedStep(self, prev: TypingResult, prev_ids: np.ndarray, cols: np.ndarray) -> Optional[TypingResult]:
        """One copy-number step through the integer bound (csrc/gk_bound.hip), or None when the step has to be
        done with float64 sums for every candidate.

        The reference sorts the float64 scores of ALL N first-occurrence candidates (567) and keeps those that
        reach the top_n-th value.  Scores of sets with different mismatch totals M differ by ~3 per unit, far
        beyond float noise, so that head lies inside {M <= M_T}: the device selects these sets by M and only
        their float64 values, per-allele sums and shares are formed -- the same numbers as in the full table.
        What the full table would add is numpy's argsort order among EQUAL values; it matters only where it
        is visible in the result: a tie across a cut (which tied sets survive) or between rows that agree in
        all three ranking keys (their order in the result).  Such a step returns None and is redone exactly."""
        m, T = self._model, self.top_n
        first = firstOfSets(prev_ids, cols, m.n_allele)
        N = int(np.count_nonzero(first))
        if N == 0:
            return None
        sel = m.boundStep(prev_ids, cols, first, T, cap=4 * T + 4096)
        if sel is None:
            return None                                     # a flood of ties at the cut
        _, _, idx, _ = sel
        t_idx, a_idx = np.divmod(idx, len(cols))
        ids = np.concatenate([prev_ids[t_idx], cols[a_idx][:, None]], axis=1)
        value, frac = m.setsum(ids)
        # the reference's head: rows of the sorted table that reach the top_n-th value (567, 605-609)
        n_top = min(max(T, N // 5), N)
        if N > T:
            v_cut = np.partition(value, len(value) - T)[len(value) - T]          # T-th largest
            head = np.flatnonzero(value >= v_cut)
            if n_top > T:
                if len(head) > n_top:
                    return None                             # ties run past the N // 5 cut
            elif len(head) != T:
                return None                                 # ties across the top_n cut
        else:
            head = np.arange(len(value))
        ids, value, frac = ids[head], value[head], frac[head]
        sum_indv = self._colsums()[ids]
        key1, key2 = -value, -sum_indv.sum(axis=1)
        if len(head) > T:
            b = np.lexsort((key2, key1))[T - 1]
            contend = np.nonzero((key1 < key1[b]) | ((key1 == key1[b]) & (key2 <= key2[b])))[0]
        else:
            contend = np.arange(len(head))
        fc = frac[contend]
        uneven = np.abs(fc - fc.mean(axis=1, keepdims=True)).sum(axis=1)
        k1, k2 = key1[contend], key2[contend]
        sub = np.lexsort((uneven, k2, k1))
        look = sub[:T + 1]                                  # rows of the result and the first one cut off
        same = (k1[look][1:] == k1[look][:-1]) & (k2[look][1:] == k2[look][:-1]) & \
               (uneven[look][1:] == uneven[look][:-1])
        if same.any():
            return None                                     # rows equal in every key: their order is argsort's
        sub = sub[:T]
        order = contend[sub]
        kept = ids[order]
        res = TypingResult(
            n=prev.n + 1, value=value[order], value_sum_indv=sum_indv[order], allele_id=kept,
            allele_name=LazyNames(kept, self.id_to_allele), allele_prob=LazyAlleleProb([(m, kept)]),
            fraction=fc[sub], fraction_uniq=np.ones(kept.shape))
        return res

    def _candidateSteps(self, candidate_allele: Optional[list[str]] = None, allow_bound: bool = True):
        m = self._model
        if not m.n_rows:
            logger.warning("[Allele] Empty reads for typing. Skip")
            e = np.array([])
            self.result.append(TypingResult(len(self.result) + 1, e, e, e, [], e, e, e))
            return self.result[-1]
        if candidate_allele is None:
            cols = np.arange(m.n_allele)
        else:
            cols = np.array([self.allele_to_id[a] for a in candidate_allele])

        if not self.result:
            score = self._colsums()[cols]                       # = log_probs[:, cols].sum(axis=0)
            top = np.argsort(score)[::-1][:self.top_n]
            top_ids = cols[:, None][top]
            res = TypingResult(
                n=1, value=score[top], value_sum_indv=score[top][:, None], allele_id=top_ids,
                allele_name=LazyNames(top_ids, self.id_to_allele), allele_prob=LazyAlleleProb([(m, top_ids)]),
                fraction=np.ones(top_ids.shape), fraction_uniq=np.ones(top_ids.shape))
            self.result.append(res)
            return res

        prev = self.result[-1]
        prev_ids = np.asarray(prev.allele_id, dtype=np.int64)
        if allow_bound and m.boundOk and len(np.unique(cols)) == len(cols):
            res = self._boundedStep(prev, prev_ids, np.asarray(cols, dtype=np.int64))
            SEARCH_STATS["bounded" if res is not None else "redone_exactly"] += 1
            if res is not None:
                self.result.append(res)
                return res
        table = getattr(self, "_pair_table", None)
        if table is not None and prev_ids.shape[1] == 1:
            score = table[prev_ids[:, 0]][:, np.asarray(cols, dtype=np.int64)].ravel()
        else:
            score = (yield ("maxsum", prev_ids, cols)).ravel()
        # candidate (t, a) sits at flat index t * len(cols) + a
        # first occurrence of every allele multiset in (t-major, a-minor) order (uniqueAllele 456-476),
        # from keys built by broadcasting -- the (T*A) x CN id table is never materialised
        cols = np.asarray(cols, dtype=np.int64)
        first = np.flatnonzero(firstOfSets(prev_ids, cols, m.n_allele))
        score = score[first]
        n_keep = max(self.top_n, score.shape[0] // 5)
        top = np.argsort(score)[::-1][:n_keep]                  # numpy's own order among tied scores
        value = score[top]
        # The reference keeps these K = max(top_n, N // 5) sets, computes their abundance fractions and
        # then keeps the first top_n under the stable order (-value, -sum of per-allele sums,
        # unevenness).  Unevenness is the last key, so only rows not worse than the top_n-th row on the
        # first two keys can make the cut: per-allele sums are formed for the rows that reach the
        # top_n-th value, fractions for the contenders among them (same final rows, same order).
        if len(top) > self.top_n:
            head = int(np.count_nonzero(value >= value[self.top_n - 1]))     # value is descending
        else:
            head = len(top)
        top, value = top[:head], value[:head]
        t_idx, a_idx = np.divmod(first[top], len(cols))
        top_ids = np.concatenate([prev_ids[t_idx], cols[a_idx][:, None]], axis=1)
        sum_indv = self._colsums()[top_ids]                     # = log_probs[:, ids].sum(axis=0)
        key1, key2 = -value, -sum_indv.sum(axis=1)
        if head > self.top_n:
            b = np.lexsort((key2, key1))[self.top_n - 1]
            contend = np.nonzero((key1 < key1[b]) | ((key1 == key1[b]) & (key2 <= key2[b])))[0]
        else:
            contend = np.arange(head)
        frac = yield ("fraction", top_ids[contend])
        uneven = np.abs(frac - frac.mean(axis=1, keepdims=True)).sum(axis=1)
        sub = np.lexsort((uneven, key2[contend], key1[contend]))[:self.top_n]
        order = contend[sub]
        kept = top_ids[order]
        res = TypingResult(
            n=prev.n + 1, value=value[order], value_sum_indv=sum_indv[order], allele_id=kept,
            allele_name=LazyNames(kept, self.id_to_allele), allele_prob=LazyAlleleProb([(m, kept)]),
            fraction=frac[sub], fraction_uniq=np.ones(kept.shape))
        self.result.append(res)
        return res

    # ---- homozygosity test on device counts
    def _variantCounts(self) -> tuple[np.ndarray, np.ndarray, np.ndarray]:
        """(ordinals, positive tally, negative tally) of the corrected lists, nonzero entries only."""
        rs = self._readset
        tab = rs.tab
        cnt = self._tally
        if cnt is not None and self._surviving is not None:   # came with the sample-wide tallies (Tabulation.prepared)
            return self._surviving
        if cnt is None:   # no correction pass ran: tally now
            cnt = self._dev.alloc(2 * tab.n_var_total, np.uint32)
            tab.countVariants(rs.rows, rs.n_rows, rs.vflag, cnt, self._span)
        out = tab.survivingCounts(cnt, rs.vflag, gene=self._tally_gene if cnt is self._tally else None)
        if cnt is not self._tally:
            cnt.free()
        return out

    @staticmethod
    def _siteVerdict(ent_pos: np.ndarray, ent_code: np.ndarray, ent_neg: np.ndarray, ent_cnt: np.ndarray,
                     cn: int) -> bool:
        """Lines 835-857 on flat (position, label code, negative?, count) observations, vectorised.

        Per position the reference sums counts per label, skips positions with one label or only
        negative labels, keeps counts > 3, needs their total >= 20, and calls the position
        heterozygous when the second largest share is > 0.1 and > 1 / (2 cn)."""
        if not len(ent_pos):
            return True
        upos, site = np.unique(ent_pos, return_inverse=True)
        key = (site.astype(np.int64) << 32) | (ent_code.astype(np.int64) << 1) | ent_neg.astype(np.int64)
        ukey, inv = np.unique(key, return_inverse=True)
        count = np.bincount(inv, weights=ent_cnt.astype(np.float64), minlength=len(ukey)).astype(np.int64)
        usite = ukey >> 32
        n_site = len(upos)
        n_label = np.bincount(usite, minlength=n_site)
        n_positive = np.bincount(usite[(ukey & 1) == 0], minlength=n_site)
        big = count > 3
        total = np.bincount(usite[big], weights=count[big].astype(np.float64), minlength=n_site)
        # second largest kept count of every position: order labels by (position, -count)
        order = np.lexsort((-count, usite))
        so_site, so_count, so_big = usite[order], count[order], big[order]
        first = np.r_[True, so_site[1:] != so_site[:-1]]
        second_idx = np.flatnonzero(first) + 1
        second_idx = second_idx[second_idx < len(order)]
        ok = (so_site[second_idx] == so_site[second_idx - 1]) & so_big[second_idx]
        second = np.zeros(n_site)
        second[so_site[second_idx[ok]]] = so_count[second_idx[ok]]
        with np.errstate(divide="ignore", invalid="ignore"):
            share = second / total                      # the reference's c / total in float64
        het = (n_label >= 2) & (n_positive >= 1) & (total >= 20) & (second > 0) & (share > 0.1) & \
              (share > (1 / (cn * 2)))
        return not bool(het.any())

    def _isHomozygous(self, cn: int) -> bool:
        """isHomozygous (807-857) from per-variant positive / negative tallies of the kept reads."""
        if cn <= 1:
            return False
        seen, pos, neg = self._variantCounts()       # surviving variants only, compacted on the device
        tab = self._readset.tab
        tables = tab.labelTables()
        if tables is not None:      # labels, the screen of lines 835-840 and the verdict in one native call
            import ctypes as C
            from ._lib import check, lib
            keys_all, ins_code = tables
            o_ = np.ascontiguousarray(seen, dtype=np.int32)
            p_, n_ = np.ascontiguousarray(pos, dtype=np.uint32), np.ascontiguousarray(neg, dtype=np.uint32)
            verdict = C.c_int32()
            check(lib().gk_site_verdict_tallies(keys_all.ctypes.data, len(keys_all), ins_code.ctypes.data, len(ins_code),
                                                o_.ctypes.data, p_.ctypes.data, n_.ctypes.data, len(o_), cn,
                                                C.byref(verdict)))
            return bool(verdict.value)
        pos, neg = pos.astype(np.int64), neg.astype(np.int64)
        fields = tab.labelCodes(seen)
        if fields is not None:
            # vectorised screen: only positions with >= 2 distinct observations, one of them positive,
            # can change the verdict (lines 835-840 skip the rest)
            vpos, code, is_del = fields
            keep = ~is_del
            vpos, code, pos, neg = vpos[keep], code[keep], pos[keep], neg[keep]
            hp, hn = pos > 0, neg > 0
            ent_pos = np.concatenate([vpos[hp], vpos[hn]])
            ent_code = np.concatenate([code[hp], code[hn]])
            ent_neg = np.concatenate([np.zeros(int(hp.sum()), bool), np.ones(int(hn.sum()), bool)])
            ent_cnt = np.concatenate([pos[hp], neg[hn]])
            import ctypes as C
            from ._lib import check, lib
            p_, c_ = np.ascontiguousarray(ent_pos, dtype=np.int64), np.ascontiguousarray(ent_code, dtype=np.int64)
            n_, k_ = np.ascontiguousarray(ent_neg, dtype=np.uint8), np.ascontiguousarray(ent_cnt, dtype=np.int64)
            verdict = C.c_int32()
            check(lib().gk_site_verdict(p_.ctypes.data, c_.ctypes.data, n_.ctypes.data, k_.ctypes.data, len(p_), cn,
                                        C.byref(verdict)))
            return bool(verdict.value)
        else:
            site = defaultdict(lambda: defaultdict(int))
            for (vpos, typ, label), np_, nn_ in zip(tab.describe(seen), pos.tolist(), neg.tolist()):
                if typ == "deletion":
                    continue
                if np_:
                    site[vpos][label] += np_
                if nn_:
                    site[vpos][f"*{label}"] += nn_
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
            if major[1] > (1 / (cn * 2)):
                hits += 1
        return hits == 0


class AlleleTypingExonFirst(AlleleTyping):
    """Exon variants select candidate allele groups, the full model refines them (622-797)."""

    def __init__(self, reads, variants: list[Variant], top_n: int = 300, exon_only: bool = False,
                 candidate_set_threshold: float = 1.0, variant_correction: bool = True,
                 force_homo: bool | None = None, *, device: Device | None = None, logs: LogTable | None = None,
                 _vbeg: int = 0, _n_span: int | None = None, _mask: DeviceBuffer | None = None,
                 _alleles: list[str] | None = None, _exon_flags: np.ndarray | None = None, _novel=None,
                 _group_cache: dict | None = None):
        """``_group_cache``: a dict owned by the caller (one per gene of an index); the exon allele
        groups, the regrouped variants and their bit rows depend on the index only and are kept there."""
        if isinstance(reads, ReadSet):
            base = reads
            template = None
        else:
            device = device or Device()
            template = list(reads)
            base, _ = ReadSet.fromLists(device, template, variants)
        tab = base.tab
        dev = tab.dev
        n_span = len(variants) if _n_span is None else _n_span
        # exon reads = same rows, non-exon ids masked out (removeIntronVariant 703-714)
        if _exon_flags is None:
            in_exon = {str(v.id) for v in variants if v.in_exon}
            _exon_flags = np.array([0 if n in in_exon else 3 for n in tab.idNames()], dtype=np.uint8)
        exon_flags = dev.put(_exon_flags)
        if variant_correction:
            tab.errorCorrection(base.rows, base.n_rows, exon_flags, span=(_vbeg, _vbeg + n_span))
        exon_set = ReadSet(tab, base.rows, base.n_rows, exon_flags)

        # alleles sharing one exon-variant set become one group (649-659)
        self.allele_group, grouped, group_names, exon_mask = self.exonGroups(variants, n_span, _group_cache, dev)
        # the base class runs errorCorrection once more on the exon lists (line 664: default True)
        super().__init__(exon_set, grouped, force_homo=force_homo, top_n=top_n, logs=logs,
                         _vbeg=_vbeg, _n_span=n_span, _mask=exon_mask, _alleles=group_names, _defer_log=True, _novel=_novel)
        self._template = template
        self.candidate_set_threshold = candidate_set_threshold
        if not exon_only:
            full_set = ReadSet(tab, base.rows, base.n_rows, base.vflag)
            self.full_model: AlleleTyping | None = AlleleTyping(
                full_set, variants, force_homo=force_homo, top_n=top_n // 5,
                variant_correction=variant_correction, logs=logs, _vbeg=_vbeg, _n_span=n_span, _mask=_mask,
                _alleles=_alleles, _defer_log=True, _novel=_novel)
            self.full_model._template = template
        else:
            self.full_model = None
        self.finish()
        if self.full_model is not None:
            self.full_model.finish()

    @classmethod
    def exonGroups(cls, variants: list[Variant], n_span: int, cache: dict | None, dev: Device):
        """(group name -> member alleles, the variants with their alleles regrouped, sorted group names, their bit rows on
        ``dev``'s GPU) of a gene (649-659): index-only data, computed once per gene and process (``cache``: a dict the
        caller owns, one per gene of an index)."""
        cache = cache if cache is not None else {}
        with _GROUP_CACHE_LOCK:       # the lanes of a process type the same gene of different samples at the same time
            if "device_mask" not in cache:
                exon_variants = [v for v in variants if v.in_exon]
                groups = cls.aggrVariantsByAllele(exon_variants)
                rest = cls.collectAlleleNames(variants) - cls.collectAlleleNames(exon_variants)
                if rest:
                    groups[tuple()] = sorted(rest)
                cache["allele_group"] = {"|".join(a): a for a in groups.values()}
                inverse = cls.createInverseMapping(cache["allele_group"])
                cache["grouped"] = cls.removeDuplicateAllele(variants, inverse)
                cache["group_names"] = sorted(cls.collectAlleleNames(cache["grouped"]))
                cache["mask"] = buildMask(cache["grouped"][:n_span], cache["group_names"])
                cache["device_mask"] = {}
            exon_mask = cache["device_mask"].get(dev.ordinal)
            if exon_mask is None:
                exon_mask = cache["device_mask"][dev.ordinal] = dev.put(cache["mask"])
            return cache["allele_group"], cache["grouped"], cache["group_names"], exon_mask

    @staticmethod
    def mergeCandidates(finals: list["TypingResult"]) -> "TypingResult":
        """The final results of the candidate searches, concatenated and ranked again (783-793)."""
        return TypingResult(
            n=finals[0].n,
            value=np.concatenate([f.value for f in finals]),
            value_sum_indv=np.concatenate([f.value_sum_indv for f in finals]),
            allele_id=np.concatenate([f.allele_id for f in finals]),
            allele_name=list(chain.from_iterable(f.allele_name for f in finals)),
            allele_prob=LazyAlleleProb([p for f in finals for p in f.allele_prob.parts]),
            fraction=np.concatenate([f.fraction for f in finals]),
            fraction_uniq=np.concatenate([f.fraction for f in finals]),
        ).sortByScoreAndEveness()

    @staticmethod
    def aggrVariantsByAllele(variants: list[Variant]) -> dict[tuple[str, ...], list[str]]:
        per_allele: dict[str, list[str]] = defaultdict(list)
        for v in variants:
            for a in v.allele:
                per_allele[a].append(str(v.id))
        by_set: dict[tuple[str, ...], list[str]] = defaultdict(list)
        for a, ids in per_allele.items():
            by_set[tuple(sorted(set(ids)))].append(a)
        return by_set

    @staticmethod
    def createInverseMapping(allele_group: dict[str, list[str]]) -> dict[str, str]:
        return {a: g for g, members in allele_group.items() for a in members}

    @staticmethod
    def removeDuplicateAllele(variants: list[Variant], allele_map: dict[str, str]) -> list[Variant]:
        import copy
        out = []
        for v in variants:
            w = copy.copy(v)
            w.allele = list(set(filter(None, [allele_map.get(a, "") for a in v.allele])))
            out.append(w)
        return out

    def typingIntron(self, exon_candidates: list[list[str]]) -> AlleleTyping:
        assert self.full_model
        model = self.full_model.fork()
        for cand in exon_candidates:
            model.addCandidate(cand)
        return model

    def typing(self, cn: int) -> TypingResult:
        result = super().typing(cn)
        result.setNameGroup(self.allele_group)
        logger.debug("[Allele] Typing exon:")
        if self.full_model is None:
            return result
        assert cn == result.n
        if not result.value.shape[0]:
            logger.warning("[Allele] Cannot typing with exon-only reads. Typing with exon+intron")
            return self.full_model.typing(cn)
        # every exon candidate is refined by an independent search on the full model (741-747):
        # the searches run in lock-step so that their device requests share launches
        finals = []
        assert self.full_model
        if self.full_model._model.n_rows:
            self.full_model._colsums()      # column sums are shared by the forks, not recomputed per search
        searches = [(self.full_model.fork(), result.allele_name_group[i])
                    for i in result.topRank(threshold=self.candidate_set_threshold)]
        AlleleTyping.runLockstep(searches)
        for model, _ in searches:
            self.result.extend(model.result)
            finals.append(model.result[-1])
        logger.debug(f"[Allele] Intron Candidate {len(finals)} Done")
        merged = self.mergeCandidates(finals)
        self.result.append(merged)
        logger.debug("[Allele] Typing intron + exon")
        merged.print()
        return merged


def isHetrozygous(gene: str) -> bool:
    """Genes typed as heterozygous by name (800-804)."""
    return "2DL1S1" in gene or "2DL5" in gene


def isHomozygous(reads, variants_map: dict[str, Variant], cn: int) -> bool:
    """Host version on ``PairRead`` lists, same rule as ``AlleleTyping._isHomozygous`` (807-857)."""
    if cn <= 1:
        return False
    site: dict[int, dict[str, int]] = defaultdict(lambda: defaultdict(int))
    for r in reads:
        for vid in chain(r.lpv, r.rpv):
            v = variants_map[vid]
            if v.typ != "deletion":
                site[v.pos][str(v.val)] += 1
        for vid in chain(r.lnv, r.rnv):
            v = variants_map[vid]
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
        if major[1] > (1 / (cn * 2)):
            hits += 1
    return hits == 0

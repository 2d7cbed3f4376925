"""
Strategy plug-in surface of the typing stage -- drop-in for ``graphkir/kir_typing.py``.

``selectKirTypingModel(method, filename_variant_json, **kwargs)`` (207-228) keeps its signature;
``filename_variant_json`` may also be a ``hisat2.SampleData`` that is already on the device (the
pipeline passes that, skipping the JSON round trip).  Methods: ``full`` (alias ``pv``),
``exonfirst[_<threshold>]`` (alias ``pv_exonfirst_<threshold>``), ``em`` (alias ``report`` -- the
reference CLI forwards ``report`` to a factory that does not know it, main.py:192 / kir_typing.py:228).
"""
from __future__ import annotations

import json
import threading
from concurrent.futures import ThreadPoolExecutor
from typing import Any

import numpy as np

from ._lib import Device
from .hisat2 import SampleData, loadReadsAndVariantsData
from .typing_em import Hisat2AlleleResult, hisat2TypingPerGene
from .typing_mulit_allele import (AlleleTyping, AlleleTypingExonFirst, ReadSet, StepList, isHetrozygous,
                                  sharedLogTable)
from .utils import NumpyEncoder, logger

_device: Device | None = None


def defaultDevice() -> Device:
    """One context per process (LOCAL_RANK picks the GPU)."""
    global _device
    if _device is None:
        _device = Device()
    return _device


_LOAD_LOCK = threading.Lock()


def _sample(source, dev: Device | None) -> SampleData:
    """A sample that is a hand-off file is uploaded (and tabulated) on ``dev`` or the process's default context.  A
    context serves one host thread at a time and the typing lanes of a process (cohort.SampleTyper) all come here with
    the same one, so the upload of one file at a time: the lock is held until the context's stream has drained."""
    if isinstance(source, SampleData):
        return source
    dev = dev or defaultDevice()
    with _LOAD_LOCK:
        if str(source).endswith(".npz"):     # compact side-format (hisat2.writeCompact)
            from .hisat2 import loadCompact
            data = loadCompact(str(source), dev)
        else:
            data = SampleData.fromHost(dev, loadReadsAndVariantsData(source))
        dev.sync()
    return data


class Typing:
    """Common driver: loop genes of the CN table, collect calls and low-depth warnings (31-74)."""

    def __init__(self) -> None:
        self._result: dict[str, Any] = {}

    def typingPerGene(self, gene: str, cn: int) -> tuple[list[str], int]:
        raise NotImplementedError

    def typing(self, gene_cn: dict[str, int], min_reads_num: int = 100) -> tuple[list[str], list[str]]:
        predict_alleles, warning_genes = [], []
        for gene, cn in gene_cn.items():
            if not cn:
                continue
            alleles, reads_num = self.typingPerGene(gene, int(cn))
            predict_alleles.extend(alleles)
            if reads_num < min_reads_num:
                warning_genes.append(gene)
        return predict_alleles, warning_genes

    def save(self, filename: str) -> None:
        with open(filename, "w") as f:
            json.dump({gene: list(steps) for gene, steps in self._result.items()}, f, cls=NumpyEncoder)

    def getAllPossibleTyping(self) -> list[dict[Any, Any]]:
        raise NotImplementedError


def _lib_slice(buf, offset: int, count: int, dev):
    from ._lib import DeviceSlice
    return DeviceSlice(buf, offset, count, dev)


class _GeneView:
    """Per-gene handles into a tabulated sample (the gene's rows are selected on first use)."""

    def __init__(self, data: SampleData, gene: str, multiple: bool, tab=None):
        self.data, self.gene = data, gene
        idx = data.index
        g = idx.gene_id.get(gene)
        self.g = g
        tab = tab or data.tab
        self.tab = tab
        self._multiple = multiple
        self._rows = None
        if g is None:
            self._rows = (tab.dev.alloc(1, np.int32), 0)
            self.vbeg = self.n_span = 0
            self.mask, self.alleles, self.variants = None, [], []
            self.novel = lambda: []
            return
        t = idx.tables[g]
        self.vbeg, self.n_span = t.vbeg, t.vend - t.vbeg
        self.mask = tab.dindex.masks[g]
        self.alleles = t.alleles
        self.variants = idx.variants[t.vbeg:t.vend]          # index part; novel ones are built lazily
        self.novel = lambda: data.novelOfGene(gene)

    def _select(self):
        if self._rows is None:
            self._rows = self.tab.selectGene(self.g, self._multiple)
        return self._rows

    @property
    def rows(self):
        return self._select()[0]

    @property
    def n_rows(self) -> int:
        return self._select()[1]

    def groupCache(self) -> dict:
        """Per-gene store of the exon allele groups (index-only data, computed on first use)."""
        store = self.data.index.__dict__.setdefault("_exon_group_cache", {})
        return store.setdefault(self.gene, {})

    def exonFlags(self) -> np.ndarray:
        idx, tab = self.data.index, self.data.tab
        flags = np.full(tab.n_var_total, 3, dtype=np.uint8)
        flags[:idx.n_variant][idx.in_exon.astype(bool)] = 0
        return flags


class _OnLane(Typing):
    """A typer that runs on one typing lane of the process: its device context, its log table (needs ``self._data``)."""

    def __init__(self) -> None:
        super().__init__()
        self._local = threading.local()
        self.slot_base = 0     # worker context of this typer (cohort.SampleTyper gives every lane its own)
        self.tables_rewritten = 0   # compatibility tables written again because the sample brought products without a log10
        self.tables_patched = 0     # ... and tables whose new products were patched in place (gk_compat_patch)

    def _context(self):
        """(tabulation bound to this thread's context, its log table)."""
        tab = getattr(self._local, "tab", None)
        if tab is None:
            base = self._data.tab
            # never the tabulation's own context: a cohort run may already be tabulating the next
            # sample there (cohort.prefetched), and a context serves one host thread at a time
            dev = base.dev.worker(self.slot_base)
            tab = self._local.tab = base.on(dev)
            self._local.logs = sharedLogTable(dev)
        return tab, self._local.logs


_SEARCH_SLOTS: dict = {}
_SEARCH_SLOTS_LOCK = threading.Lock()


def _searchSlot():
    """Context manager that admits GK_SEARCH_SLOTS whole-sample searches of this process at a time (0 / unset: any
    number).  Two searches fill the GPU; a third next to them only lengthens all three (and the tail of a short run),
    while the preamble of the samples that wait is already done -- the next search starts the moment a slot is free."""
    import contextlib
    import os
    n = int(os.environ.get("GK_SEARCH_SLOTS", "0") or 0)
    if n <= 0:
        return contextlib.nullcontext()
    with _SEARCH_SLOTS_LOCK:
        sem = _SEARCH_SLOTS.get(n)
        if sem is None:
            sem = _SEARCH_SLOTS[n] = threading.BoundedSemaphore(n)
    return sem


class TypingWithPosNegAllele(_OnLane):
    """Likelihood typing with positive / negative variants (77-150)."""

    def __init__(self, filename_variant_json, top_n: int = 300, multiple: bool = False, exon_first: bool = False,
                 exon_only: bool = False, exon_candidate_threshold: float = .9, variant_correction: bool = False,
                 device: Device | None = None):
        super().__init__()
        self._data = _sample(filename_variant_json, device)
        self._multiple = multiple
        self._top_n = top_n
        self._exon_first, self._exon_only = exon_first, exon_only
        self._exon_candidate_threshold = exon_candidate_threshold
        self._variant_correction = variant_correction

    def typing(self, gene_cn: dict[str, int], min_reads_num: int = 100) -> tuple[list[str], list[str]]:
        """The plain likelihood strategy with the sample-wide preamble goes through ``gk_sample_search``: every gene's
        table and search in ONE library call on one stream (the genes advance in lock-step; ~10 waits per sample).
        Anything else (exon-only, no correction, index tables) types gene after gene on this lane (``typingPerGene``)."""
        if self._wholeSample():
            return self._typingWholeSample(gene_cn, min_reads_num)
        if self._wholeSampleExonFirst():
            return self._typingWholeSampleExonFirst(gene_cn, min_reads_num)
        return super().typing(gene_cn, min_reads_num)

    def _wholeSampleExonFirst(self) -> bool:
        import os
        from .engine import searchMode
        return (self._exon_first and not self._exon_only and os.environ.get("GK_INDEX_TABLE", "0") != "1"
                and searchMode() == "bound")

    def _typingWholeSampleExonFirst(self, gene_cn: dict[str, int], min_reads_num: int) -> tuple[list[str], list[str]]:
        """Exon-first for ALL genes of the sample on this thread and ONE stream, in two ``gk_sample_search`` calls
        (typing_mulit_allele.py:622-797 per gene; kir_typing.py:103-132 is the gene loop):

        1. the EXON models -- ids outside the exons dropped, the lists corrected twice (``Tabulation.prepared(exon=True)``),
           alleles with one exon-variant set merged into groups -- tables and searches of every gene, pipelined;
        2. the FULL tables of every gene as table-only jobs, and for every exon set that reaches the threshold (774) a
           candidate search on its gene's table whose k-th step offers the alleles of the set's k-th group (740-746),
           all of them pipelined on the marks of the stream like the genes of the plain strategy.
        A gene the two calls do not cover (no exon reads: the reference falls back to the full model there; no reads at
        all; not in the index) takes the per-gene path on this thread.  Same results as ``AlleleTypingExonFirst``."""
        import ctypes as C
        import os
        from . import _lib
        from ._lib import check, lib
        from .typing_mulit_allele import AlleleTypingExonFirst
        tab, logs = self._context()
        udev = tab.dev.urgent()
        prep_f = tab.prepared(udev, self._multiple)
        if prep_f is None:
            return super().typing(gene_cn, min_reads_num)
        prep_e = tab.prepared(udev, self._multiple, exon=True)
        top_n, threshold = self._top_n, self._exon_candidate_threshold
        todo = [(gene, int(cn)) for gene, cn in gene_cn.items() if cn]
        views = [_GeneView(self._data, gene, self._multiple, tab=tab) for gene, _ in todo]
        verd_e = self._zygosityVerdicts(tab, prep_e, [(v.g, cn) for v, (_, cn) in zip(views, todo)])

        def slice_of(prep, view):
            vflag, cnt, rows_all, off = prep[:4]
            a, b = int(off[view.g]), int(off[view.g + 1])
            rows = _lib_slice(rows_all, a, b - a, tab.dev)
            return rows, b - a, vflag, (rows, b - a, vflag, cnt, (view.g, view.vbeg, view.vbeg + view.n_span),
                                        type(tab).survivingOfGene(prep, view.g))

        def run(jobs_list, vflag):
            jobs = (_lib.GeneJob * len(jobs_list))(*jobs_list)
            handles = (C.c_void_p * len(jobs_list))()
            with _searchSlot():
                check(lib().gk_sample_search(tab.dev.ctx, None, 0, tab.handle, vflag.ptr, logs.handle, jobs, len(jobs_list),
                                             _lib.NUMPY_ARGSORT, _lib.NUMPY_LOG10, handles))
            return jobs, handles

        def destroy(handles):
            for h in handles:
                if h:
                    lib().gk_search_destroy(C.c_void_p(h))

        # ---- 1. the exon models
        plan: dict[int, dict] = {}             # position in todo -> what the two calls hold for the gene
        jobs1 = []
        for k, ((gene, cn), view) in enumerate(zip(todo, views)):
            if view.g is None or not view.alleles:
                continue
            rows_e, n_e, vflag_e, prepared_e = slice_of(prep_e, view)
            rows_f, n_f, vflag_f, prepared_f = slice_of(prep_f, view)
            if n_e == 0 or n_f == 0:
                continue                        # the per-gene path below (the reference's fall-backs live there)
            allele_group, grouped, group_names, exon_mask = AlleleTypingExonFirst.exonGroups(
                view.variants, view.n_span, view.groupCache(), tab.dev)
            force = False if isHetrozygous(gene) else None
            typ_e = AlleleTyping(ReadSet(tab, rows_e, n_e, vflag_e), grouped, force_homo=force, top_n=top_n,
                                 variant_correction=True, logs=logs, _vbeg=view.vbeg, _n_span=view.n_span, _mask=exon_mask,
                                 _alleles=group_names, _novel=view.novel, _prepared=prepared_e, _defer_launch=True)
            job, homo = typ_e.geneJob(cn, verd_e.get(view.g) if cn > 1 else False)
            plan[k] = {"typ_e": typ_e, "homo_e": homo, "job1": len(jobs1), "groups": allele_group,
                       "full": (rows_f, n_f, vflag_f, prepared_f), "force": force}
            jobs1.append(job)
        self.tables_rewritten = self.tables_patched = 0
        self.exon_info: dict[str, dict] = {}     # per gene: exon groups, exon sets found, candidate searches run
        if jobs1:
            jobs, handles = run(jobs1, prep_e[0])
            try:
                for k, p in plan.items():
                    q = p["job1"]
                    p["typ_e"].adoptJob(jobs[q], C.c_void_p(handles[q]), todo[k][1], p["homo_e"])
                    self.tables_rewritten += max(0, int(jobs[q].passes) - 1)
                    self.tables_patched += int(jobs[q].patches)
            finally:
                destroy(handles)
        # ---- 2. the full tables and the candidate searches on them
        jobs2, keep_alive = [], []
        for k, p in plan.items():
            (gene, cn), view = todo[k], views[k]
            result = p["typ_e"].result[-1]
            result.setNameGroup(p["groups"])
            rows_f, n_f, vflag_f, prepared_f = p["full"]
            full = AlleleTyping(ReadSet(tab, rows_f, n_f, vflag_f), view.variants, force_homo=p["force"], top_n=top_n // 5,
                                variant_correction=True, logs=logs, _vbeg=view.vbeg, _n_span=view.n_span, _mask=view.mask,
                                _alleles=view.alleles, _novel=view.novel, _prepared=prepared_f, _defer_launch=True)
            p["full_model"] = full
            if not result.value.shape[0]:
                # no exon set: the reference types the gene with the full model (typing_mulit_allele.py:757-759)
                logger.warning("[Allele] Cannot typing with exon-only reads. Typing with exon+intron")
                p["fallback"] = True
                continue
            ranks = list(result.topRank(threshold=threshold))
            self.exon_info[gene] = {"exon_groups": len(p["groups"]), "exon_sets": int(result.value.shape[0]),
                                    "candidates": len(ranks)}
            if len(ranks) > 1024:
                p["per_gene"] = True            # a flood of tied exon sets: the per-gene path stacks their searches per launch
                continue
            job, _ = full.geneJob(cn, False)
            job.n_steps = 0                     # table + column sums; the searches are the jobs behind it
            p["job2"] = len(jobs2)
            jobs2.append(job)
            p["cands"] = []
            for i in ranks:
                steps = [np.ascontiguousarray([full.allele_to_id[a] for a in names], dtype=np.int32)
                         for names in result.allele_name_group[i]]
                cols = np.ascontiguousarray(np.concatenate(steps), dtype=np.int32)
                offs = np.ascontiguousarray(np.concatenate([[0], np.cumsum([len(x) for x in steps])]), dtype=np.int32)
                keep_alive += [cols, offs]
                cand = _lib.GeneJob(d_rows=job.d_rows, n_rows=job.n_rows, d_mask=job.d_mask, d_L=0, d_miss8=0, ldm=job.ldm,
                                    d_msum=0, d_flags=0, d_lidx=0, vbeg=job.vbeg, vend=job.vend, words=job.words,
                                    n_allele=job.n_allele, n_steps=len(steps), top_n=top_n // 5, bound_ok=0, passes=0,
                                    indexed=0, patches=0, table_of=p["job2"], n_step_cols=len(steps),
                                    step_cols=cols.ctypes.data, step_cols_off=offs.ctypes.data)
                p["cands"].append((len(jobs2), len(steps)))
                jobs2.append(cand)
        if jobs2:
            jobs, handles = run(jobs2, prep_f[0])
            try:
                for k, p in plan.items():
                    if "job2" not in p:
                        continue
                    q = p["job2"]
                    full = p["full_model"]
                    full.adoptTable(jobs[q], C.c_void_p(handles[q]))
                    self.tables_rewritten += max(0, int(jobs[q].passes) - 1)
                    self.tables_patched += int(jobs[q].patches)
                    steps = full.adoptSearches([handles[qc] for qc, _ in p["cands"]])
                    merged = steps.lastSteps().sortByScoreAndEveness()          # mergeCandidates (783-793)
                    merged.print()
                    results = StepList(p["typ_e"].result, steps, [merged])
                    p["results"], p["final"] = results, merged
            finally:
                destroy(handles)
        # ---- calls, in the order of the copy-number table
        predict_alleles, warning_genes = [], []
        self._result = {}
        for k, ((gene, cn), view) in enumerate(zip(todo, views)):
            p = plan.get(k)
            pure_gene = gene.split("*")[0]
            if p is None or p.get("per_gene"):
                alleles, reads_num = self.typingPerGene(gene, cn)       # not in the index / no reads: the per-gene path
            elif p.get("fallback"):
                p["full_model"]._model._launchLog()                     # its table was left to a call that never came
                res = p["full_model"].typing(cn)
                # the reference keeps the exon-first object's (failed) result here (kir_typing.py:126): no rows of this
                # gene in .possible.tsv, the calls from the full model
                self._result[gene] = p["typ_e"].result
                alleles = [x if x != "fail" else f"{pure_gene}*" for x in res.selectBest()]
                reads_num = p["typ_e"].getReadsNum()
            else:
                self._result[gene] = p["results"]
                alleles = [x if x != "fail" else f"{pure_gene}*" for x in p["final"].selectBest()]
                reads_num = p["typ_e"].getReadsNum()
            predict_alleles.extend(alleles)
            if reads_num < min_reads_num:
                warning_genes.append(gene)
        self._result = {gene: self._result[gene] for gene, _ in todo if gene in self._result}
        return predict_alleles, warning_genes

    def _wholeSample(self) -> bool:
        import os
        from .engine import searchMode
        return not self._exon_first and not self._exon_only and self._variant_correction and searchMode() in ("bound", "exact")

    def _typingWholeSample(self, gene_cn: dict[str, int], min_reads_num: int) -> tuple[list[str], list[str]]:
        import ctypes as C
        import os
        from . import _lib
        from ._lib import check, lib
        import time
        from .utils import traceOn
        trace = traceOn("bench")     # host timeline on stderr (tools/host_timeline.py)
        t_in = time.perf_counter()
        tab, logs = self._context()
        # the sample-wide preamble (error correction, empty reads, tallies: small kernels and four waits) on the lane's
        # high-priority stream: next to another sample's search it took 8 ms instead of 1.5 on a stream like any other
        prep = tab.prepared(tab.dev.urgent(), self._multiple)
        if prep is None:                 # not a gk_tabulate tabulation (host lists / compact files)
            return super().typing(gene_cn, min_reads_num)
        t_prep = time.perf_counter()
        vflag, cnt, rows_all, off = prep[:4]
        todo = [(gene, int(cn)) for gene, cn in gene_cn.items() if cn]
        entries = []                     # (gene, cn, typ or None, job, homo)
        views = [_GeneView(self._data, gene, self._multiple, tab=tab) for gene, _ in todo]
        verdicts = self._zygosityVerdicts(tab, prep, [(v.g, cn) for v, (_, cn) in zip(views, todo)])
        for (gene, cn), view in zip(todo, views):
            if view.g is None or not view.alleles:
                entries.append((gene, cn, None, None, False))
                continue
            a, b = int(off[view.g]), int(off[view.g + 1])
            rows = _lib_slice(rows_all, a, b - a, tab.dev)
            prepared = (rows, b - a, vflag, cnt, (view.g, view.vbeg, view.vbeg + view.n_span),
                        type(tab).survivingOfGene(prep, view.g))
            typ = AlleleTyping(ReadSet(tab, rows, b - a, vflag), view.variants,
                               force_homo=False if isHetrozygous(gene) else None, top_n=self._top_n,
                               variant_correction=True, logs=logs, _vbeg=view.vbeg, _n_span=view.n_span, _mask=view.mask,
                               _alleles=view.alleles, _novel=view.novel, _prepared=prepared, _defer_launch=True)
            if b - a == 0:
                entries.append((gene, cn, typ, None, False))
                continue
            job, homo = typ.geneJob(cn, verdicts.get(view.g) if cn > 1 else False)
            entries.append((gene, cn, typ, job, homo))
        live = [e for e in entries if e[3] is not None]
        if live:
            jobs = (_lib.GeneJob * len(live))(*[e[3] for e in live])
            handles = (C.c_void_p * len(live))()
            more = (C.c_void_p * 1)()
            if trace:
                import sys
                import threading
                print(f"[trace] pre {threading.get_native_id()} {t_in:.6f} {t_prep:.6f} {time.perf_counter():.6f}", file=sys.stderr, flush=True)
            with _searchSlot():
                check(lib().gk_sample_search(tab.dev.ctx, more, 0, tab.handle, vflag.ptr, logs.handle, jobs, len(live),
                                             _lib.NUMPY_ARGSORT, _lib.NUMPY_LOG10, handles))
            self.tables_rewritten = sum(max(0, int(jobs[k].passes) - 1) for k in range(len(live)))
            self.tables_patched = sum(int(jobs[k].patches) for k in range(len(live)))
            try:
                for k, (gene, cn, typ, _, homo) in enumerate(live):
                    typ.adoptJob(jobs[k], C.c_void_p(handles[k]), cn, homo)
            finally:
                for h in handles:
                    if h:
                        lib().gk_search_destroy(C.c_void_p(h))
        predict_alleles, warning_genes = [], []
        self._result = {}
        for gene, cn, typ, job, _ in entries:
            pure_gene = gene.split("*")[0]
            if typ is None:
                self._result[gene] = []
                alleles, reads_num = [f"{pure_gene}*"] * cn, 0
            else:
                res = typ.result[-1] if job is not None else typ.typing(cn)     # no rows: the reference's empty results
                self._result[gene] = typ.result
                alleles = [x if x != "fail" else f"{pure_gene}*" for x in res.selectBest()]
                reads_num = typ.getReadsNum()
            predict_alleles.extend(alleles)
            if reads_num < min_reads_num:
                warning_genes.append(gene)
        return predict_alleles, warning_genes

    @staticmethod
    def _zygosityVerdicts(tab, prep, wanted: list[tuple[int | None, int]]) -> dict[int, bool]:
        """isHomozygous (typing_mulit_allele.py:807-857) of every listed (backbone ordinal, cn) in ONE native call on the
        sample's grouped tallies (``gk_site_verdict_genes``); a gene is asked once per sample."""
        import ctypes as C
        from ._lib import check, lib
        o, p, q, bounds = prep[4]
        tables = tab.labelTables()
        if tables is None:
            return {}
        keys_all, ins_code = tables
        cn = np.ones(len(bounds) - 1, dtype=np.int32)
        for g, c in wanted:
            if g is not None:
                cn[g] = c
        out = np.zeros(len(cn), dtype=np.int32)
        check(lib().gk_site_verdict_genes(keys_all.ctypes.data, len(keys_all), ins_code.ctypes.data, len(ins_code),
                                          o.ctypes.data, p.ctypes.data, q.ctypes.data, bounds.ctypes.data, len(cn),
                                          cn.ctypes.data, out.ctypes.data))
        return {g: bool(out[g]) for g, c in wanted if g is not None and c > 1}

    def typingPerGene(self, gene: str, cn: int) -> tuple[list[str], int]:
        logger.debug(f"[Allele] {gene=} {cn=}")
        force_homo = False if isHetrozygous(gene) else None
        tab, logs = self._context()
        view = _GeneView(self._data, gene, self._multiple, tab=tab)
        pure_gene = gene.split("*")[0]
        if view.g is None or not view.alleles:
            # gene absent from the sample's variants: the reference yields "fail" calls (or crashes
            # in createHomoResult for cn >= 2 with automatic zygosity; soft-fail here)
            self._result[gene] = []
            return [f"{pure_gene}*"] * cn, 0
        if not self._exon_first and not self._exon_only:
            prep = tab.prepared(tab.dev, self._multiple) if self._variant_correction else None
            if prep is not None:
                # error correction and empty-read removal were done for every gene of the sample in one go
                vflag, cnt, rows_all, off = prep[:4]
                a, b = int(off[view.g]), int(off[view.g + 1])
                rows = _lib_slice(rows_all, a, b - a, tab.dev)
                reads = ReadSet(tab, rows, b - a, vflag)
                prepared = (rows, b - a, vflag, cnt, (view.g, view.vbeg, view.vbeg + view.n_span),
                            type(tab).survivingOfGene(prep, view.g))
            else:
                reads, prepared = ReadSet(tab, view.rows, view.n_rows), None
            typ: AlleleTyping = AlleleTyping(
                reads, view.variants, force_homo=force_homo, top_n=self._top_n,
                variant_correction=self._variant_correction, logs=logs, _vbeg=view.vbeg,
                _n_span=view.n_span, _mask=view.mask, _alleles=view.alleles, _novel=view.novel, _defer_log=True,
                _prepared=prepared)
        else:
            reads = ReadSet(tab, view.rows, view.n_rows)
            typ = AlleleTypingExonFirst(
                reads, view.variants, force_homo=force_homo, top_n=self._top_n, exon_only=self._exon_only,
                candidate_set_threshold=self._exon_candidate_threshold, logs=logs, _vbeg=view.vbeg,
                _n_span=view.n_span, _mask=view.mask, _alleles=view.alleles, _exon_flags=view.exonFlags(),
                _novel=view.novel, _group_cache=view.groupCache())
        res = typ.typing(cn)
        self._result[gene] = typ.result
        alleles = [a if a != "fail" else f"{pure_gene}*" for a in res.selectBest()]
        return alleles, typ.getReadsNum()

    def getAllPossibleTyping(self) -> list[dict[Any, Any]]:
        rows = []
        for gene, result in self._result.items():
            if not result:
                continue
            for rank, (value, alleles) in enumerate(result[-1].selectAllPossible(.9)):
                row = {"gene": gene, "rank": rank, "value": value}
                for i, allele in enumerate(alleles):
                    row[str(i + 1)] = allele
                rows.append(row)
        return rows


class TypingWithReport(_OnLane):
    """Abundance typing by the HISAT-genotype EM (153-204)."""

    def __init__(self, filename_variant_json, device: Device | None = None):
        super().__init__()
        self._data = _sample(filename_variant_json, device)
        self.em_info: dict[str, dict] = {}      # per gene: iterations used, distinct candidate sets

    def typing(self, gene_cn: dict[str, int], min_reads_num: int = 100) -> tuple[list[str], list[str]]:
        """All genes of the sample through ``gk_sample_em``: candidate sets, distinct sets and the SQUAREM loops of every
        gene in ONE library call on one host thread and one stream (a workgroup per gene solves its EM), instead of a
        thread and a stream per gene with three waits each.  A gene with more than 2^18 distinct candidate sets sends the
        sample to the per-gene calls (``typingPerGene``: they size for it).  Same reports either way (kir_typing.py:163-195)."""
        import ctypes as C
        import os
        from . import _lib
        from ._lib import lib
        tab, _ = self._context()
        todo = [(gene, int(cn)) for gene, cn in gene_cn.items() if cn]
        views = [_GeneView(self._data, gene, multiple=False, tab=tab) for gene, _ in todo]
        live = [k for k, v in enumerate(views) if v.g is not None and v.alleles and v.n_rows]
        reports: dict[int, list[Hisat2AlleleResult]] = {}
        if live:
            jobs = (_lib.EmJob * len(live))()
            for q, k in enumerate(live):
                v, t = views[k], self._data.index.tables[views[k].g]
                jobs[q] = _lib.EmJob(d_rows=v.rows.ptr, n_rows=v.n_rows, d_mask=v.mask.ptr, vbeg=v.vbeg,
                                     vend=v.vbeg + v.n_span, words=t.words, n_allele=len(v.alleles), n_distinct=0, iterations=0)
            total = sum(len(views[k].alleles) for k in live)
            prob, count = np.zeros(total, dtype=np.float64), np.zeros(total, dtype=np.int64)
            rc = lib().gk_sample_em(tab.dev.ctx, tab.handle, jobs, len(live), 300, 0.0001, prob.ctypes.data, count.ctypes.data)
            if rc == -5:                    # GK_ERR_CAPACITY: a gene with a flood of distinct sets -- the per-gene calls size for it
                return super().typing(gene_cn, min_reads_num)
            _lib.check(rc)
            if tab.dev.call_log is not None:     # launch geometries for the roofline accounting (roofmodel.emSetsLaunch)
                n_rows = sum(views[k].n_rows for k in live)
                set_words = sum(views[k].n_rows * int(jobs[q].words) for q, k in enumerate(live))
                tab.dev.call_log.append(("em_sets_groups", n_rows, tab.n_ids / max(tab.n_valid, 1) * n_rows, set_words))
                tab.dev.call_log.append(("em_sets_verify", n_rows, 0, set_words))
            at = 0
            for q, k in enumerate(live):
                names = views[k].alleles
                p, c = prob[at:at + len(names)], count[at:at + len(names)]
                at += len(names)
                reports[k] = [Hisat2AlleleResult(allele=names[a], count=int(c[a]), prob=float(p[a])) for a in np.nonzero(c)[0]]
                self.em_info[todo[k][0]] = {"iterations": int(jobs[q].iterations), "distinct_sets": int(jobs[q].n_distinct)}
        predict_alleles, warning_genes = [], []
        self._result = {}
        for k, (gene, cn) in enumerate(todo):
            alleles, reads_num = self._callsOfReport(gene, cn, reports.get(k, []), views[k].n_rows if views[k].g is not None else 0)
            predict_alleles.extend(alleles)
            if reads_num < min_reads_num:
                warning_genes.append(gene)
        return predict_alleles, warning_genes

    def typingPerGene(self, gene: str, cn: int) -> tuple[list[str], int]:
        tab, _ = self._context()
        view = _GeneView(self._data, gene, multiple=False, tab=tab)
        report: list[Hisat2AlleleResult] = []
        if view.g is not None and view.alleles and view.n_rows:
            t = self._data.index.tables[view.g]
            info: dict = {}
            report = hisat2TypingPerGene(tab, view.rows, view.n_rows, view.vbeg, view.vbeg + view.n_span,
                                         view.mask, t.words, view.alleles, info=info)
            self.em_info[gene] = info
        return self._callsOfReport(gene, cn, report, view.n_rows)

    def _callsOfReport(self, gene: str, cn: int, report: list, n_rows: int) -> tuple[list[str], int]:
        """Abundances -> calls (kir_typing.py:181-192): the copy numbers go to the alleles in descending abundance."""
        pure_gene = gene.split("*")[0]
        # descending abundance; ties by allele name (the reference leaves them to set order)
        report.sort(key=lambda r: (-r.prob, r.allele))
        if not report:
            self._result[gene] = report
            return [f"{pure_gene}*"] * cn, n_rows   # the reference raises AxisError here
        est_prob = 1 / cn
        called = []
        for rec in report:
            pred = max(1, round(rec.prob / est_prob))
            called.extend([rec.allele] * min(cn, pred))
            rec.cn = pred
            cn -= pred
            if cn <= 0:
                break
        self._result[gene] = report
        return called, n_rows

    def getAllPossibleTyping(self) -> list[dict[Any, Any]]:
        raise NotImplementedError


def selectKirTypingModel(method: str, filename_variant_json, **kwargs: Any) -> Typing:
    """Select and initialise the typing strategy (207-228)."""
    if method in ("full", "pv"):
        return TypingWithPosNegAllele(filename_variant_json, **kwargs)
    if method.startswith("pv_exonfirst"):
        method = method[len("pv_"):]
    if method.startswith("exonfirst"):
        fields = method.split("_")
        threshold = 0.0
        if len(fields) == 2:
            threshold = float(method[len("exonfirst_"):])
        return TypingWithPosNegAllele(filename_variant_json, exon_first=True,
                                      exon_candidate_threshold=threshold, **kwargs)
    if method in ("em", "report"):
        kwargs.pop("top_n", None)
        kwargs.pop("variant_correction", None)
        return TypingWithReport(filename_variant_json, **kwargs)
    raise NotImplementedError

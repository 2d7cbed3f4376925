"""
Device-resident state of the typing path and thin wrappers over the C ABI.

Layout in HBM (one sample at a time per GPU; sizes for 1 M pairs, ~2 k alleles):

* index keys ``u64[V]`` + per-gene allele bit rows ``u32[V_g][A_g/32]`` (< 1 MB, resident for the run)
* mate records ``64 B x 2 x pairs``  (128 MB)
* tabulation CSR: ``u32`` offsets ``[4 x valid + 1]`` and ``u32`` variant ordinals (~50 per pair)
* per gene: row list ``i32[R]``, variant flags ``u8[V + novel]``,
  probabilities / log-probabilities column-major ``f64[A][R]`` (so every reduction over reads
  is a coalesced stream and matches numpy's contiguous-axis summation)

Reference call sites are cited in ``include/graphkir_hip.h``.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _lib
from ._lib import Device, DeviceBuffer, DeviceSlice, TabInfo, check, lib
from .index import GkIndex, KEY_POS_SHIFT, KEY_TYP_SHIFT, KEY_VAL_MASK
from .msa2hisat import Variant

TYPE_OF_RANK = {0: "insertion", 1: "single", 2: "deletion"}


def _i32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.int32)


def searchMode() -> str:
    """``bound`` (default): a search step first bounds every candidate set with integer mismatch totals and forms
    the exact float64 sums only for the sets that can reach the cut; ``exact``: float64 sums for every candidate
    (GK_SEARCH=exact).  Both give the reference's bits; ``bound`` falls back to ``exact`` for a step whose order
    would depend on numpy's tie order among sets it did not sum."""
    import os
    mode = os.environ.get("GK_SEARCH", "bound")
    if mode not in ("bound", "exact"):
        raise ValueError(f"GK_SEARCH={mode!r}: expected 'bound' or 'exact'")
    return mode


class DeviceIndex:
    """Index tables in HBM (``gk_index``) + allele bit rows per gene."""

    def __init__(self, dev: Device, index: GkIndex):
        self.dev, self.host = dev, index
        h = C.c_void_p()
        key = np.ascontiguousarray(index.key, dtype=np.uint64)
        vbeg = _i32(index.gene_vbeg)
        check(lib().gk_index_create(dev.ctx, key.ctypes.data, len(key), vbeg.ctypes.data, len(index.genes),
                                    C.byref(h)))
        self.handle = h
        self.masks = [dev.put(t.mask) for t in index.tables]

    def close(self) -> None:
        if self.handle:
            lib().gk_index_destroy(self.handle)
            self.handle = None


class Tabulation:
    """Result of ``gk_tabulate`` for one sample (replaces the ``.variant.json`` hand-off)."""

    def __init__(self, dindex: DeviceIndex, mates, novel_base: int = 0, dev: Device | None = None,
                 correction: tuple[np.ndarray, np.ndarray] | None = None,
                 spill: tuple[np.ndarray, np.ndarray] | None = None):
        """``dev``: context (stream) that runs the tabulation; defaults to the index's own.
        ``correction``: (``pileup.correctionTable`` uint8 [positions][5], first position of every backbone
        int64 [genes + 1]) -- the pileup error correction of mismatches, off when None.
        ``spill``: (records of the pairs that do not fit ``gk_mate``, in the wide format, 2 per pair; the pairs they
        stand for, ascending) as the packer hands them out -- None when every pair fits."""
        self.dev, self.dindex = dev or dindex.dev, dindex
        self._spill, self._correction = spill, correction      # host arrays, kept for a second tabulation (ParkedRecords)
        if isinstance(mates, np.ndarray):
            assert mates.dtype == _lib.MATE_DTYPE
            self.mates = self.dev.put(mates)
        elif hasattr(mates, "toDevice"):      # packed.CompactMates: copied compact, expanded in HBM
            self.mates = mates.toDevice(self.dev)
        else:
            self.mates = mates
        self.n_pairs = self.mates.size // 2
        h = C.c_void_p()
        d_table = d_pos0 = None
        if correction is not None:
            table, pos0 = correction
            assert table.dtype == np.uint8 and table.shape == (int(pos0[-1]), 5) and len(pos0) == len(dindex.host.genes) + 1
            d_table = self.dev.put(np.ascontiguousarray(table).reshape(-1) if table.size else np.zeros(1, np.uint8))
            d_pos0 = self.dev.put(np.ascontiguousarray(pos0, dtype=np.int64))
        if spill is not None and len(spill[1]):
            wide = np.ascontiguousarray(spill[0], dtype=_lib.MATE_WIDE_DTYPE)
            which = np.ascontiguousarray(spill[1], dtype=np.int64)
            assert len(wide) == 2 * len(which)
            check(lib().gk_tabulate_spilled(self.dev.ctx, dindex.handle, self.mates.ptr, self.n_pairs,
                                            d_table.ptr if d_table else 0, d_pos0.ptr if d_pos0 else 0,
                                            wide.ctypes.data, which.ctypes.data, len(which), C.byref(h)))
        elif correction is None:
            check(lib().gk_tabulate(self.dev.ctx, dindex.handle, self.mates.ptr, self.n_pairs, C.byref(h)))
        else:
            check(lib().gk_tabulate_corrected(self.dev.ctx, dindex.handle, self.mates.ptr, self.n_pairs,
                                              d_table.ptr, d_pos0.ptr, C.byref(h)))
        self.handle = h
        info = TabInfo()
        check(lib().gk_tab_get_info(h, C.byref(info)))
        self.info = info
        self.n_valid, self.n_ids, self.n_novel = int(info.n_valid), int(info.n_ids), int(info.n_novel)
        self.novel_base = novel_base
        if info.err_flags & 1:
            raise AssertionError("variant window has left > right (graphkir/hisat2.py:744)")
        self._novel_keys = None
        if self.dev.call_log is not None:
            self.dev.call_log.append(("tab_count", self.n_pairs, self.n_valid, self.n_ids))

    @property
    def n_var_total(self) -> int:
        override = getattr(self, "_id_names", None)
        if override is not None:
            return len(override)
        return self.dindex.host.n_variant + self.n_novel

    def close(self) -> None:
        if self.handle and not getattr(self, "_borrowed", False):
            for prep in (getattr(self, "_prepared", None) or {}).values():
                for b in prep[:3]:
                    b.free()
            self._prepared = {}
            lib().gk_tab_destroy(self.handle)
        self.handle = None

    def prepared(self, dev: Device, multiple: bool = False, exon: bool = False):
        """Error correction + removal of empty reads of EVERY gene at once (``gk_sample_prepare``), computed by
        the first caller and shared by all gene threads: (drop flags, tallies, rows grouped by gene, bounds).
        ``exon``: the same for the EXON model of every gene (``gk_sample_prepare_exon``: ids outside the exons dropped
        from every list, the correction applied twice, typing_mulit_allele.py:640-664).
        None for tabulations that were not made by ``gk_tabulate`` (no index handle / novel keys on the device)."""
        if self.dindex is None or not self.info.d_pair_src:      # host lists / compact files: no index handle in the library
            return None
        import threading
        root = getattr(self, "_root", self)
        lock = root.__dict__.setdefault("_prep_lock", threading.Lock())
        store = root.__dict__.setdefault("_prepared", {})
        with lock:
            prep = store.get((bool(multiple), True) if exon else bool(multiple))
            if prep is None and exon:
                nv = max(self.n_var_total, 1)
                host = self.dindex.host
                flags = np.full(nv, 3, dtype=np.uint8)           # novel variants are never in an exon
                flags[:host.n_variant][host.in_exon.astype(bool)] = 0
                vflag = dev.put(flags)
                cnt = dev.alloc(2 * nv, np.uint32)
                rows = dev.alloc(max(self.n_valid, 1), np.int32)
                off = np.zeros(len(host.genes) + 1, dtype=np.int64)
                o, p, q = (np.empty(nv, dtype=t) for t in (np.int32, np.uint32, np.uint32))
                n_surv = C.c_int64()
                check(lib().gk_sample_prepare_exon(dev.ctx, self.handle, int(multiple), vflag.ptr, cnt.ptr, rows.ptr,
                                                   off.ctypes.data, nv, o.ctypes.data, p.ctypes.data, q.ctypes.data,
                                                   C.byref(n_surv)))
                o, p, q = o[:n_surv.value], p[:n_surv.value], q[:n_surv.value]
                gene_of = np.searchsorted(host.gene_vbeg, o, side="right") - 1
                order = np.argsort(gene_of, kind="stable")
                bounds = np.searchsorted(gene_of[order], np.arange(len(host.genes) + 1)).astype(np.int64)
                grouped = (np.ascontiguousarray(o[order], dtype=np.int32), np.ascontiguousarray(p[order], dtype=np.uint32),
                           np.ascontiguousarray(q[order], dtype=np.uint32), bounds)
                prep = store[(bool(multiple), True)] = (vflag, cnt, rows, off, grouped)
            if prep is None:
                nv = max(self.n_var_total, 1)
                vflag = dev.alloc(nv, np.uint8)
                cnt = dev.alloc(2 * nv, np.uint32)
                rows = dev.alloc(max(self.n_valid, 1), np.int32)
                off = np.zeros(len(self.dindex.host.genes) + 1, dtype=np.int64)
                # error correction, empty reads, the surviving tallies of every gene (isHomozygous reads them per gene) and
                # the sample's novel keys in ONE library call with two waits (gk_sample_prepare_all; six when the three
                # were separate calls -- each wait sits behind the long kernels of the samples typed next to this one)
                o, p, q = (np.empty(nv, dtype=t) for t in (np.int32, np.uint32, np.uint32))
                n_surv = C.c_int64()
                want_novel = self._novel_keys is None and self.n_novel > 0
                novel_keys = np.empty(self.n_novel, dtype=np.uint64) if want_novel else None
                check(lib().gk_sample_prepare_all(dev.ctx, self.handle, int(multiple), vflag.ptr, cnt.ptr, rows.ptr,
                                                  off.ctypes.data, nv, o.ctypes.data, p.ctypes.data, q.ctypes.data,
                                                  C.byref(n_surv), novel_keys.ctypes.data if want_novel else None))
                o, p, q = o[:n_surv.value], p[:n_surv.value], q[:n_surv.value]
                if want_novel:
                    self._novel_keys = root._novel_keys = novel_keys
                n_index = self.dindex.host.n_variant
                gene_of = np.searchsorted(self.dindex.host.gene_vbeg, o, side="right") - 1
                if len(o) and int(o[-1]) >= n_index:
                    is_novel = o >= n_index
                    gene_of[is_novel] = (self.novelKeys()[o[is_novel] - n_index] >> np.uint64(56)).astype(gene_of.dtype)
                # grouped by gene once (ordinals ascending inside a gene, as survivingCounts lists them): a gene's tallies
                # are a slice, and the zygosity verdicts of all genes one native call (gk_site_verdict_genes)
                order = np.argsort(gene_of, kind="stable")
                bounds = np.searchsorted(gene_of[order], np.arange(len(self.dindex.host.genes) + 1)).astype(np.int64)
                grouped = (np.ascontiguousarray(o[order], dtype=np.int32), np.ascontiguousarray(p[order], dtype=np.uint32),
                           np.ascontiguousarray(q[order], dtype=np.uint32), bounds)
                prep = store[bool(multiple)] = (vflag, cnt, rows, off, grouped)
        return prep

    @staticmethod
    def survivingOfGene(prep, g: int):
        """(ordinals, positive tally, negative tally) of gene ``g`` out of ``prepared()``'s sample-wide list: what
        ``survivingCounts(cnt, vflag, gene=(g, vbeg, vend))`` returns, without a device call."""
        o, p, q, bounds = prep[4]
        a, b = int(bounds[g]), int(bounds[g + 1])
        return o[a:b], p[a:b], q[a:b]

    def on(self, dev: Device) -> "Tabulation":
        """The same tabulation driven from another context (stream) of the same GPU.

        The CSR is read-only after ``gk_tabulate`` returned (it synchronises), so per-gene work can
        run on independent streams."""
        if dev is self.dev:
            return self
        import copy
        other = copy.copy(self)
        other.dev = dev
        other._borrowed = True
        other._root = getattr(self, "_root", self)     # shared per-sample state lives on the owning tabulation
        return other

    # ---- host views (outputs / tests)
    def offsets(self) -> np.ndarray:
        return self.dev.view(self.info.d_off, 4 * self.n_valid + 1, np.uint32)

    def ids(self) -> np.ndarray:
        return self.dev.view(self.info.d_ids, self.n_ids, np.uint32)

    def pairSrc(self) -> np.ndarray:
        return self.dev.view(self.info.d_pair_src, self.n_valid, np.int32)

    def pairGene(self) -> np.ndarray:
        return self.dev.view(self.info.d_pair_gene, self.n_valid, np.uint8)

    def pairNH(self) -> np.ndarray:
        return self.dev.view(self.info.d_pair_nh, self.n_valid, np.uint8)

    def novelKeys(self) -> np.ndarray:
        if self._novel_keys is None:
            self._novel_keys = self.dev.view(self.info.d_novel_key, self.n_novel, np.uint64)
        return self._novel_keys

    def novelVariants(self, ins_strings: list[str]) -> list[Variant]:
        """Novel variants in first-appearance order, ids ``nv{novel_base + rank}`` (hisat2.py:597-602)."""
        genes = self.dindex.host.genes
        out = []
        for rank, k in enumerate(self.novelKeys().tolist()):
            ref = genes[k >> 56]
            pos = (k >> KEY_POS_SHIFT) & 0xFFFFFF
            typ = TYPE_OF_RANK[(k >> KEY_TYP_SHIFT) & 3]
            val = k & KEY_VAL_MASK
            if typ == "single":
                v = Variant(pos=pos, typ=typ, ref=ref, val=chr(val), length=1)
            elif typ == "deletion":
                v = Variant(pos=pos, typ=typ, ref=ref, val=int(val), length=int(val))
            else:
                s = ins_strings[val]
                v = Variant(pos=pos, typ=typ, ref=ref, val=s, length=len(s))
            v.id = f"nv{self.novel_base + rank}"
            out.append(v)
        return out

    def idNames(self) -> list[str]:
        """Ordinal -> variant id string for index + novel variants."""
        override = getattr(self, "_id_names", None)
        if override is not None:
            return override
        names = [str(v.id) for v in self.dindex.host.variants]
        names += [f"nv{self.novel_base + r}" for r in range(self.n_novel)]
        self._id_names = names
        return names

    def labelTables(self):
        """(keys of every ordinal -- index variants, then this sample's novel ones --, label code of every inserted
        string) for ``gk_site_verdict_tallies``; None for list-built tabulations (no packed keys)."""
        if getattr(self, "_variant_src", None) is not None:
            return None
        root = getattr(self, "_root", self)
        tables = root.__dict__.get("_label_tables")
        if tables is None:
            keys_all = np.ascontiguousarray(np.concatenate([self.dindex.host.key, self.novelKeys()]), dtype=np.uint64)
            strings = getattr(self, "ins_strings", None) or self.dindex.host.ins_strings
            ins_code = np.array([ord(s) if len(s) == 1 else 256 + i for i, s in enumerate(strings)] or [0], dtype=np.int64)
            tables = root.__dict__["_label_tables"] = (keys_all, ins_code)
        return tables

    def labelCodes(self, ordinals):
        """Vectorised (pos, label code, is_deletion) of ordinals; equal codes <=> equal ``str(val)``.

        None for list-built tabulations (no packed keys); callers then use ``describe``.
        """
        if getattr(self, "_variant_src", None) is not None:
            return None
        keys_all = getattr(self, "_keys_all", None)
        if keys_all is None:
            keys_all = self._keys_all = np.concatenate([self.dindex.host.key, self.novelKeys()])
        k = keys_all[np.asarray(ordinals, dtype=np.int64)].astype(np.int64)
        typ = (k >> KEY_TYP_SHIFT) & 3
        val = k & KEY_VAL_MASK
        ins_code = getattr(self, "_ins_code", None)
        if ins_code is None:
            strings = getattr(self, "ins_strings", None) or self.dindex.host.ins_strings
            # a one-base insertion prints like a SNP of that base (str(val) is the key in the reference)
            ins_code = self._ins_code = np.array(
                [ord(s) if len(s) == 1 else 256 + i for i, s in enumerate(strings)] or [0], dtype=np.int64)
        code = np.where(typ == 0, ins_code[np.minimum(val, len(ins_code) - 1)], val)
        return (k >> KEY_POS_SHIFT) & 0xFFFFFF, code, typ == 2

    def describe(self, ordinals) -> list[tuple[int, str, str]]:
        """(pos, typ, str(val)) of the given ordinals without building Variant objects."""
        src = getattr(self, "_variant_src", None)
        if src is not None:   # list-built tabulation: ordinals index the caller's variant list
            return [(src[o].pos, src[o].typ, str(src[o].val)) for o in ordinals]
        keys_all = getattr(self, "_keys_all", None)
        if keys_all is None:
            keys_all = self._keys_all = np.concatenate([self.dindex.host.key, self.novelKeys()])
        strings = getattr(self, "ins_strings", None) or self.dindex.host.ins_strings
        out = []
        for k in keys_all[np.asarray(ordinals, dtype=np.int64)].tolist():
            typ = TYPE_OF_RANK[(k >> KEY_TYP_SHIFT) & 3]
            val = k & KEY_VAL_MASK
            label = chr(val) if typ == "single" else (str(val) if typ == "deletion" else strings[val])
            out.append(((k >> KEY_POS_SHIFT) & 0xFFFFFF, typ, label))
        return out

    # ---- selections
    def selectGene(self, gene: int, multiple: bool = False) -> tuple[DeviceBuffer, int]:
        rows = self.dev.alloc(max(self.n_valid, 1), np.int32)
        n = C.c_int64()
        check(lib().gk_select_gene(self.dev.ctx, self.handle, gene, int(multiple), rows.ptr, C.byref(n)))
        return rows, int(n.value)

    def selectNonEmpty(self, rows: DeviceBuffer, n_rows: int, vflag: DeviceBuffer) -> tuple[DeviceBuffer, int]:
        out = self.dev.alloc(max(n_rows, 1), np.int32)
        n = C.c_int64()
        check(lib().gk_select_nonempty(self.dev.ctx, self.handle, rows.ptr, n_rows, vflag.ptr, out.ptr, C.byref(n)))
        return out, int(n.value)

    def countVariants(self, rows: DeviceBuffer, n_rows: int, vflag: DeviceBuffer, cnt: DeviceBuffer,
                      span: tuple[int, int] = (0, 0)) -> None:
        check(lib().gk_variant_count_range(self.dev.ctx, self.handle, rows.ptr, n_rows, vflag.ptr, cnt.ptr,
                                           span[0], span[1]))

    def correctVariants(self, cnt: DeviceBuffer, vflag: DeviceBuffer) -> None:
        check(lib().gk_variant_correct(self.dev.ctx, self.handle, cnt.ptr, vflag.ptr))

    def errorCorrection(self, rows: DeviceBuffer, n_rows: int, vflag: DeviceBuffer,
                        span: tuple[int, int] = (0, 0), keep: bool = False) -> DeviceBuffer | None:
        """One pass of ``AlleleTyping.errorCorrection`` on the lists as filtered by ``vflag``.

        ``span`` = index-variant ordinal range of the rows' gene (LDS-privatised counters).
        With ``keep`` the tally buffer is returned (its counts masked by the updated ``vflag`` are
        the tallies of the corrected lists)."""
        cnt = self.dev.alloc(2 * self.n_var_total, np.uint32)
        self.countVariants(rows, n_rows, vflag, cnt, span)
        self.correctVariants(cnt, vflag)
        if keep:
            return cnt
        cnt.free()
        return None

    def survivingCounts(self, cnt: DeviceBuffer, vflag: DeviceBuffer, gene: tuple[int, int, int] | None = None
                        ) -> tuple[np.ndarray, np.ndarray, np.ndarray]:
        """(ordinals, positive tally, negative tally) of variants with a surviving observation.
        ``gene`` = (backbone ordinal, vbeg, vend): only that backbone's variants (tallies shared by all genes)."""
        cap = 1 << 16
        while True:
            o = np.empty(cap, dtype=np.int32)
            p = np.empty(cap, dtype=np.uint32)
            q = np.empty(cap, dtype=np.uint32)
            n = C.c_int64()
            if gene is None:
                rc = lib().gk_variant_surviving(self.dev.ctx, self.handle, cnt.ptr, vflag.ptr, cap, o.ctypes.data,
                                                p.ctypes.data, q.ctypes.data, C.byref(n))
            else:
                rc = lib().gk_variant_surviving_gene(self.dev.ctx, self.handle, cnt.ptr, vflag.ptr, gene[0], gene[1],
                                                     gene[2], cap, o.ctypes.data, p.ctypes.data, q.ctypes.data,
                                                     C.byref(n))
            if rc == -5 and cap < self.n_var_total:
                cap = min(cap * 8, max(self.n_var_total, 1))
                continue
            check(rc)
            return o[:n.value], p[:n.value], q[:n.value]


class LogTable:
    """``numpy.log10`` applied through a device value table (see ``csrc/gk_lut.hip``).

    One table serves every context (stream) of a GPU: kernels of any stream look values up and
    insert unknown ones; ``resolve`` (serialised by a lock, on the table's own context) evaluates
    ``numpy.log10`` for the new ones."""

    _EMPTY = np.uint64(0x7FF8DEADBEEF0001)   # kLutEmptyKey: a claimed list entry whose key is not stored yet

    def __init__(self, dev: Device, log2_capacity: int = 20, private_context: bool = False):
        self.dev = Device(dev.ordinal) if private_context else dev
        h = C.c_void_p()
        check(lib().gk_lut_create(self.dev.ctx, log2_capacity, C.byref(h)))
        self.handle = h
        self.n_host_evals = 0
        self.n_known = 0          # values with a defined log10 (== the library's count)
        self.n_undefined = 0      # entries seen by the last resolve() that could not be defined yet
        import threading
        self._lock = threading.Lock()

    def collect(self, buf: DeviceBuffer, n: int) -> None:
        check(lib().gk_lut_collect(self.handle, buf.ptr, n))

    def resolve(self) -> int:
        """Evaluate numpy.log10 for values first seen since the last call; returns how many.

        The caller's own kernels must have completed (``Device.sync``).  Kernels of other streams
        may still be inserting: the library waits for the device when the table grew, and an
        entry that is claimed but not stored yet (possible only for kernels launched meanwhile) ends
        the batch -- whoever launched that kernel resolves it.  One resolver at a time, serialised inside the
        library (``gk_lut_resolve``): host threads and ``gk_sample_search`` share the table."""
        with self._lock:
            n_new, known, undefined = C.c_int32(), C.c_int32(), C.c_int32()
            check(lib().gk_lut_resolve(self.handle, _lib.NUMPY_LOG10, C.byref(n_new), C.byref(known), C.byref(undefined)))
            self.n_host_evals += n_new.value
            self.n_known = known.value
            self.n_undefined = undefined.value      # claimed by kernels still running elsewhere
            return n_new.value

    def known(self) -> int:
        """Values with a defined log10 right now (another thread or ``gk_sample_search`` may have resolved some)."""
        n = C.c_int32()
        check(lib().gk_lut_known(self.handle, C.byref(n)))
        self.n_known = max(self.n_known, n.value)
        return n.value

    def apply(self, src: DeviceBuffer, dst: DeviceBuffer, n: int) -> None:
        check(lib().gk_lut_apply(self.handle, src.ptr, dst.ptr, n))

    def close(self) -> None:
        if self.handle:
            lib().gk_lut_destroy(self.handle)
            self.handle = None


class DeviceModel:
    """Read x allele log-likelihood table of one gene in HBM + its reductions."""

    def __init__(self, tab: Tabulation, rows: DeviceBuffer, n_rows: int, vflag: DeviceBuffer,
                 vbeg: int, vend: int, mask: DeviceBuffer, words: int, n_allele: int, logs: LogTable,
                 want_miss: bool = False, keep_empty: bool = False, launch: bool = True, indexed: bool = False):
        """``keep_empty``: rows without any kept variant are part of the model and score 0.999 for every
        allele (``no_empty=False``, typing_mulit_allele.py:372-374).  ``launch=False``: the tables are only
        allocated; ``gk_sample_search`` fills them together with those of the sample's other genes.
        ``indexed`` (with ``launch=False`` and the integer bound): the table is kept as uint16 indices into the value
        table (2 bytes per entry, ``lidx``); the float64 form ``L`` is made from it only when somebody reads it."""
        self.tab, self.dev = tab, tab.dev
        self._keep_empty = int(bool(keep_empty))
        self.rows, self.n_rows, self.n_allele = rows, n_rows, n_allele
        self.vflag = vflag
        self._L = self.miss = self.nvar = None
        self.lidx = None               # uint16 [n_allele][ldm]: the index form (valid once gk_sample_search said so)
        self._indexed = False
        self._probs = None
        self._logs = logs
        self._geom = (vbeg, vend, mask, words)
        self._want_miss = want_miss
        self._known_at_launch = -1
        # integer bound of the search (csrc/gk_bound.hip): u8 mismatch counts next to the log-likelihoods
        self.miss8 = self.msum = self._bound_flags = None
        self.ldm = 0
        self._bound_ok: bool | None = None
        if n_rows == 0 or n_allele == 0:
            return
        want_index = bool(indexed and not launch and searchMode() == "bound" and n_rows < 16_000_000)
        if not want_index:
            self._L = self.dev.alloc((n_allele, n_rows), np.float64)
        if searchMode() == "bound" and n_rows < 16_000_000:
            self.ldm = (n_rows + 63) // 64 * 64
            self.miss8 = self.dev.alloc((n_allele, self.ldm), np.uint8)
            self.msum = self.dev.alloc(n_allele, np.uint32)
            self._bound_flags = self.dev.alloc(1, np.uint32)
            if want_index:
                self.lidx = self.dev.alloc((n_allele, self.ldm), np.uint16)
        if launch:
            self._launchLog()
        if want_miss:
            self._launchProbs()

    @property
    def L(self) -> DeviceBuffer | None:
        """The float64 table [allele][read]; made from the index form on first use when that is what the model holds."""
        if self._L is None and self.lidx is not None:
            if self._indexed:
                self._L = self.dev.alloc((self.n_allele, self.n_rows), np.float64)
                check(lib().gk_expand_index(self.dev.ctx, self._logs.handle, self.lidx.ptr, self.ldm, self.n_rows,
                                            self.n_allele, self._L.ptr, self.n_rows))
            else:       # the library worked on a float64 table of its own (value table beyond 16-bit indices): write ours
                self._L = self.dev.alloc((self.n_allele, self.n_rows), np.float64)
                self._launchLog()
                self.finishLog()
        return self._L

    def _launchLog(self) -> None:
        vbeg, vend, mask, words = self._geom
        self._known_at_launch = self._logs.known()
        if self.dev.call_log is not None:    # ids of the rows: the sample's average list length (no sync for a count)
            per_row = self.tab.n_ids / max(self.tab.n_valid, 1)
            self.dev.call_log.append(("compat_kernel", self.n_rows, self.n_allele, per_row * self.n_rows, 8))
        if self.miss8 is None:
            check(lib().gk_compat_log(self.dev.ctx, self.tab.handle, self.rows.ptr, self.n_rows, self.vflag.ptr, vbeg,
                                      vend, mask.ptr, words, self.n_allele, self._keep_empty, self._logs.handle,
                                      self.L.ptr))
            return
        check(lib().gk_compat_log_miss(self.dev.ctx, self.tab.handle, self.rows.ptr, self.n_rows, self.vflag.ptr, vbeg,
                                       vend, mask.ptr, words, self.n_allele, self._keep_empty, self._logs.handle,
                                       self.L.ptr, self.miss8.ptr, self.ldm, self._bound_flags.ptr))
        check(lib().gk_miss_colsum(self.dev.ctx, self.miss8.ptr, self.ldm, self.n_allele, self.msum.ptr))
        self._bound_ok = None

    def _launchProbs(self) -> None:
        """The un-logged products (and the mismatch counts): only built when somebody asks for them."""
        vbeg, vend, mask, words = self._geom
        self._probs = self.dev.alloc((self.n_allele, self.n_rows), np.float64)
        if self._want_miss:
            self.miss = self.dev.alloc((self.n_allele, self.n_rows), np.uint8)
            self.nvar = self.dev.alloc(self.n_rows, np.uint16)
        check(lib().gk_compat(self.dev.ctx, self.tab.handle, self.rows.ptr, self.n_rows, self.vflag.ptr, vbeg, vend,
                              mask.ptr, words, self.n_allele, self._keep_empty, self._probs.ptr,
                              self.miss.ptr if self.miss else 0,
                              self.nvar.ptr if self.nvar else 0))

    @property
    def probs(self) -> DeviceBuffer | None:
        if self._probs is None and self.n_rows and self.n_allele:
            self._launchProbs()
        return self._probs

    def finishLog(self) -> None:
        """Second half of construction: if the kernel met products whose log10 the table did not
        hold yet, the host evaluates them (numpy.log10) and the table is written once more."""
        if self._L is None or self._known_at_launch < 0:
            return
        sticky = 0                              # bit 0 of the flag word survives a patch
        for _ in range(64):
            self.dev.sync()                     # this model's kernel has stored every key it claimed
            if self._bound_flags is not None:
                # the kernel itself says whether it stored NaN for a product without a log10 (bit 2 of the flag word)
                flags = int(self._bound_flags.download()[0]) | sticky
                if not flags & 4:
                    self._bound_ok = (flags & 3) == 0
                    break
                self._logs.resolve()
                if flags & 8:                   # a product that could not mark itself: the whole table again
                    sticky = 0
                    self._launchLog()
                else:                           # the entries that hold their product get their log10, nothing else moves
                    sticky = flags & 1
                    check(lib().gk_compat_patch(self.dev.ctx, self._logs.handle, self._L.ptr, self.n_rows, self.n_allele,
                                                self.miss8.ptr, self.ldm, self._bound_flags.ptr))
                    check(lib().gk_miss_colsum(self.dev.ctx, self.miss8.ptr, self.ldm, self.n_allele, self.msum.ptr))
                continue
            self._logs.resolve()
            if self._logs.n_known > self._known_at_launch:
                self._launchLog()               # some values were undefined at launch: write the table again
            elif self._logs.n_undefined == 0:
                break                           # every value this launch met had its log10 in the table
        else:
            raise _lib.GkError("log10 value table did not settle")
        self._known_at_launch = -1

    # ---- integer bound of a search step
    @property
    def boundOk(self) -> bool:
        """The mismatch table exists and no count came near the underflow range (call after ``finishLog``)."""
        if self.miss8 is None:
            return False
        if self._bound_ok is None:
            self._bound_ok = (int(self._bound_flags.download()[0]) & 3) == 0
        return self._bound_ok

    def boundStep(self, prev_ids: np.ndarray, cols: np.ndarray, first: np.ndarray, top_n: int, cap: int):
        """Candidates (t, j) = prev_ids[t] + [cols[j]] that can reach the top_n cut: those with
        M = sum_r min(miss) <= the top_n-th smallest M among ``first``.  Returns (candidates, M_T, flat indices
        ascending, their M) or None when more than ``cap`` qualify."""
        prev_ids, cols = _i32(prev_ids), _i32(cols)
        n_sets, c_prev = prev_ids.shape
        first = np.ascontiguousarray(first, dtype=np.uint8)
        assert first.size == n_sets * len(cols)
        hdr = np.zeros(4, dtype=np.uint32)
        idx = np.empty(cap, dtype=np.int32)
        mm = np.empty(cap, dtype=np.uint32)
        if self.dev.call_log is not None:
            self.dev.call_log.append(("minsum_sad", self.n_rows, n_sets, len(cols),
                                      len(np.unique(prev_ids)) if c_prev == 1 else n_sets, False))
        check(lib().gk_bound_step(self.dev.ctx, self.miss8.ptr, self.ldm, self.n_rows, self.msum.ptr,
                                  prev_ids.ctypes.data, n_sets, c_prev, cols.ctypes.data, len(cols), first.ctypes.data,
                                  top_n, cap, hdr.ctypes.data, idx.ctypes.data, mm.ctypes.data))
        n_sel = int(hdr[2])
        if n_sel > cap:
            return None
        order = np.argsort(idx[:n_sel], kind="stable")
        return int(hdr[0]), int(hdr[1]), idx[:n_sel][order].astype(np.int64), mm[:n_sel][order]

    def setsum(self, ids: np.ndarray) -> tuple[np.ndarray, np.ndarray]:
        """(value [k], fraction [k, c]) of the given sets: exact float64, numpy's summation tree."""
        ids = _i32(ids)
        value = np.empty(ids.shape[0], dtype=np.float64)
        frac = np.empty(ids.shape, dtype=np.float64)
        if self.dev.call_log is not None:
            self.dev.call_log.append(("setsum_leaves" if 2 <= ids.shape[1] <= 4 else "fraction_chunks", self.n_rows,
                                      ids.shape[0], ids.shape[1], len(np.unique(ids))))
        check(lib().gk_setsum(self.dev.ctx, self.L.ptr, self.n_rows, self.n_rows, ids.ctypes.data, ids.shape[0],
                              ids.shape[1], value.ctypes.data, frac.ctypes.data))
        return value, frac

    # ---- reductions (numpy summation tree on the device)
    def maxsum(self, prev_ids: np.ndarray | None, cols: np.ndarray) -> np.ndarray:
        cols = _i32(cols)
        if prev_ids is None or prev_ids.size == 0:
            n_sets, c_prev, ids_p = 1, 0, None
        else:
            prev_ids = _i32(prev_ids)
            n_sets, c_prev = prev_ids.shape
            ids_p = prev_ids.ctypes.data
        out = np.empty((n_sets, len(cols)), dtype=np.float64)
        if self.dev.call_log is not None:
            n_prev_cols = 0 if ids_p is None else len(np.unique(prev_ids))
            # the library runs the second allele of the search as a symmetric table (gk_search.hip: gk_maxsum)
            symmetric = bool(c_prev == 1 and n_sets == len(cols) and n_sets > 32 and
                             np.array_equal(np.sort(prev_ids.ravel()), np.sort(cols)) and len(np.unique(cols)) == n_sets)
            self.dev.call_log.append(("colsum_chunks" if ids_p is None and n_sets == 1 else "maxsum_chunks", self.n_rows,
                                      n_sets, c_prev, len(cols), n_prev_cols, symmetric))
        check(lib().gk_maxsum(self.dev.ctx, self.L.ptr, self.n_rows, self.n_rows, ids_p, n_sets, c_prev,
                              cols.ctypes.data, len(cols), out.ctypes.data))
        return out

    def pairTable(self) -> np.ndarray:
        """``sum_r max(L[r, a], L[r, b])`` for every allele pair (a, b): the second search step's scores.
        The library computes the upper triangle and mirrors it; the diagonal is the column sum."""
        every = np.arange(self.n_allele)
        return self.maxsum(every[:, None], every)

    def colsum(self, cols: np.ndarray) -> np.ndarray:
        return self.maxsum(None, cols)[0]

    def fraction(self, ids: np.ndarray) -> np.ndarray:
        ids = _i32(ids)
        out = np.empty(ids.shape, dtype=np.float64)
        if self.dev.call_log is not None:
            self.dev.call_log.append(("fraction_chunks", self.n_rows, ids.shape[0], ids.shape[1], len(np.unique(ids))))
        check(lib().gk_fraction(self.dev.ctx, self.L.ptr, self.n_rows, self.n_rows, ids.ctypes.data, ids.shape[0],
                                ids.shape[1], out.ctypes.data))
        return out

    def setmax(self, ids: np.ndarray) -> np.ndarray:
        """Host copy of allele_prob (R x T) for the given sets -- API parity only, not on the hot path."""
        ids = _i32(ids)
        buf = self.dev.alloc((ids.shape[0], self.n_rows), np.float64)
        check(lib().gk_setmax(self.dev.ctx, self.L.ptr, self.n_rows, self.n_rows, ids.ctypes.data, ids.shape[0],
                              ids.shape[1], buf.ptr))
        out = buf.download().reshape(ids.shape[0], self.n_rows).T
        buf.free()
        return out

    def hostProbs(self) -> np.ndarray:
        return self.probs.download().reshape(self.n_allele, self.n_rows).T if self.n_rows and self.n_allele else np.array([])

    def hostLogProbs(self) -> np.ndarray:
        return self.L.download().reshape(self.n_allele, self.n_rows).T if self.L else np.array([])

    def free(self) -> None:
        for b in (self._probs, self._L, self.lidx, self.miss, self.nvar, self.miss8, self.msum, self._bound_flags):
            if b is not None:
                b.free()

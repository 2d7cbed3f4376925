"""
Host-side mirror of the tabulation stage interface (``graphkir/hisat2.py``).

Same names and argument meaning as the reference for the functions the
pipeline calls -- ``getVariants``, ``readPair``, ``filterRead``,
``extractVariant``, ``extractVariantFromBam``, ``loadReadsAndVariantsData``,
``writeReadsAndVariantsData``, ``removeMultipleMapped`` -- but the per-pair
variant walk runs on the GPU (``csrc/gk_tabulate.hip``) over packed records.

``SampleData`` is the in-memory hand-off between tabulation and typing: the
device CSR plus the variant table.  The reference hands the same information
over as ``{name}.variant.json`` (hisat2.py:847-866); that file is still
written/read for compatibility, but typing does not need to go through it.
"""
from __future__ import annotations

import json
import re
from dataclasses import asdict, dataclass, field, fields as dataclass_fields
from typing import Iterable, Iterator, TypedDict

import ctypes as C
import numpy as np

from . import _lib
from ._lib import Device, check, lib
from .engine import DeviceIndex, Tabulation
from .index import GkIndex, getVariants, readExons  # noqa: F401  (re-export: reference names)
from .msa2hisat import Variant
from .packed import packPairs
from .utils import logger


@dataclass
class PairRead:
    """One read pair with its positive / negative variant ids (hisat2.py:24-52)."""

    l_sam: str = ""
    r_sam: str = ""
    multiple: int = 1
    backbone: str = ""
    lpv: list[str] = field(default_factory=list)
    lnv: list[str] = field(default_factory=list)
    rpv: list[str] = field(default_factory=list)
    rnv: list[str] = field(default_factory=list)


class ReadsAndVariantsData(TypedDict):
    variants: list[Variant]
    reads: list[PairRead]


# ---------------------------------------------------------------- text side
def getNH(sam_info: str) -> int:
    m = re.search(r"NH:i:(\d+)", sam_info)
    return int(m.group(1)) if m else 1


def readSamLines(sam_file: str) -> Iterator[str]:
    """Name-collated SAM text (what ``samtools sort -n -O SAM`` prints, hisat2.py:103-110)."""
    import gzip
    opener = gzip.open if sam_file.endswith(".gz") else open
    with opener(sam_file, "rt") as f:
        for line in f:
            yield line.rstrip("\n")


def readBam(bam_file: str) -> Iterable[str]:
    """Name-collated SAM lines of an alignment file (hisat2.py:103-110).

    ``.sam`` / ``.sam.gz`` are read directly (they must already be name-collated); ``.bam`` is
    decoded and name-sorted natively (``packed.bamChunks``).  ``GK_TEST_HOOKS=bam_reader=samtools`` runs
    ``samtools sort -n`` like the reference instead (the cross-check of tools/check_against_samtools.sh)."""
    if bam_file.endswith((".sam", ".sam.gz")):
        return readSamLines(bam_file)
    from .utils import testHook
    if testHook("bam_reader") == "samtools":
        from .external_tools import runTool
        proc = runTool("samtools", ["samtools", "sort", "-n", bam_file, "-O", "SAM"], capture_output=True)
        return str(proc.stdout).split("\n")
    from .packed import bamChunks
    return (line for chunk in bamChunks(bam_file) for line in chunk.decode().split("\n") if line)


def pairLines(lines: Iterable[str]) -> Iterator[tuple[str, str]]:
    """Mate pairing of ``readPair`` (hisat2.py:228-276) on an iterable of SAM lines."""
    waiting: dict[tuple[str, str, str, int], str] = {}
    n_reads = n_pairs = 0
    for line in lines:
        if not line or line.startswith("@") or line.startswith("[bam_sort_core]"):
            continue
        f = line.split("\t", 8)
        if f[6] != "=":
            continue
        n_reads += 1
        flag = int(f[1])
        sec = flag & 256
        want = (f[0], f[2], f[7], sec)
        mate = waiting.get(want)
        if mate is None:
            waiting[(f[0], f[2], f[3], sec)] = line
            continue
        if (int(mate.split("\t", 2)[1]) | flag) & 192 != 192:
            logger.warning(f"[Graph] Read Pair strange case: {line} {mate}")
            continue
        del waiting[want]
        n_pairs += 1
        yield line, mate
    logger.info(f"[Graph] Reads: {n_reads} Pairs: {n_pairs}")


def readPair(bam_file: str) -> Iterator[tuple[str, str]]:
    return pairLines(readBam(bam_file))


def filterRead(line: str, num_editdist: int = 4) -> bool:
    """Concordant pair flag and NM <= 4 (hisat2.py:541-578); the device applies the same test."""
    cols = line.strip().split("\t")
    if not int(cols[1]) & 2:
        return False
    nm = None
    for c in cols[11:]:
        if c.startswith("NM"):
            nm = int(c[5:])
    return nm is not None and nm <= num_editdist


# ---------------------------------------------------------------- device hand-off
class PairsText:
    """The SAM lines of the emitted pairs, kept as the collated text + (left, right) line numbers instead of
    two Python strings per pair; ``pairs[i]`` gives ``(l_sam, r_sam)`` like the list it replaces, the
    native writers (``gk_json_write_reads`` / ``gk_bam_write_lines``) take the text and the numbers."""

    def __init__(self, blob: bytes, pair_lines: np.ndarray):
        self.blob = blob
        self.pair_lines = np.ascontiguousarray(pair_lines, dtype=np.int64).reshape(-1, 2)
        self._starts: np.ndarray | None = None

    def __len__(self) -> int:
        return len(self.pair_lines)

    def _line(self, k: int) -> str:
        if self._starts is None:
            nl = np.flatnonzero(np.frombuffer(self.blob, dtype=np.uint8) == 10)
            self._starts = np.concatenate([[0], nl + 1, [len(self.blob) + 1]]).astype(np.int64)
        a, b = int(self._starts[k]), int(self._starts[k + 1]) - 1
        return self.blob[a:b].decode().rstrip("\r")

    def __getitem__(self, i: int) -> tuple[str, str]:
        a, b = self.pair_lines[i]
        return self._line(int(a)), self._line(int(b))


class SampleData:
    """Tabulated sample: device CSR (``Tabulation``) + variant table + gene tables."""

    def __init__(self, tab: Tabulation, index: GkIndex, novel: list[Variant] | None = None, pairs_text=None,
                 ins_strings: list[str] | None = None):
        self.tab, self.index = tab, index
        self._novel = novel           # built lazily from the device keys when None
        self.ins_strings = ins_strings if ins_strings is not None else index.ins_strings
        tab.ins_strings = self.ins_strings
        self.pairs_text = pairs_text  # [(l_sam, r_sam)] of the input pairs, when available
        self._reads = None

    @property
    def novel(self) -> list[Variant]:
        """Novel variants of the sample (objects are only built when somebody needs them)."""
        if self._novel is None:
            self._novel = self.tab.novelVariants(self.ins_strings)
        return self._novel

    def novelOfGene(self, gene: str) -> list[Variant]:
        return [v for v in self.novel if v.ref == gene]

    @property
    def variants(self) -> list[Variant]:
        return self.index.variants + self.novel

    def variantsOfGene(self, gene: str) -> list[Variant]:
        g = self.index.gene_id.get(gene)
        out = []
        if g is not None:
            t = self.index.tables[g]
            out = self.index.variants[t.vbeg:t.vend]
        return out + [v for v in self.novel if v.ref == gene]

    def reads(self) -> list[PairRead]:
        """Materialise ``PairRead`` objects (outputs / API parity; not on the typing path)."""
        if self._reads is None:
            tab = self.tab
            off, ids = tab.offsets().astype(np.int64), tab.ids()
            names = np.array(tab.idNames(), dtype=object)
            genes, nh = tab.pairGene(), tab.pairNH()
            src = tab.pairSrc() if tab.info.d_pair_src else np.arange(tab.n_valid)
            out = []
            for i in range(tab.n_valid):
                o = off[4 * i:4 * i + 5]
                l_sam = r_sam = ""
                if self.pairs_text is not None:
                    l_sam, r_sam = self.pairs_text[int(src[i])]
                out.append(PairRead(
                    l_sam=l_sam, r_sam=r_sam, multiple=int(nh[i]), backbone=self.index.genes[int(genes[i])],
                    lpv=list(names[ids[o[0]:o[1]]]), rpv=list(names[ids[o[1]:o[2]]]),
                    lnv=list(names[ids[o[2]:o[3]]]), rnv=list(names[ids[o[3]:o[4]]])))
            self._reads = out
        return self._reads

    def asDict(self) -> ReadsAndVariantsData:
        return {"variants": self.variants, "reads": self.reads()}

    @classmethod
    def fromHost(cls, dev: Device, data: ReadsAndVariantsData) -> "SampleData":
        """Upload ``{"variants", "reads"}`` (e.g. a loaded ``.variant.json``) as a device CSR."""
        known = sorted(v for v in data["variants"] if not str(v.id).startswith("nv"))
        novel = [v for v in data["variants"] if str(v.id).startswith("nv")]
        genes = sorted(set(v.ref for v in data["variants"]) | set(r.backbone for r in data["reads"]))
        index = GkIndex.fromVariants(known, genes=genes)
        ordinal = {str(v.id): i for i, v in enumerate(index.variants + novel)}
        reads = data["reads"]
        n = len(reads)
        off = np.zeros(4 * n + 1, dtype=np.uint32)
        flat: list[int] = []
        for i, r in enumerate(reads):
            for k, lst in enumerate((r.lpv, r.rpv, r.lnv, r.rnv)):
                flat.extend(ordinal[v] for v in lst)
                off[4 * i + k + 1] = len(flat)
        ids = np.array(flat, dtype=np.uint32)
        gene = np.array([index.gene_id[r.backbone] for r in reads], dtype=np.uint8)
        nh = np.array([min(int(r.multiple), 255) for r in reads], dtype=np.uint8)
        dindex = DeviceIndex(dev, index)
        tab = Tabulation.__new__(Tabulation)
        tab.dev, tab.dindex, tab.mates, tab.n_pairs = dev, dindex, None, n
        h = C.c_void_p()
        check(lib().gk_tab_from_csr(dev.ctx, len(ordinal), n, off.ctypes.data, ids.ctypes.data if len(ids) else None,
                                    gene.ctypes.data if n else None, nh.ctypes.data if n else None, C.byref(h)))
        tab.handle = h
        info = _lib.TabInfo()
        check(lib().gk_tab_get_info(h, C.byref(info)))
        tab.info = info
        tab.n_valid, tab.n_ids = n, int(info.n_ids)
        tab.n_novel, tab.novel_base, tab._novel_keys = len(novel), 0, None
        tab._id_names = [str(v.id) for v in index.variants + novel]
        tab._variant_src = index.variants + novel
        self = cls(tab, index, novel)
        self._reads = list(reads)
        return self


class ParkedRecords:
    """A sample between its depth and its typing, when the two are a cohort apart (``--cn-cohort``: the copy numbers come
    from ONE fit on the depths of all samples, main.py:572-589, so no sample can be typed before the last one is read).

    What waits in HBM is the sample's packed records in compact form (``gk_mates_compact``: ~30 bytes per mate, ~150 MB
    per 5 M reads) instead of its tabulation (~1 GB): the lists are made again from them when the sample's turn comes
    (``restore``: expansion + ``gk_tabulate``, a millisecond or two) -- same records, same index, same first novel id, so
    the same lists and the same ``nv`` ids."""

    def __init__(self, data: "SampleData"):
        import ctypes as C
        tab = data.tab
        if tab.mates is None:
            raise ValueError("the packed records of the sample have been released already")
        self.dev, self.dindex, self.index = tab.dev, tab.dindex, data.index
        self.ins_strings = data.ins_strings
        self.n_pairs, self.novel_base = tab.n_pairs, tab.novel_base
        self.expect = (tab.n_valid, tab.n_ids, tab.n_novel)
        self.spill, self.correction = getattr(tab, "_spill", None), getattr(tab, "_correction", None)
        ptr, nbytes = C.c_uint64(), C.c_int64()
        check(lib().gk_mates_compact(self.dev.ctx, tab.mates.ptr, 2 * self.n_pairs, C.byref(ptr), C.byref(nbytes)))
        self.ptr, self.nbytes = ptr.value, int(nbytes.value)
        # the records came from another context's pool (the copier's) and a pool orders reuse on its own stream only: the
        # packing kernel on this stream must be through with them before the block can be handed out again
        self.dev.sync()
        tab.mates.free()
        tab.mates = None
        tab.close()

    def restore(self) -> "SampleData":
        """The tabulated sample again (its compact records are released)."""
        if not self.ptr:
            raise ValueError("restored already")
        mates = self.dev.alloc(2 * self.n_pairs, _lib.MATE_DTYPE)
        check(lib().gk_mates_expand(self.dev.ctx, self.ptr, 2 * self.n_pairs, mates.ptr))
        self.release()                      # stream-ordered: the expansion is queued before the block is reused
        tab = Tabulation(self.dindex, mates, novel_base=self.novel_base, dev=self.dev, spill=self.spill,
                         correction=self.correction)
        if (tab.n_valid, tab.n_ids, tab.n_novel) != self.expect:
            raise AssertionError(f"second tabulation differs from the first: {(tab.n_valid, tab.n_ids, tab.n_novel)} "
                                 f"!= {self.expect}")
        return SampleData(tab, self.index, None, ins_strings=self.ins_strings)

    def release(self) -> None:
        if self.ptr:
            lib().gk_free(self.dev.ctx, self.ptr)
            self.ptr = 0

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


COMPACT_FORMAT = "graphkir-variant-csr-1"


def indexFingerprint(index: GkIndex) -> np.ndarray:
    """(number of variants, wrap-around sum and xor of the packed keys): ties a compact file to its index."""
    key = index.key.astype(np.uint64)
    return np.array([len(key), int(key.sum(dtype=np.uint64)), int(np.bitwise_xor.reduce(key)) if len(key) else 0],
                    dtype=np.uint64)


def writeCompact(data: SampleData, filename: str, index_ref: str = "", background=None):
    """Binary side-format of the ``.variant.json`` hand-off (hisat2.py:847-866): the tabulation's CSR
    (offsets + variant ordinals in list order lpv, rpv, lnv, rnv), backbone and NH per pair, the keys
    of the novel variants and the inserted-string table -- ~4 bytes per variant hit instead of JSON
    with embedded SAM text.  Index variants are not stored: ``index_ref`` names the index files and
    a fingerprint of the packed keys is checked on load.

    ``background``: an executor; the arrays are fetched from the device now, the file (a few hundred MB for
    a deep sample, most of the time is the archive's CRC) is written there and the future is returned."""
    tab = data.tab
    if getattr(tab, "_variant_src", None) is not None:
        raise ValueError("compact files are written from tabulations made against an index")
    path = filename if filename.endswith(".npz") else filename + ".npz"
    fields = dict(
        format=np.array(COMPACT_FORMAT), index_ref=np.array(index_ref), fingerprint=indexFingerprint(data.index),
        genes=np.array(data.index.genes), off=tab.offsets(), ids=tab.ids(), pair_gene=tab.pairGene(),
        pair_nh=tab.pairNH(), novel_key=tab.novelKeys(), novel_base=np.array(tab.novel_base),
        ins_strings=np.array(data.ins_strings if data.ins_strings else [""]),
        n_ins=np.array(len(data.ins_strings or [])))
    if background is None:
        np.savez(path, **fields)
        return None
    return background.submit(np.savez, path, **fields)


RECORDS_FORMAT = "graphkir-records-1"


def writeCompactRecords(compact, pack: dict, novel_base: int, index: GkIndex, filename: str, index_ref: str = "",
                        background=None):
    """The hand-off as the sample's PACKED RECORDS in their compact form (``packed.CompactMates``: ~30 bytes per mate --
    70 MB for 2 M reads where the tabulated lists of ``writeCompact`` are 250 MB that first have to come back from the
    device): what the command line writes when the records are at hand.  ``loadCompact`` expands and tabulates them
    again -- same records, same index, same first novel id: the same lists (0.9 ms per million pairs on the device).
    ``pack``: the dictionary of ``packAlignments`` (pairs beyond the 128-byte record, inserted strings)."""
    path = filename if filename.endswith(".npz") else filename + ".npz"
    spill = pack["counts"].get("spill")
    strings = pack["strings"]
    fields = dict(
        format=np.array(RECORDS_FORMAT), index_ref=np.array(index_ref), fingerprint=indexFingerprint(index),
        genes=np.array(index.genes), words=np.asarray(compact.words), n_mates=np.array(compact.n_mates),
        novel_base=np.array(novel_base), ins_strings=np.array(strings if strings else [""]),
        n_ins=np.array(len(strings or [])),
        spill_wide=(np.ascontiguousarray(spill[0]).view(np.uint8).reshape(-1) if spill is not None else np.zeros(0, np.uint8)),
        spill_pair=(np.ascontiguousarray(spill[1], dtype=np.int64) if spill is not None else np.zeros(0, np.int64)))
    if background is None:
        np.savez(path, **fields)
        return None
    return background.submit(np.savez, path, **fields)


def _loadCompactRecords(z, filename: str, dev: Device, index: GkIndex, dindex: DeviceIndex) -> SampleData:
    from . import _lib as gl
    from .packed import CompactMates
    compact = CompactMates.__new__(CompactMates)
    compact.words = np.ascontiguousarray(z["words"], np.uint32)
    compact.n_mates = int(z["n_mates"])
    spill = None
    if len(z["spill_pair"]):
        spill = (np.ascontiguousarray(z["spill_wide"]).view(gl.MATE_WIDE_DTYPE), np.ascontiguousarray(z["spill_pair"], np.int64))
    mates = compact.toDevice(dev, wait=True)
    tab = Tabulation(dindex, mates, novel_base=int(z["novel_base"]), dev=dev, spill=spill)
    mates.free()
    tab.mates = None
    strings = [str(x) for x in z["ins_strings"][:int(z["n_ins"])]]
    return SampleData(tab, index, None, ins_strings=strings)


_index_cache: dict[str, GkIndex] = {}


def loadCompact(filename: str, dev: Device | None = None, index: GkIndex | None = None,
                dindex: DeviceIndex | None = None) -> SampleData:
    """Load a hand-off file: ``writeCompact`` output as a device CSR (no re-tabulation, no Variant objects built), or
    ``writeCompactRecords`` output (the packed records: expanded and tabulated again on the device)."""
    z = np.load(filename, allow_pickle=False)
    if str(z["format"]) not in (COMPACT_FORMAT, RECORDS_FORMAT):
        raise ValueError(f"{filename}: unknown format {z['format']!r}")
    if index is None:
        ref = str(z["index_ref"])
        if not ref:
            raise ValueError(f"{filename} does not name its index; pass index=")
        index = _index_cache.get(ref)
        if index is None:
            index = _index_cache[ref] = GkIndex.load(ref)
    if not np.array_equal(z["fingerprint"], indexFingerprint(index)) or list(z["genes"]) != list(index.genes):
        raise ValueError(f"{filename} was tabulated against a different index")
    dev = dev or Device()
    dindex = dindex or DeviceIndex(dev, index)
    if str(z["format"]) == RECORDS_FORMAT:
        return _loadCompactRecords(z, filename, dev, index, dindex)
    off, ids = np.ascontiguousarray(z["off"], np.uint32), np.ascontiguousarray(z["ids"], np.uint32)
    gene, nh = np.ascontiguousarray(z["pair_gene"], np.uint8), np.ascontiguousarray(z["pair_nh"], np.uint8)
    novel_key = np.ascontiguousarray(z["novel_key"], np.uint64)
    n = len(gene)
    tab = Tabulation.__new__(Tabulation)
    tab.dev, tab.dindex, tab.mates, tab.n_pairs = dev, dindex, None, n
    h = C.c_void_p()
    check(lib().gk_tab_from_csr(dev.ctx, index.n_variant + len(novel_key), n, off.ctypes.data,
                                ids.ctypes.data if len(ids) else None, gene.ctypes.data if n else None,
                                nh.ctypes.data if n else None, C.byref(h)))
    tab.handle = h
    info = _lib.TabInfo()
    check(lib().gk_tab_get_info(h, C.byref(info)))
    tab.info = info
    tab.n_valid, tab.n_ids = n, int(info.n_ids)
    tab.n_novel, tab.novel_base, tab._novel_keys = len(novel_key), int(z["novel_base"]), novel_key
    strings = [str(x) for x in z["ins_strings"][:int(z["n_ins"])]]
    return SampleData(tab, index, None, ins_strings=strings)


def extractVariant(pair_reads: Iterable[tuple[str, str]], index: GkIndex | list[Variant], dev: Device | None = None,
                   dindex: DeviceIndex | None = None, pileup=None) -> SampleData:
    """SAM pairs -> tabulated sample on the device (extractVariant, hisat2.py:803-844).

    Pairs are decoded to packed records on the host, then filterRead + the variant walk +
    positive/negative extraction run in ``gk_tabulate``.  ``pileup``: the dictionary of ``getPileupBaseRatio``
    (``{(ref, pos): {base: share, "all": depth}}``); every mismatch goes through ``hisat2.errorCorrection``
    (609-654) before its id is looked up -- as a per-position table on the device (``gk_tabulate_corrected``).
    """
    if not isinstance(index, GkIndex):
        index = GkIndex.fromVariants(index)
    dev = dev or Device()
    dindex = dindex or DeviceIndex(dev, index)
    pairs = list(pair_reads)
    spill: list = []
    rec, table = packPairs(pairs, index, spill=spill)
    base = Variant.novel_id
    from .packed import spillArrays
    correction = None
    if pileup:      # the reference's `if pileup:` (hisat2.py:685): None and an empty dictionary both mean "off"
        from .pileup import correctionFromRatios
        correction = correctionFromRatios(pileup, index)
    tab = Tabulation(dindex, rec, novel_base=base, spill=spillArrays(spill), correction=correction)
    Variant.novel_id = base + tab.n_novel
    logger.info(f"[Graph] Filterd pairs: {tab.n_valid}")
    return SampleData(tab, index, None, pairs_text=pairs, ins_strings=table.strings)


def packAlignments(source, index: GkIndex, keep_text: bool = True) -> dict:
    """Host half of ``extractVariantFromText``: alignment file (or byte chunks) -> packed records.

    Native code with the GIL released (``gk_bam_*`` / ``gk_packer_*``), so a cohort run can pack the
    next sample on a helper thread (``cohort.prefetched``) while the current one is on the GPU."""
    from .packed import packBam, packText, readChunks
    if isinstance(source, str) and source.endswith(".bam") and not keep_text:
        # nobody needs the SAM text: the BAM records go to the packer in binary form
        chunks = None
        rec, table, pair_lines, counts = packBam(source, index)
    else:
        chunks = readChunks(source) if isinstance(source, str) else source
        if keep_text:
            chunks = list(chunks)
        rec, table, pair_lines, counts = packText(chunks, index)
    pairs_text = PairsText(b"".join(chunks), pair_lines) if keep_text else None
    return {"records": rec, "strings": table.strings, "pairs_text": pairs_text, "counts": counts}


def extractVariantFromPacked(pack: dict, index: GkIndex, dev: Device | None = None,
                             dindex: DeviceIndex | None = None, correction=None, mates=None) -> SampleData:
    """Device half: packed records -> tabulated sample (novel ids continue the process-wide counter).
    ``correction``: see ``Tabulation`` (pileup error correction, hisat2.py:609-654).  ``mates``: the records already in
    HBM (a device buffer: the pipeline copies them on a stream of its own), else they are uploaded here."""
    dev = dev or Device()
    dindex = dindex or DeviceIndex(dev, index)
    logger.info(f"[Graph] Reads: {pack['counts']['reads']} Pairs: {pack['counts']['pairs']}")
    base = Variant.novel_id
    spill = pack["counts"].get("spill")
    if spill is not None:
        logger.info(f"[Graph] Pairs beyond the 128-byte record, kept in the wide format: {len(spill[1])}")
    tab = Tabulation(dindex, pack["records"] if mates is None else mates, novel_base=base, dev=dev, correction=correction,
                     spill=spill)
    Variant.novel_id = base + tab.n_novel
    logger.info(f"[Graph] Filterd pairs: {tab.n_valid}")
    return SampleData(tab, index, None, pairs_text=pack["pairs_text"], ins_strings=pack["strings"])


def extractVariantFromText(source, index: GkIndex, dev: Device | None = None, dindex: DeviceIndex | None = None,
                           keep_text: bool = True, correction=None) -> SampleData:
    """Alignment file (``.sam`` / ``.sam.gz`` name-collated, or ``.bam``) or iterable of byte chunks ->
    tabulated sample.

    Same result as ``extractVariant(readPair(path), ...)`` with the pairing and decoding done natively
    (``packAlignments``).  ``keep_text``: keep the SAM lines of the emitted pairs (needed only for the
    ``l_sam`` / ``r_sam`` fields of ``.variant.json`` and the BAM rewrites)."""
    return extractVariantFromPacked(packAlignments(source, index, keep_text), index, dev, dindex, correction)


def writeReadsAndVariantsData(reads_data: ReadsAndVariantsData, filename: str) -> None:
    with open(filename, "w") as f:
        json.dump({"variants": [asdict(v) for v in reads_data["variants"]],
                   "reads": [asdict(r) for r in reads_data["reads"]]}, f)


def writeSampleJson(data: "SampleData", filename: str) -> None:
    """``writeReadsAndVariantsData(data.asDict(), filename)``, byte for byte, without materialising a
    ``PairRead`` per pair and without ``dataclasses.asdict`` / the pure-Python ``json.dump`` iterator
    (together ~75 s per million pairs): the id lists are joined from pre-encoded names, only the SAM
    lines go through the JSON string encoder.  Key order = field order of ``Variant`` / ``PairRead``."""
    from json.encoder import encode_basestring_ascii as enc   # what json.dump uses (ensure_ascii=True)
    tab = data.tab
    off, ids = tab.offsets().astype(np.int64), tab.ids()
    quoted = np.array([enc(n) for n in tab.idNames()], dtype=object)
    genes = [enc(g) for g in data.index.genes]
    gene_of, nh = tab.pairGene().tolist(), tab.pairNH().tolist()
    src = (tab.pairSrc() if tab.info.d_pair_src else np.arange(tab.n_valid)).tolist()
    text = data.pairs_text
    variants = [{"pos": v.pos, "typ": v.typ, "ref": v.ref, "val": v.val, "id": v.id, "length": v.length,
                 "allele": v.allele, "freq": v.freq, "ignore": v.ignore, "in_exon": v.in_exon} for v in data.variants]
    assert [f.name for f in dataclass_fields(Variant)] == list(variants[0]) if variants else True
    assert [f.name for f in dataclass_fields(PairRead)] == ["l_sam", "r_sam", "multiple", "backbone", "lpv", "lnv", "rpv", "rnv"]
    if isinstance(text, PairsText):   # the reads array is formatted natively from the collated text
        import ctypes as C
        with open(filename, "w") as f:
            f.write('{"variants": ')
            f.write(json.dumps(variants))
            f.write(', "reads": ')
        names = tab.idNames()
        c_names = (C.c_char_p * max(1, len(names)))(*[n.encode() for n in names])
        c_genes = (C.c_char_p * max(1, len(data.index.genes)))(*[g.encode() for g in data.index.genes])
        off32 = np.ascontiguousarray(tab.offsets(), dtype=np.uint32)
        ids32 = np.ascontiguousarray(ids, dtype=np.uint32)
        src64 = np.ascontiguousarray(src, dtype=np.int64)
        gene8 = np.ascontiguousarray(tab.pairGene(), dtype=np.uint8)
        nh8 = np.ascontiguousarray(tab.pairNH(), dtype=np.uint8)
        check(lib().gk_json_write_reads(filename.encode(), text.blob, len(text.blob), text.pair_lines.ctypes.data,
                                        len(text.pair_lines), src64.ctypes.data, tab.n_valid, off32.ctypes.data,
                                        ids32.ctypes.data if len(ids32) else None, c_names, len(names), c_genes,
                                        len(data.index.genes), gene8.ctypes.data if len(gene8) else None,
                                        nh8.ctypes.data if len(nh8) else None))
        with open(filename, "a") as f:
            f.write("}")
        return
    with open(filename, "w") as f:
        f.write('{"variants": ')
        f.write(json.dumps(variants))
        f.write(', "reads": [')
        bounds = off.tolist()
        batch: list[str] = []
        first = True
        for i in range(tab.n_valid):
            o = bounds[4 * i:4 * i + 5]
            lpv, rpv, lnv, rnv = (", ".join(quoted[ids[o[k]:o[k + 1]]]) for k in range(4))   # CSR order: lpv rpv lnv rnv
            l_sam, r_sam = text[src[i]] if text is not None else ("", "")
            batch.append(f'{{"l_sam": {enc(l_sam)}, "r_sam": {enc(r_sam)}, "multiple": {nh[i]}, '
                         f'"backbone": {genes[gene_of[i]]}, "lpv": [{lpv}], "lnv": [{lnv}], "rpv": [{rpv}], "rnv": [{rnv}]}}')
            if len(batch) == 4096:
                f.write(("" if first else ", ") + ", ".join(batch))
                first, batch = False, []
        if batch:
            f.write(("" if first else ", ") + ", ".join(batch))
        f.write("]}")


def loadReadsAndVariantsData(filename: str) -> ReadsAndVariantsData:
    with open(filename) as f:
        raw = json.load(f)
    return {"variants": [Variant(**v) for v in raw["variants"]],
            "reads": [PairRead(**r) for r in raw["reads"]]}


def removeMultipleMapped(reads_data: ReadsAndVariantsData) -> ReadsAndVariantsData:
    return {"variants": reads_data["variants"],
            "reads": [r for r in reads_data["reads"] if r.multiple == 1]}


def alignmentHeader(path: str) -> str:
    """``@`` header lines of a ``.bam`` / ``.sam`` / ``.sam.gz`` file (readBamHeader, hisat2.py:113-118)."""
    if path.endswith(".bam"):
        from .packed import bamHeader
        return bamHeader(path)
    import gzip
    opener = gzip.open if path.endswith(".gz") else open
    out = []
    with opener(path, "rt") as f:
        for line in f:
            if not line.startswith("@"):
                break
            out.append(line)
    return "".join(out)


def saveReadsToBam(data: "SampleData", filename_prefix: str, bam_file: str, filter_multi_mapped: bool = False) -> None:
    """Write the filter-passing pairs (optionally only NH == 1) as ``{filename_prefix}.bam``
    (hisat2.py:880-901: saveSam + samtobam), coordinate-sorted like ``samtools sort``, with the header
    of ``bam_file``.  Encoded and compressed natively (``packed.writeBam``)."""
    from .packed import writeBam
    if data.pairs_text is None:
        raise ValueError("the SAM text of the pairs was not kept (keep_text=False)")
    tab = data.tab
    src = tab.pairSrc() if tab.info.d_pair_src else np.arange(tab.n_valid)
    keep = src[tab.pairNH() == 1] if filter_multi_mapped else src
    text = data.pairs_text
    if isinstance(text, PairsText):   # the selected lines go to the encoder as line numbers
        header = alignmentHeader(bam_file).encode()
        picked = np.ascontiguousarray(text.pair_lines[np.asarray(keep, dtype=np.int64)].reshape(-1), dtype=np.int64)
        check(lib().gk_bam_write_lines((filename_prefix + ".bam").encode(), header, len(header), text.blob, len(text.blob),
                                       picked.ctypes.data if len(picked) else None, len(picked), 1))
        return
    body = "".join(f"{text[int(i)][0]}\n{text[int(i)][1]}\n" for i in keep)
    writeBam(filename_prefix + ".bam", alignmentHeader(bam_file) + body)


def extractVariantFromBam(index: str, bam_file: str, output_prefix: str, error_correction: bool = True,
                          dev: Device | None = None) -> SampleData:
    """index + alignments -> ``{output_prefix}.json``, ``.bam`` and ``.no_multi.bam`` (hisat2.py:904-940)."""
    gk = GkIndex.load(index)
    correction = None
    if error_correction:   # hisat2.py:925-928: pileup of the same BAM decides which mismatches are read errors
        from .pileup import correctionTable, pileupCounts
        counts, pos0 = pileupCounts(bam_file, gk)
        correction = (correctionTable(counts), pos0)
    data = extractVariantFromText(bam_file, gk, dev=dev, keep_text=True, correction=correction)
    logger.debug(f"[Graph] Save allele per reads in {output_prefix}.json")
    writeSampleJson(data, f"{output_prefix}.json")
    logger.debug(f"[Graph] Save filtered reads in {output_prefix}.bam")
    saveReadsToBam(data, output_prefix, bam_file)
    logger.debug(f"[Graph] Save filtered and unique reads in {output_prefix}.no_multi.bam")
    saveReadsToBam(data, output_prefix + ".no_multi", bam_file, filter_multi_mapped=True)
    return data

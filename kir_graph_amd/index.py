"""
Host-side container of the HISAT2-format KIR index used by the typing path.

Restates the index readers of the reference (``graphkir/hisat2.py``):
``readVariants`` 159-180 (.snp), ``readLink`` 121-134 (.link), ``readExons``
137-156 (.locus), ``isInExon`` 206-225 and ``getVariants`` 183-203, and lays
the result out as flat arrays that are uploaded once per run to HBM:

* ``key``   u64[V]  packed sort key  ref:8 | pos:24 | type:2 | val:30, sorted
  ascending == the reference's ``sorted(variants)`` (msa2hisat.py:48-53)
* ``gene_vbeg`` i32[G+1]  range of each backbone's variants in ``key``
* per gene: allele x variant membership as bit rows ``mask[v][A/32]`` (LDS
  staged by the compatibility kernel).

``val`` field of the key: ASCII code for ``single``, length for ``deletion``,
rank of the inserted string in the sorted string table for ``insertion``.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

from .msa2hisat import Variant, TYPE_RANK

KEY_POS_SHIFT = 32
KEY_REF_SHIFT = 56
KEY_TYP_SHIFT = 30
KEY_VAL_MASK = (1 << 30) - 1
MAX_POS = (1 << 24) - 1


def packKey(ref_id: int, pos: int, typ_rank: int, val: int) -> int:
    """Pack (ref, pos, type, val) into the device sort key."""
    if not (0 <= pos <= MAX_POS):
        raise ValueError(f"variant position {pos} outside 24-bit device key range")
    if not (0 <= val <= KEY_VAL_MASK):
        raise ValueError(f"variant value {val} outside 30-bit device key range")
    return (ref_id << KEY_REF_SHIFT) | (pos << KEY_POS_SHIFT) | (typ_rank << KEY_TYP_SHIFT) | val


def readSnp(index: str) -> list[Variant]:
    """``.snp``: id, type, backbone, 0-based pos, value (hisat2.py:159-180)."""
    out = []
    with open(index + ".snp") as f:
        for line in f:
            vid, typ, ref, pos, val = line.strip().split("\t")
            out.append(Variant(pos=int(pos), typ=typ, ref=ref, id=vid,
                               val=int(val) if typ == "deletion" else val))
    return out


def readLink(index: str) -> dict[str, list[str]]:
    """``.link``: variant id -> allele names (hisat2.py:121-134)."""
    links = {}
    with open(index + ".link") as f:
        for line in f:
            vid, names = line.strip().split("\t")
            links[vid] = names.split()
    return links


def readExons(index: str) -> dict[str, list[tuple[int, int]]]:
    """``.locus``: backbone -> 0-based (start-1, end-1) exon pairs (hisat2.py:137-156)."""
    exons = {}
    with open(index + ".locus") as f:
        for line in f:
            gene, _, _, _, _, exon_str, _ = line.split("\t")
            exons[gene] = [(int(s) - 1, int(e) - 1)
                           for s, e in (x.split("-") for x in exon_str.split(" "))]
    return exons


def isInExon(exons: list[tuple[int, int]], v: Variant) -> bool:
    """Exon membership rule of hisat2.py:206-225 (deletions reaching an exon count)."""
    for s, e in exons:
        if s <= v.pos < e:
            return True
        if v.typ == "deletion" and v.pos < s and v.pos + int(v.val) >= s:  # type: ignore[arg-type]
            return True
    return False


def getVariants(index: str) -> list[Variant]:
    """All index variants with alleles and exon flags, sorted (hisat2.py:183-203)."""
    variants = readSnp(index)
    links = readLink(index)
    exons = readExons(index)
    for v in variants:
        assert v.id
        v.allele = links.get(v.id, [])
        v.in_exon = isInExon(exons[v.ref], v)
    return sorted(variants)


@dataclass
class GeneTable:
    """Allele universe and bit rows of one backbone."""

    name: str
    vbeg: int
    vend: int
    alleles: list[str]
    mask: np.ndarray  # uint32 [vend-vbeg, words]

    @property
    def n_allele(self) -> int:
        return len(self.alleles)

    @property
    def words(self) -> int:
        return self.mask.shape[1]


def buildMask(variants: list[Variant], alleles: list[str]) -> np.ndarray:
    """Bit rows: bit a of row v set iff alleles[a] carries variants[v]."""
    words = max(1, (len(alleles) + 31) // 32)
    col = {a: i for i, a in enumerate(alleles)}
    mask = np.zeros((len(variants), words), dtype=np.uint32)
    rows, cols = [], []
    for i, v in enumerate(variants):          # (variant, allele) incidences as two flat lists, set in one numpy call
        js = [col[a] for a in v.allele if a in col]
        cols += js
        rows += [i] * len(js)
    if rows:
        r, c = np.asarray(rows, dtype=np.int64), np.asarray(cols, dtype=np.int64)
        np.bitwise_or.at(mask, (r, c >> 5), (np.uint32(1) << (c & 31).astype(np.uint32)))
    return mask


@dataclass
class GkIndex:
    """Flat, sorted view of the index (host copy of what lives in HBM)."""

    variants: list[Variant]
    genes: list[str]
    gene_id: dict[str, int]
    key: np.ndarray            # uint64 [V]
    gene_vbeg: np.ndarray      # int32 [G+1]
    in_exon: np.ndarray        # uint8 [V]
    ins_strings: list[str]
    ins_id: dict[str, int]
    tables: list[GeneTable] = field(default_factory=list)
    exons: dict[str, list[tuple[int, int]]] = field(default_factory=dict)

    @property
    def n_variant(self) -> int:
        return len(self.variants)

    @classmethod
    def fromVariants(cls, variants: list[Variant], genes: list[str] | None = None,
                     exons: dict[str, list[tuple[int, int]]] | None = None) -> "GkIndex":
        """Build from a (possibly unsorted) variant list; keeps the reference order."""
        variants = sorted(variants)
        refs = sorted(set(v.ref for v in variants) | set(genes or []))
        if len(refs) > 255:
            raise ValueError("more than 255 backbones are not supported by the packed key")
        gene_id = {g: i for i, g in enumerate(refs)}
        ins_strings = sorted(set(str(v.val) for v in variants if v.typ == "insertion"))
        ins_id = {s: i for i, s in enumerate(ins_strings)}

        key = np.zeros(len(variants), dtype=np.uint64)
        for i, v in enumerate(variants):
            key[i] = packKey(gene_id[v.ref], v.pos, TYPE_RANK[v.typ], valCode(v, ins_id))
        if len(key) > 1:
            if not np.all(key[1:] >= key[:-1]):
                raise AssertionError("packed key order disagrees with Variant order")
            if np.any(key[1:] == key[:-1]):
                raise ValueError("duplicate (pos, ref, typ, val) rows in the index")
        ref_ids = np.array([gene_id[v.ref] for v in variants], dtype=np.int64)
        gene_vbeg = np.searchsorted(ref_ids, np.arange(len(refs) + 1)).astype(np.int32)
        in_exon = np.array([v.in_exon for v in variants], dtype=np.uint8)

        self = cls(variants=variants, genes=refs, gene_id=gene_id, key=key,
                   gene_vbeg=gene_vbeg, in_exon=in_exon, ins_strings=ins_strings,
                   ins_id=ins_id, exons=exons or {})
        for g, name in enumerate(refs):
            b, e = int(gene_vbeg[g]), int(gene_vbeg[g + 1])
            gv = variants[b:e]
            names = sorted(set(a for v in gv for a in v.allele))
            self.tables.append(GeneTable(name, b, e, names, buildMask(gv, names)))
        return self

    @classmethod
    def load(cls, index: str) -> "GkIndex":
        """Read ``{index}.snp/.link/.locus``."""
        exons = readExons(index)
        return cls.fromVariants(getVariants(index), genes=list(exons), exons=exons)


def valCode(v: Variant, ins_id: dict[str, int]) -> int:
    """30-bit value field of the key (see module docstring)."""
    if v.typ == "single":
        s = str(v.val)
        if len(s) != 1:
            raise ValueError(f"single variant with value {s!r}")
        return ord(s)
    if v.typ == "deletion":
        return int(v.val)  # type: ignore[arg-type]
    if v.typ == "insertion":
        return ins_id[str(v.val)]
    raise ValueError(f"variant type {v.typ!r} has no device key")

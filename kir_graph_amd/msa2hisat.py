"""
The variant record type of the hot path.

Only the record type of the reference's ``graphkir/msa2hisat.py`` is on the
typing path (``Variant``, msa2hisat.py:15-63); the MSA -> HISAT2 index writer
in the rest of that file is out of scope.

Semantics kept (they are part of numeric parity):

* ordering  = (ref, pos, type-rank{insertion,single,deletion,match}, val)
* identity  = (pos, ref, typ, val) -- ``id``, ``allele`` and ``length`` are
  NOT part of equality/hash, so a walker-made variant finds the index record.
* ``novel_id`` is a process-wide counter that is never reset between samples.

On the device the same ordering is carried by a packed 64-bit key, see
``kir_graph_amd.index.variantKey``.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import ClassVar

TYPE_RANK: dict[str, int] = {"insertion": 0, "single": 1, "deletion": 2, "match": 3}


@dataclass
class Variant:
    """One graph variant (index record or read-derived)."""

    pos: int
    typ: str
    ref: str
    val: None | int | str = None
    id: None | str = None
    length: int = 0

    allele: list[str] = field(default_factory=list)
    freq: None | float = None
    ignore: None | bool = False
    in_exon: bool = False

    # process-wide counters (reference: msa2hisat.py:35-37)
    count: ClassVar[int] = 0
    haplo_id: ClassVar[int] = 0
    novel_id: ClassVar[int] = 0
    min_freq_threshold: ClassVar[float] = 0.1
    order_type: ClassVar[dict[str, int]] = TYPE_RANK
    order_nuc: ClassVar[dict[str, int]] = {"A": 0, "C": 1, "G": 2, "T": 3}

    def _sort_key(self) -> tuple:
        return (self.ref, self.pos, TYPE_RANK[self.typ], self.val)

    def _ident(self) -> tuple:
        return (self.pos, self.ref, self.typ, self.val)

    def __lt__(self, other: object) -> bool:
        if not isinstance(other, Variant):
            return NotImplemented
        return self._sort_key() < other._sort_key()

    def __eq__(self, other: object) -> bool:
        if not isinstance(other, Variant):
            return NotImplemented
        return self._ident() == other._ident()

    def __hash__(self) -> int:
        return hash(self._ident())

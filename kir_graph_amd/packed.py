"""
Packed alignment records: the device's input format (``gk_mate``, 128 bytes).

Two producers, one format (``include/graphkir_hip.h``):

* ``packPairs``  -- SAM text pairs -> records.  This is pure decoding of the
  fields the reference reads from the text (``graphkir/hisat2.py:342-350``
  POS/CIGAR/SEQ/MD/Zs, ``551-569`` FLAG/NM, ``95-100`` NH).  The CIGAR / MD /
  Zs consistency checks of ``recordToRawVariant`` (asserts at 416-417, 461,
  512-514; ``NotImplementedError`` for ``N`` at 499-502) are evaluated here,
  because they are properties of the text; the variant walk itself runs on the
  device (``csrc/gk_tabulate.hip``).
* ``packSample`` -- synthetic event arrays (``synth.SynthSample``) -> records,
  vectorised, for large inputs.

Inserted sequences are interned in ``InsTable``: ids below ``n_index`` are the
rank of the string among the index's insertion alleles (that rank is the
``val`` field of the variant key), larger ids are novel strings in first-seen
order.
"""
from __future__ import annotations

import re

import numpy as np

from ._lib import MATE_DTYPE, CIG_M, CIG_I, CIG_D, CIG_S, NM_ABSENT, MAX_CIG, MAX_MM, MAX_INS, MAX_EV
from .index import GkIndex

_CIGAR = re.compile(r"(\d+)(\w)")
_MDTOK = re.compile(r"\d+|.")
_NH = re.compile(r"NH:i:(\d+)")
MAX_OPLEN = 4095


class PackCapacityError(ValueError):
    """A filter-passing record does not fit the fixed-size device record."""


class InsTable:
    """Inserted-sequence interning shared by index and reads."""

    def __init__(self, index: GkIndex):
        self.strings = list(index.ins_strings)
        self.ids = dict(index.ins_id)
        self.n_index = len(self.strings)

    def intern(self, s: str) -> int:
        i = self.ids.get(s)
        if i is None:
            i = len(self.strings)
            self.ids[s] = i
            self.strings.append(s)
        return i


def _decodeTags(cols: list[str]):
    nm = None
    md = zs = None
    for c in cols:
        if c.startswith("NM"):
            nm = int(c[5:])
        elif md is None and c.startswith("MD"):
            md = c[5:]
        elif zs is None and c.startswith("Zs"):
            zs = c[5:]
    return nm, md, zs


def _passes(flag: int, nm) -> bool:
    return bool(flag & 2) and nm is not None and nm <= 4


def _walkText(pos0: int, cigar: str, seq: str, md_s: str | None, zs_s: str | None, table: InsTable):
    """Co-walk CIGAR / MD / Zs text; returns (ops, mismatches, insertion ids, clipped)."""
    md = [int(t) if t.isdigit() else t for t in _MDTOK.findall(md_s)] if md_s is not None else []
    zs = [(int(a), b) for a, b, _ in (x.split("|") for x in zs_s.split(","))] if zs_s else []
    ops: list[tuple[int, int]] = []
    mms: list[tuple[int, int]] = []
    ins: list[int] = []
    ref = 0          # reference offset from pos0
    ri = 0           # read offset
    mi = 0           # MD cursor
    owed = 0         # matched bases still owed by the last MD number
    zi = zpos = 0
    clipped = False

    def take_zs(kind: str) -> None:
        nonlocal zi, zpos
        if zi < len(zs) and zs[zi][1] == kind and ri + owed == zpos + zs[zi][0]:
            zpos += zs[zi][0] + (1 if kind == "S" else 0)
            zi += 1

    for k, (n_s, op) in enumerate(_CIGAR.findall(cigar)):
        n = int(n_s)
        if mi < len(md) and md[mi] == 0:
            mi += 1
        if op == "M":
            ops.append((CIG_M, n))
            done = 0
            while True:
                if owed <= done and mi < len(md) and type(md[mi]) is int:
                    owed += md[mi]
                    mi += 1
                if owed >= n:
                    owed -= n
                    break
                base = seq[ri + owed]
                if md[mi] == 0:
                    mi += 1
                assert str(md[mi]) in "ACGT", "MD mismatch token is not a base"
                assert str(md[mi]) != base, "MD reference base equals the read base"
                mi += 1
                take_zs("S")
                mms.append((ref + owed, ord(base)))
                owed += 1
                done = owed
                if owed == n:
                    owed = 0
                    break
            ref += n
            ri += n
        elif op == "I":
            ops.append((CIG_I, n))
            take_zs("I")
            ins.append(table.intern(seq[ri:ri + n]))
            ri += n
        elif op == "D":
            ops.append((CIG_D, n))
            assert md[mi] == "^", "MD has no deletion at a D op"
            mi += 1
            while mi < len(md) and type(md[mi]) is not int and str(md[mi]) in "ACGT":
                mi += 1
            take_zs("D")
            ref += n
        elif op == "S":
            clipped = True
            zpos += n
            ri += n
        elif op == "N":
            raise NotImplementedError("Cannot typing with splicing")
        else:
            raise NotImplementedError
    if mi < len(md) and md[mi] == 0:
        mi += 1
    assert zi == len(zs), "Zs entries do not line up with the alignment"
    assert mi == len(md), "MD not fully consumed"
    assert ri == len(seq), "CIGAR does not cover the read"
    return ops, mms, ins, clipped


WIDE_CIG, WIDE_MM, WIDE_INS, WIDE_EV = 128, 256, 120, 384      # gk_mate_wide (include/graphkir_hip.h)


def packPairs(pairs, index: GkIndex, table: InsTable | None = None, spill: list | None = None) -> tuple[np.ndarray, InsTable]:
    """[(left_line, right_line)] -> mate records (2 per pair, left first).

    ``spill``: a list that receives ``(pair index, two gk_mate_wide records)`` for every pair that does not fit
    ``gk_mate`` (``spillArrays`` turns it into what ``Tabulation`` takes); without it such a pair raises
    ``PackCapacityError``."""
    from ._lib import MATE_WIDE_DTYPE, SPILLED
    table = table or InsTable(index)
    pairs = list(pairs)
    rec = np.zeros(2 * len(pairs), dtype=MATE_DTYPE)
    for p, pair in enumerate(pairs):
        walked = [None, None]          # per side: (ops, mismatches, string ids) when the pair has to go wide
        too_big = False
        heads = []
        for line in pair:
            cols = line.strip().split("\t")
            nm, md, zs = _decodeTags(cols[11:])
            heads.append((cols, nm, md, zs))
        both = all(_passes(int(c[1]), nm) for c, nm, _, _ in heads)
        for side, (cols, nm, md, zs) in enumerate(heads):
            r = rec[2 * p + side]
            gid = index.gene_id.get(cols[2])
            if gid is None:
                raise ValueError(f"reference {cols[2]!r} is not a backbone of the index")
            r["pos0"] = int(cols[3]) - 1
            r["flag"] = int(cols[1]) & 0xFFFF
            r["ref"] = gid
            m = _NH.search(pair[side])
            r["nh"] = min(int(m.group(1)), 255) if m else 1
            r["nm"] = NM_ABSENT if nm is None else min(nm, 254)
            if not both:
                continue
            ops, mms, ins, clipped = _walkText(int(cols[3]) - 1, cols[5], cols[9], md, zs, table)
            if clipped:
                # a soft-clipped mate yields no variants (hisat2.py:682-683) but still counts for read
                # depth: keep its CIGAR when it fits (S ops included), else just the clip marker
                full = [(CIG_S if o == "S" else {"M": CIG_M, "I": CIG_I, "D": CIG_D}[o], int(n))
                        for n, o in _CIGAR.findall(cols[5])]
                narrow = len(full) <= MAX_CIG and all(n <= MAX_OPLEN for _, n in full)
                wide = len(full) <= WIDE_CIG and all(n < (1 << 28) for _, n in full)
                if not narrow and wide and spill is not None:
                    too_big = True            # its M runs count for the depth: the whole CIGAR goes wide
                elif not narrow:
                    full = [(CIG_S, 0)]
                walked[side] = (full, [], [])
                if narrow or not wide or spill is None:
                    r["n_cig"] = len(full)
                    for i, (o, n) in enumerate(full):
                        r["cig"][i] = (n << 4) | o
                continue
            n_ev = len(mms) + sum(1 for o, _ in ops if o in (CIG_I, CIG_D))
            walked[side] = (ops, mms, ins)
            if (len(ops) > MAX_CIG or len(mms) > MAX_MM or len(ins) > MAX_INS or n_ev > MAX_EV
                    or any(n > MAX_OPLEN for _, n in ops) or any(off > 0xFFFF for off, _ in mms)):
                fits_wide = (len(ops) <= WIDE_CIG and len(mms) <= WIDE_MM and len(ins) <= WIDE_INS and n_ev <= WIDE_EV
                             and all(n < (1 << 28) for _, n in ops) and all(off < (1 << 24) for off, _ in mms))
                if spill is None or not fits_wide:
                    raise PackCapacityError(f"record does not fit gk_mate{'_wide' if spill is not None else ''}: {cols[0]} {cols[5]}")
                too_big = True
                continue
            r["n_cig"], r["n_mm"], r["n_ins"] = len(ops), len(mms), len(ins)
            for i, (o, n) in enumerate(ops):
                r["cig"][i] = (n << 4) | o
            for i, (off, b) in enumerate(mms):
                r["mm"][i]["ref_off"] = off
                r["mm"][i]["base"] = b
            for i, s in enumerate(ins):
                r["ins"][i] = s
        if too_big:
            wide = np.zeros(2, dtype=MATE_WIDE_DTYPE)
            for side in range(2):
                r, x = rec[2 * p + side], wide[side]
                for f in ("pos0", "flag", "ref", "nh", "nm"):
                    x[f] = r[f]
                ops, mms, ins = walked[side]
                x["n_cig"], x["n_mm"], x["n_ins"] = len(ops), len(mms), len(ins)
                for i, (o, n) in enumerate(ops):
                    x["cig"][i] = (n << 4) | o
                for i, (off, b) in enumerate(mms):
                    x["mm"][i] = (off << 8) | b
                for i, sid in enumerate(ins):
                    x["ins"][i] = sid
                head = np.zeros(1, dtype=MATE_DTYPE)[0]
                for f in ("pos0", "flag", "ref", "nh", "nm"):
                    head[f] = r[f]
                head["n_cig"] = SPILLED
                head["ins"][0] = len(spill)
                rec[2 * p + side] = head
            spill.append((p, wide))
    return rec, table


def spillArrays(spill: list):
    """``packPairs``' spill list -> (wide records, pair indices) as ``Tabulation(spill=...)`` takes them; None when empty."""
    from ._lib import MATE_WIDE_DTYPE
    if not spill:
        return None
    return (np.concatenate([w for _, w in spill]).astype(MATE_WIDE_DTYPE, copy=False),
            np.array([p for p, _ in spill], dtype=np.int64))


_ERR_KINDS = {1: AssertionError, 2: NotImplementedError, 3: PackCapacityError, 4: ValueError}


def _withPacker(index: GkIndex, table: InsTable | None, feed, capacity: int | None = None):
    """Create a native packer, let ``feed(handle)`` push alignments into it, collect the result.
    ``capacity``: upper bound of the records the input can yield, when known beforehand -- the decoder then
    writes them straight into the (pinned) array that is returned instead of into storage of its own."""
    import ctypes as C
    from ._lib import check, lib, pinnedEmpty
    table = table or InsTable(index)
    genes = (C.c_char_p * len(index.genes))(*[g.encode() for g in index.genes])
    strings = (C.c_char_p * max(1, len(table.strings)))(*[s.encode() for s in table.strings])
    pk = C.c_void_p()
    check(lib().gk_packer_create(genes, len(index.genes), strings, len(table.strings), C.byref(pk)))
    try:
        rec = None
        if capacity:
            rec = pinnedEmpty(capacity, MATE_DTYPE)
            check(lib().gk_packer_set_output(pk, rec.ctypes.data, capacity))
        feed(pk)
        kind, line = C.c_int32(), C.c_int64()
        check(lib().gk_packer_error(pk, C.byref(kind), C.byref(line)))
        if kind.value:
            msg = lib().gk_last_error().decode(errors="replace")
            raise _ERR_KINDS.get(kind.value, ValueError)(msg)
        n_lines, n_reads, n_pairs, n_strange, n_str = (C.c_int64() for _ in range(5))
        check(lib().gk_packer_counts(pk, C.byref(n_lines), C.byref(n_reads), C.byref(n_pairs), C.byref(n_strange),
                                     C.byref(n_str)))
        if rec is None:
            rec = pinnedEmpty(2 * n_pairs.value, MATE_DTYPE)         # filled by gk_packer_records; pinned when a GPU is there
            dst = rec.ctypes.data if len(rec) else None
        else:
            rec, dst = rec[:2 * n_pairs.value], None                 # already in place
        pair_lines = np.empty((n_pairs.value, 2), dtype=np.int64)
        check(lib().gk_packer_records(pk, dst, pair_lines.ctypes.data if len(pair_lines) else None))
        for i in range(len(table.strings), n_str.value):
            table.intern(lib().gk_packer_string(pk, i).decode())
        counts = {"lines": n_lines.value, "reads": n_reads.value, "pairs": n_pairs.value, "strange": n_strange.value}
        n_spill = C.c_int64()
        check(lib().gk_packer_spilled(pk, C.byref(n_spill)))
        if n_spill.value:   # pairs that do not fit gk_mate travel in the wide format beside the records
            from ._lib import MATE_WIDE_DTYPE
            wide = np.empty(2 * n_spill.value, dtype=MATE_WIDE_DTYPE)
            which = np.empty(n_spill.value, dtype=np.int64)
            check(lib().gk_packer_spill_records(pk, wide.ctypes.data, which.ctypes.data))
            counts["spill"] = (wide, which)
        return rec, table, pair_lines, counts
    finally:
        lib().gk_packer_destroy(pk)


def packText(chunks, index: GkIndex, table: InsTable | None = None):
    """Name-collated SAM text (iterable of ``bytes`` chunks) -> (records, table, pair_lines, counts).

    Native pairing + decoding (``csrc/gk_sampack.cpp``): same records, same pairing order and the same
    exceptions as ``hisat2.pairLines`` + ``packPairs``, at about a million lines per second.
    ``pair_lines[p] = (line index of left, line index of right)`` for the emitted pairs."""
    from ._lib import lib

    def feed(pk):
        for chunk in chunks:
            if lib().gk_packer_feed(pk, chunk, len(chunk), 0):
                return
        lib().gk_packer_feed(pk, b"", 0, 1)   # flush the last (unterminated) line

    return _withPacker(index, table, feed)


def packBam(path: str, index: GkIndex, table: InsTable | None = None, name_sorted: bool = True):
    """``.bam`` file -> (records, table, pair_lines, counts) without going through SAM text.

    Same result as ``packText(bamChunks(path), ...)``: the records are inflated, name-collated and handed
    to the packer in binary form (``gk_bam_pack``); ``pair_lines`` index the collated record stream."""
    import ctypes as C
    from ._lib import check, lib
    h = C.c_void_p()
    check(lib().gk_bam_open(path.encode(), int(name_sorted), C.byref(h)))
    try:
        n_rec = C.c_int64()
        check(lib().gk_bam_info(h, C.byref(n_rec), None, None))
        return _withPacker(index, table, lambda pk: lib().gk_bam_pack(h, pk), capacity=n_rec.value)
    finally:
        lib().gk_bam_close(h)


def bamChunks(path: str, chunk_bytes: int = 1 << 24, name_sorted: bool = True):
    """SAM text of a ``.bam`` file in query-name order, as byte chunks of whole lines.

    Native replacement of ``samtools sort -n bam -O SAM`` (hisat2.py:103-110): BGZF inflate, BAM
    decode and name collation happen in ``csrc/gk_bamread.cpp``; no external tool is started."""
    import ctypes as C
    from ._lib import check, lib
    h = C.c_void_p()
    check(lib().gk_bam_open(path.encode(), int(name_sorted), C.byref(h)))
    try:
        buf = C.create_string_buffer(chunk_bytes)
        n = C.c_int64()
        while True:
            check(lib().gk_bam_next(h, buf, chunk_bytes, C.byref(n)))
            if n.value == 0:
                return
            yield buf.raw[:n.value]
    finally:
        lib().gk_bam_close(h)


def bamHeader(path: str) -> str:
    """``@`` header lines of a BAM file (``samtools view -H``, hisat2.py:113-118)."""
    import ctypes as C
    from ._lib import check, lib
    h = C.c_void_p()
    check(lib().gk_bam_open(path.encode(), 0, C.byref(h)))
    try:
        n = C.c_int64()
        check(lib().gk_bam_info(h, None, C.byref(n), None))
        buf = C.create_string_buffer(max(int(n.value), 1))
        check(lib().gk_bam_header(h, buf, n.value))
        return buf.raw[:n.value].decode()
    finally:
        lib().gk_bam_close(h)


def writeBam(path: str, sam_text: bytes | str, coordinate_sort: bool = True) -> None:
    """SAM text (header + alignment lines) -> BGZF BAM, natively (``gk_bam_write``): what
    ``samtools sort`` of the SAM does for ``utils.samtobam`` (hisat2.py:869-901); a coordinate-sorted file gets its ``.bai`` next to it."""
    from ._lib import check, lib
    if isinstance(sam_text, str):
        sam_text = sam_text.encode()
    check(lib().gk_bam_write(path.encode(), sam_text, len(sam_text), int(coordinate_sort)))


def readChunks(path: str, chunk_bytes: int = 1 << 24):
    """Byte chunks of a ``.sam`` / ``.sam.gz`` file, or of the name-collated text of a ``.bam``."""
    if path.endswith(".bam"):
        yield from bamChunks(path, chunk_bytes)
        return
    import gzip
    opener = gzip.open if path.endswith(".gz") else open
    with opener(path, "rb") as f:
        while True:
            b = f.read(chunk_bytes)
            if not b:
                return
            yield b


def packSample(sample, index: GkIndex, table: InsTable | None = None) -> tuple[np.ndarray, InsTable]:
    """Vectorised ``synth.SynthSample`` -> mate records in readPair emission order.

    The synthetic stream lists READ1 then READ2 of each name, so the pair is emitted when
    READ2 is seen: left = READ2 (mate index 2r+1), right = READ1 (2r).
    """
    from .synth import EV_SINGLE, EV_INS, EV_DEL
    table = table or InsTable(index)
    n = sample.n_pairs
    n_m = 2 * n
    gmap = np.array([index.gene_id[g] for g in sample.index.genes], dtype=np.uint8)
    ins_map = np.array([table.intern(s) for s in sample.ins_strings], dtype=np.uint32) \
        if sample.ins_strings else np.zeros(1, np.uint32)
    rec = np.zeros(n_m, dtype=MATE_DTYPE)
    # source mate for each record slot: slot 2r <- mate 2r+1, slot 2r+1 <- mate 2r
    src = np.arange(n_m) ^ 1
    pair = np.arange(n_m) >> 1
    rec["pos0"] = sample.pos0[src]
    rec["flag"] = sample.flag[src] | np.where(sample.pair_secondary[pair], 256, 0).astype(np.uint16)
    rec["ref"] = gmap[sample.pair_gene[pair]]
    rec["nh"] = sample.pair_nh[pair]
    nm = sample.nm[src]
    rec["nm"] = np.where(nm < 0, NM_ABSENT, np.minimum(nm, 254)).astype(np.uint8)
    flag = rec["flag"]
    ok = ((flag & 2) != 0) & (nm >= 0) & (nm <= 4)
    ok_pair = ok[0::2] & ok[1::2]
    walk = np.repeat(ok_pair, 2)
    clip = sample.clip[src]
    clipped = (clip.sum(axis=1) > 0) & walk
    cig = rec["cig"]
    # synthetic clipped mates carry no events: CIGAR = [head S] span M [tail S]
    span_all = sample.span[src].astype(np.int64)
    for side_mask, head, tail in ((clipped & (clip[:, 0] > 0) & (clip[:, 1] == 0), True, False),
                                  (clipped & (clip[:, 0] == 0) & (clip[:, 1] > 0), False, True),
                                  (clipped & (clip[:, 0] > 0) & (clip[:, 1] > 0), True, True)):
        idx = np.nonzero(side_mask)[0]
        if not len(idx):
            continue
        k = 0
        if head:
            cig[idx, k] = ((clip[idx, 0].astype(np.int64) << 4) | CIG_S).astype(np.uint16)
            k += 1
        cig[idx, k] = ((span_all[idx] << 4) | CIG_M).astype(np.uint16)
        k += 1
        if tail:
            cig[idx, k] = ((clip[idx, 1].astype(np.int64) << 4) | CIG_S).astype(np.uint16)
            k += 1
        rec["n_cig"][idx] = k
    todo = walk & ~clipped

    ev_cnt = (sample.ev_off[1:] - sample.ev_off[:-1])[src]
    if np.any(todo & (ev_cnt > MAX_EV)):
        raise PackCapacityError(f"a filter-passing synthetic mate has more than {MAX_EV} events")
    if np.any(todo):
        kinds_all = sample.ev_kind
        per_mate = lambda sel: np.bincount(  # noqa: E731
            np.repeat(np.arange(n_m), sample.ev_off[1:] - sample.ev_off[:-1])[sel], minlength=n_m)[src]
        if (np.any(todo & (per_mate(kinds_all == EV_SINGLE) > MAX_MM))
                or np.any(todo & (per_mate(kinds_all == EV_INS) > MAX_INS))
                or np.any(todo & (2 * per_mate(kinds_all != EV_SINGLE) + 1 > MAX_CIG))):
            raise PackCapacityError("a filter-passing synthetic mate does not fit gk_mate")
    # flatten the events of the records to pack
    slots = np.nonzero(todo)[0]
    cnt = ev_cnt[slots]
    tot = int(cnt.sum())
    span = sample.span[src]
    if tot:
        rep = np.repeat(np.arange(len(slots)), cnt)
        within = np.arange(tot) - np.repeat(np.cumsum(cnt) - cnt, cnt)
        e_idx = np.repeat(sample.ev_off[:-1][src][slots], cnt) + within
        kind = sample.ev_kind[e_idx]
        epos = sample.ev_pos[e_idx].astype(np.int64)
        eval_ = sample.ev_val[e_idx].astype(np.int64)
        slot_of = slots[rep]
        off = epos - rec["pos0"][slot_of].astype(np.int64)
        # mismatches
        is_s = kind == EV_SINGLE
        s_rank = _rankWithin(rep, is_s)
        ss = slot_of[is_s]
        mm = rec["mm"]
        mm["ref_off"][ss, s_rank[is_s]] = off[is_s]
        mm["base"][ss, s_rank[is_s]] = eval_[is_s]
        np.add.at(rec["n_mm"], ss, 1)
        # insertions
        is_i = kind == EV_INS
        i_rank = _rankWithin(rep, is_i)
        si = slot_of[is_i]
        rec["ins"][si, i_rank[is_i]] = ins_map[eval_[is_i]]
        np.add.at(rec["n_ins"], si, 1)
        # CIGAR: every indel event contributes M(gap) + I/D
        is_x = ~is_s
        x_rank = _rankWithin(rep, is_x)
        sx = slot_of[is_x]
        xoff = off[is_x]
        xk = kind[is_x]
        xlen = np.where(xk == EV_DEL, eval_[is_x], 0)
        if sample.ins_strings:
            ilen = np.array([len(s) for s in sample.ins_strings], dtype=np.int64)
            xlen_ins = np.where(xk == EV_INS, ilen[np.where(xk == EV_INS, eval_[is_x], 0)], 0)
        else:
            xlen_ins = np.zeros(len(xk), dtype=np.int64)
        # reference offset where the previous indel of the same record ended
        prev_end = np.zeros(len(xk), dtype=np.int64)
        same = np.zeros(len(xk), dtype=bool)
        if len(xk) > 1:
            same[1:] = sx[1:] == sx[:-1]
            prev_end[1:] = np.where(same[1:], (xoff + xlen)[:-1], 0)
        gap = xoff - prev_end
        xr = x_rank[is_x]
        cig[sx, 2 * xr] = ((gap << 4) | CIG_M).astype(np.uint16)
        oplen = np.where(xk == EV_DEL, xlen, xlen_ins)
        cig[sx, 2 * xr + 1] = ((oplen << 4) | np.where(xk == EV_DEL, CIG_D, CIG_I)).astype(np.uint16)
        n_x = np.bincount(sx, minlength=n_m)
        last_end = np.zeros(n_m, dtype=np.int64)
        # last indel end per record = max over its indels of (off + dellen)
        np.maximum.at(last_end, sx, xoff + xlen)
    else:
        n_x = np.zeros(n_m, dtype=np.int64)
        last_end = np.zeros(n_m, dtype=np.int64)
    fin = span.astype(np.int64) - last_end
    t = np.nonzero(todo)[0]
    cig[t, 2 * n_x[t]] = ((fin[t] << 4) | CIG_M).astype(np.uint16)
    rec["n_cig"][t] = (2 * n_x[t] + 1).astype(np.uint8)
    if np.any(fin[t] <= 0):
        raise PackCapacityError("synthetic mate ends in an indel")
    return rec, table


def _rankWithin(group: np.ndarray, sel: np.ndarray) -> np.ndarray:
    """For non-decreasing ``group`` ids: rank of each element among the selected ones of its group."""
    n = len(group)
    if not n:
        return np.zeros(0, dtype=np.int64)
    before = np.cumsum(sel) - sel               # selected elements before i, globally
    start = np.ones(n, dtype=bool)
    start[1:] = group[1:] != group[:-1]
    first = np.maximum.accumulate(np.where(start, np.arange(n), 0))
    return (before - before[first]).astype(np.int64)


class CompactMates:
    """Packed records in compact form on the host (``gk_mates_compact_host``): uint32 word offsets ``[n_mates + 1]``
    followed by the words the mates use -- ~30 bytes per mate instead of 128.  This is what crosses PCIe for a sample;
    ``toDevice`` queues the copy and the expansion into 128-byte records (``gk_mates_expand``) on a context's stream."""

    def __init__(self, records: np.ndarray, threads: int = 4):
        import ctypes as C
        from ._lib import MATE_DTYPE, check, lib, pinnedEmpty
        assert records.dtype == MATE_DTYPE
        records = np.ascontiguousarray(records)
        self.n_mates = len(records)
        n_words = C.c_int64()
        check(lib().gk_mates_compact_size(records.ctypes.data, self.n_mates, threads, C.byref(n_words)))
        self.words = pinnedEmpty(self.n_mates + 1 + n_words.value, np.uint32)
        check(lib().gk_mates_compact_host(records.ctypes.data, self.n_mates, threads, self.words.ctypes.data, len(self.words)))

    @property
    def nbytes(self) -> int:
        return int(self.words.nbytes)

    def toDevice(self, dev, wait: bool = False):
        """The 128-byte records in HBM (a device buffer of ``MATE_DTYPE``): compact words copied (queued on ``dev``'s
        stream; ``wait``: synchronised) and expanded there."""
        import ctypes as C
        from ._lib import MATE_DTYPE, check, lib
        compact = dev.alloc(len(self.words), np.uint32)
        check(lib().gk_h2d_async(dev.ctx, compact.ptr, C.c_void_p(self.words.ctypes.data), self.words.nbytes))
        mates = dev.alloc(self.n_mates, MATE_DTYPE)
        check(lib().gk_mates_expand(dev.ctx, compact.ptr, self.n_mates, mates.ptr))
        compact.free()          # the pool reuses the block in stream order: after the expansion
        if wait:
            dev.sync()
        return mates

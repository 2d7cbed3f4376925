"""Test infrastructure: a minimal SAM-text -> BAM (BGZF) encoder following the SAM/BAM specification
(sections 4.1 / 4.2), used to build inputs for the native reader (``csrc/gk_bamread.cpp``).
samtools is not available in this image, so this is the suite's own encoder."""
import struct
import zlib

_OPS = {c: i for i, c in enumerate("MIDNSHP=X")}
_BASES = {c: i for i, c in enumerate("=ACMGRSVTWYHKDBN")}


def _tag(field: str) -> bytes:
    tag, typ, val = field.split(":", 2)
    head = tag.encode()
    if typ == "i":
        v = int(val)
        for code, fmt, lo, hi in (("c", "<b", -128, 127), ("C", "<B", 0, 255), ("s", "<h", -32768, 32767),
                                  ("S", "<H", 0, 65535), ("i", "<i", -2 ** 31, 2 ** 31 - 1), ("I", "<I", 0, 2 ** 32 - 1)):
            if lo <= v <= hi:
                return head + code.encode() + struct.pack(fmt, v)
        raise ValueError(field)
    if typ == "A":
        return head + b"A" + val.encode()
    if typ == "f":
        return head + b"f" + struct.pack("<f", float(val))
    if typ in "ZH":
        return head + typ.encode() + val.encode() + b"\0"
    if typ == "B":
        sub, *items = val.split(",")
        fmt = {"c": "<b", "C": "<B", "s": "<h", "S": "<H", "i": "<i", "I": "<I", "f": "<f"}[sub]
        conv = float if sub == "f" else int
        return head + b"B" + sub.encode() + struct.pack("<I", len(items)) + b"".join(struct.pack(fmt, conv(x)) for x in items)
    raise ValueError(field)


def _reg2bin(beg: int, end: int) -> int:
    """UCSC bin of the 0-based half-open interval (SAM specification, section 5.3)."""
    end -= 1
    for shift, base in ((14, 4681), (17, 585), (20, 73), (23, 9), (26, 1)):
        if beg >> shift == end >> shift:
            return base + (beg >> shift)
    return 0


def _record(line: str, ref_id: dict[str, int]) -> bytes:
    f = line.split("\t")
    name, flag, rname, pos, mapq, cigar, rnext, pnext, tlen, seq, qual = f[:11]
    rid = ref_id.get(rname, -1)
    nid = rid if rnext == "=" else ref_id.get(rnext, -1)
    ops = []
    if cigar != "*":
        num = ""
        for ch in cigar:
            if ch.isdigit():
                num += ch
            else:
                ops.append(int(num) << 4 | _OPS[ch])
                num = ""
    l_seq = 0 if seq == "*" else len(seq)
    packed = bytearray((l_seq + 1) // 2)
    for i in range(l_seq):
        packed[i >> 1] |= _BASES[seq[i]] << (4 if i % 2 == 0 else 0)
    quals = b"\xff" * l_seq if qual == "*" else bytes(ord(c) - 33 for c in qual)
    ref_len = sum(o >> 4 for o in ops if (o & 15) in (0, 2, 3, 7, 8)) or 1
    body = struct.pack("<iiBBHHHIiii", rid, int(pos) - 1, len(name) + 1, int(mapq),
                       _reg2bin(int(pos) - 1, int(pos) - 1 + ref_len), len(ops), int(flag), l_seq,
                       nid, int(pnext) - 1, int(tlen))
    body += name.encode() + b"\0" + b"".join(struct.pack("<I", o) for o in ops) + bytes(packed) + quals
    body += b"".join(_tag(t) for t in f[11:])
    return struct.pack("<I", len(body)) + body


def _bgzf_block(data: bytes) -> bytes:
    comp = zlib.compressobj(6, zlib.DEFLATED, -15)
    deflated = comp.compress(data) + comp.flush()
    bsize = len(deflated) + 25
    return (b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", bsize) + deflated +
            struct.pack("<II", zlib.crc32(data), len(data)))


def bamBytes(lines: list[str]) -> bytes:
    """Uncompressed BAM stream of header ('@' lines) and records of ``lines``, in the given order."""
    header = [l for l in lines if l.startswith("@")]
    refs = []
    for l in header:
        if l.startswith("@SQ"):
            kv = dict(x.split(":", 1) for x in l.split("\t")[1:])
            refs.append((kv["SN"], int(kv["LN"])))
    ref_id = {n: i for i, (n, _) in enumerate(refs)}
    text = ("\n".join(header) + "\n").encode() if header else b""
    raw = b"BAM\x01" + struct.pack("<I", len(text)) + text + struct.pack("<I", len(refs))
    for n, ln in refs:
        raw += struct.pack("<I", len(n) + 1) + n.encode() + b"\0" + struct.pack("<I", ln)
    return raw + b"".join(_record(l, ref_id) for l in lines if l and not l.startswith("@"))


def samToBam(lines: list[str], path: str, block: int = 40000) -> None:
    """Write ``bamBytes(lines)`` to ``path`` as BGZF blocks of ``block`` bytes plus the EOF block."""
    raw = bamBytes(lines)
    with open(path, "wb") as f:
        for i in range(0, len(raw), block):
            f.write(_bgzf_block(raw[i:i + block]))
        f.write(_bgzf_block(b""))   # EOF marker block

#!/usr/bin/env python3
"""Runs inside a process started with LD_PRELOAD=libasan: loads the sanitized host library and drives the
BGZF / BAM / SAM parsers with valid inputs and with a loop of mutated ones (truncated streams, corrupted
deflate blocks, random byte flips in the uncompressed records, hostile length fields).  Any out-of-bounds read,
overflow or other undefined behaviour aborts the process; a clean run prints OK.

    python tests/asan/driver.py <library> <n_mutants> <seed>
"""
import ctypes as C
import os
import random
import struct
import sys
import tempfile
import zlib

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from bamwriter import _bgzf_block, bamBytes  # noqa: E402

lib = C.CDLL(sys.argv[1])
n_mutants, seed = int(sys.argv[2]), int(sys.argv[3])
lib.gk_last_error.restype = C.c_char_p
lib.gk_packer_string.restype = C.c_char_p

GENES = ["KIR2DL1*BACKBONE", "KIR3DL3*BACKBONE"]


def sam_lines(rng, n_pairs=120):
    """Name-collated SAM lines with the tags the decoder reads (hand-rolled: numpy is kept out of the
    sanitized process)."""
    lines = ["@HD\tVN:1.0\tSO:queryname"] + [f"@SQ\tSN:{g}\tLN:9000" for g in GENES]
    for q in range(n_pairs):
        g = rng.choice(GENES)
        a = rng.randrange(1, 8000)
        b = a + rng.randrange(100, 400)
        for flag, pos, mpos in ((99, a, b), (147, b, a)):
            kind = rng.randrange(6)
            seq = "".join(rng.choice("ACGT") for _ in range(150))
            if kind == 0:
                cigar, md, extra = "150M", "150", ""
            elif kind == 1:
                k = rng.randrange(1, 148)
                cigar, md, extra = "150M", f"{k}A{149 - k}", f"\tZs:Z:{k}|S|hv{rng.randrange(99)}"
            elif kind == 2:
                cigar, md, extra = "50M2D100M", "50^CA100", "\tZs:Z:50|D|hv7"
            elif kind == 3:
                cigar, md, extra = "10M2I138M", "148", ""
            elif kind == 4:
                cigar, md, extra = "5S145M", "145", ""
            else:
                cigar, md, extra = "70M80S", "70", ""
            nm = rng.choice(["NM:i:0", "NM:i:1", "NM:i:5", ""])
            tags = "\t".join(t for t in (nm, f"MD:Z:{md}", "NH:i:%d" % rng.choice([1, 1, 2]), "XS:A:+", "ZB:B:c,1,2,3") if t)
            lines.append(f"r{q}\t{flag}\t{g}\t{pos}\t60\t{cigar}\t=\t{mpos}\t{b - a}\t{seq}\t{'I' * 150}\t{tags}{extra}")
    return lines


def write_bgzf(raw: bytes, path: str, block=5000):
    with open(path, "wb") as f:
        for i in range(0, len(raw), block):
            f.write(_bgzf_block(raw[i:i + block]))
        f.write(_bgzf_block(b""))


def packed_pairs(pk, ext, blob: bytes, header: bytes, path: str) -> None:
    """What the host does with a packer's result: the records and the pairs' line numbers, the wide pairs, the compact
    form that crosses PCIe, the BAM rewrite of the pairs' lines and the "reads" array of a .variant.json."""
    n_lines, n_reads, n_pairs, n_strange, n_str = (C.c_int64() for _ in range(5))
    lib.gk_packer_counts(pk, C.byref(n_lines), C.byref(n_reads), C.byref(n_pairs), C.byref(n_strange), C.byref(n_str))
    kind, line = C.c_int32(), C.c_int64()
    lib.gk_packer_error(pk, C.byref(kind), C.byref(line))
    n = n_pairs.value
    own = None if ext is not None else (C.c_uint8 * (128 * max(2 * n, 1)))()
    pair_lines = (C.c_int64 * max(2 * n, 1))()
    if lib.gk_packer_records(pk, own, pair_lines if n else None):
        return
    n_spill = C.c_int64()
    lib.gk_packer_spilled(pk, C.byref(n_spill))
    if n_spill.value:
        wide = (C.c_uint8 * (2048 * 2 * n_spill.value))()
        which = (C.c_int64 * n_spill.value)()
        lib.gk_packer_spill_records(pk, wide, which)
    if not n:
        return
    rec = ext if ext is not None else own
    words = C.c_int64()
    if lib.gk_mates_compact_size(rec, C.c_int64(2 * n), 2, C.byref(words)) == 0:
        out = (C.c_uint32 * (2 * n + 1 + words.value))()
        lib.gk_mates_compact_host(rec, C.c_int64(2 * n), 2, out, C.c_int64(len(out)))
    # every second pair's two lines as a BAM file of their own; every pair as a row of the JSON hand-off
    sel = []
    for k in range(0, 2 * n, 4):
        sel += [pair_lines[k], pair_lines[k + 1]]
    pick = (C.c_int64 * len(sel))(*sel)
    lib.gk_bam_write_lines((path + ".lines.bam").encode(), header, C.c_int64(len(header)), blob, C.c_int64(len(blob)),
                           pick, C.c_int64(len(sel)), 1)
    src = (C.c_int64 * n)(*range(n))
    off = (C.c_uint32 * (4 * n + 1))(*[(k + 3) // 4 for k in range(4 * n + 1)])      # one id per row, in its lpv list
    ids = (C.c_uint32 * n)(*[k % 3 for k in range(n)])
    names = (C.c_char_p * 3)(b"KIR2DL1*v1", b"KIR2DL1*\"q\"", b"KIR3DL3*\xc3\xa9")
    genes = (C.c_char_p * len(GENES))(*[g.encode() for g in GENES])
    gene_of = (C.c_uint8 * n)(*[k % len(GENES) for k in range(n)])
    nh = (C.c_uint8 * n)(*[1 + k % 3 for k in range(n)])
    js = path + ".reads.json"
    open(js, "wb").close()
    lib.gk_json_write_reads(js.encode(), blob, C.c_int64(len(blob)), pair_lines, C.c_int64(n), src, C.c_int64(n), off, ids,
                            names, C.c_int64(3), genes, len(GENES), gene_of, nh)


def exercise(path: str) -> int:
    """Open + every walker of the reader on one file; returns the open() code."""
    names = (C.c_char_p * len(GENES))(*[g.encode() for g in GENES])
    for name_sorted in (1, 0):
        h = C.c_void_p()
        rc = lib.gk_bam_open(path.encode(), name_sorted, C.byref(h))
        if rc:
            return rc
        n_rec, n_hdr, n_ref = C.c_int64(), C.c_int64(), C.c_int32()
        lib.gk_bam_info(h, C.byref(n_rec), C.byref(n_hdr), C.byref(n_ref))
        hdr = C.create_string_buffer(max(n_hdr.value, 1))
        lib.gk_bam_header(h, hdr, n_hdr.value)
        buf = C.create_string_buffer(1 << 16)
        wrote = C.c_int64(1)
        text = []
        while wrote.value:
            if lib.gk_bam_next(h, buf, len(buf), C.byref(wrote)):
                break
            text.append(buf.raw[:wrote.value])
        lib.gk_bam_close(h)
        # binary packing of the same records
        h = C.c_void_p()
        if lib.gk_bam_open(path.encode(), name_sorted, C.byref(h)) == 0:
            pk = C.c_void_p()
            if lib.gk_packer_create(names, len(GENES), None, 0, C.byref(pk)) == 0:
                if name_sorted:   # records straight into a caller's buffer of exactly the promised size
                    out_buf = (C.c_uint8 * (128 * max(n_rec.value, 1)))()
                    lib.gk_packer_set_output(pk, out_buf, C.c_int64(n_rec.value))
                lib.gk_bam_pack(h, pk)
                packed_pairs(pk, out_buf if name_sorted else None, b"".join(text), hdr.raw[:n_hdr.value], path)
                lib.gk_packer_destroy(pk)
            off = (C.c_int64 * (n_ref.value + 1))(*[9000 * i for i in range(n_ref.value + 1)])
            counts = (C.c_uint32 * (9000 * max(n_ref.value, 1) * 6))()
            if 0 < n_ref.value <= 4:
                lib.gk_bam_pileup(h, off, n_ref.value, counts)
            lib.gk_bam_close(h)
        # the rendered text through the SAM packer and the BAM writer
        blob = b"".join(text)
        pk = C.c_void_p()
        if blob and lib.gk_packer_create(names, len(GENES), None, 0, C.byref(pk)) == 0:
            lib.gk_packer_feed(pk, blob, len(blob), 1)
            lib.gk_packer_destroy(pk)
        if blob:
            out = path + ".rewrite.bam"
            sam = hdr.raw[:n_hdr.value] + blob
            lib.gk_bam_write(out.encode(), sam, len(sam), 1)
    return 0


def main():
    rng = random.Random(seed)
    tmp = tempfile.mkdtemp(prefix="gk_asan_")
    raw = bamBytes(sam_lines(rng))
    good = os.path.join(tmp, "good.bam")
    write_bgzf(raw, good)
    assert exercise(good) == 0, lib.gk_last_error()
    opened = refused = 0
    first_rec = raw.index(b"r0\0") - 36 if b"r0\0" in raw else 64
    recs, o = [], first_rec                     # offset of every record's fixed fields in the uncompressed stream
    while o + 36 <= len(raw):
        size = struct.unpack_from("<I", raw, o)[0]
        if size < 32 or o + 4 + size > len(raw):
            break
        recs.append(o + 4)
        o += 4 + size
    for k in range(n_mutants):
        kind = rng.randrange(8)
        path = os.path.join(tmp, f"m{k % 8}.bam")
        if kind == 0:      # truncated uncompressed stream
            write_bgzf(raw[:rng.randrange(0, len(raw))], path)
        elif kind == 1:    # random byte flips in the records
            b = bytearray(raw)
            for _ in range(rng.randrange(1, 12)):
                b[rng.randrange(first_rec, len(b))] = rng.randrange(256)
            write_bgzf(bytes(b), path)
        elif kind == 2:    # hostile length fields (block_size, l_read_name, n_cigar_op, l_seq, B-array counts)
            b = bytearray(raw)
            at = rng.randrange(first_rec, len(b) - 4)
            b[at:at + 4] = struct.pack("<I", rng.choice([0, 1, 31, 32, 0x7FFFFFFF, 0xFFFFFFFF, rng.randrange(1 << 20)]))
            write_bgzf(bytes(b), path)
        elif kind == 3:    # corrupted compressed file
            with open(good, "rb") as f:
                z = bytearray(f.read())
            for _ in range(rng.randrange(1, 6)):
                z[rng.randrange(len(z))] = rng.randrange(256)
            with open(path, "wb") as f:
                f.write(z)
        elif kind == 4:    # truncated compressed file
            with open(good, "rb") as f:
                z = f.read()
            with open(path, "wb") as f:
                f.write(z[:rng.randrange(len(z))])
        elif kind == 7 and recs:   # a well-formed record whose CIGAR lies: I / S runs longer than the read, D / N runs of 2^28 bases
            b = bytearray(raw)
            for _ in range(rng.randrange(1, 4)):
                r = rng.choice(recs)
                l_name, n_cig = b[r + 8], struct.unpack_from("<H", b, r + 12)[0]
                if n_cig:
                    at = r + 32 + l_name + 4 * rng.randrange(n_cig)
                    run = rng.choice([151, 400, 1 << 20, (1 << 28) - 1])
                    b[at:at + 4] = struct.pack("<I", run << 4 | rng.choice([1, 4, 2, 3, 0]))
            write_bgzf(bytes(b), path)
        elif kind == 5:    # header damage: magic, l_text, n_ref, reference names
            b = bytearray(raw)
            at = rng.randrange(0, first_rec)
            b[at] = rng.randrange(256)
            write_bgzf(bytes(b), path)
        else:              # a plain gzip member instead of BGZF, or garbage
            with open(path, "wb") as f:
                f.write(zlib.compress(raw[:rng.randrange(len(raw))]) if rng.random() < 0.5 else os.urandom(rng.randrange(4000)))
        if exercise(path) == 0:
            opened += 1
        else:
            refused += 1
    # SAM text straight into the packer: damaged lines must end in an error code, never in a bad read
    names = (C.c_char_p * len(GENES))(*[g.encode() for g in GENES])
    all_lines = sam_lines(rng, 60)
    header_lines, lines = all_lines[:3], all_lines[3:]
    for k in range(n_mutants):
        bad = list(lines)
        for _ in range(rng.randrange(1, 5)):
            i = rng.randrange(len(bad))
            s = bytearray(bad[i].encode())
            if s:
                how = rng.randrange(4)
                if how == 3:      # numbers of more digits than a long holds: CIGAR run, POS / PNEXT, MD
                    f = s.split(b"\t")
                    if len(f) > 8:
                        big = b"9" * rng.randrange(19, 40)
                        which = rng.randrange(4)
                        if which == 0:
                            f[5] = big + b"M"
                        elif which == 1:
                            f[3] = big
                        elif which == 2:
                            f[7] = b"-" + big
                        else:
                            f = [x if not x.startswith(b"MD:Z:") else b"MD:Z:" + big for x in f]
                        s = bytearray(b"\t".join(f))
                elif how == 0:
                    s[rng.randrange(len(s))] = rng.choice(b"\t:0A^|,*\x00\xff")
                elif how == 1:
                    del s[rng.randrange(len(s)):]
                else:
                    s += b"\t" + rng.choice([b"Zs:Z:", b"Zs:Z:9|S", b"MD:Z:", b"NM:i:", b"NH:i:99999999999", b"Zs:Z:|||",
                                             b"NH:i:" + b"9" * 30, b"NM:i:" + b"9" * 30, b"Zs:Z:" + b"9" * 30 + b"|S|x"])
            bad[i] = s.decode("latin1")
        blob = ("\n".join(bad) + ("\n" if rng.random() < 0.8 else "")).encode("latin1")
        pk = C.c_void_p()
        assert lib.gk_packer_create(names, len(GENES), None, 0, C.byref(pk)) == 0
        cut = rng.randrange(len(blob) + 1)
        lib.gk_packer_feed(pk, blob[:cut], cut, 0)
        lib.gk_packer_feed(pk, blob[cut:], len(blob) - cut, 1)
        lib.gk_packer_destroy(pk)
        if k % 4 == 0:     # the same damaged lines through the BAM writer (its own number and CIGAR parsing)
            sam = ("\n".join(header_lines) + "\n").encode() + blob
            lib.gk_bam_write(os.path.join(tmp, "damaged.bam").encode(), sam, len(sam), k % 8 == 0)
    print(f"OK mutants {n_mutants}: {opened} opened and walked, {refused} refused")


if __name__ == "__main__":
    main()

"""CPU-side tests of the product's host logic (no GPU): decoding, pairing, packing, ranking helpers."""
import gzip
import json
import os
import re

import numpy as np
import pytest

from kir_graph_amd import _lib, packed, synth
from kir_graph_amd.hisat2 import pairLines, filterRead
from kir_graph_amd.index import GkIndex, getVariants, packKey
from kir_graph_amd.typing_mulit_allele import firstOccurrence, rankScore
from oracle import tabulate as ot

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    with gzip.open(os.path.join(GOLD, name), "rt") as f:
        return json.load(f)


def test_abi_library_exports_every_declared_symbol():
    """The C-ABI library loads on a machine without a GPU and exports all of include/graphkir_hip.h."""
    header = open(os.path.join(os.path.dirname(GOLD), "..", "include", "graphkir_hip.h")).read()
    declared = set(re.findall(r"^(?:int|const char\*)\s+(gk_\w+)\s*\(", header, flags=re.M))
    assert declared, "no declarations parsed"
    lib = _lib.lib()
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in the header but not exported"
    assert declared <= set(_lib.EXPORTED) | {"gk_abi_version"}, declared - set(_lib.EXPORTED)
    assert lib.gk_abi_version() == 1


def test_ctypes_signatures_take_as_many_arguments_as_the_header_declares():
    """Every prototype of include/graphkir_hip.h against the binding table of kir_graph_amd/_lib.py: same names, same
    number of arguments (a call with one argument too few reads a register of garbage: nothing else would say so)."""
    header = open(os.path.join(os.path.dirname(GOLD), "..", "include", "graphkir_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(?:int|const char\*|void)\s+(gk_\w+)\s*\(([^;{]*?)\)\s*;", header, flags=re.S):
        args = m.group(2).strip()
        protos[m.group(1)] = 0 if args in ("", "void") else len(re.split(r",(?![^()]*\))", args))
    assert len(protos) > 100
    assert set(protos) == set(_lib._SIGS), set(protos) ^ set(_lib._SIGS)
    wrong = {n: (protos[n], len(_lib._SIGS[n][1])) for n in protos if protos[n] != len(_lib._SIGS[n][1])}
    assert not wrong, wrong


def test_no_device_is_a_loud_error():
    """Without a HIP device the typing path must raise, not fall back."""
    if _lib.deviceCount() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(_lib.GkNoDevice):
        _lib.Device(0)


def test_mate_record_layout():
    assert _lib.MATE_DTYPE.itemsize == 128
    off = {k: v[1] for k, v in _lib.MATE_DTYPE.fields.items()}
    assert (off["pos0"], off["flag"], off["ref"], off["nh"], off["nm"], off["n_cig"], off["n_mm"], off["n_ins"],
            off["cig"], off["mm"], off["ins"]) == (0, 4, 6, 7, 8, 9, 10, 11, 12, 40, 104)


def test_pairing_matches_reference_fixture():
    t2 = load("t2_pairing.json.gz")
    lines = t2["lines"]
    got = [[lines.index(a), lines.index(b)] for a, b in pairLines(lines)]
    assert got == t2["pairs"]


def test_filter_read_matches_oracle():
    t1 = load("t1_tabulation.json.gz")
    for line in t1["lines"]:
        assert filterRead(line) == ot.passesFilter(line)


def _index_from(text, tmp_path):
    for ext, body in text.items():
        (tmp_path / f"ix.{ext}").write_text(body)
    return GkIndex.load(str(tmp_path / "ix"))


def test_decoder_raises_like_the_reference(tmp_path):
    """CIGAR/MD/Zs consistency errors of recordToRawVariant surface from packPairs."""
    t1 = load("t1_tabulation.json.gz")
    gidx = _index_from(t1["index"], tmp_path)
    good = t1["lines"][1]
    for rec in t1["records"]:
        if "error" not in rec:
            continue
        exc = {"AssertionError": AssertionError, "NotImplementedError": NotImplementedError}[rec["error"]]
        with pytest.raises(exc):
            packed.packPairs([(rec["line"], good)], gidx)


def test_key_order_equals_variant_order(tmp_path):
    t1 = load("t1_tabulation.json.gz")
    gidx = _index_from(t1["index"], tmp_path)
    assert list(gidx.key) == sorted(gidx.key)
    assert [v.id for v in gidx.variants] == [v[0] for v in t1["variants"] if not v[0].startswith("nv")]
    # window keys of getVariantsBoundary: single 'A' sorts after insertions, 'T' after A/C/G
    k = lambda pos, typ, val: packKey(0, pos, typ, val)  # noqa: E731
    assert k(100, 0, 5) < k(100, 1, ord("A")) < k(100, 1, ord("C")) < k(100, 1, ord("T")) < k(100, 2, 1)


def test_packers_agree_on_synthetic_sample():
    sidx = synth.makeIndex(seed=5, n_genes=3, var_range=(200, 300), allele_range=(10, 20))
    gidx = GkIndex.fromVariants(sidx.variants, genes=sidx.genes, exons=sidx.exons)
    sample = synth.makeSample(sidx, seed=9, n_pairs=1500)
    lines = synth.toSamLines(sample)
    a, _ = packed.packPairs(list(pairLines(lines)), gidx)
    b, _ = packed.packSample(sample, gidx)
    assert a.tobytes() == b.tobytes()
    # a fair share of pairs must survive the filter, and events of all three kinds must occur
    assert 0.7 * sample.n_pairs < sum(ot.passesFilter(l) for l in lines) / 2
    assert set(np.unique(sample.ev_kind)) == {0, 1, 2}


def test_index_files_round_trip(tmp_path):
    sidx = synth.makeIndex(seed=3, n_genes=2, var_range=(100, 150), allele_range=(8, 12))
    sidx.write(str(tmp_path / "ix"))
    back = getVariants(str(tmp_path / "ix"))
    assert [(v.id, v.pos, v.typ, v.val, v.allele, v.in_exon) for v in back] == \
           [(v.id, v.pos, v.typ, v.val, v.allele, v.in_exon) for v in sidx.variants]


def test_rank_equals_python_sorted():
    rng = np.random.default_rng(1)
    for _ in range(20):
        n, c = int(rng.integers(1, 300)), int(rng.integers(1, 5))
        value = -np.round(rng.random(n) * 5, 1)           # many ties
        sums = -np.round(rng.random((n, c)) * 3, 1)
        frac = rng.integers(0, 4, (n, c)) / 4
        keys = np.array([-value, -sums.sum(axis=1), np.abs(frac - frac.mean(axis=1, keepdims=True)).sum(axis=1)]).T
        want = sorted(range(n), key=lambda i: tuple(keys[i]))
        assert list(rankScore(value, sums, frac)) == want


def test_first_occurrence_equals_set_walk():
    rng = np.random.default_rng(2)
    for c in (1, 2, 3, 4):
        ids = rng.integers(0, 12, (500, c))
        seen, want = set(), []
        for row in ids:
            k = tuple(sorted(row))
            want.append(k not in seen)
            seen.add(k)
        assert list(firstOccurrence(ids, 12)) == want


def test_native_packer_equals_python_decoder(tmp_path):
    """csrc/gk_sampack.cpp == hisat2.pairLines + packed.packPairs: records, pairing order, string table."""
    sidx = synth.makeIndex(seed=5, n_genes=3, var_range=(200, 300), allele_range=(10, 20))
    gidx = GkIndex.fromVariants(sidx.variants, genes=sidx.genes, exons=sidx.exons)
    sample = synth.makeSample(sidx, seed=9, n_pairs=1500)
    lines = ["@HD\tVN:1.0\tSO:queryname", "[bam_sort_core] merging from 1 files"] + synth.toSamLines(sample)
    want, table_py = packed.packPairs(list(pairLines(lines)), gidx)
    text = ("\n".join(lines) + "\n").encode()
    for chunk in (len(text), 4096, 997):           # lines straddling chunk boundaries
        chunks = [text[i:i + chunk] for i in range(0, len(text), chunk)]
        got, table, pair_lines, counts = packed.packText(chunks, gidx)
        assert got.tobytes() == want.tobytes()
        assert table.strings == table_py.strings
        assert counts["pairs"] == len(want) // 2
        expect_pairs = [[lines.index(a), lines.index(b)] for a, b in pairLines(lines)]
        assert pair_lines.tolist() == expect_pairs
    # no trailing newline
    got, *_ = packed.packText([text.rstrip(b"\n")], gidx)
    assert got.tobytes() == want.tobytes()


def test_native_packer_raises_like_the_reference(tmp_path):
    t1 = load("t1_tabulation.json.gz")
    gidx = _index_from(t1["index"], tmp_path)
    good = t1["lines"][:2]
    got, _, pair_lines, _ = packed.packText([("\n".join(t1["lines"]) + "\n").encode()], gidx)
    want, _ = packed.packPairs(list(pairLines(t1["lines"])), gidx)
    assert got.tobytes() == want.tobytes()
    for rec in t1["records"]:
        if "error" not in rec:
            continue
        exc = {"AssertionError": AssertionError, "NotImplementedError": NotImplementedError}[rec["error"]]
        # make the bad record the first mate of a pair with a good second mate
        c = rec["line"].split("\t")
        mate = good[1].split("\t")
        c[0] = mate[0]; c[7] = mate[3]
        mate[7] = c[3]
        with pytest.raises(exc):
            packed.packText([("\t".join(c) + "\n" + "\t".join(mate) + "\n").encode()], gidx)


def test_first_of_sets_equals_stacked_table():
    from kir_graph_amd.typing_mulit_allele import firstOfSets
    rng = np.random.default_rng(4)
    for k in (1, 2, 3):
        prev = rng.integers(0, 20, (37, k))
        cols = rng.permutation(20)[:13]
        ids = np.hstack([np.repeat(prev, len(cols), axis=0), np.tile(cols, len(prev))[:, None]])
        assert list(firstOfSets(prev, cols, 20)) == list(firstOccurrence(ids, 20))


def test_site_verdict_equals_reference_loop():
    """Vectorised isHomozygous tail (typing_mulit_allele.py:835-857) vs the reference's per-position loop."""
    from collections import defaultdict
    from kir_graph_amd.typing_mulit_allele import AlleleTyping
    rng = np.random.default_rng(7)
    verdicts = set()
    for trial in range(300):
        n = int(rng.integers(1, 60))
        pos = rng.integers(0, 12, n)
        code = rng.integers(65, 70, n)
        neg = rng.integers(0, 2, n).astype(bool)
        cnt = np.where(rng.random(n) < 0.3, rng.integers(1, 5, n), rng.integers(1, 60, n))
        cn = int(rng.integers(2, 5))
        site = defaultdict(lambda: defaultdict(int))
        for p_, c_, ng, ct in zip(pos.tolist(), code.tolist(), neg.tolist(), cnt.tolist()):
            site[p_][f"*{c_}" if ng else f"{c_}"] += ct
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
        got = AlleleTyping._siteVerdict(pos, code, neg, cnt, cn)
        assert got == (hits == 0), trial
        verdicts.add(got)
        # the native form (gk_site_verdict, host code of the library) gives the same verdict
        import ctypes as C
        from kir_graph_amd._lib import check, lib
        p64, c64 = np.ascontiguousarray(pos, dtype=np.int64), np.ascontiguousarray(code, dtype=np.int64)
        n8, k64 = np.ascontiguousarray(neg, dtype=np.uint8), np.ascontiguousarray(cnt, dtype=np.int64)
        verdict = C.c_int32()
        check(lib().gk_site_verdict(p64.ctypes.data, c64.ctypes.data, n8.ctypes.data, k64.ctypes.data, n, cn,
                                    C.byref(verdict)))
        assert bool(verdict.value) == got, trial
    assert verdicts == {True, False}


def _name_order(a: str, b: str) -> int:
    """Query-name order restated in Python: digit runs compare as numbers, everything else by code."""
    i = j = 0
    while i < len(a) and j < len(b):
        if a[i].isdigit() and b[j].isdigit():
            while i < len(a) and a[i] == "0":
                i += 1
            while j < len(b) and b[j] == "0":
                j += 1
            ea, eb = i, j
            while ea < len(a) and a[ea].isdigit():
                ea += 1
            while eb < len(b) and b[eb].isdigit():
                eb += 1
            if ea - i != eb - j:
                return 1 if ea - i > eb - j else -1
            if a[i:ea] != b[j:eb]:
                return 1 if a[i:ea] > b[j:eb] else -1
            i, j = ea, eb
            if i != j:
                return 1 if i < j else -1
        else:
            if a[i] != b[j]:
                return 1 if a[i] > b[j] else -1
            i += 1
            j += 1
    return 1 if i < len(a) else -1 if j < len(b) else 0


def test_native_bam_reader_collates_and_renders_like_sam_text(tmp_path):
    """csrc/gk_bamread.cpp: BGZF inflate + BAM decode + query-name collation == the SAM text it was made of."""
    import functools
    from bamwriter import samToBam
    sidx = synth.makeIndex(seed=5, n_genes=3, var_range=(200, 300), allele_range=(10, 20))
    gidx = GkIndex.fromVariants(sidx.variants, genes=sidx.genes, exons=sidx.exons)
    sample = synth.makeSample(sidx, seed=9, n_pairs=1500)
    records = synth.toSamLines(sample)
    # rename some reads so that the order depends on the numeric rule, add tags of every type
    tricky = ["r2", "r10", "r1", "r01", "a", "r1a", "r1b10", "r1b9", "x007y3", "x7y12"]
    for k, name in enumerate(tricky):
        for m in (0, 1):
            f = records[2 * k + m].split("\t")
            f[0] = name
            records[2 * k + m] = "\t".join(f) + "\tXA:A:q\tXf:f:0.5\tXB:B:s,-3,7\tXH:H:1AE3\tXI:i:3000000000\tXS:i:-70000"
    header = ["@HD\tVN:1.0\tSO:coordinate"] + [f"@SQ\tSN:{g}\tLN:{len(sidx.backbone[g])}" for g in sidx.genes]
    by_coord = sorted(records, key=lambda l: (l.split("\t")[2], int(l.split("\t")[3])))   # a coordinate-sorted BAM
    path = str(tmp_path / "s.bam")
    samToBam(header + by_coord, path, block=30000)

    assert packed.bamHeader(path) == "\n".join(header) + "\n"
    got = b"".join(packed.bamChunks(path, chunk_bytes=1 << 16)).decode().split("\n")
    assert got[-1] == ""
    got = got[:-1]

    def order(x, y):
        fx, fy = x.split("\t", 2), y.split("\t", 2)
        return _name_order(fx[0], fy[0]) or (int(fx[1]) & 192) - (int(fy[1]) & 192)
    want = sorted(by_coord, key=functools.cmp_to_key(order))          # Python's sort is stable, like the reader's
    assert got == want
    names = [l.split("\t", 1)[0] for l in got]
    firsts = [n for i, n in enumerate(names) if (i == 0 or names[i - 1] != n) and n in tricky]
    assert firsts == ["a", "r01", "r1", "r1a", "r1b9", "r1b10", "r2", "r10", "x007y3", "x7y12"]
    # file order without collation
    raw = b"".join(packed.bamChunks(path, name_sorted=False)).decode().split("\n")[:-1]
    assert raw == by_coord
    # the packer sees the same stream either way
    a, _, pa, _ = packed.packText(packed.readChunks(path), gidx)
    b, _, pb, _ = packed.packText([("\n".join(want) + "\n").encode()], gidx)
    assert a.tobytes() == b.tobytes() and pa.tolist() == pb.tolist()
    # ... and so does the binary hand-over that skips the text
    c, table_c, pc, counts_c = packed.packBam(path, gidx)
    assert c.tobytes() == b.tobytes() and pc.tolist() == pb.tolist()
    assert table_c.strings == packed.packText([("\n".join(want) + "\n").encode()], gidx)[1].strings
    assert counts_c["pairs"] == len(b) // 2 and counts_c["lines"] == len(want)
    # hisat2.readBam yields the same lines
    from kir_graph_amd.hisat2 import readBam
    assert list(readBam(path)) == want
    # a plain gzip stream without BGZF block sizes takes the sequential inflate path
    from bamwriter import bamBytes
    plain = str(tmp_path / "plain.bam")
    with open(plain, "wb") as f:
        f.write(gzip.compress(bamBytes(header + by_coord)))
    assert b"".join(packed.bamChunks(plain)).decode().split("\n")[:-1] == want
    # one decoding thread or many: same text
    os.environ["GK_PACK_THREADS"] = "1"
    try:
        assert b"".join(packed.bamChunks(path)).decode().split("\n")[:-1] == want
    finally:
        del os.environ["GK_PACK_THREADS"]


def test_native_bam_reader_rejects_garbage(tmp_path):
    p = tmp_path / "bad.bam"
    p.write_bytes(b"not a bam file at all")
    with pytest.raises(_lib.GkError):
        list(packed.bamChunks(str(p)))


def test_native_bam_writer_matches_the_test_encoder_and_round_trips(tmp_path):
    """gk_bam_write: same bytes as the suite's Python encoder for the same record order; coordinate sort;
    what it writes is read back unchanged by the native reader."""
    from bamwriter import bamBytes
    sidx = synth.makeIndex(seed=5, n_genes=3, var_range=(200, 300), allele_range=(10, 20))
    sample = synth.makeSample(sidx, seed=9, n_pairs=800)
    records = synth.toSamLines(sample)
    records[0] += "\tXA:A:q\tXf:f:0.5\tXB:B:s,-3,7\tXH:H:1AE3\tXI:i:3000000000\tXS:i:-70000\tXC:i:200\tXc:i:-5"
    header = ["@HD\tVN:1.0\tSO:unsorted"] + [f"@SQ\tSN:{g}\tLN:{len(sidx.backbone[g])}" for g in sidx.genes]
    text = "\n".join(header + records) + "\n"
    # file order kept: byte-identical BAM stream to the independent Python encoder
    path = str(tmp_path / "a.bam")
    packed.writeBam(path, text, coordinate_sort=False)
    inflated = b"".join(gzip.decompress(b) for b in [open(path, "rb").read()])
    assert inflated == bamBytes(header + records)
    assert b"".join(packed.bamChunks(path, name_sorted=False)).decode().split("\n")[:-1] == records
    assert packed.bamHeader(path) == "\n".join(header) + "\n"
    # coordinate sort = stable sort by (reference order of the header, position)
    path2 = str(tmp_path / "b.bam")
    packed.writeBam(path2, text, coordinate_sort=True)
    order = {g: i for i, g in enumerate(sidx.genes)}
    want = sorted(records, key=lambda l: (order[l.split("\t")[2]], int(l.split("\t")[3])))
    assert b"".join(packed.bamChunks(path2, name_sorted=False)).decode().split("\n")[:-1] == want
    with pytest.raises(_lib.GkError):
        packed.writeBam(str(tmp_path / "c.bam"), "@HD\tVN:1.0\nbroken line\n")


def _pileup_case(tmp_path, seed=9, n_pairs=1500):
    """A coordinate-sorted BAM with varied base qualities, flags of every skipped kind and overlapping mates."""
    from bamwriter import samToBam
    sidx = synth.makeIndex(seed=5, n_genes=3, var_range=(200, 300), allele_range=(10, 20))
    gidx = GkIndex.fromVariants(sidx.variants, genes=sidx.genes, exons=sidx.exons)
    sample = synth.makeSample(sidx, seed=seed, n_pairs=n_pairs)
    records = synth.toSamLines(sample)
    rng = np.random.default_rng(3)
    for k in range(len(records)):
        f = records[k].split("\t")
        q = rng.integers(2, 41, size=len(f[9]))
        q[rng.random(len(q)) < 0.05] = 5                         # below mpileup's default --min-BQ
        f[10] = "".join(chr(33 + x) for x in q)
        pair = k // 2
        if pair % 97 == 0:
            f[1] = str(int(f[1]) | 256)                          # secondary: skipped
        elif pair % 89 == 0:
            f[1] = str(int(f[1]) | 1024)                         # duplicate: skipped
        elif pair % 83 == 0:
            f[1] = str(int(f[1]) & ~2)                           # not a proper pair: skipped
        elif pair % 7 == 0 and k % 2 == 1:                       # move the second mate onto the first one
            left = records[k - 1].split("\t")
            f[3] = str(int(left[3]) + int(rng.integers(0, 60)))
        records[k] = "\t".join(f)
    g0 = sidx.genes[0]
    hand = [   # equal and unequal overlapping bases, a deletion over a base, quality ties
        f"h1\t99\t{g0}\t101\t60\t6M\t=\t103\t8\tACGTAC\t" + "".join(chr(33 + q) for q in (30, 9, 9, 20, 20, 12)),
        f"h1\t147\t{g0}\t103\t60\t2M2D4M\t=\t101\t-8\tGAACTT\t" + "".join(chr(33 + q) for q in (30, 8, 20, 20, 20, 40)),
        f"h2\t99\t{g0}\t101\t60\t4M\t=\t101\t4\tAAAA\t" + "".join(chr(33 + q) for q in (20, 20, 10, 30)),
        f"h2\t147\t{g0}\t101\t60\t4M\t=\t101\t-4\tACAC\t" + "".join(chr(33 + q) for q in (20, 20, 10, 31)),
        f"h3\t0\t{g0}\t100\t60\t3M\t*\t0\t0\tNNA\t*",
    ]
    records += hand
    header = ["@HD\tVN:1.0\tSO:coordinate"] + [f"@SQ\tSN:{g}\tLN:{len(sidx.backbone[g])}" for g in sidx.genes]
    by_coord = sorted(records, key=lambda l: (l.split("\t")[2], int(l.split("\t")[3])))
    path = str(tmp_path / "p.bam")
    samToBam(header + by_coord, path, block=30000)
    return sidx, gidx, by_coord, path


def test_native_pileup_matches_the_restatement(tmp_path):
    """gk_bam_pileup + pileup.ratiosOf vs oracle/pileup.py on the same alignments (pileup.py:57-81)."""
    from kir_graph_amd import pileup
    from oracle import pileup as opile, tabulate as ot
    from kir_graph_amd.msa2hisat import Variant
    sidx, gidx, lines, path = _pileup_case(tmp_path)
    counts, pos0 = pileup.pileupCounts(path, gidx)
    got = pileup.ratiosOf(counts, pos0, gidx.genes)
    want = opile.pileupOfLines(lines)
    assert got.keys() == want.keys()
    for key in want:
        assert got[key] == want[key], key            # same integer counts -> the same float64 quotients
    g0 = sidx.genes[0]
    assert want[(g0, 99)]["all"] >= 1 and "N" in want[(g0, 99)]
    # the device table = hisat2.errorCorrection on every (position, read base)
    table = pileup.correctionTable(counts)
    assert table.shape == (int(pos0[-1]), 5)
    changed = 0
    for (ref, pos), p in want.items():
        at = int(pos0[gidx.gene_id[ref]]) + pos
        for j, b in enumerate("ACGTN"):
            v = ot.pileupCorrect(Variant(typ="single", ref=ref, pos=pos, val=b), want)
            expect = 0 if v.val == b else ord(v.val)
            assert table[at, j] == expect, (ref, pos, b, p)
            changed += expect != 0
    assert changed > 100
    assert not table[counts.sum(axis=1) == 0].any()


def test_bam_pairing_of_name_groups_matches_python(tmp_path):
    """gk_bam_pack's per-name pairing (2-record fast path, larger groups through the table) emits the pairs
    of hisat2.pairLines (readPair 248-270) on the collated stream: secondary copies, singletons, mates whose
    PNEXT does not point at each other, two READ1 records, groups cut at thread boundaries."""
    from bamwriter import samToBam
    from kir_graph_amd.hisat2 import readBam
    sidx = synth.makeIndex(seed=5, n_genes=3, var_range=(200, 300), allele_range=(10, 20))
    gidx = GkIndex.fromVariants(sidx.variants, genes=sidx.genes, exons=sidx.exons)
    sample = synth.makeSample(sidx, seed=12, n_pairs=6000)
    base = synth.toSamLines(sample)
    rng = np.random.default_rng(8)
    records = []
    for p in range(len(base) // 2):
        l, r = base[2 * p].split("\t"), base[2 * p + 1].split("\t")
        kind = int(rng.integers(0, 12))
        if kind == 0:        # a secondary copy of the pair elsewhere (its own key: flag 256)
            records += ["\t".join(l), "\t".join(r)]
            l2, r2 = list(l), list(r)
            shift = int(rng.integers(1, 40))
            l2[1], r2[1] = str(int(l[1]) | 256), str(int(r[1]) | 256)
            l2[3], r2[7] = str(int(l[3]) + shift), str(int(r[7]) + shift)
            records += ["\t".join(l2), "\t".join(r2)]
        elif kind == 1:      # mate lost
            records += ["\t".join(l)]
        elif kind == 2:      # PNEXT of the second mate points elsewhere: both keep waiting
            r[7] = str(int(r[7]) + 3)
            records += ["\t".join(l), "\t".join(r)]
        elif kind == 3:      # two READ1 records
            r[1] = str((int(r[1]) & ~128) | 64)
            records += ["\t".join(l), "\t".join(r)]
        elif kind == 4:      # mate on another reference
            r[6] = sidx.genes[0] if r[2] != sidx.genes[0] else sidx.genes[1]
            records += ["\t".join(l), "\t".join(r)]
        elif kind == 5:      # three copies of the same mate pair (same keys): the third record waits again
            records += ["\t".join(l), "\t".join(r), "\t".join(l)]
        else:
            records += ["\t".join(l), "\t".join(r)]
    header = ["@HD\tVN:1.0\tSO:coordinate"] + [f"@SQ\tSN:{g}\tLN:{len(sidx.backbone[g])}" for g in sidx.genes]
    by_coord = sorted(records, key=lambda x: (x.split("\t")[2], int(x.split("\t")[3])))
    path = str(tmp_path / "g.bam")
    samToBam(header + by_coord, path)
    collated = list(readBam(path))
    pairs = list(pairLines(collated))
    want, table_py = packed.packPairs(pairs, gidx)
    got, table_c, pair_lines, counts = packed.packBam(path, gidx)
    assert counts["pairs"] == len(pairs) and counts["lines"] == len(collated)
    assert got.tobytes() == want.tobytes()
    assert table_c.strings == table_py.strings
    where = {id(x): i for i, x in enumerate(collated)}
    assert pair_lines.tolist() == [[where[id(a)], where[id(b)]] for a, b in pairs]
    # the text packer agrees as well (its pairing runs over the whole stream)
    txt, _, pl_txt, counts_txt = packed.packText([("\n".join(collated) + "\n").encode()], gidx)
    assert txt.tobytes() == want.tobytes() and pl_txt.tolist() == pair_lines.tolist()
    assert counts_txt["strange"] == counts["strange"] > 0 and counts_txt["reads"] == counts["reads"]


def test_name_collation_on_random_name_shapes(tmp_path):
    """The keyed name sort (16-byte prefix keys, full comparison when they do not decide) orders any mix of
    name shapes like the plain comparator: long names, equal prefixes beyond the key, digit runs of more
    than 9 digits, leading zeros, digits against letters, names that are prefixes of others."""
    import functools
    from bamwriter import samToBam
    rng = np.random.default_rng(17)
    stems = ["r", "read", "A00123:45:HXXXXXXXX:1:", "x", "", "r0", "sample_long_prefix_over_sixteen_bytes/", "7", "00"]
    pieces = ["", "a", "b", ":", "_", "-", "0", "00", "000", "1", "9", "10", "007", "12345678901", "123456789", "99999999999999"]
    names = set()
    while len(names) < 17000:     # 34 k records and more: splitters, buckets and the per-bucket sorts take part
        k = int(rng.integers(1, 6))
        name = stems[int(rng.integers(len(stems)))] + "".join(
            pieces[int(rng.integers(len(pieces)))] if rng.random() < 0.5 else str(int(rng.integers(0, 3000))).zfill(int(rng.integers(0, 7)))
            for _ in range(k))
        if name and len(name) < 200:
            names.add(name)
    names = sorted(names)
    rng.shuffle(names)
    g = "KIR_TEST*BACKBONE"
    lines = []
    for i, name in enumerate(names):
        pos = 100 + i
        lines.append(f"{name}\t147\t{g}\t{pos + 200}\t60\t10M\t=\t{pos}\t-210\tACGTACGTAC\tIIIIIIIIII\tNM:i:0")   # READ2 first in the file
        lines.append(f"{name}\t99\t{g}\t{pos}\t60\t10M\t=\t{pos + 200}\t210\tACGTACGTAC\tIIIIIIIIII\tNM:i:0")
    for i, name in enumerate(names[::40]):     # more records of some names, far away in the file: ties keep the file's order
        for flag in (355, 403, 99, 147, 355):
            lines.append(f"{name}\t{flag}\t{g}\t{7 + i}\t60\t10M\t=\t{9 + i}\t0\tACGTACGTAC\tIIIIIIIIII\tNM:i:{flag % 3}")
    header = ["@HD\tVN:1.0\tSO:unsorted", f"@SQ\tSN:{g}\tLN:100000"]
    path = str(tmp_path / "n.bam")
    samToBam(header + lines, path, block=20000)
    got = b"".join(packed.bamChunks(path, chunk_bytes=1 << 15)).decode().split("\n")[:-1]

    def order(x, y):
        fx, fy = x.split("\t", 2), y.split("\t", 2)
        return _name_order(fx[0], fy[0]) or (int(fx[1]) & 192) - (int(fy[1]) & 192)
    want = sorted(lines, key=functools.cmp_to_key(order))
    assert got == want


def _reg2bins(beg: int, end: int) -> list[int]:
    """Bins that may hold records overlapping [beg, end) (SAM specification, section 5.3)."""
    end -= 1
    bins = [0]
    for shift, base in ((26, 1), (23, 9), (20, 73), (17, 585), (14, 4681)):
        bins += list(range(base + (beg >> shift), base + (end >> shift) + 1))
    return bins


def test_bam_index_finds_the_records_of_a_region(tmp_path):
    """gk_bam_write writes {path}.bai next to a coordinate-sorted BAM (utils.samtobam: samtools sort + index).
    The index is read back the way the specification prescribes -- candidate bins, linear-index floor, chunks
    as virtual offsets into the BGZF file -- and must lead to exactly the records that overlap a region."""
    import struct
    import zlib
    sidx = synth.makeIndex(seed=5, n_genes=3, var_range=(200, 300), allele_range=(10, 20), len_range=(30000, 40000))
    sample = synth.makeSample(sidx, seed=4, n_pairs=4000)
    lines = synth.toSamLines(sample)
    header = ["@HD\tVN:1.0\tSO:unsorted"] + [f"@SQ\tSN:{g}\tLN:{len(sidx.backbone[g])}" for g in sidx.genes]
    path = str(tmp_path / "x.bam")
    packed.writeBam(path, "\n".join(header + lines) + "\n")
    raw = open(path, "rb").read()
    bai = open(path + ".bai", "rb").read()

    # ---- parse the index
    assert bai[:4] == b"BAI\x01"
    o = 4
    (n_ref,) = struct.unpack_from("<i", bai, o); o += 4
    assert n_ref == len(sidx.genes)
    refs = []
    for _ in range(n_ref):
        (n_bin,) = struct.unpack_from("<i", bai, o); o += 4
        bins, meta = {}, None
        for _ in range(n_bin):
            b, n_chunk = struct.unpack_from("<Ii", bai, o); o += 8
            chunks = [struct.unpack_from("<QQ", bai, o + 16 * k) for k in range(n_chunk)]
            o += 16 * n_chunk
            if b == 37450:
                meta = chunks
            else:
                bins[b] = chunks
        (n_intv,) = struct.unpack_from("<i", bai, o); o += 4
        linear = list(struct.unpack_from(f"<{n_intv}Q", bai, o)); o += 8 * n_intv
        refs.append((bins, linear, meta))
    (n_no_coor,) = struct.unpack_from("<Q", bai, o); o += 8
    assert o == len(bai) and n_no_coor == 0

    # ---- BGZF random access: virtual offset = file offset of a block << 16 | offset inside the inflated block
    raw_at, stream, coff = {}, b"", 0
    while coff < len(raw):
        xlen = struct.unpack_from("<H", raw, coff + 10)[0]
        bsize = struct.unpack_from("<H", raw, coff + 16)[0] + 1      # the BC subfield is the only one written
        raw_at[coff] = len(stream)
        stream += zlib.decompress(raw[coff + 12 + xlen:coff + bsize - 8], -15)
        coff += bsize

    def position(v):
        return raw_at[v >> 16] + (v & 0xFFFF)

    def records(beg_v, end_v):
        at, stop = position(beg_v), position(end_v)
        while at < stop:
            size = struct.unpack_from("<i", stream, at)[0]
            yield stream[at + 4:at + 4 + size]
            at += 4 + size
        assert at == stop

    def span_of(rec):
        ref_id, pos, l_name, _, _, n_cig = struct.unpack_from("<iiBBHH", rec, 0)
        ops = struct.unpack_from(f"<{n_cig}I", rec, 32 + l_name)
        ref_len = sum(v >> 4 for v in ops if (v & 15) in (0, 2, 3, 7, 8)) or 1
        name = rec[32:32 + l_name - 1].decode()
        return ref_id, pos, pos + ref_len, name, struct.unpack_from("<H", rec, 14)[0]

    everything = {}
    for l in lines:
        f = l.split("\t")
        ref_len = sum(int(n) for n, op in __import__("re").findall(r"(\d+)([MDN=X])", f[5])) or 1
        everything.setdefault(sidx.genes.index(f[2]), []).append((int(f[3]) - 1, int(f[3]) - 1 + ref_len, f[0], int(f[1])))
    rng = np.random.default_rng(2)
    found = 0
    for _ in range(40):
        rid = int(rng.integers(0, n_ref))
        glen = len(sidx.backbone[sidx.genes[rid]])
        beg = int(rng.integers(0, glen - 10))
        end = min(glen, beg + int(rng.choice([1, 50, 700, 20000])))
        bins, linear, meta = refs[rid]
        floor = linear[beg >> 14] if (beg >> 14) < len(linear) else (linear[-1] if linear else 0)
        got = set()
        for b in _reg2bins(beg, end):
            for c_beg, c_end in bins.get(b, []):
                if c_end <= floor:
                    continue
                for rec in records(c_beg, c_end):
                    r, p0, p1, name, flag = span_of(rec)
                    assert r == rid
                    if p0 < end and p1 > beg:
                        got.add((p0, p1, name, flag))
        want = {t for t in everything.get(rid, []) if t[0] < end and t[1] > beg}
        assert got == want, (rid, beg, end, len(got), len(want))
        found += len(want)
    assert found > 500
    # metadata pseudo-bin: the counts of the reference
    for rid, (bins, linear, meta) in enumerate(refs):
        assert meta is not None and meta[1] == (len(everything[rid]), 0)


def test_native_json_reads_array_matches_json_dumps(tmp_path):
    """gk_json_write_reads == json.dumps([asdict(PairRead), ...]): escapes of quotes, backslashes, control
    characters, DEL and non-ASCII text (BMP and beyond), CRLF line ends, empty id lists, a last line
    without a line feed."""
    import ctypes as C
    import json
    from kir_graph_amd._lib import check, lib
    lines = ['r1\t99\tg*BACKBONE\t1\t60\t4M\t=\t9\t12\tACGT\tII"I\\\tXX:Z:tab\there',
             'r1\t147\tg*BACKBONE\t9\t60\t4M\t=\t1\t-12\tACGT\tIIII\tCO:Z:café 中 \U0001F9EC \x01\x7f',
             "r2\t99\tg*BACKBONE\t5\t60\t4M\t=\t7\t6\tACGT\tIIII\r",
             "r2\t147\tg*BACKBONE\t7\t60\t4M\t=\t5\t-6\tACGT\tIIII"]
    text = "\n".join(lines).encode()                      # no final line feed
    pair_lines = np.array([[1, 0], [3, 2]], dtype=np.int64)
    src = np.array([1, 0, 1], dtype=np.int64)             # rows may repeat / reorder pairs
    names = ["hv0", "hv1", 'n"v\\2', "nv3"]
    genes = ["g*BACKBONE", "KIRé"]
    off = np.array([0, 2, 2, 3, 4,   4, 4, 4, 4,   5, 6, 6, 8], dtype=np.uint32)   # row 1 has four empty lists
    ids = np.array([0, 1, 2, 3, 3, 0, 1, 2], dtype=np.uint32)
    gene_of = np.array([0, 1, 0], dtype=np.uint8)
    nh = np.array([1, 7, 255], dtype=np.uint8)
    path = str(tmp_path / "reads.json")
    c_names = (C.c_char_p * len(names))(*[n.encode() for n in names])
    c_genes = (C.c_char_p * len(genes))(*[g.encode() for g in genes])
    check(lib().gk_json_write_reads(path.encode(), text, len(text), pair_lines.ctypes.data, len(pair_lines),
                                    src.ctypes.data, len(src), off.ctypes.data, ids.ctypes.data, c_names, len(names),
                                    c_genes, len(genes), gene_of.ctypes.data, nh.ctypes.data))
    want = []
    for i in range(len(src)):
        o = off[4 * i:4 * i + 5]
        l, r = (lines[k].rstrip("\r") for k in pair_lines[src[i]])
        want.append({"l_sam": l, "r_sam": r, "multiple": int(nh[i]), "backbone": genes[gene_of[i]],
                     "lpv": [names[v] for v in ids[o[0]:o[1]]], "lnv": [names[v] for v in ids[o[2]:o[3]]],
                     "rpv": [names[v] for v in ids[o[1]:o[2]]], "rnv": [names[v] for v in ids[o[3]:o[4]]]})
    assert open(path).read() == json.dumps(want)
    assert json.loads(open(path).read()) == want


def test_pileup_ratios_and_correction_table_match_reference_fixture():
    """Product host logic of a21 (pileup.ratiosOf / correctionTable) against the REFERENCE's outputs on
    hand-written mpileup columns (tests/golden/t11_pileup.json.gz: getPileupBaseRatio 57-81, hisat2.errorCorrection
    609-654).  The per-position base counts are taken from the reference's own parse of the columns."""
    import gzip
    import json
    from kir_graph_amd import pileup as pp
    with gzip.open(os.path.join(os.path.dirname(__file__), "golden", "t11_pileup.json.gz"), "rt") as f:
        t11 = json.load(f)
    parsed = {c["bases"]: c["out"] for c in t11["parse"]}
    n_pos = 1000
    counts = np.zeros((n_pos, 6), dtype=np.uint32)
    for _, pos, depth, column in t11["rows"]:
        if depth == 0:
            continue
        bases = parsed.get(column)
        if bases is None:      # plain columns (letters only) parse to themselves
            assert not set(column) & set("$^+-")
            bases = column
        for b in bases.upper():
            counts[pos, pp.BASES.index(b)] += 1
    pos0 = np.array([0, n_pos], dtype=np.int64)
    got = pp.ratiosOf(counts, pos0, [t11["gene"]])
    want = {(t11["gene"], r["pos"]): r["entry"] for r in t11["ratio"]}
    assert set(got) == set(want)
    for key, entry in got.items():
        assert set(entry) == set(want[key])
        for k, v in entry.items():
            assert v == (want[key][k] if k == "all" else float.fromhex(want[key][k])), (key, k)   # same v / s bits
    table = pp.correctionTable(counts)
    for pos, val, fixed in t11["fixes"]:
        if ":" in val:
            continue                       # non-SNP variants never reach the table
        j = "ACGTN".index(val)
        shown = chr(table[pos, j]) if table[pos, j] else val
        assert shown == fixed, (pos, val, shown, fixed)


def test_correction_table_from_the_references_ratio_dictionary():
    """``extractVariant(pairs, variants, pileup=...)`` (hisat2.py:803-844) takes the dictionary of getPileupBaseRatio;
    ``pileup.correctionFromRatios`` turns it into the device table.  Against the REFERENCE's own ratio entries and
    its ``errorCorrection`` outputs (T11)."""
    import gzip
    import json
    from types import SimpleNamespace
    from kir_graph_amd import pileup as pp
    with gzip.open(os.path.join(os.path.dirname(__file__), "golden", "t11_pileup.json.gz"), "rt") as f:
        t11 = json.load(f)
    g = t11["gene"]
    ratios = {}
    for r in t11["ratio"]:          # the reference's entries, keys in the reference's order
        ratios[(g, r["pos"])] = {k: (r["entry"][k] if k == "all" else float.fromhex(r["entry"][k])) for k in r["order"]}
    ratios[("KIRX*BACKBONE", 3)] = {"A": 1.0, "all": 50}        # a backbone the index does not have: ignored
    index = SimpleNamespace(genes=["KIR0*BACKBONE", g], gene_id={"KIR0*BACKBONE": 0, g: 1})
    table, pos0 = pp.correctionFromRatios(ratios, index)
    assert pos0.tolist() == [0, 0, max(r["pos"] for r in t11["ratio"]) + 1] and table.shape == (pos0[-1], 5)
    seen = 0
    for pos, val, fixed in t11["fixes"]:
        if ":" in val or pos >= pos0[-1]:
            continue
        j = "ACGTN".index(val)
        shown = chr(table[pos0[1] + pos, j]) if table[pos0[1] + pos, j] else val
        assert shown == fixed, (pos, val, shown, fixed)
        seen += shown != val
    assert seen >= 5
    assert pp.correctionFromRatios({}, index)[0].shape == (0, 5)


def test_native_depth_tsv_equals_pandas(tmp_path):
    """gk_depth_write_tsv writes the text DataFrame.to_csv(sep='\\t', header=False, index=False) gives for the
    ``samtools depth -aa`` table (gene, 1-based position, depth)."""
    import ctypes as C
    import pandas as pd
    from kir_graph_amd._lib import check, lib
    rng = np.random.default_rng(3)
    genes = ["KIR2DL1*BACKBONE", "KIR2DL4*BACKBONE", "KIR3DL3*BACKBONE"]
    lens = np.array([1500, 1, 977])
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    depth = rng.integers(0, 5000, int(off[-1])).astype(np.uint32)
    depth[:3] = [0, 4294967295, 10]
    names = (C.c_char_p * len(genes))(*[g.encode() for g in genes])
    path = str(tmp_path / "d.tsv")
    check(lib().gk_depth_write_tsv(path.encode(), names, off.ctypes.data, len(genes), depth.ctypes.data))
    df = pd.DataFrame({"gene": np.repeat(np.array(genes, dtype=object), lens),
                       "pos": np.concatenate([np.arange(1, n + 1) for n in lens]), "depth": depth.astype(np.int64)})
    df.to_csv(tmp_path / "p.tsv", sep="\t", header=False, index=False)
    assert open(path).read() == (tmp_path / "p.tsv").read_text()


def test_bam_binary_hand_over_equals_text_path_on_awkward_records(tmp_path):
    """gk_bam_pack hands CIGAR / SEQ over as they lie in the record when every op is one of M I D N S and as text
    otherwise: either way the records, the string table and the exceptions are those of the SAM-text packer
    fed with the rendered lines (csrc/gk_sampack.cpp: TextSource / BamSource under one walk)."""
    from bamwriter import samToBam
    sidx = synth.makeIndex(seed=5, n_genes=3, var_range=(200, 300), allele_range=(10, 20))
    gidx = GkIndex.fromVariants(sidx.variants, genes=sidx.genes, exons=sidx.exons)
    sample = synth.makeSample(sidx, seed=21, n_pairs=400)
    records = synth.toSamLines(sample)
    header = ["@HD\tVN:1.0\tSO:queryname"] + [f"@SQ\tSN:{g}\tLN:{len(sidx.backbone[g])}" for g in sidx.genes]

    def edit(lines, k, **kw):
        """Pair k with fields of its first mate replaced (cigar / seq / md / zs / nm)."""
        f = lines[2 * k].split("\t")
        if "cigar" in kw:
            f[5] = kw["cigar"]
        if "seq" in kw:
            f[9] = kw["seq"]
            f[10] = "I" * len(f[9]) if f[9] != "*" else "*"
        for i in range(11, len(f)):
            for tag, key in (("MD:Z:", "md"), ("Zs:Z:", "zs"), ("NM:i:", "nm")):
                if f[i].startswith(tag) and key in kw:
                    f[i] = tag + str(kw[key])
        f = [c for c in f if not (c.startswith("Zs:Z:") and kw.get("zs") == "")]
        lines[2 * k] = "\t".join(f)

    def both_ways(lines, tag):
        path = str(tmp_path / f"{tag}.bam")
        samToBam(header + lines, path, block=20000)
        outcomes = []
        for run in (lambda: packed.packBam(path, gidx), lambda: packed.packText(packed.bamChunks(path), gidx)):
            try:
                rec, table, pl, counts = run()
                spill = counts.pop("spill", None)
                spill = (spill[0].tobytes(), spill[1].tolist()) if spill else None
                outcomes.append(("ok", rec.tobytes(), table.strings, pl.tolist(), counts, spill))
            except Exception as e:    # noqa: BLE001 -- the kind and the line are what is compared
                outcomes.append((type(e).__name__, str(e).split(":")[0].split()[-1], str(e).split(":", 1)[-1]))   # kind, line, why
        assert outcomes[0] == outcomes[1], tag
        return outcomes[0]

    n = 150
    # shapes that must come out as records: clips (short, long, too many ops to keep), long runs, many ops
    good = list(records)
    edit(good, 3, cigar=f"10S{n - 10}M", md=str(n - 10), zs="", nm=0)
    edit(good, 5, cigar=f"5S{n - 12}M7S", md=str(n - 12), zs="", nm=0)
    edit(good, 7, cigar="1S" + "1M1I" * 8 + f"{n - 17}M", md=str(n - 17 + 8), zs="", nm=0)      # 18 ops, clipped
    edit(good, 9, cigar="2M1I" * 6 + f"{n - 18}M", md=str(n - 18 + 12), zs="", nm=0)            # 13 ops, 6 strings
    edit(good, 11, cigar=f"{n}M", md="0A0C0G" + str(n - 3), seq="TTT" + "A" * (n - 3), zs="", nm=3)
    out = both_ways(good, "good")
    assert out[0] == "ok" and out[5][1] == [7]      # the clipped mate with 18 ops keeps its whole CIGAR in the wide format
    # shapes the reference raises on, or that only the text walk spells right: one file each, the bad pair in the middle
    cases = {
        "hard_clip": dict(cigar=f"5H{n}M", md=str(n), zs="", nm=0),          # H: NotImplementedError
        "pad": dict(cigar=f"70M2P{n - 70}M", md=str(n), zs="", nm=0),
        "eq_single_digit": dict(cigar=f"{n - 5}M5=", md=str(n - 5), zs="", nm=0),   # "5=" is skipped by (\d+)(\w)
        "eq_two_digits": dict(cigar=f"{n - 15}M15=", md=str(n - 15), zs="", nm=0),  # "15=" reads as op '5' of length 1
        "x_op": dict(cigar=f"{n - 5}M5X", md=str(n - 5), zs="", nm=0),
        "splice": dict(cigar=f"70M100N{n - 70}M", md=str(n), zs="", nm=0),
        "no_cigar": dict(cigar="*", md=str(n), zs="", nm=0),
        "no_seq": dict(cigar=f"{n}M", seq="*", md=str(n), zs="", nm=0),
        "short_cigar": dict(cigar=f"{n - 1}M", md=str(n - 1), zs="", nm=0),
        "md_short": dict(cigar=f"{n}M", md=str(n - 1), zs="", nm=0),
        "md_same_base": dict(cigar=f"{n}M", md="0A" + str(n - 1), seq="A" * n, zs="", nm=1),
        "md_no_deletion": dict(cigar=f"70M2D{n - 70}M", md=str(n), zs="", nm=0),
        "zs_unused": dict(cigar=f"{n}M", md=str(n), zs="3|S|hv1", nm=0),
        "too_many_ops": dict(cigar="2M1I" * 8 + f"{n - 24}M", md=str(n - 24 + 16), zs="", nm=0),   # 17 ops, not clipped
        "long_insert_run": dict(cigar=f"4M4096I{n - 4}M", md=str(n), zs="", nm=0),
    }
    kinds = {}
    for tag, kw in cases.items():
        lines = list(records)
        edit(lines, 200, **kw)
        out = both_ways(lines, tag)
        kinds[tag] = out[0]
        if tag == "too_many_ops":
            assert out[5][1] == [200]               # beyond gk_mate, inside gk_mate_wide
    assert kinds["splice"] == "NotImplementedError" and kinds["hard_clip"] == "NotImplementedError"
    assert kinds["md_short"] == "AssertionError" and kinds["too_many_ops"] == "ok"
    # beyond the wide format too: the capacity error stays
    lines = list(records)
    edit(lines, 200, cigar="1M1I" * 70 + f"{n - 140}M", md=str(n - 70), zs="", nm=0)     # 141 ops
    assert both_ways(lines, "beyond_wide")[0] == "PackCapacityError"


def test_record_index_is_the_same_however_the_stream_is_cut(tmp_path, monkeypatch):
    """gk_bam_open indexes the records on several threads from guessed starting points and keeps a guess only from
    where the true chain meets it: any number of segments gives the records of the one-segment walk, and a damaged
    block size is refused wherever the cuts fall."""
    import struct
    from bamwriter import bamBytes, _bgzf_block
    sidx = synth.makeIndex(seed=5, n_genes=3, var_range=(200, 300), allele_range=(10, 20))
    sample = synth.makeSample(sidx, seed=33, n_pairs=700)
    header = ["@HD\tVN:1.0\tSO:unsorted"] + [f"@SQ\tSN:{g}\tLN:{len(sidx.backbone[g])}" for g in sidx.genes]
    lines = synth.toSamLines(sample)
    # names and tags full of bytes that look like record heads: zeros, small integers, printable runs
    for k in range(0, len(lines), 7):
        lines[k] += "\tXB:B:I," + ",".join(str(v) for v in (40, 0, 0, 300, 1, 0, 36, 0))
    raw = bamBytes(header + lines)

    def write(data, name):
        path = str(tmp_path / name)
        with open(path, "wb") as f:
            for i in range(0, len(data), 30000):
                f.write(_bgzf_block(data[i:i + 30000]))
            f.write(_bgzf_block(b""))
        return path

    good = write(raw, "good.bam")
    want = None
    for n_seg in ("1", "2", "5", "64", "1000", "100000"):
        monkeypatch.setenv("GK_TEST_HOOKS", f"bam_segments={n_seg}")
        got = b"".join(packed.bamChunks(good, name_sorted=False))
        want = want or got
        assert got == want, n_seg
    assert want.decode().split("\n")[:-1] == lines
    # a block size that points past the end, one that is too small, one that lands inside the next record
    first = raw.index(b"r0000")
    starts = [first - 36]
    while starts[-1] < len(raw):
        starts.append(starts[-1] + 4 + struct.unpack_from("<I", raw, starts[-1])[0])
    assert starts[-1] == len(raw)
    for which, value in ((len(starts) // 2, 0x7FFFFFF0), (len(starts) // 3, 31), (len(starts) - 3, None)):
        at = starts[which]
        size = struct.unpack_from("<I", raw, at)[0]
        bad = bytearray(raw)
        bad[at:at + 4] = struct.pack("<I", value if value is not None else size + 9)
        path = write(bytes(bad), f"bad{which}.bam")
        for n_seg in ("1", "3", "64", "1000"):
            monkeypatch.setenv("GK_TEST_HOOKS", f"bam_segments={n_seg}")
            with pytest.raises(_lib.GkError, match="malformed BAM"):
                list(packed.bamChunks(path, name_sorted=False))


def test_wide_records_of_the_native_and_the_python_packer_agree():
    """Pairs that do not fit gk_mate: both packers file the same gk_mate_wide records under the same pair numbers
    and leave the same marker records (n_cig == GK_SPILLED, ins[0] = place in the wide array) behind."""
    sidx = synth.makeIndex(seed=5, n_genes=3, var_range=(200, 300), allele_range=(10, 20))
    gidx = GkIndex.fromVariants(sidx.variants, genes=sidx.genes, exons=sidx.exons)
    sample = synth.makeSample(sidx, seed=21, n_pairs=300)
    lines = synth.toSamLines(sample)
    n = 150

    def edit(k, side, cigar, md, seq=None):
        f = lines[2 * k + side].split("\t")
        f[5] = cigar
        if seq:
            f[9] = seq
        f = [c for c in f if not c.startswith("Zs:Z:")]
        f = ["MD:Z:" + md if c.startswith("MD:Z:") else "NM:i:0" if c.startswith("NM:i:") else c for c in f]
        lines[2 * k + side] = "\t".join(f)

    ok = [k for k in range(300) if all(int(l.split("\t")[1]) & 2 and "NM:i:" in l for l in lines[2 * k:2 * k + 2])]
    a, b, c, d = ok[3], ok[10], ok[11], ok[40]
    edit(a, 0, "2M1I" * 8 + f"{n - 24}M", str(n - 8))                          # 17 ops, 8 inserted strings
    edit(b, 1, "1S" + "1M1I" * 8 + f"{n - 17}M", str(n - 9))                   # clipped, 18 ops
    edit(c, 0, f"{n}M", "".join("0C" for _ in range(20)) + str(n - 20), "A" * n)   # 20 mismatches
    edit(c, 1, f"70M5000D{n - 70}M", "70^" + "A" * 5000 + str(n - 70))          # a 5000-base deletion
    edit(d, 0, "1M1I" * 60 + f"{n - 120}M", str(n - 60))                        # 121 ops, 60 strings: still wide
    text = ("\n".join(lines) + "\n").encode()
    rec, table, pair_lines, counts = packed.packText([text], gidx)
    wide, which = counts["spill"]
    spill = []
    rec_py, table_py = packed.packPairs(list(pairLines(lines)), gidx, spill=spill)
    wide_py, which_py = packed.spillArrays(spill)
    assert sorted(which.tolist()) == which.tolist() == which_py.tolist() == sorted([a, b, c, d])
    assert wide.tobytes() == wide_py.tobytes() and rec.tobytes() == rec_py.tobytes()
    assert table.strings == table_py.strings
    marked = np.flatnonzero(rec["n_cig"] == _lib.SPILLED)
    assert marked.tolist() == sorted([2 * k + s for k in (a, b, c, d) for s in (0, 1)])
    assert rec["ins"][marked, 0].tolist() == [0, 0, 1, 1, 2, 2, 3, 3]
    assert int(wide["n_ins"].max()) == 60 and int(wide["n_mm"].max()) == 20 and int((wide["cig"] >> 4).max()) == 5000
    # without a spill list the Python packer keeps its old limit
    with pytest.raises(packed.PackCapacityError):
        packed.packPairs(list(pairLines(lines)), gidx)


def test_site_verdict_from_tallies_equals_reference_loop():
    """gk_site_verdict_tallies: ordinals + positive / negative tallies -> the verdict of typing_mulit_allele.py:829-857,
    labels being str(val) (a one-base insertion prints like a substitution of that base, deletions are skipped)."""
    import ctypes as C
    from collections import defaultdict
    from kir_graph_amd.index import packKey
    rng = np.random.default_rng(11)
    strings = ["A", "C", "AC", "GGT", "T"]
    ins_code = np.array([ord(s) if len(s) == 1 else 256 + i for i, s in enumerate(strings)], dtype=np.int64)
    verdicts = set()
    for trial in range(300):
        n_keys = int(rng.integers(5, 80))
        keys, labels, is_del, positions = [], [], [], []
        for _ in range(n_keys):
            pos = int(rng.integers(0, 10))
            typ = int(rng.choice([0, 1, 1, 1, 2]))            # insertion, substitution, deletion ranks of the key
            if typ == 1:
                val = int(rng.choice([65, 67, 71, 84]))
                label = chr(val)
            elif typ == 0:
                val = int(rng.integers(len(strings)))
                label = strings[val]
            else:
                val = int(rng.integers(1, 9))
                label = str(val)
            keys.append(packKey(int(rng.integers(0, 3)), pos, typ, val))
            labels.append(label); is_del.append(typ == 2); positions.append(pos)
        keys = np.array(keys, dtype=np.uint64)
        n = int(rng.integers(1, n_keys + 1))
        ords = np.sort(rng.choice(n_keys, size=n, replace=False)).astype(np.int32)
        big = rng.random(n) < 0.7
        pcount = np.where(rng.random(n) < 0.6, np.where(big, rng.integers(4, 60, n), rng.integers(1, 4, n)), 0).astype(np.uint32)
        ncount = np.where(rng.random(n) < 0.6, np.where(big, rng.integers(4, 60, n), rng.integers(1, 4, n)), 0).astype(np.uint32)
        cn = int(rng.integers(2, 5))
        site = defaultdict(lambda: defaultdict(int))
        for o, p_, n_ in zip(ords.tolist(), pcount.tolist(), ncount.tolist()):
            if is_del[o]:
                continue
            if p_:
                site[positions[o]][labels[o]] += p_
            if n_:
                site[positions[o]][f"*{labels[o]}"] += n_
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
        verdict = C.c_int32()
        _lib.check(_lib.lib().gk_site_verdict_tallies(keys.ctypes.data, len(keys), ins_code.ctypes.data, len(ins_code),
                                                      ords.ctypes.data, pcount.ctypes.data, ncount.ctypes.data, n, cn,
                                                      C.byref(verdict)))
        assert bool(verdict.value) == (hits == 0), trial
        verdicts.add(bool(verdict.value))
    assert verdicts == {True, False}


def test_site_verdicts_of_all_genes_in_one_call():
    """gk_site_verdict_genes over tallies grouped by gene = gk_site_verdict_tallies group by group (0 where cn <= 1,
    where the question is not asked); empty groups included."""
    import ctypes as C
    from kir_graph_amd.index import packKey
    rng = np.random.default_rng(5)
    ins_code = np.array([65, 67, 258], dtype=np.int64)
    seen = set()
    for trial in range(60):
        n_keys = int(rng.integers(20, 200))
        keys = np.array([packKey(int(rng.integers(0, 6)), int(rng.integers(0, 12)), int(rng.choice([0, 1, 1, 2])),
                                 int(rng.integers(0, 3))) for _ in range(n_keys)], dtype=np.uint64)
        n_groups = int(rng.integers(1, 9))
        sizes = rng.integers(0, 40, n_groups)
        sizes[rng.integers(n_groups)] = 0
        bounds = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        n = int(bounds[-1])
        ords = rng.integers(0, n_keys, n).astype(np.int32)
        pcount = np.where(rng.random(n) < 0.7, rng.integers(1, 60, n), 0).astype(np.uint32)
        ncount = np.where(rng.random(n) < 0.7, rng.integers(1, 60, n), 0).astype(np.uint32)
        cn = rng.integers(0, 5, n_groups).astype(np.int32)
        got = np.full(n_groups, -1, dtype=np.int32)
        _lib.check(_lib.lib().gk_site_verdict_genes(keys.ctypes.data, n_keys, ins_code.ctypes.data, len(ins_code),
                                                    ords.ctypes.data, pcount.ctypes.data, ncount.ctypes.data,
                                                    bounds.ctypes.data, n_groups, cn.ctypes.data, got.ctypes.data))
        for g in range(n_groups):
            a, b = int(bounds[g]), int(bounds[g + 1])
            want = 0
            if cn[g] > 1:
                v = C.c_int32()
                o_, p_, q_ = (np.ascontiguousarray(x[a:b]) for x in (ords, pcount, ncount))
                _lib.check(_lib.lib().gk_site_verdict_tallies(keys.ctypes.data, n_keys, ins_code.ctypes.data, len(ins_code),
                                                              o_.ctypes.data, p_.ctypes.data, q_.ctypes.data, b - a,
                                                              int(cn[g]), C.byref(v)))
                want = v.value
            assert got[g] == want, (trial, g)
            seen.add((bool(cn[g] > 1), int(got[g])))
    assert {(True, 0), (True, 1), (False, 0)} <= seen


def test_name_collation_with_a_long_common_prefix(tmp_path):
    """Names of one sequencing run share instrument / run / flow cell / lane: the sort keys start after the prefix
    common to all names (cut back to the start of a digit run it would split), and the order is still the plain
    comparator's -- also when the common prefix ends in digits, in zeros, or is the whole of the shortest name."""
    import functools
    from bamwriter import samToBam
    rng = np.random.default_rng(23)
    g = "KIR_TEST*BACKBONE"
    header = ["@HD\tVN:1.0\tSO:unsorted", f"@SQ\tSN:{g}\tLN:100000"]

    def order(x, y):
        fx, fy = x.split("\t", 2), y.split("\t", 2)
        return _name_order(fx[0], fy[0]) or (int(fx[1]) & 192) - (int(fy[1]) & 192)

    families = {
        "illumina": lambda i: f"A00123:45:HXXXXXXXX:{1 + i % 2}:{1101 + int(rng.integers(0, 80))}:{int(rng.integers(1000, 30000))}:{int(rng.integers(1000, 99999))}",
        "digits_at_the_cut": lambda i: f"run7_read10{int(rng.integers(0, 500000))}",        # common prefix ends inside a number
        "zeros_at_the_cut": lambda i: f"s00{int(rng.integers(0, 300000)):0{int(rng.integers(1, 8))}d}x",
        "one_is_the_prefix": lambda i: "frag" if i == 0 else f"frag{'' if i % 3 else '.'}{int(rng.integers(0, 900000))}",
        # the buckets of the sample sort are ordered by radix passes over the first eight key bytes, then by the full
        # comparison inside runs of equal first halves: long runs (two groups that share eight characters each) ...
        "equal_first_halves": lambda i: f"{'aaaaaaaa' if i % 2 else 'bbbbbbbb'}{'x' * int(rng.integers(0, 3))}{int(rng.integers(0, 700000))}",
        # ... and keys that stop before their eighth byte (a digit run too long for a key ends it): such buckets keep
        # the comparison sort
        "keys_that_stop_early": lambda i: f"{'q' * int(rng.integers(0, 4))}{int(rng.integers(10**10, 10**12))}_{int(rng.integers(0, 50))}",
    }
    for tag, make in families.items():
        names = []
        seen = set()
        i = 0
        while len(names) < 9000:          # 18 k records: the bucket path of the sample sort
            nm = make(i)
            i += 1
            if nm not in seen:
                seen.add(nm)
                names.append(nm)
        lines = []
        for k, name in enumerate(names):
            pos = 100 + k
            lines.append(f"{name}\t147\t{g}\t{pos + 200}\t60\t10M\t=\t{pos}\t-210\tACGTACGTAC\tIIIIIIIIII\tNM:i:0")
            lines.append(f"{name}\t99\t{g}\t{pos}\t60\t10M\t=\t{pos + 200}\t210\tACGTACGTAC\tIIIIIIIIII\tNM:i:0")
        path = str(tmp_path / f"{tag}.bam")
        samToBam(header + lines, path, block=20000)
        got = b"".join(packed.bamChunks(path, chunk_bytes=1 << 15)).decode().split("\n")[:-1]
        assert got == sorted(lines, key=functools.cmp_to_key(order)), tag


def test_compact_records_on_the_host_hold_exactly_the_used_words():
    """gk_mates_compact_host (what crosses PCIe for a sample: ~24 bytes per mate instead of 128): word offsets + the
    header, CIGAR, mismatch and inserted-string words each mate uses, in record order, whatever the thread count."""
    import numpy as np
    from kir_graph_amd import packed, synth
    from kir_graph_amd.index import GkIndex
    sidx = synth.makeIndex(seed=2022, n_genes=3, var_range=(200, 300), allele_range=(12, 24))
    gidx = GkIndex.fromVariants(sidx.variants, genes=sidx.genes, exons=sidx.exons)
    sample = synth.makeSample(sidx, seed=1031, n_pairs=3000)
    rec, _ = packed.packSample(sample, gidx)
    w = rec.view(np.uint32).reshape(len(rec), 32)
    want, offsets = [], [0]
    for m in range(len(rec)):
        h = int(w[m, 2])
        n_cig, n_mm, n_ins = (h >> 8) & 255, (h >> 16) & 255, h >> 24
        used = np.concatenate([w[m, :3], w[m, 3:3 + (min(n_cig, 14) + 1) // 2], w[m, 10:10 + min(n_mm, 16)],
                               w[m, 26:26 + min(n_ins, 6)]])
        want.append(used)
        offsets.append(offsets[-1] + len(used))
    want = np.concatenate(want)
    assert int(rec["n_ins"].sum()) > 0 and int(rec["n_mm"].max()) > 1      # the case has insertions and mismatches
    for threads in (1, 3, 8):
        cm = packed.CompactMates(rec, threads=threads)
        assert np.array_equal(cm.words[:len(rec) + 1], np.array(offsets, dtype=np.uint32))
        assert np.array_equal(cm.words[len(rec) + 1:], want)
        assert cm.nbytes * 4 < rec.nbytes


def test_pipeline_defaults_follow_the_cores_of_the_rank(monkeypatch):
    """cohort.pipelineDefaults: sample lanes x searches at a time by the host cores a rank has (the affinity, shared by the
    ranks of the node unless the rank was pinned to cores of its own); whatever the user set stays."""
    from kir_graph_amd import cohort
    for name in ("GK_SAMPLE_LANES", "GK_SEARCH_SLOTS", "GK_WAIT_POLICY", "WORLD_SIZE", "LOCAL_WORLD_SIZE",
                 "GK_PRIVATE_CORES"):
        monkeypatch.delenv(name, raising=False)
    monkeypatch.setattr(os, "cpu_count", lambda: 64)
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(range(64)), raising=False)
    quota = cohort.hostCoresPerRank()                  # the container's cgroup quota may cap the 64
    assert 1 <= quota <= 64
    monkeypatch.setenv("WORLD_SIZE", "8")
    assert cohort.hostCoresPerRank() == max(1, quota // 8)
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "4")         # two nodes of four ranks
    assert cohort.hostCoresPerRank() == max(1, quota // 4)
    monkeypatch.delenv("GK_PRIVATE_CORES", raising=False)
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(range(16)), raising=False)   # a cpuset of the NODE: shared
    assert cohort.hostCoresPerRank() == max(1, min(16, quota) // 4)
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: {0, 1, 2}, raising=False)
    monkeypatch.setenv("GK_PRIVATE_CORES", "1")         # pinned by the launcher: cores of its own
    assert cohort.hostCoresPerRank() == min(3, quota)
    for cores, want in ((16, ("5", "3")), (6, ("5", "3")), (4, ("4", "2")), (3, ("4", "2")), (2, ("3", "2"))):
        monkeypatch.delenv("GK_SAMPLE_LANES", raising=False)
        monkeypatch.delenv("GK_SEARCH_SLOTS", raising=False)
        cohort.pipelineDefaults(cores=cores)
        assert (os.environ["GK_SAMPLE_LANES"], os.environ["GK_SEARCH_SLOTS"]) == want, cores
        assert os.environ["GK_WAIT_POLICY"] == "block"
    monkeypatch.setenv("GK_SAMPLE_LANES", "2")          # the user's choice stays
    monkeypatch.delenv("GK_SEARCH_SLOTS", raising=False)
    cohort.pipelineDefaults(cores=16)
    assert os.environ["GK_SAMPLE_LANES"] == "2" and os.environ["GK_SEARCH_SLOTS"] == "3"
    for name in ("GK_SAMPLE_LANES", "GK_SEARCH_SLOTS", "GK_WAIT_POLICY"):
        os.environ.pop(name, None)                      # set by pipelineDefaults itself, not through monkeypatch


def test_trace_and_test_hook_lists(monkeypatch):
    """GK_TRACE / GK_TEST_HOOKS are comma-separated lists (csrc/gk_env.h reads them the same way on the native side)."""
    from kir_graph_amd.utils import testHook, traceOn
    monkeypatch.delenv("GK_TRACE", raising=False)
    monkeypatch.delenv("GK_TEST_HOOKS", raising=False)
    assert not traceOn("pool") and testHook("two_walks") is None and testHook("setsum", "leaves") == "leaves"
    monkeypatch.setenv("GK_TRACE", "pool, search")
    assert traceOn("pool") and traceOn("search") and not traceOn("ingest") and not traceOn("poo")
    monkeypatch.setenv("GK_TEST_HOOKS", "two_walks,novel_log2cap=4,setsum=tiles")
    assert testHook("two_walks") == "" and testHook("novel_log2cap") == "4" and testHook("setsum") == "tiles"
    assert testHook("bam_segments") is None and testHook("novel") is None


def test_roofline_entry_is_the_largest_priced_kernel():
    """roofmodel.dominant: kernels are timed under their own names, so on a tiny sample a launch-bound kernel without a
    model can be the largest by time -- the entry is then the largest PRICED one, the other is named beside it."""
    from kir_graph_amd import roofmodel
    prof = {"scan_tiles": (10, 5.0), "compat_kernel": (2, 3.0), "tab_count": (1, 1.0), "em_sets_groups": (1, 0.5)}
    log = [("compat_kernel", 1000, 100, 60000.0, 8), ("compat_kernel", 1000, 100, 60000.0, 8), ("tab_count", 500, 480, 30000),
           ("em_sets_groups", 1000, 60000.0, 4000)]
    r = roofmodel.dominant(prof, log)
    assert r["kernel"] == "compat_kernel" and r["largest_kernel_by_time"] == "scan_tiles"
    assert r["bound"] == "valu" and r["achieved"] > 0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert "em_sets_groups" not in r.get("other_kernels", {}) or r["other_kernels"]["em_sets_groups"]["bound"] == "hbm"
    by, ops = roofmodel.emSetsLaunch(1000, 60000.0, 4000)
    assert by == 24.0 * 1000 + 4.0 * 60000.0 + 4.0 * 4000 and ops == 0.0
    step = roofmodel.stepRoofline(log, 1, 2.0)
    assert step["kernels"]["em_sets_groups"]["bytes_per_step"] == by


def test_samples_are_admitted_to_the_lanes_by_their_footprint():
    """cohort.SampleTyper: a sample starts when the samples in flight leave room for its estimated HBM footprint; one sample
    always may; a sample that is still a file (no tabulation to size) is not held back."""
    import threading
    import time
    from types import SimpleNamespace
    from kir_graph_amd import cohort

    def fake(n_valid, n_alleles=100, n_genes=2):
        tab = SimpleNamespace(n_valid=n_valid, n_ids=60 * n_valid, n_pairs=n_valid, mates=None, dev=None)
        return SimpleNamespace(tab=tab, index=SimpleNamespace(tables=[SimpleNamespace(n_allele=n_alleles)] * n_genes))

    big = fake(1_000_000)
    fp = cohort.sampleFootprint(big, "pv")
    assert fp > 9.2 * 100 * 1_000_000 and cohort.sampleFootprint(big, "exonfirst_1") > 1.9 * fp - 1e9
    assert cohort.sampleFootprint(big, "em") < fp / 3 and cohort.sampleFootprint("s.variant.npz", "pv") == 0      # lists dominate the EM
    typer = cohort.SampleTyper("pv", lanes=3)
    typer._budget = int(2.5 * fp)
    a, b = typer._admit(big), typer._admit(big)            # two fit
    assert a == b == fp and typer._inflight_bytes == 2 * fp
    got = []
    th = threading.Thread(target=lambda: got.append(typer._admit(big)))     # the third waits for room
    th.start()
    time.sleep(0.3)
    assert th.is_alive() and not got
    typer._release(a)
    th.join(5)
    assert got == [fp] and typer._inflight_bytes == 2 * fp
    assert typer._admit("sample.variant.npz") == 0                          # a file: nothing to size
    typer._release(b)
    typer._release(fp)
    typer._budget = 1                                                       # smaller than any sample: one at a time still runs
    assert typer._admit(big) == fp
    typer._release(fp)
    typer.close()


def test_pair_lists_of_the_pairing_threads_join_in_stream_order(tmp_path, monkeypatch):
    """gk_packer_feed_records on a name-collated stream large enough for several pairing threads (more than 4096 records
    each, 2^16 list entries or more): the pieces' pair lists are joined by one thread per piece into a block that is not
    zero-filled first -- records, line numbers of the pairs and counts are those of the one-piece text path
    (hisat2.py:248-270 pairs in stream order)."""
    sidx = synth.makeIndex(seed=5, n_genes=3, var_range=(200, 300), allele_range=(10, 20))
    gidx = GkIndex.fromVariants(sidx.variants, genes=sidx.genes, exons=sidx.exons)
    sample = synth.makeSample(sidx, seed=77, n_pairs=36000)
    lines = synth.toSamLines(sample)
    header = ["@HD\tVN:1.0\tSO:coordinate"] + [f"@SQ\tSN:{g}\tLN:{len(sidx.backbone[g])}" for g in sidx.genes]
    path = str(tmp_path / "big.bam")
    packed.writeBam(path, "\n".join(header + lines) + "\n")
    monkeypatch.setenv("GK_PACK_THREADS", "4")
    rec_b, table_b, pl_b, counts_b = packed.packBam(path, gidx)
    rec_t, table_t, pl_t, counts_t = packed.packText(packed.bamChunks(path), gidx)
    assert counts_b == counts_t and counts_b["pairs"] >= 32768
    assert rec_b.tobytes() == rec_t.tobytes()
    assert pl_b.tolist() == pl_t.tolist()
    assert table_b.strings == table_t.strings


def test_bgzf_blocks_inflate_the_same_with_zlib_as_with_libdeflate(tmp_path):
    """csrc/gk_bamread.cpp binds libdeflate by name when the image has its runtime library and falls back to zlib
    otherwise (test hook no_libdeflate): the name-collated text of a BAM file (hisat2.py:103-110) is the same either way.
    The choice is made once per process, so the zlib run is a child process."""
    import hashlib
    import subprocess
    import sys
    sidx = synth.makeIndex(seed=5, n_genes=3, var_range=(200, 300), allele_range=(10, 20))
    sample = synth.makeSample(sidx, seed=3, n_pairs=3000)
    header = ["@HD\tVN:1.0\tSO:coordinate"] + [f"@SQ\tSN:{g}\tLN:{len(sidx.backbone[g])}" for g in sidx.genes]
    path = str(tmp_path / "z.bam")
    packed.writeBam(path, "\n".join(header + synth.toSamLines(sample)) + "\n")
    here = hashlib.sha256(b"".join(packed.bamChunks(path))).hexdigest()
    code = ("import hashlib, sys; from kir_graph_amd import packed; "
            "print(hashlib.sha256(b''.join(packed.bamChunks(sys.argv[1]))).hexdigest())")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, "-c", code, path], capture_output=True, text=True, timeout=300, cwd=root,
                         env=dict(os.environ, GK_TEST_HOOKS="no_libdeflate", GK_TRACE="ingest", PYTHONPATH=root))
    assert res.returncode == 0, res.stderr[-2000:]
    assert res.stdout.strip() == here

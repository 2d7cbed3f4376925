"""GPU parity: device tabulation / compatibility / reductions vs the CPU oracle (bit-exact)."""
import numpy as np
import pytest

from kir_graph_amd import packed, synth
from kir_graph_amd.engine import DeviceIndex, Tabulation, LogTable, DeviceModel
from oracle import tabulate as ot, typing as oty

pytestmark = pytest.mark.gpu


def oracle_lists(sample, gidx):
    lines = synth.toSamLines(sample)
    return ot.tabulateLines(lines, gidx.variants)


def device_lists(tab):
    off, ids = tab.offsets(), tab.ids()
    names = tab.idNames()
    out = []
    for i in range(tab.n_valid):
        o = off[4 * i:4 * i + 5]
        out.append({"lpv": [names[v] for v in ids[o[0]:o[1]]], "rpv": [names[v] for v in ids[o[1]:o[2]]],
                    "lnv": [names[v] for v in ids[o[2]:o[3]]], "rnv": [names[v] for v in ids[o[3]:o[4]]]})
    return out


def test_tabulate_matches_oracle(device, small_case):
    sidx, gidx, sample = small_case
    rec, table = packed.packSample(sample, gidx)
    dindex = DeviceIndex(device, gidx)
    tab = Tabulation(dindex, rec)
    ref = oracle_lists(sample, gidx)
    assert tab.n_valid == len(ref["reads"])
    got = device_lists(tab)
    for i, (g, r) in enumerate(zip(got, ref["reads"])):
        for k in ("lpv", "rpv", "lnv", "rnv"):
            assert g[k] == r[k], (i, k)
    # novel variants: same ids, same order, same fields
    nov = tab.novelVariants(table.strings)
    ref_nov = [v for v in ref["variants"] if str(v.id).startswith("nv")]
    assert [(v.id, v.pos, v.typ, v.ref, v.val, v.length) for v in nov] == \
           [(v.id, v.pos, v.typ, v.ref, v.val, v.length) for v in ref_nov]
    genes = tab.pairGene()
    assert [gidx.genes[g] for g in genes] == [r["backbone"] for r in ref["reads"]]
    assert list(tab.pairNH()) == [r["multiple"] for r in ref["reads"]]
    tab.close()
    dindex.close()


def test_model_and_reductions_match_oracle(device, small_case):
    sidx, gidx, sample = small_case
    rec, table = packed.packSample(sample, gidx)
    dindex = DeviceIndex(device, gidx)
    tab = Tabulation(dindex, rec)
    ref = oracle_lists(sample, gidx)
    logs = LogTable(device)
    for g, gname in enumerate(gidx.genes):
        t = gidx.tables[g]
        reads = [dict(r) for r in ref["reads"] if r["backbone"] == gname and r["multiple"] == 1]
        variants = [v for v in ref["variants"] if v.ref == gname]
        om = oty.GeneModel(reads, variants, top_n=600, variant_correction=True)
        rows, n = tab.selectGene(g)
        vflag = device.alloc(tab.n_var_total, np.uint8).zero()
        tab.errorCorrection(rows, n, vflag)
        rows2, n2 = tab.selectNonEmpty(rows, n, vflag)
        assert n2 == om.readsNum(), gname
        if n2 == 0:
            continue
        dm = DeviceModel(tab, rows2, n2, vflag, t.vbeg, t.vend, dindex.masks[g], t.words, t.n_allele, logs,
                         want_miss=True)
        logs.resolve()
        dm.finishLog()
        assert om.id_to_allele == dict(enumerate(t.alleles))
        assert np.array_equal(dm.hostProbs(), om.probs), gname
        assert np.array_equal(dm.hostLogProbs(), om.log_probs), gname
        miss, nvar = oty.missTable(om.reads, om.variants, om.allele_to_id)
        assert np.array_equal(dm.miss.download().reshape(t.n_allele, n2).T, np.minimum(miss, 255))
        assert np.array_equal(dm.nvar.download(), nvar)
        cols = np.arange(t.n_allele)
        L = om.log_probs
        assert np.array_equal(dm.colsum(cols), L[:, cols].sum(axis=0)), gname
        prev = np.argsort(L.sum(axis=0))[::-1][:7][:, None]
        want = np.maximum(L[:, cols], L[:, prev[:, 0]].T[:, :, None]).sum(axis=1)
        assert np.array_equal(dm.maxsum(prev, cols), want), gname
        ids = np.stack([prev[:, 0], prev[::-1, 0]], axis=1)
        gathered = L[:, ids]
        best = gathered.max(axis=2)
        owns = np.equal(gathered, best[:, :, None])
        frac = (owns / owns.sum(axis=2)[:, :, None]).sum(axis=0) / L.shape[0]
        assert np.array_equal(dm.fraction(ids), frac), gname
        assert np.array_equal(dm.setmax(ids), best), gname
        dm.free()
    logs.close()
    tab.close()
    dindex.close()


def test_pileup_error_correction_matches_oracle(device, tmp_path):
    """error_correction=True (hisat2.py:925-928): mismatches that the pileup calls read errors are rewritten
    before the variant lookup; product (native pileup + corrected tabulation) vs oracle, ids and novel variants."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(__file__))
    from bamwriter import samToBam
    from kir_graph_amd import pileup
    from kir_graph_amd.hisat2 import extractVariantFromText
    from kir_graph_amd.index import GkIndex
    from oracle import pileup as opile
    sidx = synth.makeIndex(seed=21, n_genes=3, len_range=(4200, 6000), var_range=(200, 300), allele_range=(10, 20))
    gidx = GkIndex.fromVariants(sidx.variants, genes=sidx.genes, exons=sidx.exons)
    sample = synth.makeSample(sidx, seed=77, n_pairs=3000, err_rate=0.01)
    lines = synth.toSamLines(sample)
    header = ["@HD\tVN:1.0\tSO:coordinate"] + [f"@SQ\tSN:{g}\tLN:{len(sidx.backbone[g])}" for g in sidx.genes]
    by_coord = sorted(lines, key=lambda l: (l.split("\t")[2], int(l.split("\t")[3])))
    path = str(tmp_path / "s.bam")
    samToBam(header + by_coord, path)

    counts, pos0 = pileup.pileupCounts(path, gidx)
    table = pileup.correctionTable(counts)
    assert table.any()
    dindex = DeviceIndex(device, gidx)
    results = {}
    from kir_graph_amd.msa2hisat import Variant
    for name, corr in (("off", None), ("on", (table, pos0))):
        Variant.novel_id = 0
        data = extractVariantFromText(path, gidx, dev=device, dindex=dindex, correction=corr)
        results[name] = (device_lists(data.tab), data.tab.novelVariants(data.ins_strings))
        data.tab.close()
    from kir_graph_amd.hisat2 import readBam
    collated = list(readBam(path))
    for name, pile in (("off", None), ("on", opile.pileupOfLines(by_coord))):
        ref = ot.tabulateLines(collated, gidx.variants, pileup=pile)
        got, nov = results[name]
        assert len(got) == len(ref["reads"])
        for i, (g, r) in enumerate(zip(got, ref["reads"])):
            for k in ("lpv", "rpv", "lnv", "rnv"):
                assert g[k] == r[k], (name, i, k)
        ref_nov = [v for v in ref["variants"] if str(v.id).startswith("nv")]
        assert [(v.id, v.pos, v.typ, v.ref, v.val, v.length) for v in nov] == \
               [(v.id, v.pos, v.typ, v.ref, v.val, v.length) for v in ref_nov]
    # the text API with the reference's own argument (extractVariant(pairs, variants, pileup=...), hisat2.py:803-844):
    # the ratio dictionary becomes the device table (pileup.correctionFromRatios); same lists, same novel variants
    from kir_graph_amd.hisat2 import extractVariant, pairLines
    Variant.novel_id = 0
    data = extractVariant(pairLines(collated), gidx, dev=device, pileup=opile.pileupOfLines(by_coord))
    assert device_lists(data.tab) == results["on"][0]
    assert [(v.id, v.pos, v.typ, v.val) for v in data.tab.novelVariants(data.ins_strings)] == \
           [(v.id, v.pos, v.typ, v.val) for v in results["on"][1]]
    data.tab.close()
    Variant.novel_id = 0
    data = extractVariant(pairLines(collated), gidx, dev=device, pileup={})      # `if pileup:` -- empty means off
    assert device_lists(data.tab) == results["off"][0]
    data.tab.close()
    # the correction did something: distinct read errors at one site collapse onto the majority base
    assert len(results["on"][1]) < len(results["off"][1])
    assert results["on"][0] != results["off"][0]
    dindex.close()
    # the reference's entry point with its default error_correction=True writes the corrected lists
    import json
    from kir_graph_amd.hisat2 import extractVariantFromBam
    prefix = str(tmp_path / "idx")
    sidx.write(prefix)
    Variant.novel_id = 0
    extractVariantFromBam(prefix, path, str(tmp_path / "out"), error_correction=True, dev=device).tab.close()
    with open(tmp_path / "out.json") as f:
        saved = json.load(f)
    assert [{k: r[k] for k in ("lpv", "rpv", "lnv", "rnv")} for r in saved["reads"]] == results["on"][0]


def test_both_second_passes_give_the_same_tabulation(device, small_case, monkeypatch):
    """Pass 2 normally writes the lists from what pass 1 saved (events, window start, kept bits); windows of
    more than 256 variants take the second walk instead.  Both must give the same CSR -- also on an index
    dense enough to need the second walk."""
    from kir_graph_amd.index import GkIndex
    sidx, gidx, sample = small_case
    # an index with ~1 variant per base: every SNP site gets its other two alternative bases, carried by an
    # allele no sample draws, so a 300-base mate sees a window of ~300 variants but few events of its own
    from kir_graph_amd.msa2hisat import Variant
    dense = synth.makeIndex(seed=4, n_genes=1, len_range=(4000, 4200), var_range=(3500, 3600), allele_range=(20, 24),
                            frac_del=0.02, frac_ins=0.01)
    g = dense.genes[0]
    taken = {(v.pos, v.val) for v in dense.variants if v.typ == "single"}
    extra = []
    for v in dense.variants:
        if v.typ != "single":
            continue
        for alt in "ACGT":
            if alt != chr(dense.backbone[g][v.pos]) and (v.pos, alt) not in taken:
                taken.add((v.pos, alt))
                extra.append(Variant(pos=v.pos, typ="single", ref=g, val=alt, allele=[g.split("*")[0] + "*99999"],
                                     in_exon=v.in_exon))
    dense.variants = sorted(dense.variants + extra)
    for i, v in enumerate(dense.variants):
        v.id = f"hv{i}"
    dense_idx = GkIndex.fromVariants(dense.variants, genes=dense.genes, exons=dense.exons)
    dense_sample = synth.makeSample(dense, seed=3, n_pairs=600, read_len=300, frag_mean=700.0, frag_sd=20.0,
                                    err_rate=0.0005)
    for index, smp, needs_walk in ((gidx, sample, False), (dense_idx, dense_sample, True)):
        rec, _ = packed.packSample(smp, index)
        dindex = DeviceIndex(device, index)
        got = []
        for two_walks in (False, True):
            if two_walks:
                monkeypatch.setenv("GK_TEST_HOOKS", "two_walks")
            else:
                monkeypatch.delenv("GK_TEST_HOOKS", raising=False)
            tab = Tabulation(dindex, rec)
            got.append((tab.offsets().tobytes(), tab.ids().tobytes(), tab.n_novel))
            longest = int(np.diff(tab.offsets()).max())
            tab.close()
        assert got[0] == got[1]
        assert (longest > 256) == needs_walk
        if needs_walk:   # ... and the dense case agrees with the oracle
            ref = oracle_lists(smp, index)
            tab = Tabulation(dindex, rec)
            lists = device_lists(tab)
            assert [tuple(map(tuple, (g["lpv"], g["rpv"], g["lnv"], g["rnv"]))) for g in lists] == \
                   [tuple(map(tuple, (r["lpv"], r["rpv"], r["lnv"], r["rnv"]))) for r in ref["reads"]]
            tab.close()
        dindex.close()


def _rewrite(line, **kw):
    f = line.split("\t")
    if "cigar" in kw:
        f[5] = kw["cigar"]
    if "seq" in kw:
        f[9] = kw["seq"]
        f[10] = "I" * len(f[9])
    out = []
    for c in f:
        for tag, key in (("MD:Z:", "md"), ("Zs:Z:", "zs"), ("NM:i:", "nm")):
            if c.startswith(tag) and key in kw:
                c = None if kw[key] is None else tag + str(kw[key])
                break
        if c is not None:
            out.append(c)
    return "\t".join(out)


def test_pairs_beyond_the_128_byte_record_take_the_wide_format(device, tmp_path):
    """Mates with more mismatches / ops / events than gk_mate holds, an op longer than 4095 or a clipped CIGAR of
    many ops are kept as gk_mate_wide and walked by tab_count_wide / tab_emit_wide: lists, novel-variant ids in
    first-appearance order and read depth equal the oracle's, through the native packers (SAM text, BAM) and the
    Python one; the rest of the sample is unchanged by their presence."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(__file__))
    from bamwriter import samToBam
    from kir_graph_amd.hisat2 import extractVariant, extractVariantFromText, pairLines
    from kir_graph_amd.index import GkIndex
    from kir_graph_amd.msa2hisat import Variant
    from kir_graph_amd.samtools_utils import depthOfSample
    from oracle import depth as odepth
    sidx = synth.makeIndex(seed=21, n_genes=3, len_range=(9000, 11000), var_range=(200, 300), allele_range=(10, 20))
    gidx = GkIndex.fromVariants(sidx.variants, genes=sidx.genes, exons=sidx.exons)
    sample = synth.makeSample(sidx, seed=5, n_pairs=600)
    lines = synth.toSamLines(sample)
    rng = np.random.default_rng(3)

    def backbone_of(line):
        f = line.split("\t")
        bb = sidx.backbone[f[2]]
        return (bb if isinstance(bb, str) else bytes(bytearray(bb)).decode()), int(f[3]) - 1

    def many_mismatches(line, n_mm):
        """150M with n_mm substitutions against the backbone (mostly novel variants), MD to match."""
        bb, pos0 = backbone_of(line)
        ref = bb[pos0:pos0 + 150]
        assert len(ref) == 150
        at = sorted(rng.choice(150, size=n_mm, replace=False).tolist())
        seq, md, last = list(ref), "", 0
        for p in at:
            seq[p] = "ACGT"[("ACGT".index(ref[p]) + 1 + int(rng.integers(3))) % 4]
            md += f"{p - last}{ref[p]}"
            last = p + 1
        md += str(150 - last)
        return _rewrite(line, cigar="150M", seq="".join(seq), md=md, zs=None, nm=0)

    def plain(line, cigar, read_len):
        """A read that copies the backbone under `cigar` (M and D only): MD = match runs and deleted bases."""
        bb, pos0 = backbone_of(line)
        import re
        seq, md, cur, run = "", "", pos0, 0
        for n, op in re.findall(r"(\d+)([MDS])", cigar):
            n = int(n)
            if op == "M":
                seq += bb[cur:cur + n]; cur += n; run += n
            elif op == "D":
                md += f"{run}^{bb[cur:cur + n]}"; run = 0; cur += n
            else:
                seq += "A" * n
        md += str(run)
        assert len(seq) == read_len
        return _rewrite(line, cigar=cigar, seq=seq, md=md, zs=None, nm=0)

    def roomy(p, span=150):
        return all(l.split("\t")[5] == "150M" and ot.passesFilter(l) and
                   backbone_of(l)[1] + span + 150 < len(backbone_of(l)[0]) for l in lines[2 * p:2 * p + 2])
    picks = [p for p in range(600) if roomy(p)]
    far = [p for p in picks if roomy(p, 4500)]
    assert len(picks) >= 8 and far
    wide_pairs = {}
    # 30 mismatches on one mate, 20 on the other (more than 16 per mate / 22 events)
    k0 = picks[2]
    lines[2 * k0] = many_mismatches(lines[2 * k0], 30)
    lines[2 * k0 + 1] = many_mismatches(lines[2 * k0 + 1], 20)
    wide_pairs[k0] = "mismatches"
    # a novel deletion longer than 4095 (the mate yields no lists, its M runs count for the depth)
    k = next(p for p in far if p not in wide_pairs)
    lines[2 * k] = plain(lines[2 * k], "70M4200D80M", 150)
    wide_pairs[k] = "long deletion"
    # 19 ops of short novel deletions on one mate (more than 14)
    k2 = next(p for p in picks if p not in wide_pairs)
    lines[2 * k2 + 1] = plain(lines[2 * k2 + 1], "10M1D" * 9 + "60M", 150)
    wide_pairs[k2] = "many ops"
    # a clipped mate with 18 ops: no variants, but its CIGAR counts for the depth
    k3 = next(p for p in picks if p not in wide_pairs)
    lines[2 * k3] = plain(lines[2 * k3], "5S" + "8M1D" * 8 + "81M", 150)
    wide_pairs[k3] = "clipped, many ops"

    ref = ot.tabulateLines(lines, gidx.variants)
    ref_nov = [(v.id, v.pos, v.typ, v.ref, v.val, v.length) for v in ref["variants"] if str(v.id).startswith("nv")]
    gene_len = {g: len(sidx.backbone[g]) for g in sidx.genes}
    kept = [(l, r, ot.nhOf(l)) for l, r in ot.pairMates(lines) if ot.passesFilter(l) and ot.passesFilter(r)]
    want_depth = odepth.depthFromPairs(kept, gene_len)

    header = ["@HD\tVN:1.0\tSO:queryname"] + [f"@SQ\tSN:{g}\tLN:{len(sidx.backbone[g])}" for g in sidx.genes]
    sam = str(tmp_path / "s.sam")
    with open(sam, "w") as f:
        f.write("\n".join(header + lines) + "\n")
    bam = str(tmp_path / "s.bam")
    samToBam(header + lines, bam)
    dindex = DeviceIndex(device, gidx)
    spilled = None
    for how in ("sam", "bam", "python"):
        Variant.novel_id = 0
        if how == "python":
            data = extractVariant(pairLines(lines), gidx, dev=device, dindex=dindex)
        else:
            data = extractVariantFromText(sam if how == "sam" else bam, gidx, dev=device, dindex=dindex, keep_text=False)
        n_spill = int(data.tab.mates.download()["n_cig"][::2].tolist().count(0xFF)) if hasattr(data.tab.mates, "download") else None
        got = device_lists(data.tab)
        assert len(got) == len(ref["reads"]), how
        for i, (g, r) in enumerate(zip(got, ref["reads"])):
            for key in ("lpv", "rpv", "lnv", "rnv"):
                assert g[key] == r[key], (how, i, key)
        nov = data.tab.novelVariants(data.ins_strings)
        assert [(v.id, v.pos, v.typ, v.ref, v.val, v.length) for v in nov] == ref_nov, how
        df = depthOfSample(data, gene_len)
        for g in sidx.genes:
            assert np.array_equal(df[df["gene"] == g]["depth"].to_numpy(), want_depth[g]), (how, g)
        if n_spill is not None:
            assert n_spill == len(wide_pairs), how
        data.tab.close()
    # the 30-mismatch mate really carries more positives than a gk_mate could
    longest = max(len(r["lpv"]) + len(r["rpv"]) for r in ref["reads"])
    assert longest > 22
    dindex.close()


def test_a_novel_table_that_fills_up_is_retried_larger(device, small_case, monkeypatch):
    """The hash table of the novel variants starts at a slot per mate; a sample that fills half of it (or whose keys
    find no slot within the probe bound) is tabulated again with a table eight times the size.  Forced here with a
    16-slot table: same lists, same novel variants in the same order as with the default size."""
    sidx, gidx, sample = small_case
    rec, table = packed.packSample(sample, gidx)
    dindex = DeviceIndex(device, gidx)
    want_tab = Tabulation(dindex, rec)
    want = (want_tab.offsets().tolist(), want_tab.ids().tolist(), want_tab.novelKeys().tolist())
    assert want_tab.n_novel > 8          # more than half of 16 slots: the first attempts must fail
    want_tab.close()
    monkeypatch.setenv("GK_TEST_HOOKS", "novel_log2cap=4")
    tab = Tabulation(dindex, rec)
    assert (tab.offsets().tolist(), tab.ids().tolist(), tab.novelKeys().tolist()) == want
    tab.close()
    dindex.close()

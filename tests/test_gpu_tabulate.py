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
                monkeypatch.setenv("GK_TAB_TWO_WALKS", "1")
            else:
                monkeypatch.delenv("GK_TAB_TWO_WALKS", raising=False)
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

"""GPU parity: full / exon-first / EM strategies through the drop-in API vs the CPU oracle."""
import copy

import numpy as np
import pytest

from kir_graph_amd import packed, synth
from kir_graph_amd.engine import DeviceIndex, Tabulation
from kir_graph_amd.hisat2 import SampleData
from kir_graph_amd.kir_typing import selectKirTypingModel
from oracle import tabulate as ot, typing as oty, em as oem

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tabulated(device, small_case):
    sidx, gidx, sample = small_case
    rec, table = packed.packSample(sample, gidx)
    dindex = DeviceIndex(device, gidx)
    tab = Tabulation(dindex, rec)
    data = SampleData(tab, gidx, tab.novelVariants(table.strings))
    ref = ot.tabulateLines(synth.toSamLines(sample), gidx.variants)
    return data, ref, sample


def same_result(a, b):
    assert a.n == b.n
    for f in ("value", "value_sum_indv", "allele_id", "fraction", "fraction_uniq"):
        x, y = np.asarray(getattr(a, f)), np.asarray(getattr(b, f))
        assert x.shape == y.shape, (f, x.shape, y.shape)
        assert np.array_equal(x, y), f
    assert a.allele_name == b.allele_name
    assert np.array_equal(np.asarray(a.allele_prob), b.allele_prob)


@pytest.mark.parametrize("method", ["full", "exonfirst_1", "exonfirst_0.9"])
def test_likelihood_strategies(tabulated, method):
    data, ref, sample = tabulated
    gpu = selectKirTypingModel(method, data, top_n=600, variant_correction=True)
    got = gpu.typing(sample.gene_cn)
    cpu = oty.makeTyper(method, copy.deepcopy(ref), top_n=600, variant_correction=True)
    want = cpu.typing(sample.gene_cn)
    assert got == want
    for gene, steps in cpu.results.items():
        assert len(gpu._result[gene]) == len(steps), gene
        for a, b in zip(gpu._result[gene], steps):
            same_result(a, b)
    assert gpu.getAllPossibleTyping() == cpu.allPossible()


def test_em_strategy(tabulated):
    data, ref, sample = tabulated
    gpu = selectKirTypingModel("report", data)
    got = gpu.typing(sample.gene_cn)
    cpu = oem.ReportTyper(copy.deepcopy(ref))
    want = cpu.typing(sample.gene_cn)
    for gene, report in cpu.results.items():
        a = {r.allele: (r.count, r.prob) for r in gpu._result[gene]}
        b = {r["allele"]: (r["count"], r["prob"]) for r in report}
        assert a.keys() == b.keys()
        for k in a:
            assert a[k][0] == b[k][0]
            assert a[k][1] == pytest.approx(b[k][1], rel=1e-5, abs=1e-9)   # tolerance of BASELINE.json north_star
    assert sorted(got[0]) == sorted(want[0])
    assert got[1] == want[1]


def test_em_distinct_sets_equal_numpy_unique(tabulated):
    """gk_em_distinct (device grouping of the candidate bit sets) == np.unique(axis=0) of the per-read sets."""
    from kir_graph_amd.kir_typing import _GeneView
    from kir_graph_amd.typing_em import candidateSets, candidateSetsDistinct, distinctSets
    data, ref, sample = tabulated
    for gene in data.index.genes:
        view = _GeneView(data, gene, multiple=False)
        t = data.index.tables[view.g]
        args = (data.tab, view.rows, view.n_rows, view.vbeg, view.vbeg + view.n_span, view.mask, t.words)
        per_read = candidateSets(*args)
        want_sets, want_count = np.unique(per_read, axis=0, return_counts=True)
        got_sets, got_count = candidateSetsDistinct(*args)
        assert np.array_equal(got_sets, want_sets) and np.array_equal(got_count, want_count), gene
        host_sets, host_count = distinctSets(per_read)
        assert np.array_equal(host_sets, want_sets) and np.array_equal(host_count, want_count), gene


def test_json_roundtrip_path(device, tabulated, tmp_path):
    """selectKirTypingModel(method, '<file>.json') -- the reference's calling convention."""
    from kir_graph_amd.hisat2 import writeReadsAndVariantsData
    data, ref, sample = tabulated
    path = str(tmp_path / "s.variant.json")
    writeReadsAndVariantsData(data.asDict(), path)
    a = selectKirTypingModel("full", path, top_n=600, variant_correction=True, device=device).typing(sample.gene_cn)
    b = selectKirTypingModel("full", data, top_n=600, variant_correction=True).typing(sample.gene_cn)
    assert a == b


def test_fast_json_writer_is_byte_identical(device, tmp_path):
    """hisat2.writeSampleJson == writeReadsAndVariantsData(data.asDict()) (json.dump of dataclasses.asdict,
    hisat2.py:847-856), with SAM text kept, novel variants, insertion values and empty lists in the file."""
    from kir_graph_amd.hisat2 import extractVariantFromText, writeReadsAndVariantsData, writeSampleJson
    from kir_graph_amd.index import GkIndex
    from kir_graph_amd.msa2hisat import Variant
    sidx = synth.makeIndex(seed=31, n_genes=2, var_range=(200, 300), allele_range=(10, 20))
    gidx = GkIndex.fromVariants(sidx.variants, genes=sidx.genes, exons=sidx.exons)
    for n_pairs in (9000, 3):       # more than two batches of the writer / less than one
        sample = synth.makeSample(sidx, seed=8, n_pairs=n_pairs, err_rate=0.004)
        lines = synth.toSamLines(sample)
        Variant.novel_id = 0
        data = extractVariantFromText([("\n".join(lines) + "\n").encode()], gidx, dev=device, keep_text=True)
        a, b = str(tmp_path / f"a{n_pairs}.json"), str(tmp_path / f"b{n_pairs}.json")
        writeReadsAndVariantsData(data.asDict(), a)
        writeSampleJson(data, b)                      # native: the collated text + line numbers (hisat2.PairsText)
        assert open(a, "rb").read() == open(b, "rb").read()
        data.pairs_text = [data.pairs_text[i] for i in range(len(data.pairs_text))]
        writeSampleJson(data, b)                      # ... and from a plain list of (l_sam, r_sam)
        assert open(a, "rb").read() == open(b, "rb").read()
        if n_pairs > 100:
            assert data.tab.n_novel > 0 and data.tab.n_valid > 8192
        data.tab.close()


def test_bounded_search_equals_exact_search(device, small_case, monkeypatch):
    """GK_SEARCH=bound (integer bound first, float64 sums for the sets that can reach the cut) and GK_SEARCH=exact
    (float64 sums for every candidate) give the same bits in every field of every copy-number step; the mismatch
    table the bound works on equals the counts read off the log-likelihoods."""
    from kir_graph_amd.hisat2 import extractVariant, pairLines
    from kir_graph_amd.kir_typing import selectKirTypingModel
    sidx, gidx, sample = small_case
    data = extractVariant(pairLines(synth.toSamLines(sample)), gidx, dev=device)
    gene_cn = {g: (k % 4) + 1 for k, g in enumerate(sidx.genes)}
    results = {}
    from kir_graph_amd.typing_mulit_allele import SEARCH_STATS
    stats = {}
    # exact = float64 sums for every candidate; the others must give its bits: the integer bound through the whole-sample
    # calls (gk_sample_search) and through the gene-after-gene path (Typing.typing -> typingPerGene -> gk_search_run)
    from kir_graph_amd.kir_typing import Typing
    for mode, search, per_gene in (("exact", "exact", False), ("bound", "bound", False), ("per_gene", "bound", True),
                                   ("per_gene_exact", "exact", True)):
        monkeypatch.setenv("GK_SEARCH", search)
        before = dict(SEARCH_STATS)
        for method, top_n in (("full", 600), ("full", 7), ("exonfirst_1", 60)):
            typer = selectKirTypingModel(method, data, top_n=top_n, variant_correction=True)
            calls = Typing.typing(typer, gene_cn) if per_gene else typer.typing(gene_cn)
            results[(mode, method, top_n)] = (calls, typer._result)
        stats[mode] = {k: SEARCH_STATS[k] - before[k] for k in before}
    # the comparison below covers both routes of a step only if both were taken: steps served by the integer bound
    # and steps it handed back to the exact kernels (ties across a cut / rows equal in all three keys)
    for mode in ("bound", "per_gene"):
        assert stats[mode]["bounded"] > 0 and stats[mode]["redone_exactly"] > 0, (mode, stats)
    for (mode, method, top_n), (calls, res) in results.items():
        if mode == "exact":
            continue
        want_calls, want = results[("exact", method, top_n)]
        assert calls == want_calls
        for gene in want:
            assert len(res[gene]) == len(want[gene])
            for x, y in zip(res[gene], want[gene]):
                for f in ("value", "value_sum_indv", "allele_id", "fraction"):
                    assert np.array_equal(np.asarray(getattr(x, f)), np.asarray(getattr(y, f))), (method, top_n, gene, f)


def test_mismatch_table_equals_counts_from_the_lists(device, small_case, monkeypatch):
    """The u8 table written next to the log-likelihoods (gk_compat_log_miss) holds exactly the integer
    mismatch counts of gk_compat's own tally (miss[r, a])."""
    from kir_graph_amd.engine import DeviceModel
    from kir_graph_amd.hisat2 import extractVariant, pairLines
    from kir_graph_amd.typing_mulit_allele import sharedLogTable
    monkeypatch.setenv("GK_SEARCH", "bound")
    sidx, gidx, sample = small_case
    data = extractVariant(pairLines(synth.toSamLines(sample)), gidx, dev=device)
    tab = data.tab
    from kir_graph_amd.engine import DeviceIndex
    for g, t in enumerate(gidx.tables):
        rows, n_rows = tab.selectGene(g, False)
        if not n_rows:
            continue
        vflag = device.alloc(max(tab.n_var_total, 1), np.uint8).zero()
        mask = tab.dindex.masks[g]
        a = DeviceModel(tab, rows, n_rows, vflag, t.vbeg, t.vend, mask, t.words, t.n_allele, sharedLogTable(device))
        a.finishLog()
        b = DeviceModel(tab, rows, n_rows, vflag, t.vbeg, t.vend, mask, t.words, t.n_allele, sharedLogTable(device),
                        want_miss=True)
        assert a.boundOk
        got = a.miss8.download().reshape(t.n_allele, a.ldm)
        want = b.miss.download().reshape(t.n_allele, n_rows)
        assert np.array_equal(got[:, :n_rows], want)
        assert not got[:, n_rows:].any()
        assert np.array_equal(a.msum.download(), want.sum(axis=1, dtype=np.uint32))


def test_index_table_equals_the_float_table(device, small_case, monkeypatch):
    """The index form of a gene's table (gk_compat_index: uint16 dense indices into the value table, 2 bytes per entry)
    expands (gk_expand_index) to exactly the float64 table of gk_compat_log_miss, with the same mismatch bytes; a whole
    sample typed on index tables (GK_INDEX_TABLE=1) gives the bits of the float64 tables (the default) in every field
    of every copy-number step."""
    import ctypes as C
    from kir_graph_amd._lib import check, lib
    from kir_graph_amd.engine import DeviceModel
    from kir_graph_amd.hisat2 import extractVariant, pairLines
    from kir_graph_amd.kir_typing import selectKirTypingModel
    from kir_graph_amd.typing_mulit_allele import sharedLogTable
    monkeypatch.setenv("GK_SEARCH", "bound")
    sidx, gidx, sample = small_case
    data = extractVariant(pairLines(synth.toSamLines(sample)), gidx, dev=device)
    tab, logs = data.tab, sharedLogTable(device)
    for g, t in enumerate(gidx.tables):
        rows, n_rows = tab.selectGene(g, False)
        if not n_rows:
            continue
        vflag = device.alloc(max(tab.n_var_total, 1), np.uint8).zero()
        a = DeviceModel(tab, rows, n_rows, vflag, t.vbeg, t.vend, tab.dindex.masks[g], t.words, t.n_allele, logs)
        a.finishLog()                                   # every value of the gene is defined from here on
        lidx = device.alloc((t.n_allele, a.ldm), np.uint16)
        miss8 = device.alloc((t.n_allele, a.ldm), np.uint8)
        flags = device.alloc(1, np.uint32)
        check(lib().gk_compat_index(device.ctx, tab.handle, rows.ptr, n_rows, vflag.ptr, t.vbeg, t.vend,
                                    tab.dindex.masks[g].ptr, t.words, t.n_allele, 0, logs.handle, lidx.ptr, miss8.ptr,
                                    a.ldm, flags.ptr))
        back = device.alloc((t.n_allele, n_rows), np.float64)
        check(lib().gk_expand_index(device.ctx, logs.handle, lidx.ptr, a.ldm, n_rows, t.n_allele, back.ptr, n_rows))
        assert int(flags.download()[0]) == int(a._bound_flags.download()[0]) == 0
        idx = lidx.download().reshape(t.n_allele, a.ldm)[:, :n_rows]
        assert idx.max() < 0xFFFF                         # nothing undefined, nothing beyond 16 bits
        assert np.array_equal(back.download(), a.L.download())
        assert np.array_equal(miss8.download(), a.miss8.download())
    gene_cn = {g: (k % 4) + 1 for k, g in enumerate(sidx.genes)}
    results = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("GK_INDEX_TABLE", mode)
        typer = selectKirTypingModel("full", data, top_n=600, variant_correction=True)
        calls = typer.typing(gene_cn)
        results[mode] = (calls, typer._result, typer)
    models = [steps[-1].allele_prob.parts[0][0] for steps in results["1"][1].values() if steps]
    assert models and all(m._indexed and m._L is None for m in models)      # the index form really was what ran
    assert results["1"][0] == results["0"][0]
    for gene, want in results["0"][1].items():
        for x, y in zip(results["1"][1][gene], want):
            for f in ("value", "value_sum_indv", "allele_id", "fraction"):
                assert np.array_equal(np.asarray(getattr(x, f)), np.asarray(getattr(y, f))), (gene, f)
    # the float64 form is made on demand from the indices (allele_prob / log_probs readers)
    m = models[0]
    assert m.L is not None and np.array_equal(np.asarray(results["1"][1][next(iter(results["1"][1]))][-1].allele_prob),
                                             np.asarray(results["0"][1][next(iter(results["0"][1]))][-1].allele_prob))



@pytest.mark.gpu
def test_forms_of_the_sample_search_and_of_the_factor_agree(device, small_case, monkeypatch):
    """A whole sample typed two ways gives the same bits in every field of every copy-number step: all genes pipelined on
    one stream in one library call (gk_sample_search: the default), or gene after gene (Typing.typing -> typingPerGene:
    a table and a gk_search_run per gene).  (The lock-step form of the gene loop serves index tables:
    test_index_table_equals_the_float_table.)"""
    from kir_graph_amd.hisat2 import extractVariant, pairLines
    monkeypatch.setenv("GK_SEARCH", "bound")
    sidx, gidx, sample = small_case
    data = extractVariant(pairLines(synth.toSamLines(sample)), gidx, dev=device)
    gene_cn = {g: (k % 4) + 1 for k, g in enumerate(sidx.genes)}
    results = {}
    from kir_graph_amd.kir_typing import Typing
    for name in ("default", "gene after gene"):
        typer = selectKirTypingModel("full", data, top_n=600, variant_correction=True)
        assert typer._wholeSample()
        results[name] = (typer.typing(gene_cn) if name == "default" else Typing.typing(typer, gene_cn), typer._result)
    want_calls, want = results["default"]
    assert any(len(steps) > 1 for steps in want.values())
    for name, (calls, got) in results.items():
        assert calls == want_calls, name
        for gene, steps in want.items():
            assert len(got[gene]) == len(steps), (name, gene)
            for x, y in zip(got[gene], steps):
                for f in ("value", "value_sum_indv", "allele_id", "fraction"):
                    assert np.array_equal(np.asarray(getattr(x, f)), np.asarray(getattr(y, f))), (name, gene, f)


@pytest.mark.gpu
def test_new_values_settle_per_gene_and_give_the_same_bits(device, small_case, monkeypatch):
    """A cohort of DISTINCT samples: every sample may bring products the log10 value table has not seen.  The pipelined
    gene loop settles them per gene -- a compatibility kernel that met such a product leaves it in the entry's place and
    says so in its gene's flag word, only that gene's table is patched (`tables_patched`) -- and every field of every copy-number step equals the lock-step
    form's and the oracle's.  The samples grow (more pairs, more errors), so later ones do bring new products."""
    from kir_graph_amd.hisat2 import extractVariant, pairLines
    monkeypatch.setenv("GK_SEARCH", "bound")
    sidx, gidx, _ = small_case
    rewritten, n_genes = [], []
    for k, (pairs, err) in enumerate(((1500, 0.0), (2500, 0.001), (4000, 0.004), (4000, 0.004), (6000, 0.01))):
        sample = synth.makeSample(sidx, seed=4000 + (k if k != 3 else 2), n_pairs=pairs, err_rate=err)
        lines = synth.toSamLines(sample)
        data = extractVariant(pairLines(lines), gidx, dev=device)
        # (a gene without reads and a copy number >= 2 makes the reference -- and the oracle -- raise AxisError)
        gene_cn = {g: ((k + i) % 3) + 1 for i, g in enumerate(sidx.genes) if sample.gene_cn.get(g, 0) > 0}
        typer = selectKirTypingModel("full", data, top_n=600, variant_correction=True)
        assert typer._wholeSample()
        calls = typer.typing(gene_cn)
        rewritten.append(typer.tables_rewritten + typer.tables_patched)
        n_genes.append(sum(1 for steps in typer._result.values() if steps))
        cpu = oty.makeTyper("full", ot.tabulateLines(lines, gidx.variants), top_n=600, variant_correction=True)
        assert calls == cpu.typing(gene_cn), k
        for gene, steps in cpu.results.items():
            assert len(typer._result[gene]) == len(steps), (k, gene)
            for a, b in zip(typer._result[gene], steps):
                same_result(a, b)
        data.tab.close()
    # sample 3 repeats sample 2: nothing new, no table written twice; no sample rewrites a table more than a few times
    assert rewritten[3] == 0, rewritten
    assert all(r <= 3 * g for r, g in zip(rewritten, n_genes)), (rewritten, n_genes)


@pytest.mark.gpu
def test_em_of_all_genes_in_one_call_equals_the_per_gene_calls(tabulated, monkeypatch):
    """gk_sample_em (candidate sets, distinct sets and the SQUAREM loops of every gene in one library call, a workgroup
    per gene) gives the reports of the per-gene path -- abundances bit for bit (same sets in the same order through the
    same loop), read counts, iterations, distinct sets -- and the same calls and warnings."""
    data, ref, sample = tabulated
    got = {}
    from kir_graph_amd.kir_typing import Typing
    for form in ("1", "0"):
        typer = selectKirTypingModel("em", data)
        calls = typer.typing(sample.gene_cn) if form == "1" else Typing.typing(typer, sample.gene_cn)
        got[form] = (calls, {g: [(r.allele, r.count, r.prob, r.cn) for r in rep] for g, rep in typer._result.items()},
                     dict(typer.em_info))
    assert got["1"][0] == got["0"][0]
    assert list(got["1"][1]) == list(got["0"][1])
    assert got["1"][1] == got["0"][1]
    assert got["1"][2] == got["0"][2] and any(v["iterations"] > 0 for v in got["1"][2].values())


@pytest.mark.gpu
@pytest.mark.parametrize("method", ["exonfirst_1", "exonfirst_0.9"])
def test_exon_first_of_all_genes_in_two_calls_equals_the_per_gene_path(tabulated, monkeypatch, method):
    """Exon-first through gk_sample_search twice (exon models of every gene; full tables + the candidate searches, each
    step offering the alleles of one exon group) gives every field of every result of the per-gene path
    (AlleleTypingExonFirst on a thread per gene): exon steps, candidate steps, the merged ranking, calls, warnings."""
    data, ref, sample = tabulated
    got = {}
    from kir_graph_amd.kir_typing import Typing
    for form in ("1", "0"):
        typer = selectKirTypingModel(method, data, top_n=600, variant_correction=True)
        assert typer._wholeSampleExonFirst()
        calls = typer.typing(sample.gene_cn) if form == "1" else Typing.typing(typer, sample.gene_cn)
        got[form] = (calls, typer._result, typer.getAllPossibleTyping())
    assert got["1"][0] == got["0"][0]
    assert got["1"][2] == got["0"][2]
    assert list(got["1"][1]) == list(got["0"][1])
    for gene, want in got["0"][1].items():
        have = got["1"][1][gene]
        assert len(have) == len(want), gene
        for a, b in zip(have, want):
            same_result(a, b)

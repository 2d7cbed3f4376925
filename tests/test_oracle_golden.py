"""
The CPU oracle against fixtures produced by the reference itself (tests/golden/make_golden.py).

Integers, ids, orders and allele calls must be identical; floats within 1e-9 relative (numpy's
log10 / argsort tie order depend on the host's SIMD level, see oracle/__init__.py).
"""
import copy
import gzip
import io
import json
import os
import tempfile

import numpy as np
import pandas as pd
import pytest

from kir_graph_amd.index import getVariants
from oracle import cn as ocn, em as oem, tabulate as ot, typing as oty

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    with gzip.open(os.path.join(GOLD, name), "rt") as f:
        return json.load(f)


def unhex(xs):
    return np.array([float.fromhex(x) for x in xs])


def write_index(text):
    d = tempfile.mkdtemp()
    for ext, body in text.items():
        with open(f"{d}/ix.{ext}", "w") as f:
            f.write(body)
    return d + "/ix"


def close(a, b, rel=1e-9):
    a, b = np.asarray(a, dtype=np.float64).ravel(), np.asarray(b, dtype=np.float64).ravel()
    assert a.shape == b.shape
    assert np.allclose(a, b, rtol=rel, atol=0), float(np.max(np.abs(a - b)))


@pytest.fixture(scope="module")
def t1():
    return load("t1_tabulation.json.gz")


@pytest.fixture(scope="module")
def case():
    return load("typing_case.json.gz")


def test_index_reader_matches_reference_variants(t1):
    variants = getVariants(write_index(t1["index"]))
    ref = [v for v in t1["variants"] if not v[0].startswith("nv")]
    assert [[v.id, v.typ, v.pos, v.val, v.length, v.allele, v.in_exon] for v in variants] == ref


def test_walk_records(t1):
    for rec in t1["records"]:
        if "error" in rec:
            with pytest.raises((AssertionError, NotImplementedError)) as e:
                ot.walkRecord(rec["line"])
            assert type(e.value).__name__ == rec["error"]
        else:
            vs, clip = ot.walkRecord(rec["line"])
            assert [[v.typ, v.pos, v.length, v.val, v.id] for v in vs] == rec["variants"]
            assert clip == rec["clip"]


def test_tabulate_hand_built_pairs(t1):
    variants = getVariants(write_index(t1["index"]))
    pairs = list(ot.pairMates(t1["lines"]))
    assert len(pairs) == t1["n_pairs"]
    data = ot.tabulateLines(t1["lines"], variants)
    assert len(data["reads"]) == t1["n_kept"]
    for got, want in zip(data["reads"], t1["reads"]):
        for k in ("lpv", "lnv", "rpv", "rnv", "multiple", "backbone"):
            assert got[k] == want[k], k
    assert [[v.id, v.typ, v.pos, v.val, v.length, v.allele, v.in_exon] for v in data["variants"]] == t1["variants"]


def test_pairing_order():
    t2 = load("t2_pairing.json.gz")
    lines = t2["lines"]
    got = [[lines.index(a), lines.index(b)] for a, b in ot.pairMates(lines)]
    assert got == t2["pairs"]


def test_tabulate_synthetic_sample(case):
    variants = getVariants(write_index(case["index"]))
    data = ot.tabulateLines(case["lines"], variants)
    assert len(data["reads"]) == len(case["reads"])
    for got, want in zip(data["reads"], case["reads"]):
        for k in ("lpv", "lnv", "rpv", "rnv", "multiple", "backbone"):
            assert got[k] == want[k]
    novel = [[v.id, v.typ, v.pos, v.val, v.length, v.ref] for v in data["variants"] if str(v.id).startswith("nv")]
    assert novel == case["novel"]


@pytest.fixture(scope="module")
def tabulated(case):
    variants = getVariants(write_index(case["index"]))
    return ot.tabulateLines(case["lines"], variants)


def test_error_correction_and_probabilities(case, tabulated):
    m = case["model"]
    g = m["gene"]
    reads = [dict(r) for r in copy.deepcopy(tabulated["reads"]) if r["backbone"] == g and r["multiple"] == 1]
    variants = [v for v in tabulated["variants"] if v.ref == g]
    model = oty.GeneModel(reads, variants, top_n=600, variant_correction=True)
    assert model.readsNum() == m["n_reads"]
    assert [[r["lpv"], r["rpv"], r["lnv"], r["rnv"]] for r in model.reads[:60]] == m["kept_lists"]
    assert [model.id_to_allele[i] for i in range(len(model.id_to_allele))] == m["alleles"]
    assert np.array_equal(model.probs[:40].ravel(), unhex(m["probs_head"]))       # products are exact IEEE
    close(model.log_probs[:40], unhex(m["log_probs_head"]))
    close(model.log_probs.sum(axis=0), unhex(m["colsum"]))
    # integer hit table: log-probabilities follow from (nvar, miss) up to rounding
    miss, nvar = oty.missTable(model.reads, model.variants, model.allele_to_id)
    approx = (nvar[:, None] - miss) * np.log10(0.999) + miss * np.log10(0.001)
    close(model.log_probs, approx, rel=1e-10)


@pytest.mark.parametrize("method", ["full", "exonfirst_1", "exonfirst_0.9"])
def test_likelihood_typing(case, tabulated, method):
    want = case["methods"][method]
    typer = oty.makeTyper(method, copy.deepcopy(tabulated), top_n=600, variant_correction=True)
    calls, warn = typer.typing(case["gene_cn"])
    assert calls == want["calls"]
    assert warn == want["warnings"]
    for gene, w in want["genes"].items():
        steps = typer.results[gene]
        assert len(steps) == w["steps"]
        last = steps[-1]
        assert last.n == w["n"]
        close(last.value[:50], unhex(w["value"]))
        close(last.value_sum_indv[:50], unhex(w["value_sum_indv"]))
        close(last.fraction[:50], unhex(w["fraction"]))
        # ids agree wherever the reference's values are not tied (tie order is host dependent)
        v = unhex(w["value"])
        untied = np.ones(len(v), dtype=bool)
        untied[1:] &= v[1:] != v[:-1]
        untied[:-1] &= v[:-1] != v[1:]
        got_ids = np.asarray(last.allele_id[:50])
        want_ids = np.asarray(w["allele_id"])
        assert np.array_equal(np.sort(got_ids[untied], axis=1), np.sort(want_ids[untied], axis=1))
    poss = typer.allPossible()
    assert len(poss) == len(want["possible"])
    for a, b in zip(poss, want["possible"]):
        assert a["gene"] == b["gene"] and a["rank"] == b["rank"]
        assert a["value"] == pytest.approx(float.fromhex(b["value"]), rel=1e-9)


def test_em_typing(case, tabulated):
    want = case["methods"]["em"]
    typer = oem.ReportTyper(copy.deepcopy(tabulated))
    calls, warn = typer.typing(case["gene_cn"])
    assert sorted(calls) == sorted(want["calls"])
    assert warn == want["warnings"]
    for gene, rows in want["genes"].items():
        got = sorted([[r["allele"], r["count"], r["prob"]] for r in typer.results[gene]])
        assert [g[:2] for g in got] == [r[:2] for r in rows]
        close([g[2] for g in got], unhex([r[2] for r in rows]), rel=1e-9)


def test_copy_number():
    t8 = load("t8_cn.json.gz")
    tables = [pd.DataFrame(rows, columns=["gene", "pos", "depth"]) for rows in t8["depth_tables"]]
    kw = {"base_dev": 0.08, "start_base": 2}
    for mode, res in t8["per_sample"].items():
        for df, want in zip(tables, res):
            cns, samples, model = ocn.predictCN([df], mode, "LCND", kw, assume_3DL3_diploid=True)
            ref = pd.read_csv(io.StringIO(want["tsv"]), sep="\t")
            assert dict(zip(ref["gene"], ref["cn"])) == {k: int(v) for k, v in cns[0].items()}
            # the TSV prints depths with pandas' float format (15-17 digits): compare to 1e-13
            assert list(ref["gene"]) == list(samples[0])
            close(list(ref["depth"]), [float(v) for v in samples[0].values()], rel=1e-13)
            assert model.base == pytest.approx(float.fromhex(want["base"]), rel=1e-12)
            assert model.x_max == float.fromhex(want["x_max"])
            assert model.bin_num == want["bin_num"]
    for method, files in t8["cohort"].items():
        cns, _, _ = ocn.predictCN(tables, "p75", method, kw if method == "LCND" else {}, False)
        for got, want in zip(cns, files):
            ref = pd.read_csv(io.StringIO(want), sep="\t")
            assert dict(zip(ref["gene"], ref["cn"])) == {k: int(v) for k, v in got.items()}


def test_copy_number_per_gene():
    """predictSamplesCN(per_gene=True) of the reference (kir_cn.py:195-222) on six samples: one model per gene."""
    t8 = load("t8_cn.json.gz")
    rows = t8["depth_tables"] + t8["per_gene"]["depth_tables_extra"]
    tables = [pd.DataFrame(r, columns=["gene", "pos", "depth"]) for r in rows]
    kw = {"base_dev": 0.08, "start_base": 2}
    for method in ("LCND", "KDE"):
        want = t8["per_gene"][method]
        cns, samples, models = ocn.predictCNPerGene(tables, "p75", method, kw if method == "LCND" else {})
        for got, text in zip(cns, want["tsv"]):
            ref = pd.read_csv(io.StringIO(text), sep="\t")
            assert dict(zip(ref["gene"], ref["cn"])) == {k: int(v) for k, v in got.items()}, method
        for m in want["models"]:
            mine = models[m["gene"]]
            assert mine.x_max == float.fromhex(m["x_max"]), (method, m["gene"])
            if method == "LCND":
                assert mine.base == pytest.approx(float.fromhex(m["base"]), rel=1e-12)
                assert mine.bin_num == m["bin_num"]
            else:
                close(mine.local_min, unhex(m["local_min"]), rel=1e-9)


def test_numpy_reduction_tree():
    """Pure-Python restatement of numpy's add.reduce tree (the device follows the same tree)."""
    from oracle.sumtree import numpySum
    t10 = load("t10_sums.json.gz")
    rng = np.random.default_rng(11)   # the vectors are regenerated exactly as make_golden.py drew them
    for c in t10["cases"]:
        x = -rng.random(c["n"]) * 10
        if c["seed_vec"] is not None:
            assert np.array_equal(x, unhex(c["seed_vec"]))
        assert numpySum(x) == float.fromhex(c["sum"])
        assert float(np.add.reduce(x)) == float.fromhex(c["sum"])
        col = unhex(c["colsum"])
        assert numpySum(x) == col[0] and numpySum(x[::-1]) == col[1]


# ---------------------------------------------------------------- T11: pileup error correction (a21)
@pytest.fixture(scope="module")
def t11():
    return load("t11_pileup.json.gz")


def test_pileup_column_parser_and_ratios(t11):
    """oracle/pileup.py against the reference's parsePileupBase / getPileupBaseRatio (pileup.py:13-37, 57-81)."""
    from oracle import pileup as op
    for c in t11["parse"]:
        assert op.basesOfColumn(c["bases"]) == c["out"], c["bases"]
    got = op.ratiosOfColumns([tuple(r) for r in t11["rows"]])
    want = {(t11["gene"], r["pos"]): r for r in t11["ratio"]}
    assert set(got) == set(want)
    for key, entry in got.items():
        w = want[key]
        assert list(entry) == w["order"]                      # dict order = first appearance (decides max() ties)
        for k, v in entry.items():
            assert v == (w["entry"][k] if k == "all" else float.fromhex(w["entry"][k])), (key, k)


def test_pileup_correction_rule(t11):
    """oracle/tabulate.pileupCorrect against hisat2.errorCorrection (609-654) at the depth-20 / 0.2 / 0.8 edges."""
    from kir_graph_amd.msa2hisat import Variant
    from oracle import pileup as op
    ratio = op.ratiosOfColumns([tuple(r) for r in t11["rows"]])
    for pos, val, want in t11["fixes"]:
        if ":" in val:
            typ, raw = val.split(":")
            v = Variant(pos=pos, typ=typ, ref=t11["gene"], val=None if raw == "None" else raw, length=2)
        else:
            v = Variant(pos=pos, typ="single", ref=t11["gene"], val=val, length=1)
        assert str(ot.pileupCorrect(v, ratio).val) == str(want), (pos, val)


def test_tabulate_reads_beyond_the_128_byte_record():
    """T12: the reference's lists for pairs with 17-60 substitutions per mate, a 4200-base novel deletion and 19 CIGAR
    ops -- the inputs the HIP path keeps in its wide record format; the oracle has no capacity and must agree."""
    t12 = load("t12_wide.json.gz")
    variants = getVariants(write_index(t12["index"]))
    data = ot.tabulateLines(t12["lines"], variants)
    assert len(data["reads"]) == len(t12["reads"])
    for got, want in zip(data["reads"], t12["reads"]):
        for k in ("lpv", "lnv", "rpv", "rnv", "multiple", "backbone"):
            assert got[k] == want[k]
    novel = [[v.id, v.typ, v.pos, v.val, v.length, v.ref] for v in data["variants"] if str(v.id).startswith("nv")]
    assert novel == t12["novel"]
    assert t12["most_positives"] > 44        # more events than two gk_mate records could hold

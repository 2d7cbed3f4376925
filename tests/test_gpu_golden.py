"""GPU parity against fixtures produced by the reference itself (tests/golden): SAM text in,
the reference's lists / calls out, through the product's text decoder and the HIP path."""
import gzip
import json
import os

import numpy as np
import pytest

from kir_graph_amd.hisat2 import extractVariant, pairLines
from kir_graph_amd.index import GkIndex
from kir_graph_amd.kir_typing import selectKirTypingModel
from kir_graph_amd.msa2hisat import Variant

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    with gzip.open(os.path.join(GOLD, name), "rt") as f:
        return json.load(f)


def index_from(text, tmp_path):
    for ext, body in text.items():
        (tmp_path / f"ix.{ext}").write_text(body)
    return GkIndex.load(str(tmp_path / "ix"))


def unhex(xs):
    return np.array([float.fromhex(x) for x in xs])


def test_hand_built_records(device, tmp_path):
    t1 = load("t1_tabulation.json.gz")
    gidx = index_from(t1["index"], tmp_path)
    Variant.novel_id = 0
    data = extractVariant(pairLines(t1["lines"]), gidx, dev=device)
    reads = data.reads()
    assert len(reads) == t1["n_kept"]
    for got, want in zip(reads, t1["reads"]):
        assert (got.lpv, got.lnv, got.rpv, got.rnv) == (want["lpv"], want["lnv"], want["rpv"], want["rnv"])
        assert got.multiple == want["multiple"] and got.backbone == want["backbone"]
    assert [[v.id, v.typ, v.pos, v.val, v.length, v.allele, v.in_exon] for v in data.variants] == t1["variants"]
    assert Variant.novel_id == sum(1 for v in t1["variants"] if v[0].startswith("nv"))


@pytest.fixture(scope="module")
def typed_case(device, tmp_path_factory):
    case = load("typing_case.json.gz")
    gidx = index_from(case["index"], tmp_path_factory.mktemp("ix"))
    Variant.novel_id = 0
    data = extractVariant(pairLines(case["lines"]), gidx, dev=device)
    return case, data


def test_synthetic_sample_lists(typed_case):
    case, data = typed_case
    reads = data.reads()
    assert len(reads) == len(case["reads"])
    for got, want in zip(reads, case["reads"]):
        assert (got.lpv, got.lnv, got.rpv, got.rnv) == (want["lpv"], want["lnv"], want["rpv"], want["rnv"])
    assert [[v.id, v.typ, v.pos, v.val, v.length, v.ref] for v in data.novel] == case["novel"]


@pytest.mark.parametrize("method", ["full", "exonfirst_1", "exonfirst_0.9"])
def test_allele_calls_and_likelihoods(typed_case, method):
    case, data = typed_case
    want = case["methods"][method]
    typer = selectKirTypingModel(method, data, top_n=600, variant_correction=True)
    calls, warn = typer.typing(case["gene_cn"])
    assert calls == want["calls"]
    assert warn == want["warnings"]
    for gene, w in want["genes"].items():
        last = typer._result[gene][-1]
        assert len(typer._result[gene]) == w["steps"]
        assert np.allclose(last.value[:50], unhex(w["value"]), rtol=1e-9, atol=0)       # north_star: 1e-5
        assert np.allclose(np.asarray(last.value_sum_indv[:50]).ravel(), unhex(w["value_sum_indv"]), rtol=1e-9, atol=0)
        assert np.allclose(np.asarray(last.fraction[:50]).ravel(), unhex(w["fraction"]), rtol=1e-9, atol=0)
        assert last.n == w["n"]
        # ids and names agree wherever the reference's values are not tied with a neighbour (the order inside a
        # group of tied values is numpy's argsort tie order, which depends on the host's SIMD level)
        v = unhex(w["value"])
        apart = np.abs(np.diff(v)) > 1e-7 * np.abs(v[1:])      # well beyond cross-host last-bit differences
        untied = np.ones(len(v), dtype=bool)
        untied[1:] &= apart
        untied[:-1] &= apart
        got_ids, want_ids = np.asarray(last.allele_id[:50]), np.asarray(w["allele_id"])
        assert np.array_equal(np.sort(got_ids[untied], axis=1), np.sort(want_ids[untied], axis=1)), gene
        names = last.allele_name[:50]
        for k in np.flatnonzero(untied):
            assert sorted(names[k]) == sorted(w["allele_name"][k]), (gene, k)
    got_rows, want_rows = typer.getAllPossibleTyping(), want["possible"]
    assert [(r["gene"], r["rank"]) for r in got_rows] == [(r["gene"], r["rank"]) for r in want_rows]


def test_em_abundances(typed_case):
    case, data = typed_case
    want = case["methods"]["em"]
    typer = selectKirTypingModel("em", data)
    calls, warn = typer.typing(case["gene_cn"])
    assert sorted(calls) == sorted(want["calls"])
    for gene, rows in want["genes"].items():
        got = sorted([[r.allele, r.count, r.prob] for r in typer._result[gene]])
        assert [g[:2] for g in got] == [r[:2] for r in rows]
        assert np.allclose([g[2] for g in got], unhex([r[2] for r in rows]), rtol=1e-5, atol=1e-9)


@pytest.mark.parametrize("how", ["python packer", "native packer"])
def test_reads_beyond_the_128_byte_record_equal_the_reference(device, tmp_path, how):
    """T12 (fixture made by the reference itself): pairs with 17-60 substitutions per mate, a 4200-base novel deletion
    and 19 CIGAR ops take the wide record format (gk_mate_wide, tab_count_wide / tab_emit_wide) and give the reference's
    lists and novel variants -- ids in first-appearance order included."""
    from kir_graph_amd.hisat2 import extractVariantFromText
    t12 = load("t12_wide.json.gz")
    gidx = index_from(t12["index"], tmp_path)
    Variant.novel_id = 0
    if how == "python packer":
        data = extractVariant(pairLines(t12["lines"]), gidx, dev=device)
    else:
        sam = tmp_path / "t12.sam"
        sam.write_text("\n".join(t12["lines"]) + "\n")
        data = extractVariantFromText(str(sam), gidx, dev=device, keep_text=False)
    assert int(np.count_nonzero(data.tab.mates.download()["n_cig"] == 0xFF)) >= 2 * 8      # the wide pairs are there
    reads = data.reads()
    assert len(reads) == len(t12["reads"])
    for got, want in zip(reads, t12["reads"]):
        assert (got.lpv, got.lnv, got.rpv, got.rnv) == (want["lpv"], want["lnv"], want["rpv"], want["rnv"])
        assert got.multiple == want["multiple"] and got.backbone == want["backbone"]
    novel = [[v.id, v.typ, v.pos, v.val, v.length, v.ref] for v in data.variants if str(v.id).startswith("nv")]
    assert novel == t12["novel"]
    data.tab.close()


def test_a_parked_sample_comes_back_with_the_reference_lists(device, tmp_path):
    """--cn-cohort keeps a sample between its depth and its typing as its packed records in compact form
    (ParkedRecords: gk_mates_compact); expanded and tabulated again they give the reference's lists and novel ids of
    fixture T12 -- pairs of the wide format, inserted strings and a first novel id that is not 0 included -- and the records
    themselves come back word for word where they were used."""
    import ctypes as C
    from kir_graph_amd import _lib
    from kir_graph_amd._lib import check, lib
    from kir_graph_amd.hisat2 import ParkedRecords
    t12 = load("t12_wide.json.gz")
    gidx = index_from(t12["index"], tmp_path)
    Variant.novel_id = 0
    data = extractVariant(pairLines(t12["lines"]), gidx, dev=device)
    before = data.tab.mates.download()
    n_mates = len(before)
    off0, ids0 = data.tab.offsets().copy(), data.tab.ids().copy()
    from kir_graph_amd.packed import CompactMates
    on_host = CompactMates(before, threads=2)      # the same compact form made by the host packer's side
    parked = ParkedRecords(data)
    assert parked.nbytes < before.nbytes // 2 and data.tab.handle is None and data.tab.mates is None
    assert parked.nbytes == on_host.nbytes
    assert np.array_equal(device.view(parked.ptr, len(on_host.words), np.uint32), on_host.words)
    # the records, expanded: every field a kernel reads is back (unused array entries are zero now)
    again = device.alloc(n_mates, _lib.MATE_DTYPE)
    check(lib().gk_mates_expand(device.ctx, parked.ptr, n_mates, again.ptr))
    after = again.download()
    for f in ("pos0", "flag", "ref", "nh", "nm", "n_cig", "n_mm", "n_ins"):
        assert np.array_equal(before[f], after[f]), f
    for m in range(n_mates):
        r, q = before[m], after[m]
        if r["n_cig"] == 0xFF:
            assert r["ins"][0] == q["ins"][0]
            continue
        assert np.array_equal(r["cig"][:r["n_cig"]], q["cig"][:r["n_cig"]])
        assert np.array_equal(r["mm"][:r["n_mm"]], q["mm"][:r["n_mm"]])
        assert np.array_equal(r["ins"][:r["n_ins"]], q["ins"][:r["n_ins"]])
    again.free()
    back = parked.restore()
    assert parked.ptr == 0
    assert np.array_equal(back.tab.offsets(), off0) and np.array_equal(back.tab.ids(), ids0)
    reads = back.reads()
    assert len(reads) == len(t12["reads"])
    for got, want in zip(reads, t12["reads"]):
        assert (got.lpv, got.lnv, got.rpv, got.rnv) == (want["lpv"], want["lnv"], want["rpv"], want["rnv"])
    novel = [[v.id, v.typ, v.pos, v.val, v.length, v.ref] for v in back.variants if str(v.id).startswith("nv")]
    assert novel == t12["novel"]
    back.tab.close()

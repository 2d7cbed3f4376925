"""GPU parity on edge cases: row counts around numpy's summation block sizes, small top_n, copy numbers
1-4, forced / automatic zygosity, duplicated alleles (tie heavy), empty genes -- through the list-based
drop-in constructors (``AlleleTyping(reads, variants, ...)``) vs the CPU oracle."""
import copy

import numpy as np
import pytest

from kir_graph_amd import synth
from kir_graph_amd.hisat2 import PairRead
from kir_graph_amd.typing_mulit_allele import AlleleTyping, AlleleTypingExonFirst
from oracle import tabulate as ot, typing as oty

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def one_gene():
    sidx = synth.makeIndex(seed=77, n_genes=1, var_range=(260, 320), allele_range=(22, 28), len_range=(5000, 6000))
    g = sidx.genes[0]
    # duplicate one allele under a second name: every set containing either ties exactly
    twin_of, twin = sidx.alleles[g][0], sidx.alleles[g][0] + "X"
    for v in sidx.variants:
        if twin_of in v.allele:
            v.allele = sorted(v.allele + [twin])
    sidx.alleles[g].append(twin)
    sample = synth.makeSample(sidx, seed=5, n_pairs=21000, gene_cn={g: 2}, frac_multi=0.0)
    data = ot.tabulateLines(synth.toSamLines(sample), sidx.variants)
    variants = data["variants"]
    return g, variants, data["reads"]


def to_pairs(reads):
    return [PairRead(lpv=list(r["lpv"]), lnv=list(r["lnv"]), rpv=list(r["rpv"]), rnv=list(r["rnv"]),
                     multiple=r["multiple"], backbone=r["backbone"]) for r in reads]


def same(a, b):
    assert a.n == b.n
    for f in ("value", "value_sum_indv", "allele_id", "fraction"):
        x, y = np.asarray(getattr(a, f)), np.asarray(getattr(b, f))
        assert x.shape == y.shape, (f, x.shape, y.shape)
        assert np.array_equal(x, y), f
    assert list(map(list, a.allele_name)) == b.allele_name


@pytest.mark.parametrize("n_rows", [1, 7, 8, 9, 127, 128, 129, 1000, 8191, 8192, 8193, 20000])
def test_row_counts_around_block_sizes(device, one_gene, n_rows):
    g, variants, reads = one_gene
    sub = reads[:n_rows]
    cpu = oty.GeneModel(copy.deepcopy(sub), variants, force_homo=False, top_n=600, variant_correction=False)
    gpu = AlleleTyping(to_pairs(sub), variants, force_homo=False, top_n=600, variant_correction=False, device=device)
    assert gpu.getReadsNum() == cpu.readsNum()
    want = cpu.typing(2)
    got = gpu.typing(2)
    if cpu.readsNum():
        assert np.array_equal(gpu.log_probs, cpu.log_probs)
    same(got, want)
    assert got.selectBest() == oty.selectBest(want)


@pytest.mark.parametrize("cn,top_n,force", [(1, 600, None), (2, 5, False), (3, 600, False), (4, 40, False),
                                            (2, 600, None), (3, 600, None), (2, 600, True)])
def test_copy_numbers_topn_zygosity(device, one_gene, cn, top_n, force):
    g, variants, reads = one_gene
    sub = reads[:3000]
    cpu = oty.GeneModel(copy.deepcopy(sub), variants, force_homo=force, top_n=top_n, variant_correction=True)
    gpu = AlleleTyping(to_pairs(sub), variants, force_homo=force, top_n=top_n, variant_correction=True, device=device)
    want, got = cpu.typing(cn), gpu.typing(cn)
    assert len(gpu.result) == len(cpu.result)
    for a, b in zip(gpu.result, cpu.result):
        same(a, b)
    assert got.selectBest() == oty.selectBest(want)
    # the kept reads (after error correction / empty removal) are exposed like the reference's .reads
    assert [(r.lpv, r.rpv, r.lnv, r.rnv) for r in gpu.reads[:50]] == \
           [(r["lpv"], r["rpv"], r["lnv"], r["rnv"]) for r in cpu.reads[:50]]


def test_exon_first_from_lists(device, one_gene):
    g, variants, reads = one_gene
    sub = reads[:4000]
    cpu = oty.ExonFirstModel(copy.deepcopy(sub), variants, top_n=600, candidate_set_threshold=1.0)
    gpu = AlleleTypingExonFirst(to_pairs(sub), variants, top_n=600, candidate_set_threshold=1.0, device=device)
    want, got = cpu.typing(2), gpu.typing(2)
    same(got, want)
    assert gpu.allele_group == cpu.allele_group


def test_gene_without_reads_fails_softly(device, one_gene):
    g, variants, _ = one_gene
    gpu = AlleleTyping([], variants, force_homo=False, top_n=600, device=device)
    res = gpu.typing(2)
    assert res.isFail() and res.selectBest() == ["fail", "fail"]
    assert gpu.getReadsNum() == 0


@pytest.mark.parametrize("n_allele", [64, 65, 130, 200, 256, 257, 330])
def test_wide_genes_cover_every_allele_slot_layout(device, n_allele):
    """Allele counts around the 64-lane slot / 256-allele pass boundaries of the compatibility kernel and
    the 16 / 32 wide tiles of the search kernel."""
    sidx = synth.makeIndex(seed=400 + n_allele, n_genes=1, var_range=(250, 300), allele_range=(n_allele, n_allele),
                           len_range=(5000, 6000))
    g = sidx.genes[0]
    sample = synth.makeSample(sidx, seed=6, n_pairs=700, gene_cn={g: 2}, frac_multi=0.0)
    data = ot.tabulateLines(synth.toSamLines(sample), sidx.variants)
    variants, reads = data["variants"], data["reads"]
    cpu = oty.GeneModel(copy.deepcopy(reads), variants, force_homo=False, top_n=40, variant_correction=True)
    gpu = AlleleTyping(to_pairs(reads), variants, force_homo=False, top_n=40, variant_correction=True, device=device)
    assert len(gpu.id_to_allele) == n_allele
    assert np.array_equal(gpu.probs, cpu.probs)
    assert np.array_equal(gpu.log_probs, cpu.log_probs)
    want, got = cpu.typing(3), gpu.typing(3)
    for a, b in zip(gpu.result, cpu.result):
        same(a, b)
    assert got.selectBest() == oty.selectBest(want)


@pytest.mark.parametrize("correction", [False, True])
def test_reads_without_information_stay_when_asked(device, one_gene, correction):
    """no_empty=False (the way novel_discover.py:273 builds its model): reads without any variant -- from the
    start, or once error correction dropped theirs -- stay in the model and score 0.999 for every allele
    (typing_mulit_allele.py:259-260, 372-374)."""
    g, variants, reads = one_gene
    sub = copy.deepcopy(reads[:2500])
    for i in range(0, len(sub), 9):
        for k in ("lpv", "lnv", "rpv", "rnv"):
            sub[i][k] = []
    cpu = oty.GeneModel(copy.deepcopy(sub), variants, force_homo=False, top_n=600, no_empty=False,
                        variant_correction=correction)
    gpu = AlleleTyping(to_pairs(sub), variants, force_homo=False, top_n=600, no_empty=False,
                       variant_correction=correction, device=device)
    assert gpu.getReadsNum() == cpu.readsNum() == len(sub)
    assert np.array_equal(gpu.probs, cpu.probs)
    assert (gpu.probs[0] == 0.999).all()
    assert np.array_equal(gpu.log_probs, cpu.log_probs)
    want, got = cpu.typing(2), gpu.typing(2)
    for a, b in zip(gpu.result, cpu.result):
        same(a, b)
    assert got.selectBest() == oty.selectBest(want)


def test_a_row_of_thousands_of_factors_takes_the_exact_search(device, monkeypatch):
    """The mismatch bytes of the integer bound are read back from the log-likelihood (m = floor(-L / 3 + 1/4)), which is
    exact only while a row lists fewer than ~5000 variants.  A row beyond 4096 factors (wide records and windows of
    more than 256 variants can produce one) raises the gene's flag, so the search runs on float64 sums for every
    candidate -- and still equals the oracle."""
    from kir_graph_amd.msa2hisat import Variant
    monkeypatch.setenv("GK_SEARCH", "bound")
    rng = np.random.default_rng(3)
    alleles = [f"KIRX*{k:03d}" for k in range(200)]
    variants = []
    for i in range(4300):       # every variant belongs to one allele: a long row then stays far below 100 mismatches per allele
        v = Variant(pos=10 + 2 * i, typ="single", ref="KIRX*BACKBONE", val="ACGT"[i % 4], length=1)
        v.id = f"hv{i}"
        v.allele = [alleles[i % 200]]
        variants.append(v)
    ids = [str(v.id) for v in variants]
    long_row = {"lpv": ids[:6], "rpv": ids[6:10], "lnv": ids[10:2100], "rnv": ids[2100:4200],
                "multiple": 1, "backbone": "KIRX*BACKBONE"}
    short = [{"lpv": [ids[int(k)] for k in rng.choice(4300, 3, replace=False)], "rpv": [],
              "lnv": [ids[int(k)] for k in rng.choice(4300, 6, replace=False)], "rnv": [],
              "multiple": 1, "backbone": "KIRX*BACKBONE"} for _ in range(300)]
    reads = [long_row] + short
    cpu = oty.GeneModel(copy.deepcopy(reads), variants, force_homo=False, top_n=20, variant_correction=False)
    gpu = AlleleTyping(to_pairs(reads), variants, force_homo=False, top_n=20, variant_correction=False, device=device)
    want, got = cpu.typing(2), gpu.typing(2)
    assert gpu._model.miss8 is not None and not gpu._model.boundOk        # 4200 factors in one row: the flag is up
    assert np.isfinite(np.asarray(want.value)).all()                       # ... not because a product underflowed
    same(got, want)


"""GPU parity: read depth, copy-number model and the command line end to end vs the CPU oracle."""
import copy
import gzip
import io
import json
import os

import numpy as np
import pandas as pd
import pytest

from kir_graph_amd import synth
from kir_graph_amd.hisat2 import extractVariant, pairLines
from kir_graph_amd.index import GkIndex
from kir_graph_amd.kir_cn import depthToCN, predictSamplesCN, loadCN
from kir_graph_amd.samtools_utils import depthOfSample
from oracle import cn as ocn, depth as odepth, tabulate as ot, typing as oty

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_depth_matches_oracle(device, small_case):
    sidx, gidx, sample = small_case
    lines = synth.toSamLines(sample)
    data = extractVariant(pairLines(lines), gidx, dev=device)
    gene_len = {g: len(sidx.backbone[g]) for g in sidx.genes}
    df = depthOfSample(data, gene_len)
    kept = [(l, r, ot.nhOf(l)) for l, r in ot.pairMates(lines) if ot.passesFilter(l) and ot.passesFilter(r)]
    want = odepth.depthFromPairs(kept, gene_len)
    for g in sidx.genes:
        got = df[df["gene"] == g]["depth"].to_numpy()
        assert np.array_equal(got, want[g]), g
    assert list(df[df["gene"] == sidx.genes[0]]["pos"][:3]) == [1, 2, 3]


def test_copy_number_matches_reference_fixture(device):
    with gzip.open(os.path.join(GOLD, "t8_cn.json.gz"), "rt") as f:
        t8 = json.load(f)
    tables = [pd.DataFrame(rows, columns=["gene", "pos", "depth"]) for rows in t8["depth_tables"]]
    kw = {"base_dev": 0.08, "start_base": 2}
    for mode, res in t8["per_sample"].items():
        for df, want in zip(tables, res):
            depths = ocn.geneDepths(df, mode)
            cns, model = depthToCN([depths], cluster_method="LCND", cluster_method_kwargs=kw,
                                   assume_3DL3_diploid=True)
            ref = pd.read_csv(io.StringIO(want["tsv"]), sep="\t")
            assert {k: int(v) for k, v in cns[0].items()} == dict(zip(ref["gene"], ref["cn"]))   # integer CN: exact
            assert model.base == pytest.approx(float.fromhex(want["base"]), rel=1e-12)
            assert model.bin_num == want["bin_num"]
    for method, files in t8["cohort"].items():
        depths = [ocn.geneDepths(df, "p75") for df in tables]
        cns, _ = depthToCN(depths, cluster_method=method, cluster_method_kwargs=kw if method == "LCND" else {})
        for got, want in zip(cns, files):
            ref = pd.read_csv(io.StringIO(want), sep="\t")
            assert {k: int(v) for k, v in got.items()} == dict(zip(ref["gene"], ref["cn"]))


def test_likelihood_curve_close_to_scipy(device):
    rng = np.random.default_rng(3)
    values = list(np.concatenate([rng.normal(30 * k, 2 + k, 6) for k in (1, 2, 2, 3)]).clip(0))
    _, dev_model = depthToCN([dict(enumerate(values))], cluster_method="LCND",
                             cluster_method_kwargs={"base_dev": 0.08, "start_base": 2})
    _, cpu_model = ocn.depthsToCN([dict(enumerate(values))], "LCND", {"base_dev": 0.08, "start_base": 2})
    assert np.allclose(dev_model.likelihood, cpu_model.likelihood, rtol=1e-10, atol=1e-9)
    assert dev_model.base == cpu_model.base


def test_command_line_two_samples(device, tmp_path):
    """graphkir CLI on two synthetic samples: same files, same allele calls as the oracle pipeline."""
    from kir_graph_amd import main as cli
    sidx = synth.makeIndex(seed=11, n_genes=3, var_range=(200, 300), allele_range=(12, 20))
    folder = tmp_path / "index"
    folder.mkdir()
    prefix = str(folder / "kir_2100_withexon_ab_2dl1s1.leftalign.mut01")
    sidx.write(prefix)
    gidx = GkIndex.load(prefix)
    # 3DL3 must be present for the per-sample CN model: rename the last backbone is not needed in cohort mode
    sams, cns, samples = [], [], []
    for k in range(2):
        s = synth.makeSample(sidx, seed=50 + k, n_pairs=2500)
        path = tmp_path / f"s{k}.sam.gz"
        with gzip.open(path, "wt") as f:
            f.write("@HD\tVN:1.0\tSO:queryname\n" +
                    "".join(f"@SQ\tSN:{g}\tLN:{len(sidx.backbone[g])}\n" for g in sidx.genes) +
                    "\n".join(synth.toSamLines(s)) + "\n")
        cn_path = tmp_path / f"s{k}.cn.tsv"
        cn_path.write_text("gene\tcn\n" + "".join(f"{g}\t{c}\n" for g, c in s.gene_cn.items()))
        sams.append(str(path)); cns.append(str(cn_path)); samples.append(s)
    args = cli.createParser().parse_args(
        ["--step-skip-extraction", "--index-folder", str(folder), "--output-folder", str(tmp_path / "out"),
         "--allele-strategy", "pv", "--cn-provided", *cns, "--alignment", sams[0], "--alignment", sams[1]])
    cli.main(args)
    out = pd.read_csv(tmp_path / "out" / "cohort.allele.tsv", sep="\t")
    assert list(out.columns) == ["name", "alleles", "warnings"]
    assert len(out) == 2
    for k, s in enumerate(samples):
        ref = ot.tabulateLines(synth.toSamLines(s), gidx.variants)
        calls, warn = oty.makeTyper("full", copy.deepcopy(ref), top_n=600, variant_correction=True).typing(s.gene_cn)
        assert out["alleles"][k] == "_".join(calls)
        assert (out["warnings"][k] if isinstance(out["warnings"][k], str) else "") == "_".join(warn)
        assert out["name"][k].endswith(".full")
    cn = pd.read_csv(tmp_path / "out" / "cohort.cn.tsv", sep="\t", index_col=0)
    assert list(cn.columns) == cns
    # the per-sample artefacts of the reference's naming chain exist
    produced = sorted(p.name for p in (tmp_path / "out").iterdir())
    assert any(n.endswith(".variant.json") for n in produced)
    assert any(n.endswith(".variant.no_multi.depth.tsv") for n in produced)
    assert any(n.endswith(".full.possible.tsv") for n in produced)
    # the filtered alignments are rewritten as BAM like extractVariantFromBam does (hisat2.py:936-940)
    from kir_graph_amd import packed
    bams = sorted(n for n in produced if n.endswith(".variant.bam"))
    uniq = sorted(n for n in produced if n.endswith(".variant.no_multi.bam"))
    assert len(bams) == 2 and len(uniq) == 2
    for k, s in enumerate(samples):
        ref = ot.tabulateLines(synth.toSamLines(s), gidx.variants)
        got = b"".join(packed.bamChunks(str(tmp_path / "out" / bams[k]), name_sorted=False)).decode().split("\n")[:-1]
        assert len(got) == 2 * len(ref["reads"])
        pos = [(l.split("\t")[2], int(l.split("\t")[3])) for l in got]
        assert pos == sorted(pos, key=lambda x: (sidx.genes.index(x[0]), x[1]))      # coordinate-sorted
        only = b"".join(packed.bamChunks(str(tmp_path / "out" / uniq[k]), name_sorted=False)).decode().split("\n")[:-1]
        assert len(only) == 2 * sum(r["multiple"] == 1 for r in ref["reads"])


def test_command_line_reads_coordinate_sorted_bam(device, tmp_path):
    """A coordinate-sorted BAM goes through the native reader (no samtools) and gives the calls of the SAM text."""
    import sys
    sys.path.insert(0, os.path.dirname(__file__))
    from bamwriter import samToBam
    from kir_graph_amd import main as cli
    sidx = synth.makeIndex(seed=11, n_genes=3, var_range=(200, 300), allele_range=(12, 20))
    folder = tmp_path / "index"
    folder.mkdir()
    prefix = str(folder / "kir_2100_withexon_ab_2dl1s1.leftalign.mut01")
    sidx.write(prefix)
    s = synth.makeSample(sidx, seed=50, n_pairs=2500)
    lines = synth.toSamLines(s)
    header = ["@HD\tVN:1.0\tSO:coordinate"] + [f"@SQ\tSN:{g}\tLN:{len(sidx.backbone[g])}" for g in sidx.genes]
    by_coord = sorted(lines, key=lambda l: (l.split("\t")[2], int(l.split("\t")[3])))
    samToBam(header + by_coord, str(tmp_path / "s.bam"))
    with gzip.open(tmp_path / "s.sam.gz", "wt") as f:
        f.write("\n".join(lines) + "\n")
    cn_path = tmp_path / "s.cn.tsv"
    cn_path.write_text("gene\tcn\n" + "".join(f"{g}\t{c}\n" for g, c in s.gene_cn.items()))
    calls = []
    # BAM through the rendered text (JSON wanted), BAM in binary form (no JSON: compact hand-off), SAM text
    for k, (aln, extra) in enumerate((("s.bam", []), ("s.bam", ["--no-variant-json"]), ("s.sam.gz", []))):
        out = tmp_path / f"out{k}"
        args = cli.createParser().parse_args(
            ["--step-skip-extraction", "--index-folder", str(folder), "--output-folder", str(out),
             "--allele-strategy", "pv", "--cn-provided", str(cn_path), "--alignment", str(tmp_path / aln)] + extra)
        cli.main(args)
        calls.append(pd.read_csv(out / "cohort.allele.tsv", sep="\t")["alleles"][0])
        produced = [p.name for p in out.iterdir()]
        assert any(n.endswith(".variant.npz") for n in produced) == bool(extra)
        assert any(n.endswith(".variant.json") for n in produced) != bool(extra)
    assert calls[0] == calls[1] == calls[2] and "*" in calls[0]


def test_compact_side_format_round_trip(device, tmp_path):
    """hisat2.writeCompact / loadCompact: the CSR hand-off types like the in-memory tabulation."""
    from kir_graph_amd.hisat2 import extractVariantFromText, loadCompact, writeCompact
    from kir_graph_amd.kir_typing import selectKirTypingModel
    sidx = synth.makeIndex(seed=12, n_genes=3, var_range=(200, 300), allele_range=(12, 20))
    prefix = str(tmp_path / "idx")
    sidx.write(prefix)
    gidx = GkIndex.load(prefix)
    s = synth.makeSample(sidx, seed=8, n_pairs=3000)
    text = ("\n".join(synth.toSamLines(s)) + "\n").encode()
    data = extractVariantFromText([text], gidx, dev=device, keep_text=False)
    want = selectKirTypingModel("pv", data, top_n=600, variant_correction=True).typing(s.gene_cn)
    path = str(tmp_path / "s.variant.npz")
    writeCompact(data, path, index_ref=prefix)
    back = loadCompact(path, device)
    assert back.tab.n_valid == data.tab.n_valid and back.tab.n_novel == data.tab.n_novel
    assert np.array_equal(back.tab.ids(), data.tab.ids()) and np.array_equal(back.tab.offsets(), data.tab.offsets())
    assert [str(v.id) for v in back.novel] == [str(v.id) for v in data.novel]
    assert selectKirTypingModel("pv", path, top_n=600, variant_correction=True).typing(s.gene_cn) == want
    for strategy in ("exonfirst_1", "em"):
        a = selectKirTypingModel(strategy, data, top_n=600, variant_correction=True).typing(s.gene_cn)
        b = selectKirTypingModel(strategy, back, top_n=600, variant_correction=True).typing(s.gene_cn)
        assert a == b, strategy
    # a file made against another index is refused
    other = synth.makeIndex(seed=13, n_genes=3, var_range=(200, 300), allele_range=(12, 20))
    with pytest.raises(ValueError):
        loadCompact(path, device, index=GkIndex.fromVariants(other.variants, genes=other.genes, exons=other.exons))


def test_hand_off_files_through_three_lanes(device, tmp_path, monkeypatch):
    """main.alleleTyping with FILE entries and three sample lanes: every lane uploads its file on the process's default
    context, one at a time (kir_typing._sample holds a lock until the stream has drained); the files written are the ones
    a one-lane run writes."""
    from kir_graph_amd import main as cli
    from kir_graph_amd.hisat2 import extractVariantFromText, writeCompact
    sidx = synth.makeIndex(seed=21, n_genes=3, var_range=(200, 300), allele_range=(12, 20))
    prefix = str(tmp_path / "idx")
    sidx.write(prefix)
    gidx = GkIndex.load(prefix)
    names, cn_files = [], []
    for k in range(6):
        s = synth.makeSample(sidx, seed=30 + k, n_pairs=2500, gene_cn={g: 1 + (k + j) % 2 for j, g in enumerate(sidx.genes)})
        text = ("\n".join(synth.toSamLines(s)) + "\n").encode()
        data = extractVariantFromText([text], gidx, dev=device, keep_text=False)
        name = str(tmp_path / f"s{k}.variant")
        writeCompact(data, name + ".npz", index_ref=prefix)
        data.tab.close()
        cn = name + ".cn.tsv"
        with open(cn, "w") as f:
            f.write("gene\tcn\tdepth\n" + "".join(f"{g}\t{c}\t{30.0 * c}\n" for g, c in s.gene_cn.items()))
        names.append(name)
        cn_files.append(cn)
    out = {}
    for lanes in ("1", "3"):
        monkeypatch.setenv("GK_SAMPLE_LANES", lanes)
        files = cli.alleleTyping([(n, n + ".npz") for n in names], cn_files, method="full")
        out[lanes] = [open(f).read() for f in files] + [open(f[:-4] + ".possible.tsv").read() for f in files]
        assert len(files) == 6 and all("*" in text for text in out[lanes][:6])
    assert out["1"] == out["3"]


def test_records_side_format_round_trip(device, tmp_path):
    """hisat2.writeCompactRecords / loadCompact: the hand-off as the sample's packed records in compact form (what the
    command line writes with --no-variant-json) gives the tabulation it was written beside -- lists, novel numbering,
    pairs of the wide record format included -- and types like it."""
    from kir_graph_amd.hisat2 import extractVariantFromPacked, loadCompact, packAlignments, writeCompactRecords
    from kir_graph_amd.kir_typing import selectKirTypingModel
    from kir_graph_amd.packed import CompactMates
    sidx = synth.makeIndex(seed=14, n_genes=3, var_range=(200, 300), allele_range=(12, 20))
    prefix = str(tmp_path / "idx")
    sidx.write(prefix)
    gidx = GkIndex.load(prefix)
    s = synth.makeSample(sidx, seed=9, n_pairs=3000, err_rate=0.004)
    rng = np.random.default_rng(5)
    lines = synth.withManyMismatches(synth.toSamLines(s), sidx, rng.choice(s.n_pairs, size=5, replace=False).tolist(), rng)
    sam = tmp_path / "s.sam"
    sam.write_text("\n".join(lines) + "\n")
    pack = packAlignments(str(sam), gidx, keep_text=False)
    assert pack["counts"].get("spill") is not None and len(pack["counts"]["spill"][1]) > 0
    compact = CompactMates(pack["records"], threads=2)
    data = extractVariantFromPacked(pack, gidx, dev=device, mates=compact.toDevice(device, wait=True))
    path = str(tmp_path / "s.variant.npz")
    writeCompactRecords(compact, pack, data.tab.novel_base, gidx, path, index_ref=prefix)
    back = loadCompact(path, device)
    assert back.tab.n_valid == data.tab.n_valid and back.tab.n_novel == data.tab.n_novel and back.tab.n_novel > 0
    assert np.array_equal(back.tab.ids(), data.tab.ids()) and np.array_equal(back.tab.offsets(), data.tab.offsets())
    assert [str(v.id) for v in back.novel] == [str(v.id) for v in data.novel]
    want = selectKirTypingModel("pv", data, top_n=600, variant_correction=True).typing(s.gene_cn)
    assert selectKirTypingModel("pv", path, top_n=600, variant_correction=True).typing(s.gene_cn) == want
    other = synth.makeIndex(seed=13, n_genes=3, var_range=(200, 300), allele_range=(12, 20))
    with pytest.raises(ValueError):
        loadCompact(path, device, index=GkIndex.fromVariants(other.variants, genes=other.genes, exons=other.exons))

#!/usr/bin/env python3
"""The pieces of the path that stand in for samtools / HISAT2, against the real tools -- for a box that has them
(this image has neither: DESIGN.md section 2 lists these four as "parity unpinned").  Called by
tools/check_against_samtools.sh; every check prints PASS / FAIL / SKIP and the first difference.

  1. name collation   packed.bamChunks (native BAM reader + query-name order)   vs  `samtools sort -n -O SAM`
                      (graphkir/hisat2.py:103-110: the order of the lines fixes the order of the read pairs)
  2. read depth       samtools_utils.depthOfSample (gk_depth)                     vs  `samtools depth -aa` on the
                      .no_multi.bam the product writes (samtools_utils.py:9-22)
  3. pileup counts    pileup.pileupCounts (gk_bam_pileup)                         vs  `samtools mpileup -a`, parsed with
                      the restatement of parsePileupBase (pileup.py:13-55)
  4. end to end       `python -m kir_graph_amd.main` on example/test00 + test01    vs  the reference CLI
                      (needs hisat2, the example_index and GK_REFERENCE_DIR=<clone of linnil1/KIR_graph>)

    python tests/check_against_samtools.py [--bam aligned.bam --index-prefix <...leftalign.mut01>] [--example DIR]

Without --bam a synthetic sample is rendered (synth.py) and written as a coordinate-sorted BAM with the product's own
writer -- the comparison then covers the readers, not the writer's view of real HISAT2 records; with the BAM of a real
HISAT2 run (`hisat2 ... | samtools sort`) it covers real Zs / MD shapes too."""
import argparse
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
RESULTS = []


def report(name, ok, detail=""):
    RESULTS.append((name, ok))
    print(f"[{'PASS' if ok else 'FAIL' if ok is False else 'SKIP'}] {name}" + (f": {detail}" if detail else ""), flush=True)


def sh(cmd):
    return subprocess.run(cmd, capture_output=True, text=True, check=True).stdout


def key(line):
    c = line.split("\t")
    return c[0], int(c[1]), c[2], int(c[3])


def check_collation(bam):
    from kir_graph_amd.packed import bamChunks
    mine = [l for chunk in bamChunks(bam) for l in chunk.decode().split("\n") if l and not l.startswith("@")]
    theirs = [l for l in sh(["samtools", "sort", "-n", bam, "-O", "SAM"]).split("\n") if l and not l.startswith("@")]
    if len(mine) != len(theirs):
        return report("name collation", False, f"{len(mine)} lines here, {len(theirs)} from samtools sort -n")
    for i, (a, b) in enumerate(zip(mine, theirs)):
        if key(a) != key(b):
            return report("name collation", False, f"line {i}: {key(a)} here, {key(b)} from samtools")
    same_text = sum(a == b for a, b in zip(mine, theirs))
    report("name collation", True, f"{len(mine)} records in samtools' order; {same_text} of them identical as SAM text")
    # the pairs the reference would form from either stream (hisat2.py:228-276) are then the same as well
    from kir_graph_amd.hisat2 import pairLines
    pa, pb = list(pairLines(mine)), list(pairLines(theirs))
    report("pairing of the collated stream", [tuple(map(key, p)) for p in pa] == [tuple(map(key, p)) for p in pb],
           f"{len(pa)} pairs")


def check_depth(bam, index_prefix, tmp):
    import pandas as pd
    from kir_graph_amd._lib import Device
    from kir_graph_amd.hisat2 import extractVariantFromBam
    from kir_graph_amd.samtools_utils import depthOfSample, readLocusLengths
    dev = Device()
    out = os.path.join(tmp, "sample")
    data = extractVariantFromBam(index_prefix, bam, out, error_correction=False, dev=dev)     # writes out.no_multi.bam
    mine = depthOfSample(data, readLocusLengths(index_prefix))
    sh(["samtools", "index", out + ".no_multi.bam"])
    theirs = pd.read_csv(__import__("io").StringIO(sh(["samtools", "depth", "-aa", out + ".no_multi.bam"])), sep="\t",
                         header=None, names=["gene", "pos", "depth"])
    merged = mine.merge(theirs, on=["gene", "pos"], how="outer", suffixes=("_here", "_samtools")).fillna(-1)
    bad = merged[merged["depth_here"] != merged["depth_samtools"]]
    report("read depth (-aa, NH == 1 pairs)", bad.empty, f"{len(merged)} positions" if bad.empty else
           f"{len(bad)} of {len(merged)} positions differ, first: {bad.iloc[0].to_dict()}")
    data.tab.close()


def check_pileup(bam, index_prefix):
    from kir_graph_amd import pileup
    from kir_graph_amd.index import GkIndex
    from oracle.pileup import basesOfColumn
    idx = GkIndex.load(index_prefix)
    counts, pos0 = pileup.pileupCounts(bam, idx)
    bad = n = 0
    first = ""
    for line in sh(["samtools", "mpileup", "-a", bam]).split("\n"):
        if not line:
            continue
        ref, pos, _, depth, column = line.split("\t")[:5]
        g = idx.gene_id.get(ref)
        if g is None:
            continue
        want = [0] * 6
        for b in basesOfColumn(column).upper():
            want[pileup.BASES.index(b)] += 1
        got = counts[pos0[g] + int(pos) - 1].tolist()
        n += 1
        if got != want:
            bad += 1
            first = first or f"{ref}:{pos} here {got} mpileup {want}"
    report("pileup base counts", bad == 0, f"{n} positions" if not bad else f"{bad} of {n} positions differ, first: {first}")


def check_example(example, reference_dir, tmp):
    index = os.path.join(example, "..", "example_index")
    if not (shutil.which("hisat2") and os.path.isdir(index) and reference_dir):
        return report("example/test00 + test01 end to end", None, "needs hisat2, example_index and GK_REFERENCE_DIR")
    outs = {}
    for who, cmd in (("reference", [sys.executable, "-m", "graphkir.main"]), ("here", [sys.executable, "-m", "kir_graph_amd.main"])):
        out = os.path.join(tmp, who)
        env = dict(os.environ, PYTHONPATH=(reference_dir if who == "reference" else ROOT))
        res = subprocess.run(cmd + ["--input-csv", os.path.join(example, "cohort.csv"), "--index-folder", index,
                                    "--output-folder", out, "--output-cohort-name", os.path.join(out, "cohort")],
                             env=env, capture_output=True, text=True)
        if res.returncode:
            return report("example/test00 + test01 end to end", None, f"the {who} command line failed: {res.stderr[-300:]}")
        outs[who] = out
    same = all(open(os.path.join(outs["here"], f)).read().replace(outs["here"], "@") ==
               open(os.path.join(outs["reference"], f)).read().replace(outs["reference"], "@")
               for f in ("cohort.allele.tsv", "cohort.cn.tsv"))
    report("example/test00 + test01 end to end", same, "cohort.allele.tsv and cohort.cn.tsv")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bam")
    ap.add_argument("--index-prefix")
    ap.add_argument("--example", help="the reference's example/ directory (test00, test01, cohort.csv)")
    ap.add_argument("--pairs", type=int, default=20000)
    args = ap.parse_args()
    if not shutil.which("samtools"):
        print("[SKIP] samtools is not installed: nothing to compare against")
        return 0
    tmp = tempfile.mkdtemp(prefix="gk_vs_samtools_")
    try:
        bam, prefix = args.bam, args.index_prefix
        if not bam:
            from kir_graph_amd import packed, synth
            sidx = synth.makeIndex(seed=2022, n_genes=6)
            prefix = os.path.join(tmp, "kir_2100_withexon_ab_2dl1s1.leftalign.mut01")
            sidx.write(prefix)
            s = synth.makeSample(sidx, seed=7, n_pairs=args.pairs, err_rate=0.004, frac_multi=0.1)
            header = ["@HD\tVN:1.0\tSO:coordinate"] + [f"@SQ\tSN:{g}\tLN:{len(sidx.backbone[g])}" for g in sidx.genes]
            bam = os.path.join(tmp, "synthetic.bam")
            packed.writeBam(bam, "\n".join(header + synth.toSamLines(s)) + "\n")
        check_collation(bam)
        if prefix:
            check_depth(bam, prefix, tmp)
            check_pileup(bam, prefix)
        else:
            report("read depth / pileup counts", None, "--index-prefix needed with --bam")
        if args.example:
            check_example(args.example, os.environ.get("GK_REFERENCE_DIR"), tmp)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return 1 if any(ok is False for _, ok in RESULTS) else 0


if __name__ == "__main__":
    sys.exit(main())

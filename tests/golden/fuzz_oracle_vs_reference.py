#!/usr/bin/env python3
"""
Randomised run of the CPU oracle against the REFERENCE itself (build container only, like
make_golden.py: the reference is imported from /root/reference with the two stand-in modules and is
never copied; not collected by pytest, nothing here travels to the GPU box).

Same generator as tests/fuzz_parity.py (random indices of 1-3 genes, 3-100 alleles, 40-4000 pairs,
copy numbers 0-4, sequencing errors, multi-mapping shares, top_n 5 / 60 / 600, correction on / off).
Compared per case, all exactly (same host, same numpy):

* tabulation: the four id lists, NH and backbone of every pair, the variant list incl. novel ids/order
  (hisat2.extractVariant 803-844);
* ``full``, ``exonfirst_1``, ``exonfirst_0.9``: the calls, the warnings and every field of every
  copy-number step (typing_mulit_allele.py:478-598, 622-797);
* ``em``: alleles, counts and abundances of the report (typing_em.py:107-215), 1e-5 relative.

    python tests/golden/fuzz_oracle_vs_reference.py [seconds] [first_seed]  > profiles/rNN_fuzz_oracle_vs_reference.txt
"""
from __future__ import annotations

import contextlib
import copy
import io
import logging
import os
import sys
import tempfile
import time
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.dont_write_bytecode = True


def _stub(name, attrs=()):
    m = types.ModuleType(name)
    for a in attrs:
        setattr(m, a, type(a, (), {}))
    sys.modules[name] = m
    return m


_stub("pyhlamsa", ["Genemsa", "KIRmsa"])
_bio = _stub("Bio")
for _sub in ("SeqIO", "SeqRecord", "Align", "Seq"):
    setattr(_bio, _sub, _stub("Bio." + _sub, ["SeqRecord", "MultipleSeqAlignment", "Seq"]))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, ROOT)

import graphkir.hisat2 as rh                      # noqa: E402
import graphkir.kir_typing as rkt                 # noqa: E402
from graphkir.msa2hisat import Variant as RV      # noqa: E402

from kir_graph_amd import synth                   # noqa: E402
from oracle import em as oem, tabulate as ot, typing as oty   # noqa: E402

logging.getLogger("graphkir").setLevel(logging.ERROR)


def one(seed: int) -> str:
    rng = np.random.default_rng(seed)
    n_genes = int(rng.integers(1, 4))
    a_lo = int(rng.choice([3, 12, 40, 70]))
    sidx = synth.makeIndex(seed=seed, n_genes=n_genes, var_range=(60, 400),
                           allele_range=(a_lo, a_lo + int(rng.integers(1, 30))), len_range=(2500, 6000),
                           frac_del=float(rng.choice([0.0, 0.09, 0.2])), frac_ins=float(rng.choice([0.0, 0.03, 0.1])))
    gene_cn = {g: int(rng.choice([0, 1, 2, 2, 3, 4])) for g in sidx.genes}
    if not any(gene_cn.values()):
        gene_cn[sidx.genes[0]] = 2
    sample = synth.makeSample(sidx, seed=seed + 1, n_pairs=int(rng.choice([40, 300, 1500, 4000])), gene_cn=gene_cn,
                              err_rate=float(rng.choice([0.0, 0.001, 0.01])),
                              frac_multi=float(rng.choice([0.0, 0.05, 0.3])))
    lines = synth.toSamLines(sample)
    if os.environ.get("GK_FUZZ_SPILL") == "1":   # a few pairs with 17-60 mismatches per mate: what the HIP path keeps in its wide record format
        lines = synth.withManyMismatches(lines, sidx, rng.choice(sample.n_pairs, size=min(6, sample.n_pairs), replace=False).tolist(), rng)
    d = tempfile.mkdtemp()
    sidx.write(d + "/ix")
    rv = rh.getVariants(d + "/ix")

    # ---- tabulation
    rh.readBam = lambda f: lines
    kept = [p for p in rh.readPair("x") if rh.filterRead(p[0]) and rh.filterRead(p[1])]
    RV.novel_id = 0
    data = rh.extractVariant(kept, rv)
    mine = ot.tabulateLines(lines, sidx.variants)
    assert len(mine["reads"]) == len(data["reads"]), "pair count"
    for a, b in zip(mine["reads"], data["reads"]):
        assert (a["lpv"], a["lnv"], a["rpv"], a["rnv"], a["multiple"], a["backbone"]) == \
               (b.lpv, b.lnv, b.rpv, b.rnv, b.multiple, b.backbone), "lists"
    assert [(str(v.id), v.typ, v.pos, v.val, v.ref) for v in mine["variants"]] == \
           [(str(v.id), v.typ, v.pos, v.val, v.ref) for v in data["variants"]], "variant list"

    # ---- typing
    js = d + "/s.variant.json"
    rh.writeReadsAndVariantsData(data, js)
    top_n = int(rng.choice([5, 60, 600]))
    corr = bool(rng.integers(0, 2))
    note = ""
    for method in ("full", "exonfirst_1", "exonfirst_0.9"):
        ref = rkt.selectKirTypingModel(method, js, top_n=top_n, variant_correction=corr)
        cpu = oty.makeTyper(method, copy.deepcopy(mine), top_n=top_n, variant_correction=corr)
        try:
            with contextlib.redirect_stdout(io.StringIO()):
                want = ref.typing(gene_cn)
        except np.exceptions.AxisError:
            note = " (reference AxisError on an empty gene)"
            try:
                cpu.typing(gene_cn)
                raise AssertionError(f"{method}: the oracle did not raise where the reference does")
            except np.exceptions.AxisError:
                continue
        got = cpu.typing(gene_cn)
        assert got == want, f"{method} calls"
        for gene, steps in ref._result.items():
            assert len(cpu.results[gene]) == len(steps), f"{method} {gene} steps"
            for x, y in zip(cpu.results[gene], steps):
                for f in ("value", "value_sum_indv", "allele_id", "fraction", "fraction_uniq"):
                    assert np.array_equal(np.asarray(getattr(x, f)), np.asarray(getattr(y, f))), f"{method} {gene} {f}"
                assert [list(r) for r in x.allele_name] == [list(r) for r in y.allele_name], f"{method} {gene} names"
    ref = rkt.selectKirTypingModel("em", js)
    cpu = oem.ReportTyper(copy.deepcopy(mine))
    try:
        want = ref.typing(gene_cn)
    except np.exceptions.AxisError:
        try:
            cpu.typing(gene_cn)
            raise AssertionError("em: the oracle did not raise where the reference does")
        except np.exceptions.AxisError:
            return f"genes {n_genes} pairs {sample.n_pairs}: em AxisError in both"
    got = cpu.typing(gene_cn)
    assert sorted(got[0]) == sorted(want[0]) and got[1] == want[1], "em calls"
    for gene, report in ref._result.items():
        a = {r["allele"]: (r["count"], r["prob"]) for r in cpu.results[gene]}
        b = {r.allele: (r.count, r.prob) for r in report}
        assert a.keys() == b.keys(), "em alleles"
        for k in a:
            assert a[k][0] == b[k][0] and abs(a[k][1] - b[k][1]) <= 1e-5 * abs(b[k][1]) + 1e-12, "em abundance"
    return (f"genes {n_genes} alleles>={a_lo} pairs {sample.n_pairs} top_n {top_n} corr {corr} "
            f"cn {list(gene_cn.values())}{note}")


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    print(f"numpy {np.__version__}; oracle vs reference, first seed {seed}", flush=True)
    t0, n = time.time(), 0
    while time.time() - t0 < budget:
        try:
            info = one(seed)
        except AssertionError as e:
            print(f"MISMATCH seed {seed}: {e}", flush=True)
            raise
        n += 1
        if n % 10 == 0:
            print(f"[fuzz] {n} cases ok ({time.time() - t0:.0f}s), last: seed {seed}: {info}", flush=True)
        seed += 1
    print(f"[fuzz] {n} cases (seeds {seed - n}..{seed - 1}), all equal to the reference")


if __name__ == "__main__":
    main()

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

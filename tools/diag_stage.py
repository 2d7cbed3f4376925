"""Dev tool: the staging of a sample step by step, each checked on the host (used to localise a fault).
usage: python tools/diag_stage.py expand|tab|type [pairs]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from kir_graph_amd import _lib, packed
from kir_graph_amd.engine import DeviceIndex, Tabulation
from kir_graph_amd.hisat2 import SampleData
from kir_graph_amd.kir_typing import selectKirTypingModel

mode = sys.argv[1]
pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
sidx, gidx, by_gene = bench.build_index()
sample, rec, table = bench.build_sample(sidx, gidx, by_gene, 1031, pairs)
dev = _lib.Device(0)
cm = packed.CompactMates(rec, threads=4)
print("compact", cm.nbytes, "of", rec.nbytes, flush=True)
mates = cm.toDevice(dev, wait=True)
print("expanded", flush=True)
back = mates.download()
for f in ("pos0", "flag", "ref", "nh", "nm", "n_cig", "n_mm", "n_ins"):
    assert np.array_equal(back[f], rec[f]), f
k = np.arange(14)[None, :] < rec["n_cig"][:, None]
assert np.array_equal(np.where(k, back["cig"], 0), np.where(k, rec["cig"], 0))
print("records equal where used", flush=True)
if mode == "expand":
    sys.exit(0)
dindex = DeviceIndex(dev, gidx)
tab = Tabulation(dindex, mates)
ref = Tabulation(dindex, rec)
assert (tab.n_valid, tab.n_ids, tab.n_novel) == (ref.n_valid, ref.n_ids, ref.n_novel)
assert np.array_equal(tab.ids(), ref.ids()) and np.array_equal(tab.offsets(), ref.offsets())
print("tabulation equal", tab.n_valid, tab.n_ids, flush=True)
if mode == "tab":
    sys.exit(0)
data = SampleData(tab, gidx, None, ins_strings=table.strings)
typer = selectKirTypingModel("full", data, top_n=600, variant_correction=True)
calls = typer.typing(sample.gene_cn)
print("typed", len(calls[0]), "calls", flush=True)

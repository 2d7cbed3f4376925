#!/usr/bin/env python3
"""Dev tool: where one process spends a sample's typing time -- cProfile of the typing of a few bench samples on ONE
native calls (ctypes) against interpreter time."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from kir_graph_amd import _lib
from kir_graph_amd.engine import DeviceIndex, Tabulation
from kir_graph_amd.hisat2 import SampleData
from kir_graph_amd.kir_typing import selectKirTypingModel

dev = _lib.Device(0)
sidx, gidx, by_gene = bench.build_index()
sample, rec, table = bench.build_sample(sidx, gidx, by_gene, 1031, 1_000_000)
dindex = DeviceIndex(dev, gidx)
mates = dev.put(rec)

def one():
    tab = Tabulation(dindex, mates)
    data = SampleData(tab, gidx, None, ins_strings=table.strings)
    typer = selectKirTypingModel("pv", data, top_n=600, variant_correction=True)
    typer.typing(sample.gene_cn)
    tab.close()

for _ in range(3):
    one()
n = 8
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
for _ in range(n):
    one()
pr.disable()
wall = (time.perf_counter() - t0) / n * 1e3
print(f"wall {wall:.2f} ms per sample (one gene thread, profiler on)")
st = pstats.Stats(pr)
native = sum(v[2] for k, v in st.stats.items() if k[0] == "~" and "_ctypes" in k[2] or "CFuncPtr" in k[2] or "callproc" in k[2])
total = sum(v[2] for v in st.stats.values())
print(f"tottime in all functions {total / n * 1e3:.2f} ms per sample")
st.sort_stats("tottime").print_stats(28)

#!/usr/bin/env python3
"""Dev tool: cProfile of one single-threaded bench step sorted by own time (host hot spots)."""
import cProfile, io, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from kir_graph_amd import _lib, kir_typing
from kir_graph_amd.engine import DeviceIndex, Tabulation
from kir_graph_amd.hisat2 import SampleData

pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dev = _lib.Device(0)
sidx, gidx, sample, rec, table = bench.build_inputs(1031, pairs)
dindex = DeviceIndex(dev, gidx)
mates = dev.put(rec)

def step():
    tab = Tabulation(dindex, mates)
    data = SampleData(tab, gidx, None, ins_strings=table.strings)
    typer = kir_typing.selectKirTypingModel(os.environ.get("GK_METHOD", "pv"), data, top_n=600, variant_correction=True)
    typer.typing(sample.gene_cn)
    tab.close()

for _ in range(3):
    step()
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    step()
pr.disable()
out = io.StringIO()
pstats.Stats(pr, stream=out).sort_stats("tottime").print_stats(40)
print(out.getvalue())

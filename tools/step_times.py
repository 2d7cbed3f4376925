#!/usr/bin/env python3
"""Dev tool: completion time of every step of one bench-like run (variance between steps)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from kir_graph_amd import _lib
from kir_graph_amd.cohort import overlapped, prefetched
from kir_graph_amd.engine import DeviceIndex, Tabulation
from kir_graph_amd.hisat2 import SampleData
from kir_graph_amd.kir_typing import hostThreads, selectKirTypingModel

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = _lib.Device(0)
sidx, gidx, sample, rec, table = bench.build_inputs(1031, 1_000_000)
dindex = DeviceIndex(dev, gidx)
mates = dev.put(rec)
ingest = dev.worker(hostThreads())

def type_one(tab, lane):
    data = SampleData(tab, gidx, None, ins_strings=table.strings)
    typer = selectKirTypingModel("pv", data, top_n=600, variant_correction=True)
    typer.typing(sample.gene_cn)
    tab.close()
    return time.perf_counter()

tabs = prefetched(range(n), lambda _: Tabulation(dindex, mates, dev=ingest), depth=1)
t0 = time.perf_counter()
stamps = [t0] + [t for t in overlapped(tabs, type_one, lanes=1)]
d = np.diff(stamps) * 1e3
print("steps ms:", " ".join(f"{x:.1f}" for x in d))
print(f"median {np.median(d[3:]):.2f}  mean {d[3:].mean():.2f}  min {d[3:].min():.2f}  max {d[3:].max():.2f}")

#!/usr/bin/env python3
"""Dev tool: wall time of the two halves of a bench step in one process -- stage (pinned H2D + tabulation)
and typing -- run back to back, then pipelined (what bounds a single process?)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from kir_graph_amd import _lib
from kir_graph_amd.engine import DeviceIndex, Tabulation
from kir_graph_amd.hisat2 import SampleData
from kir_graph_amd.kir_typing import hostThreads, selectKirTypingModel

dev = _lib.Device(0)
sidx, gidx, by_gene = bench.build_index()
inputs = []
for i in range(3):
    sample, rec, table = bench.build_sample(sidx, gidx, by_gene, 1031 + i, 1_000_000)
    inputs.append((bench.PinnedRecords(rec), table, sample.gene_cn))
dindex = DeviceIndex(dev, gidx)
ingest = dev.worker(hostThreads())

def stage(k):
    pinned, table, gene_cn = inputs[k % 3]
    t0 = time.perf_counter()
    mates = pinned.toDevice(ingest)
    ingest.sync()
    t1 = time.perf_counter()
    tab = Tabulation(dindex, mates, dev=ingest)
    t2 = time.perf_counter()
    return (tab, table, gene_cn), (t1 - t0) * 1e3, (t2 - t1) * 1e3

def type_one(item):
    tab, table, gene_cn = item
    t0 = time.perf_counter()
    data = SampleData(tab, gidx, None, ins_strings=table.strings)
    typer = selectKirTypingModel("pv", data, top_n=600, variant_correction=True)
    typer.typing(gene_cn)
    tab.close(); tab.mates.free()
    return (time.perf_counter() - t0) * 1e3

for k in range(3):
    it, a, b = stage(k); type_one(it)
rows = []
for k in range(9):
    it, a, b = stage(k)
    c = type_one(it)
    rows.append((a, b, c))
r = np.array(rows)
print("per sample ms: h2d %.2f  tabulate %.2f  typing %.2f" % tuple(r.mean(axis=0)))
print("by sample (h2d, tab, typing):", np.round(r[:3], 2).tolist())
t0 = time.perf_counter()
out = bench.run_steps(12, dev, dindex, gidx, inputs, "pv")
print("pipelined: %.2f ms per step" % ((time.perf_counter() - t0) * 1e3 / 12))

#!/usr/bin/env python3
"""Dev tool: the tabulation stage alone (gk_tabulate) on the bench workload, per-kernel times."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from kir_graph_amd import _lib
from kir_graph_amd.engine import DeviceIndex, Tabulation

pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = _lib.Device(0)
sidx, gidx, sample, rec, table = bench.build_inputs(1031, pairs)
dindex = DeviceIndex(dev, gidx)
mates = dev.put(rec)
for _ in range(2):
    Tabulation(dindex, mates).close()
dev.profEnable(True)
dev.profCollect()
t = time.perf_counter()
for _ in range(reps):
    tab = Tabulation(dindex, mates)
    info = (tab.n_valid, tab.n_ids, tab.n_novel)
    tab.close()
wall = (time.perf_counter() - t) / reps
print("valid pairs, ids, novel:", info)
for k, (n, ms) in dev.profCollect().items():
    print(f"{k:16s} {n / reps:5.1f} launches/step  {ms / reps:8.3f} ms/step")
print(f"wall {wall * 1e3:.2f} ms per tabulation of {pairs} pairs")

#!/usr/bin/env python3
"""Dev tool: the shapes of a bench sample's search steps -- per gene and step: rows, previous sets, candidates, sets the
integer bound selects (= sets whose exact float64 sums are formed), their distinct columns."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from kir_graph_amd import _lib
from kir_graph_amd.engine import DeviceIndex, Tabulation
from kir_graph_amd.hisat2 import SampleData
from kir_graph_amd.kir_typing import selectKirTypingModel

pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dev = _lib.Device(0)
sidx, gidx, sample, rec, table = bench.build_inputs(1031, pairs)
dindex = DeviceIndex(dev, gidx)
tab = Tabulation(dindex, dev.put(rec))
data = SampleData(tab, gidx, None, ins_strings=table.strings)
selectKirTypingModel("pv", data, top_n=600, variant_correction=True).typing(sample.gene_cn)   # creates the worker contexts
for d in _lib.Device.instances:
    d.call_log = []
typer = selectKirTypingModel("pv", data, top_n=600, variant_correction=True)
typer.typing(sample.gene_cn)
log = [c for d in _lib.Device.instances for c in (d.call_log or [])]
for c in log:
    if c[0] == "minsum_sad":
        print(f"bound   rows {c[1]:7d}  prev sets {c[2]:4d}  candidate alleles {c[3]:4d}  prev columns {c[4]:4d}")
    elif c[0] == "fraction_chunks":
        print(f"setsum  rows {c[1]:7d}  sets {c[2]:5d}  alleles per set {c[3]}  distinct columns {c[4]:4d}")
    elif c[0] == "compat_kernel":
        print(f"compat  rows {c[1]:7d}  alleles {c[2]:4d}  ids {int(c[3]):9d}")
sel = [c[2] for c in log if c[0] == "fraction_chunks"]
print(f"{len(sel)} set-sum launches, sets per launch: mean {np.mean(sel):.0f}, max {max(sel)}; genes {sample.gene_cn}")

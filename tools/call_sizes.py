#!/usr/bin/env python3
"""Dev tool: launch geometry of one bench step (maxsum / fraction calls per gene)."""
import os, sys
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
tab = Tabulation(dindex, mates)
data = SampleData(tab, gidx, None, ins_strings=table.strings)
dev.call_log = []
dev.worker(0).call_log = []      # single-threaded typing runs on worker 0
typer = kir_typing.selectKirTypingModel(os.environ.get("GK_METHOD", "pv"), data, top_n=600, variant_correction=True)
typer.typing(sample.gene_cn)
for d in _lib.Device.instances:
    for c in d.call_log or []:
        print(c)

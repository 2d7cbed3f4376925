#!/usr/bin/env python3
"""Dev tool: wall-clock phases of one bench step (tabulation vs per-gene typing) at 1 M pairs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from kir_graph_amd import _lib
from kir_graph_amd.engine import DeviceIndex, Tabulation
from kir_graph_amd.hisat2 import SampleData
from kir_graph_amd import kir_typing

pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dev = _lib.Device(0)
sidx, gidx, sample, rec, table = bench.build_inputs(1031, pairs)
dindex = DeviceIndex(dev, gidx)
mates = dev.put(rec)
orig = kir_typing.TypingWithPosNegAllele.typingPerGene
times = {}
def timed(self, gene, cn):
    t = time.perf_counter(); r = orig(self, gene, cn); times[gene] = (time.perf_counter() - t, cn, t - T0[0]); return r
kir_typing.TypingWithPosNegAllele.typingPerGene = timed
T0 = [0.0]
for it in range(3):
    times.clear()
    t0 = time.perf_counter(); T0[0] = t0
    tab = Tabulation(dindex, mates)
    t1 = time.perf_counter()
    data = SampleData(tab, gidx, None, ins_strings=table.strings)
    typer = kir_typing.selectKirTypingModel("pv", data, top_n=600, variant_correction=True)
    typer.typing(sample.gene_cn)
    t2 = time.perf_counter()
    tab.close()
    print(f"iter {it}: tabulate {1e3*(t1-t0):.1f} ms, typing {1e3*(t2-t1):.1f} ms, total {1e3*(t2-t0):.1f} ms")
for g, (dt, cn, st) in sorted(times.items(), key=lambda kv: kv[1][2]):
    A = gidx.tables[gidx.gene_id[g]].n_allele
    print(f"  {g:22s} A={A:3d} cn={cn} start {1e3*st:6.1f} ms  took {1e3*dt:6.1f} ms")

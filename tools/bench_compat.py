#!/usr/bin/env python3
"""Dev tool: the compatibility stage (gk_compat_log) of every gene of the bench sample, per-kernel times."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from kir_graph_amd import _lib
from kir_graph_amd.engine import DeviceIndex, DeviceModel, LogTable, Tabulation

pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dev = _lib.Device(0)
sidx, gidx, sample, rec, table = bench.build_inputs(1031, pairs)
dindex = DeviceIndex(dev, gidx)
tab = Tabulation(dindex, dev.put(rec))
logs = LogTable(dev)
models = []
def run(products_only=False):
    for g, t in enumerate(gidx.tables):
        rows, n = tab.selectGene(g)
        vflag = dev.alloc(tab.n_var_total, np.uint8).zero()
        dm = DeviceModel(tab, rows, n, vflag, t.vbeg, t.vend, dindex.masks[g], t.words, t.n_allele, logs)
        dm.finishLog()
        if products_only:
            dm.probs          # gk_compat: the ordered products alone (no value table, no mismatch bytes)
        dm.free(); rows.free(); vflag.free()

def measure(label, **kw):
    run(**kw)
    dev.profEnable(True); dev.profCollect()
    t0 = time.perf_counter()
    for _ in range(3):
        run(**kw)
    dev.sync()
    wall = (time.perf_counter() - t0) / 3
    print(f"--- {label}")
    for k, (n, ms) in dev.profCollect().items():
        print(f"{k:16s} {n / 3:5.1f} launches/step  {ms / 3:8.3f} ms/step")
    print(f"wall {wall * 1e3:.2f} ms")
    dev.profEnable(False)

os.environ["GK_SEARCH"] = "bound"
measure("log-likelihoods + u8 mismatch table (gk_compat_log_miss): the product path")
os.environ["GK_SEARCH"] = "exact"
measure("log-likelihoods only (gk_compat_log)")
measure("log-likelihoods, then the plain products once more (gk_compat): compat_kernel = both", products_only=True)

#!/usr/bin/env python3
"""Dev tool: how many read pairs of a gene share their four id lists (= identical rows of the compatibility table).
  python tools/dup_rows.py [pairs, default 10000000]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                            # noqa: E402
from kir_graph_amd import _lib                          # noqa: E402
from kir_graph_amd.engine import DeviceIndex, Tabulation   # noqa: E402

pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
dev = _lib.Device(0)
sidx, gidx, by_gene = bench.build_index()
t = time.time()
sample, rec, table = bench.build_sample(sidx, gidx, by_gene, 1031, pairs)
print(f"sample of {pairs} pairs in {time.time() - t:.1f}s", flush=True)
dindex = DeviceIndex(dev, gidx)
mates = dev.put(rec)
tab = Tabulation(dindex, mates, dev=dev)
off = tab.offsets().astype(np.int64)
ids = tab.ids().astype(np.uint64)
gene = tab.pairGene()
nh = tab.pairNH()
n = tab.n_valid
print(f"valid {n}, ids {len(ids)} ({len(ids) / n:.1f} per pair)", flush=True)
# an order-independent hash per list (lists are sorted), the four lists mixed with different multipliers
h = (ids + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15)
h ^= h >> np.uint64(29)
h *= np.uint64(0xBF58476D1CE4E5B9)
h ^= h >> np.uint64(32)
csum = np.concatenate([[np.uint64(0)], np.cumsum(h, dtype=np.uint64)])
per_list = csum[off[1:]] - csum[off[:-1]]               # [4 * n]
per_list = per_list.reshape(n, 4)
lens = np.diff(off).reshape(n, 4).astype(np.uint64)
key = np.zeros(n, dtype=np.uint64)
for q, m in enumerate((0x94D049BB133111EB, 0xD6E8FEB86659FD93, 0xA0761D6478BD642F, 0xE7037ED1A0B428DB)):
    key += (per_list[:, q] + lens[:, q] * np.uint64(0x632BE59BD9B4E019)) * np.uint64(m)
tot_r = tot_d = 0
for g in range(len(gidx.genes)):
    sel = (gene == g) & (nh == 1)
    k = key[sel]
    if not len(k):
        continue
    u, c = np.unique(k, return_counts=True)
    top = np.sort(c)[::-1]
    print(f"gene {g:2d} {gidx.genes[g]:10s} rows {len(k):8d} distinct {len(u):8d} ({len(u) / len(k):.3f}); "
          f"largest classes {top[:4].tolist()}; rows in classes >= 2: {int(c[c >= 2].sum()) / len(k):.3f}", flush=True)
    tot_r += len(k)
    tot_d += len(u)
print(f"all genes: rows {tot_r}, distinct {tot_d} ({tot_d / tot_r:.3f})")

#!/usr/bin/env python3
"""Micro-benchmark of gk_maxsum / gk_fraction / gk_compat-sized problems on random data (dev tool)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kir_graph_amd import _lib
from kir_graph_amd._lib import lib, check

R = int(sys.argv[1]) if len(sys.argv) > 1 else 67000
A = int(sys.argv[2]) if len(sys.argv) > 2 else 160
T = int(sys.argv[3]) if len(sys.argv) > 3 else 160
C = int(sys.argv[4]) if len(sys.argv) > 4 else 1
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 10
dev = _lib.Device(0)
rng = np.random.default_rng(0)
L = -rng.integers(0, 6, (A, R)).astype(np.float64) * 3.0 - rng.integers(20, 60, (1, R)) * 0.000434
dL = dev.put(L)
ids = np.ascontiguousarray(rng.integers(0, A, (T, C)), dtype=np.int32)
cols = np.arange(A, dtype=np.int32)
if os.environ.get("GK_BENCH_SAMECOL"):   # every tile reads the same column: cache-resident operands (latency experiment)
    cols[:] = 0
    ids[:] = 0
out = np.empty((T, A))
dev.profEnable(True)
for _ in range(2):
    check(lib().gk_maxsum(dev.ctx, dL.ptr, R, R, ids.ctypes.data, T, C, cols.ctypes.data, A, out.ctypes.data))
dev.profCollect()
t = time.perf_counter()
for _ in range(reps):
    check(lib().gk_maxsum(dev.ctx, dL.ptr, R, R, ids.ctypes.data, T, C, cols.ctypes.data, A, out.ctypes.data))
wall = (time.perf_counter() - t) / reps
prof = dev.profCollect()
ops = 2.0 * R * T * A
for k, (n, ms) in prof.items():
    print(f"{k:16s} avg {ms / n:8.4f} ms")
ms = prof["maxsum_chunks"][1] / prof["maxsum_chunks"][0]
print(f"R={R} A={A} T={T} c={C}: kernel {ms:.3f} ms, wall {wall*1e3:.3f} ms, {ops / ms / 1e9:.2f} Tops/s f64 "
      f"({ops / ms / 1e9 / 39.3 * 100:.1f}% of VALU peak)")
# check vs numpy
want = np.maximum(L.T[:, None, :], L[ids[:, 0]].T[:, :, None] if C == 1 else L[ids].max(axis=1).T[:, :, None]).sum(axis=0) if R <= 20000 else None
if want is not None:
    print("exact vs numpy:", np.array_equal(out, want))

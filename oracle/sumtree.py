"""
Oracle: numpy's ``add.reduce`` summation tree for a contiguous float64 axis, in plain Python.

TEST INFRASTRUCTURE (see oracle/__init__.py).  numpy is the reference's arithmetic for
``log_probs[:, ids].sum(axis=0)`` and ``np.maximum(...).sum(axis=1)``
(``graphkir/typing_mulit_allele.py:514, 542, 571, 580``); the device kernels in
``csrc/gk_search.hip`` follow this tree, and tests/golden/t10_sums pins it against numpy 2.2.6:

* the axis is cut into buffers of 8192 elements, buffer sums are accumulated left to right;
* a buffer of n elements: n < 8 sequential from 0.0; n <= 128 eight strided accumulators combined
  as ((0+1)+(2+3))+((4+5)+(6+7)) followed by the n % 8 tail; otherwise split at
  n2 = n//2 - (n//2) % 8 and add the two halves.
"""
from __future__ import annotations

BUFFER = 8192
BLOCK = 128


def pairwise(a) -> float:
    n = len(a)
    if n < 8:
        r = 0.0
        for x in a:
            r += x
        return r
    if n <= BLOCK:
        acc = [a[j] for j in range(8)]
        i = 8
        while i < n - (n % 8):
            for j in range(8):
                acc[j] += a[i + j]
            i += 8
        r = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]))
        while i < n:
            r += a[i]
            i += 1
        return r
    n2 = n // 2
    n2 -= n2 % 8
    return pairwise(a[:n2]) + pairwise(a[n2:])


def numpySum(a) -> float:
    a = [float(x) for x in a]
    total = None
    for s in range(0, len(a), BUFFER):
        part = pairwise(a[s:s + BUFFER])
        total = part if total is None else total + part
    return 0.0 if total is None else total

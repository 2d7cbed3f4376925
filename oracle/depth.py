"""
Oracle: per-position read depth of the filtered, uniquely mapped pairs.

TEST INFRASTRUCTURE (see oracle/__init__.py).  The reference gets this table from
``samtools depth -aa {name}.no_multi.bam`` (``graphkir/samtools_utils.py:9-14``, ``main.py:152-158``)
where the BAM holds both mates of every filter-passing pair with ``NH == 1``
(``hisat2.saveReadsToBam`` 883-901).  samtools is not available in the build image, so the counting
rule is restated from the samtools-depth documentation: a read covers the reference positions of its
M / = / X runs; deletions (D), reference skips (N), insertions and soft clips do not count; mates
are counted independently; secondary / unmapped / QC-fail / duplicate reads are skipped.
No reference output pins this file ("parity unpinned" for the depth rule itself; the copy-number
stage downstream IS pinned by tests/golden/t8_cn).
"""
from __future__ import annotations

import re

import numpy as np

_CIGAR = re.compile(r"(\d+)([MIDNSHP=X])")


def depthFromPairs(pairs, gene_len: dict[str, int], multiple: bool = False) -> dict[str, np.ndarray]:
    """pairs: iterable of (l_sam, r_sam, NH) of the filter-passing pairs."""
    depth = {g: np.zeros(n, dtype=np.int64) for g, n in gene_len.items()}
    for l_sam, r_sam, nh in pairs:
        if not multiple and nh != 1:
            continue
        for line in (l_sam, r_sam):
            c = line.split("\t")
            if int(c[1]) & (4 | 256 | 512 | 1024):
                continue
            d = depth[c[2]]
            pos = int(c[3]) - 1
            for n, op in _CIGAR.findall(c[5]):
                n = int(n)
                if op in "M=X":
                    d[max(pos, 0):pos + n] += 1
                    pos += n
                elif op in "DN":
                    pos += n
    return depth

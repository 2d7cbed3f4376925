"""
Read depth per backbone position -- drop-in for ``graphkir/samtools_utils.py``.

``bam2Depth(file_bam, file_depth)`` keeps the reference's behaviour (``samtools depth -aa``) when
samtools is installed.  ``depthOfSample`` computes the same table on the GPU from the packed records
of a tabulated sample (filter-passing pairs with NH == 1, i.e. the content of the reference's
``.no_multi.bam``), which is what the pipeline uses: no BAM rewrite, no external process.
"""
from __future__ import annotations

import numpy as np
import pandas as pd

from ._lib import check, lib
from .external_tools import runTool


def bam2Depth(file_bam: str, file_depth: str, get_all: bool = True) -> None:
    args = ["samtools", "depth"] + (["-aa"] if get_all else []) + [file_bam, "-o", file_depth]
    runTool("samtools", args)


def readSamtoolsDepth(depth_filename: str) -> pd.DataFrame:
    return pd.read_csv(depth_filename, sep="\t", header=None, names=["gene", "pos", "depth"])


def readLocusLengths(index: str) -> dict[str, int]:
    """Backbone lengths from ``.locus`` (column 5, hisat2.py:146-155 reads the same file)."""
    out = {}
    with open(index + ".locus") as f:
        for line in f:
            cols = line.split("\t")
            out[cols[0]] = int(cols[4])
    return out


def depthOfSample(data, gene_len: dict[str, int], file_depth: str | None = None, multiple: bool = False,
                  want_frame: bool = True) -> pd.DataFrame | None:
    """``gene, pos (1-based), depth`` for every position of every backbone, like ``samtools depth -aa``.

    The file (when asked for) is written natively (``gk_depth_write_tsv``: the same text ``DataFrame.to_csv`` gives,
    at a fraction of the time); ``want_frame=False`` skips building the DataFrame (the pipeline only needs the file)."""
    import ctypes as C
    tab = data.tab
    genes = data.index.genes
    lens = np.array([gene_len[g] for g in genes], dtype=np.int64)
    off = np.zeros(len(genes) + 1, dtype=np.int64)
    np.cumsum(lens, out=off[1:])
    depth = np.empty(int(off[-1]), dtype=np.uint32)
    check(lib().gk_depth(tab.dev.ctx, tab.handle, tab.mates.ptr, int(multiple), off.ctypes.data, len(genes),
                         depth.ctypes.data))
    if file_depth:
        names = (C.c_char_p * len(genes))(*[g.encode() for g in genes])
        check(lib().gk_depth_write_tsv(file_depth.encode(), names, off.ctypes.data, len(genes), depth.ctypes.data))
    if not want_frame:
        return None
    return pd.DataFrame({
        "gene": np.repeat(np.array(genes, dtype=object), lens),
        "pos": np.concatenate([np.arange(1, n + 1) for n in lens]) if len(lens) else np.zeros(0, np.int64),
        "depth": depth.astype(np.int64),
    })

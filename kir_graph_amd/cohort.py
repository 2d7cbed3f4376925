"""
Cohort sharding over the GPUs of one node.

Samples are independent in the reference (sequential loops, main.py:139-167, 178-220), so a cohort
shards embarrassingly: one process per GPU, every rank tabulates and types its own samples, rank 0
merges the per-sample TSVs.  The only exchange is ``--cn-cohort``: the reference pools the raw gene
depths of all samples into one list before fitting ONE copy-number model (kir_cn.py:61, 167-186).
Here every rank contributes the depths of its samples through a single all-gather
(``comm.Comm.allgatherF64``: ``gk_allgather_f64`` = RCCL over xGMI when every rank has its own GPU, the
rendezvous directory otherwise and in the CPU tests); the payload is 8 x (genes + 1) bytes per sample, so
the collective is latency bound.
"""
from __future__ import annotations

import os

import numpy as np


def prefetched(items, prepare, depth: int = 1, workers: int = 1):
    """Yield ``prepare(item)`` for every item, in order, preparing up to ``depth`` items ahead on
    ``workers`` helper threads while the caller works on the current one.

    Samples of a cohort are independent, so the ingest + tabulation of the next sample (native code:
    the GIL is released, its kernels run on the device's own stream) overlaps the typing of the
    current one, which runs on the worker streams.  An exception in ``prepare`` is re-raised at the
    matching ``next()``."""
    from concurrent.futures import ThreadPoolExecutor
    items = iter(items)
    with ThreadPoolExecutor(max_workers=max(1, workers), thread_name_prefix="gk-prefetch") as pool:
        pending = []
        for item in items:
            pending.append(pool.submit(prepare, item))
            if len(pending) > depth:
                yield pending.pop(0).result()
        while pending:
            yield pending.pop(0).result()


def overlapped(items, work, lanes: int = 2):
    """Yield ``work(item, lane)`` for every item, in order, with up to ``lanes`` items in flight on
    helper threads (``lane`` = 0 .. lanes-1 tells ``work`` which set of device contexts is its own).

    Typing one sample ends with a tail -- the last, largest gene finishes alone -- and starts with
    a ramp; with two samples in flight the GPU and the host threads of one fill the other's gaps."""
    from concurrent.futures import ThreadPoolExecutor
    import queue
    if lanes <= 1:
        for item in items:
            yield work(item, 0)
        return
    free = queue.SimpleQueue()
    for lane in range(lanes):
        free.put(lane)

    def run(item):
        lane = free.get()
        try:
            return work(item, lane)
        finally:
            free.put(lane)

    with ThreadPoolExecutor(max_workers=lanes, thread_name_prefix="gk-sample") as pool:
        pending = []
        for item in items:
            pending.append(pool.submit(run, item))
            if len(pending) >= lanes:
                yield pending.pop(0).result()
        while pending:
            yield pending.pop(0).result()


def shardSamples(n_samples: int, world: int, weights=None) -> list[list[int]]:
    """Assignment of sample indices to ranks, deterministic and known to every rank.

    With ``weights`` (a cost per sample, e.g. the size of its read files): longest processing time first --
    samples in descending weight go to the least loaded rank (ties: lowest index / lowest rank), SURVEY.md
    section 8(e); a rank works through its samples in cohort order.  Without: round robin."""
    if weights is None:
        return [list(range(r, n_samples, world)) for r in range(world)]
    assert len(weights) == n_samples
    load = [0.0] * world
    shards: list[list[int]] = [[] for _ in range(world)]
    for i in sorted(range(n_samples), key=lambda i: (-float(weights[i]), i)):
        r = min(range(world), key=lambda r: (load[r], len(shards[r]), r))
        shards[r].append(i)
        load[r] += float(weights[i])
    return [sorted(s) for s in shards]


def sampleWeights(paths: list) -> list[float] | None:
    """Cost estimate per sample = bytes of its input files (every rank sees the same files); None when a
    file is missing (the caller then falls back to round robin)."""
    out = []
    for group in paths:
        total = 0
        for p in ([group] if isinstance(group, str) else group):
            if not p:
                continue
            try:
                total += os.path.getsize(p)
            except OSError:
                return None
        out.append(float(total))
    return out if any(out) else None


class Comm:
    """The cohort's samples over the ranks of a launch (``comm.Comm`` does the exchange)."""

    def __init__(self, n_samples: int, transport, weights=None):
        self.transport = transport
        self.rank, self.world = transport.rank, transport.world
        self.n_samples = n_samples
        self.shards = shardSamples(n_samples, self.world, weights)

    @property
    def mine(self) -> list[int]:
        return self.shards[self.rank]

    def allgatherDepths(self, local_depths: list[dict[str, float]]) -> list[float]:
        """Pool the gene depths of the whole cohort, in cohort sample order then gene order.

        ``local_depths[i]`` belongs to cohort sample ``self.mine[i]``.  One all-gather of a
        ``[max_local, 1 + genes]`` float64 block per rank (kir_cn.py:61, 167-177)."""
        assert len(local_depths) == len(self.mine)
        n_gene = [len(d) for d in local_depths]
        G = int(self.transport.maxF64(float(max(n_gene, default=0))))
        max_local = max(len(s) for s in self.shards)
        buf = np.full((max_local, 1 + G), np.nan, dtype=np.float64)
        buf[:, 0] = -1
        for i, (gi, d) in enumerate(zip(self.mine, local_depths)):
            if len(d) != G:
                raise ValueError("all samples of a cohort must report the same genes (samtools depth -aa)")
            buf[i, 0] = gi
            buf[i, 1:] = [d[g] for g in d]          # the sample's own (sorted-gene) order
        rows = self.transport.allgatherF64(buf.ravel()).reshape(-1, 1 + G)
        rows = rows[rows[:, 0] >= 0]
        rows = rows[np.argsort(rows[:, 0], kind="stable")]
        if len(rows) != self.n_samples:
            raise RuntimeError(f"cohort all-gather returned {len(rows)} of {self.n_samples} samples")
        return [float(v) for v in rows[:, 1:].reshape(-1)]

    def gatherInCohortOrder(self, mine: list) -> list:
        """Per-sample items of every rank (``mine[k]`` belongs to sample ``self.mine[k]``) back in cohort order."""
        everyone = self.transport.allgatherObject(list(mine))
        out = [None] * self.n_samples
        for r, idxs in enumerate(self.shards):
            for k, gi in enumerate(idxs):
                if k < len(everyone[r]):
                    out[gi] = everyone[r][k]
        return out

    def barrier(self) -> None:
        self.transport.barrier()

    def close(self) -> None:
        self.transport.close()


def initFromEnv(dev=None, backend: str | None = None):
    """The launch's ``comm.Comm`` (RANK / WORLD_SIZE / LOCAL_RANK from torchrun or ``bench.py --gpus``), None if single."""
    from . import comm
    return comm.initFromEnv(dev=dev, backend=backend)

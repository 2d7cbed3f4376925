"""
Cohort sharding over the GPUs of one node.

Samples are independent in the reference (sequential loops, main.py:139-167, 178-220), so a cohort
shards embarrassingly: one process per GPU, every rank tabulates and types its own samples, rank 0
merges the per-sample TSVs.  The only exchange is ``--cn-cohort``: the reference pools the raw gene
depths of all samples into one list before fitting ONE copy-number model (kir_cn.py:61, 167-186).
Here every rank contributes the depths of its samples through a single all-gather
(``torch.distributed``: backend ``nccl`` = RCCL over xGMI on the GPUs, ``gloo`` in CPU tests); the
payload is 8 x (genes + 1) bytes per sample, so the collective is latency bound.
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


def shardSamples(n_samples: int, world: int) -> list[list[int]]:
    """Round-robin assignment of sample indices to ranks (deterministic, known to every rank)."""
    return [list(range(r, n_samples, world)) for r in range(world)]


class Comm:
    """Thin wrapper over an initialised ``torch.distributed`` process group."""

    def __init__(self, n_samples: int, device: str | None = None):
        import torch.distributed as dist
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.dist = dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.n_samples = n_samples
        self.shards = shardSamples(n_samples, self.world)
        self.device = device or ("cuda" if dist.get_backend() == "nccl" else "cpu")

    @property
    def mine(self) -> list[int]:
        return self.shards[self.rank]

    def allgatherDepths(self, local_depths: list[dict[str, float]]) -> list[float]:
        """Pool the gene depths of the whole cohort, in cohort sample order then gene order.

        ``local_depths[i]`` belongs to cohort sample ``self.mine[i]``.  One all-gather of a
        ``[max_local, 1 + genes]`` float64 tensor per rank."""
        import torch
        assert len(local_depths) == len(self.mine)
        genes = sorted(set().union(*[d.keys() for d in local_depths])) if local_depths else []
        n_gene = torch.tensor([len(genes)], dtype=torch.int64, device=self.device)
        self.dist.all_reduce(n_gene, op=self.dist.ReduceOp.MAX)
        G = int(n_gene.item())
        max_local = max(len(s) for s in self.shards)
        buf = np.full((max_local, 1 + G), np.nan, dtype=np.float64)
        buf[:, 0] = -1
        for i, (gi, d) in enumerate(zip(self.mine, local_depths)):
            if len(d) != G:
                raise ValueError("all samples of a cohort must report the same genes (samtools depth -aa)")
            buf[i, 0] = gi
            buf[i, 1:] = [d[g] for g in d]          # the sample's own (sorted-gene) order
        mine = torch.from_numpy(buf).to(self.device)
        out = torch.empty((self.world * max_local, 1 + G), dtype=mine.dtype, device=self.device)
        self.dist.all_gather_into_tensor(out, mine)   # ranks concatenated along dim 0
        rows = out.reshape(-1, 1 + G).cpu().numpy()
        rows = rows[rows[:, 0] >= 0]
        rows = rows[np.argsort(rows[:, 0], kind="stable")]
        if len(rows) != self.n_samples:
            raise RuntimeError(f"cohort all-gather returned {len(rows)} of {self.n_samples} samples")
        return [float(v) for v in rows[:, 1:].reshape(-1)]

    def barrier(self) -> None:
        self.dist.barrier()


def initFromEnv(backend: str | None = None):
    """Initialise ``torch.distributed`` from RANK / WORLD_SIZE / MASTER_* (torchrun); None if single."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return None
    import torch
    import torch.distributed as dist
    if backend is None:
        # more ranks than GPUs on this node (several processes per GPU, each with its own interpreter lock):
        # RCCL wants one rank per device, and the only exchange is a few hundred bytes -- gloo carries it
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
        one_per_gpu = torch.cuda.is_available() and local_world <= torch.cuda.device_count()
        backend = "nccl" if one_per_gpu else "gloo"
    if backend == "nccl":
        local = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        dist.init_process_group(backend)
    return dist

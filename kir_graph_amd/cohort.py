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


def hostCoresPerRank() -> int:
    """Host cores this process may count on: the cores it is allowed on (affinity), capped by the container's CPU quota
    (cgroup v2 ``cpu.max``), shared between the ranks of this node (LOCAL_WORLD_SIZE / WORLD_SIZE of any launcher).
    An affinity smaller than the machine is NOT taken for a private one -- a cpuset-limited container or a Slurm cgroup
    gives every rank of the node the same smaller set; a launcher that pins each rank to cores of its own says so with
    GK_PRIVATE_CORES=1 (``bench.py --cores-per-gpu`` does)."""
    try:
        cores = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        cores = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            cores = min(cores, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    private = os.environ.get("GK_PRIVATE_CORES") == "1"
    ranks = 1 if private else max(1, int(os.environ.get("LOCAL_WORLD_SIZE") or os.environ.get("WORLD_SIZE") or 1))
    return max(1, cores // ranks)


def pipelineDefaults(procs: int = 1, cores: int | None = None) -> None:
    """Environment defaults of the sample pipeline, set before the first HIP call of the process (the wait policy is
    read when a context is made).  ``main.main`` and ``bench.py`` both call this: ONE process per GPU types several
    samples at a time (``GK_SAMPLE_LANES``), some of them inside their search (``GK_SEARCH_SLOTS``), the sample preamble
    on a high-priority stream, waits that put the thread to sleep.  How many depends on the host cores the rank has
    (``cores``, default ``hostCoresPerRank()``): five lanes and three searches where it has six or more (7.2 - 7.7 ms per
    configs[1] sample on 3 - 4 busy cores), four and two from three cores (8.0 ms on 2.8), three and two below that
    (8.3 ms on 2.6; a rank of an 8-GPU node on a 16-core quota has two) -- profiles/r04_sample_lanes.txt.  Anything the
    user set stays."""
    cores = hostCoresPerRank() if cores is None else cores
    lanes, slots = (5, 3) if cores >= 6 else (4, 2) if cores >= 3 else (3, 2)
    os.environ.setdefault("GK_WAIT_POLICY", "block")
    os.environ.setdefault("GK_SAMPLE_LANES", str(lanes))
    os.environ.setdefault("GK_SEARCH_SLOTS", str(slots))


def sampleLanes() -> int:
    """Samples typed at a time by this process, each on a host thread and a block of device contexts of its own."""
    return max(1, int(os.environ.get("GK_SAMPLE_LANES", "3")))


def stagingContexts(dev, lanes: int | None = None):
    """(copier, ingest): the two device contexts of the staging stages, beyond the contexts of the typing lanes
    (workers 0 .. lanes - 1 belong to those): one stream for the copy of a sample's records into
    HBM, one -- of the device's highest priority -- for its tabulation, whose short kernels would otherwise sit behind
    the long kernels of the samples being typed."""
    lanes = sampleLanes() if lanes is None else lanes
    base = lanes
    ingest = dev.worker(base, urgent=True)       # made first: a context's priority is fixed when it is made
    return dev.worker(base + 1), ingest


def stagedSamples(items, copy_in, tabulate, depth: int | None = None):
    """Yield ``tabulate(copy_in(item))`` for every item, in order, staged ahead of the consumer as a two-stage
    pipeline: while sample k is typed, sample k + 1 is tabulated and the records of k + 2 are on their way to HBM,
    each stage on a thread (and a device context, ``stagingContexts``) of its own (both stages in one took 6 - 7 ms of
    wall time per configs[1] sample next to the typing kernels, 80 % of a process's budget).  ``copy_in=None``: the records
    are in HBM already, one stage.  ``depth`` <= 0 (GK_PREFETCH=0): nothing ahead, everything on the calling thread."""
    depth = int(os.environ.get("GK_PREFETCH", "1")) if depth is None else depth
    if depth <= 0:
        for item in items:
            yield tabulate(item if copy_in is None else copy_in(item))
        return
    if copy_in is None:
        yield from prefetched(items, tabulate, depth=depth)
    else:
        yield from prefetched(prefetched(items, copy_in, depth=depth), tabulate, depth=depth)


def sampleFootprint(data, method: str) -> int:
    """Bytes of HBM the typing of a tabulated sample holds at its peak, estimated before it starts: the compatibility
    tables of its genes (a float64 and a mismatch byte per read and allele; twice for exon-first, whose exon model's tables
    stand beside the full ones; candidate bit sets only for the EM), its lists and its records.  0 for a sample that is
    still a file."""
    tab = getattr(data, "tab", None)
    if tab is None:
        return 0
    tables = getattr(data.index, "tables", None) or []
    mean_alleles = sum(t.n_allele for t in tables) / max(len(tables), 1)
    n = int(tab.n_valid)
    if method in ("em", "report"):
        per_read = 4.0 * (mean_alleles / 32 + 1) + 16
    else:
        per_read = 9.2 * mean_alleles * (2 if method.startswith(("exonfirst", "pv_exonfirst")) else 1)
    lists = 4.0 * int(tab.n_ids) + 48.0 * n
    records = 256.0 * int(tab.n_pairs) if getattr(tab, "mates", None) is not None else 0.0
    return int(1.15 * per_read * n + lists + records)


def hbmBudget(dev=None) -> int:
    """Bytes of HBM the samples in flight of this process may hold together (``sampleFootprint``): GK_HBM_BUDGET_GB, else
    55 % of the device -- the rest is for the samples staged ahead, the blocks the pools keep idle and the ranks that may
    share the card."""
    env = os.environ.get("GK_HBM_BUDGET_GB")
    if env:
        return int(float(env) * (1 << 30))
    try:
        if dev is None:
            from .kir_typing import defaultDevice
            dev = defaultDevice()
        return int(0.55 * dev.memory()[1])
    except Exception:
        return 0                  # unknown: no admission control


class SampleTyper:
    """The typing stage of one process: ``submit`` hands a tabulated sample to one of ``lanes`` host threads (each
    with its own block of device contexts: ``typer.slot_base``), results come back in submission order.

    This is the one code path behind ``main.alleleTyping`` / ``main._runCohort`` (the command line) and ``bench.py``
    (the measurement): a sample is typed by ONE host thread on one stream (``gk_sample_search``), up to ``lanes``
    samples at a time, at most GK_SEARCH_SLOTS of them inside their search (``kir_typing._searchSlot``), the
    preamble of a sample on the lane's high-priority stream, waits that block.

    ``finish(typer, calls, warnings, item)`` runs on the lane's thread once the sample is typed (write its files,
    release its HBM); its return value is the sample's result."""

    def __init__(self, method: str, lanes: int | None = None, top_n: int = 600, variant_correction: bool = True,
                 finish=None):
        from concurrent.futures import ThreadPoolExecutor
        import queue
        self.method = "exonfirst_1" if method == "exonfirst" else method     # main.py:186-187
        self.lanes = sampleLanes() if lanes is None else max(1, lanes)
        self.top_n, self.variant_correction = top_n, variant_correction
        self.finish = finish
        self._pool = ThreadPoolExecutor(max_workers=self.lanes, thread_name_prefix="gk-sample") if self.lanes > 1 else None
        self._free = queue.SimpleQueue()
        for lane in range(self.lanes):
            self._free.put(lane)
        self._pending: list = []
        # admission by memory: a sample starts when the samples in flight leave room for it (one sample always may) --
        # the lanes follow the host cores, but five 20 M-read samples typed exon-first do not fit one card
        import threading
        self._room = threading.Condition()
        self._inflight_bytes, self._budget = 0, None

    def _admit(self, data) -> int:
        need = sampleFootprint(data, self.method)
        if need <= 0 or self._pool is None:
            return 0
        if self._budget is None:
            self._budget = hbmBudget(getattr(getattr(data, "tab", None), "dev", None))
        if self._budget <= 0:
            return 0
        with self._room:
            while self._inflight_bytes > 0 and self._inflight_bytes + need > self._budget:
                self._room.wait(timeout=1.0)
            self._inflight_bytes += need
        return need

    def _release(self, need: int) -> None:
        if need:
            with self._room:
                self._inflight_bytes -= need
                self._room.notify_all()

    def typeOne(self, data, gene_cn, item=None, lane: int = 0):
        """One sample on the calling thread, on lane ``lane``'s contexts."""
        from .kir_typing import selectKirTypingModel
        typer = selectKirTypingModel(self.method, data, top_n=self.top_n, variant_correction=self.variant_correction)
        typer.slot_base = lane
        calls, warnings = typer.typing(gene_cn() if callable(gene_cn) else gene_cn)
        if self.finish is not None:
            return self.finish(typer, calls, warnings, item)
        return calls, warnings, typer

    def _run(self, data, gene_cn, item, need: int = 0):
        lane = self._free.get()
        try:
            return self.typeOne(data, gene_cn, item, lane)
        finally:
            self._free.put(lane)
            self._release(need)

    def submit(self, data, gene_cn, item=None) -> None:
        """Queue a sample (``gene_cn``: the copy numbers, or a callable that returns them on the lane's thread)."""
        if self._pool is None:
            self._pending.append(_Done(self.typeOne(data, gene_cn, item, 0)))
        else:
            self._pending.append(self._pool.submit(self._run, data, gene_cn, item, self._admit(data)))

    def inFlight(self) -> int:
        return len(self._pending)

    def next(self):
        """The oldest sample's result (waits for it); an exception of its typing is raised here."""
        return self._pending.pop(0).result()

    def drain(self):
        while self._pending:
            yield self.next()

    def close(self) -> None:
        if self._pool is not None:
            self._pool.shutdown(wait=True)
            self._pool = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


class _Done:
    def __init__(self, value):
        self._value = value

    def result(self):
        return self._value


def typeSamples(samples, method: str, lanes: int | None = None, finish=None, top_n: int = 600,
                variant_correction: bool = True):
    """Type every ``(SampleData, gene_cn, item)`` of ``samples`` -- an iterable that may stage its samples ahead
    (``stagedSamples``) -- and yield the results in order, with up to ``lanes`` samples in flight (``SampleTyper``)."""
    with SampleTyper(method, lanes=lanes, top_n=top_n, variant_correction=variant_correction, finish=finish) as typer:
        for data, gene_cn, item in samples:
            typer.submit(data, gene_cn, item)
            if typer.inFlight() >= typer.lanes:
                yield typer.next()
        yield from typer.drain()


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

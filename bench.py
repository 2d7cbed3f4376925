#!/usr/bin/env python3
"""
bench.py -- typed 150 bp PE reads/s of the Graph-KIR hot path on MI355X.

One "step" = one synthetic sample taken from packed alignment records to per-gene allele calls on the host: the tabulation
(gk_tabulate), the sample preamble (error correction, empty reads, zygosity tallies), per gene the compatibility table
through the log10 value table and the allele selection of the strategy -- the greedy multi-allele likelihood search
(gk_sample_search: pv; exon-first: the exon models, then the candidate searches on the full tables) or the EM
(gk_sample_em).  The steps run through the package's own sample pipeline (kir_graph_amd.cohort: stagedSamples ->
typeSamples), the one `python -m kir_graph_amd.main` types a cohort with.

Workloads.  Without --pairs / --method the line's headline (`value`) is BASELINE.json configs[2] -- 20 M reads = 10 M
pairs per sample, --allele-strategy exonfirst (what the reference's own pipeline runs, kir/graphkir.py:76-87) -- and, on one
GPU, two more workloads are measured the same way and reported as objects of the same line: `em` (configs[2],
--allele-strategy em: EM to convergence) and `configs1_pv` (configs[1]: 2 M reads, --allele-strategy pv -- the headline of
rounds 1 - 4).  Each carries its own legs, `kernels_serial`, `roofline` and `cpu_baseline`.  With --pairs and / or --method:
that one workload.  All use the synthetic example_index-shaped index (~2.4 k alleles in 15 genes), top_n 600, variant
correction on.

Two kinds of timed leg over the same K steps, each bracketed by a barrier + device synchronise on both sides, each kind
timed `--legs` times (default 3) with the MEDIAN leg reported:
  host  (`value`)         a sample's records start in PINNED HOST MEMORY, in the compact form they cross PCIe in (~30 bytes
                          per mate: packed.CompactMates): the host-to-device copy and the expansion into 128-byte
                          records are inside the region, staged two samples ahead of the typing;
  hbm   (`hbm_resident`)  the records of the distinct samples are resident in HBM before the clock starts.
`--inputs hbm` swaps the two (`value` from resident records, `pcie_inclusive` beside it).
Consecutive steps take DIFFERENT samples: `--distinct N` distinct ones per rank in rotation (2 at configs[2] size -- a
sample takes ~30 s to make -- 8 below), seeds 1031 + 7 rank + i.  With N >= steps + warmup the first leg types only samples
nobody has typed before -- `legs[0]` then says what a NEW sample costs (value_table_new_per_sample, samples_repeated_pass).

``--gpus N``: N ranks, one per GPU, every rank types its own samples (cohort sharding: weak scaling, no
data-path collective); value = reads of all ranks / max-over-ranks time.  Started without a launcher
(`python bench.py --gpus N`) this process starts the N ranks itself -- as fresh processes, before
anything here touches the GPU -- and relays rank 0's line; under `torchrun` (RANK / WORLD_SIZE set) it
is one of the ranks.  Ranks meet through kir_graph_amd/comm.py (RCCL: barrier + max of the times).

Prints ONE JSON line on rank 0 (contract in the task statement) with extra objects:
  roofline        the dominant kernel of the step, from a SERIAL pass (one sample at a time on one stream, no prefetch)
                  run right after the timed legs: HIP-event time per launch, algorithmic bytes / operations per launch
                  (kir_graph_amd/roofmodel.py); `roofline.step` = the algorithmic bytes of ALL
                  launches of a step over the reported ms_per_step against 8 TB/s; `traffic` = HBM bytes per launch from the
                  committed PMC passes of the same workload on the same device sources (profiles/)
  kernels_serial  per-kernel launches and time per step of that pass, under the kernels' own names (the basis rocprofv3
                  reproduces, profiles/)
  legs            every timed leg of the headline kind in the order they ran
  cpu_baseline    the oracle (CPU restatement of the reference) on a bounded sample of the same workload and strategy,
                  one core and (headline) N-way over the host's cores
  host            what the step costs on the host: core-seconds per step (user + system time of rank 0's process over the
                  reported leg, getrusage), cores busy on average, the cores the rank was allowed (``--cores-per-gpu K``
                  pins every rank to K cores of its own before anything touches HIP: the budget an 8-GPU node leaves a rank)
  em, configs1_pv the other two workloads (see above)
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import tempfile
import threading
import time


ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
N_DISTINCT = 8          # distinct samples a rank rotates through (--distinct)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def allowed_cores() -> list[int]:
    try:
        return sorted(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        return list(range(os.cpu_count() or 1))


def cgroup_cores() -> int | None:
    """The container's CPU quota in cores (cgroup v2 cpu.max), None when unlimited / unknown."""
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            return max(1, int(quota) // int(period))
    except (OSError, ValueError):
        pass
    return None


def pin_rank(local_rank: int, k: int) -> list[int]:
    """``--cores-per-gpu K``: this rank (and every worker process / thread it starts later) runs on K cores of its
    own -- cores [local_rank * K, (local_rank + 1) * K) of the allowed set, wrapping around when the set is smaller.
    Must run before the first HIP call (the runtime's helper threads inherit the mask)."""
    cores = allowed_cores()
    mine = [cores[(local_rank * k + i) % len(cores)] for i in range(min(k, len(cores)))]
    os.sched_setaffinity(0, set(mine))
    os.environ["GK_PRIVATE_CORES"] = "1"      # cohort.hostCoresPerRank: this affinity is the rank's own, not the node's
    return sorted(set(mine))


def pin_all_threads(cores) -> int:
    """Apply the rank's core set to EVERY thread this process has by now: the HIP / ROCr runtime starts helper threads
    (signal and event handling) that do not keep the mask they inherit -- without this a rank pinned to 2 cores showed
    2.8 cores busy.  Called after the warm-up, when those threads exist.  Returns the number of threads bound."""
    n = 0
    for tid in os.listdir("/proc/self/task"):
        try:
            os.sched_setaffinity(int(tid), set(cores))
            n += 1
        except (OSError, ValueError):
            pass
    return n


def thread_cpu_table() -> list[tuple[str, int, float]]:
    """(name, thread id, user + system seconds so far) of every thread of this process, from /proc/self/task."""
    out = []
    tick = os.sysconf("SC_CLK_TCK")
    for tid in os.listdir("/proc/self/task"):
        try:
            with open(f"/proc/self/task/{tid}/stat") as f:
                text = f.read()
        except OSError:
            continue
        name = text[text.index("(") + 1:text.rindex(")")]
        rest = text[text.rindex(")") + 2:].split()
        out.append((name, int(tid), (int(rest[11]) + int(rest[12])) / tick))
    return out


def cpu_seconds() -> float:
    """User + system time of this process (all its threads) so far."""
    import resource
    ru = resource.getrusage(resource.RUSAGE_SELF)
    return ru.ru_utime + ru.ru_stime


# ------------------------------------------------------------------------------------------ inputs
def build_index(index_seed: int = 2022):
    from kir_graph_amd import synth
    from kir_graph_amd.index import GkIndex
    sidx = synth.makeIndex(seed=index_seed)
    gidx = GkIndex.fromVariants(sidx.variants, genes=sidx.genes, exons=sidx.exons)
    by_gene = {}
    for v in sidx.variants:
        by_gene.setdefault(v.ref, []).append(v)
    return sidx, gidx, by_gene


def build_sample(sidx, gidx, by_gene, seed: int, n_pairs: int):
    from kir_graph_amd import synth, packed
    sample = synth.makeSample(sidx, seed=seed, n_pairs=n_pairs, variants_by_gene=by_gene)
    rec, table = packed.packSample(sample, gidx)
    return sample, rec, table


def build_inputs(seed: int, n_pairs: int, index_seed: int = 2022):
    """(index, packed index, sample, records, string table) of one synthetic sample (tests use this)."""
    t = time.time()
    sidx, gidx, by_gene = build_index(index_seed)
    sample, rec, table = build_sample(sidx, gidx, by_gene, seed, n_pairs)
    log(f"[bench] inputs: {len(gidx.variants)} variants, {sum(len(t.alleles) for t in gidx.tables)} alleles, "
        f"{n_pairs} pairs in {time.time() - t:.1f}s")
    return sidx, gidx, sample, rec, table


class PinnedRecords:
    """The packed records of a sample in pinned host memory, in the compact form they cross PCIe in
    (``packed.CompactMates``: ~30 bytes per mate instead of 128): the start of a step."""

    def __init__(self, rec):
        from kir_graph_amd.packed import CompactMates
        self.compact = CompactMates(rec, threads=4)
        self.nbytes, self.count, self.dtype = self.compact.nbytes, len(rec), rec.dtype
        self.record_bytes = rec.nbytes

    def toDevice(self, dev):
        """Queue the copy and the expansion into 128-byte records on ``dev``'s stream; kernels launched on that stream
        afterwards see the records."""
        return self.compact.toDevice(dev)

    def free(self):
        self.compact = None


# ------------------------------------------------------------------------------------------ steps
def run_steps(items, dev, dindex, gidx, inputs, method, depth=None, threads=None, resident=None, stats=None):
    """Types the samples ``inputs[k % len(inputs)]`` for k in ``items``: pinned records -> HBM -> tabulation ->
    typing -> calls, through the package's sample pipeline -- ``cohort.stagedSamples`` (copy and tabulation of the next
    samples on their own contexts while the current ones are typed) feeding ``cohort.typeSamples`` (the typing lanes),
    the same two calls ``kir_graph_amd.main`` makes for the samples of a cohort.  Every copy, tabulation and typing of
    the listed samples starts and ends inside this call.  ``resident``: the records of the distinct samples already in
    HBM (one device buffer per entry of ``inputs``) -- a step then starts at the tabulation.  ``stats`` (a dict):
    receives what the value table and the searches did over these samples."""
    from kir_graph_amd import cohort
    from kir_graph_amd.engine import Tabulation
    from kir_graph_amd.hisat2 import SampleData
    lanes = cohort.sampleLanes() if depth is None or depth > 0 else 1
    copier, ingest = cohort.stagingContexts(dev, cohort.sampleLanes())
    from kir_graph_amd.utils import traceOn
    trace = traceOn("bench")      # GK_TRACE=bench: a timeline of the host threads on stderr (tools/host_timeline.py)

    def note(what, k, t0):
        if trace:
            log(f"[trace] {what} {k} {threading.get_native_id()} {t0:.6f} {time.perf_counter():.6f}")

    def copy_in(k):
        t0 = time.perf_counter()
        mates = inputs[k % len(inputs)][0].toDevice(copier)
        copier.sync()                       # the records are in HBM when the next stage takes them
        note("copy", k, t0)
        return k, mates

    def tabulate(item):
        t0 = time.perf_counter()
        k, mates = item if isinstance(item, tuple) else (item, resident[item % len(inputs)])
        _, table, gene_cn = inputs[k % len(inputs)]
        tab = Tabulation(dindex, mates, dev=ingest)
        note("stage", k, t0)
        return SampleData(tab, gidx, None, ins_strings=table.strings), gene_cn, (k, time.perf_counter())

    def finish(typer, calls, warn, item):
        k, t0 = item
        tab = typer._data.tab
        n_valid = tab.n_valid
        tab.close()
        if resident is None:
            tab.mates.free()
        note("type", k, t0)
        if stats is not None:
            stats["samples"] = stats.get("samples", 0) + 1
            again = getattr(typer, "tables_rewritten", 0) + getattr(typer, "tables_patched", 0)
            stats["samples_repeated_pass"] = stats.get("samples_repeated_pass", 0) + (1 if again else 0)
            stats["tables_rewritten"] = stats.get("tables_rewritten", 0) + getattr(typer, "tables_rewritten", 0)
            stats["tables_patched"] = stats.get("tables_patched", 0) + getattr(typer, "tables_patched", 0)
        return calls, warn, n_valid, None        # the typer (and the tables of its genes) ends with its lane's turn

    items = range(items) if isinstance(items, int) else items
    staged = cohort.stagedSamples(items, None if resident is not None else copy_in, tabulate, depth=depth)
    out = None
    for out in cohort.typeSamples(staged, method, lanes=lanes, finish=finish):
        pass
    return out


def cli_typing_stage(n, dev, dindex, gidx, inputs, resident, method):
    """The typing stage of the COMMAND LINE (`kir_graph_amd.main.alleleTyping`, main.py:171-220 of the reference) on `n`
    tabulated samples, timed: copy-number files read, every sample typed through the process's typing lanes, its
    `.tsv` / `.possible.tsv` written, its tabulation released.  The samples are tabulated (and their copy-number files
    written) before the clock starts -- `main` does that in its mapping stage -- so the figure compares with a bench step
    minus its tabulation.  Same lanes, slots, streams and waits as the timed legs: one code path (cohort.SampleTyper)."""
    import shutil
    from kir_graph_amd import cohort, main as gk_main
    from kir_graph_amd.engine import Tabulation
    from kir_graph_amd.hisat2 import SampleData
    _, ingest = cohort.stagingContexts(dev, cohort.sampleLanes())
    tmp = tempfile.mkdtemp(prefix="gk_bench_cli_")
    try:
        processed, cn_files = [], []
        for k in range(n):
            _, table, gene_cn = inputs[k % len(inputs)]
            tab = Tabulation(dindex, resident[k % len(inputs)], dev=ingest)
            tab.mates = None        # the records are the bench's resident inputs: not this tabulation's to release
            name = os.path.join(tmp, f"s{k:03d}.variant")
            cn_file = name + ".no_multi.depth.p75.LCND.tsv"
            with open(cn_file, "w") as f:
                f.write("gene\tcn\tdepth\n" + "".join(f"{g}\t{c}\t{30.0 * c}\n" for g, c in gene_cn.items()))
            processed.append((name, SampleData(tab, gidx, None, ins_strings=table.strings)))
            cn_files.append(cn_file)
        ingest.sync()
        cpu0, t0 = cpu_seconds(), time.perf_counter()
        files = gk_main.alleleTyping(processed, cn_files, method="full" if method in ("pv", "full") else method, release=True)
        for d in list(type(dev).instances):
            d.sync()
        elapsed, cpu = time.perf_counter() - t0, cpu_seconds() - cpu0
        assert len(files) == n and all(os.path.getsize(f) > 0 for f in files)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return {"samples": n, "ms_per_sample": 1e3 * elapsed / n, "host_core_s_per_sample": cpu / n,
            "what": "kir_graph_amd.main.alleleTyping on tabulated samples (copy-number files read, typing lanes, .tsv + "
                    ".possible.tsv written, tabulations released): the command line's typing stage, timed in this process"}


def _oracle_leg(job, barrier=None, out=None):
    """One CPU-baseline job (runs in a fresh process for the N-way leg): tabulate + type with the oracle.  With a
    ``barrier`` the inputs are made first, then every process waits for the others: the clock of the N-way leg holds the
    oracle's work only."""
    seed, n_pairs, method = job
    sys.path.insert(0, ROOT)
    from kir_graph_amd import synth
    from oracle import tabulate as ot, typing as oty
    sidx, gidx, by_gene = build_index()
    sample = synth.makeSample(sidx, seed=seed, n_pairs=n_pairs, variants_by_gene=by_gene)
    lines = synth.toSamLines(sample, with_zs=False)
    if barrier is not None:
        barrier.wait(timeout=600)
    t0 = time.time()
    data = ot.tabulateLines(lines, gidx.variants)
    t1 = time.time()
    if method in ("em", "report"):
        from oracle import em as oem
        typer = oem.ReportTyper(data)
    else:     # the command line types `exonfirst` as exonfirst_1 (main.py:186-187)
        typer = oty.makeTyper({"pv": "full", "exonfirst": "exonfirst_1"}.get(method, method), data, top_n=600, variant_correction=True)
    typer.typing(sample.gene_cn)
    t2 = time.time()
    if out is not None:
        out.put((t0, t2))
    return t1 - t0, t2 - t1


def cpu_baseline(method, n_pairs, n_way=True):
    """Oracle on a bounded sample of the same workload: one core, then (``n_way``) N-way -- one sample per core, the way
    the reference's own speed test runs it: research/test_speed.graphkir.par.sh, `parallel -j 14 --thread 1`."""
    import multiprocessing as mp
    t_tab, t_typ = _oracle_leg((99, n_pairs, method))
    single = 2 * n_pairs / (t_tab + t_typ)
    cores = len(allowed_cores())
    quota = cgroup_cores()      # a container's CPU quota: the cores that can really run at once
    if quota is not None:
        cores = max(1, min(cores, quota))
    n_way = max(1, min(cores, 32)) if n_way else 1
    out = {"value": single, "unit": "reads/s", "cores": 1, "kind": "port",
           "sample": f"{n_pairs} pairs of the same synthetic workload, --allele-strategy {method} (R_g <= 8 k per gene: the "
                     f"reference's own real-depth regime); oracle: tabulate {t_tab:.1f}s + typing {t_typ:.1f}s on one core",
           "host_cores": cores}
    if n_way > 1:
        ctx = mp.get_context("spawn")
        barrier, results = ctx.Barrier(n_way), ctx.Queue()
        procs = [ctx.Process(target=_oracle_leg, args=((100 + i, n_pairs, method), barrier, results)) for i in range(n_way)]
        for p in procs:
            p.start()
        spans = [results.get(timeout=900) for _ in procs]
        for p in procs:
            p.join(timeout=60)
        wall = max(t1 for _, t1 in spans) - min(t0 for t0, _ in spans)
        out["n_way"] = {"value": 2 * n_pairs * n_way / wall, "unit": "reads/s", "cores": n_way,
                        "sample": f"{n_way} samples of {n_pairs} pairs, one process per core, wall {wall:.1f}s from the "
                                  "first process's start to the last one's end (inputs made before a common barrier)"}
    return out


# ------------------------------------------------------------------------------------------ workloads
WORKLOADS = {
    # name in the JSON line -> (configs[] entry of BASELINE.json, pairs per sample, --allele-strategy, distinct samples)
    "configs2_exonfirst": ("configs[2]", 10_000_000, "exonfirst", 2),
    "em": ("configs[2]", 10_000_000, "em", 2),
    "configs1_pv": ("configs[1]", 1_000_000, "pv", 8),
}


def configName(pairs: int) -> str:
    return "configs[1]" if pairs == 1_000_000 else "configs[2]" if pairs == 10_000_000 else "custom"


def make_inputs(sidx, gidx, by_gene, rank: int, pairs: int, distinct: int, threads: int = 1):
    """``distinct`` synthetic samples of ``pairs`` pairs as (pinned compact records, string table, copy numbers), seeds
    1031 + 7 rank + i.  ``threads`` > 1: that many samples are made at a time (numpy releases the interpreter lock in
    its large operations; a 10 M-pair sample takes ~30 s and ~12 GB of host memory while it is made)."""
    from concurrent.futures import ThreadPoolExecutor

    def one(i):
        sample, rec, table = build_sample(sidx, gidx, by_gene, 1031 + 7 * rank + i, pairs)
        pinned = PinnedRecords(rec)
        return pinned, table, sample.gene_cn

    t0 = time.time()
    if threads > 1 and distinct > 1:
        with ThreadPoolExecutor(max_workers=min(threads, distinct)) as pool:
            inputs = list(pool.map(one, range(distinct)))
    else:
        inputs = [one(i) for i in range(distinct)]
    log(f"[bench] rank {rank}: {len(inputs)} samples of {pairs} pairs in pinned memory, {inputs[0][0].nbytes / 1e6:.0f} MB each "
        f"in compact form ({inputs[0][0].record_bytes / 1e6:.0f} MB as 128-byte records; {time.time() - t0:.1f}s)")
    return inputs


class RankContext:
    """What a rank keeps across its workloads: the device, the index on it, the communicator."""

    def __init__(self, args, rank: int, local_rank: int):
        from kir_graph_amd import _lib, comm as gk_comm
        from kir_graph_amd.engine import DeviceIndex
        self.args, self.rank = args, rank
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        n_dev = _lib.deviceCount()
        if n_dev == 0:
            raise RuntimeError("bench.py: no HIP device visible (the typing path has no CPU fallback)")
        backend = {"nccl": "rccl", "gloo": "file"}.get(os.environ.get("GK_COMM_BACKEND", "rccl"),
                                                       os.environ.get("GK_COMM_BACKEND", "rccl"))
        if self.world > 1 and backend == "rccl" and self.world > n_dev:
            raise RuntimeError(f"bench.py: {self.world} ranks but {n_dev} GPU(s): one rank per GPU "
                               "(GK_COMM_BACKEND=file rehearses the multi-rank path on fewer GPUs)")
        self.dev = _lib.Device(local_rank % n_dev)
        self.sidx, self.gidx, self.by_gene = build_index()
        self.dindex = DeviceIndex(self.dev, self.gidx)
        self.dev.sync()
        self.comm, self.comm_note = None, None
        if self.world > 1:
            # The data path has no collective: the communicator only carries the barriers around the timed legs and the
            # max over ranks.  When the RCCL communicator cannot be made on some rank (RCCL with more than one rank has
            # never run on the builder's one-GPU boxes) all ranks agree to carry those few messages through the
            # rendezvous directory instead -- and the line SAYS so (config.rank_barrier = "file" + rank_barrier_note)
            # rather than the run producing no number at all.  A set-up that never returns counts as failed after 90 s.
            os.environ.setdefault("GK_RCCL_INIT_TIMEOUT", "90")
            try:
                self.comm = gk_comm.initFromEnv(dev=self.dev, backend=backend, fallback=True)
            except gk_comm.CommError as e:
                log(f"[bench] rank {rank}: {e}")
                os._exit(4)
            if self.comm.world != args.gpus:
                raise RuntimeError(f"bench.py: --gpus {args.gpus} but {self.comm.world} ranks joined")
            if self.comm.backend != backend:
                self.comm_note = (f"the {backend} communicator could not be made on every rank: barriers and the max over "
                                  f"ranks went through the rendezvous directory ({self.comm.backend} backend); the data path "
                                  "has no collective either way")

    def devices(self):
        from kir_graph_amd import _lib
        return list(_lib.Device.instances)

    def profiled(self, on):
        for d in self.devices():
            d.profEnable(on)
            if on:
                d.profCollect()
                d.call_log = []

    def collect(self):
        prof, call_log = {}, []
        for d in self.devices():
            for k, (n, ms) in d.profCollect().items():
                n0, ms0 = prof.get(k, (0, 0.0))
                prof[k] = (n0 + n, ms0 + ms)
            call_log += d.call_log or []
        return prof, call_log


def measure(rc: RankContext, inputs, pairs: int, method: str, opts) -> dict:
    """One workload on this rank: warm-up, the timed legs (two kinds, ``opts.legs`` each), the command line's typing stage
    and the one-process serial pass the roofline is taken from.  Returns what the JSON object of the workload is made of."""
    from kir_graph_amd.typing_mulit_allele import sharedLogTable, SEARCH_STATS
    from kir_graph_amd.utils import traceOn
    args, dev, dindex, gidx, comm = rc.args, rc.dev, rc.dindex, rc.gidx, rc.comm
    resident = [pinned.toDevice(dev) for pinned, _, _ in inputs]
    dev.sync()
    n_valid = 0
    if opts.warmup:
        n_valid = run_steps(opts.warmup, dev, dindex, gidx, inputs, method, resident=resident)[2]
        run_steps(min(opts.warmup, 4), dev, dindex, gidx, inputs, method)      # the copy path's contexts and pools
    if getattr(args, "pinned_to", None):      # --cores-per-gpu: the runtime's own threads too (they exist by now)
        pin_all_threads(args.pinned_to)
    if getattr(args, "profile_host", False):
        import cProfile
        import pstats
        n_prof = 1     # with GK_SAMPLE_LANES=1 GK_PREFETCH=0 everything is on this thread
        pr = cProfile.Profile()
        pr.enable()
        run_steps(n_prof, dev, dindex, gidx, inputs, method, resident=resident)
        pr.disable()
        log(f"[bench] host profile of {n_prof} step(s)")
        pstats.Stats(pr, stream=sys.stderr).sort_stats("cumulative").print_stats(45)
        pstats.Stats(pr, stream=sys.stderr).sort_stats("tottime").print_stats(45)
    # per-kernel events inside the timed region cost ~3 ms per step (a profiling signal per dispatch): only on request;
    # the roofline comes from the serial pass after the region
    in_region = bool(getattr(args, "verbose", False))

    def timed_leg(leg_resident):
        """EXACTLY opts.steps steps between a barrier + device synchronise on both sides; (seconds, host CPU seconds of
        this rank, the last step's result, what the value table and the searches did)."""
        for d in rc.devices():
            d.sync()
        if comm is not None:
            comm.barrier()          # RCCL all-reduce + stream synchronise: every rank is ready
        stats = {"value_table_at_start": sharedLogTable(dev).known()}
        t0 = time.perf_counter()
        cpu0 = cpu_seconds()
        by_thread0 = {tid: c for _, tid, c in thread_cpu_table()} if traceOn("bench") else None
        last = run_steps(opts.steps, dev, dindex, gidx, inputs, method, resident=leg_resident, stats=stats)
        for d in rc.devices():
            d.sync()
        cpu = cpu_seconds() - cpu0          # this rank's host time for the steps (waits that spin included)
        if by_thread0 is not None:          # GK_TRACE=bench: which threads the host time of the leg went to
            rows = sorted(((c - by_thread0.get(tid, 0.0), name, tid) for name, tid, c in thread_cpu_table()), reverse=True)
            log("[trace] host CPU of the leg by thread (ms per step): " +
                ", ".join(f"{name}/{tid} {1e3 * c / max(opts.steps, 1):.2f}" for c, name, tid in rows if c > 0))
        if comm is not None:
            comm.barrier()
        elapsed = time.perf_counter() - t0
        stats["value_table_new"] = sharedLogTable(dev).known() - stats.pop("value_table_at_start")
        return elapsed, cpu, last, stats

    kinds = [k for k in ("host", "hbm") if k == opts.inputs or opts.both_legs]
    kinds.sort(key=lambda k: k != opts.inputs)          # the headline's kind first: its first leg meets the new samples
    legs = {k: [] for k in kinds}
    prof, call_log = {}, []
    for n in range(max(1, opts.legs)):
        for kind in kinds:
            profile_this = in_region and n == 0 and kind == opts.inputs
            if profile_this:
                rc.profiled(True)
            elapsed, cpu_s, last, stats = timed_leg(resident if kind == "hbm" else None)
            if profile_this:
                prof, call_log = rc.collect()
                rc.profiled(False)
            if last is not None:
                n_valid = last[2]
            if comm is not None:
                elapsed = comm.maxF64(elapsed)
            legs[kind].append(dict(stats, elapsed=elapsed, cpu_s=cpu_s))
    cli_stage = None
    if opts.cli_samples > 0 and rc.rank == 0 and rc.world == 1:
        cli_stage = cli_typing_stage(opts.cli_samples, dev, dindex, gidx, inputs, resident, method)
    # ---- the roofline basis: the same step one sample at a time on ONE stream, no prefetch (kernels back to back)
    serial = None
    if opts.serial_steps > 0 and rc.rank == 0:
        run_steps(1, dev, dindex, gidx, inputs, method, depth=0, resident=resident)      # contexts of this mode warm
        rc.profiled(True)
        t1 = time.perf_counter()
        run_steps(opts.serial_steps, dev, dindex, gidx, inputs, method, depth=0, resident=resident)
        for d in rc.devices():
            d.sync()
        s_elapsed = time.perf_counter() - t1
        s_prof, s_log = rc.collect()
        rc.profiled(False)
        serial = {"prof": s_prof, "call_log": s_log, "steps": opts.serial_steps,
                  "ms_per_step": 1e3 * s_elapsed / opts.serial_steps}
    for buf in resident:
        buf.free()
    return {"legs": legs, "prof": prof, "call_log": call_log, "n_valid": n_valid, "serial": serial,
            "n_values": sharedLogTable(dev).known(),      # distinct probabilities met so far = entries of the log10 value table
            "search_steps": dict(SEARCH_STATS), "cli_stage": cli_stage,
            "h2d_bytes": inputs[0][0].nbytes, "record_bytes": inputs[0][0].record_bytes,
            "pairs": pairs, "method": method, "distinct": len(inputs), "opts": opts}


def report(rc: RankContext, res: dict, name: str, pinned, cores_before, head: bool) -> dict:
    """The JSON object of one workload (rank 0): the contract's keys for the headline, the same measurements under the
    workload's name for the others."""
    from kir_graph_amd import cohort, roofmodel
    args, opts, legs = rc.args, res["opts"], res["legs"]
    world, pairs, method, gidx = rc.world, res["pairs"], res["method"], rc.gidx

    def median_leg(kind):
        """The leg of ``kind`` with the median time (the slower of the middle two for an even count)."""
        rows = sorted(legs[kind], key=lambda r: r["elapsed"])
        return rows[len(rows) // 2]

    lead = median_leg(opts.inputs)
    elapsed = lead["elapsed"]
    ms_per_step = 1e3 * elapsed / opts.steps
    reads_per_step = 2 * pairs * world
    value = reads_per_step / (elapsed / opts.steps)
    prof, call_log = res["prof"], res["call_log"]
    if args.verbose:
        for k, (n, ms) in sorted(prof.items(), key=lambda kv: -kv[1][1]):
            log(f"[bench] {k:18s} launches {n:6d}  total {ms:9.3f} ms  avg {ms / n:8.4f} ms")
        log(f"[bench] kernel time {sum(v[1] for v in prof.values()) / opts.steps:.2f} ms of {ms_per_step:.2f} ms per step")
    lanes = os.environ.get("GK_SAMPLE_LANES", "3")
    out = {
        "metric": "typed 150 bp PE reads/s (pileup+EM) per GPU; achieved HBM GB/s vs roofline",
        "value": value, "unit": "reads/s", "n_gpus": world, "steps": opts.steps, "warmup": opts.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"{configName(pairs)}: 1 synthetic sample per step and GPU ({res['distinct']} distinct samples in "
                               f"rotation), {2 * pairs} 150 bp PE reads, synthetic example_index-shaped index "
                               f"({sum(len(t.alleles) for t in gidx.tables)} alleles, 15 genes), "
                               f"--allele-strategy {method}, top_n 600; "
                               + ("a step starts with the sample's packed records in pinned host memory: host-to-device "
                                  "copy + tabulation + typing + calls inside the timed region (SURVEY.md 8(d)); "
                                  "hbm_resident: the same steps with the records resident in HBM"
                                  if opts.inputs == "host" else
                                  "a step starts with the sample's records resident in HBM (tabulation + typing + calls "
                                  "inside the timed region); pcie_inclusive: the same steps from pinned host memory")
                               + f"; median of {len(legs[opts.inputs])} timed legs of {opts.steps} steps",
                   "inputs": opts.inputs, "allele_strategy": method,
                   "h2d_bytes_per_sample": res.get("h2d_bytes"), "record_bytes_per_sample": res.get("record_bytes"),
                   "pairs_per_sample": pairs, "pairs_passing_filter": int(res["n_valid"]),
                   "parallelism": f"samples sharded over {world} GPU(s), no data-path collective; one process per GPU, "
                                  f"{lanes} samples in flight (one host thread and one stream per sample: "
                                  + ("gk_sample_em)" if method in ("em", "report") else "gk_sample_search)"),
                   "rank_barrier": (rc.comm.backend if rc.comm is not None else None),
                   "rank_barrier_note": rc.comm_note},
    }
    serial = res.get("serial")
    if serial:
        s_prof, steps = serial["prof"], serial["steps"]
        table = {k: {"launches_per_step": n / steps, "ms_per_step": ms / steps, "avg_launch_ms": ms / n}
                 for k, (n, ms) in sorted(s_prof.items(), key=lambda kv: -kv[1][1])}
        out["kernels_serial"] = {"ms_per_step_wall": serial["ms_per_step"],
                                 "kernel_ms_per_step": sum(v[1] for v in s_prof.values()) / steps,
                                 "mode": "one sample at a time on one stream, no prefetch (GK_SAMPLE_LANES=1 "
                                         "GK_PREFETCH=0): kernels run back to back; names are the "
                                         "kernels' own (rocprofv3 --kernel-trace --stats lists the same names)",
                                 "kernels": table}
        out["roofline"] = roofmodel.dominant(s_prof, serial["call_log"])
        out["roofline"]["step"] = roofmodel.stepRoofline(serial["call_log"], steps, ms_per_step)
    else:
        out["roofline"] = roofmodel.dominant(prof, call_log)
        out["roofline"]["note_basis"] = "launch times taken inside the timed region (other samples share the GPU)"
    out["roofline"]["traffic"], out["roofline"]["traffic_source"] = measured_traffic(out["roofline"].get("kernel"), pairs, method)
    if prof:     # --verbose: launch times inside the timed region (kernels of the samples in flight overlap there)
        out["kernel_ms_per_step"] = {k: v[1] / opts.steps for k, v in prof.items()}
    out["search_steps"] = res.get("search_steps")    # steps bounded by integers / redone with f64 only (so far in this process)
    out["value_table_entries"] = res.get("n_values")  # distinct probabilities = log10 evaluations on the host
    cpu_s = float(lead["cpu_s"])
    out["host"] = {"host_core_s_per_step": cpu_s / opts.steps, "cores_busy": cpu_s / elapsed if elapsed else None,
                   "cores_per_gpu": args.cores_per_gpu or None, "pinned_to": pinned,
                   "cores_allowed": len(cores_before), "cgroup_quota_cores": cgroup_cores(),
                   "worker_processes": 1, "sample_lanes": int(lanes),
                   "search_slots": int(os.environ.get("GK_SEARCH_SLOTS", "0") or 0),
                   "cores_per_rank": cohort.hostCoresPerRank(),
                   "wait_policy": os.environ.get("GK_WAIT_POLICY", "runtime default"),
                   "note": "user + system time of rank 0's process over the reported leg (getrusage); "
                           "a host thread that spins on the GPU counts as busy"}

    def leg_rows(kind):
        """Every timed leg of a kind, in the order they ran: what a NEW sample costs shows in the first leg of a run
        with --distinct >= steps + warmup (value_table_new > 0 there, 0 in the later legs, which meet the same samples
        again); samples_repeated_pass = samples of the leg that brought a product without a log10, tables_patched = the
        gene tables that got those values patched in (gk_compat_patch: one pass over the table), tables_rewritten = the
        ones written again by the compatibility kernel."""
        return [{"ms_per_step": 1e3 * r["elapsed"] / opts.steps, "value_table_new": r.get("value_table_new", 0),
                 "value_table_new_per_sample": r.get("value_table_new", 0) / max(opts.steps, 1),
                 "samples_repeated_pass": r.get("samples_repeated_pass", 0),
                 "tables_rewritten": r.get("tables_rewritten", 0),
                 "tables_patched": r.get("tables_patched", 0)} for r in legs[kind]]

    out["legs"] = leg_rows(opts.inputs)
    first = legs[opts.inputs][0]
    out["value_table_new_per_sample"] = first.get("value_table_new", 0) / max(opts.steps, 1)
    out["samples_repeated_pass"] = first.get("samples_repeated_pass", 0)
    out["distinct_samples"] = res["distinct"]
    other_kind = [k for k in legs if k != opts.inputs]
    if other_kind:
        kind = other_kind[0]
        o = median_leg(kind)
        key = "pcie_inclusive" if kind == "host" else "hbm_resident"
        out[key] = {
            "value": reads_per_step / (o["elapsed"] / opts.steps), "unit": "reads/s",
            "ms_per_step": 1e3 * o["elapsed"] / opts.steps,
            "host_core_s_per_step": float(o["cpu_s"]) / opts.steps,
            "legs": leg_rows(kind),
            "note": (f"the same {opts.steps} steps with every sample's packed records starting in pinned host memory: "
                     "the host-to-device copy of each sample (compact records) and their expansion are inside the timed "
                     "region (staged two samples ahead of the typing)" if kind == "host" else
                     f"the same {opts.steps} steps with the records of the distinct samples resident in HBM before the "
                     "clock starts (a step = tabulation + typing + calls)") + f"; median of {len(legs[kind])} legs"}
    if res.get("cli_stage"):
        out["cli_typing_stage"] = dict(res["cli_stage"], vs_bench_step=res["cli_stage"]["ms_per_sample"] / ms_per_step)
    if opts.cpu_pairs and world == 1:      # the CPU leg runs on rank 0 of the single-GPU run only
        out["cpu_baseline"] = cpu_baseline(method, opts.cpu_pairs, n_way=head)
    if not head:        # a secondary workload: the contract's run-level keys stay with the headline
        for key in ("n_gpus", "higher_is_better", "scaling", "vs_baseline", "data", "search_steps", "value_table_entries"):
            out.pop(key, None)
        out["workload"] = name
    return out


# ------------------------------------------------------------------------------------------ launcher
def launch_ranks(args, argv):
    """`python bench.py --gpus N` without a launcher: start the N ranks as fresh processes (this process has
    not touched HIP and never will), supervise them -- a rank that fails ends the launch for all, at once -- and
    pass rank 0's JSON line on."""
    import uuid
    from kir_graph_amd.comm import superviseRanks
    rdzv = tempfile.mkdtemp(prefix="gk_bench_rdzv_")
    token = uuid.uuid4().hex
    procs = []
    with tempfile.TemporaryFile() as out0:
        for r in range(args.gpus):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus),
                       LOCAL_WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", GK_RDZV_DIR=rdzv, GK_RDZV_TOKEN=token)
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                          stdout=out0 if r == 0 else subprocess.DEVNULL))
        failed = superviseRanks(procs, rdzv, token)
        out0.seek(0)
        text = out0.read().decode(errors="replace")
    if failed:
        log(f"[bench] rank exit codes {[p.returncode for p in procs]}: fewer than {args.gpus} ranks finished")
        sys.exit(1)
    line = [x for x in text.splitlines() if x.startswith("{")]
    if not line:
        log("[bench] rank 0 printed no result line")
        sys.exit(1)
    print(line[-1], flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--pairs", type=int, default=None,
                    help="read pairs per sample: 10000000 = configs[2], 1000000 = configs[1].  Given (or --method given): "
                         "that ONE workload.  Neither given: configs[2] with --allele-strategy exonfirst as the line's "
                         "headline, and -- on one GPU -- `em` (configs[2]) and `configs1_pv` (configs[1]) as objects beside it")
    ap.add_argument("--method", default=None, help="--allele-strategy of the typing: pv, exonfirst, em")
    ap.add_argument("--distinct", type=int, default=None,
                    help="distinct samples a rank rotates through (default: 2 at configs[2] size, 8 below; a cohort types "
                         "every sample once: with N >= steps + warmup the first leg only meets samples nobody has typed before)")
    ap.add_argument("--legs", type=int, default=3, help="timed legs per kind of input; the median leg is reported")
    ap.add_argument("--cpu-pairs", type=int, default=20000, help="pairs for the CPU baseline sample (0 = skip)")
    ap.add_argument("--serial-steps", type=int, default=2,
                    help="steps of the one-process serial pass after the timed region (roofline basis; 0 = skip)")
    ap.add_argument("--inputs", choices=("host", "hbm"), default="host",
                    help="where a sample's records are when its step starts -- what `value` is measured on: pinned host "
                         "memory (the copy inside the step: SURVEY.md section 8(d), the metric), or resident in "
                         "HBM (a step = tabulation + typing + calls).  The other kind is timed too and reported beside it")
    ap.add_argument("--one-kind", "--no-pcie-leg", dest="both_legs", action="store_false",
                    help="time only the kind of leg --inputs names (no second object in the JSON line)")
    ap.add_argument("--cli-samples", type=int, default=12,
                    help="samples for the command line's typing stage (main.alleleTyping), timed after the legs as "
                         "`cli_typing_stage` (rank 0 of the one-GPU run only; 0 = skip)")
    ap.add_argument("--pairs-scale", type=float, default=1.0, help=argparse.SUPPRESS)      # tests: the default workloads, smaller
    ap.add_argument("--no-secondary", dest="secondary", action="store_false",
                    help="the headline workload only (no `em` / `configs1_pv` objects)")
    ap.add_argument("--synth-threads", type=int, default=0,
                    help="samples synthesised at a time (0: by the host cores and ranks of the node)")
    ap.add_argument("--cores-per-gpu", type=int, default=0,
                    help="pin every rank (its threads) to this many host cores of its own, before anything touches HIP "
                         "(0 = no pinning)")
    ap.add_argument("--verbose", action="store_true")
    ap.add_argument("--profile-host", action="store_true", help="cProfile one extra step to stderr")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return launch_ranks(args, sys.argv[1:])
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        log(f"[bench] --gpus {args.gpus} does not match WORLD_SIZE {world}: start one rank per GPU "
            f"(torchrun --nproc-per-node {args.gpus}, or no launcher at all)")
        sys.exit(2)
    cores_before = allowed_cores()
    pinned = pin_rank(local_rank, args.cores_per_gpu) if args.cores_per_gpu > 0 else None
    args.pinned_to = pinned
    # the host threads of a process (sample lanes, ingest) hold the interpreter lock only between library calls: a short
    # switch interval keeps one lane's Python from delaying another lane's next launch by the default 5 ms
    sys.setswitchinterval(0.0005)
    # blocking waits, sample lanes and search slots by the rank's host cores, the preamble on a high-priority stream: the
    # package's own defaults for a process that types a cohort (kir_graph_amd.main sets the same ones).  ONE process per
    # GPU: a sample is typed by one host thread on one stream, three to five samples at a time (GK_SAMPLE_LANES)
    from kir_graph_amd import cohort
    cohort.pipelineDefaults(1)

    if args.pairs is None and args.method is None:
        scale = args.pairs_scale      # tests: the three workloads at a fraction of their size
        names = ["configs2_exonfirst"] + (["em", "configs1_pv"] if args.secondary and world == 1 else [])
        plan = [(name, max(1000, int(WORKLOADS[name][1] * scale)), WORKLOADS[name][2], WORKLOADS[name][3]) for name in names]
    else:
        pairs = 1_000_000 if args.pairs is None else args.pairs
        plan = [("custom", pairs, args.method or "pv", 2 if pairs >= 5_000_000 else N_DISTINCT)]
    rc = RankContext(args, rank, local_rank)
    threads = args.synth_threads or max(1, min(4, cohort.hostCoresPerRank() // 3))
    results, by_size = [], {}
    from types import SimpleNamespace
    for k, (name, pairs, method, distinct) in enumerate(plan):
        distinct = max(1, min(args.distinct or distinct, args.steps + args.warmup))
        if (pairs, distinct) not in by_size:
            by_size.clear()                     # the samples of the workload before go (their pinned memory with them)
            by_size[(pairs, distinct)] = make_inputs(rc.sidx, rc.gidx, rc.by_gene, rank, pairs, distinct, threads)
        head = k == 0
        opts = SimpleNamespace(steps=args.steps, warmup=args.warmup, legs=args.legs if head else min(args.legs, 3),
                               inputs=args.inputs, both_legs=args.both_legs, serial_steps=args.serial_steps,
                               cli_samples=args.cli_samples if (head or name == "configs1_pv") else 0, cpu_pairs=args.cpu_pairs)
        t0 = time.time()
        res = measure(rc, by_size[(pairs, distinct)], pairs, method, opts)
        log(f"[bench] rank {rank}: workload {name} ({configName(pairs)}, {method}) measured in {time.time() - t0:.1f}s")
        results.append((name, res))
    if rank == 0:
        out = None
        for k, (name, res) in enumerate(results):
            obj = report(rc, res, name, pinned, cores_before, head=(k == 0))
            if k == 0:
                out = obj
            else:
                out[name] = obj
        print(json.dumps(out), flush=True)
    if rc.comm is not None:
        rc.comm.close()


def workloadTag(pairs: int, method: str) -> str:
    """Name of a workload in the file names under profiles/: cfg2_exonfirst, cfg2_em, cfg1_pv, ..."""
    size = {1_000_000: "cfg1", 10_000_000: "cfg2"}.get(pairs, f"p{pairs}")
    return f"{size}_{method}"


def measured_traffic(kernel, pairs, method):
    """(HBM bytes per launch of ``kernel``, where the figure comes from) from the committed PMC passes over the serial
    form of this workload (profiles/rNN_traffic_<workload>_<kernel>.json, made by tools/collect_profiles.sh) -- but only
    from a file that was measured on THIS code: the file records the digest of the device sources it ran
    (kir_graph_amd.build.sourceDigest) and a file with another digest, or none, is refused.  (None, why) then."""
    import glob
    from kir_graph_amd.build import sourceDigest
    digest = sourceDigest(kernel)
    tag = workloadTag(pairs, method)
    stale = []
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_traffic_{tag}_{kernel}.json")), reverse=True):
        try:
            with open(path) as f:
                t = json.load(f)
            if t.get("kernel") != kernel:
                continue
            if t.get("csrc_sha16") != digest:
                stale.append(os.path.basename(path))
                continue
            return float(t["traffic_bytes_per_launch"]), (f"profiles/{os.path.basename(path)} (rocprofv3 --pmc FETCH_SIZE / "
                                                          f"WRITE_SIZE passes of this workload, one sample at a time; "
                                                          f"device sources {digest} = the running code)")
        except (OSError, ValueError, KeyError):
            continue
    return None, (f"no PMC pass of workload {tag} on the running device sources ({digest}) is committed"
                  + (f"; refused as stale: {', '.join(stale)}" if stale else ""))


if __name__ == "__main__":
    main()

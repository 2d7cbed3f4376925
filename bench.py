#!/usr/bin/env python3
"""
bench.py -- typed 150 bp PE reads/s of the Graph-KIR hot path on MI355X.

One "step" = one synthetic sample (BASELINE.json configs[1]: 2 M reads = 1 M pairs, ~2 k alleles in 15 genes,
--allele-strategy pv == full, top_n 600, variant correction on) taken from packed alignment records to per-gene allele
calls on the host: the tabulation (gk_tabulate), the sample preamble (error correction, empty reads, zygosity tallies),
per gene the compatibility table through the log10 value table and the greedy multi-allele likelihood search
(gk_sample_search), and the allele selection.  The steps run through the package's own sample pipeline
(kir_graph_amd.cohort: stagedSamples -> typeSamples), the one `python -m kir_graph_amd.main` types a cohort with.

Two kinds of timed leg over the same K steps, each bracketed by a barrier + device synchronise on both sides, each kind
timed `--legs` times (default 3) with the MEDIAN leg reported:
  host  (`value`)         a sample's records start in PINNED HOST MEMORY, in the compact form they cross PCIe in (~30 bytes
                          per mate: packed.CompactMates): the host-to-device copy and the expansion into 128-byte
                          records are inside the region, staged two samples ahead of the typing;
  hbm   (`hbm_resident`)  the records of the distinct samples are resident in HBM before the clock starts.
`--inputs hbm` swaps the two (`value` from resident records, `pcie_inclusive` beside it).
Consecutive steps take DIFFERENT samples: `--distinct N` (default 8) distinct ones per rank in rotation, seeds
1031 + 7 rank + i.  With N >= steps + warmup the first leg types only samples nobody has typed before -- `legs[0]`
then says what a NEW sample costs (value_table_new_per_sample, samples_repeated_pass), the later legs what a repeated one.

``--gpus N``: N ranks, one per GPU, every rank types its own samples (cohort sharding: weak scaling, no
data-path collective); value = reads of all ranks / max-over-ranks time.  Started without a launcher
(`python bench.py --gpus N`) this process starts the N ranks itself -- as fresh processes, before
anything here touches the GPU -- and relays rank 0's line; under `torchrun` (RANK / WORLD_SIZE set) it
is one of the ranks.  Ranks meet through kir_graph_amd/comm.py (RCCL: barrier + max of the times).

Prints ONE JSON line on rank 0 (contract in the task statement) with extra objects:
  roofline        the dominant kernel of the step, from a ONE-PROCESS, SERIAL pass (one sample at a time on one
                  stream, no prefetch) run right after the timed legs: HIP-event time per launch, algorithmic
                  bytes / operations per launch (kir_graph_amd/roofmodel.py); `roofline.step` =
                  the algorithmic bytes of ALL launches of a step over the reported ms_per_step against 8 TB/s
  kernels_serial  per-kernel launches and time per step of that pass (the basis rocprofv3 reproduces, profiles/)
  legs            every timed leg of the headline kind in the order they ran
  cpu_baseline    the oracle (CPU restatement of the reference) on a bounded sample of the same
                  workload, one core and N-way over the host's cores
  host            what the step costs on the host: core-seconds per step (user + system time of every worker
                  process of rank 0 over the reported leg, getrusage), cores busy on average, the cores the rank was
                  allowed (``--cores-per-gpu K`` pins every rank and its workers to K cores of its own before
                  anything touches HIP: the budget an 8-GPU node leaves each rank)
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
    trace = os.environ.get("GK_BENCH_TRACE") == "1"      # a timeline of the host threads on stderr (tools/host_timeline.py)

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
        return calls, warn, n_valid, typer

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
    typer = oty.makeTyper("full" if method in ("pv", "full") else method, data, top_n=600, variant_correction=True)
    typer.typing(sample.gene_cn)
    t2 = time.time()
    if out is not None:
        out.put((t0, t2))
    return t1 - t0, t2 - t1


def cpu_baseline(method, n_pairs):
    """Oracle on a bounded sample of the same workload: one core, then N-way (one sample per core, the way the
    reference's own speed test runs it: research/test_speed.graphkir.par.sh, `parallel -j 14 --thread 1`)."""
    import multiprocessing as mp
    t_tab, t_typ = _oracle_leg((99, n_pairs, method))
    single = 2 * n_pairs / (t_tab + t_typ)
    cores = len(allowed_cores())
    quota = cgroup_cores()      # a container's CPU quota: the cores that can really run at once
    if quota is not None:
        cores = max(1, min(cores, quota))
    n_way = max(1, min(cores, 32))
    out = {"value": single, "unit": "reads/s", "cores": 1, "kind": "port",
           "sample": f"{n_pairs} pairs of the same synthetic workload (R_g <= 8 k per gene: the reference's own "
                     f"real-depth regime); oracle: tabulate {t_tab:.1f}s + typing {t_typ:.1f}s on one core",
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


# ------------------------------------------------------------------------------------------ one worker process
def worker(j, procs, opts, rank, local_rank, gang, timing=None, helpers_done=None):
    """Worker j of `procs` on this rank's GPU: builds the inputs, warms up, then types its share of the
    rank's `steps` samples between the common start and end.

    One Python process drives the GPU through ~25 host threads at most (gene workers, prefetch) and its
    interpreter lock serialises their host work; samples are independent, so a rank may run
    GK_PROCS_PER_GPU processes on its GPU, the same way a cohort run may place several ranks on one GPU.
    Worker 0 is the rank's own process and keeps the clock: the timed region starts when every worker
    (and every rank) is ready and ends when every worker's last sample is typed."""
    from types import SimpleNamespace
    args = SimpleNamespace(**opts)
    # the host threads of a process (sample lanes, ingest) hold the interpreter lock only between library calls: a short
    # switch interval keeps one lane's Python from delaying another lane's next launch by the default 5 ms
    sys.setswitchinterval(float(os.environ.get("GK_SWITCH_INTERVAL", "0.0005")))
    if j and os.environ.get("GK_BENCH_KILL_WORKER") == str(j):   # test hook: this worker dies at once
        os._exit(3)
    from kir_graph_amd import _lib, comm as gk_comm
    from kir_graph_amd.engine import DeviceIndex
    world = int(os.environ.get("WORLD_SIZE", "1"))
    n_dev = _lib.deviceCount()
    if n_dev == 0:
        raise RuntimeError("bench.py: no HIP device visible (the typing path has no CPU fallback)")
    backend = {"nccl": "rccl", "gloo": "file"}.get(os.environ.get("GK_BENCH_BACKEND", "rccl"),
                                                   os.environ.get("GK_BENCH_BACKEND", "rccl"))
    if world > 1 and backend == "rccl" and world > n_dev:
        raise RuntimeError(f"bench.py: {world} ranks but {n_dev} GPU(s): one rank per GPU "
                           "(GK_BENCH_BACKEND=file rehearses the multi-rank path on fewer GPUs)")
    dev = _lib.Device(local_rank % n_dev)
    sidx, gidx, by_gene = build_index()
    t_in = time.time()
    inputs, samples = [], []
    for i in range(max(1, min(args.distinct, args.steps + args.warmup))):
        sample, rec, table = build_sample(sidx, gidx, by_gene, 1031 + 7 * rank + i, args.pairs)
        inputs.append((PinnedRecords(rec), table, sample.gene_cn))
        samples.append(sample)
        del rec
    log(f"[bench] rank {rank} worker {j}: {len(inputs)} samples of {args.pairs} pairs in pinned memory, "
        f"{inputs[0][0].nbytes / 1e6:.0f} MB each in compact form ({inputs[0][0].record_bytes / 1e6:.0f} MB as 128-byte records; "
        f"{time.time() - t_in:.1f}s)")
    dindex = DeviceIndex(dev, gidx)
    dev.sync()
    comm = None
    if j == 0 and world > 1:
        # a scaling run must not quietly measure something else: when the RCCL communicator cannot be made on some
        # rank every rank stops with a non-zero code (the file backend is used only when it was asked for)
        try:
            comm = gk_comm.initFromEnv(dev=dev, backend=backend, fallback=False)
        except gk_comm.CommError as e:
            log(f"[bench] rank {rank}: {e}; not falling back (GK_BENCH_BACKEND=file rehearses the launch without RCCL)")
            os._exit(4)
        if comm.world != args.gpus or comm.backend != backend:
            raise RuntimeError(f"bench.py: --gpus {args.gpus} on {backend} but {comm.world} ranks joined on {comm.backend}")

    def claims():
        """Samples of the timed region for this worker: all of them, or whatever it gets from the shared counter."""
        if gang is None:
            yield from range(args.steps)
            return
        while True:
            with gang["next"].get_lock():
                k = gang["next"].value
                gang["next"].value = k + 1
            if k >= args.steps:
                return
            yield k

    def all_devices():
        return list(_lib.Device.instances)

    def gang_wait(name):
        if gang is not None:
            gang[name].wait(timeout=600)

    def profiled(on):
        for d in all_devices():
            d.profEnable(on)
            if on:
                d.profCollect()
                d.call_log = []

    def collect():
        prof, call_log = {}, []
        for d in all_devices():
            for k, (n, ms) in d.profCollect().items():
                n0, ms0 = prof.get(k, (0, 0.0))
                prof[k] = (n0 + n, ms0 + ms)
            call_log += d.call_log or []
        return prof, call_log

    # Two kinds of timed leg over the same steps.  "host": every sample's packed records start in pinned host memory, the
    # host-to-device copy (compact records) + expansion are inside the region (the metric) -- `value`.  "hbm": the records
    # of the distinct samples are resident in HBM before any clock starts, a step = tabulation + typing + calls --
    # `hbm_resident`.  Each kind is timed `--legs` times (alternating), the median leg is reported.
    from kir_graph_amd.typing_mulit_allele import sharedLogTable
    resident = [pinned.toDevice(dev) for pinned, _, _ in inputs]
    dev.sync()
    n_valid = 0
    if args.warmup:
        n_valid = run_steps(args.warmup, dev, dindex, gidx, inputs, args.method, resident=resident)[2]
        run_steps(min(args.warmup, 4), dev, dindex, gidx, inputs, args.method)      # the copy path's contexts and pools
    if getattr(args, "pinned_to", None):      # --cores-per-gpu: the runtime's own threads too (they exist by now)
        pin_all_threads(args.pinned_to)
    if j == 0 and getattr(args, "profile_host", False):
        import cProfile
        import pstats
        n_prof = int(os.environ.get("GK_PROFILE_STEPS", "1"))     # with GK_SAMPLE_LANES=1 GK_PREFETCH=0 everything is on this thread
        pr = cProfile.Profile()
        pr.enable()
        run_steps(n_prof, dev, dindex, gidx, inputs, args.method, resident=resident)
        pr.disable()
        log(f"[bench] host profile of {n_prof} step(s)")
        pstats.Stats(pr, stream=sys.stderr).sort_stats("cumulative").print_stats(45)
        pstats.Stats(pr, stream=sys.stderr).sort_stats("tottime").print_stats(45)
    # per-kernel events inside the timed region cost ~3 ms per step (a profiling signal per dispatch): only on request;
    # the roofline comes from the serial pass after the region
    in_region = bool(getattr(args, "verbose", False)) or os.environ.get("GK_BENCH_PROFILE") == "1"

    def timed_leg(leg_resident):
        """EXACTLY args.steps steps between a barrier + device synchronise on both sides; (seconds, host CPU seconds of
        this worker, the last step's result, what the value table and the searches did)."""
        if j == 0 and gang is not None:
            gang["next"].value = 0          # nobody claims before the "go" barrier below
        dev.sync()
        gang_wait("ready")
        if comm is not None:
            comm.barrier()          # RCCL all-reduce + stream synchronise: every rank is ready
        gang_wait("go")
        stats = {"value_table_at_start": sharedLogTable(dev).known()}
        t0 = time.perf_counter()
        cpu0 = cpu_seconds()
        by_thread0 = {tid: c for _, tid, c in thread_cpu_table()} if os.environ.get("GK_BENCH_TRACE") == "1" else None
        last = run_steps(claims(), dev, dindex, gidx, inputs, args.method, resident=leg_resident, stats=stats)
        for d in all_devices():
            d.sync()
        cpu = cpu_seconds() - cpu0          # this worker's host time for its share of the steps (waits that spin included)
        if by_thread0 is not None:          # GK_BENCH_TRACE=1: which threads the host time of the leg went to
            rows = sorted(((c - by_thread0.get(tid, 0.0), name, tid) for name, tid, c in thread_cpu_table()), reverse=True)
            log("[trace] host CPU of the leg by thread (ms per step): " +
                ", ".join(f"{name}/{tid} {1e3 * c / max(args.steps, 1):.2f}" for c, name, tid in rows if c > 0))
        gang_wait("done")
        if comm is not None:
            comm.barrier()
        elapsed = time.perf_counter() - t0
        stats["value_table_new"] = sharedLogTable(dev).known() - stats.pop("value_table_at_start")
        return elapsed, cpu, last, stats

    kinds = [k for k in ("host", "hbm") if k == args.inputs or args.both_legs]
    kinds.sort(key=lambda k: k != args.inputs)          # the headline's kind first: its first leg meets the new samples
    legs = {k: [] for k in kinds}
    prof, call_log = {}, []
    for n in range(max(1, args.legs)):
        for kind in kinds:
            profile_this = in_region and n == 0 and kind == args.inputs
            if profile_this:
                profiled(True)
            elapsed, cpu_s, last, stats = timed_leg(resident if kind == "hbm" else None)
            if profile_this:
                prof, call_log = collect()
                profiled(False)
            if last is not None:
                n_valid = last[2]
            if comm is not None:
                elapsed = comm.maxF64(elapsed)
            legs[kind].append(dict(stats, elapsed=elapsed, cpu_s=cpu_s))
    if j:
        gang["results"].put({"prof": prof, "call_log": call_log, "legs": legs})
        return None
    timing["legs"] = legs
    others = helpers_done() if helpers_done is not None else []      # the other workers have left the GPU
    cli_stage = None
    if args.cli_samples > 0 and rank == 0 and world == 1:
        cli_stage = cli_typing_stage(args.cli_samples, dev, dindex, gidx, inputs, resident, args.method)
    # ---- the roofline basis: the same step in ONE process, ONE gene thread, no prefetch (kernels back to back)
    serial = None
    if args.serial_steps > 0 and rank == 0:
        keep = os.environ.get("GK_THREADS")
        keep_streams = os.environ.get("GK_SAMPLE_STREAMS")
        os.environ["GK_THREADS"] = "1"
        os.environ["GK_SAMPLE_STREAMS"] = "1"       # one stream: the kernels of a sample run back to back
        try:
            run_steps(1, dev, dindex, gidx, inputs, args.method, depth=0, resident=resident)      # contexts of this mode warm
            profiled(True)
            t1 = time.perf_counter()
            run_steps(args.serial_steps, dev, dindex, gidx, inputs, args.method, depth=0, resident=resident)
            for d in all_devices():
                d.sync()
            s_elapsed = time.perf_counter() - t1
            s_prof, s_log = collect()
            profiled(False)
            serial = {"prof": s_prof, "call_log": s_log, "steps": args.serial_steps,
                      "ms_per_step": 1e3 * s_elapsed / args.serial_steps}
        finally:
            for name, val in (("GK_THREADS", keep), ("GK_SAMPLE_STREAMS", keep_streams)):
                if val is None:
                    os.environ.pop(name, None)
                else:
                    os.environ[name] = val
    from kir_graph_amd.typing_mulit_allele import SEARCH_STATS
    n_values = sharedLogTable(dev).known()      # distinct probabilities met so far = entries of the log10 value table
    for other in others:                        # the other worker processes of this rank: their host time counts too
        for kind, rows in other.get("legs", {}).items():
            for mine, theirs in zip(legs.get(kind, []), rows):
                mine["cpu_s"] += theirs["cpu_s"]
                for key in ("samples", "samples_repeated_pass", "tables_rewritten", "tables_patched", "value_table_new"):
                    mine[key] = mine.get(key, 0) + theirs.get(key, 0)
    return {"prof": prof, "call_log": call_log, "n_valid": n_valid, "gidx": gidx, "serial": serial, "comm": comm, "n_values": n_values,
            "search_steps": dict(SEARCH_STATS), "others": others, "cli_stage": cli_stage,
            "h2d_bytes": inputs[0][0].nbytes, "record_bytes": inputs[0][0].record_bytes}


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
    ap.add_argument("--steps", type=int, default=96)     # long enough for the ramp and the uneven finish of the workers not to show (1 % at 96 steps, 4 % at 24)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--pairs", type=int, default=1_000_000, help="read pairs per sample (config 2: 1e6)")
    ap.add_argument("--method", default="pv")
    ap.add_argument("--distinct", type=int, default=N_DISTINCT,
                    help="distinct samples a rank rotates through (a cohort types every sample once: with N >= steps + "
                         "warmup the first leg only meets samples nobody has typed before; 3.4 s of generation each)")
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
    ap.add_argument("--cores-per-gpu", type=int, default=0,
                    help="pin every rank (its worker processes and threads) to this many host cores of its own, "
                         "before anything touches HIP (0 = no pinning)")
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
    args.pinned_to = pinned          # handed to the workers (vars(args)): they bind the runtime's threads after the warm-up
    # Worker processes of this rank on its GPU (see worker()): started first, before anything touches HIP.
    # A sample is typed by ONE host thread on one stream (gk_sample_search: its genes pipelined on marks of the stream),
    # three to five samples at a time (GK_SAMPLE_LANES: by the host cores the rank has, cohort.pipelineDefaults) plus the
    # staging thread(s): ONE process keeps the GPU fed from two to four host cores (profiles/r04_sample_lanes.txt);
    # GK_PROCS_PER_GPU=2 adds a second worker process.  Waits block instead of spinning: a rank of an 8-GPU node may
    # have about two cores.
    procs = max(1, int(os.environ.get("GK_PROCS_PER_GPU", "1")))
    procs = min(procs, max(1, args.steps))
    # blocking waits, sample lanes and search slots by the rank's host cores, the preamble on a high-priority stream: the
    # package's own defaults for a process that types a cohort (kir_graph_amd.main sets the same ones)
    from kir_graph_amd import cohort
    cohort.pipelineDefaults(procs)
    own_threads = procs > 1 and "GK_THREADS" not in os.environ and os.environ.get("GK_SAMPLE_SEARCH") == "0"
    if own_threads:
        os.environ["GK_THREADS"] = "3"   # per-gene threads (the round-2 path): four processes of three shared the host cores
    gang, helpers = None, []
    if procs > 1:
        import multiprocessing as mp
        ctx = mp.get_context("spawn")
        gang = {"ready": ctx.Barrier(procs), "go": ctx.Barrier(procs), "done": ctx.Barrier(procs), "results": ctx.Queue(),
                "next": ctx.Value("i", 0)}   # the samples of the timed region are handed out one by one
        helpers = [ctx.Process(target=worker, args=(j, procs, vars(args), rank, local_rank, gang), daemon=True)
                   for j in range(1, procs)]
        try:
            for h in helpers:
                h.start()
        except OSError as e:   # no child processes here (e.g. under a profiler that forbids them): one process
            log(f"[bench] cannot start worker processes ({e}); running in one process")
            for h in helpers:
                if h.is_alive():
                    h.terminate()
            procs, gang = 1, None
            if own_threads:
                del os.environ["GK_THREADS"]

    finished = threading.Event()
    if gang is not None:
        def watch():   # a worker that exits early breaks the barriers at once instead of after their timeout
            while not finished.wait(0.5):
                if any(h.exitcode not in (None, 0) for h in helpers):
                    for name in ("ready", "go", "done"):
                        gang[name].abort()
                    return
        threading.Thread(target=watch, daemon=True).start()

    def helpers_done():
        """Results of the other workers, collected as soon as the timed region is over; the workers have exited
        (and released the GPU) when this returns."""
        got = [gang["results"].get(timeout=600) for _ in range(len(helpers))] if gang is not None else []
        for h in helpers:
            h.join(timeout=60)
        return got

    timing = {}
    try:
        res = worker(0, procs, vars(args), rank, local_rank, gang, timing=timing, helpers_done=helpers_done)
    except threading.BrokenBarrierError:
        # a worker process died or never came up; a single-GPU run starts over in one process, a multi-rank
        # run cannot (the other ranks are past their barriers)
        for h in helpers:
            if h.is_alive():
                h.terminate()
        if world > 1:
            raise
        log("[bench] a worker process failed; running the measurement in one process")
        procs, gang = 1, None
        if own_threads:
            del os.environ["GK_THREADS"]
        res = worker(0, 1, vars(args), rank, local_rank, None, timing=timing)
    legs = timing["legs"]

    def median_leg(kind):
        """The leg of ``kind`` with the median time (the slower of the middle two for an even count)."""
        rows = sorted(legs[kind], key=lambda r: r["elapsed"])
        return rows[len(rows) // 2]

    head = median_leg(args.inputs)
    elapsed = head["elapsed"]
    prof, call_log, n_valid, gidx = res["prof"], res["call_log"], res["n_valid"], res["gidx"]
    for other in res.get("others", []):
        for k, (n, ms) in other["prof"].items():
            n0, ms0 = prof.get(k, (0, 0.0))
            prof[k] = (n0 + n, ms0 + ms)
        call_log += other["call_log"]
    finished.set()
    if gang is not None:
        for h in helpers:
            h.join(timeout=30)

    if rank == 0:
        from kir_graph_amd import roofmodel
        ms_per_step = 1e3 * elapsed / args.steps
        reads_per_step = 2 * args.pairs * world
        value = reads_per_step / (elapsed / args.steps)
        if args.verbose:
            for k, (n, ms) in sorted(prof.items(), key=lambda kv: -kv[1][1]):
                log(f"[bench] {k:18s} launches {n:6d}  total {ms:9.3f} ms  avg {ms / n:8.4f} ms")
            log(f"[bench] kernel time {sum(v[1] for v in prof.values()) / args.steps:.2f} ms of {ms_per_step:.2f} ms per step")
        out = {
            "metric": "typed 150 bp PE reads/s (pileup+EM) per GPU; achieved HBM GB/s vs roofline",
            "value": value, "unit": "reads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{'configs[1]' if args.pairs == 1_000_000 else 'configs[2]' if args.pairs == 10_000_000 else 'custom'}: "
                                   f"1 synthetic sample per step and GPU ({args.distinct} distinct samples in "
                                   f"rotation), {2 * args.pairs} 150 bp PE reads, synthetic example_index-shaped index "
                                   f"({sum(len(t.alleles) for t in gidx.tables)} alleles, 15 genes), "
                                   f"--allele-strategy {args.method}, top_n 600; "
                                   + ("a step starts with the sample's packed records in pinned host memory: host-to-device "
                                      "copy + tabulation + typing + calls inside the timed region (SURVEY.md 8(d)); "
                                      "hbm_resident: the same steps with the records resident in HBM"
                                      if args.inputs == "host" else
                                      "a step starts with the sample's records resident in HBM (tabulation + typing + calls "
                                      "inside the timed region); pcie_inclusive: the same steps from pinned host memory")
                                   + f"; median of {len(legs[args.inputs])} timed legs of {args.steps} steps",
                       "inputs": args.inputs,
                       "h2d_bytes_per_sample": res.get("h2d_bytes"), "record_bytes_per_sample": res.get("record_bytes"),
                       "pairs_per_sample": args.pairs, "pairs_passing_filter": int(n_valid),
                       "parallelism": f"samples sharded over {world} GPU(s), no data-path collective; "
                                      f"{procs} worker process(es) per GPU, {os.environ.get('GK_SAMPLE_LANES', '2')} samples in flight "
                                      "each (one host thread and one stream per sample: gk_sample_search)"
                                      if os.environ.get("GK_SAMPLE_SEARCH") != "0" else
                                      f"samples sharded over {world} GPU(s), no data-path collective; {procs} worker "
                                      f"process(es) per GPU, {os.environ.get('GK_THREADS', '6')} gene threads each",
                       "rank_barrier": (res["comm"].backend if res.get("comm") is not None else None)},
        }
        serial = res.get("serial")
        if serial:
            s_prof, steps = serial["prof"], serial["steps"]
            table = {k: {"launches_per_step": n / steps, "ms_per_step": ms / steps, "avg_launch_ms": ms / n}
                     for k, (n, ms) in sorted(s_prof.items(), key=lambda kv: -kv[1][1])}
            out["kernels_serial"] = {"ms_per_step_wall": serial["ms_per_step"],
                                     "kernel_ms_per_step": sum(v[1] for v in s_prof.values()) / steps,
                                     "mode": "one process, one sample at a time on one stream, no prefetch (GK_PROCS_PER_GPU=1 "
                                             "GK_SAMPLE_LANES=1 GK_SAMPLE_STREAMS=1 GK_PREFETCH=0): kernels run back to back",
                                     "kernels": table}
            out["roofline"] = roofmodel.dominant(s_prof, serial["call_log"])
            out["roofline"]["step"] = roofmodel.stepRoofline(serial["call_log"], steps, ms_per_step)
        else:
            out["roofline"] = roofmodel.dominant(prof, call_log)
            out["roofline"]["note_basis"] = "launch times taken inside the timed region (other workers share the GPU)"
        out["roofline"]["traffic"], out["roofline"]["traffic_source"] = measured_traffic(out["roofline"].get("kernel"))
        if prof:     # --verbose: launch times inside the timed region (kernels of all workers overlap there)
            out["kernel_ms_per_step"] = {k: v[1] / args.steps for k, v in prof.items()}
        out["search_steps"] = res.get("search_steps")    # worker 0: steps bounded by integers / redone with f64 only
        out["value_table_entries"] = res.get("n_values")  # worker 0: distinct probabilities = log10 evaluations on the host
        cpu_s = float(head["cpu_s"])
        out["host"] = {"host_core_s_per_step": cpu_s / args.steps, "cores_busy": cpu_s / elapsed if elapsed else None,
                       "cores_per_gpu": args.cores_per_gpu or None, "pinned_to": pinned,
                       "cores_allowed": len(cores_before), "cgroup_quota_cores": cgroup_cores(),
                       "worker_processes": procs, "sample_lanes": int(os.environ.get("GK_SAMPLE_LANES", "2")),
                       "search_slots": int(os.environ.get("GK_SEARCH_SLOTS", "0") or 0),
                       "cores_per_rank": cohort.hostCoresPerRank(),
                       "wait_policy": os.environ.get("GK_WAIT_POLICY", "runtime default"),
                       "note": "user + system time of rank 0's worker processes over the reported leg (getrusage); "
                               "a host thread that spins on the GPU counts as busy"}

        def leg_rows(kind):
            """Every timed leg of a kind, in the order they ran: what a NEW sample costs shows in the first leg of a run
            with --distinct >= steps + warmup (value_table_new > 0 there, 0 in the later legs, which meet the same samples
            again); samples_repeated_pass = samples of the leg that brought a product without a log10, tables_patched = the
            gene tables that got those values patched in (gk_compat_patch: one pass over the table), tables_rewritten = the
            ones written again by the compatibility kernel."""
            return [{"ms_per_step": 1e3 * r["elapsed"] / args.steps, "value_table_new": r.get("value_table_new", 0),
                     "value_table_new_per_sample": r.get("value_table_new", 0) / max(args.steps, 1),
                     "samples_repeated_pass": r.get("samples_repeated_pass", 0),
                     "tables_rewritten": r.get("tables_rewritten", 0),
                     "tables_patched": r.get("tables_patched", 0)} for r in legs[kind]]

        out["legs"] = leg_rows(args.inputs)
        first = legs[args.inputs][0]
        out["value_table_new_per_sample"] = first.get("value_table_new", 0) / max(args.steps, 1)
        out["samples_repeated_pass"] = first.get("samples_repeated_pass", 0)
        out["distinct_samples"] = args.distinct
        other_kind = [k for k in legs if k != args.inputs]
        if other_kind:
            kind = other_kind[0]
            o = median_leg(kind)
            name = "pcie_inclusive" if kind == "host" else "hbm_resident"
            out[name] = {
                "value": reads_per_step / (o["elapsed"] / args.steps), "unit": "reads/s",
                "ms_per_step": 1e3 * o["elapsed"] / args.steps,
                "host_core_s_per_step": float(o["cpu_s"]) / args.steps,
                "legs": leg_rows(kind),
                "note": (f"the same {args.steps} steps with every sample's packed records starting in pinned host memory: "
                         "the host-to-device copy of each sample (compact records) and their expansion are inside the timed "
                         "region (staged two samples ahead of the typing)" if kind == "host" else
                         f"the same {args.steps} steps with the records of the distinct samples resident in HBM before the "
                         "clock starts (a step = tabulation + typing + calls)") + f"; median of {len(legs[kind])} legs"}
        if res.get("cli_stage"):
            out["cli_typing_stage"] = dict(res["cli_stage"], vs_bench_step=res["cli_stage"]["ms_per_sample"] / ms_per_step)
        if args.cpu_pairs and world == 1:      # the CPU leg runs on rank 0 of the single-GPU run only
            out["cpu_baseline"] = cpu_baseline(args.method, args.cpu_pairs)
        print(json.dumps(out), flush=True)
    if res.get("comm") is not None:
        res["comm"].close()


def measured_traffic(kernel):
    """(HBM bytes per launch of ``kernel``, where the figure comes from) from the committed PMC passes over the serial
    form of this command (profiles/rNN_traffic_<kernel>.json, made by tools/collect_profiles.sh) -- but only from a
    file that was measured on THIS code: the file records the digest of the device sources it ran
    (kir_graph_amd.build.sourceDigest) and a file with another digest, or none, is refused.  (None, why) then."""
    import glob
    from kir_graph_amd.build import sourceDigest
    digest = sourceDigest(kernel)
    stale = []
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_traffic_{kernel}.json")), reverse=True):
        try:
            with open(path) as f:
                t = json.load(f)
            if t.get("kernel") != kernel:
                continue
            if t.get("csrc_sha16") != digest:
                stale.append(os.path.basename(path))
                continue
            return float(t["traffic_bytes_per_launch"]), (f"profiles/{os.path.basename(path)} (rocprofv3 --pmc FETCH_SIZE / "
                                                          f"WRITE_SIZE passes of this command, one process, serial; "
                                                          f"device sources {digest} = the running code)")
        except (OSError, ValueError, KeyError):
            continue
    return None, (f"no PMC pass of the running device sources ({digest}) is committed"
                  + (f"; refused as stale: {', '.join(stale)}" if stale else ""))


if __name__ == "__main__":
    main()

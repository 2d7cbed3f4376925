#!/usr/bin/env python3
"""
bench.py -- typed 150 bp PE reads/s of the Graph-KIR hot path on MI355X.

One "step" = one synthetic sample (BASELINE.json configs[1]: 2 M reads = 1 M pairs, ~2 k alleles in
15 genes, --allele-strategy pv == full, top_n 600, variant correction on) taken from packed alignment
records in PINNED HOST MEMORY to per-gene allele calls on the host: the
host-to-device copy of the records (256 MB), the tabulation (gk_tabulate), per gene the error
correction, compatibility table, log table and greedy multi-allele likelihood search, and the allele
selection all lie inside the timed region.  Consecutive steps take DIFFERENT samples (three distinct
ones per rank, in rotation) and are pipelined like the samples of a cohort: the copy + tabulation of the
next sample runs while the current one is typed.

``--gpus N``: N ranks, one per GPU, every rank types its own samples (cohort sharding: weak scaling, no
data-path collective); value = reads of all ranks / max-over-ranks time.  Started without a launcher
(`python bench.py --gpus N`) this process starts the N ranks itself -- as fresh processes, before
anything here touches the GPU -- and relays rank 0's line; under `torchrun` (RANK / WORLD_SIZE set) it
is one of the ranks.  Ranks meet through kir_graph_amd/comm.py (RCCL: barrier + max of the times).

Prints ONE JSON line on rank 0 (contract in the task statement) with extra objects:
  roofline        the dominant kernel of the step, from a ONE-PROCESS, SERIAL pass (one gene thread, no
                  prefetch) run right after the timed region: HIP-event time per launch, algorithmic
                  bytes / operations per launch (kir_graph_amd/roofmodel.py)
  kernels_serial  per-kernel launches and time per step of that pass (the basis rocprofv3 reproduces:
                  GK_PROCS_PER_GPU=1 GK_THREADS=1 GK_PREFETCH=0, profiles/)
  cpu_baseline    the oracle (CPU restatement of the reference) on a bounded sample of the same
                  workload, one core and N-way over the host's cores
  host            what the step costs on the host: core-seconds per step (user + system time of every worker
                  process of rank 0 over the timed region, getrusage), cores busy on average, the cores the rank was
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
N_DISTINCT = 3          # distinct samples a rank rotates through


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
    """The packed records of a sample in pinned host memory (gk_host_alloc): the start of a step."""

    def __init__(self, rec):
        import numpy as np
        from kir_graph_amd._lib import check, lib
        self.nbytes, self.count, self.dtype = rec.nbytes, len(rec), rec.dtype
        p = C.c_void_p()
        check(lib().gk_host_alloc(self.nbytes, C.byref(p)))
        self.ptr = p.value
        view = np.ctypeslib.as_array((C.c_uint8 * self.nbytes).from_address(self.ptr))
        view[:] = rec.view(np.uint8).reshape(-1)

    def toDevice(self, dev):
        """Queue the copy on ``dev``'s stream; kernels launched on that stream afterwards see the records."""
        from kir_graph_amd._lib import check, lib
        buf = dev.alloc(self.count, self.dtype)
        check(lib().gk_h2d_async(dev.ctx, buf.ptr, C.c_void_p(self.ptr), self.nbytes))
        return buf

    def free(self):
        from kir_graph_amd._lib import lib
        if self.ptr:
            lib().gk_host_free(C.c_void_p(self.ptr))
            self.ptr = 0


# ------------------------------------------------------------------------------------------ steps
def run_steps(items, dev, dindex, gidx, inputs, method, depth=None, threads=None, resident=None):
    """Types the samples ``inputs[k % len(inputs)]`` for k in ``items``: pinned records -> HBM -> tabulation ->
    typing -> calls.  Like a cohort run the samples go through ``cohort.prefetched``: copy + tabulation of
    the next sample are issued (on their own stream) while the current one is typed.  Every copy,
    tabulation and typing of the listed samples starts and ends inside this call.  ``resident``: the records of the
    distinct samples already in HBM (one device buffer per entry of ``inputs``) -- a step then starts at the tabulation."""
    from kir_graph_amd.cohort import overlapped, prefetched
    from kir_graph_amd.engine import Tabulation
    from kir_graph_amd.hisat2 import SampleData
    from kir_graph_amd.kir_typing import hostThreads, selectKirTypingModel
    if method == "exonfirst":      # the command line types `--allele-strategy exonfirst` as exonfirst_1 (main.py:186-187)
        method = "exonfirst_1"
    depth = int(os.environ.get("GK_PREFETCH", "1")) if depth is None else depth
    lanes = int(os.environ.get("GK_SAMPLE_LANES", "2"))   # samples typed at a time, each on a host thread and a stream of its own
    # Staging has contexts of its own (the typing lanes use workers 0..lanes*n-1): one for the copy of a sample's
    # records into HBM, one -- with a high-priority stream -- for its tabulation.  The two are stages of a pipeline
    # (GK_COPY_AHEAD=1, default): while sample k is typed, sample k+1 is tabulated and the records of k+2 are on their
    # way, each stage on a thread of its own.  In one stage (GK_COPY_AHEAD=0) a sample's staging took 6 - 7 ms of wall time
    # next to the typing kernels -- 80 % of a worker process's budget per sample.
    ingest = dev.worker(lanes * hostThreads(), urgent=True)
    copier = dev.worker(lanes * hostThreads() + 1)
    copy_ahead = os.environ.get("GK_COPY_AHEAD", "1") != "0"

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
        k, mates = item
        _, table, gene_cn = inputs[k % len(inputs)]
        tab = Tabulation(dindex, mates, dev=ingest)
        note("stage", k, t0)
        return tab, table, gene_cn, k

    def stage(k):                           # both in one go: queued on one stream, one after the other
        t0 = time.perf_counter()
        pinned, table, gene_cn = inputs[k % len(inputs)]
        tab = Tabulation(dindex, pinned.toDevice(ingest), dev=ingest)
        note("stage", k, t0)
        return tab, table, gene_cn, k

    def from_hbm(k):                        # the records are in HBM already: the step starts here
        return tabulate((k, resident[k % len(inputs)]))

    def type_one(item, lane):
        t0 = time.perf_counter()
        tab, table, gene_cn, k = item
        data = SampleData(tab, gidx, None, ins_strings=table.strings)
        typer = selectKirTypingModel(method, data, top_n=600, variant_correction=True)
        typer.slot_base = lane * hostThreads()
        calls, warn = typer.typing(gene_cn)
        n_valid = tab.n_valid
        tab.close()
        if resident is None:
            tab.mates.free()
        note("type", k, t0)
        return calls, warn, n_valid, typer

    out = None
    items = range(items) if isinstance(items, int) else items
    if depth <= 0:
        for k in items:
            out = type_one(from_hbm(k) if resident is not None else stage(k), 0)
        return out
    if resident is not None:
        staged = prefetched(items, from_hbm, depth=depth)
    elif copy_ahead:
        staged = prefetched(prefetched(items, copy_in, depth=depth), tabulate, depth=depth)
    else:
        staged = prefetched(items, stage, depth=depth)
    for out in overlapped(staged, type_one, lanes=lanes):
        pass
    return out


def _oracle_leg(job):
    """One CPU-baseline job (runs in a fresh process for the N-way leg): tabulate + type with the oracle."""
    seed, n_pairs, method = job
    sys.path.insert(0, ROOT)
    from kir_graph_amd import synth
    from oracle import tabulate as ot, typing as oty
    sidx, gidx, by_gene = build_index()
    sample = synth.makeSample(sidx, seed=seed, n_pairs=n_pairs, variants_by_gene=by_gene)
    lines = synth.toSamLines(sample, with_zs=False)
    t0 = time.time()
    data = ot.tabulateLines(lines, gidx.variants)
    t1 = time.time()
    typer = oty.makeTyper("full" if method in ("pv", "full") else method, data, top_n=600, variant_correction=True)
    typer.typing(sample.gene_cn)
    t2 = time.time()
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
        t0 = time.time()
        with ctx.Pool(n_way) as pool:
            pool.map(_oracle_leg, [(100 + i, n_pairs, method) for i in range(n_way)])
        wall = time.time() - t0
        out["n_way"] = {"value": 2 * n_pairs * n_way / wall, "unit": "reads/s", "cores": n_way,
                        "sample": f"{n_way} samples of {n_pairs} pairs, one process per core, wall {wall:.1f}s "
                                  "(inputs generated inside the clock's processes: ~10 % of it)"}
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
    log(f"[bench] rank {rank} worker {j}: {len(inputs)} samples of {args.pairs} pairs in pinned memory "
        f"({time.time() - t_in:.1f}s)")
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

    # The records of the distinct samples in HBM before any clock starts (--inputs hbm, the default): `value` is measured
    # with the inputs resident, a step = tabulation + typing + calls.  A second leg times the same steps with every
    # sample's records starting in pinned host memory (the 256 MB copy inside the region): `pcie_inclusive`.
    resident = None
    if args.inputs == "hbm":
        resident = [pinned.toDevice(dev) for pinned, _, _ in inputs]
        dev.sync()
    pcie_leg = resident is not None and args.pcie_leg
    n_valid = 0
    if args.warmup:
        n_valid = run_steps(args.warmup, dev, dindex, gidx, inputs, args.method, resident=resident)[2]
        if pcie_leg:
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
        this worker, the last step's result)."""
        if j == 0 and gang is not None:
            gang["next"].value = 0          # nobody claims before the "go" barrier below
        dev.sync()
        gang_wait("ready")
        if comm is not None:
            comm.barrier()          # RCCL all-reduce + stream synchronise: every rank is ready
        gang_wait("go")
        t0 = time.perf_counter()
        cpu0 = cpu_seconds()
        by_thread0 = {tid: c for _, tid, c in thread_cpu_table()} if os.environ.get("GK_BENCH_TRACE") == "1" else None
        last = run_steps(claims(), dev, dindex, gidx, inputs, args.method, resident=leg_resident)
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
        return time.perf_counter() - t0, cpu, last

    if in_region:
        profiled(True)
    elapsed, cpu_s, last = timed_leg(resident)
    if last is not None:
        n_valid = last[2]
    prof, call_log = collect() if in_region else ({}, [])
    profiled(False)
    pcie_elapsed, pcie_cpu_s = None, 0.0
    if pcie_leg:
        pcie_elapsed, pcie_cpu_s, _ = timed_leg(None)
    if j:
        gang["results"].put({"prof": prof, "call_log": call_log, "cpu_s": cpu_s, "pcie_cpu_s": pcie_cpu_s})
        return None
    if comm is not None:
        elapsed = comm.maxF64(elapsed)
        if pcie_elapsed is not None:
            pcie_elapsed = comm.maxF64(pcie_elapsed)
    timing["elapsed"] = elapsed
    timing["pcie_elapsed"] = pcie_elapsed
    others = helpers_done() if helpers_done is not None else []      # the other workers have left the GPU
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
    from kir_graph_amd.typing_mulit_allele import SEARCH_STATS, sharedLogTable
    n_values = sharedLogTable(dev).known()      # distinct probabilities met so far = entries of the log10 value table
    return {"prof": prof, "call_log": call_log, "n_valid": n_valid, "gidx": gidx, "serial": serial, "comm": comm, "n_values": n_values,
            "search_steps": dict(SEARCH_STATS), "others": others, "cpu_s": cpu_s + sum(o.get("cpu_s", 0.0) for o in others),
            "pcie_cpu_s": pcie_cpu_s + sum(o.get("pcie_cpu_s", 0.0) for o in others)}


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
    ap.add_argument("--distinct", type=int, default=N_DISTINCT, help="distinct samples a rank rotates through")
    ap.add_argument("--cpu-pairs", type=int, default=20000, help="pairs for the CPU baseline sample (0 = skip)")
    ap.add_argument("--serial-steps", type=int, default=2,
                    help="steps of the one-process serial pass after the timed region (roofline basis; 0 = skip)")
    ap.add_argument("--inputs", choices=("hbm", "host"), default="hbm",
                    help="where a sample's records are when its step starts: resident in HBM (the metric), or in pinned "
                         "host memory (the 256 MB copy inside the step)")
    ap.add_argument("--no-pcie-leg", dest="pcie_leg", action="store_false",
                    help="skip the second timed leg (the same steps from pinned host memory, reported as pcie_inclusive)")
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
    # three samples at a time (GK_SAMPLE_LANES) plus the staging thread(s): ONE process keeps the GPU fed from two to
    # three host cores (profiles/r03_host_budget.txt, profiles/r03_default_layout.txt); GK_PROCS_PER_GPU=2 adds a second
    # worker process (the default until the end of round 3).  Waits block instead of spinning: a rank of an 8-GPU
    # node may have about two cores.
    os.environ.setdefault("GK_WAIT_POLICY", "block")
    procs = max(1, int(os.environ.get("GK_PROCS_PER_GPU", "1")))
    procs = min(procs, max(1, args.steps))
    # the sample preamble on a high-priority stream: what lets ONE process keep the GPU busy (8.9 against 10.2 ms per
    # sample with three lanes); with two processes it takes CUs from the other process's search at the wrong moments
    # (9.4 against 8.3 ms) -- profiles/r03_stream_priority.txt
    os.environ.setdefault("GK_URGENT_PREAMBLE", "1" if procs == 1 else "0")
    os.environ.setdefault("GK_SAMPLE_LANES", "3" if procs == 1 else "2")      # samples in flight per worker process
    # ... of which two at a time are in their search (the third has its preamble done and starts the moment a search
    # ends): two searches fill the GPU, a third next to them lengthens all three and the tail of a short run
    # (profiles/r03_search_slots.txt: 8.13 against 8.84 ms per sample on the driver's 20 steps, the same on 64)
    if procs == 1:
        os.environ.setdefault("GK_SEARCH_SLOTS", "2")
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
    elapsed = timing["elapsed"]
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
                                   + ("a step starts with the sample's records resident in HBM (tabulation + typing + calls "
                                      "inside the timed region); pcie_inclusive: the same steps from pinned host memory"
                                      if args.inputs == "hbm" else
                                      "records start in pinned host memory (H2D inside the timed region)"),
                       "inputs": args.inputs,
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
        else:
            out["roofline"] = roofmodel.dominant(prof, call_log)
            out["roofline"]["note_basis"] = "launch times taken inside the timed region (other workers share the GPU)"
        out["roofline"]["traffic"], out["roofline"]["traffic_source"] = measured_traffic(out["roofline"].get("kernel"))
        if prof:     # --verbose: launch times inside the timed region (kernels of all workers overlap there)
            out["kernel_ms_per_step"] = {k: v[1] / args.steps for k, v in prof.items()}
        out["search_steps"] = res.get("search_steps")    # worker 0: steps bounded by integers / redone with f64 only
        out["value_table_entries"] = res.get("n_values")  # worker 0: distinct probabilities = log10 evaluations on the host
        cpu_s = float(res.get("cpu_s", 0.0))
        out["host"] = {"host_core_s_per_step": cpu_s / args.steps, "cores_busy": cpu_s / elapsed if elapsed else None,
                       "cores_per_gpu": args.cores_per_gpu or None, "pinned_to": pinned,
                       "cores_allowed": len(cores_before), "cgroup_quota_cores": cgroup_cores(),
                       "worker_processes": procs, "sample_lanes": int(os.environ.get("GK_SAMPLE_LANES", "2")),
                       "wait_policy": os.environ.get("GK_WAIT_POLICY", "runtime default"),
                       "note": "user + system time of rank 0's worker processes over the timed region (getrusage); "
                               "a host thread that spins on the GPU counts as busy"}
        pcie_elapsed = timing.get("pcie_elapsed")
        if pcie_elapsed:
            out["pcie_inclusive"] = {
                "value": reads_per_step / (pcie_elapsed / args.steps), "unit": "reads/s",
                "ms_per_step": 1e3 * pcie_elapsed / args.steps,
                "host_core_s_per_step": float(res.get("pcie_cpu_s", 0.0)) / args.steps,
                "note": f"a second timed leg of the same {args.steps} steps with every sample's packed records starting in "
                        f"pinned host memory: the {256 * args.pairs // 1_000_000} MB host-to-device copy of each sample is "
                        "inside the timed region (staged two samples ahead of the typing)"}
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

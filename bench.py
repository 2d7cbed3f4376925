#!/usr/bin/env python3
"""
bench.py -- typed 150 bp PE reads/s of the Graph-KIR hot path on MI355X.

One "step" = one synthetic sample (BASELINE.json configs[1]: 2 M reads = 1 M pairs, ~2 k alleles in
15 genes, --allele-strategy pv == full, top_n 600, variant correction on) taken from packed
alignment records ALREADY RESIDENT IN HBM to per-gene allele calls on the host:
tabulation (gk_tabulate) -> per gene: error correction, compatibility table, log table, greedy
multi-allele likelihood search -> allele selection.  N > 1: every rank types its own sample(s)
(cohort sharding, no data-path collective), value = reads of all ranks / max-over-ranks time.

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline      dominant kernel, HIP-event time measured in this run (gk_prof_*), algorithmic bytes
  cpu_baseline  the oracle (CPU restatement of the reference) on a bounded sample of the same workload
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import threading
import time


ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def build_inputs(seed: int, n_pairs: int, index_seed: int = 2022):
    from kir_graph_amd import synth, packed
    from kir_graph_amd.index import GkIndex
    t = time.time()
    sidx = synth.makeIndex(seed=index_seed)
    gidx = GkIndex.fromVariants(sidx.variants, genes=sidx.genes, exons=sidx.exons)
    by_gene = {}
    for v in sidx.variants:
        by_gene.setdefault(v.ref, []).append(v)
    sample = synth.makeSample(sidx, seed=seed, n_pairs=n_pairs, variants_by_gene=by_gene)
    rec, table = packed.packSample(sample, gidx)
    log(f"[bench] inputs: {len(gidx.variants)} variants, {sum(len(t.alleles) for t in gidx.tables)} alleles, "
        f"{n_pairs} pairs in {time.time() - t:.1f}s")
    return sidx, gidx, sample, rec, table


def run_steps(n_steps, dev, dindex, gidx, mates_buf, table, gene_cn, method):
    """``n_steps`` samples: tabulation + typing of records resident in HBM.

    Like a cohort run, the samples go through ``cohort.prefetched`` and ``cohort.overlapped``: the
    tabulation of the next sample is issued while the current one is typed; with GK_SAMPLE_LANES=2
    two samples are typed at a time (each on its own block of worker streams).  Every tabulation and
    typing of the ``n_steps`` samples starts and ends inside this call."""
    from kir_graph_amd.cohort import overlapped, prefetched
    from kir_graph_amd.engine import Tabulation
    from kir_graph_amd.hisat2 import SampleData
    from kir_graph_amd.kir_typing import hostThreads, selectKirTypingModel
    depth = int(os.environ.get("GK_PREFETCH", "1"))
    lanes = int(os.environ.get("GK_SAMPLE_LANES", "1"))   # 2: two samples typed at a time (gain varies from box to box)
    ingest = dev.worker(lanes * hostThreads())   # a context of its own: the typing lanes use workers 0..lanes*n-1

    def type_one(tab, lane):
        data = SampleData(tab, gidx, None, ins_strings=table.strings)
        typer = selectKirTypingModel(method, data, top_n=600, variant_correction=True)
        typer.slot_base = lane * hostThreads()
        calls, warn = typer.typing(gene_cn)
        n_valid = tab.n_valid
        tab.close()
        return calls, warn, n_valid, typer

    out = None
    # n_steps: a count, or an iterator that hands out the samples of a region shared with other workers
    items = range(n_steps) if isinstance(n_steps, int) else n_steps
    tabs = prefetched(items, lambda _: Tabulation(dindex, mates_buf, dev=ingest), depth=depth)
    for out in overlapped(tabs, type_one, lanes=lanes):
        pass
    return out


def cpu_baseline(sidx, gidx, gene_cn, method, n_pairs, seed):
    """Oracle on a bounded sample of the same workload (single core)."""
    from kir_graph_amd import synth
    from oracle import tabulate as ot, typing as oty
    sample = synth.makeSample(sidx, seed=seed, n_pairs=n_pairs, gene_cn=gene_cn)
    lines = synth.toSamLines(sample, with_zs=False)
    t0 = time.time()
    data = ot.tabulateLines(lines, gidx.variants)
    t1 = time.time()
    typer = oty.makeTyper("full" if method in ("pv", "full") else method, data, top_n=600, variant_correction=True)
    typer.typing(gene_cn)
    t2 = time.time()
    return {"value": 2 * n_pairs / (t2 - t0), "unit": "reads/s", "cores": 1, "kind": "port",
            "sample": f"{n_pairs} pairs of the same synthetic workload (oracle: tabulate {t1 - t0:.1f}s + "
                      f"typing {t2 - t1:.1f}s)"}


def worker(j, procs, opts, rank, local_rank, gang, rank_barrier=None, timing=None):
    """Worker j of `procs` on this rank's GPU: builds the inputs, warms up, then types its share of the
    rank's `steps` samples between the common start and end.

    One Python process drives the GPU through ~25 host threads at most (gene workers, prefetch) and its
    interpreter lock serialises their host work; samples are independent, so a rank runs GK_PROCS_PER_GPU
    processes on its GPU (default 3, with 4 gene threads each), the same way a cohort run may place
    several ranks on one GPU.  Worker 0 is the rank's own process and keeps the clock: the timed region
    starts when every worker (and every rank) is ready and ends when every worker's last sample is typed."""
    from types import SimpleNamespace
    args = SimpleNamespace(**opts)
    if j and os.environ.get("GK_BENCH_KILL_WORKER") == str(j):   # test hook: this worker dies at once
        os._exit(3)
    from kir_graph_amd import _lib
    from kir_graph_amd.engine import DeviceIndex
    world = int(os.environ.get("WORLD_SIZE", "1"))
    n_dev = max(1, _lib.deviceCount())
    dev = _lib.Device(local_rank % n_dev if world > 1 else 0)
    sidx, gidx, sample, rec, table = build_inputs(seed=1031 + rank, n_pairs=args.pairs)
    gene_cn = sample.gene_cn
    dindex = DeviceIndex(dev, gidx)
    mates = dev.put(rec)
    dev.sync()

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
            gang[name].wait(timeout=300)

    n_valid = 0
    if args.warmup:
        n_valid = run_steps(args.warmup, dev, dindex, gidx, mates, table, gene_cn, args.method)[2]
    if j == 0 and getattr(args, "profile_host", False):
        import cProfile
        import pstats
        pr = cProfile.Profile()
        pr.enable()
        run_steps(1, dev, dindex, gidx, mates, table, gene_cn, args.method)
        pr.disable()
        pstats.Stats(pr, stream=sys.stderr).sort_stats("cumulative").print_stats(45)
    for d in all_devices():
        d.profEnable(True)
        d.profCollect()
        d.call_log = []
    dev.sync()
    gang_wait("ready")
    if rank_barrier is not None:
        rank_barrier()
    gang_wait("go")
    t0 = time.perf_counter()
    last = run_steps(claims(), dev, dindex, gidx, mates, table, gene_cn, args.method)
    if last is not None:
        n_valid = last[2]
    for d in all_devices():
        d.sync()
    gang_wait("done")
    if rank_barrier is not None:
        rank_barrier()
    if timing is not None:
        timing["elapsed"] = time.perf_counter() - t0
    prof, call_log = {}, []
    for d in all_devices():
        for k, (n, ms) in d.profCollect().items():
            n0, ms0 = prof.get(k, (0, 0.0))
            prof[k] = (n0 + n, ms0 + ms)
        call_log += d.call_log or []
        d.profEnable(False)
    if j:
        gang["results"].put({"prof": prof, "call_log": call_log})
        return None
    alone = None
    if procs > 1:   # the same launches with the GPU to this process alone (the other workers are done): untimed
        for d in all_devices():
            d.profEnable(True)
            d.profCollect()
        run_steps(2, dev, dindex, gidx, mates, table, gene_cn, args.method)
        alone = {}
        for d in all_devices():
            for k, (n, ms) in d.profCollect().items():
                n0, ms0 = alone.get(k, (0, 0.0))
                alone[k] = (n0 + n, ms0 + ms)
            d.profEnable(False)
    return {"prof": prof, "call_log": call_log, "n_valid": n_valid, "sidx": sidx, "gidx": gidx, "gene_cn": gene_cn,
            "alone": alone}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--pairs", type=int, default=1_000_000, help="read pairs per sample (config 2: 1e6)")
    ap.add_argument("--method", default="pv")
    ap.add_argument("--cpu-pairs", type=int, default=20000, help="pairs for the CPU baseline sample (0 = skip)")
    ap.add_argument("--verbose", action="store_true")
    ap.add_argument("--profile-host", action="store_true", help="cProfile one extra step to stderr")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # Worker processes of this rank on its GPU (see worker()): started first, before anything touches HIP.
    procs = max(1, int(os.environ.get("GK_PROCS_PER_GPU", "3")))
    procs = min(procs, max(1, args.steps))
    own_threads = procs > 1 and "GK_THREADS" not in os.environ
    if own_threads:
        os.environ["GK_THREADS"] = "4"   # gene threads per process: three processes share the host cores
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

    dist = None
    backend = os.environ.get("GK_BENCH_BACKEND", "nccl")   # "gloo": rehearsal of the N > 1 path on one GPU
    if world > 1:
        import torch
        import torch.distributed as dist_
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist_.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist_.init_process_group(backend)
        dist = dist_

    def rank_barrier():
        if dist is not None:
            import torch
            if backend == "nccl":
                torch.cuda.synchronize()
            dist.barrier()
            if backend == "nccl":
                torch.cuda.synchronize()

    timing = {}
    try:
        res = worker(0, procs, vars(args), rank, local_rank, gang, rank_barrier=rank_barrier, timing=timing)
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
        res = worker(0, 1, vars(args), rank, local_rank, None, rank_barrier=rank_barrier, timing=timing)
    elapsed = timing["elapsed"]
    prof, call_log, n_valid = res["prof"], res["call_log"], res["n_valid"]
    sidx, gidx, gene_cn = res["sidx"], res["gidx"], res["gene_cn"]
    for _ in range(procs - 1):
        other = gang["results"].get(timeout=600)
        for k, (n, ms) in other["prof"].items():
            n0, ms0 = prof.get(k, (0, 0.0))
            prof[k] = (n0 + n, ms0 + ms)
        call_log += other["call_log"]
    finished.set()
    if gang is not None:
        for h in helpers:
            h.join(timeout=30)

    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        reads_per_step = 2 * args.pairs * world
        value = reads_per_step / (elapsed / args.steps)
        total_kernel_ms = sum(v[1] for v in prof.values())
        dom = max(prof.items(), key=lambda kv: kv[1][1]) if prof else ("none", (1, 0.0))
        if args.verbose:
            for k, (n, ms) in sorted(prof.items(), key=lambda kv: -kv[1][1]):
                log(f"[bench] {k:18s} launches {n:6d}  total {ms:9.3f} ms  avg {ms / n:8.4f} ms")
            log(f"[bench] kernel time {total_kernel_ms / args.steps:.2f} ms of {ms_per_step:.2f} ms per step")
        roof = roofline(dom, call_log)
        # the kernel's share of the whole timed region: work of all its launches over the elapsed time (launches of
        # different workers overlap, so this is not bounded by the per-launch figure above)
        n_l, t_ms = dom[1]
        if elapsed > 0 and t_ms > 0 and roof.get("achieved") is not None:
            share = (t_ms * 1e-3) / elapsed          # sum of launch durations / wall time
            roof["region"] = {"achieved": roof["achieved"] * share, "unit": roof["unit"], "frac": roof["frac"] * share,
                              "valu_frac": roof["valu"]["frac"] * share if "valu" in roof else None,
                              "note": "algorithmic bytes (ops) of all launches of the kernel / elapsed time of the region"}
        if res.get("alone") and dom[0] in res["alone"] and roof.get("achieved") is not None:
            # launch durations in the timed region include the time the kernel shares the GPU with the kernels of
            # the other worker processes; the same launches right after it, one process on the GPU:
            n1, ms1 = res["alone"][dom[0]]
            scale = (dom[1][1] / dom[1][0]) / (ms1 / n1) if n1 and ms1 else None
            roof["one_process"] = {"avg_launch_ms": ms1 / n1, "launches": n1,
                                   "achieved": roof["achieved"] * scale if scale else None,
                                   "frac": roof["frac"] * scale if scale else None,
                                   "valu_frac": roof["valu"]["frac"] * scale if scale and "valu" in roof else None,
                                   "note": "2 untimed steps after the timed region, no other worker on the GPU"}
        out = {
            "metric": "typed 150 bp PE reads/s (pileup+EM) per GPU; achieved HBM GB/s vs roofline",
            "value": value, "unit": "reads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"configs[1]: 1 synthetic sample per GPU, {2 * args.pairs} 150 bp PE reads, "
                                   f"synthetic example_index-shaped index ({sum(len(t.alleles) for t in gidx.tables)} "
                                   f"alleles, 15 genes), --allele-strategy {args.method}, top_n 600",
                       "pairs_per_sample": args.pairs, "pairs_passing_filter": int(n_valid),
                       "parallelism": f"samples sharded over {world} GPU(s), no data-path collective; "
                                      f"{procs} worker process(es) per GPU, {os.environ.get('GK_THREADS', '6')} gene threads each"},
            "roofline": roof,
            "kernel_ms_per_step": {k: v[1] / args.steps for k, v in prof.items()},
        }
        if args.cpu_pairs and world == 1:      # the CPU leg runs on rank 0 of the single-GPU run only
            out["cpu_baseline"] = cpu_baseline(sidx, gidx, gene_cn, args.method, args.cpu_pairs, seed=99)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def measured_traffic(kernel):
    """HBM bytes per launch of ``kernel`` from the committed PMC passes over this same command
    (profiles/r01_bench_traffic.json, made by tools/collect_profiles.sh), or None."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_bench_traffic.json")
    try:
        with open(path) as f:
            t = json.load(f)
        return float(t["traffic_bytes_per_launch"]) if t.get("kernel") == kernel else None
    except (OSError, ValueError, KeyError):
        return None


def roofline(dom, call_log):
    """Roofline entry of the dominant kernel (algorithmic bytes stated in DESIGN.md)."""
    from kir_graph_amd import roofmodel
    name, (launches, total_ms) = dom
    out = roofmodel.summarise(call_log, name, total_ms, launches)
    traffic = measured_traffic(name)
    if traffic is not None:
        out["traffic"] = traffic
        out["traffic_source"] = "profiles/r01_bench_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command)"
    return out


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""configs[3] on ONE GPU: a synthetic cohort through `python -m kir_graph_amd.main --cn-cohort`, sharded over N ranks
(file backend: the ranks share the GPU, the pooled-depth exchange goes through the rendezvous directory -- on a node
with a GPU per rank the same calls go over RCCL), compared with the single-process run:

  * every per-sample TSV (`.cn`, allele, `.possible`), `cohort.cn.tsv` and `cohort.allele.tsv` byte for byte;
  * the CN column against the ORACLE's fit (oracle/cn.py) on the pooled depths of the depth files the run wrote;
  * a log line per run: samples/s, reads/s, peak host RSS over all ranks, peak HBM in use (rocm-smi, polled).

    python tests/run_cohort_cfg3.py [--samples 64] [--pairs 2500000] [--ranks 6] [--distinct 8] [--out DIR]

BASELINE.json configs[3] = 64 samples x 5 M reads (2.5 M pairs) over 8 GPUs.  The GPU boxes of this pool allow at most
6 processes on a card, hence `--ranks 6` by default here (8 ranks need 8 GPUs or a box without that guard).  Only
`--distinct` samples are synthesised (seeds 100 + k, their own copy numbers); the others are hard links to them under
their own names -- the pipeline does the same work for a copy, and the pooled fit sees every sample."""
import argparse
import io
import json
import os
import resource
import shutil
import subprocess
import sys
import tempfile
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def make_inputs(tmp, n_samples, n_pairs, n_distinct):
    from kir_graph_amd import packed, synth
    folder = os.path.join(tmp, "index")
    os.makedirs(folder, exist_ok=True)
    prefix = os.path.join(folder, "kir_2100_withexon_ab_2dl1s1.leftalign.mut01")
    sidx = synth.makeIndex(seed=2022)
    sidx.write(prefix)
    header = ["@HD\tVN:1.0\tSO:coordinate"] + [f"@SQ\tSN:{g}\tLN:{len(sidx.backbone[g])}" for g in sidx.genes]
    bams = [os.path.join(tmp, f"s{k:02d}.bam") for k in range(n_samples)]
    t = time.time()
    for k in range(n_samples):
        if k >= n_distinct:
            os.link(bams[k % n_distinct], bams[k])
            continue
        s = synth.makeSample(sidx, seed=100 + k, n_pairs=n_pairs)
        packed.writeBam(bams[k], "\n".join(header + synth.toSamLines(s)) + "\n")
        print(f"[cfg3] sample {k}: {os.path.getsize(bams[k]) / 1e6:.0f} MB ({time.time() - t:.0f}s)", file=sys.stderr, flush=True)
    return folder, bams


class HbmWatch(threading.Thread):
    """Peak VRAM in use, polled through rocm-smi (the ranks are other processes: their pools are not visible from here)."""

    def __init__(self):
        super().__init__(daemon=True)
        self.peak, self.stop = 0, threading.Event()

    def run(self):
        while not self.stop.wait(0.5):
            try:
                out = subprocess.run(["rocm-smi", "--showmeminfo", "vram", "--json"], capture_output=True, text=True,
                                     timeout=5).stdout
                for card in json.loads(out).values():
                    self.peak = max(self.peak, int(card.get("VRAM Total Used Memory (B)", 0)))
            except Exception:      # noqa: BLE001 -- no rocm-smi: the figure stays 0
                pass


def run_cli(folder, out, bams, ranks):
    cmd = [sys.executable, "-m", "kir_graph_amd.main", "--step-skip-extraction", "--index-folder", folder,
           "--output-folder", out, "--allele-strategy", "pv", "--no-variant-json", "--cn-cohort", "--log-level", "WARNING"]
    for b in bams:
        cmd += ["--alignment", b]
    if ranks > 1:
        cmd += ["--ranks", str(ranks)]
    # no hand-off files unless a sample has to leave HBM before the pooled fit (64 x 660 MB of .npz that nothing reads)
    env = dict(os.environ, PYTHONPATH=ROOT, GK_COMM_BACKEND="file", GK_HANDOFF="lazy")
    watch = HbmWatch()
    watch.start()
    before = resource.getrusage(resource.RUSAGE_CHILDREN)
    t = time.time()
    res = subprocess.run(cmd, env=env, cwd=ROOT)
    wall = time.time() - t
    watch.stop.set()
    after = resource.getrusage(resource.RUSAGE_CHILDREN)
    if res.returncode:
        raise SystemExit(f"[cfg3] the {ranks}-rank run failed with code {res.returncode}")
    return {"ranks": ranks, "wall_s": wall, "cpu_s": (after.ru_utime + after.ru_stime) - (before.ru_utime + before.ru_stime),
            "peak_rss_mb_largest_child": after.ru_maxrss / 1024, "peak_hbm_gb": watch.peak / 2**30}


def compare(one, many):
    names = sorted(os.listdir(one))
    assert names == sorted(os.listdir(many)), "the runs wrote different files"
    checked = 0
    for n in names:
        if not n.endswith(".tsv") or n.endswith(".depth.tsv"):
            continue
        a = open(os.path.join(one, n)).read().replace(one, "@")
        b = open(os.path.join(many, n)).read().replace(many, "@")
        assert a == b, f"{n} differs between the runs"
        checked += 1
    return checked


def oracle_check(out, n_samples):
    import pandas as pd
    from oracle import cn as ocn
    depth_files = sorted(os.path.join(out, f) for f in os.listdir(out) if f.endswith(".no_multi.depth.tsv"))
    assert len(depth_files) == n_samples
    tables = [pd.read_csv(f, sep="\t", header=None, names=["gene", "pos", "depth"]) for f in depth_files]
    want = ocn.predictCN(tables, "p75", "LCND", {"base_dev": 0.08, "start_base": 2}, False)[0]
    merged = pd.read_csv(os.path.join(out, "cohort.cn.tsv"), sep="\t", index_col=0)
    for f, w in zip(depth_files, want):
        cn_file = f[:-len(".tsv")] + ".p75.cohort.LCND.tsv"
        got = pd.read_csv(cn_file, sep="\t")
        assert dict(zip(got["gene"], got["cn"])) == {g: int(c) for g, c in w.items()}, cn_file
        assert {g: int(merged[cn_file][g]) for g in merged.index} == {g: int(c) for g, c in w.items()}
    return len(depth_files)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", type=int, default=64)
    ap.add_argument("--pairs", type=int, default=2_500_000)
    ap.add_argument("--ranks", type=int, default=6)
    ap.add_argument("--distinct", type=int, default=8)
    ap.add_argument("--out", default=None, help="keep inputs and outputs here (default: a temporary directory)")
    args = ap.parse_args()
    tmp = args.out or tempfile.mkdtemp(prefix="gk_cfg3_")
    os.makedirs(tmp, exist_ok=True)
    try:
        folder, bams = make_inputs(tmp, args.samples, args.pairs, min(args.distinct, args.samples))
        runs = []
        for ranks in (1, args.ranks):
            out = os.path.join(tmp, f"out_{ranks}")
            r = run_cli(folder, out, bams, ranks)
            r.update(samples=args.samples, reads_per_sample=2 * args.pairs,
                     samples_per_s=args.samples / r["wall_s"], reads_per_s=2 * args.pairs * args.samples / r["wall_s"],
                     cpu_s_per_sample=r["cpu_s"] / args.samples)
            runs.append(r)
            print("[cfg3] " + json.dumps(r), flush=True)
        n_files = compare(os.path.join(tmp, "out_1"), os.path.join(tmp, f"out_{args.ranks}"))
        n_cn = oracle_check(os.path.join(tmp, "out_1"), args.samples)
        print(f"[cfg3] {n_files} TSV files byte-identical between 1 and {args.ranks} ranks "
              f"(cohort.cn.tsv, cohort.allele.tsv, per-sample cn / allele / possible); copy numbers of {n_cn} samples equal "
              f"the oracle's fit on the pooled depths")
    finally:
        if not args.out:
            shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()

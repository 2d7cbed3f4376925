#!/usr/bin/env python3
"""Dev tool: wall time of the command line on a synthetic cohort of coordinate-sorted BAM files
(native BAM ingest -> tabulation -> depth / CN -> typing -> cohort tables), per sample."""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kir_graph_amd import main as cli, packed, synth

n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
n_samples = int(sys.argv[2]) if len(sys.argv) > 2 else 3
extra = sys.argv[3:]
n_made = min(n_samples, int(os.environ.get("GK_CLI_DISTINCT", "3")))   # distinct samples synthesised; the rest are copies
# under torchrun every rank runs this script: rank 0 writes the inputs, the others wait for them
rank = int(os.environ.get("RANK", "0"))
tmp = os.path.join(tempfile.gettempdir(), f"gk_cli_{n_pairs}_{n_samples}_{os.environ.get('MASTER_PORT', os.getppid())}")
folder = os.path.join(tmp, "index")
prefix = os.path.join(folder, "kir_2100_withexon_ab_2dl1s1.leftalign.mut01")
bams = [os.path.join(tmp, f"s{k}.bam") for k in range(n_samples)]
cns = [os.path.join(tmp, f"s{k}.cn.tsv") for k in range(n_samples)]
if rank == 0 and not os.path.exists(os.path.join(tmp, "ready")):   # a second run from the same shell reuses the inputs
    os.makedirs(folder, exist_ok=True)
    sidx = synth.makeIndex(seed=2022)
    sidx.write(prefix)
    header = ["@HD\tVN:1.0\tSO:coordinate"] + [f"@SQ\tSN:{g}\tLN:{len(sidx.backbone[g])}" for g in sidx.genes]
    t = time.time()
    import shutil
    for k in range(n_samples):
        if k >= n_made:       # a copy of an earlier sample under its own name (same work for the pipeline)
            shutil.copy(bams[k % n_made], bams[k])
            shutil.copy(cns[k % n_made], cns[k])
            continue
        s = synth.makeSample(sidx, seed=100 + k, n_pairs=n_pairs)
        lines = synth.toSamLines(s)
        packed.writeBam(bams[k], "\n".join(header + lines) + "\n")
        with open(cns[k], "w") as f:
            f.write("gene\tcn\n" + "".join(f"{g}\t{c}\n" for g, c in s.gene_cn.items()))
    print(f"{n_samples} samples of {n_pairs} pairs written in {time.time() - t:.0f}s "
          f"({os.path.getsize(bams[0]) / 1e6:.0f} MB each)", file=sys.stderr)
    open(os.path.join(tmp, "ready"), "w").close()
while not os.path.exists(os.path.join(tmp, "ready")):
    time.sleep(0.2)
out = os.path.join(tmp, "out")
argv = ["--step-skip-extraction", "--index-folder", folder, "--output-folder", out, "--allele-strategy", "pv"]
for b in bams:
    argv += ["--alignment", b]
argv += ["--cn-provided"] + cns
args = cli.createParser().parse_args(argv + extra)
import resource


def cpu_now():      # user, system seconds of this process (its ingest / typing threads included)
    ru = resource.getrusage(resource.RUSAGE_SELF)
    return ru.ru_utime, ru.ru_stime


cpu0 = cpu_now()    # whatever writing the inputs cost is before this point
t = time.time()
if os.environ.get("GK_CLI_PROFILE") == "1":     # where does the main thread spend its time?
    import cProfile, pstats
    pr = cProfile.Profile()
    pr.enable()
    cli.main(args)
    pr.disable()
    pstats.Stats(pr, stream=sys.stderr).sort_stats("cumulative").print_stats(35)
else:
    cli.main(args)
dt = time.time() - t
cpu1 = cpu_now()
user, system = cpu1[0] - cpu0[0], cpu1[1] - cpu0[1]
print(f"process CPU of the command line alone: {user:.1f}s user + {system:.1f}s system over {dt:.1f}s of wall = {(user + system) / dt:.1f} cores busy, "
      f"{(user + system) / n_samples:.2f} core-s per sample of {2 * n_pairs} reads (ingest + typing + outputs; typing alone is "
      f"~0.025 core-s per 2 M reads, bench.py host.host_core_s_per_step)", file=sys.stderr)
if rank == 0:
  print(f"command line: {dt:.2f}s for {n_samples} samples = {dt / n_samples:.2f}s per sample of {2 * n_pairs} reads "
      f"({2 * n_pairs * n_samples / dt / 1e6:.2f} M reads/s end to end; flags {extra})")

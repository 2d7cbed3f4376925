#!/usr/bin/env python3
"""Dev tool: the native BAM ingest alone (coordinate-sorted BAM -> packed records, no GPU work) on a synthetic sample:
wall time and process CPU time of `packed.packBam`, and the phases the library reports (GK_TRACE=ingest), with one
thread and with the default thread count.     python tools/ingest_profile.py [pairs, default 500000] [repeats, default 4]
"""
import os
import resource
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(path, repeats):
    from kir_graph_amd import packed, synth
    from kir_graph_amd.index import GkIndex
    sidx = synth.makeIndex(seed=2022)
    gidx = GkIndex.fromVariants(sidx.variants, genes=sidx.genes, exons=sidx.exons)

    def cpu():
        ru = resource.getrusage(resource.RUSAGE_SELF)
        return ru.ru_utime, ru.ru_stime
    for rep in range(repeats):
        u0, s0 = cpu()
        t0 = time.time()
        rec = packed.packBam(path, gidx)[0]
        u1, s1 = cpu()
        print(f"[run {rep}] {1e3 * (time.time() - t0):.0f} ms wall, {u1 - u0:.2f} s user + {s1 - s0:.2f} s system for {len(rec)} records",
              file=sys.stderr, flush=True)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        return child(sys.argv[2], int(sys.argv[3]))
    n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 500000
    repeats = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    path = f"/tmp/gk_ingest_{n_pairs}.bam"
    if not os.path.exists(path):
        from kir_graph_amd import packed, synth
        t = time.time()
        sidx = synth.makeIndex(seed=2022)
        s = synth.makeSample(sidx, seed=100, n_pairs=n_pairs)
        header = ["@HD\tVN:1.0\tSO:coordinate"] + [f"@SQ\tSN:{g}\tLN:{len(sidx.backbone[g])}" for g in sidx.genes]
        packed.writeBam(path, "\n".join(header + synth.toSamLines(s)) + "\n")
        print(f"wrote {path} in {time.time() - t:.0f}s, {os.path.getsize(path) / 1e6:.0f} MB")
    for threads in ("1", None):
        env = dict(os.environ, GK_TRACE="ingest")
        if threads:
            env["GK_PACK_THREADS"] = threads
        print(f"== GK_PACK_THREADS={threads or 'default'}")
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", path, str(repeats)], env=env,
                             capture_output=True, text=True).stderr.splitlines()
        runs = [x for x in out if x.startswith("[run")]
        last = [x for x in out if x.startswith("[ingest]")][-6:]       # phases of the last run
        print("\n".join(runs + last))


if __name__ == "__main__":
    main()

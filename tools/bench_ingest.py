#!/usr/bin/env python3
"""Dev tool: host ingest rates (no GPU): SAM text -> records, BAM -> text -> records."""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from kir_graph_amd import synth, packed
from kir_graph_amd.index import GkIndex
from bamwriter import samToBam

n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
sidx = synth.makeIndex(seed=5, n_genes=5, var_range=(500, 800), allele_range=(20, 40))
gidx = GkIndex.fromVariants(sidx.variants, genes=sidx.genes, exons=sidx.exons)
sample = synth.makeSample(sidx, seed=9, n_pairs=n_pairs)
lines = synth.toSamLines(sample)
text = ("\n".join(lines) + "\n").encode()
header = ["@HD\tVN:1.0\tSO:coordinate"] + [f"@SQ\tSN:{g}\tLN:{len(sidx.backbone[g])}" for g in sidx.genes]
tmp = tempfile.mkdtemp()
bam = os.path.join(tmp, "s.bam")
samToBam(header + sorted(lines, key=lambda l: (l.split("\t")[2], int(l.split("\t")[3]))), bam)
print(f"{len(lines)} lines, {len(text) / 1e6:.0f} MB of SAM text, BAM {os.path.getsize(bam) / 1e6:.0f} MB")
t = time.time(); rec, *_ = packed.packText([text], gidx); dt = time.time() - t
print(f"SAM text -> records      {len(lines) / dt / 1e6:.2f} M lines/s")
t = time.time(); chunks = list(packed.bamChunks(bam)); dt1 = time.time() - t
print(f"BAM -> collated text     {len(lines) / dt1 / 1e6:.2f} M lines/s")
t = time.time(); rec2, *_ = packed.packText(chunks, gidx); dt2 = time.time() - t
print(f"BAM -> records (total)   {len(lines) / (dt1 + dt2) / 1e6:.2f} M lines/s")
assert rec.tobytes() == rec2.tobytes()
t = time.time(); rec3, *_ = packed.packBam(bam, gidx); dt3 = time.time() - t
print(f"BAM -> records (binary)  {len(lines) / dt3 / 1e6:.2f} M lines/s")
assert rec.tobytes() == rec3.tobytes()

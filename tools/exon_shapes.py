"""Dev tool: what exon-first looks like on the bench samples -- per gene the exon groups, the exon sets that reach the
threshold (candidate searches), and where the host time of a sample goes.   python tools/exon_shapes.py [pairs] [samples]"""
import cProfile, pstats, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from kir_graph_amd import _lib, cohort
from kir_graph_amd.engine import DeviceIndex, Tabulation
from kir_graph_amd.hisat2 import SampleData
from kir_graph_amd.kir_typing import selectKirTypingModel

pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 3
cohort.pipelineDefaults()
sidx, gidx, by_gene = bench.build_index()
dev = _lib.Device(0)
dindex = DeviceIndex(dev, gidx)
for k in range(n):
    sample, rec, table = bench.build_sample(sidx, gidx, by_gene, 1031 + k, pairs)
    tab = Tabulation(dindex, rec)
    data = SampleData(tab, gidx, None, ins_strings=table.strings)
    for rep in range(2):
        typer = selectKirTypingModel("exonfirst_1", data, top_n=600, variant_correction=True)
        pr = cProfile.Profile()
        t0 = time.perf_counter()
        pr.enable()
        typer.typing(sample.gene_cn)
        pr.disable()
        ms = 1e3 * (time.perf_counter() - t0)
    print(f"sample {k}: {ms:.1f} ms (second typing);", {g.split('*')[0]: (v['exon_groups'], v['exon_sets'], v['candidates'])
                                                        for g, v in getattr(typer, 'exon_info', {}).items()}, flush=True)
    if k == n - 1:
        pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
    tab.close()

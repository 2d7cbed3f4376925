#!/bin/bash
# L2-level read requests of the exact set sums, leaf-wise (setsum_leaves) against by tiles (fraction_chunks, GK_SETSUM=tiles):
# rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCC_REQ_sum over the bench's one-process serial mode, per-step totals.
#   bash tools/l2_reads_setsum.sh   (on the GPU box; prints the table)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/l2reads
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export GK_PROCS_PER_GPU=1 GK_THREADS=1 GK_PREFETCH=0 GK_SAMPLE_LANES=1 GK_SAMPLE_STREAMS=1
for form in leaves tiles; do
  rm -rf $O/$form
  GK_SETSUM=$form rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCC_REQ_sum --kernel-trace -d $O/$form -o p --output-format csv -- python3 $R/bench.py --cpu-pairs 0 --steps 8 --warmup 3 --serial-steps 0 --inputs hbm --one-kind --legs 1 --cli-samples 0 > /dev/null 2> $O/$form.err
  python3 - <<PY
import collections, csv
acc, n = collections.defaultdict(float), collections.Counter()
samples = 11
for r in csv.DictReader(open("$O/$form/p_counter_collection.csv")):
    name = r["Kernel_Name"]
    for k in ("setsum_leaves", "fraction_chunks", "fold_leaves", "combine_chunks"):
        if k in name:
            acc[(k, r["Counter_Name"])] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
print("GK_SETSUM=$form")
for key in sorted(acc):
    print("  %-16s %-22s %12.4e per launch  x %6.1f launches per sample = %12.4e per sample" % (key[0], key[1], acc[key] / n[key], n[key] / samples, acc[key] / samples))
PY
  rm -rf $O/$form
done

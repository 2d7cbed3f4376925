# Dev tool: SQ / scalar-cache counters of the compatibility kernel over tools/bench_compat.py (separate --pmc passes).
#   bash tools/pmc_compat.sh [output tag under gpurun_out/, default compat_pmc]
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-compat_pmc}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM --kernel-trace -d $O/sq1 -o p --output-format csv -- python3 $R/tools/bench_compat.py > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVES SQ_INSTS_SMEM --kernel-trace -d $O/sq2 -o p --output-format csv -- python3 $R/tools/bench_compat.py > /dev/null 2>&1
rocprofv3 --pmc SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_TC_DATA_READ_REQ SQ_INST_LEVEL_SMEM SQ_WAIT_INST_LDS --kernel-trace -d $O/sq3 -o p --output-format csv -- python3 $R/tools/bench_compat.py > $O/sq3.log 2>&1 || echo "sq3 pass failed (counter names?)"
python3 - <<PY
import csv, collections, glob
for d in ("sq1", "sq2", "sq3"):
    fs = glob.glob("$O/%s/**/p_counter_collection.csv" % d, recursive=True)
    if not fs: print(d, "no output"); continue
    acc = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open(fs[0])):
        if "compat_kernel" in r["Kernel_Name"] and "Li4E" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    for k in sorted(acc): print(f"{d} {k:24s} {acc[k]/n[k]:.4e}  ({n[k]} dispatches)")
PY

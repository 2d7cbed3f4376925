# Dev tool: SQ counters of the compatibility kernel over tools/bench_compat.py (two --pmc passes).
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/compat_pmc
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM --kernel-trace -d $O/sq1 -o p --output-format csv -- python3 $R/tools/bench_compat.py > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVES SQ_INSTS_SMEM --kernel-trace -d $O/sq2 -o p --output-format csv -- python3 $R/tools/bench_compat.py > /dev/null 2>&1
python3 - <<PY
import csv, collections, glob
for d in ("sq1", "sq2"):
    f = glob.glob("$O/%s/**/p_counter_collection.csv" % d, recursive=True)[0]
    acc = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        if "compat_kernel" in r["Kernel_Name"] and "Li4E" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    for k in sorted(acc): print(f"{d} {k:24s} {acc[k]/n[k]:.4e}  ({n[k]} dispatches)")
PY

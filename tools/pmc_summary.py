#!/usr/bin/env python3
"""Dev tool: mean counter values of one kernel over the dispatches of the rocprofv3 --pmc passes collected
by tools/collect_profiles.sh (pmc_sq1, pmc_sq2, pmc_fetch, pmc_write, cal_fetch, cal_write).

    python tools/pmc_summary.py gpurun_out/r01 maxsum_chunks > profiles/rNN_maxsum_pmc_counters.txt
"""
import collections
import csv
import os
import sys

root, kernel = sys.argv[1], sys.argv[2]
for d in ("pmc_sq1", "pmc_sq2", "bench_fetch", "bench_write", "pmc_fetch", "pmc_write", "cal_fetch", "cal_write"):
    path = os.path.join(root, d, "p_counter_collection.csv")
    if not os.path.exists(path):
        continue
    want = "" if d.startswith("cal") else kernel
    acc, n, dur = collections.defaultdict(float), collections.Counter(), collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"]
        if want and want not in name:
            continue
        if not want and "rocclr" in name:
            continue
        key = (name.split("(")[0].split("::")[-1] if not want else kernel, r["Counter_Name"])
        acc[key] += float(r["Counter_Value"])
        n[key] += 1
        dur[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    print(f"[{d}]")
    for key in sorted(acc):
        print(f"{key[0]:24s} {key[1]:24s} {acc[key] / n[key]:14.4e}   mean of {n[key]} dispatches, kernel {sum(dur[key]) / len(dur[key]):9.1f} us")
    print()

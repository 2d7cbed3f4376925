"""Durations of every launch of one kernel, in launch order, from a rocprofv3 --kernel-trace csv (tools/collect_profiles.sh
keeps the summaries only).   python tools/launch_durations.py <kernel_trace.csv> <kernel name prefix> [launches per step]"""
import csv
import sys

path, name = sys.argv[1], sys.argv[2]
per_step = int(sys.argv[3]) if len(sys.argv) > 3 else 0
rows = []
with open(path) as fh:
    for r in csv.DictReader(fh):
        if r["Kernel_Name"].startswith(name) or (" " + name) in r["Kernel_Name"] or ("::" + name) in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
                         r.get("Grid_Size_X", r.get("Grid_Size", "?")), r["Kernel_Name"][:60]))
rows.sort()
if per_step:
    rows = rows[-per_step:]
for i, (_, us, grid, kn) in enumerate(rows):
    print(f"{i:4d} {us:10.1f} us  grid {grid:>10s}  {kn}")
print(f"{len(rows)} launches, {sum(r[1] for r in rows) / 1e3:.3f} ms")

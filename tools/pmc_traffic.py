#!/usr/bin/env python3
"""HBM traffic per launch of one kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE).

gfx950 corrections (MI355X_MICROARCH.md, HBM section; calibrated with tools/fetch_calib.hip): both
counters are KiB, FETCH_SIZE reports half of the bytes of coalesced reads, WRITE_SIZE is exact."""
import csv, json, os, sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kir_graph_amd.build import sourceDigest

def total(path, counter, kernel):
    s = n = 0
    for r in csv.DictReader(open(path)):
        if kernel in r["Kernel_Name"] and r["Counter_Name"] == counter:
            s += float(r["Counter_Value"]); n += 1
    return s, n

fetch, n_f = total(sys.argv[1], "FETCH_SIZE", sys.argv[3])
write, n_w = total(sys.argv[2], "WRITE_SIZE", sys.argv[3])
workload = sys.argv[4] if len(sys.argv) > 4 else None
print(json.dumps({"kernel": sys.argv[3], "workload": workload, "csrc_sha16": sourceDigest(sys.argv[3]), "launches": n_f, "fetch_size_kib": fetch, "write_size_kib": write,
                  "traffic_bytes_per_launch": (2 * fetch / max(n_f, 1) + write / max(n_w, 1)) * 1024,
                  "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over the workload's serial form "
                            "(tools/collect_profiles.sh); "
                            "bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024"}))

"""Dev tool: the [trace] lines of a bench run (GK_TRACE=bench) as a table relative to the first event, one line per
stage of a sample: which thread, from when to when (ms).   python3 tools/trace_dump.py bench.err [last N lines]"""
import re
import sys

rows = []
for line in open(sys.argv[1], errors="replace"):
    m = re.match(r"\[trace\] (copy|stage|type) (\d+) (\d+) ([\d.]+) ([\d.]+)", line)
    if m:
        rows.append((float(m.group(4)), float(m.group(5)), m.group(1), int(m.group(2)), m.group(3)))
rows.sort()
n = int(sys.argv[2]) if len(sys.argv) > 2 else len(rows)
rows = rows[-n:]
t0 = rows[0][0]
for a, b, what, k, thread in rows:
    print(f"{what:5s} sample {k:3d} thread {thread:>7s}  {1e3 * (a - t0):8.2f} -> {1e3 * (b - t0):8.2f}  ({1e3 * (b - a):6.2f} ms)")

"""Timeline of the host threads of ONE bench worker process (GK_TRACE=bench,search, stderr of bench.py):
per sample the staging (copy + tabulation), the Python before the search call, the call, the Python after it; and how
the searches of the lanes overlap.     python3 tools/host_timeline.py bench.err [samples to skip at the start]
"""
import re
import sys


def main():
    path = sys.argv[1]
    skip = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    stage, typ, search, pre = [], [], [], []
    for line in open(path, errors="replace"):
        m = re.match(r"\[trace\] pre (\d+) ([\d.]+) ([\d.]+) ([\d.]+)", line)
        if m:
            pre.append((m.group(1), float(m.group(2)), float(m.group(3)), float(m.group(4))))
            continue
        m = re.match(r"\[trace\] (\w+) (\d+) (\S+) ([\d.]+) ([\d.]+)", line)
        if m:
            (stage if m.group(1) == "stage" else typ).append((int(m.group(2)), m.group(3), float(m.group(4)), float(m.group(5))))
            continue
        m = re.search(r"gk_sample_search (\S+)\] thread (\d+), .* host ([\d.]+) ms, waits ([\d.]+) ms in (\d+) waits; from ([\d.]+) to ([\d.]+) s", line)
        if m:
            search.append((m.group(1), float(m.group(3)), float(m.group(4)), float(m.group(6)), float(m.group(7)), m.group(2)))
    typ.sort(key=lambda r: r[2])
    typ, stage = typ[skip:], sorted(stage, key=lambda r: r[2])[skip:]
    rows = []
    for k, thread, t0, t1 in typ:
        inside = [s for s in search if s[3] >= t0 and s[4] <= t1 and s[5] == thread]
        if len(inside) != 1:
            continue
        s = inside[0]
        rows.append((k, thread, t0, s[3], s[4], t1, s[1], s[2]))
    n = len(rows)
    if not n:
        print("no samples matched")
        return
    ms = lambda x: 1e3 * x
    span = rows[-1][5] - rows[0][2]
    print(f"{n} samples in {ms(span):.1f} ms = {ms(span) / n:.3f} ms per sample; lanes: {sorted(set(r[1] for r in rows))}")
    print(f"per sample, mean ms: before the call {sum(ms(r[3] - r[2]) for r in rows) / n:.2f}, the call {sum(ms(r[4] - r[3]) for r in rows) / n:.2f} "
          f"(host inside {sum(r[6] for r in rows) / n:.2f}, waits {sum(r[7] for r in rows) / n:.2f}), after the call {sum(ms(r[5] - r[4]) for r in rows) / n:.2f}")
    mine = [p for p in pre if any(p[0] == r[1] and r[2] <= p[1] and p[3] <= r[3] + 1e-4 for r in rows)]
    if mine:
        first = {(r[1], i): r[2] for i, r in enumerate(rows)}
        print(f"before the call, mean ms: SampleData / typer objects {sum(ms(p[1] - max(r[2] for r in rows if r[1] == p[0] and r[2] <= p[1])) for p in mine) / len(mine):.2f}, "
              f"prepared() {sum(ms(p[2] - p[1]) for p in mine) / len(mine):.2f}, gene objects + jobs {sum(ms(p[3] - p[2]) for p in mine) / len(mine):.2f}")
    if stage:
        print(f"staging (copy + tabulation) per sample: {sum(ms(r[3] - r[2]) for r in stage) / len(stage):.2f} ms")
    # how many searches are in flight over the window
    events = []
    for r in rows:
        events += [(r[3], 1), (r[4], -1)]
    events.sort()
    level, last, hist = 0, rows[0][2], {}
    for t, d in events:
        hist[level] = hist.get(level, 0.0) + (t - last)
        level, last = level + d, t
    tot = sum(hist.values())
    print("search calls in flight (share of the window): " + ", ".join(f"{k}: {100 * v / tot:.0f} %" for k, v in sorted(hist.items())))
    # idle between a lane's samples
    by_lane = {}
    for r in rows:
        by_lane.setdefault(r[1], []).append(r)
    for lane, rs in sorted(by_lane.items()):
        gaps = [ms(b[2] - a[5]) for a, b in zip(rs, rs[1:])]
        if gaps:
            print(f"{lane}: {len(rs)} samples, mean gap between its samples {sum(gaps) / len(gaps):.2f} ms (max {max(gaps):.2f})")


if __name__ == "__main__":
    main()

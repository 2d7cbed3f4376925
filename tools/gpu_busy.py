"""How busy is the GPU while bench.py runs in ONE process?  Reads a rocprofv3 kernel trace (``*_kernel_trace.csv``) and
reports, over the samples in the middle of the run (from the sample at the first fraction to the one at the last): wall span, the union of the kernel
intervals (GPU busy), their sum (overlap = sum / union) and the longest idle gaps.

    rocprofv3 --kernel-trace -d gpurun_out/r03/busy -o p --output-format csv -- python3 bench.py ...
    python3 tools/gpu_busy.py gpurun_out/r03/busy/p_kernel_trace.csv [first fraction, default 0.3] [last, default 0.8]
"""
import csv
import sys
from collections import defaultdict


def short(name):
    name = name.replace("void ", "").replace("(anonymous namespace)::", "")
    return name.split("<")[0].split("(")[0].strip()


def main():
    path = sys.argv[1]
    lo = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
    hi = float(sys.argv[3]) if len(sys.argv) > 3 else 0.8
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    rows = [(s, e, short(n)) for s, e, n in rows]
    tabs = [s for s, _, n in rows if n == "tab_count"]          # one per sample: the window is cut in samples
    a, b = tabs[int(lo * len(tabs))], tabs[int(hi * len(tabs))]
    sel = [r for r in rows if r[0] >= a and r[0] < b]
    span = sel[-1][1] - sel[0][0]
    union, total, end = 0, 0, sel[0][0]
    gaps = []
    by_kernel = defaultdict(lambda: [0, 0])
    for s, e, name in sel:
        total += e - s
        k = name
        by_kernel[k][0] += 1
        by_kernel[k][1] += e - s
        if s > end:
            gaps.append((s - end, k))
            union += e - s
        elif e > end:
            union += e - end
        end = max(end, e)
    n_tab = by_kernel.get("tab_count", [1, 0])[0]
    print(f"window {span / 1e6:.1f} ms, {len(sel)} launches, {n_tab} samples -> {span / 1e6 / n_tab:.3f} ms per sample")
    print(f"GPU busy (union of kernels) {union / 1e6:.1f} ms = {100 * union / span:.1f} %  ({union / 1e6 / n_tab:.3f} ms per sample)")
    print(f"sum of kernel durations {total / 1e6:.1f} ms = {total / 1e6 / n_tab:.3f} ms per sample; overlap factor {total / union:.2f}")
    idle = span - union
    print(f"idle {idle / 1e6:.1f} ms = {idle / 1e6 / n_tab:.3f} ms per sample in {len(gaps)} gaps; "
          f"gaps > 20 us: {sum(1 for g, _ in gaps if g > 20000)} holding {sum(g for g, _ in gaps if g > 20000) / 1e6 / n_tab:.3f} ms per sample")
    after = defaultdict(lambda: [0, 0])
    for g, k in gaps:
        after[k][0] += 1
        after[k][1] += g
    print("idle time by the kernel that ended the gap (ms per sample):")
    for k, (n, g) in sorted(after.items(), key=lambda kv: -kv[1][1])[:8]:
        print(f"  {k:28s} {n / n_tab:6.1f} gaps  {g / 1e6 / n_tab:.3f} ms")
    print("kernel time per sample (ms), concurrent run:")
    for k, (n, t) in sorted(by_kernel.items(), key=lambda kv: -kv[1][1])[:10]:
        print(f"  {k:28s} {n / n_tab:6.1f} launches  {t / 1e6 / n_tab:.3f} ms")


if __name__ == "__main__":
    main()

#!/bin/bash
# Copies the summaries that tools/collect_profiles.sh left under gpurun_out/<tag> (merged back by gpurun) into
# profiles/ under the names DESIGN.md and profiles/README.md cite.   bash tools/install_profiles.sh [tag, default r05]
set -e
TAG=${1:-r05}
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out/$TAG
[ -f $O/bench.json ] && cp $O/bench.json $R/profiles/${TAG}_bench.json
for W in $(ls -d $O/*/ 2>/dev/null); do
  W=$(basename $W)
  [ -f $O/$W/kernel_stats.csv ] || continue
  cp $O/$W/kernel_stats.csv $R/profiles/${TAG}_${W}_serial_kernel_stats.csv
  cp $O/$W/bench_under_rocprof.json $R/profiles/${TAG}_${W}_serial_under_rocprof.json
  for f in $O/$W/traffic_*.json; do k=$(basename $f .json); cp $f $R/profiles/${TAG}_traffic_${W}_${k#traffic_}.json; done
  for f in $O/$W/pmc_*.txt; do k=$(basename $f .txt); cp $f $R/profiles/${TAG}_pmc_${W}_${k#pmc_}.txt; done
done
python3 - <<PY
import json, os
p = "$R/profiles/${TAG}_bench.json"
if os.path.exists(p):
    d = json.load(open(p))
    for name, o in [("headline", d)] + [(k, d[k]) for k in ("em", "configs1_pv") if k in d]:
        r = o["roofline"]
        print(name, o["config"]["workload"][:40], "ms_per_step", round(o["ms_per_step"], 3), "value", round(o["value"] / 1e6, 1), "M reads/s;",
              r["kernel"], r["bound"], "frac", r["frac"] and round(r["frac"], 3), "traffic", r.get("traffic"),
              "; serial kernel sum", round(o["kernels_serial"]["kernel_ms_per_step"], 3), "ms")
PY

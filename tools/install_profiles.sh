#!/bin/bash
# Copies the summaries that tools/collect_profiles.sh left under gpurun_out/<tag> (merged back by gpurun) into
# profiles/ under the names DESIGN.md and profiles/README.md cite.   bash tools/install_profiles.sh [tag, default r03]
set -e
TAG=${1:-r04}
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out/$TAG
cp $O/bench.json $R/profiles/${TAG}_bench.json
cp $O/stats/b_kernel_stats.csv $R/profiles/${TAG}_bench_serial_kernel_stats.csv
cp $O/bench_under_rocprof.json $R/profiles/${TAG}_bench_serial_under_rocprof.json
cp $O/traffic_compat_kernel.json $R/profiles/${TAG}_bench_traffic.json
for k in compat_kernel tab_count minsum_sad setsum_leaves fraction_chunks maxsum_chunks select_cut count_ids patch_pending; do
  cp $O/pmc_$k.txt $R/profiles/${TAG}_pmc_$k.txt
  cp $O/traffic_$k.json $R/profiles/${TAG}_traffic_$k.json
done
python3 - <<PY
import json
d = json.load(open("$R/profiles/${TAG}_bench.json"))
print("ms_per_step", round(d["ms_per_step"], 3), "value", round(d["value"] / 1e6, 1), "M reads/s;",
      d["roofline"]["kernel"], d["roofline"]["bound"], "frac", round(d["roofline"]["frac"], 3),
      "; serial kernel sum", round(d["kernels_serial"]["kernel_ms_per_step"], 3), "ms")
PY

#!/bin/bash
# Timing probe of the compatibility kernel's halves (WRONG results on purpose): the kernel as it is, without the walk over
# the kept variants (what remains: list loads, bit rows through LDS, the way out), without the way out of a tile (what
# remains: loads + the factor products).   bash tools/compat_phases.sh
# The probe is compiled out of the shipped library: this script rebuilds it with -DGK_TIMING_PROBES=1 and restores the
# normal build at the end.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
GK_EXTRA_HIPCC_FLAGS="-DGK_TIMING_PROBES=1" python -c "from kir_graph_amd import build; build.buildNative(force=True)" > /dev/null 2>&1 || { echo "build failed"; exit 1; }
trap 'python -c "from kir_graph_amd import build; build.buildNative(force=True)" > /dev/null 2>&1' EXIT
for rep in 1 2; do
  for p in 0 1 2 3 4; do
    GK_TIMING_PROBES=1 GK_COMPAT_PROBE=$p timeout -k 10 200 python tools/bench_compat.py 2>&1 | grep -m1 "compat_kernel" | sed "s/^/probe $p: /"
  done
done

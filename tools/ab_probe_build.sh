#!/bin/bash
# Dev tool (GPU box): the compatibility kernel with its timing probe compiled out (the shipped build) against the build that
# carries it as a runtime argument (-DGK_TIMING_PROBES=1, probe 0), library rebuilt per setting, tools/bench_compat.py.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
trap 'python -c "from kir_graph_amd import build; build.buildNative(force=True)" > /dev/null 2>&1' EXIT
for rep in 1 2; do
  for flags in "" "-DGK_TIMING_PROBES=1"; do
    GK_EXTRA_HIPCC_FLAGS="$flags" python -c "from kir_graph_amd import build; build.buildNative(force=True)" > /dev/null 2>&1 || { echo "build failed"; exit 1; }
    timeout -k 10 200 python tools/bench_compat.py 2>&1 | grep -m1 "compat_kernel" | sed "s/^/build [$flags]: /"
  done
done

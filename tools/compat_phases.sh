#!/bin/bash
# Timing probe of the compatibility kernel's halves (WRONG results on purpose): the kernel as it is, without the walk over
# the kept variants (what remains: list loads, bit rows through LDS, the way out), without the way out of a tile (what
# remains: loads + the factor products).   bash tools/compat_phases.sh
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
for rep in 1 2; do
  for p in 0 1 2 3 4; do
    GK_TIMING_PROBES=1 GK_COMPAT_PROBE=$p timeout -k 10 200 python tools/bench_compat.py 2>&1 | grep -m1 "compat_kernel" | sed "s/^/probe $p: /"
  done
done

#!/bin/bash
# A / B on one box, every variant twice (interleaved): staging as a two-stage pipeline (copy of sample k+2 while k+1 is
# tabulated; GK_COPY_AHEAD=1, default) against copy + tabulation in one stage.   bash tools/ab_copy_ahead.sh [steps]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
STEPS=${1:-64}
cd $R
run() {
  local label=$1; shift
  local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done
  shift
  env "${envs[@]}" python bench.py --steps $STEPS --warmup 8 --cpu-pairs 0 --serial-steps 0 --no-pcie-leg "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('$label |', round(d['ms_per_step'],3), 'ms/step', round(d['host']['host_core_s_per_step']*1e3,1), 'core-ms/step', round(d['host']['cores_busy'],2), 'cores busy')"
}
for rep in 1 2; do
  for A in 1 0; do
    run "1 process x 3 lanes, 3 cores, copy ahead=$A" GK_PROCS_PER_GPU=1 GK_COPY_AHEAD=$A -- --cores-per-gpu 3
    run "1 process x 3 lanes, unpinned, copy ahead=$A" GK_PROCS_PER_GPU=1 GK_COPY_AHEAD=$A --
    run "2 processes x 2 lanes, unpinned, copy ahead=$A" GK_COPY_AHEAD=$A --
    run "2 processes x 2 lanes, 2 cores, copy ahead=$A" GK_COPY_AHEAD=$A -- --cores-per-gpu 2
  done
done

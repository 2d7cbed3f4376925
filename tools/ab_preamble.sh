#!/bin/bash
# A / B on one box, every variant three times (interleaved): the sample preamble as one library call with two waits
# (gk_sample_prepare_all, default) against the three calls with six.   bash tools/ab_preamble.sh [steps]
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
d=json.loads(sys.stdin.readlines()[-1]); print('$label |', round(d['ms_per_step'],3), 'ms/step', round(d['host']['host_core_s_per_step']*1e3,1), 'core-ms/step')"
}
for rep in 1 2 3; do
  for A in 1 0; do
    run "one call=$A, unpinned" GK_PREAMBLE_ONE_CALL=$A --
    run "one call=$A, 2 cores" GK_PREAMBLE_ONE_CALL=$A -- --cores-per-gpu 2
    run "one call=$A, 20 steps" GK_PREAMBLE_ONE_CALL=$A -- --steps 20 --warmup 5
  done
done

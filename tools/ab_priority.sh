#!/bin/bash
# A / B on one box, every variant twice (interleaved): high-priority streams for staging and for the sample preamble
# against plain ones (GK_STREAM_PRIORITY=0 turns both off, GK_URGENT_PREAMBLE=0 the second).
#   bash tools/ab_priority.sh [steps, default 64]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
STEPS=${1:-64}
cd $R
run() {
  local label=$1; shift
  env "$@" python bench.py --steps $STEPS --warmup 8 --cpu-pairs 0 --serial-steps 0 --no-pcie-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('$label |', round(d['ms_per_step'],3), 'ms/step', round(d['host']['host_core_s_per_step']*1e3,1), 'core-ms/step', round(d['host']['cores_busy'],2), 'cores busy')"
}
for rep in 1 2; do
  run "1 process x 3 lanes, staging + preamble urgent" GK_PROCS_PER_GPU=1 GK_SAMPLE_LANES=3
  run "1 process x 3 lanes, staging urgent" GK_PROCS_PER_GPU=1 GK_SAMPLE_LANES=3 GK_URGENT_PREAMBLE=0
  run "2 processes x 2 lanes, staging + preamble urgent" GK_PROCS_PER_GPU=2 GK_SAMPLE_LANES=2
  run "2 processes x 2 lanes, staging urgent" GK_PROCS_PER_GPU=2 GK_SAMPLE_LANES=2 GK_URGENT_PREAMBLE=0
  run "2 processes x 2 lanes, no priorities" GK_PROCS_PER_GPU=2 GK_SAMPLE_LANES=2 GK_STREAM_PRIORITY=0
  run "2 processes x 2 lanes, no priorities, lock-step search" GK_PROCS_PER_GPU=2 GK_SAMPLE_LANES=2 GK_STREAM_PRIORITY=0 GK_SAMPLE_PIPELINE=0
  run "4 processes x 1 lane, staging urgent" GK_PROCS_PER_GPU=4 GK_SAMPLE_LANES=1 GK_URGENT_PREAMBLE=0
  run "2 processes x 3 lanes, staging urgent" GK_PROCS_PER_GPU=2 GK_SAMPLE_LANES=3 GK_URGENT_PREAMBLE=0
done

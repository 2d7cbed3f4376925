#!/bin/bash
# A / B in one run on one box: high-priority streams for staging and for the sample preamble against plain ones
# (GK_STREAM_PRIORITY=0 turns both off, GK_URGENT_PREAMBLE=0 the second).   bash tools/ab_priority.sh [steps, default 48]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
STEPS=${1:-48}
cd $R
run() {
  local label=$1; shift
  env "$@" python bench.py --steps $STEPS --warmup 8 --cpu-pairs 0 --serial-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('$label |', round(d['ms_per_step'],3), 'ms/step', round(d['host']['host_core_s_per_step']*1e3,1), 'core-ms/step', round(d['host']['cores_busy'],2), 'cores busy')"
}
export GK_INGEST_THREADS=${GK_INGEST_THREADS:-1}
for PL in "1 3" "2 2" "2 1" "4 1"; do
  set -- $PL
  run "$1 processes x $2 lanes, priority: staging + preamble" GK_PROCS_PER_GPU=$1 GK_SAMPLE_LANES=$2
  run "$1 processes x $2 lanes, priority: staging only" GK_PROCS_PER_GPU=$1 GK_SAMPLE_LANES=$2 GK_URGENT_PREAMBLE=0
  run "$1 processes x $2 lanes, priority: none" GK_PROCS_PER_GPU=$1 GK_SAMPLE_LANES=$2 GK_STREAM_PRIORITY=0
done

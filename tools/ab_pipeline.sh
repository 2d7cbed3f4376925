#!/bin/bash
# A / B in one run on one box: the whole-sample search pipelined (marks, default) against lock-step (GK_SAMPLE_PIPELINE=0),
# in the one-process layouts and in the default one.   bash tools/ab_pipeline.sh [steps, default 48]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
STEPS=${1:-48}
cd $R
run() {
  local label=$1; shift
  env "$@" python bench.py --steps $STEPS --warmup 8 --cpu-pairs 0 --serial-steps 0 --no-pcie-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('$label |', round(d['ms_per_step'],3), 'ms/step', round(d['host']['host_core_s_per_step']*1e3,1), 'core-ms/step', round(d['host']['cores_busy'],2), 'cores busy')"
}
for P in 1 0; do
  for L in 1 2 3; do
    run "1 process x $L lanes, pipeline=$P" GK_PROCS_PER_GPU=1 GK_SAMPLE_LANES=$L GK_SAMPLE_PIPELINE=$P
  done
  run "2 processes x 2 lanes, pipeline=$P" GK_SAMPLE_PIPELINE=$P
  run "2 processes x 1 lane, pipeline=$P" GK_SAMPLE_LANES=1 GK_SAMPLE_PIPELINE=$P
done

#!/bin/bash
# Which worker layout for the driver's own command (20 steps, 5 warm-up)?  Every layout four times, interleaved, one box.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
for rep in 1 2 3 4; do
  for PL in "1 3" "1 4" "2 2" "2 3"; do
    set -- $PL
    GK_PROCS_PER_GPU=$1 GK_SAMPLE_LANES=$2 python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-pairs 0 --serial-steps 0 --no-pcie-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('$1 process(es) x $2 lanes |', round(d['ms_per_step'],3), 'ms/step', round(d['host']['host_core_s_per_step']*1e3,1), 'core-ms/step')"
  done
done

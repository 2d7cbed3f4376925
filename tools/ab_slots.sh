#!/bin/bash
# A / B on one box: at most S whole-sample searches at a time (GK_SEARCH_SLOTS) x sample lanes, for the driver's command
# (20 steps) and a longer run, every variant several times (interleaved).   bash tools/ab_slots.sh
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
for rep in 1 2 3 4; do
  for cfg in "0 3" "2 3" "2 4" "3 4"; do
    set -- $cfg
    for steps in "20 5" "64 8"; do
      set -- $cfg $steps
      GK_SEARCH_SLOTS=$1 GK_SAMPLE_LANES=$2 python bench.py --gpus 1 --steps $3 --warmup $4 --cpu-pairs 0 --serial-steps 0 --no-pcie-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('slots $1, lanes $2, $3 steps |', round(d['ms_per_step'],3), 'ms/step', round(d['host']['host_core_s_per_step']*1e3,1), 'core-ms')"
    done
  done
done

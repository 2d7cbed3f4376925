#!/bin/bash
# A / B on one box, round 4: sample lanes x whole-sample searches at a time, host-memory legs of 48 steps (one leg per
# setting and repeat, interleaved).   bash tools/ab_lanes_r04.sh
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
for rep in 1 2 3 4 5; do
  for cfg in ${GK_AB_CFGS:-2:3 2:2 3:3 2:4 3:4 1:3}; do      # slots:lanes
    set -- ${cfg/:/ }
    GK_SEARCH_SLOTS=$1 GK_SAMPLE_LANES=$2 python bench.py --gpus 1 --steps 48 --warmup 10 --cpu-pairs 0 --serial-steps 0 --cli-samples 0 --legs 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('slots $1, lanes $2 | host', round(d['ms_per_step'],3), 'hbm', round(d['hbm_resident']['ms_per_step'],3), 'ms/step', round(d['host']['host_core_s_per_step']*1e3,1), 'core-ms')"
  done
done

#!/bin/bash
# Host budget of one rank (VERDICT r02 #1b): the bench step with the rank pinned to K cores, K = 2 4 8 16 and unpinned,
# with the default worker layout and with ONE worker process.  One JSON line per run -> gpurun_out/<tag>/host_budget.jsonl
#   bash tools/host_budget_sweep.sh [tag, default r03] [steps, default 48]
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r03}
STEPS=${2:-48}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
: > $O/host_budget.jsonl
run() {   # label, env assignments..., then -- bench flags
  local label=$1; shift
  local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done
  shift
  echo "[sweep] $label" >&2
  env "${envs[@]}" python bench.py --steps $STEPS --warmup 8 --cpu-pairs 0 --serial-steps 0 "$@" 2>> $O/host_budget.err \
    | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); d['label']='$label'; print(json.dumps({k: d[k] for k in ('label','ms_per_step','value','host','config')}))" >> $O/host_budget.jsonl
  tail -1 $O/host_budget.jsonl | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['label'], round(d['ms_per_step'],3), 'ms/step', round(d['host']['host_core_s_per_step']*1e3,2), 'core-ms/step', round(d['host']['cores_busy'],2), 'cores busy')"
}
run "default layout, unpinned" X=1 --
for K in 16 8 4 2; do
  run "default layout, $K cores" X=1 -- --cores-per-gpu $K
done
run "one process, unpinned" GK_PROCS_PER_GPU=1 --
for K in 8 4 3 2; do
  run "one process, $K cores" GK_PROCS_PER_GPU=1 -- --cores-per-gpu $K
done
run "one process, 3 cores, blocking waits" GK_PROCS_PER_GPU=1 GK_WAIT_POLICY=block -- --cores-per-gpu 3
run "default layout, 4 cores, blocking waits" GK_WAIT_POLICY=block -- --cores-per-gpu 4
run "default layout, 2 cores, blocking waits" GK_WAIT_POLICY=block -- --cores-per-gpu 2
cat $O/host_budget.jsonl | wc -l

#!/bin/bash
# One worker process per GPU with the whole-sample search (gk_sample_search): sample lanes x cores x wait policy.
#   bash tools/lane_sweep.sh [tag, default r03] [steps, default 48]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r03}
STEPS=${2:-48}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
: > $O/lane_sweep.jsonl
run() {
  local label=$1; shift
  local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done
  shift
  env "${envs[@]}" python bench.py --steps $STEPS --warmup 8 --cpu-pairs 0 --serial-steps 0 --no-pcie-leg "$@" 2>> $O/lane_sweep.err \
    | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); d['label']='$label'; print(json.dumps({k: d[k] for k in ('label','ms_per_step','value','host','config','search_steps')}))" >> $O/lane_sweep.jsonl
  tail -1 $O/lane_sweep.jsonl | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['label'], '|', round(d['ms_per_step'],3), 'ms/step', round(d['host']['host_core_s_per_step']*1e3,2), 'core-ms/step', round(d['host']['cores_busy'],2), 'cores busy')"
}
for L in 1 2 3; do
  run "1 proc, $L lanes, unpinned" GK_PROCS_PER_GPU=1 GK_SAMPLE_LANES=$L --
done
run "1 proc, 2 lanes, 3 cores" GK_PROCS_PER_GPU=1 GK_SAMPLE_LANES=2 -- --cores-per-gpu 3
run "1 proc, 2 lanes, 2 cores" GK_PROCS_PER_GPU=1 GK_SAMPLE_LANES=2 -- --cores-per-gpu 2
run "1 proc, 2 lanes, 3 cores, block" GK_PROCS_PER_GPU=1 GK_SAMPLE_LANES=2 GK_WAIT_POLICY=block -- --cores-per-gpu 3
run "1 proc, 2 lanes, 2 cores, block" GK_PROCS_PER_GPU=1 GK_SAMPLE_LANES=2 GK_WAIT_POLICY=block -- --cores-per-gpu 2
run "1 proc, 3 lanes, 3 cores, block" GK_PROCS_PER_GPU=1 GK_SAMPLE_LANES=3 GK_WAIT_POLICY=block -- --cores-per-gpu 3
run "1 proc, 2 lanes, 3 cores, yield" GK_PROCS_PER_GPU=1 GK_SAMPLE_LANES=2 GK_WAIT_POLICY=yield -- --cores-per-gpu 3
run "2 procs, 2 lanes, 4 cores, block" GK_PROCS_PER_GPU=2 GK_SAMPLE_LANES=2 GK_WAIT_POLICY=block -- --cores-per-gpu 4
run "1 proc, per-gene threads (GK_SAMPLE_SEARCH=0), unpinned" GK_PROCS_PER_GPU=1 GK_SAMPLE_SEARCH=0 --

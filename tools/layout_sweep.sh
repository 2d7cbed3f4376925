#!/bin/bash
# Worker layout of one rank with the whole-sample search: processes x sample lanes x cores x wait policy (bench step).
#   bash tools/layout_sweep.sh [tag] [steps]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r03}
STEPS=${2:-48}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
: > $O/layout_sweep.jsonl
run() {
  local label=$1; shift
  local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done
  shift
  env "${envs[@]}" python bench.py --steps $STEPS --warmup 8 --cpu-pairs 0 --serial-steps 0 "$@" 2>> $O/layout_sweep.err \
    | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); d['label']='$label'; print(json.dumps({k: d[k] for k in ('label','ms_per_step','value','host','config','search_steps')}))" >> $O/layout_sweep.jsonl
  tail -1 $O/layout_sweep.jsonl | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['label'], '|', round(d['ms_per_step'],3), 'ms/step', round(d['host']['host_core_s_per_step']*1e3,2), 'core-ms/step', round(d['host']['cores_busy'],2), 'cores busy')"
}
for cfg in "$@"; do :; done
run "2p x 2l block unpinned" GK_PROCS_PER_GPU=2 GK_SAMPLE_LANES=2 GK_WAIT_POLICY=block --
run "3p x 2l block unpinned" GK_PROCS_PER_GPU=3 GK_SAMPLE_LANES=2 GK_WAIT_POLICY=block --
run "2p x 3l block unpinned" GK_PROCS_PER_GPU=2 GK_SAMPLE_LANES=3 GK_WAIT_POLICY=block --
run "2p x 2l spin unpinned" GK_PROCS_PER_GPU=2 GK_SAMPLE_LANES=2 --
run "4p x 1l block unpinned" GK_PROCS_PER_GPU=4 GK_SAMPLE_LANES=1 GK_WAIT_POLICY=block --
run "2p x 2l block 2 cores" GK_PROCS_PER_GPU=2 GK_SAMPLE_LANES=2 GK_WAIT_POLICY=block -- --cores-per-gpu 2
run "1p x 3l block 2 cores" GK_PROCS_PER_GPU=1 GK_SAMPLE_LANES=3 GK_WAIT_POLICY=block -- --cores-per-gpu 2
run "2p x 2l block 3 cores" GK_PROCS_PER_GPU=2 GK_SAMPLE_LANES=2 GK_WAIT_POLICY=block -- --cores-per-gpu 3

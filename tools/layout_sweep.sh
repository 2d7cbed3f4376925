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
  env "${envs[@]}" python bench.py --steps $STEPS --warmup 8 --cpu-pairs 0 --serial-steps 0 --no-pcie-leg "$@" 2>> $O/layout_sweep.err \
    | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); d['label']='$label'; print(json.dumps({k: d[k] for k in ('label','ms_per_step','value','host','config','search_steps')}))" >> $O/layout_sweep.jsonl
  tail -1 $O/layout_sweep.jsonl | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['label'], '|', round(d['ms_per_step'],3), 'ms/step', round(d['host']['host_core_s_per_step']*1e3,2), 'core-ms/step', round(d['host']['cores_busy'],2), 'cores busy')"
}
for PL in "1 3" "1 4" "2 2" "2 3" "3 2" "1 2" "2 1" "4 1"; do
  set -- $PL
  run "$1p x $2l unpinned" GK_PROCS_PER_GPU=$1 GK_SAMPLE_LANES=$2 --
done
run "1p x 3l 3 cores" GK_PROCS_PER_GPU=1 GK_SAMPLE_LANES=3 -- --cores-per-gpu 3
run "1p x 3l 2 cores" GK_PROCS_PER_GPU=1 GK_SAMPLE_LANES=3 -- --cores-per-gpu 2
run "1p x 4l 2 cores" GK_PROCS_PER_GPU=1 GK_SAMPLE_LANES=4 -- --cores-per-gpu 2
run "2p x 2l 2 cores" GK_PROCS_PER_GPU=2 GK_SAMPLE_LANES=2 -- --cores-per-gpu 2
run "2p x 2l 3 cores" GK_PROCS_PER_GPU=2 GK_SAMPLE_LANES=2 -- --cores-per-gpu 3

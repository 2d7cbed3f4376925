#!/bin/bash
# Host budget of one rank (VERDICT r02 #1b): the bench step with the rank pinned to K cores, K = 16 8 4 3 2 and unpinned,
# with the default layout (ONE worker process, 3 samples in flight, whole-sample search, blocking waits), with two
# worker processes, and with the round-2 layout (a thread and a stream per gene, 4 processes x 3 threads, spinning waits).
# One JSON line per run -> gpurun_out/<tag>/host_budget.jsonl, one summary line each on stdout.
#   bash tools/host_budget_sweep.sh [tag, default r03] [steps, default 48]
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
  env "${envs[@]}" python bench.py --steps $STEPS --warmup 8 --cpu-pairs 0 --serial-steps 0 --no-pcie-leg "$@" 2>> $O/host_budget.err \
    | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); d['label']='$label'; print(json.dumps({k: d[k] for k in ('label','ms_per_step','value','host','config')}))" >> $O/host_budget.jsonl
  tail -1 $O/host_budget.jsonl | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-58s %7.3f ms/step %7.2f core-ms/step %5.2f cores busy' % (d['label'], d['ms_per_step'], d['host']['host_core_s_per_step']*1e3, d['host']['cores_busy']))"
}
run "default (1 process x 3 lanes, block), unpinned" X=1 --
for K in 16 8 4 3 2; do
  run "default (1 process x 3 lanes, block), $K cores" X=1 -- --cores-per-gpu $K
done
run "two processes x 2 lanes, block, unpinned" GK_PROCS_PER_GPU=2 --
for K in 4 3 2; do
  run "two processes x 2 lanes, block, $K cores" GK_PROCS_PER_GPU=2 -- --cores-per-gpu $K
done
run "one process x 4 lanes, block, 2 cores" GK_SAMPLE_LANES=4 -- --cores-per-gpu 2
run "round-2 layout (per-gene threads, 4 procs x 3, spin), unpinned" GK_SAMPLE_SEARCH=0 GK_PROCS_PER_GPU=4 GK_WAIT_POLICY=spin --
run "round-2 layout (per-gene threads, 4 procs x 3, spin), 4 cores" GK_SAMPLE_SEARCH=0 GK_PROCS_PER_GPU=4 GK_WAIT_POLICY=spin -- --cores-per-gpu 4
run "round-2 layout (per-gene threads, 4 procs x 3, spin), 2 cores" GK_SAMPLE_SEARCH=0 GK_PROCS_PER_GPU=4 GK_WAIT_POLICY=spin -- --cores-per-gpu 2
wc -l < $O/host_budget.jsonl

#!/bin/bash
# GPU occupancy of the one-process layouts (tools/gpu_busy.py): bash tools/one_process_busy.sh [tag]
TAG=${1:-r03}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export GK_PROCS_PER_GPU=1
for lanes in 1 2 3; do
  export GK_SAMPLE_LANES=$lanes
  rm -rf $O/busy_$lanes
  rocprofv3 --kernel-trace -d $O/busy_$lanes -o p --output-format csv -- python3 $R/bench.py --cpu-pairs 0 --serial-steps 0 --no-pcie-leg --steps 40 --warmup 8 > $O/busy_$lanes.json 2> $O/busy_$lanes.err
  echo "== one process, $lanes lane(s): $(python3 -c "import json; print(round(json.load(open('$O/busy_$lanes.json'))['ms_per_step'], 3))") ms/step under the profiler"
  python3 $R/tools/gpu_busy.py $O/busy_$lanes/p_kernel_trace.csv 0.4 0.9
  rm -rf $O/busy_$lanes
done

#!/bin/bash
# Dev tool (GPU box): the LDS stride of setsum_leaves' staged columns, 33 doubles (odd) against 40 (windows of 16 banks at
# multiples of 16: two columns collide only when their distance is a multiple of 4), library rebuilt per setting.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
for rep in 1 2; do
  for ld in 33 40; do
    GK_EXTRA_HIPCC_FLAGS="-DGK_STAGE_LD=$ld" python -c "from kir_graph_amd import build; build.buildNative(force=True)" > /dev/null 2>&1 || { echo "build failed"; exit 1; }
    python bench.py --steps 12 --warmup 4 --legs 1 --one-kind --inputs hbm --cli-samples 0 --cpu-pairs 0 > gpurun_out/ab_ld_$ld.json 2> gpurun_out/ab_ld_$ld.log
    python3 -c "
import json
d = json.load(open('gpurun_out/ab_ld_$ld.json')); k = d['kernels_serial']['kernels']
print('GK_STAGE_LD=$ld  fraction %.3f ms per sample  (serial sum %.3f)' % (k['fraction_chunks']['ms_per_step'], d['kernels_serial']['kernel_ms_per_step']))"
  done
done
python -c "from kir_graph_amd import build; build.buildNative(force=True)" > /dev/null 2>&1

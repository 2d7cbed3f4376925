#!/bin/bash
# Dev tool: forms of the compatibility kernel side by side on the bench sample, twice each (interleaved), then the
# parity tests with the form under test.  A variant is a list of NAME=value settings joined by commas, e.g.
#   GK_AB_VARIANTS="GK_COMPAT_FORM=select GK_COMPAT_FORM=fma" GK_AB_TEST="GK_COMPAT_FORM=fma" bash tools/ab_compat.sh
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
mkdir -p gpurun_out/r03
for rep in 1 2; do
  for v in ${GK_AB_VARIANTS:-GK_COMPAT_FORM=select GK_COMPAT_FORM=fma}; do
    env ${v//,/ } timeout -k 10 200 python tools/bench_compat.py > gpurun_out/r03/bench_compat_ab.txt 2>&1
    echo "== $v"; grep -E "compat_kernel|wall" gpurun_out/r03/bench_compat_ab.txt | head -2
  done
done
if [ -n "$GK_AB_TEST" ]; then
  env ${GK_AB_TEST//,/ } timeout -k 10 600 python -m pytest tests/test_gpu_typing.py tests/test_gpu_golden.py tests/test_gpu_edge_cases.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -2
fi

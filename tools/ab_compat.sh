#!/bin/bash
# Dev tool: the compatibility kernel's forms side by side (GK_COMPAT) on the bench sample + PMC of the default form.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
mkdir -p gpurun_out/r03
tools/valu_rate.bin > gpurun_out/r03/valu_rate.txt 2>&1
grep "select factor" gpurun_out/r03/valu_rate.txt | cut -c1-140
for v in ${GK_AB_VARIANTS:-scalar lds ldsvcc}; do
  GK_COMPAT=$v timeout -k 10 200 python tools/bench_compat.py > gpurun_out/r03/bench_compat_$v.txt 2>&1
  echo "== $v"; grep -E "compat_kernel|wall" gpurun_out/r03/bench_compat_$v.txt | head -2
done
if [ -n "$GK_AB_TEST" ]; then
  GK_COMPAT=$GK_AB_TEST timeout -k 10 400 python -m pytest tests/test_gpu_typing.py tests/test_gpu_golden.py tests/test_gpu_edge_cases.py -m gpu -x -q 2>&1 | tail -2
fi
if [ -n "$GK_AB_PMC" ]; then
  GK_COMPAT=$GK_AB_PMC timeout -k 10 500 bash tools/pmc_compat.sh r03/pmc_compat_$GK_AB_PMC > gpurun_out/r03/pmc_compat_$GK_AB_PMC.txt 2>&1
  tail -30 gpurun_out/r03/pmc_compat_$GK_AB_PMC.txt
fi

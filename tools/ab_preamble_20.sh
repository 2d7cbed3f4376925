cd $GRAFT_REPO_ROOT
for rep in 1 2 3 4 5 6; do
  for A in 1 0; do
    GK_PREAMBLE_ONE_CALL=$A python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-pairs 0 --serial-steps 0 --no-pcie-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('one call=$A, 20 steps |', round(d['ms_per_step'],3), 'ms/step')"
  done
done

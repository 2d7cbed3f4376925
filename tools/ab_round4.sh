#!/bin/bash
# A/B of round 4's switches on one box: every line = the driver's command (20 steps) under one setting
# usage: tools/ab_round4.sh OUTDIR
O=${1:-gpurun_out}
run() {
  name=$1; shift
  env "$@" python bench.py --steps 20 --warmup 5 --cpu-pairs 0 --cli-samples 0 > $O/ab_$name.json 2> $O/ab_$name.log || return 1
  python - "$O/ab_$name.json" "$name" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
ks = d.get("kernels_serial", {}).get("kernels", {})
pick = lambda k: round(ks.get(k, {}).get("ms_per_step", 0.0), 3)
print(f"{sys.argv[2]:24s} host {d['ms_per_step']:.3f} ms  hbm {d.get('hbm_resident', {}).get('ms_per_step', 0):.3f} ms  "
      f"serial {d['kernels_serial']['kernel_ms_per_step']:.3f}  compat {pick('compat_kernel')}  tab_count {pick('tab_count')}  "
      f"fraction {pick('fraction_chunks')}  combine {pick('combine_chunks')}  minsum {pick('minsum_sad')}", flush=True)
PY
}
run default GK_DUMMY=1 && run h2d_kernel GK_H2D=kernel && run no_uniform_cut GK_COMPAT_UNIFORM=0 && run setsum_tiles GK_SETSUM=tiles && run default_again GK_DUMMY=1

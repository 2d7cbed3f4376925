#!/bin/bash
# A/B of round 4's switches on one box: every line = the driver's command (20 steps) under one setting
# usage: tools/ab_round4.sh OUTDIR name=ENV=VAL[,ENV=VAL...] ...      (name=- : no setting)
O=${1:-gpurun_out}; shift
run() {
  name=$1; shift
  env "$@" python bench.py --steps 20 --warmup 5 --cpu-pairs 0 --cli-samples 0 > $O/ab_$name.json 2> $O/ab_$name.log || return 1
  python - "$O/ab_$name.json" "$name" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
ks = d.get("kernels_serial", {}).get("kernels", {})
pick = lambda k: round(ks.get(k, {}).get("ms_per_step", 0.0), 3)
legs = lambda rows: "/".join(f"{r['ms_per_step']:.2f}" for r in rows)
print(f"{sys.argv[2]:18s} host {d['ms_per_step']:.3f} ({legs(d['legs'])})  hbm {d.get('hbm_resident', {}).get('ms_per_step', 0):.3f} "
      f"({legs(d.get('hbm_resident', {}).get('legs', []))})  serial {d['kernels_serial']['kernel_ms_per_step']:.3f}  compat {pick('compat_kernel')}  "
      f"tab_count {pick('tab_count')}  fraction {pick('fraction_chunks')}  combine {pick('combine_chunks')}  minsum {pick('minsum_sad')}", flush=True)
PY
}
for spec in "$@"; do
  name=${spec%%=*}; envs=${spec#*=}
  if [ "$envs" = "-" ]; then run $name GK_DUMMY=1 || exit 1; else run $name ${envs//,/ } || exit 1; fi
done

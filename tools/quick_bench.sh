#!/bin/bash
# Dev tool: the bench step with the default layout + the serial per-kernel table, one line each.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
python bench.py --steps ${1:-48} --warmup 8 --cpu-pairs 0 --serial-steps 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print(round(d['ms_per_step'],3), 'ms/step', round(d['value']/1e6,1), 'M reads/s; host', round(d['host']['host_core_s_per_step']*1e3,1), 'core-ms/step; value table', d.get('value_table_entries'))
ks=d['kernels_serial']; print('serial kernels', round(ks['kernel_ms_per_step'],3), {k: round(v['ms_per_step'],3) for k,v in ks['kernels'].items()})
r=d['roofline']; print('roofline', r['kernel'], 'frac', round(r['frac'],3), 'guide', round(r.get('frac_guide',0),3), 'avg_launch_ms', round(r['avg_launch_ms'],4))"

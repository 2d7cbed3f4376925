#!/bin/bash
# Round 4: threads per ingest x samples ingested at a time for the command line (tools/bench_cli.py, 1 M pairs per sample,
# hand-off as compact records), inputs made once (the runs share a parent shell).   bash tools/ingest_sweep_r04.sh [samples]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
N=${1:-24}
cd $R
for rep in 1 2; do
  for cfg in "8 3" "8 4" "6 4" "5 5" "4 6" "12 2"; do
    set -- $cfg
    GK_PACK_THREADS=$1 GK_INGEST_AHEAD=$2 python tools/bench_cli.py 1000000 $N --no-variant-json > /tmp/cli_sweep.txt 2> /tmp/cli_sweep.err
    echo "threads per ingest $1, samples ahead $2 | $(cut -d'(' -f1 /tmp/cli_sweep.txt | sed 's/command line: //') | $(grep 'process CPU' /tmp/cli_sweep.err | sed 's/.*= \([0-9.]* cores busy, [0-9.]* core-s per sample\).*/\1/')"
  done
done

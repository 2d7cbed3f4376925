#!/bin/bash
# Threads per ingest x samples ingested at a time: wall time and host core-seconds per sample of the command line
# (tools/bench_cli.py, 1 M pairs per sample).   bash tools/ingest_threads_sweep.sh [samples, default 24]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
N=${1:-24}
cd $R
for cfg in "8 3" "4 4" "2 6" "2 8" "1 8" "1 12" "1 16" "8 3"; do
  set -- $cfg
  GK_PACK_THREADS=$1 GK_INGEST_AHEAD=$2 python tools/bench_cli.py 1000000 $N --no-variant-json > /tmp/cli_sweep.txt 2> /tmp/cli_sweep.err
  echo "threads per ingest $1, samples ahead $2 | $(cut -d'(' -f1 /tmp/cli_sweep.txt | sed 's/command line: //') | $(grep 'process CPU' /tmp/cli_sweep.err | sed 's/.*= \([0-9.]* cores busy, [0-9.]* core-s per sample\).*/\1/')"
done

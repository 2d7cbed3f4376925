cd $GRAFT_REPO_ROOT
for cfg in "8 3 always" "8 3 lazy" "2 6 lazy" "1 8 lazy" "1 6 lazy" "2 4 lazy"; do
  set -- $cfg
  GK_PACK_THREADS=$1 GK_INGEST_AHEAD=$2 GK_HANDOFF=$3 python tools/bench_cli.py 1000000 24 --no-variant-json > /tmp/cli_sweep.txt 2> /tmp/cli_sweep.err
  echo "threads per ingest $1, samples ahead $2, hand-off files $3 | $(cut -d'(' -f1 /tmp/cli_sweep.txt | sed 's/command line: //') | $(grep 'process CPU' /tmp/cli_sweep.err | sed 's/.*= \([0-9.]* cores busy, [0-9.]* core-s per sample\).*/\1/')"
done

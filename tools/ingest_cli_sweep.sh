mkdir -p gpurun_out/sweep
run() { # threads ahead
  echo "== GK_PACK_THREADS=$1 ingest_ahead=$2" >> gpurun_out/sweep/cli_sweep.txt
  GK_PACK_THREADS=$1 GK_TEST_HOOKS=ingest_ahead=$2 timeout -k 5 120 python tools/bench_cli.py 1000000 24 --no-variant-json >> gpurun_out/sweep/cli_sweep.txt 2>&1
}
run 8 3 && run 16 2 && run 16 3 && run 12 4 && run 8 3

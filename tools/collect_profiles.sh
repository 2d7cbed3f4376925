#!/bin/bash
# Regenerates the round's evidence under gpurun_out/<tag> (run on the GPU box through gpurun); the summaries that are
# judged are then copied into profiles/ in the build container: bash tools/install_profiles.sh <tag>.
#   bash tools/collect_profiles.sh [round tag, default r05] [workloads, default "cfg2_exonfirst cfg2_em cfg1_pv"] [bench: 1|0]
# Per workload (BASELINE.json configs[2] exon-first / EM, configs[1] pv) one sample at a time on ONE stream -- the mode of
# the bench's own serial pass, the basis of `roofline` and `kernels_serial` (children are not allowed under the profiler
# on this pool):
#   1. rocprofv3 --kernel-trace --stats: per-kernel launches and average durations (the kernels' own names);
#   2. --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes: HBM bytes per launch (tools/pmc_traffic.py; FETCH_SIZE
#      counts half of coalesced reads on gfx950 -- the guide's correction);
#   3. two passes of SQ counters (issue / wait / LDS), summarised per kernel (tools/pmc_summary.py).
# Then the driver's own command, AFTER the traffic files of this code are in place (bench.py only reports a traffic
# figure whose recorded source digest is the running code's).
set -e
R=$GRAFT_REPO_ROOT
TAG=${1:-r05}
WORKLOADS=${2:-"cfg2_exonfirst cfg2_em cfg1_pv"}
RUN_BENCH=${3:-1}
O=$R/gpurun_out/$TAG
mkdir -p $O
KERNELS="compat_kernel tab_count tab_expand minsum_sad setsum_leaves colsum_chunks count_ids_genes flag_nonempty flag_pairs fraction_chunks maxsum_chunks patch_pending em_sets_groups em_sets_verify em_sets_emit em_kernel_genes"
cd /tmp && export TMPDIR=/tmp
export GK_PREFETCH=0 GK_SAMPLE_LANES=1
for W in $WORKLOADS; do
  case $W in
    cfg2_*) PAIRS=10000000; STEPS=6; PSTEPS=3;;
    *) PAIRS=1000000; STEPS=40; PSTEPS=16;;
  esac
  METHOD=${W#*_}
  CMD="python3 $R/bench.py --pairs $PAIRS --method $METHOD --distinct 1 --warmup 2 --legs 1 --one-kind --inputs hbm --serial-steps 2 --cpu-pairs 0 --cli-samples 0"
  mkdir -p $O/$W
  rocprofv3 --kernel-trace --stats -d $O/$W/stats -o b --output-format csv -- $CMD --steps $STEPS > $O/$W/bench_under_rocprof.json 2> $O/$W/stats.err
  echo "[collect] $W kernel stats done"
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/$W/bench_fetch -o p --output-format csv -- $CMD --steps $PSTEPS > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/$W/bench_write -o p --output-format csv -- $CMD --steps $PSTEPS > /dev/null 2>&1
  echo "[collect] $W traffic passes done"
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-trace -d $O/$W/pmc_sq1 -o p --output-format csv -- $CMD --steps $PSTEPS > /dev/null 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_INSTS_SMEM --kernel-trace -d $O/$W/pmc_sq2 -o p --output-format csv -- $CMD --steps $PSTEPS > /dev/null 2>&1
  echo "[collect] $W SQ passes done"
  F=$(find $O/$W/bench_fetch -name 'p_counter_collection.csv' | head -1)
  G=$(find $O/$W/bench_write -name 'p_counter_collection.csv' | head -1)
  for k in $KERNELS; do
    python3 $R/tools/pmc_traffic.py $F $G $k "$W" > $O/$W/traffic_$k.json
    python3 $R/tools/pmc_summary.py $O/$W $k > $O/$W/pmc_$k.txt
    if grep -q '"launches": 0' $O/$W/traffic_$k.json; then rm -f $O/$W/traffic_$k.json $O/$W/pmc_$k.txt; fi
  done
  cp $(find $O/$W/stats -name 'b_kernel_stats.csv' | head -1) $O/$W/kernel_stats.csv
  # gpurun merges at most 64 MiB back: the raw per-dispatch counter files have been summarised above
  rm -rf $O/$W/bench_fetch $O/$W/bench_write $O/$W/pmc_sq1 $O/$W/pmc_sq2 $O/$W/stats
  for f in $O/$W/traffic_*.json; do
    k=$(basename $f .json); k=${k#traffic_}
    cp $f $R/profiles/${TAG}_traffic_${W}_$k.json
  done
done
unset GK_PREFETCH GK_SAMPLE_LANES      # the serial mode was for the profiler only
if [ "$RUN_BENCH" = "1" ]; then
  cd $R
  python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
  echo "[collect] bench done"
  tail -c 1500 $O/bench.json
fi
ls $O $O/*
echo "then, in the build container: bash tools/install_profiles.sh $TAG   (copies the summaries into profiles/)"

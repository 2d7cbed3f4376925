#!/bin/bash
# Regenerates the round's evidence under gpurun_out/rNN (run on the GPU box through gpurun); the summaries that are
# judged are then copied into profiles/ in the build container: bash tools/install_profiles.sh <tag>.
#   bash tools/collect_profiles.sh [round tag, default r04]
set -e
R=$GRAFT_REPO_ROOT
TAG=${1:-r04}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
# 1. the roofline basis under rocprofv3: ONE process, ONE gene thread, no prefetch -- kernels back to back, the
#    same mode as the bench's own serial pass (children are not allowed under the profiler on this pool)
cd /tmp && export TMPDIR=/tmp
export GK_PROCS_PER_GPU=1 GK_THREADS=1 GK_PREFETCH=0 GK_SAMPLE_LANES=1 GK_SAMPLE_STREAMS=1
SERIAL="python3 $R/bench.py --cpu-pairs 0 --steps 40 --warmup 8 --serial-steps 2 --inputs hbm --one-kind --legs 1 --cli-samples 0"
rocprofv3 --kernel-trace --stats -d $O/stats -o b --output-format csv -- $SERIAL > $O/bench_under_rocprof.json 2> $O/stats.err
echo "[collect] kernel stats done"
# the counter passes run fewer steps: their figures are per-launch averages and the raw per-dispatch CSVs are large
SERIAL="python3 $R/bench.py --cpu-pairs 0 --steps 16 --warmup 4 --serial-steps 2 --inputs hbm --one-kind --legs 1 --cli-samples 0"
# 2. HBM traffic of every kernel over the same command: FETCH_SIZE and WRITE_SIZE in separate passes (the guide's
#    recipe; FETCH_SIZE counts half of coalesced reads on gfx950, corrected in tools/pmc_traffic.py)
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/bench_fetch -o p --output-format csv -- $SERIAL > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/bench_write -o p --output-format csv -- $SERIAL > /dev/null 2>&1
echo "[collect] traffic passes done"
# 3. issue / stall / LDS counters of the same command, two passes (counter groups that fit together)
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-trace -d $O/pmc_sq1 -o p --output-format csv -- $SERIAL > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_INSTS_SMEM --kernel-trace -d $O/pmc_sq2 -o p --output-format csv -- $SERIAL > /dev/null 2>&1
echo "[collect] SQ passes done"
for k in compat_kernel tab_count minsum_sad setsum_leaves fraction_chunks maxsum_chunks select_cut count_ids patch_pending; do
  python3 $R/tools/pmc_traffic.py $O/bench_fetch/p_counter_collection.csv $O/bench_write/p_counter_collection.csv $k > $O/traffic_$k.json
  python3 $R/tools/pmc_summary.py $O $k > $O/pmc_$k.txt
done
# gpurun merges at most 64 MiB back: the raw per-dispatch counter files have been summarised above
rm -f $O/bench_fetch/*.csv $O/bench_write/*.csv $O/pmc_sq1/*.csv $O/pmc_sq2/*.csv
# 4. the driver's own command, AFTER the traffic files of this code are in place (bench.py only reports a traffic
#    figure whose recorded source digest is the running code's): throughput line with roofline (serial pass) and CPU baseline
cd $R
unset GK_PROCS_PER_GPU GK_THREADS GK_PREFETCH GK_SAMPLE_LANES GK_SAMPLE_STREAMS      # the serial mode was for the profiler only
cp $O/traffic_compat_kernel.json $R/profiles/${TAG}_bench_traffic.json
for k in compat_kernel tab_count minsum_sad setsum_leaves fraction_chunks maxsum_chunks select_cut count_ids patch_pending; do
  cp $O/traffic_$k.json $R/profiles/${TAG}_traffic_$k.json
done
python bench.py > $O/bench.json 2> $O/bench.err
echo "[collect] bench done"
ls $O $O/stats
cat $O/bench.json
echo "then, in the build container: bash tools/install_profiles.sh $TAG   (copies the summaries into profiles/)"

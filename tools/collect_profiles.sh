set -e
R=$GRAFT_REPO_ROOT
# the calibration tool is built on first use
[ -x $R/tools/fetch_calib.bin ] || hipcc --offload-arch=gfx950 -O3 $R/tools/fetch_calib.hip -o $R/tools/fetch_calib.bin
O=$R/gpurun_out/r01
mkdir -p $O
cd $R
python bench.py --verbose > $O/bench.json 2> $O/bench.err
cd /tmp && export TMPDIR=/tmp
# the profiled runs are one process per GPU (GK_PROCS_PER_GPU=1): child processes under the profiler are not
# allowed on this pool, and the per-kernel durations are then not stretched by the other workers' kernels
export GK_PROCS_PER_GPU=1
rocprofv3 --kernel-trace --stats -d $O/stats -o b --output-format csv -- python3 $R/bench.py --cpu-pairs 0 > $O/bench_under_rocprof.json 2> $O/stats.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-trace -d $O/pmc_sq1 -o p --output-format csv -- python3 $R/tools/bench_maxsum.py 67000 220 600 2 3 > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_INSTS_SMEM --kernel-trace -d $O/pmc_sq2 -o p --output-format csv -- python3 $R/tools/bench_maxsum.py 67000 220 600 2 3 > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_fetch -o p --output-format csv -- python3 $R/tools/bench_maxsum.py 67000 220 600 2 3 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_write -o p --output-format csv -- python3 $R/tools/bench_maxsum.py 67000 220 600 2 3 > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/cal_fetch -o p --output-format csv -- $R/tools/fetch_calib.bin > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/cal_write -o p --output-format csv -- $R/tools/fetch_calib.bin > /dev/null 2>&1
ls $O $O/stats
cat $O/bench.json
# HBM traffic of the dominant kernel over the bench's own launch mix (two more passes, as the guide prescribes)
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/bench_fetch -o p --output-format csv -- python3 $R/bench.py --cpu-pairs 0 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/bench_write -o p --output-format csv -- python3 $R/bench.py --cpu-pairs 0 > /dev/null 2>&1
python3 $R/tools/pmc_traffic.py $O/bench_fetch/p_counter_collection.csv $O/bench_write/p_counter_collection.csv maxsum_chunks > $O/bench_traffic.json
cat $O/bench_traffic.json

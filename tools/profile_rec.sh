#!/bin/bash
# Profiling run of the recurrent bench (rec_mappo, BASELINE config 4 shape): bench line, rocprofv3 kernel stats, PMC passes.
#   bash tools/profile_rec.sh <tag>      -> profiles/<tag>_kernel_stats.csv, profiles/<tag>_pmc_traffic.json (+ gpurun_out/<tag>/)
set -o pipefail
TAG=$1; shift
OUT=gpurun_out/$TAG
ARGS="--system rec_mappo --env smax --scenario 3s5z --envs 2048 $@"
mkdir -p $OUT
export TMPDIR=/tmp
python3 bench.py --steps 6 --warmup 2 $ARGS > $OUT/bench.json 2> $OUT/bench.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timers --no-secondary --repeats 1 $ARGS > $OUT/stats.json 2> $OUT/stats.err || exit 1
python3 tools/trace_by_kernel.py $OUT/stats 0.05 > $OUT/by_kernel.txt || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_r -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timers --no-secondary --repeats 1 $ARGS > /dev/null 2> $OUT/pmc_r.err || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_w -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timers --no-secondary --repeats 1 $ARGS > /dev/null 2> $OUT/pmc_w.err || exit 1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timers --no-secondary --repeats 1 $ARGS > /dev/null 2> $OUT/pmc_sq.err || exit 1
python3 tools/profile_summarise.py $TAG $OUT/stats $OUT/pmc_r $OUT/pmc_w $OUT/pmc_sq > $OUT/summary.log 2>&1 || exit 1
cp profiles/${TAG}_kernel_stats.csv profiles/${TAG}_pmc_traffic.json $OUT/
cp $OUT/by_kernel.txt profiles/${TAG}_by_kernel.txt
rm -rf $OUT/stats $OUT/pmc_r $OUT/pmc_w $OUT/pmc_sq
cat $OUT/summary.log; cat $OUT/by_kernel.txt

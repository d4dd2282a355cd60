#!/bin/bash
# One profiling run of the headline bench on the GPU box: bench line, rocprofv3 kernel stats, PMC traffic passes.
#   bash tools/profile_round.sh <out_dir under gpurun_out/> [extra bench args]
# rocprofv3 gets the program itself after `--` (python3 bench.py ...), counters and tracing in separate passes.
set -o pipefail
OUT=gpurun_out/$1; shift
EXTRA="$@"
mkdir -p $OUT
export TMPDIR=/tmp
python3 bench.py --steps 20 --warmup 5 --no-secondary $EXTRA > $OUT/bench.json 2> $OUT/bench.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-kernel-timers --no-secondary --repeats 1 $EXTRA > $OUT/stats.json 2> $OUT/stats.err || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_r -- python3 bench.py --steps 1 --warmup 2 --no-cpu-baseline --no-kernel-timers --no-secondary --repeats 1 $EXTRA > /dev/null 2> $OUT/pmc_r.err || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_w -- python3 bench.py --steps 1 --warmup 2 --no-cpu-baseline --no-kernel-timers --no-secondary --repeats 1 $EXTRA > /dev/null 2> $OUT/pmc_w.err || exit 1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -- python3 bench.py --steps 1 --warmup 2 --no-cpu-baseline --no-kernel-timers --no-secondary --repeats 1 $EXTRA > /dev/null 2> $OUT/pmc_sq.err || exit 1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmc_lds -- python3 bench.py --steps 1 --warmup 2 --no-cpu-baseline --no-kernel-timers --no-secondary --repeats 1 $EXTRA > /dev/null 2> $OUT/pmc_lds.err || exit 1
python3 tools/lds_conflicts.py $OUT/pmc_lds profiles/$(basename $OUT)_lds_conflicts.json > $OUT/lds.log 2>&1 || exit 1
cp profiles/$(basename $OUT)_lds_conflicts.json $OUT/
# keep the merged-back output small: the raw per-dispatch CSVs are summarised here, on the box
python3 tools/profile_summarise.py $(basename $OUT) $OUT/stats $OUT/pmc_r $OUT/pmc_w $OUT/pmc_sq > $OUT/summary.log 2>&1 || exit 1
cp profiles/$(basename $OUT)_kernel_stats.csv profiles/$(basename $OUT)_pmc_traffic.json $OUT/
rm -rf $OUT/stats $OUT/pmc_r $OUT/pmc_w $OUT/pmc_sq $OUT/pmc_lds
cat $OUT/summary.log $OUT/lds.log

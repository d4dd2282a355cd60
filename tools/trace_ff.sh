#!/bin/bash
# Kernel timeline of the headline bench: per-kernel totals and the device idle time inside two steady-state updates.
#   bash tools/trace_ff.sh <tag> [extra bench args]
set -o pipefail
TAG=$1; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-kernel-timers --no-secondary --repeats 1 "$@" > $OUT/stats.json 2> $OUT/stats.err || exit 1
python3 tools/trace_by_kernel.py $OUT/stats 0.01 rollout_h2 > $OUT/by_kernel.txt || exit 1
rm -rf $OUT/stats
cat $OUT/stats.json | cut -c1-200; cat $OUT/by_kernel.txt

#!/bin/bash
# kernel trace (stats) of the default bench command, short: tools/scratch/kt6.sh [extra bench args]
export TMPDIR=/tmp
ROOT=$(pwd); OUT=gpurun_out/kt6; mkdir -p $OUT
(cd /tmp && rocprofv3 --kernel-trace --stats -d "$ROOT/$OUT/kt" -o run --output-format csv -- python3 "$ROOT/bench.py" --no-build --no-cpu-baseline --no-host-leg --no-from-bam --no-also --steps 6 --warmup 2 "$@" > "$ROOT/$OUT/bench.json" 2> "$ROOT/$OUT/bench.err")
cp $OUT/kt/run_kernel_stats.csv $OUT/kernel_stats.csv
head -30 $OUT/kernel_stats.csv | cut -c1-200
rm -rf $OUT/kt

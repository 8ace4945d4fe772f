#!/bin/bash
# A/B: windows of the consensus rounds from the break-round table (default) vs the walk over the positions every round (SPX_WINDOW_TABLE=2,0: the
# device takes the parameters from the host's logic_params).  Every result is APPENDED to gpurun_out/r06/flank.txt as it comes (the first attempt
# printed at the end and was cut off by the budget).
OUT=gpurun_out/r06; mkdir -p $OUT
run() { # platform, table setting ("" = default), lanes
  if [ -n "$2" ]; then export SPX_WINDOW_TABLE=$2; else unset SPX_WINDOW_TABLE; fi
  if [ -n "$3" ]; then export SPX_PREP_LANES=$3; else unset SPX_PREP_LANES; fi
  python3 bench.py --platform $1 --no-from-bam --no-host-leg --no-cpu-baseline --no-also --no-build --steps 16 --warmup 3 --verify 64 --guard-exposure 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 table=${2:-default}', d['value'], d['ms_per_step'], d['kernel_ms_per_step'])" | tee -a $OUT/flank.txt
}
for r in 1 2; do
  run mixed "" 6; run mixed "2,0" 6; run mixed "" ""; run mixed "2,0" ""
done
run ont "" ""; run ont "2,0" ""
run hifi "" ""; run hifi "2,0" ""

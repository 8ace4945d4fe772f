#!/bin/bash
# A/B: slices alternating between two scratch areas, MAP on the last side stream beside the next slice's DP kernels (SPX_DP_PINGPONG=1) vs one area
OUT=gpurun_out/r06; mkdir -p $OUT
run() { # label platform steps env...
  L=$1; P=$2; S=$3; shift 3
  env "$@" python3 bench.py --platform $P --no-from-bam --no-host-leg --no-cpu-baseline --no-also --no-build --steps $S --warmup 3 --verify 64 --guard-exposure 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$P $L', d['value'], d['ms_per_step'], d['kernel_ms_per_step'], d['config'].get('verified_timed_groups'))" | tee -a $OUT/pingpong2.txt
}
run "one area" hifi 12 A=1
run "pingpong 2x12" hifi 12 SPX_DP_PINGPONG=1
run "pingpong 2x8" hifi 12 SPX_DP_PINGPONG=1 SPX_DP_SLICE_GB=8
run "one area" hifi 12 A=1
run "pingpong 2x12" hifi 12 SPX_DP_PINGPONG=1
run "one area" ont 6 A=1
run "pingpong" ont 6 SPX_DP_PINGPONG=1
run "one area" mixed 8 A=1
run "pingpong" mixed 8 SPX_DP_PINGPONG=1

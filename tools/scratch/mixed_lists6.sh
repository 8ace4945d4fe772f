#!/bin/bash
# mixed workload: list size (groups per step) against the fixed latency of the heaviest groups' single-lane walks
OUT=gpurun_out/r06; mkdir -p $OUT
run() { # label gps distinct depth lanes steps env...
  L=$1; G=$2; D=$3; P=$4; N=$5; S=$6; shift 6
  env SPX_PREP_LANES=$N "$@" python3 bench.py --platform mixed --groups-per-step $G --distinct $D --depth $P --no-from-bam --no-host-leg --no-cpu-baseline --no-also --no-build --steps $S --warmup 3 --verify 64 --guard-exposure 0 2>/tmp/ml.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('mixed $L gps $G depth $P lanes $N:', d['value'], d['ms_per_step'], d['kernel_ms_per_step'], d['dp_tiers']['on'])" | tee -a $OUT/mixed_lists.txt || tail -3 /tmp/ml.err
}
run "default" 16384 9 8 6 24 A=1
run "default tiers off" 16384 9 8 6 24 SPX_DP_TIERS=0
run "64k" 65536 3 4 4 8 A=1
run "64k lanes 6" 65536 3 4 6 8 A=1
run "64k tiers off" 65536 3 4 4 8 SPX_DP_TIERS=0
run "32k depth 5" 32768 5 5 6 12 A=1

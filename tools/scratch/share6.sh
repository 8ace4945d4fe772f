#!/bin/bash
# share of a list's cells from which a class takes the fast tier: mixed (65 536-group lists) and HiFi
OUT=gpurun_out/r06; mkdir -p $OUT
run() { # label platform gps distinct depth lanes steps env...
  L=$1; PF=$2; G=$3; D=$4; P=$5; N=$6; S=$7; shift 7
  env SPX_PREP_LANES=$N "$@" python3 bench.py --platform $PF --groups-per-step $G --distinct $D --depth $P --no-from-bam --no-host-leg --no-cpu-baseline --no-also --no-build --steps $S --warmup 3 --verify 64 --guard-exposure 0 2>/tmp/ml.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$PF $L:', d['value'], d['ms_per_step'], d['kernel_ms_per_step'], d['dp_tiers'])" | tee -a $OUT/share.txt
}
for s in 2 1 0; do run "share $s" mixed 65536 3 4 4 8 SPX_FAST_MIN_SHARE=$s; done
for s in 2 1 0; do run "share $s" hifi 131072 8 3 4 12 SPX_FAST_MIN_SHARE=$s; done
run "share 0" ont 16384 8 4 4 6 SPX_FAST_MIN_SHARE=0
run "share 2" ont 16384 8 4 4 6 SPX_FAST_MIN_SHARE=2

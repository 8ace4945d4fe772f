#!/bin/bash
# side streams for the band classes: 3 (default) vs 6 / 8, hardware queues 14 vs 20
OUT=gpurun_out/r05; mkdir -p $OUT
run() { # platform, side, queues
  export SPX_SIDE_STREAMS=$2 GPU_MAX_HW_QUEUES=$3
  python3 bench.py --platform $1 --no-from-bam --no-host-leg --no-cpu-baseline --no-also --no-build --steps 12 --warmup 3 --verify 64 --guard-exposure 0 > $OUT/sd_$1_$2_$3.json 2>$OUT/sd.err
  python3 -c "
import json
d=json.loads(open('$OUT/sd_$1_$2_$3.json').read().strip().splitlines()[-1]); print('$1 side $2 queues $3', d['value'], d['ms_per_step'], d['kernel_ms_per_step'])" || tail -3 $OUT/sd.err
}
for r in 1 2; do
 for plat in mixed hifi ont; do
  run $plat 3 14; run $plat 6 14; run $plat 6 20; run $plat 8 24
 done
done

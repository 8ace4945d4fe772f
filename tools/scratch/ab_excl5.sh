#!/bin/bash
OUT=gpurun_out/r05; mkdir -p $OUT
for plat in mixed hifi; do
 for v in 0 16 32 0 16 32; do
  SPX_DP_EXCLUDE_CUS=$v python3 bench.py --platform $plat --no-from-bam --no-host-leg --no-cpu-baseline --no-also --steps 12 --warmup 3 --verify 64 --guard-exposure 0 > $OUT/excl_${plat}_$v.json 2>$OUT/excl.err
  python3 -c "
import json
d=json.loads(open('$OUT/excl_${plat}_$v.json').read().strip().splitlines()[-1]); print('$plat dp_exclude=$v', d['value'], d['ms_per_step'], d['kernel_ms_per_step'])"
 done
done

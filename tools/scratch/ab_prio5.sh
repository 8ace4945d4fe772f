#!/bin/bash
OUT=gpurun_out/r05; mkdir -p $OUT
for plat in mixed hifi ont; do
 for v in cur noprio cur noprio; do
  if [ $v = noprio ]; then export SPX_LIB=$PWD/ab/noprio/libspx.so; else unset SPX_LIB; fi
  python3 bench.py --platform $plat --no-from-bam --no-host-leg --no-cpu-baseline --no-also --no-build --steps 12 --warmup 3 --verify 64 --guard-exposure 0 > $OUT/prio_${plat}_$v.json 2>$OUT/prio.err
  python3 -c "
import json
d=json.loads(open('$OUT/prio_${plat}_$v.json').read().strip().splitlines()[-1]); print('$plat $v', d['value'], d['ms_per_step'], d['kernel_ms_per_step'])"
 done
done

#!/bin/bash
# A/B on one box: band class 14 = (2,28) for W 49..55 (default) vs the build before it (ab/cls13: those widths in (4,16))
OUT=gpurun_out/r05; mkdir -p $OUT
timeout 900 python3 -m pytest tests -x -q -m gpu -k "band or probaln or batch_ or posterior or mixed" 2>&1 | tail -3
for plat in mixed hifi; do
 for v in cur old cur old; do
  if [ $v = old ]; then export SPX_LIB=$PWD/ab/cls13/libspx.so; else unset SPX_LIB; fi
  python3 bench.py --platform $plat --no-from-bam --no-host-leg --no-cpu-baseline --no-also --no-build --steps 12 --warmup 3 --verify 64 --guard-exposure 0 > $OUT/cls_${plat}_$v.json 2>$OUT/cls.err
  python3 -c "
import json
d=json.loads(open('$OUT/cls_${plat}_$v.json').read().strip().splitlines()[-1]); print('$plat $v', d['value'], d['ms_per_step'], d['kernel_ms_per_step'])"
 done
done

#!/bin/bash
# A/B on one box: score / result kernels on the result stream (default) vs on the main stream (SPX_SCORE_MAIN=1)
OUT=gpurun_out/r05; mkdir -p $OUT
timeout 900 python3 -m pytest tests -x -q -m gpu -k "batch_ or pipeline or command_line or dist or quality or write_bam or probaln" 2>&1 | tail -3
for plat in mixed hifi ont; do
 for v in cur main cur main; do
  if [ $v = main ]; then export SPX_SCORE_MAIN=1; else unset SPX_SCORE_MAIN; fi
  python3 bench.py --platform $plat --no-from-bam --no-host-leg --no-cpu-baseline --no-also --no-build --steps 12 --warmup 3 --verify 64 --guard-exposure 0 > $OUT/tail_${plat}_$v.json 2>$OUT/tail.err
  python3 -c "
import json
d=json.loads(open('$OUT/tail_${plat}_$v.json').read().strip().splitlines()[-1]); print('$plat $v', d['value'], d['ms_per_step'], d['kernel_ms_per_step'])"
 done
done

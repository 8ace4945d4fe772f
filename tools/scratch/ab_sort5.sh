#!/bin/bash
OUT=gpurun_out/r05; mkdir -p $OUT
python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "work_list or md_tagged or preparation_table or batch_ or trim" 2>&1 | tail -4
for plat in mixed hifi ont; do
 for v in 1 0 1 0; do
  SPX_PREP_SORT=$v python3 bench.py --platform $plat --no-from-bam --no-host-leg --no-cpu-baseline --no-also --steps 12 --warmup 3 --verify 64 > $OUT/sort_${plat}_$v.json 2>$OUT/sort.err
  python3 -c "
import json
d=json.loads(open('$OUT/sort_${plat}_$v.json').read().strip().splitlines()[-1]); print('$plat sort=$v', d['value'], d['ms_per_step'], d['kernel_ms_per_step'])"
 done
done

#!/bin/bash
OUT=gpurun_out/r05; mkdir -p $OUT
python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "work_list or md_tagged or preparation_table or batch_ or trim" 2>&1 | tail -4
for plat in mixed hifi ont; do
 for v in 2048 0 2048 0; do
  SPX_PREP_HEAVY=$v python3 bench.py --platform $plat --no-from-bam --no-host-leg --no-cpu-baseline --no-also --steps 12 --warmup 3 --verify 64 --guard-exposure 0 > $OUT/heavy_${plat}_$v.json 2>$OUT/heavy.err
  python3 -c "
import json
d=json.loads(open('$OUT/heavy_${plat}_$v.json').read().strip().splitlines()[-1]); print('$plat heavy=$v', d['value'], d['ms_per_step'], d['kernel_ms_per_step'])"
 done
done
for v in 512 8192; do
  SPX_PREP_HEAVY=$v python3 bench.py --platform mixed --no-from-bam --no-host-leg --no-cpu-baseline --no-also --steps 12 --warmup 3 --verify 64 > $OUT/heavy_mixed_$v.json 2>$OUT/heavy.err
  python3 -c "
import json
d=json.loads(open('$OUT/heavy_mixed_$v.json').read().strip().splitlines()[-1]); print('mixed heavy=$v', d['value'], d['ms_per_step'], d['kernel_ms_per_step'])"
done

#!/bin/bash
# A/B on one box: slices alternate between two scratch areas (MAP kernels beside the next slice's DP kernels) vs one area
OUT=gpurun_out/r05; mkdir -p $OUT
timeout 900 python3 -m pytest tests -x -q -m gpu -k "batch_ or pipeline or command_line or dist or quality or write_bam or probaln or slice" 2>&1 | tail -3
for plat in hifi ont mixed; do
 for v in two one two32 two one two32; do
  unset SPX_DP_SCRATCH_COPIES SPX_DP_SLICE_GB
  if [ $v = one ]; then export SPX_DP_SCRATCH_COPIES=1; fi
  if [ $v = two32 ]; then export SPX_DP_SLICE_GB=32; fi
  if [ $plat = mixed ] && [ $v = two32 ]; then continue; fi
  python3 bench.py --platform $plat --no-from-bam --no-host-leg --no-cpu-baseline --no-also --no-build --steps 12 --warmup 3 --verify 64 --guard-exposure 0 > $OUT/cp_${plat}_$v.json 2>$OUT/cp.err
  python3 -c "
import json
d=json.loads(open('$OUT/cp_${plat}_$v.json').read().strip().splitlines()[-1]); print('$plat $v', d['value'], d['ms_per_step'], d['kernel_ms_per_step'], d.get('dp_slices'), d['roofline'].get('phase',{}).get('frac'))"
 done
done

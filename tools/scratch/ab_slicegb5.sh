#!/bin/bash
# scratch budget per DP slice (SPX_DP_SLICE_GB, default 16): fewer, larger slices
OUT=gpurun_out/r05; mkdir -p $OUT
timeout 600 python3 -m pytest tests -x -q -m gpu -k "slice or batch_ or pipeline or quality" 2>&1 | tail -3
run() { # platform, GB
  export SPX_DP_SLICE_GB=$2
  python3 bench.py --platform $1 --no-from-bam --no-host-leg --no-cpu-baseline --no-also --no-build --steps 12 --warmup 3 --verify 64 --guard-exposure 0 > $OUT/sg_$1_$2.json 2>$OUT/sg.err
  python3 -c "
import json
d=json.loads(open('$OUT/sg_$1_$2.json').read().strip().splitlines()[-1]); print('$1 $2', d['value'], d['ms_per_step'], d['kernel_ms_per_step'], d['roofline'].get('kernel_launches_per_step'))" || tail -3 $OUT/sg.err
}
for r in 1 2; do
 for gb in 16 32 64 8; do run mixed $gb; done
 for gb in 16 32 48; do run ont $gb; done
 for gb in 16 24; do run hifi $gb; done
done

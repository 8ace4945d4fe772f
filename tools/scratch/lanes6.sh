#!/bin/bash
for r in 1 2; do
for cfg in "4 6 7 14" "6 8 9 16" "8 10 11 18" "5 7 8 15"; do set -- $cfg
  GPU_MAX_HW_QUEUES=$4 SPX_PREP_LANES=$1 python3 bench.py --platform mixed --depth $2 --distinct $3 --no-from-bam --no-host-leg --no-cpu-baseline --no-also --no-build --steps 14 --warmup 4 --verify 64 --guard-exposure 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('mixed lanes=$1 depth=$2 queues=$4', d['value'], d['ms_per_step'], d['kernel_ms_per_step'])"
done
done

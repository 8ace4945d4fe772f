#!/bin/bash
for cfg in "4 6 7" "6 8 9" "8 10 11" "3 4 5"; do set -- $cfg
  SPX_PREP_LANES=$1 python3 bench.py --platform mixed --depth $2 --distinct $3 --no-from-bam --no-host-leg --no-cpu-baseline --no-also --steps 14 --warmup 4 --verify 64 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('mixed lanes=$1 depth=$2', d['value'], d['ms_per_step'], d['kernel_ms_per_step'])"
done

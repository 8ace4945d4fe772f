#!/bin/bash
run() {
  SPX_PREP_LANES=$1 SPX_SIDE_STREAMS=$2 python3 bench.py --platform ont --steps 8 --warmup 2 --no-also --no-host-leg --no-build --verify 64 --no-cpu-baseline --no-from-bam --distinct $3 --depth 4 --guard-exposure 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ont lanes=$1 side=$2 distinct=$3', d['value'], d['ms_per_step'], d['kernel_ms_per_step'])"
}
run 6 6 8
run 4 6 8
run 6 3 8
run 4 3 8
run 5 6 8

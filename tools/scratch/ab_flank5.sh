#!/bin/bash
# A/B: windows of the consensus rounds from the break-round table (default) vs the walk over the positions every round (ab/prev)
export SPX_PREP_LANES=6
run() {
  if [ $2 = old ]; then export SPX_LIB=$PWD/ab/prev/libspx.so; else unset SPX_LIB; fi
  python3 bench.py --platform $1 --no-from-bam --no-host-leg --no-cpu-baseline --no-also --no-build --steps 16 --warmup 3 --verify 64 --guard-exposure 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 $2', d['value'], d['ms_per_step'], d['kernel_ms_per_step'])"
}
run mixed new; run mixed old; run mixed new; run mixed old
unset SPX_PREP_LANES
run hifi new; run hifi old

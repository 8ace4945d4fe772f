#!/bin/bash
run() {
  python3 bench.py --platform ont --steps 8 --warmup $1 --no-also --no-host-leg --no-build --verify 64 --cpu-runs 3 --cpu-threads 32 --no-host-input-leg --from-bam 131072 --distinct 8 --depth 4 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ont exact also-leg command, warmup $1:', d['value'], d['ms_per_step'], d['kernel_ms_per_step'], (d.get('from_bam') or {}).get('loop_groups_per_s'))"
}
run 2
run 2

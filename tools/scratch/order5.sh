#!/bin/bash
for plat in mixed hifi ont; do for v in 0 1 0 1; do
  SPX_PIPE_LAUNCH_IN_ORDER=$v python3 bench.py --platform $plat --no-from-bam --no-host-leg --no-cpu-baseline --no-also --steps 14 --warmup 4 --verify 256 --guard-exposure 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$plat in_order=$v', d['value'], d['ms_per_step'], d['kernel_ms_per_step'], d['config']['verified_timed_groups'], d['config']['verified_own_relabel_list']['oracle_list_is_byte_prefix_of_this_runs_list'])"
done; done

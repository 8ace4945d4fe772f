#!/bin/bash
OUT=gpurun_out/r05; mkdir -p $OUT
python3 tools/e2e_cli.py --platform mixed --groups 114688 --threads 64 --sweep SPX_DEPTH=3,5,7 > $OUT/e2e_mixed_depth.json 2> $OUT/e2e_mixed_depth.err
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r05/e2e_mixed_depth.json'))
print({k:d.get(k) for k in ('groups','wall_s','loop_s','groups_per_s','loop_groups_per_s','relabel_list_prefix_identical')})
print(json.dumps(d.get('sweep'),indent=0)[:1500])
PY
for plat in mixed hifi ont; do
  python3 bench.py --platform $plat --no-from-bam --no-host-leg --no-cpu-baseline --no-also --no-build --steps 16 --warmup 3 --verify 64 --guard-exposure 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$plat', d['value'], d['ms_per_step'], d['kernel_ms_per_step'])"
done

#!/bin/bash
# kernel-trace stats of a short bench run: preparation kernels' average durations
export TMPDIR=/tmp
PLAT=${1:-mixed}
shift
cd /tmp; rm -rf /tmp/ktp
rocprofv3 --kernel-trace --stats -d /tmp/ktp -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --platform $PLAT --steps 4 --warmup 2 --no-build --no-cpu-baseline --no-host-leg --no-from-bam --no-also --depth 1 --verify 0 "$@" 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('bench', d['value'], d['ms_per_step'], d['kernel_ms_per_step'])"
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open('/tmp/ktp/run_kernel_stats.csv')))
for r in rows:
    n=r['Name']
    if any(k in n for k in ('aln_','group_','recode','rows_unpack')):
        print("%-60s calls %4s avg %9.3f ms total %9.3f ms"%(n[:60], r['Calls'], float(r['AverageNs'])/1e6, float(r['TotalDurationNs'])/1e6))
PY

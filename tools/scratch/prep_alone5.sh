#!/bin/bash
# prep kernels of ONE mixed list alone on the chip (kernel-only bench: one prepare, then DP replays), with and without heavy extraction
export TMPDIR=/tmp
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r05; mkdir -p $OUT
for v in "$@"; do
  (cd /tmp && SPX_PREP_HEAVY=$v rocprofv3 --kernel-trace --stats -d $OUT/kt_$v -o run --output-format csv -- python3 $ROOT/bench.py --platform mixed --kernel-only --no-build --no-cpu-baseline --steps 1 --warmup 0 --verify 0 > $OUT/prep_alone_$v.json 2> $OUT/prep_alone_$v.err)
  echo "== SPX_PREP_HEAVY=$v"
  python3 - $OUT/kt_$v/run_kernel_stats.csv <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
keep=[r for r in rows if any(k in r['Name'] for k in ('aln_','group_','heavy','rocprim','recode','plan_','scan','order','rows_unpack','problem_const'))]
keep.sort(key=lambda r:-float(r['TotalDurationNs']))
tot=0
for r in keep[:22]:
    print(f"{float(r['TotalDurationNs'])/1e6:9.2f} ms {r['Calls']:>4} calls  {r['Name'][:70]}")
for r in keep: tot+=float(r['TotalDurationNs'])
print('prep total ms', tot/1e6)
PY
  cp $OUT/kt_$v/run_kernel_stats.csv $OUT/prep_alone_stats_$v.csv; rm -rf $OUT/kt_$v
done

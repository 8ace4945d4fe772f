#!/bin/bash
# round 5: kernel trace of the mixed bench + CU-mask experiments
export TMPDIR=/tmp
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r05
mkdir -p $OUT
python3 bench.py --platform mixed --no-from-bam --no-host-leg --no-cpu-baseline --steps 12 --warmup 3 > $OUT/mixed_base.json 2> $OUT/mixed_base.err
for cus in 32 64; do
  SPX_PREP_CUS=$cus python3 bench.py --platform mixed --no-from-bam --no-host-leg --no-cpu-baseline --steps 12 --warmup 3 > $OUT/mixed_cus$cus.json 2> $OUT/mixed_cus$cus.err
done
(cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/kt -o run --output-format csv -- python3 $ROOT/bench.py --platform mixed --no-build --no-cpu-baseline --no-host-leg --no-from-bam --no-also --steps 8 --warmup 2 --verify 0 > $OUT/mixed_under_rocprof.json 2> $OUT/mixed_under_rocprof.err)
cp $OUT/kt/run_kernel_stats.csv $OUT/mixed_kernel_stats.csv
python3 - $OUT <<'PY'
import csv, sys, collections, json
out = sys.argv[1]
rows = list(csv.DictReader(open(f"{out}/kt/run_kernel_trace.csv")))
print("columns", list(rows[0].keys()))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60], r.get("Queue_Id") or r.get("Stream_Id") or "") for r in rows]
ev.sort()
t0, t1 = ev[0][0], max(e[1] for e in ev)
# take the last 40 % of the run (timed steps), compute busy time (union of intervals) and per-kernel-name union
lo = t0 + int(0.6 * (t1 - t0))
sel = [e for e in ev if e[0] >= lo]
def union(iv):
    iv = sorted(iv); tot = 0; cs, ce = iv[0]
    for s, e in iv[1:]:
        if s > ce: tot += ce - cs; cs, ce = s, e
        else: ce = max(ce, e)
    return tot + ce - cs
span = max(e[1] for e in sel) - sel[0][0]
print("window ms", span / 1e6, "any-kernel busy ms", union([(s, e) for s, e, _, _ in sel]) / 1e6)
dp = [(s, e) for s, e, n, _ in sel if "baq_" in n or "map_kernel" in n]
print("DP-kernel union ms", union(dp) / 1e6, "sum ms", sum(e - s for s, e in dp) / 1e6)
byname = collections.defaultdict(list)
for s, e, n, q in sel: byname[n].append((s, e))
tab = sorted(((union(v) / 1e6, sum(e - s for s, e in v) / 1e6, len(v), k) for k, v in byname.items()), reverse=True)
for u, sm, c, k in tab[:30]: print(f"{u:9.2f} union ms {sm:9.2f} sum ms {c:5d} calls  {k}")
json.dump({"window_ms": span / 1e6, "busy_ms": union([(s, e) for s, e, _, _ in sel]) / 1e6, "dp_union_ms": union(dp) / 1e6,
           "by_kernel": [{"kernel": k, "union_ms": u, "sum_ms": sm, "calls": c} for u, sm, c, k in tab]}, open(f"{out}/mixed_trace_summary.json", "w"), indent=1)
PY
gzip -c $OUT/kt/run_kernel_trace.csv > $OUT/mixed_kernel_trace.csv.gz
rm -rf $OUT/kt
for f in mixed_base mixed_cus32 mixed_cus64 mixed_under_rocprof; do python3 -c "
import json,sys
d=json.loads(open('$OUT/$f.json').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'], d['kernel_ms_per_step'])"; done
